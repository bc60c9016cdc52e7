// sequencer_module.cpp -- the `tksm sequence` module on top of the C-ABI.
//
// Mirrors (file:line into vpc-ccg/tksm):
//   Sequencer_module::impl::run     src/sequence.cpp:30-54   ($TKSM_MODELS handling, then the sequencer)
//   parse_args                      py/sequence.py:34-165    (flags, defaults, validation texts, exit codes)
//   main block                      py/sequence.py:323-376   (load reference + models, stream the MDF, write)
//   get_output_file                 py/sequence.py:291-300   (extension decides FASTQ/FASTA; .gz ok)
//   utility flags                   src/module.h:75-104      (-s/--seed default 42, --verbosity, --log-file)
//   worker pool                     py/sequence.py:354-366   (multiprocessing.Pool + imap_unordered over molecules)
// Streaming: the reader cuts the MDF text into batches of whole molecules and numbers their reads; two parser threads per
// device group (contexts of their own) turn the text into device batches ahead of time; --in-flight worker threads (one
// context each, sharing the packed reference and the model tables) run and download a batch each, so that parsing, the
// device work of consecutive batches, the copies and the writes overlap.  Regular output files are written by the workers
// themselves at their final offsets (pwrite, MDF order); pipes and devices by one writer that puts the record streams back
// into MDF order.
// Exit codes: 0 ok; 1 for `sys.exit("msg")`-style validation and runtime errors; 2 for argparse
// usage errors (missing -i, neither -o nor --perfect) -- what the embedded interpreter returns.
#include "sequencer_module.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tksmseq.h"

namespace {

struct Args {
    std::string input, badread, perfect, output_format, identity = "84.0,99.0,5.5";
    std::string error_model, qscore_model, tail_model = "no_noise";      // "": nanopore2020 if discoverable, else random
    std::vector<std::string> references;
    bool skip_qual = false, list = false, help = false;
    int threads = 1, in_flight = 3;
    std::vector<int> devices{0};
    long long seed = 42;
    uint64_t batch_bytes = 64ull << 20;
    std::string verbosity = "INFO", log_file = "stderr";
};

const char* OPTION_DESTS[] = {"help", "input", "references", "badread", "perfect", "skip_qual_compute", "output_format",
                              "threads", "badread_identity", "badread_error_model", "badread_qscore_model",
                              "badread_tail_model", "list", "seed", "devices", "verbosity", "log_file"};

void usage(FILE* f) {
    fprintf(f,
            "usage: sequence [-h] -i INPUT [-r REFERENCES [REFERENCES ...]] [-o BADREAD] [--perfect PERFECT]\n"
            "                [--skip-qual-compute] [-O {fastq,fasta}] [-t THREADS] [--badread-identity BADREAD_IDENTITY]\n"
            "                [--badread-error-model M] [--badread-qscore-model M] [--badread-tail-model M] [--list]\n"
            "                [-s SEED] [--devices D[,D...]] [--batch-bytes B] [--in-flight N] [--verbosity L] [--log-file F]\n");
}

// one gzip member (RFC 1952) holding d[0..n): members simply follow each other in a .gz file, so batches -- and pieces of
// a batch -- are compressed independently, on the worker threads, and the writer appends bytes
bool gzip_member(const uint8_t* d, size_t n, std::vector<uint8_t>& out) {
    z_stream z{};
    if (deflateInit2(&z, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&z, (uLong)n) + 64);
    z.next_in = const_cast<Bytef*>(d); z.avail_in = (uInt)n;
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    const int rc = deflate(&z, Z_FINISH);
    out.resize(rc == Z_STREAM_END ? z.total_out : 0);
    deflateEnd(&z);
    return rc == Z_STREAM_END;
}

struct Writer {
    int fd = -1; bool gz = false, fastq = false, wrote = false;
    bool positional = false;                             // a regular file: the workers write their batches at their offsets (pwrite)
    bool open(const std::string& path) {                 // get_output_file, py/sequence.py:291-300
        std::string p = path;
        if (p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0) { gz = true; p.resize(p.size() - 3); }
        fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) return false;
        struct stat st;
        positional = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
        auto ends = [&](const char* s) { size_t n = strlen(s); return p.size() >= n && p.compare(p.size() - n, n, s) == 0; };
        fastq = ends(".fastq") || ends(".fq");
        return true;
    }
    bool write(const uint8_t* d, size_t n) {             // appends; for .gz outputs the bytes are finished gzip members
        wrote = wrote || n;
        while (n) {
            const ssize_t w = ::write(fd, d, std::min<size_t>(n, (size_t)1 << 30));
            if (w < 0) { if (errno == EINTR) continue; return false; }
            d += w; n -= (size_t)w;
        }
        return true;
    }
    bool write_at(const uint8_t* d, size_t n, uint64_t off) const {
        while (n) {
            const ssize_t w = ::pwrite(fd, d, std::min<size_t>(n, (size_t)1 << 30), (off_t)off);
            if (w < 0) { if (errno == EINTR) continue; return false; }
            d += w; n -= (size_t)w; off += (uint64_t)w;
        }
        return true;
    }
    bool close() {                                       // false: the last bytes could not be written
        if (fd < 0) return true;
        bool ok = true;
        if (gz && !wrote) { std::vector<uint8_t> e; if (gzip_member(nullptr, 0, e)) ok = write(e.data(), e.size()); }   // a valid empty .gz
        ok = ::close(fd) == 0 && ok;
        fd = -1;
        return ok;
    }
};

// --verbosity / --log-file (src/module.h:95-122, src/util.h:94-120): levels DEBUG < INFO < WARN < ERROR < OFF; the file is
// "stderr", "stdout" or a path.  What the reference's Python prints unconditionally (model loading progress, "Loading
// reference") stays unconditional; the module's own diagnostics go through here.
struct Logger {
    enum Level { DEBUG = 0, INFO = 1, WARN = 2, ERROR = 3, OFF = 4 };
    int level = INFO; FILE* f = stderr; bool own = false; std::mutex m;
    static int parse(const std::string& v) {
        static const char* names[] = {"DEBUG", "INFO", "WARN", "ERROR", "OFF"};
        for (int i = 0; i < 5; i++) if (v == names[i]) return i;
        return -1;
    }
    bool open(const std::string& path) {
        if (path == "stderr") { f = stderr; return true; }
        if (path == "stdout") { f = stdout; return true; }
        FILE* g = fopen(path.c_str(), "a");
        if (!g) return false;
        f = g; own = true;
        return true;
    }
    void log(int lv, const char* fmt, ...) __attribute__((format(printf, 3, 4))) {
        if (lv < level || level == OFF) return;
        static const char* tag[] = {"DBG", "INF", "WRN", "ERR"};
        std::lock_guard<std::mutex> l(m);
        fprintf(f, "[sequence %s] ", tag[lv]);
        va_list ap; va_start(ap, fmt); vfprintf(f, fmt, ap); va_end(ap);
        fputc('\n', f); fflush(f);
    }
    ~Logger() { if (own) fclose(f); }
};

// reads a batch of MDF text will produce: the depth column of every molecule header (mdf_generator, py/sequence.py:206-213)
uint64_t count_reads(const char* p, size_t len) {
    uint64_t n = 0;
    const char* end = p + len;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (*p == '+') {
            const char* t = (const char*)memchr(p, '\t', (size_t)(le - p));
            long long d = 0;
            if (t) { const char* q = t + 1; bool neg = false; if (q < le && *q == '-') { neg = true; q++; } while (q < le && *q >= '0' && *q <= '9') d = d * 10 + (*q++ - '0'); if (neg) d = 0; }
            n += (uint64_t)d;
        }
        p = nl ? nl + 1 : end;
    }
    return n;
}

struct Chunk { uint64_t seq = 0, first_read = 0, n_reads = 0; std::vector<char> text; };

struct ChunkQueue {                                       // bounded, closed by the reader at end of input
    std::mutex m; std::condition_variable cv_put, cv_get; std::deque<Chunk> q; size_t cap = 2; bool closed = false;
    void push(Chunk&& c) { std::unique_lock<std::mutex> l(m); cv_put.wait(l, [&] { return q.size() < cap || closed; }); if (closed) return; q.push_back(std::move(c)); cv_get.notify_one(); }
    bool pop(Chunk& c) { std::unique_lock<std::mutex> l(m); cv_get.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; c = std::move(q.front()); q.pop_front(); cv_put.notify_one(); return true; }
    void close() { std::lock_guard<std::mutex> l(m); closed = true; cv_get.notify_all(); cv_put.notify_all(); }
};

// a parsed batch on its way from a parser thread to a worker of the same device group
struct Parsed { uint64_t seq = 0, first_read = 0, n_reads = 0; tksmseq_batch* b = nullptr; };
struct ParsedQueue {
    std::mutex m; std::condition_variable cv_put, cv_get; std::deque<Parsed> q; size_t cap = 2; bool closed = false;
    // a group's parsers hand their batches over in the order in which they took the chunks (a worker that ran a later batch first
    // would wait for the earlier one's place in the output while that one waits for a worker)
    std::mutex take_m; uint64_t taken = 0;
    std::mutex order_m; std::condition_variable order_cv; uint64_t handed = 0;
    bool push(const Parsed& c) { std::unique_lock<std::mutex> l(m); cv_put.wait(l, [&] { return q.size() < cap || closed; }); if (closed) return false; q.push_back(c); cv_get.notify_one(); return true; }
    bool pop(Parsed& c) { std::unique_lock<std::mutex> l(m); cv_get.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; c = q.front(); q.pop_front(); cv_put.notify_one(); return true; }
    void close() { std::lock_guard<std::mutex> l(m); closed = true; cv_get.notify_all(); cv_put.notify_all(); }       // (what is queued is still handed out)
};

struct Worker {                                           // one batch in flight: context + page-locked record buffers
    tksmseq_ctx* ctx = nullptr;
    uint8_t* host[2] = {nullptr, nullptr}; uint64_t host_cap[2] = {0, 0};
    std::vector<uint8_t> packed[2];                     // .gz outputs: the batch as gzip members
    std::mutex m; std::condition_variable cv; bool host_busy = false;     // the writer still reads the buffers
    // regular uncompressed files: the records pass through two page-locked pieces (one being written while the next arrives)
    uint64_t piece = 64ull << 20;                       // (TKSMSEQ_PIECE_BYTES: small pieces for the tests)
    uint8_t* ring[2] = {nullptr, nullptr};
    bool ring_ready() {
        for (int q = 0; q < 2; q++)
            if (!ring[q]) { void* p = nullptr; if (tksmseq_host_alloc(piece, &p)) return false; ring[q] = (uint8_t*)p; }
        return true;
    }
    bool reserve(int k, uint64_t bytes) {
        if (bytes <= host_cap[k]) return true;
        tksmseq_host_free(host[k]); host[k] = nullptr; host_cap[k] = 0;
        const uint64_t want = bytes + bytes / 4 + 4096;
        void* p = nullptr;
        if (tksmseq_host_alloc(want, &p)) return false;
        host[k] = (uint8_t*)p; host_cap[k] = want;
        return true;
    }
};

struct Finished { int worker = -1; uint64_t bytes[2] = {0, 0}; uint64_t n_reads = 0; };

}  // namespace

class Sequencer_module::impl {
    int argc; char** argv;
    Args a;

    int parse() {
        auto need = [&](int& i) -> const char* { if (i + 1 >= argc) { usage(stderr); fprintf(stderr, "sequence: error: argument %s: expected one argument\n", argv[i]); return nullptr; } return argv[++i]; };
        for (int i = 1; i < argc; i++) {
            std::string o = argv[i];
            const char* v;
            if (o == "-h" || o == "--help") a.help = true;
            else if (o == "-i" || o == "--input") { if (!(v = need(i))) return 2; a.input = v; }
            else if (o == "-r" || o == "--references") {
                while (i + 1 < argc && argv[i + 1][0] != '-') a.references.push_back(argv[++i]);
                if (a.references.empty()) { usage(stderr); fprintf(stderr, "sequence: error: argument -r/--references: expected at least one argument\n"); return 2; }
            }
            else if (o == "-o" || o == "--badread") { if (!(v = need(i))) return 2; a.badread = v; }
            else if (o == "--perfect") { if (!(v = need(i))) return 2; a.perfect = v; }
            else if (o == "--skip-qual-compute") a.skip_qual = true;
            else if (o == "-O" || o == "--output-format") {
                if (!(v = need(i))) return 2;
                a.output_format = v;      // parsed and, like the reference, not used (py/sequence.py:65-72)
                if (a.output_format != "fastq" && a.output_format != "fasta") { usage(stderr); fprintf(stderr, "sequence: error: argument -O/--output-format: invalid choice: '%s' (choose from 'fastq', 'fasta')\n", v); return 2; }
            }
            else if (o == "-t" || o == "--threads") { if (!(v = need(i))) return 2; a.threads = atoi(v); }
            else if (o == "--badread-identity") { if (!(v = need(i))) return 2; a.identity = v; }
            else if (o == "--badread-error-model") { if (!(v = need(i))) return 2; a.error_model = v; }
            else if (o == "--badread-qscore-model") { if (!(v = need(i))) return 2; a.qscore_model = v; }
            else if (o == "--badread-tail-model") { if (!(v = need(i))) return 2; a.tail_model = v; }
            else if (o == "--list") a.list = true;
            else if (o == "-s" || o == "--seed") { if (!(v = need(i))) return 2; a.seed = atoll(v); }
            else if (o == "--devices") {
                // comma-separated device list: one group of --in-flight contexts per entry (an entry may repeat)
                if (!(v = need(i))) return 2;
                a.devices.clear();
                const char* q = v;
                for (;;) {
                    char* e = nullptr;
                    const long d = strtol(q, &e, 10);
                    if (e == q || d < 0 || (*e && *e != ',')) { usage(stderr); fprintf(stderr, "sequence: error: argument --devices: invalid device list: '%s'\n", v); return 2; }
                    a.devices.push_back((int)d);
                    if (!*e) break;
                    q = e + 1;
                }
            }
            else if (o == "--batch-bytes") {
                if (!(v = need(i))) return 2;
                char* e = nullptr;
                a.batch_bytes = strtoull(v, &e, 10);
                if (e == v || *e || a.batch_bytes < 1) { usage(stderr); fprintf(stderr, "sequence: error: argument --batch-bytes: expected a positive integer, got '%s'\n", v); return 2; }
            }
            else if (o == "--in-flight") {
                if (!(v = need(i))) return 2;
                a.in_flight = atoi(v);
                if (a.in_flight < 1) { usage(stderr); fprintf(stderr, "sequence: error: argument --in-flight: expected a positive integer, got '%s'\n", v); return 2; }
            }
            else if (o == "--verbosity") { if (!(v = need(i))) return 2; a.verbosity = v; }
            else if (o == "--log-file") { if (!(v = need(i))) return 2; a.log_file = v; }
            else { usage(stderr); fprintf(stderr, "sequence: error: unrecognized arguments: %s\n", argv[i]); return 2; }
        }
        return 0;
    }

    static int die(const std::string& msg) { fprintf(stderr, "%s\n", msg.c_str()); return 1; }

public:
    impl(int argc, char** argv) : argc(argc), argv(argv) {}

    int run() {
        int rc = parse();
        if (rc) return rc;
        if (a.help) { usage(stdout); return 0; }
        if (a.list) { for (const char* d : OPTION_DESTS) printf("%s\n", d); return 0; }
        if (a.input.empty()) { usage(stderr); fprintf(stderr, "sequence: error: the following arguments are required: -i/--input\n"); return 2; }
        // py/sequence.py:134-164
        double idv[3]; int nid = 0; bool bad = false;
        {
            size_t p = 0;
            while (p <= a.identity.size()) {
                size_t q = a.identity.find(',', p);
                if (q == std::string::npos) q = a.identity.size();
                std::string t = a.identity.substr(p, q - p);
                char* e = nullptr;
                double v = strtod(t.c_str(), &e);
                if (t.empty() || *e) bad = true;
                if (nid < 3) idv[nid] = v;
                nid++; p = q + 1;
            }
        }
        if (bad) return die("Error: could not parse --identity values");
        if (nid != 3) return die("AssertionError: Must specify 3 values for --badread-identity");
        const double mean = idv[0], maxi = idv[1], sd = idv[2];
        if (mean > 100.0) return die("Error: mean read identity cannot be more than 100");
        if (maxi > 100.0) return die("Error: max read identity cannot be more than 100");
        if (mean <= 50) return die("Error: mean read identity must be at least 50");
        if (maxi <= 50) return die("Error: max read identity must be at least 50");
        if (mean > maxi) { char b[200]; snprintf(b, sizeof b, "Error: mean identity (%g) cannot be larger than max identity (%g)", mean, maxi); return die(b); }
        if (sd < 0.0) return die("Error: read identity stdev cannot be negative");
        if (a.badread.empty() && a.perfect.empty()) { usage(stderr); fprintf(stderr, "sequence: error: Must specify either --output or --perfect.\n"); return 2; }

        // utility flags (src/module.h:106-125)
        Logger log;
        {
            const int lv = Logger::parse(a.verbosity);
            if (lv < 0) return die("Error: unknown verbosity level '" + a.verbosity + "' (choose from DEBUG, INFO, WARN, ERROR, OFF)");
            log.level = lv;
            if (!log.open(a.log_file)) return die("Error: cannot open log file " + a.log_file);
        }
        // $TKSM_MODELS handling of the shim (src/sequence.cpp:38-52) happens in the library's model lookup: the built-in model
        // directory comes first, then the entries of $TKSM_MODELS in order.  Default models: nanopore2020 if it can be found,
        // else `random` (py/sequence.py:86-107).
        if (const char* env = getenv("TKSM_MODELS")) log.log(Logger::DEBUG, "TKSM_MODELS was set to %s; the built-in model directory is searched first", env);
        else log.log(Logger::DEBUG, "TKSM_MODELS not set: built-in model directory only");
        if (a.error_model.empty()) a.error_model = tksmseq_model_available("nanopore2020", "error") ? "nanopore2020" : "random";
        if (a.qscore_model.empty()) a.qscore_model = tksmseq_model_available("nanopore2020", "qscore") ? "nanopore2020" : "random";
        if (a.threads < 1) a.threads = 1;

        const auto t_begin = std::chrono::steady_clock::now();
        Writer wb, wp;
        bool compute_q = false;
        if (!a.badread.empty()) {
            if (!wb.open(a.badread)) return die("Error: cannot open " + a.badread);
            compute_q = !a.skip_qual && wb.fastq;
        }
        if (!a.perfect.empty() && !wp.open(a.perfect)) return die("Error: cannot open " + a.perfect);
        if (!a.badread.empty() && !a.perfect.empty())
            log.log(Logger::WARN, "with both -o and --perfect the reference writes the badread sequence (quals 'K') to the "
                                  "--perfect file (py/sequence.py:317-319); reproduced here");

        // a hardware queue per stream in flight: with the runtime's default of 4 the main streams of several contexts share queues,
        // and kernels of different batches that could run side by side run one after the other (tools/calib/hwq_check.hip).  Read
        // when the runtime starts, i.e. at the first device call below; a value set by the user wins.
        setenv("GPU_MAX_HW_QUEUES", "16", 0);
        // One group of --in-flight contexts per entry of --devices: the first context of a group loads the reference and the
        // models onto its device, the others share them (tksmseq_clone).  Reads are numbered by the reader, batches go to
        // whichever context is free, the writer restores MDF order: the output does not depend on the device list.
        const int n_groups = (int)a.devices.size();
        const int per_group = std::max(1, std::min(a.in_flight, 8));
        const int n_workers = n_groups * per_group;
        std::vector<std::unique_ptr<Worker>> workers;
        for (int w = 0; w < n_workers; w++) workers.emplace_back(new Worker());
        if (const char* pb = getenv("TKSMSEQ_PIECE_BYTES")) for (auto& W : workers) W->piece = std::max<uint64_t>(4096, strtoull(pb, nullptr, 10));
        std::vector<tksmseq_ctx*> pctx;                                     // the parser threads' contexts (clones)
        auto destroy_all = [&]() {
            // clones before the contexts they borrow from
            for (auto& c : pctx) if (c) { tksmseq_destroy(c); c = nullptr; }
            for (int g = 0; g < n_groups; g++) for (int j = per_group - 1; j >= 0; j--) { tksmseq_ctx*& c = workers[(size_t)g * per_group + j]->ctx; if (c) { tksmseq_destroy(c); c = nullptr; } }
        };
        // the models are parsed (and the identity quantile table computed) on threads of their own while the devices are set up and
        // the reference is read and packed: the loaders below find them parsed (models.cpp keeps parsed models)
        std::vector<std::thread> prefetch;
        if (!a.badread.empty()) {
            prefetch.emplace_back([&]() { (void)tksmseq_prefetch_identity(mean, maxi, sd); });
            prefetch.emplace_back([&]() { (void)tksmseq_prefetch_model(a.error_model.c_str(), "error"); });
            if (compute_q) prefetch.emplace_back([&]() { (void)tksmseq_prefetch_model(a.qscore_model.c_str(), "qscore"); });
        }
        std::once_flag prefetch_joined;
        auto join_prefetch = [&]() { std::call_once(prefetch_joined, [&]() { for (auto& t : prefetch) t.join(); }); };
        {
            std::vector<std::string> gerr((size_t)n_groups);
            std::vector<std::thread> gt;
            std::mutex out_m;
            for (int g = 0; g < n_groups; g++)
                gt.emplace_back([&, g]() {
                    tksmseq_ctx* ctx = nullptr;
                    if (tksmseq_create(a.devices[(size_t)g], &ctx)) { gerr[(size_t)g] = std::string("Error: ") + tksmseq_last_error(nullptr); return; }
                    workers[(size_t)g * per_group]->ctx = ctx;
                    auto fail = [&](const std::string& what) { gerr[(size_t)g] = "Error: " + what + ": " + tksmseq_last_error(ctx); };
                    tksmseq_set_host_threads(ctx, a.threads);
                    for (auto& r : a.references) {
                        if (g == 0) { std::lock_guard<std::mutex> l(out_m); printf("Loading reference %s...\n", r.c_str()); fflush(stdout); }
                        if (tksmseq_reference_add_fasta(ctx, r.c_str())) return fail("loading reference");
                    }
                    if (!a.badread.empty()) {
                        join_prefetch();
                        if (tksmseq_set_identity(ctx, mean, maxi, sd)) return fail("identity distribution");
                        if (g == 0) fprintf(stderr, "\nLoading error model from %s\n", a.error_model.c_str());
                        if (tksmseq_load_error_model(ctx, a.error_model.c_str())) return fail("error model");
                        if (compute_q) {
                            if (g == 0) fprintf(stderr, "\nLoading qscore model from %s\n", a.qscore_model.c_str());
                            if (tksmseq_load_qscore_model(ctx, a.qscore_model.c_str())) return fail("qscore model");
                        }
                        if (tksmseq_load_tail_model(ctx, a.tail_model.c_str())) return fail("tail model");     // py/sequence.py:343-345
                    }
                    for (int j = 1; j < per_group; j++)
                        if (tksmseq_clone(ctx, &workers[(size_t)g * per_group + j]->ctx)) return fail("second context");
                });
            for (auto& t : gt) t.join();
            join_prefetch();
            for (auto& e : gerr) if (!e.empty()) { destroy_all(); return die(e); }
        }
        // MDF text is parsed (and its tables uploaded) ahead of the workers, by two parser threads per device group with contexts of
        // their own, so that a worker's cycle is run + copy + write only
        const int parsers_per_group = 2;
        pctx.assign((size_t)n_groups * parsers_per_group, nullptr);
        for (int g = 0; g < n_groups; g++)
            for (int j = 0; j < parsers_per_group; j++) {
                if (tksmseq_clone(workers[(size_t)g * per_group]->ctx, &pctx[(size_t)g * parsers_per_group + j])) {
                    const std::string e = std::string("Error: parser context: ") + tksmseq_last_error(workers[(size_t)g * per_group]->ctx);
                    destroy_all();
                    return die(e);
                }
                tksmseq_set_host_threads(pctx[(size_t)g * parsers_per_group + j], a.threads);
            }
        log.log(Logger::INFO, "%d device group(s) x %d contexts in flight, %d parser(s) per group with %d host thread(s) each", n_groups, per_group,
                parsers_per_group, a.threads);

        FILE* in = fopen(a.input.c_str(), "rb");
        if (!in) { destroy_all(); return die("Error: cannot open " + a.input); }
        ChunkQueue queue;
        queue.cap = (size_t)n_workers;
        std::vector<std::unique_ptr<ParsedQueue>> pq;
        for (int g = 0; g < n_groups; g++) { pq.emplace_back(new ParsedQueue()); pq.back()->cap = (size_t)per_group; }
        std::mutex done_m; std::condition_variable done_cv; std::map<uint64_t, Finished> done;   // by batch number
        std::atomic<bool> failed{false};
        std::mutex err_m; std::string first_error;
        auto set_error = [&](const std::string& msg) {
            std::lock_guard<std::mutex> l(err_m);
            if (!failed.exchange(true)) first_error = msg;
            queue.close();
            for (auto& q2 : pq) q2->close();
            done_cv.notify_all();
            // a worker may be waiting for the writer to release its host buffers: nobody will (the writer stops at the
            // first error), so wake it -- its wait also checks `failed`
            for (auto& W : workers) { std::lock_guard<std::mutex> wl(W->m); W->cv.notify_all(); }
        };
        uint64_t n_batches = 0; bool reader_done = false;                                          // guarded by done_m
        // every open output is a regular file: positional writes from the workers, no writer thread
        const bool positional = (a.badread.empty() || wb.positional) && (a.perfect.empty() || wp.positional);
        uint64_t next_place[2] = {0, 0}, place[2] = {0, 0};                                        // per output; guarded by done_m
        // stage clocks (TKSMSEQ_VERBOSE): seconds spent parsing, running, copying, writing, reading
        const bool verbose = getenv("TKSMSEQ_VERBOSE") != nullptr || log.level <= Logger::DEBUG;
        std::mutex clk_m; double clk[6] = {0, 0, 0, 0, 0, 0};
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto add_clk = [&](int k, std::chrono::steady_clock::time_point t0) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::lock_guard<std::mutex> l(clk_m); clk[k] += dt;
        };
        const bool verbose2 = getenv("TKSMSEQ_VERBOSE") && atoi(getenv("TKSMSEQ_VERBOSE")) >= 2;
        const auto t_start = now();
        if (verbose) fprintf(stderr, "[sequence] device, reference and models ready after %.2f s\n", std::chrono::duration<double>(t_start - t_begin).count());
        uint64_t total_reads = 0;

        auto parse_ahead = [&](int pi) {
            tksmseq_ctx* pc = pctx[(size_t)pi];
            ParsedQueue& out = *pq[(size_t)(pi / parsers_per_group)];
            for (;;) {
                Chunk c;
                uint64_t ticket;
                { std::lock_guard<std::mutex> l(out.take_m); if (!queue.pop(c)) break; ticket = out.taken++; }
                Parsed pr; pr.seq = c.seq; pr.first_read = c.first_read; pr.n_reads = c.n_reads;
                bool ok = !failed;
                if (ok) {
                    const auto t_parse = now();
                    if (tksmseq_batch_from_mdf_text(pc, c.text.data(), c.text.size(), &pr.b)) { set_error(tksmseq_last_error(pc)); ok = false; }
                    add_clk(0, t_parse);
                }
                std::unique_lock<std::mutex> l(out.order_m);
                out.order_cv.wait(l, [&] { return out.handed == ticket; });
                if (ok && !out.push(pr)) tksmseq_batch_free(pc, pr.b);     // (closed after an error)
                out.handed++;
                out.order_cv.notify_all();
            }
        };
        // regular files: batch `seq` takes its place in output k once every earlier batch has (sizes are announced in batch order)
        const int first_out = a.badread.empty() ? 1 : 0;
        auto take_place = [&](int k, uint64_t seq_no, uint64_t bytes, uint64_t n_reads_of_batch, uint64_t& off) -> bool {
            {
                std::unique_lock<std::mutex> l(done_m);
                done_cv.wait(l, [&] { return next_place[k] == seq_no || failed.load(); });
                if (failed) return false;
                off = place[k]; place[k] += bytes; next_place[k]++;
                if (k == first_out) total_reads += n_reads_of_batch;
            }
            done_cv.notify_all();
            return true;
        };
        auto work = [&](int wi) {
            Worker& W = *workers[wi];
            ParsedQueue& in_q = *pq[(size_t)(wi / per_group)];
            Parsed c;
            while (in_q.pop(c)) {
                tksmseq_batch* b = c.b;
                if (failed) { tksmseq_batch_free(W.ctx, b); continue; }
                uint64_t n = 0;
                tksmseq_batch_info(b, &n, nullptr, nullptr);
                Finished fin; fin.worker = wi; fin.n_reads = n;
                bool ok = n == c.n_reads;
                if (!ok) set_error("internal: the reader and the parser disagree on the number of reads of a batch");
                bool waited = false;
                auto emit = [&](int k, bool fastq, int mode, int quirk) -> bool {
                    tksmseq_run_params p{};
                    p.seed = (uint64_t)a.seed; p.first_read_index = c.first_read; p.read_index_stride = 1;
                    p.mode = mode; p.fastq = fastq; p.compute_qual = compute_q; p.perfect_of_badread = quirk;
                    tksmseq_result r{};
                    const auto t_run = now();
                    if (tksmseq_run(W.ctx, b, &p, &r)) { set_error(tksmseq_last_error(W.ctx)); return false; }
                    add_clk(1, t_run);
                    if (verbose2) fprintf(stderr, "[sequence] batch %llu worker %d: run %.3f s (%llu reads) at %.3f s\n", (unsigned long long)c.seq, wi,
                                          std::chrono::duration<double>(now() - t_run).count(), (unsigned long long)n, std::chrono::duration<double>(now() - t_start).count());
                    Writer& wr = k == 0 ? wb : wp;
                    if (positional && !wr.gz) {
                        // a regular uncompressed file: the batch's place in it is known as soon as every earlier batch has announced
                        // its size; the records go there in pieces, straight from the device (pwrite into the page cache takes
                        // ~4 GB/s per thread; the copy of the next piece runs meanwhile)
                        uint64_t off = 0;
                        if (!take_place(k, c.seq, r.records_bytes, n, off)) return false;
                        if (!W.ring_ready()) { set_error("out of page-locked host memory"); return false; }
                        const uint64_t np = (r.records_bytes + W.piece - 1) / W.piece;
                        auto piece_bytes = [&](uint64_t q) { return std::min<uint64_t>(W.piece, r.records_bytes - q * W.piece); };
                        if (np && tksmseq_result_download_range(W.ctx, W.ring[0], 0, piece_bytes(0), 1)) { set_error(tksmseq_last_error(W.ctx)); return false; }
                        for (uint64_t q = 0; q < np; q++) {
                            const auto t_copy = now();
                            if (tksmseq_synchronize(W.ctx)) { set_error(tksmseq_last_error(W.ctx)); return false; }
                            if (q + 1 < np && tksmseq_result_download_range(W.ctx, W.ring[(q + 1) & 1], (q + 1) * W.piece, piece_bytes(q + 1), 1)) { set_error(tksmseq_last_error(W.ctx)); return false; }
                            add_clk(2, t_copy);
                            const auto t_write = now();
                            const bool wok = wr.write_at(W.ring[q & 1], piece_bytes(q), off + q * W.piece);
                            add_clk(3, t_write);
                            if (!wok) { (void)tksmseq_synchronize(W.ctx); set_error("write failed"); return false; }
                        }
                        fin.bytes[k] = r.records_bytes;
                        return true;
                    }
                    const auto t_wait = now();
                    if (!waited) {                                  // the previous batch of this worker has been written
                        std::unique_lock<std::mutex> l(W.m);
                        W.cv.wait(l, [&] { return !W.host_busy || failed.load(); });
                        waited = true;
                    }
                    add_clk(4, t_wait);
                    if (failed) return false;
                    const auto t_copy = now();
                    if (!W.reserve(k, r.records_bytes)) { set_error("out of page-locked host memory"); return false; }
                    const auto t_copy2 = now();
                    if (tksmseq_result_download(W.ctx, W.host[k], nullptr)) { set_error(tksmseq_last_error(W.ctx)); return false; }
                    add_clk(2, t_copy);
                    if (verbose2) fprintf(stderr, "[sequence] batch %llu worker %d: host buffer %.3f s, copy %.3f s\n", (unsigned long long)c.seq, wi,
                                          std::chrono::duration<double>(t_copy2 - t_copy).count(), std::chrono::duration<double>(now() - t_copy2).count());
                    fin.bytes[k] = r.records_bytes;
                    if ((k == 0 ? wb : wp).gz) {
                        // 16 MB pieces, compressed side by side (level 1), concatenated in order
                        const size_t piece = 16u << 20, np = (size_t)((r.records_bytes + piece - 1) / piece);
                        std::vector<std::vector<uint8_t>> parts(np);
                        std::vector<char> okp(np, 0);
                        std::vector<std::thread> zt;
                        std::atomic<size_t> nextp{0};
                        auto zwork = [&]() { for (size_t q; (q = nextp++) < np;) okp[q] = gzip_member(W.host[k] + q * piece, (size_t)std::min<uint64_t>(piece, r.records_bytes - q * piece), parts[q]); };
                        for (size_t q = 0; q < std::min<size_t>(np, 6); q++) zt.emplace_back(zwork);
                        for (auto& t2 : zt) t2.join();
                        size_t total = 0;
                        for (size_t q = 0; q < np; q++) { if (!okp[q]) { set_error("gzip compression failed"); return false; } total += parts[q].size(); }
                        W.packed[k].resize(total);
                        size_t at = 0;
                        for (size_t q = 0; q < np; q++) { memcpy(W.packed[k].data() + at, parts[q].data(), parts[q].size()); at += parts[q].size(); }
                        fin.bytes[k] = total;
                        if (positional) {                           // a regular .gz file: the worker writes its members at their place
                            uint64_t off = 0;
                            if (!take_place(k, c.seq, total, n, off)) return false;
                            const auto t_write = now();
                            const bool wok = wr.write_at(W.packed[k].data(), total, off);
                            add_clk(3, t_write);
                            if (!wok) { set_error("write failed"); return false; }
                        }
                    }
                    return true;
                };
                if (ok && n) {
                    if (!a.badread.empty()) ok = emit(0, wb.fastq, TKSMSEQ_MODE_BADREAD, 0);
                    if (ok && !a.perfect.empty()) ok = a.badread.empty() ? emit(1, wp.fastq, TKSMSEQ_MODE_PERFECT, 0) : emit(1, wp.fastq, TKSMSEQ_MODE_BADREAD, 1);
                } else if (ok && positional) {                      // an empty batch still takes its (empty) place
                    uint64_t off = 0;
                    if (!a.badread.empty()) ok = take_place(0, c.seq, 0, 0, off);
                    if (ok && !a.perfect.empty()) ok = take_place(1, c.seq, 0, 0, off);
                }
                tksmseq_batch_free(W.ctx, b);
                if (!ok || positional) continue;                    // (regular files: written inside emit)
                { std::lock_guard<std::mutex> l(W.m); W.host_busy = true; }
                { std::lock_guard<std::mutex> l(done_m); done[c.seq] = fin; }
                done_cv.notify_all();
            }
        };
        auto write_all = [&]() {
            uint64_t next = 0;
            for (;;) {
                Finished fin;
                {
                    std::unique_lock<std::mutex> l(done_m);
                    done_cv.wait(l, [&] { return done.count(next) || failed.load() || (reader_done && next >= n_batches); });
                    if (failed || !done.count(next)) return;
                    fin = done[next]; done.erase(next);
                }
                Worker& W = *workers[fin.worker];
                bool ok = true;
                const auto t_write = now();
                if (fin.bytes[0]) ok = wb.write(wb.gz ? W.packed[0].data() : W.host[0], fin.bytes[0]);
                if (ok && fin.bytes[1]) ok = wp.write(wp.gz ? W.packed[1].data() : W.host[1], fin.bytes[1]);
                add_clk(3, t_write);
                { std::lock_guard<std::mutex> l(W.m); W.host_busy = false; }
                W.cv.notify_all();
                if (!ok) { set_error("write failed"); return; }
                total_reads += fin.n_reads;
                next++;
            }
        };
        std::vector<std::thread> threads, parsers;
        for (int w = 0; w < n_workers; w++) threads.emplace_back(work, w);
        for (int pi = 0; pi < n_groups * parsers_per_group; pi++) parsers.emplace_back(parse_ahead, pi);
        std::thread writer;
        if (!positional) writer = std::thread(write_all);

        // reader: batches of whole molecules, numbered; the first read index of a batch is known before it is parsed
        std::vector<char> buf;
        uint64_t read_index = 0, seq = 0;
        bool eof = false;
        size_t have = 0;
        while ((!eof || have) && !failed) {
            // fill up to batch_bytes, then cut at the last molecule header so a batch holds whole molecules
            const auto t_read = now();
            buf.resize(have + a.batch_bytes);
            size_t got = eof ? 0 : fread(buf.data() + have, 1, a.batch_bytes, in);
            if (got < a.batch_bytes) eof = true;
            have += got;
            size_t cut = have;
            if (!eof) {
                size_t p = have;
                while (p > 1 && !(buf[p - 1] == '\n' && p < have && buf[p] == '+')) p--;
                if (p <= 1) { buf.resize(have); continue; }   // no boundary yet: read more
                cut = p;
            }
            if (cut == 0) break;
            Chunk c;
            c.seq = seq++; c.first_read = read_index;
            c.text.assign(buf.begin(), buf.begin() + (ptrdiff_t)cut);
            c.n_reads = count_reads(c.text.data(), c.text.size());
            read_index += c.n_reads;
            add_clk(5, t_read);
            queue.push(std::move(c));
            memmove(buf.data(), buf.data() + cut, have - cut);
            have -= cut;
        }
        { std::lock_guard<std::mutex> l(done_m); n_batches = seq; reader_done = true; }
        queue.close();
        done_cv.notify_all();
        for (auto& t : parsers) t.join();
        for (auto& q2 : pq) q2->close();                                                           // (the workers take what is still queued)
        for (auto& t : threads) t.join();
        done_cv.notify_all();
        if (writer.joinable()) writer.join();
        if (positional && !failed) { wb.wrote = place[0] != 0; wp.wrote = place[1] != 0; }
        int status = failed ? 1 : 0;
        if (status) fprintf(stderr, "Error: %s\n", first_error.c_str());
        if (verbose)
            fprintf(stderr, "[sequence] %d batches, %d in flight, %.2f s streaming: parse %.2f, run %.2f, copy %.2f, wait for writer %.2f "
                            "(summed over workers); write %.2f; read + count %.2f\n", (int)seq, n_workers,
                    std::chrono::duration<double>(now() - t_start).count(), clk[0], clk[1], clk[2], clk[4], clk[3], clk[5]);
        for (auto& W : workers) { tksmseq_host_free(W->host[0]); tksmseq_host_free(W->host[1]); tksmseq_host_free(W->ring[0]); tksmseq_host_free(W->ring[1]); }
        const auto t_close = now();
        fclose(in);
        if ((!wb.close() || !wp.close()) && !status) { status = 1; fprintf(stderr, "Error: write failed\n"); }
        const auto t_destroy = now();
        destroy_all();
        if (verbose)
            fprintf(stderr, "[sequence] closing the outputs %.2f s, releasing the device %.2f s\n",
                    std::chrono::duration<double>(t_destroy - t_close).count(), std::chrono::duration<double>(now() - t_destroy).count());
        if (!status) log.log(Logger::INFO, "Sequencing: %llu reads", (unsigned long long)total_reads);
        return status;
    }
};

Sequencer_module::Sequencer_module(int argc, char** argv) : pimpl{std::make_unique<impl>(argc, argv)} {}
Sequencer_module::~Sequencer_module() = default;
int Sequencer_module::run() { return pimpl->run(); }

extern "C" int tksmseq_sequence_main(int argc, char** argv) { return Sequencer_module{argc, argv}.run(); }
