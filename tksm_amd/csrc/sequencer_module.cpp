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
// device work of consecutive batches, the copies and the writes overlap.  Regular output files are written at their final
// offsets (pwrite, MDF order) by a writer thread per worker, from a device-side copy of the batch's records, while the worker's
// context runs its next batch; pipes and devices by one writer that puts the record streams back into MDF order.
// Exit codes: 0 ok; 1 for `sys.exit("msg")`-style validation and runtime errors; 2 for argparse
// usage errors (missing -i, neither -o nor --perfect) -- what the embedded interpreter returns.
#include "sequencer_module.h"

#include <fcntl.h>
#include <poll.h>
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
#include "module_log.h"

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
    // chained stages (BASELINE config 5): `tksm pcr` and / or `tksm truncate` in front of the sequencer, molecule tables staying on
    // the device -- same results as the three-module route over MDF files with the same -s (src/pcr.cpp:91-260, src/truncate.cpp:236-451)
    bool pcr_on = false, pcr_have_cycles = false, pcr_have_count = false, pcr_have_er = false, pcr_have_ef = false;
    tksmseq_pcr_params pcr{};
    std::string pcr_preset;
    uint64_t pcr_slice = 2000000;
    int trc_n = 0;
    tksmseq_trc_params trc{};
    std::string trc_kde;
};

const char* OPTION_DESTS[] = {"help", "input", "references", "badread", "perfect", "skip_qual_compute", "output_format",
                              "threads", "badread_identity", "badread_error_model", "badread_qscore_model",
                              "badread_tail_model", "list", "seed", "devices", "verbosity", "log_file"};

void usage(FILE* f) {
    fprintf(f,
            "usage: sequence [-h] -i INPUT [-r REFERENCES [REFERENCES ...]] [-o BADREAD] [--perfect PERFECT]\n"
            "                [--skip-qual-compute] [-O {fastq,fasta}] [-t THREADS] [--badread-identity BADREAD_IDENTITY]\n"
            "                [--badread-error-model M] [--badread-qscore-model M] [--badread-tail-model M] [--list]\n"
            "                [-s SEED] [--devices D[,D...]] [--batch-bytes B] [--in-flight N] [--verbosity L] [--log-file F]\n"
            "                [--pcr-cycles C --pcr-molecule-count N (--pcr-preset X | --pcr-error-rate E --pcr-efficiency F)]\n"
            "                [--truncate-normal MU,SIGMA | --truncate-lognormal MU,SIGMA | --truncate-kde-model M.json\n"
            "                 [--truncate-always-end] [--truncate-kde-models-length]]\n");
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
    void classify(const std::string& path) {             // get_output_file, py/sequence.py:291-300: the NAME decides format and compression
        std::string p = path;
        gz = false;
        if (p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0) { gz = true; p.resize(p.size() - 3); }
        auto ends = [&](const char* s) { size_t n = strlen(s); return p.size() >= n && p.compare(p.size() - n, n, s) == 0; };
        fastq = ends(".fastq") || ends(".fq");
    }
    bool open(const std::string& path) {                 // creates / truncates: only once devices, references, models and the input are usable
        classify(path);
        fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) return false;
        struct stat st;
        positional = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
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

// --verbosity / --log-file: module_log.h.  What the reference's Python prints unconditionally (model loading progress, "Loading
// reference") stays unconditional; the module's own diagnostics go through the logger.
using tkmod::Logger;

using tkmod::count_reads;

struct Chunk { uint64_t seq = 0, first_read = 0, n_reads = 0; std::vector<char> text; uint64_t t_begin = 0, t_end = 0; };   // text, or (chained PCR) a slice of the templates

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
    // ... behind the worker's back: the batch's records are copied into one of two device buffers of the writer's (a device copy
    // takes a millisecond), and the worker's context runs the next batch while a thread of its own streams the buffer out
    struct Job { int stage = 0, k = 0; uint64_t bytes = 0, off = 0, seq = 0; };
    tksmseq_ctx* wctx = nullptr;                        // the writer thread's context (its stream carries the copies)
    void* stage[2] = {nullptr, nullptr}; uint64_t stage_cap[2] = {0, 0};
    bool stage_busy[2] = {false, false};                // guarded by m
    std::deque<Job> jobs; bool jobs_closed = false;     // guarded by m
    int cur_stage = 0;
    bool stage_reserve(int q, uint64_t bytes) {
        if (bytes <= stage_cap[q]) return true;
        tksmseq_device_free(wctx, stage[q]); stage[q] = nullptr; stage_cap[q] = 0;
        const uint64_t want = bytes + bytes / 8 + 4096;
        if (tksmseq_device_alloc(wctx, want, &stage[q])) return false;
        stage_cap[q] = want;
        return true;
    }
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

// The reference's parser is argparse with its defaults (py/sequence.py:35-40, :124): a long option may be abbreviated to any
// unambiguous prefix, `--opt=value` and a value glued to a short option (`-t8`) are accepted.  The command line is rewritten into
// the plain `--opt value` form the loop below reads; `glued` marks values that came with their option (for -r/--references,
// nargs="+": an explicit value is the option's only one).
struct OptSpec { const char* name; bool takes_value; };
const OptSpec LONG_OPTS[] = {
    {"--help", false}, {"--input", true}, {"--references", true}, {"--badread", true}, {"--perfect", true}, {"--skip-qual-compute", false},
    {"--output-format", true}, {"--threads", true}, {"--badread-identity", true}, {"--badread-error-model", true},
    {"--badread-qscore-model", true}, {"--badread-tail-model", true}, {"--list", false}, {"--seed", true}, {"--devices", true},
    {"--batch-bytes", true}, {"--in-flight", true}, {"--pcr-cycles", true}, {"--pcr-molecule-count", true}, {"--pcr-error-rate", true},
    {"--pcr-efficiency", true}, {"--pcr-preset", true}, {"--pcr-slice-molecules", true}, {"--truncate-normal", true},
    {"--truncate-lognormal", true}, {"--truncate-kde-model", true}, {"--truncate-always-end", false},
    {"--truncate-kde-models-length", false}, {"--verbosity", true}, {"--log-file", true}};
const char SHORT_WITH_VALUE[] = "iroOts";

int normalise_args(int argc, char** argv, std::vector<std::string>& out, std::vector<char>& glued) {
    out.clear(); glued.clear();
    auto put = [&](const std::string& t, bool g) { out.push_back(t); glued.push_back(g ? 1 : 0); };
    if (argc > 0) put(argv[0], false);
    for (int i = 1; i < argc; i++) {
        const std::string t = argv[i];
        if (t.size() > 2 && t[0] == '-' && t[1] == '-') {
            const size_t eq = t.find('=');
            const std::string name = t.substr(0, eq);
            const OptSpec* hit = nullptr; std::string could;
            int n_hit = 0;
            for (const OptSpec& o : LONG_OPTS) if (name == o.name) { hit = &o; n_hit = 1; break; }
            if (!hit)
                for (const OptSpec& o : LONG_OPTS)
                    if (strncmp(o.name, name.c_str(), name.size()) == 0) { hit = &o; n_hit++; could += (could.empty() ? "" : ", ") + std::string(o.name); }
            if (n_hit > 1) { usage(stderr); fprintf(stderr, "sequence: error: ambiguous option: %s could match %s\n", name.c_str(), could.c_str()); return 2; }
            if (n_hit == 0) { put(t, false); continue; }                       // the loop reports it as unrecognized
            put(hit->name, false);
            if (eq != std::string::npos) {
                if (!hit->takes_value) { usage(stderr); fprintf(stderr, "sequence: error: argument %s: ignored explicit argument '%s'\n", hit->name, t.substr(eq + 1).c_str()); return 2; }
                put(t.substr(eq + 1), true);
            }
        } else if (t.size() > 2 && t[0] == '-' && t[1] != '-' && strchr(SHORT_WITH_VALUE, t[1])) {
            put(t.substr(0, 2), false);
            put(t.substr(t[2] == '=' ? 3 : 2), true);
        } else put(t, false);
    }
    return 0;
}

class Sequencer_module::impl {
    int argc; char** argv;
    Args a;
    std::vector<std::string> norm; std::vector<char> glued; std::vector<char*> norm_argv;

    int parse() {
        if (int rc = normalise_args(argc, argv, norm, glued)) return rc;
        norm_argv.clear();
        for (auto& t : norm) norm_argv.push_back(const_cast<char*>(t.c_str()));
        const int argc = (int)norm_argv.size(); char** const argv = norm_argv.data();
        auto need = [&](int& i) -> const char* { if (i + 1 >= argc) { usage(stderr); fprintf(stderr, "sequence: error: argument %s: expected one argument\n", argv[i]); return nullptr; } return argv[++i]; };
        for (int i = 1; i < argc; i++) {
            std::string o = argv[i];
            const char* v;
            if (o == "-h" || o == "--help") a.help = true;
            else if (o == "-i" || o == "--input") { if (!(v = need(i))) return 2; a.input = v; }
            else if (o == "-r" || o == "--references") {
                if (i + 1 < argc && glued[(size_t)i + 1]) a.references.push_back(argv[++i]);
                else while (i + 1 < argc && argv[i + 1][0] != '-') a.references.push_back(argv[++i]);
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
                if (!tkmod::parse_device_list(v, a.devices)) { usage(stderr); fprintf(stderr, "sequence: error: argument --devices: invalid device list: '%s'\n", v); return 2; }
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
            else if (o == "--pcr-cycles") { if (!(v = need(i))) return 2; a.pcr.cycles = atoi(v); a.pcr_have_cycles = true; }
            else if (o == "--pcr-molecule-count") { if (!(v = need(i))) return 2; a.pcr.target_count = strtoull(v, nullptr, 10); a.pcr_have_count = true; }
            else if (o == "--pcr-error-rate") { if (!(v = need(i))) return 2; a.pcr.error_rate = atof(v); a.pcr_have_er = true; }
            else if (o == "--pcr-efficiency") { if (!(v = need(i))) return 2; a.pcr.efficiency = atof(v); a.pcr_have_ef = true; }
            else if (o == "--pcr-preset") { if (!(v = need(i))) return 2; a.pcr_preset = v; }
            else if (o == "--pcr-slice-molecules") { if (!(v = need(i))) return 2; a.pcr_slice = std::max<uint64_t>(1, strtoull(v, nullptr, 10)); }
            else if (o == "--truncate-normal" || o == "--truncate-lognormal") {
                if (!(v = need(i))) return 2;
                char* e = nullptr;
                a.trc.mu = strtod(v, &e);
                if (!e || *e != ',') { usage(stderr); fprintf(stderr, "sequence: error: argument %s: expected MU,SIGMA\n", o.c_str()); return 2; }
                a.trc.sigma = strtod(e + 1, &e);
                if (!e || *e) { usage(stderr); fprintf(stderr, "sequence: error: argument %s: expected MU,SIGMA\n", o.c_str()); return 2; }
                a.trc.mode = o == "--truncate-normal" ? TKSMSEQ_TRC_NORMAL : TKSMSEQ_TRC_LOGNORMAL; a.trc_n++;
            }
            else if (o == "--truncate-kde-model") { if (!(v = need(i))) return 2; a.trc_kde = v; a.trc.mode = TKSMSEQ_TRC_KDE; a.trc_n++; }
            else if (o == "--truncate-always-end") a.trc.always_end = 1;
            else if (o == "--truncate-kde-models-length") a.trc.kde_models_length = 1;
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
        // the chained stages' own argument checks (src/pcr.cpp:148-185, src/truncate.cpp:278-300)
        a.pcr_on = a.pcr_have_cycles || a.pcr_have_count || a.pcr_have_er || a.pcr_have_ef || !a.pcr_preset.empty();
        if (a.pcr_on) {
            int missing = 0;
            if (!a.pcr_have_count) { fprintf(stderr, "molecule-count is required!\n"); missing++; }
            if (!a.pcr_have_cycles) { fprintf(stderr, "cycles is required!\n"); missing++; }
            if (!a.pcr_preset.empty()) {
                double er = 0, ef = 0;
                if (tksmseq_pcr_preset(a.pcr_preset.c_str(), &er, &ef)) { fprintf(stderr, "Preset %s not found\n", a.pcr_preset.c_str()); missing++; }
                else { if (!a.pcr_have_er) a.pcr.error_rate = er; if (!a.pcr_have_ef) a.pcr.efficiency = ef; }
            } else {
                if (!a.pcr_have_er) { fprintf(stderr, "Error rate is required!\n"); missing++; }
                if (!a.pcr_have_ef) { fprintf(stderr, "Efficiency is required!\n"); missing++; }
            }
            if (missing) return 1;
            a.pcr.seed = (uint64_t)a.seed;
        }
        if (a.trc_n > 1) return die("Only one of kde-model, normal or lognormal is allowed!");
        if (a.trc_n == 1) { a.trc.seed = (uint64_t)a.seed; if (a.trc.mode == TKSMSEQ_TRC_KDE) a.trc.kde_model_path = a.trc_kde.c_str(); }

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
        // the output NAMES decide what is computed (py/sequence.py:349-353); the files are created below, after the devices, references,
        // models and the input have turned out usable (a run that fails before that leaves no empty output behind)
        Writer wb, wp;
        bool compute_q = false;
        if (!a.badread.empty()) { wb.classify(a.badread); compute_q = !a.skip_qual && wb.fastq; }
        if (!a.perfect.empty()) wp.classify(a.perfect);
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
        // the input is read through its descriptor: a pipe (Snakemake's `tksm ... | tksm sequence -i /dev/stdin`, Snakefile:283-305) may stay
        // open with nothing to read while a worker has already failed -- the reader polls it and gives up then, instead of sleeping in a
        // read() until the producer closes
        const int in_fd = fileno(in);
        bool in_regular = false;
        { struct stat st_in; in_regular = fstat(in_fd, &st_in) == 0 && S_ISREG(st_in.st_mode); }
        if (!a.badread.empty() && !wb.open(a.badread)) { fclose(in); destroy_all(); return die("Error: cannot open " + a.badread); }
        if (!a.perfect.empty() && !wp.open(a.perfect)) { fclose(in); wb.close(); destroy_all(); return die("Error: cannot open " + a.perfect); }
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
            // (under done_m: a writer that has just evaluated its predicate -- `failed` still false -- holds done_m until it blocks, so the
            // notification cannot fall between its check and its wait)
            { std::lock_guard<std::mutex> dl(done_m); done_cv.notify_all(); }
            // a worker may be waiting for the writer to release its host buffers: nobody will (the writer stops at the
            // first error), so wake it -- its wait also checks `failed`
            for (auto& W : workers) { std::lock_guard<std::mutex> wl(W->m); W->cv.notify_all(); }
        };
        uint64_t n_batches = 0; bool reader_done = false;                                          // guarded by done_m
        // every open output is a regular file: positional writes from the workers, no writer thread
        const bool positional = (a.badread.empty() || wb.positional) && (a.perfect.empty() || wp.positional);
        // uncompressed outputs (files, pipes, devices) are written BEHIND the workers: a batch's records move into a staging buffer on
        // the device, and a writer thread per worker streams them out -- at their place in a regular file, in batch order into
        // anything else -- while the worker's context runs its next batch.  (.gz outputs keep whole-batch host buffers: the members
        // are compressed side by side.)
        const bool behind = (a.badread.empty() || !wb.gz) && (a.perfect.empty() || !wp.gz);
        uint64_t written_upto[2] = {0, 0};                                                          // non-seekable outputs: batches written; guarded by done_m
        uint64_t next_place[2] = {0, 0}, place[2] = {0, 0};                                        // per output; guarded by done_m
        // stage clocks (TKSMSEQ_VERBOSE): seconds spent parsing, running, copying, writing, reading
        const bool verbose = getenv("TKSMSEQ_VERBOSE") != nullptr || log.level <= Logger::DEBUG;
        std::mutex clk_m; double clk[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // 0 parse, 1 run, 2 device copy / download, 3 write, 4 wait for writer, 5 read + count, 6 wait for device-to-host pieces
        std::atomic<uint64_t> bytes_in{0}, bytes_d2h{0};
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto add_clk = [&](int k, std::chrono::steady_clock::time_point t0) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::lock_guard<std::mutex> l(clk_m); clk[k] += dt;
        };
        const bool verbose2 = getenv("TKSMSEQ_VERBOSE") && atoi(getenv("TKSMSEQ_VERBOSE")) >= 2;
        const auto t_start = now();
        if (verbose) fprintf(stderr, "[sequence] device, reference and models ready after %.2f s\n", std::chrono::duration<double>(t_start - t_begin).count());
        uint64_t total_reads = 0;

        std::vector<tksmseq_batch*> templates((size_t)n_groups, nullptr);        // chained PCR: the whole input, one batch per device group
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
                    if (a.pcr_on) {
                        // a slice of the templates amplified on the device: its copies are numbered from c.first_read on
                        tksmseq_pcr_params q = a.pcr;
                        q.flags = TKSMSEQ_MOL_NO_COMMENTS;             // (Seq never reads header comments: no per-molecule text on the host)
                        q.template_begin = c.t_begin; q.template_end = c.t_end;
                        if (c.t_begin == c.t_end) { q.template_begin = q.template_end = 0; q.cycles = 0; }
                        if (tksmseq_pcr(pc, templates[(size_t)(pi / parsers_per_group)], &q, &pr.b)) { set_error(tksmseq_last_error(pc)); ok = false; }
                        if (verbose2) fprintf(stderr, "[sequence] slice %llu parser %d: pcr %.3f s at %.3f s\n", (unsigned long long)c.seq, pi,
                                              std::chrono::duration<double>(now() - t_parse).count(), std::chrono::duration<double>(now() - t_start).count());
                    } else if ((a.trc_n ? tksmseq_molecules_from_mdf_text : tksmseq_batch_from_mdf_text)(pc, c.text.data(), c.text.size(), &pr.b)) { set_error(tksmseq_last_error(pc)); ok = false; }
                    if (ok && a.trc_n) {
                        tksmseq_trc_params q = a.trc;
                        q.flags = TKSMSEQ_MOL_NO_COMMENTS;
                        q.first_molecule_index = c.first_read;
                        tksmseq_batch* cut = nullptr;
                        if (tksmseq_truncate(pc, pr.b, &q, &cut)) { set_error(tksmseq_last_error(pc)); ok = false; }
                        tksmseq_batch_free(pc, pr.b);
                        pr.b = cut;
                    }
                    add_clk(0, t_parse);
                    if (verbose2) fprintf(stderr, "[sequence] batch %llu parser %d: parsed / made in %.3f s at %.3f s\n", (unsigned long long)c.seq, pi,
                                          std::chrono::duration<double>(now() - t_parse).count(), std::chrono::duration<double>(now() - t_start).count());
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
                    if (behind || (positional && !wr.gz)) {
                        // an uncompressed output (or the plain regular file next to a .gz one); a regular file: the batch's place in it is known as soon as every earlier batch has announced
                        // its size (writes into ONE file are serialised by the file system: 11 - 13.5 GB/s on the test box whatever
                        // the number of threads, tools/fs_write_probe.py -- the bound of the end-to-end rate)
                        uint64_t off = 0;
                        if (!take_place(k, c.seq, r.records_bytes, n, off)) return false;
                        // (blocks allocated ahead of the writes: the writes into one file are serialised by the file system, and
                        // the allocation would happen inside them -- 12 -> 13.5 GB/s on the test box, tools/fs_write_probe.py)
                        if (r.records_bytes && wr.positional) (void)posix_fallocate(wr.fd, (off_t)off, (off_t)r.records_bytes);
                        // the records move into a staging buffer of the writer thread's (device to device), which streams them
                        // out while this context runs its next batch
                        const auto t_copy = now();
                        int q;
                        {
                            std::unique_lock<std::mutex> l(W.m);
                            q = W.cur_stage;
                            W.cv.wait(l, [&] { return !W.stage_busy[q] || failed.load(); });
                            if (failed) return false;
                            W.stage_busy[q] = true;
                            W.cur_stage ^= 1;
                        }
                        if (!W.stage_reserve(q, r.records_bytes)) { set_error("out of device memory for the output staging buffers"); return false; }
                        if (r.records_bytes && (tksmseq_result_copy_device(W.ctx, W.stage[q], nullptr) || tksmseq_synchronize(W.ctx))) { set_error(tksmseq_last_error(W.ctx)); return false; }
                        add_clk(2, t_copy);
                        { std::lock_guard<std::mutex> l(W.m); Worker::Job j; j.stage = q; j.k = k; j.bytes = r.records_bytes; j.off = off; j.seq = c.seq; W.jobs.push_back(j); }
                        W.cv.notify_all();
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
                } else if (ok && (positional || behind)) {          // an empty batch still takes its (empty) place
                    uint64_t off = 0;
                    for (int k2 = 0; k2 < 2 && ok; k2++) {
                        if ((k2 == 0 ? a.badread : a.perfect).empty()) continue;
                        ok = take_place(k2, c.seq, 0, 0, off);
                        if (ok && behind && !(k2 == 0 ? wb : wp).positional) {     // ... and its turn in a non-seekable output
                            std::unique_lock<std::mutex> l(done_m);
                            done_cv.wait(l, [&] { return written_upto[k2] == c.seq || failed.load(); });
                            written_upto[k2]++;
                            done_cv.notify_all();
                        }
                    }
                }
                tksmseq_batch_free(W.ctx, b);
                if (!ok || positional || behind) continue;          // (written inside emit / behind the worker)
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
                { std::lock_guard<std::mutex> l(done_m); place[0] += fin.bytes[0]; place[1] += fin.bytes[1]; }     // (bytes written, for the run's summary)
                next++;
            }
        };
        // positional outputs: a writer thread per worker takes the staged batches in order, copies them to the host in pieces
        // (two page-locked pieces: the copy of one under the write of the other) and writes them at their place
        auto write_behind = [&](int wi) {
            Worker& W = *workers[wi];
            for (;;) {
                Worker::Job j;
                {
                    std::unique_lock<std::mutex> l(W.m);
                    W.cv.wait(l, [&] { return !W.jobs.empty() || W.jobs_closed; });
                    if (W.jobs.empty()) return;
                    j = W.jobs.front(); W.jobs.pop_front();
                }
                Writer& wr = j.k == 0 ? wb : wp;
                bool ok = !failed && W.ring_ready();
                if (!ok && !failed) set_error("out of page-locked host memory");
                const uint64_t np = (j.bytes + W.piece - 1) / W.piece;
                auto piece_bytes = [&](uint64_t q) { return std::min<uint64_t>(W.piece, j.bytes - q * W.piece); };
                const uint8_t* src = (const uint8_t*)W.stage[j.stage];
                if (ok && np && tksmseq_copy_to_host(W.wctx, W.ring[0], src, piece_bytes(0), 1)) { set_error(tksmseq_last_error(W.wctx)); ok = false; }
                if (ok && !wr.positional) {                         // a pipe / device: the batches before this one have been written
                    std::unique_lock<std::mutex> l(done_m);
                    done_cv.wait(l, [&] { return written_upto[j.k] == j.seq || failed.load(); });
                    ok = !failed;
                }
                for (uint64_t q = 0; q < np && ok; q++) {
                    const auto t_d2h = now();
                    const bool sync_failed = tksmseq_synchronize(W.wctx) != 0;
                    add_clk(6, t_d2h);
                    bytes_d2h += piece_bytes(q);
                    if (sync_failed) { set_error(tksmseq_last_error(W.wctx)); ok = false; break; }
                    if (q + 1 < np && tksmseq_copy_to_host(W.wctx, W.ring[(q + 1) & 1], src + (q + 1) * W.piece, piece_bytes(q + 1), 1)) { set_error(tksmseq_last_error(W.wctx)); ok = false; break; }
                    const auto t_write = now();
                    const bool wok = wr.positional ? wr.write_at(W.ring[q & 1], piece_bytes(q), j.off + q * W.piece) : wr.write(W.ring[q & 1], piece_bytes(q));
                    add_clk(3, t_write);
                    if (!wok) { (void)tksmseq_synchronize(W.wctx); set_error("write failed"); ok = false; }
                }
                if (!wr.positional) { { std::lock_guard<std::mutex> l(done_m); if (ok) written_upto[j.k]++; } done_cv.notify_all(); }
                { std::lock_guard<std::mutex> l(W.m); W.stage_busy[j.stage] = false; }
                W.cv.notify_all();
            }
        };
        std::vector<std::thread> threads, parsers, writers;
        if (behind || positional)
            for (int w = 0; w < n_workers; w++) {
                if (tksmseq_clone(workers[(size_t)w]->ctx, &workers[(size_t)w]->wctx)) { set_error(std::string("writer context: ") + tksmseq_last_error(workers[(size_t)w]->ctx)); break; }
                writers.emplace_back(write_behind, w);
            }
        for (int w = 0; w < n_workers; w++) threads.emplace_back(work, w);
        for (int pi = 0; pi < n_groups * parsers_per_group; pi++) parsers.emplace_back(parse_ahead, pi);
        std::thread writer;
        if (!positional && !behind) writer = std::thread(write_all);

        auto read_full = [&](char* dst, size_t n) -> size_t {          // like fread(dst, 1, n, in): n bytes unless the input ends -- or the run has failed
            size_t got = 0;
            while (got < n && !failed) {
                if (!in_regular) {
                    struct pollfd pf; pf.fd = in_fd; pf.events = POLLIN; pf.revents = 0;
                    const int pr = poll(&pf, 1, 200);
                    if (pr == 0) continue;
                    if (pr < 0) { if (errno == EINTR) continue; break; }
                }
                const ssize_t r = ::read(in_fd, dst + got, std::min<size_t>(n - got, (size_t)1 << 30));
                if (r < 0) { if (errno == EINTR) continue; break; }
                if (r == 0) break;
                got += (size_t)r;
            }
            return got;
        };
        // reader: batches of whole molecules, numbered; the first read index of a batch is known before it is parsed
        std::vector<char> buf;
        uint64_t read_index = 0, seq = 0;
        bool eof = false;
        size_t have = 0, scan_floor = 0;
        if (a.pcr_on) {
            // chained PCR (src/pcr.cpp:215: the module holds its whole input): the templates go to every device group once; the
            // copies per template (tksmseq_pcr_template_counts) cut them into slices of about --pcr-slice-molecules copies, which
            // the parser threads amplify (and truncate) in place of parsing text
            std::vector<char> all;
            { std::vector<char> tmp(1 << 20); size_t n2; while ((n2 = read_full(tmp.data(), tmp.size())) > 0) all.insert(all.end(), tmp.begin(), tmp.begin() + (ptrdiff_t)n2); }
            for (int g = 0; g < n_groups && !failed; g++)
                if (tksmseq_molecules_from_mdf_text(pctx[(size_t)g * parsers_per_group], all.data(), all.size(), &templates[(size_t)g])) set_error(tksmseq_last_error(pctx[(size_t)g * parsers_per_group]));
            uint64_t nt = 0;
            std::vector<uint64_t> counts;
            if (!failed) {
                tksmseq_batch_info(templates[0], &nt, nullptr, nullptr);
                counts.resize(nt);
                if (tksmseq_pcr_template_counts(pctx[0], templates[0], &a.pcr, counts.data())) set_error(tksmseq_last_error(pctx[0]));
            }
            uint64_t u0 = 0, acc = 0;
            auto emit_slice = [&](uint64_t b0, uint64_t e0, uint64_t n_out) {
                Chunk c;
                c.seq = seq++; c.first_read = read_index; c.n_reads = n_out; c.t_begin = b0; c.t_end = e0;
                read_index += n_out;
                queue.push(std::move(c));
            };
            for (uint64_t u = 0; u < nt && !failed; u++) {
                acc += counts[u];
                if (acc >= a.pcr_slice && u + 1 < nt) { emit_slice(u0, u + 1, acc); u0 = u + 1; acc = 0; }
            }
            if (!failed) emit_slice(u0, nt, acc);                   // the last slice (the only, empty one of an input without molecules)
            eof = true; have = 0;
        }
        while ((!eof || have) && !failed) {
            // fill up to batch_bytes, then cut at the last molecule header so a batch holds whole molecules
            const auto t_read = now();
            buf.resize(have + a.batch_bytes);
            size_t got = eof ? 0 : read_full(buf.data() + have, a.batch_bytes);
            if (got < a.batch_bytes) eof = true;
            have += got;
            size_t cut = have;
            if (!eof) {
                cut = tkmod::last_molecule_boundary(buf.data(), have, scan_floor);
                if (!cut) { scan_floor = have ? have - 1 : 0; buf.resize(have); continue; }   // no boundary yet: read more (what was scanned is not scanned again)
            }
            if (cut == 0) break;
            Chunk c;
            c.seq = seq++; c.first_read = read_index;
            c.text.assign(buf.begin(), buf.begin() + (ptrdiff_t)cut);
            c.n_reads = count_reads(c.text.data(), c.text.size());
            read_index += c.n_reads;
            bytes_in += c.text.size();
            add_clk(5, t_read);
            queue.push(std::move(c));
            memmove(buf.data(), buf.data() + cut, have - cut);
            have -= cut;
            scan_floor = have ? have - 1 : 0;                       // (the cut was the last boundary: the rest holds none)
        }
        { std::lock_guard<std::mutex> l(done_m); n_batches = seq; reader_done = true; }
        queue.close();
        done_cv.notify_all();
        for (auto& t : parsers) t.join();
        for (auto& q2 : pq) q2->close();                                                           // (the workers take what is still queued)
        for (auto& t : threads) t.join();
        for (auto& W : workers) { { std::lock_guard<std::mutex> l(W->m); W->jobs_closed = true; } W->cv.notify_all(); }
        for (auto& t : writers) t.join();
        for (int g = 0; g < n_groups; g++) if (templates[(size_t)g]) tksmseq_batch_free(pctx[(size_t)g * parsers_per_group], templates[(size_t)g]);
        done_cv.notify_all();
        if (writer.joinable()) writer.join();
        if ((positional || behind) && !failed) { wb.wrote = wb.wrote || place[0] != 0; wp.wrote = wp.wrote || place[1] != 0; }
        int status = failed ? 1 : 0;
        if (status) fprintf(stderr, "Error: %s\n", first_error.c_str());
        const double t_stream = std::chrono::duration<double>(now() - t_start).count();        // first chunk read -> last record byte written
        if (const char* sf = getenv("TKSMSEQ_STATS_FILE")) {
            // machine-readable stage clocks of this run (bench.py's end-to-end leg): seconds are summed over the threads of a stage
            if (FILE* f = fopen(sf, "w")) {
                const uint64_t out_b = place[0] + place[1];
                fprintf(f, "{\"reads\": %llu, \"batches\": %llu, \"workers\": %d, \"parsers\": %d, \"parse_threads\": %d, \"mdf_bytes\": %llu, "
                           "\"record_bytes\": %llu, \"d2h_bytes\": %llu, \"setup_s\": %.4f, \"stream_s\": %.4f, \"parse_s\": %.4f, \"run_s\": %.4f, "
                           "\"device_copy_s\": %.4f, \"d2h_wait_s\": %.4f, \"write_s\": %.4f, \"wait_for_writer_s\": %.4f, \"read_count_s\": %.4f, "
                           "\"written_behind\": %s, \"positional\": %s, \"status\": %d}\n",
                        (unsigned long long)total_reads, (unsigned long long)seq, n_workers, n_groups * parsers_per_group, a.threads,
                        (unsigned long long)bytes_in.load(), (unsigned long long)out_b, (unsigned long long)bytes_d2h.load(),
                        std::chrono::duration<double>(t_start - t_begin).count(), t_stream, clk[0], clk[1], clk[2], clk[6], clk[3], clk[4], clk[5],
                        behind ? "true" : "false", positional ? "true" : "false", status);
                fclose(f);
            }
        }
        if (verbose)
            fprintf(stderr, "[sequence] %d batches, %d in flight, %.2f s streaming: parse %.2f, run %.2f, copy %.2f, wait for writer %.2f "
                            "(summed over workers); waiting for device-to-host pieces %.2f, write %.2f (summed over writers); read + count %.2f\n", (int)seq, n_workers,
                    t_stream, clk[0], clk[1], clk[2], clk[4], clk[6], clk[3], clk[5]);
        for (auto& W : workers) {
            tksmseq_host_free(W->host[0]); tksmseq_host_free(W->host[1]); tksmseq_host_free(W->ring[0]); tksmseq_host_free(W->ring[1]);
            if (W->wctx) { tksmseq_device_free(W->wctx, W->stage[0]); tksmseq_device_free(W->wctx, W->stage[1]); tksmseq_destroy(W->wctx); W->wctx = nullptr; }
        }
        const auto t_close = now();
        fclose(in);
        if ((!wb.close() || !wp.close()) && !status) { status = 1; fprintf(stderr, "Error: write failed\n"); }
        const auto t_destroy = now();
        destroy_all();
        if (verbose)
            fprintf(stderr, "[sequence] closing the outputs %.2f s, releasing the device %.2f s\n",
                    std::chrono::duration<double>(t_destroy - t_close).count(), std::chrono::duration<double>(now() - t_destroy).count());
        if (!status) log.log(Logger::INFO, "Sequencing: %llu reads, %llu record bytes, %.2f s streaming (%.2f M reads/s)", (unsigned long long)total_reads,
                             (unsigned long long)(place[0] + place[1]), t_stream, t_stream > 0 ? total_reads / t_stream / 1e6 : 0.0);
        return status;
    }
};

Sequencer_module::Sequencer_module(int argc, char** argv) : pimpl{std::make_unique<impl>(argc, argv)} {}
Sequencer_module::~Sequencer_module() = default;
int Sequencer_module::run() { return pimpl->run(); }

extern "C" int tksmseq_sequence_main(int argc, char** argv) { return Sequencer_module{argc, argv}.run(); }
