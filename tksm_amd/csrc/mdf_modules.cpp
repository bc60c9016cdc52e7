// mdf_modules.cpp -- the `tksm pcr` and `tksm truncate` modules on top of the C-ABI (MDF file in, MDF file out).
//
// Mirrors (file:line into vpc-ccg/tksm):
//   PCR_module::impl        src/pcr.cpp:91-260       flags -i -o --molecule-count --cycles --error-rate --efficiency -x/--preset,
//                                                    mandatory-argument and preset checks; the whole input is held (:215: the drop
//                                                    ratio needs the number of templates), the output is streamed
//   Truncate_module::impl   src/truncate.cpp:236-451 flags -i -o --kde-model --always-end --kde-models-length --normal --lognormal,
//                                                    "exactly one of kde-model, normal or lognormal"; a stream transform (:322-351)
//   utility flags           src/module.h:75-104      -s/--seed (default 42), --verbosity, --log-file, -h
// Both stream: `truncate` reads the input in batches of whole molecules (--batch-bytes), `pcr` amplifies its templates in slices
// of about --slice-molecules output molecules (tksmseq_pcr_params::template_begin / _end); the pieces go round the entries of
// --devices D[,D...] (two contexts per entry: one formats its text while the other computes) and are written in input order, so the
// output does not depend on the device list or the piece size.
// Exit codes as the reference's run(): 0 ok (also for --help), 1 for missing / inconsistent arguments and runtime errors.
#include <deque>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/tksmseq.h"
#include "module_log.h"
#include "sequencer_module.h"

namespace {

using tkmod::Logger;

bool read_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

struct Common {
    std::string input, output, verbosity = "INFO", log_file = "stderr";
    long long seed = 42;
    std::vector<int> devices{0};
    uint64_t batch_bytes = 64ull << 20;          // truncate: MDF text per batch
    uint64_t slice_molecules = 2000000;          // pcr: output molecules per slice of templates
    bool help = false;
};

// returns 1 if the flag was one of the common ones (i advanced), 0 if not, -1 on a missing / malformed value
int common_flag(int argc, char** argv, int& i, Common& c) {
    const std::string o = argv[i];
    auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
    const char* v = nullptr;
    if (o == "-h" || o == "--help") { c.help = true; return 1; }
    if (o == "-i" || o == "--input") { if (!(v = val())) return -1; c.input = v; return 1; }
    if (o == "-o" || o == "--output") { if (!(v = val())) return -1; c.output = v; return 1; }
    if (o == "-s" || o == "--seed") { if (!(v = val())) return -1; c.seed = atoll(v); return 1; }
    if (o == "--devices") { if (!(v = val()) || !tkmod::parse_device_list(v, c.devices)) return -1; return 1; }
    if (o == "--verbosity") { if (!(v = val())) return -1; c.verbosity = v; return 1; }
    if (o == "--log-file") { if (!(v = val())) return -1; c.log_file = v; return 1; }
    if (o == "--batch-bytes") { if (!(v = val()) || strtoull(v, nullptr, 10) < 1) return -1; c.batch_bytes = strtoull(v, nullptr, 10); return 1; }
    if (o == "--slice-molecules") { if (!(v = val()) || strtoull(v, nullptr, 10) < 1) return -1; c.slice_molecules = strtoull(v, nullptr, 10); return 1; }
    return 0;
}

bool open_log(const Common& c, const char* module, Logger& log) {
    log.module = module;
    const int lv = Logger::parse(c.verbosity);
    if (lv < 0) { fprintf(stderr, "Error: unknown verbosity level '%s' (choose from DEBUG, INFO, WARN, ERROR, OFF)\n", c.verbosity.c_str()); return false; }
    log.level = lv;
    if (!log.open(c.log_file)) { fprintf(stderr, "Error: cannot open log file %s\n", c.log_file.c_str()); return false; }
    return true;
}

// pieces of output text, written in the order of their numbers whatever the order in which they are finished
struct OrderedOut {
    FILE* f = nullptr; std::mutex m; std::condition_variable cv; uint64_t next = 0; bool failed = false;
    bool put(uint64_t k, const char* text, uint64_t len) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return next == k || failed; });
        if (failed) return false;
        if (len && fwrite(text, 1, len, f) != len) failed = true;
        next++;
        cv.notify_all();
        return !failed;
    }
    void fail() { std::lock_guard<std::mutex> l(m); failed = true; cv.notify_all(); }
};

struct Piece { uint64_t seq = 0, first = 0, begin = 0, end = 0; std::vector<char> text; };   // truncate: text + first molecule index; pcr: template slice

// The common engine: `n_ctx` worker threads (two per entry of --devices), each with a context of its own; prepare() runs once per
// context (pcr: parse the templates), pieces come from next_piece() (serialised), work() turns one into a batch, whose MDF text is
// written in piece order.
template <class Prepare, class Next, class Work>
int run_pieces(const Common& c, Logger& log, const char* what, Prepare prepare, Next next_piece, Work work) {
    OrderedOut out;
    out.f = fopen(c.output.c_str(), "wb");
    if (!out.f) { fprintf(stderr, "Error: cannot write %s\n", c.output.c_str()); return 1; }
    const int per_device = 2, n_ctx = (int)c.devices.size() * per_device;
    std::mutex err_m, next_m; std::string first_error; std::atomic<bool> failed{false};
    auto set_error = [&](const std::string& e) { std::lock_guard<std::mutex> l(err_m); if (!failed.exchange(true)) first_error = e; out.fail(); };
    std::atomic<uint64_t> molecules{0};
    const auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int wi) {
        tksmseq_ctx* ctx = nullptr;
        if (tksmseq_create(c.devices[(size_t)(wi / per_device)], &ctx)) { set_error(tksmseq_last_error(nullptr)); return; }
        // host threads for the MDF parser and writer (the reference's modules are single-threaded; results do not depend on it)
        tksmseq_set_host_threads(ctx, (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency() / (unsigned)std::max(1, n_ctx / 2))));
        void* state = nullptr;
        if (!prepare(ctx, &state)) set_error(tksmseq_last_error(ctx));
        while (!failed) {
            Piece p;
            { std::lock_guard<std::mutex> l(next_m); if (failed || !next_piece(p)) break; }
            tksmseq_batch* b = nullptr;
            char* text = nullptr; uint64_t len = 0;
            int rc = work(ctx, state, p, &b);
            if (!rc) rc = tksmseq_batch_to_mdf_text(ctx, b, &text, &len);
            if (rc) set_error(std::string(what) + ": " + tksmseq_last_error(ctx));
            else {
                uint64_t n = 0;
                tksmseq_batch_info(b, &n, nullptr, nullptr);
                molecules += n;
                log.log(Logger::DEBUG, "piece %llu: %llu molecules, %.1f MB of text (context %d)", (unsigned long long)p.seq, (unsigned long long)n, len / 1e6, wi);
                if (!out.put(p.seq, text, len) && !failed) set_error("cannot write " + c.output);
            }
            tksmseq_text_free(text);
            if (b) tksmseq_batch_free(ctx, b);
        }
        if (state) prepare(ctx, &state);                        // (second call: releases what the first one made)
        tksmseq_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int w = 0; w < n_ctx; w++) th.emplace_back(worker, w);
    for (auto& t : th) t.join();
    const bool close_ok = fclose(out.f) == 0;
    if (failed) { fprintf(stderr, "Error: %s\n", first_error.c_str()); return 1; }
    if (!close_ok || out.failed) { fprintf(stderr, "Error: cannot write %s\n", c.output.c_str()); return 1; }
    log.log(Logger::INFO, "%s: %llu molecules written in %.2f s (%d device group(s))", what, (unsigned long long)molecules.load(),
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), (int)c.devices.size());
    return 0;
}

}  // namespace

extern "C" int tksmseq_pcr_main(int argc0, char** argv0) {
    std::vector<std::string> arg_store; std::vector<char*> arg_ptrs;
    tkmod::split_equals(argc0, argv0, arg_store, arg_ptrs);
    const int argc = (int)arg_ptrs.size(); char** const argv = arg_ptrs.data();
    Common c;
    bool have_count = false, have_cycles = false, have_er = false, have_ef = false;
    std::string preset;
    tksmseq_pcr_params p{};
    for (int i = 1; i < argc; i++) {
        const int k = common_flag(argc, argv, i, c);
        if (k < 0) { fprintf(stderr, "Option '%s' is missing an argument or has a malformed one\n", argv[i]); return 1; }
        if (k) continue;
        const std::string o = argv[i];
        const char* v = i + 1 < argc ? argv[i + 1] : nullptr;
        if (o == "--molecule-count" && v) { p.target_count = strtoull(v, nullptr, 10); have_count = true; i++; }
        else if (o == "--cycles" && v) { p.cycles = atoi(v); have_cycles = true; i++; }
        else if (o == "--error-rate" && v) { p.error_rate = atof(v); have_er = true; i++; }
        else if (o == "--efficiency" && v) { p.efficiency = atof(v); have_ef = true; i++; }
        else if ((o == "-x" || o == "--preset") && v) { preset = v; i++; }
        else { fprintf(stderr, "Option '%s' does not exist or is missing an argument\n", argv[i]); return 1; }
    }
    if (c.help) { printf("PCR amplification module\nusage: pcr -i INPUT -o OUTPUT --molecule-count N --cycles C [-x PRESET | --error-rate E --efficiency F] [-s SEED]\n"
                         "           [--devices D[,D...]] [--slice-molecules N] [--verbosity L] [--log-file F]\n"); return 0; }
    int missing = 0;
    if (c.input.empty()) { fprintf(stderr, "input is required!\n"); missing++; }
    if (c.output.empty()) { fprintf(stderr, "output is required!\n"); missing++; }
    if (!have_count) { fprintf(stderr, "molecule-count is required!\n"); missing++; }
    if (!have_cycles) { fprintf(stderr, "cycles is required!\n"); missing++; }
    if (!preset.empty()) {
        double er = 0, ef = 0;
        if (tksmseq_pcr_preset(preset.c_str(), &er, &ef)) { fprintf(stderr, "Preset %s not found\n", preset.c_str()); missing++; }
        else { if (!have_er) p.error_rate = er; if (!have_ef) p.efficiency = ef; }       // explicit values override the preset (src/pcr.cpp:205-210)
    } else {
        if (!have_er) { fprintf(stderr, "Error rate is required!\n"); missing++; }
        if (!have_ef) { fprintf(stderr, "Efficiency is required!\n"); missing++; }
    }
    if (missing) return 1;
    Logger log;
    if (!open_log(c, "pcr", log)) return 1;
    p.seed = (uint64_t)c.seed;
    std::string text;
    if (!read_file(c.input, text)) { fprintf(stderr, "Could not open file %s\n", c.input.c_str()); return 1; }
    // the slices of templates: consecutive runs that write about slice_molecules copies each, from the per-template counts
    std::vector<uint64_t> cuts;                                  // template index where slice k begins; cuts.back() = number of templates
    std::mutex cuts_m; bool cuts_done = false, cuts_failed = false; std::condition_variable cuts_cv;
    uint64_t next_slice = 0;
    auto prepare = [&](tksmseq_ctx* ctx, void** state) -> bool {
        if (*state) { tksmseq_batch_free(ctx, (tksmseq_batch*)*state); *state = nullptr; return true; }
        tksmseq_batch* T = nullptr;
        if (tksmseq_molecules_from_mdf_text(ctx, text.data(), text.size(), &T)) return false;
        *state = T;
        bool mine = false;
        { std::lock_guard<std::mutex> l(cuts_m); if (cuts.empty()) { cuts.push_back(0); mine = true; } }
        if (mine) {
            uint64_t n = 0;
            tksmseq_batch_info(T, &n, nullptr, nullptr);
            std::vector<uint64_t> counts(n);
            const bool ok = tksmseq_pcr_template_counts(ctx, T, &p, counts.data()) == 0;
            std::lock_guard<std::mutex> l(cuts_m);
            uint64_t acc = 0, total = 0;
            for (uint64_t u = 0; u < n && ok; u++) { acc += counts[u]; total += counts[u]; if (acc >= c.slice_molecules && u + 1 < n) { cuts.push_back(u + 1); acc = 0; } }
            cuts.push_back(n);
            cuts_done = true; cuts_failed = !ok;
            cuts_cv.notify_all();
            if (ok) log.log(Logger::INFO, "%llu templates -> %llu molecules in %zu slice(s)", (unsigned long long)n, (unsigned long long)total, cuts.size() - 1);
            return ok;
        }
        std::unique_lock<std::mutex> l(cuts_m);
        cuts_cv.wait(l, [&] { return cuts_done; });
        return !cuts_failed;
    };
    auto next_piece = [&](Piece& pc) -> bool {
        std::lock_guard<std::mutex> l(cuts_m);
        if (next_slice + 1 >= cuts.size()) return false;
        pc.seq = next_slice; pc.begin = cuts[next_slice]; pc.end = cuts[next_slice + 1];
        next_slice++;
        return true;
    };
    auto work = [&](tksmseq_ctx* ctx, void* state, const Piece& pc, tksmseq_batch** out) -> int {
        tksmseq_pcr_params q = p;
        q.template_begin = pc.begin; q.template_end = pc.end;
        if (pc.begin == pc.end) { q.template_begin = 0; q.template_end = 0; q.cycles = 0; }     // (an input without molecules: one empty piece)
        return tksmseq_pcr(ctx, (const tksmseq_batch*)state, &q, out);
    };
    return run_pieces(c, log, "PCR", prepare, next_piece, work);
}

extern "C" int tksmseq_truncate_main(int argc0, char** argv0) {
    std::vector<std::string> arg_store; std::vector<char*> arg_ptrs;
    tkmod::split_equals(argc0, argv0, arg_store, arg_ptrs);
    const int argc = (int)arg_ptrs.size(); char** const argv = arg_ptrs.data();
    Common c;
    tksmseq_trc_params p{};
    std::string kde;
    int n_dist = 0;
    auto two = [](const char* v, double& a, double& b) { char* e = nullptr; a = strtod(v, &e); if (!e || *e != ',') return false; b = strtod(e + 1, &e); return e && !*e; };
    for (int i = 1; i < argc; i++) {
        const int k = common_flag(argc, argv, i, c);
        if (k < 0) { fprintf(stderr, "Option '%s' is missing an argument or has a malformed one\n", argv[i]); return 1; }
        if (k) continue;
        const std::string o = argv[i];
        const char* v = i + 1 < argc ? argv[i + 1] : nullptr;
        if (o == "--kde-model" && v) { kde = v; n_dist++; i++; }
        else if (o == "--always-end") p.always_end = 1;
        else if (o == "--kde-models-length") p.kde_models_length = 1;
        else if (o == "--normal" && v) { if (!two(v, p.mu, p.sigma)) { fprintf(stderr, "--normal needs mu,sigma\n"); return 1; } p.mode = TKSMSEQ_TRC_NORMAL; n_dist++; i++; }
        else if (o == "--lognormal" && v) { if (!two(v, p.mu, p.sigma)) { fprintf(stderr, "--lognormal needs mu,sigma\n"); return 1; } p.mode = TKSMSEQ_TRC_LOGNORMAL; n_dist++; i++; }
        else { fprintf(stderr, "Option '%s' does not exist or is missing an argument\n", argv[i]); return 1; }
    }
    if (c.help) { printf("Truncate module\nusage: truncate -i INPUT -o OUTPUT (--kde-model M.json [--always-end] [--kde-models-length] | --normal MU,SIGMA | --lognormal MU,SIGMA) [-s SEED]\n"
                         "                [--devices D[,D...]] [--batch-bytes B] [--verbosity L] [--log-file F]\n"); return 0; }
    int missing = 0;
    if (c.input.empty()) { fprintf(stderr, "input is required!\n"); missing++; }
    if (c.output.empty()) { fprintf(stderr, "output is required!\n"); missing++; }
    if (n_dist == 0) { fprintf(stderr, "One of kde-model, normal or lognormal is required!\n"); missing++; }
    if (n_dist > 1) { fprintf(stderr, "Only one of kde-model, normal or lognormal is allowed!\n"); missing++; }
    if (missing) return 1;
    Logger log;
    if (!open_log(c, "truncate", log)) return 1;
    if (!kde.empty()) { p.mode = TKSMSEQ_TRC_KDE; p.kde_model_path = kde.c_str(); }
    p.seed = (uint64_t)c.seed;
    // a reader thread of its own cuts the input into pieces of whole molecules and numbers them (input order), a few pieces ahead of the
    // workers: what the engine serialises is a pop from this queue, not the read + scan of a piece.  Its state lives on the heap and is
    // shared with the thread: after an error the module returns without waiting for a reader that may sit in a read() on a pipe nobody
    // closes (the thread is detached and ends with the process).
    struct ReaderState {
        tkmod::ChunkReader rd;
        std::mutex m; std::condition_variable put, get; std::deque<Piece> pieces; bool done = false, stop = false; size_t cap = 3;
    };
    auto rs = std::make_shared<ReaderState>();
    rs->rd.in = fopen(c.input.c_str(), "rb");
    if (!rs->rd.in) { fprintf(stderr, "Could not open file %s\n", c.input.c_str()); return 1; }
    rs->rd.bytes = c.batch_bytes;
    rs->cap = c.devices.size() * 2 + 1;
    std::thread reader([rs]() {
        uint64_t seq = 0, first = 0;
        for (;;) {
            Piece pc;
            if (!rs->rd.next(pc.text)) {
                if (seq) break;
                pc.text.clear();                                  // an empty input still makes an (empty) output
            }
            pc.seq = seq++; pc.first = first;
            first += tkmod::count_reads(pc.text.data(), pc.text.size());
            std::unique_lock<std::mutex> l(rs->m);
            rs->put.wait(l, [&] { return rs->pieces.size() < rs->cap || rs->stop; });
            if (rs->stop) break;
            rs->pieces.push_back(std::move(pc));
            rs->get.notify_one();
        }
        { std::lock_guard<std::mutex> l(rs->m); rs->done = true; }
        rs->get.notify_all();
    });
    auto prepare = [&](tksmseq_ctx*, void**) -> bool { return true; };
    auto next_piece = [&](Piece& pc) -> bool {
        std::unique_lock<std::mutex> l(rs->m);
        rs->get.wait(l, [&] { return !rs->pieces.empty() || rs->done; });
        if (rs->pieces.empty()) return false;
        pc = std::move(rs->pieces.front()); rs->pieces.pop_front();
        rs->put.notify_one();
        return true;
    };
    auto work = [&](tksmseq_ctx* ctx, void*, const Piece& pc, tksmseq_batch** out) -> int {
        tksmseq_batch* in = nullptr;
        int rc = tksmseq_molecules_from_mdf_text(ctx, pc.text.data(), pc.text.size(), &in);
        if (rc) return rc;
        tksmseq_trc_params q = p;
        q.first_molecule_index = pc.first;
        rc = tksmseq_truncate(ctx, in, &q, out);
        tksmseq_batch_free(ctx, in);
        return rc;
    };
    const int rc = run_pieces(c, log, "truncate", prepare, next_piece, work);
    bool finished;
    { std::lock_guard<std::mutex> l(rs->m); rs->stop = true; finished = rs->done; }      // (an error: the reader may be waiting for room, or for input)
    rs->put.notify_all();
    if (rc == 0 || finished) { reader.join(); fclose(rs->rd.in); }
    else reader.detach();
    return rc;
}
