// mdf_modules.cpp -- the `tksm pcr` and `tksm truncate` modules on top of the C-ABI (MDF file in, MDF file out).
//
// Mirrors (file:line into vpc-ccg/tksm):
//   PCR_module::impl        src/pcr.cpp:91-260       flags -i -o --molecule-count --cycles --error-rate --efficiency -x/--preset,
//                                                    mandatory-argument and preset checks, whole input read into memory (:215)
//   Truncate_module::impl   src/truncate.cpp:236-451 flags -i -o --kde-model --always-end --kde-models-length --normal --lognormal,
//                                                    "exactly one of kde-model, normal or lognormal"
//   utility flags           src/module.h:75-104      -s/--seed (default 42), --verbosity, --log-file, -h
// Exit codes as the reference's run(): 0 ok (also for --help), 1 for missing / inconsistent arguments and runtime errors.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/tksmseq.h"
#include "sequencer_module.h"

namespace {

bool read_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

struct Common { std::string input, output; long long seed = 42; int device = 0; bool help = false; };

// returns 1 if the flag was one of the common ones (i advanced), 0 if not, -1 on a missing value
int common_flag(int argc, char** argv, int& i, Common& c) {
    const std::string o = argv[i];
    auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
    const char* v = nullptr;
    if (o == "-h" || o == "--help") { c.help = true; return 1; }
    if (o == "-i" || o == "--input") { if (!(v = val())) return -1; c.input = v; return 1; }
    if (o == "-o" || o == "--output") { if (!(v = val())) return -1; c.output = v; return 1; }
    if (o == "-s" || o == "--seed") { if (!(v = val())) return -1; c.seed = atoll(v); return 1; }
    if (o == "--devices") { if (!(v = val())) return -1; c.device = atoi(v); return 1; }
    if (o == "--verbosity" || o == "--log-file") { if (!val()) return -1; return 1; }
    return 0;
}

int run_transform(const Common& c, const char* what, int (*apply)(tksmseq_ctx*, const tksmseq_batch*, void*, tksmseq_batch**), void* arg) {
    const bool verbose = getenv("TKSMSEQ_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* name) {
        const auto t = std::chrono::steady_clock::now();
        if (verbose) fprintf(stderr, "[%s] %s %.3f s\n", what, name, std::chrono::duration<double>(t - t_last).count());
        t_last = t;
    };
    std::string text;
    if (!read_file(c.input, text)) { fprintf(stderr, "Could not open file %s\n", c.input.c_str()); return 1; }
    lap("input read");
    tksmseq_ctx* ctx = nullptr;
    if (tksmseq_create(c.device, &ctx)) { fprintf(stderr, "Error: %s\n", tksmseq_last_error(nullptr)); return 1; }
    lap("device ready");
    // host threads for the MDF parser and writer (the reference's modules are single-threaded; results do not depend on it)
    tksmseq_set_host_threads(ctx, (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
    tksmseq_batch *in = nullptr, *out = nullptr;
    char* otext = nullptr; uint64_t olen = 0;
    int rc = tksmseq_molecules_from_mdf_text(ctx, text.data(), text.size(), &in);
    lap("molecules parsed and uploaded");
    if (!rc) rc = apply(ctx, in, arg, &out);
    lap("device pass");
    if (!rc) rc = tksmseq_batch_to_mdf_text(ctx, out, &otext, &olen);
    lap("MDF text written");
    int status = 0;
    if (rc) { fprintf(stderr, "Error: %s: %s\n", what, tksmseq_last_error(ctx)); status = 1; }
    else {
        FILE* f = fopen(c.output.c_str(), "wb");
        if (!f || fwrite(otext, 1, olen, f) != olen || fclose(f)) { fprintf(stderr, "Error: cannot write %s\n", c.output.c_str()); status = 1; }
    }
    lap("output file written");
    tksmseq_text_free(otext);
    if (out) tksmseq_batch_free(ctx, out);
    if (in) tksmseq_batch_free(ctx, in);
    tksmseq_destroy(ctx);
    lap("released");
    return status;
}

int apply_pcr(tksmseq_ctx* ctx, const tksmseq_batch* in, void* arg, tksmseq_batch** out) { return tksmseq_pcr(ctx, in, (const tksmseq_pcr_params*)arg, out); }
int apply_trc(tksmseq_ctx* ctx, const tksmseq_batch* in, void* arg, tksmseq_batch** out) { return tksmseq_truncate(ctx, in, (const tksmseq_trc_params*)arg, out); }

}  // namespace

extern "C" int tksmseq_pcr_main(int argc, char** argv) {
    Common c;
    bool have_count = false, have_cycles = false, have_er = false, have_ef = false;
    std::string preset;
    tksmseq_pcr_params p{};
    for (int i = 1; i < argc; i++) {
        const int k = common_flag(argc, argv, i, c);
        if (k < 0) { fprintf(stderr, "Option '%s' is missing an argument\n", argv[i]); return 1; }
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
    if (c.help) { printf("PCR amplification module\nusage: pcr -i INPUT -o OUTPUT --molecule-count N --cycles C [-x PRESET | --error-rate E --efficiency F] [-s SEED]\n"); return 0; }
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
    p.seed = (uint64_t)c.seed;
    return run_transform(c, "PCR", apply_pcr, &p);
}

extern "C" int tksmseq_truncate_main(int argc, char** argv) {
    Common c;
    tksmseq_trc_params p{};
    std::string kde;
    int n_dist = 0;
    auto two = [](const char* v, double& a, double& b) { char* e = nullptr; a = strtod(v, &e); if (!e || *e != ',') return false; b = strtod(e + 1, &e); return e && !*e; };
    for (int i = 1; i < argc; i++) {
        const int k = common_flag(argc, argv, i, c);
        if (k < 0) { fprintf(stderr, "Option '%s' is missing an argument\n", argv[i]); return 1; }
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
    if (c.help) { printf("Truncate module\nusage: truncate -i INPUT -o OUTPUT (--kde-model M.json [--always-end] [--kde-models-length] | --normal MU,SIGMA | --lognormal MU,SIGMA) [-s SEED]\n"); return 0; }
    int missing = 0;
    if (c.input.empty()) { fprintf(stderr, "input is required!\n"); missing++; }
    if (c.output.empty()) { fprintf(stderr, "output is required!\n"); missing++; }
    if (n_dist == 0) { fprintf(stderr, "One of kde-model, normal or lognormal is required!\n"); missing++; }
    if (n_dist > 1) { fprintf(stderr, "Only one of kde-model, normal or lognormal is allowed!\n"); missing++; }
    if (missing) return 1;
    if (!kde.empty()) { p.mode = TKSMSEQ_TRC_KDE; p.kde_model_path = kde.c_str(); }
    p.seed = (uint64_t)c.seed;
    return run_transform(c, "truncate", apply_trc, &p);
}
