// sequencer_module.cpp -- the `tksm sequence` module on top of the C-ABI.
//
// Mirrors (file:line into vpc-ccg/tksm):
//   Sequencer_module::impl::run     src/sequence.cpp:30-54   ($TKSM_MODELS handling, then the sequencer)
//   parse_args                      py/sequence.py:34-165    (flags, defaults, validation texts, exit codes)
//   main block                      py/sequence.py:323-376   (load reference + models, stream the MDF, write)
//   get_output_file                 py/sequence.py:291-300   (extension decides FASTQ/FASTA; .gz ok)
//   utility flags                   src/module.h:75-104      (-s/--seed default 42, --verbosity, --log-file)
// Exit codes: 0 ok; 1 for `sys.exit("msg")`-style validation and runtime errors; 2 for argparse
// usage errors (missing -i, neither -o nor --perfect) -- what the embedded interpreter returns.
#include "sequencer_module.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tksmseq.h"

namespace {

struct Args {
    std::string input, badread, perfect, output_format, identity = "84.0,99.0,5.5";
    std::string error_model = "nanopore2020", qscore_model = "nanopore2020", tail_model = "no_noise";
    std::vector<std::string> references;
    bool skip_qual = false, list = false, help = false;
    int threads = 1, device = 0;
    long long seed = 42;
    uint64_t batch_bytes = 256ull << 20;
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
            "                [-s SEED] [--devices D] [--verbosity L] [--log-file F]\n");
}

struct Writer {
    FILE* f = nullptr; gzFile g = nullptr; bool fastq = false;
    bool open(const std::string& path) {                 // get_output_file, py/sequence.py:291-300
        std::string p = path;
        if (p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0) { g = gzopen(path.c_str(), "wb"); p.resize(p.size() - 3); if (!g) return false; }
        else { f = fopen(path.c_str(), "wb"); if (!f) return false; }
        auto ends = [&](const char* s) { size_t n = strlen(s); return p.size() >= n && p.compare(p.size() - n, n, s) == 0; };
        fastq = ends(".fastq") || ends(".fq");
        return true;
    }
    bool write(const uint8_t* d, size_t n) {
        while (n) {
            size_t c = n > (1u << 30) ? (1u << 30) : n;
            if (g) { if (gzwrite(g, d, (unsigned)c) != (int)c) return false; }
            else if (fwrite(d, 1, c, f) != c) return false;
            d += c; n -= c;
        }
        return true;
    }
    void close() { if (g) gzclose(g); if (f) fclose(f); g = nullptr; f = nullptr; }
};

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
            else if (o == "--devices") { if (!(v = need(i))) return 2; a.device = atoi(v); }
            else if (o == "--batch-bytes") { if (!(v = need(i))) return 2; a.batch_bytes = strtoull(v, nullptr, 10); }
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
        if (a.tail_model != "no_noise") return die("Error: tail-noise models other than no_noise are not supported by this build yet");

        tksmseq_ctx* ctx = nullptr;
        if (tksmseq_create(a.device, &ctx)) return die(std::string("Error: ") + tksmseq_last_error(nullptr));
        auto fail = [&](const std::string& what) { std::string m = "Error: " + what + ": " + tksmseq_last_error(ctx); tksmseq_destroy(ctx); return die(m); };
        for (auto& r : a.references) {
            printf("Loading reference %s...\n", r.c_str());
            if (tksmseq_reference_add_fasta(ctx, r.c_str())) return fail("loading reference");
        }
        Writer wb, wp;
        bool compute_q = false;
        if (!a.badread.empty()) {
            if (tksmseq_set_identity(ctx, mean, maxi, sd)) return fail("identity distribution");
            fprintf(stderr, "\nLoading error model from %s\n", a.error_model.c_str());
            if (tksmseq_load_error_model(ctx, a.error_model.c_str())) return fail("error model");
            if (!wb.open(a.badread)) { tksmseq_destroy(ctx); return die("Error: cannot open " + a.badread); }
            compute_q = !a.skip_qual && wb.fastq;
            if (compute_q) {
                fprintf(stderr, "\nLoading qscore model from %s\n", a.qscore_model.c_str());
                if (tksmseq_load_qscore_model(ctx, a.qscore_model.c_str())) return fail("qscore model");
            }
        }
        if (!a.perfect.empty() && !wp.open(a.perfect)) { tksmseq_destroy(ctx); return die("Error: cannot open " + a.perfect); }
        if (!a.badread.empty() && !a.perfect.empty())
            fprintf(stderr, "note: with both -o and --perfect the reference writes the badread sequence (quals 'K') to the "
                            "--perfect file (py/sequence.py:317-319); reproduced here\n");

        FILE* in = fopen(a.input.c_str(), "rb");
        if (!in) { tksmseq_destroy(ctx); return die("Error: cannot open " + a.input); }
        std::vector<char> buf;
        std::vector<uint8_t> rec;
        uint64_t read_index = 0, total_reads = 0;
        bool eof = false;
        size_t have = 0;
        int status = 0;
        while (!eof || have) {
            // fill up to batch_bytes, then cut at the last molecule header so a batch holds whole molecules
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
            tksmseq_batch* b = nullptr;
            if (tksmseq_batch_from_mdf_text(ctx, buf.data(), cut, &b)) { status = 1; fprintf(stderr, "Error: %s\n", tksmseq_last_error(ctx)); break; }
            uint64_t n = 0;
            tksmseq_batch_info(b, &n, nullptr, nullptr);
            auto emit = [&](Writer& w, int mode, int quirk) -> bool {
                tksmseq_run_params p{};
                p.seed = (uint64_t)a.seed; p.first_read_index = read_index; p.read_index_stride = 1;
                p.mode = mode; p.fastq = w.fastq; p.compute_qual = compute_q; p.perfect_of_badread = quirk;
                tksmseq_result r{};
                if (tksmseq_run(ctx, b, &p, &r)) return false;
                rec.resize(r.records_bytes);
                if (tksmseq_result_download(ctx, rec.data(), nullptr)) return false;
                return w.write(rec.data(), rec.size());
            };
            bool ok = true;
            if (n) {
                if (!a.badread.empty()) ok = emit(wb, TKSMSEQ_MODE_BADREAD, 0);
                if (ok && !a.perfect.empty()) ok = a.badread.empty() ? emit(wp, TKSMSEQ_MODE_PERFECT, 0) : emit(wp, TKSMSEQ_MODE_BADREAD, 1);
            }
            tksmseq_batch_free(ctx, b);
            if (!ok) { status = 1; fprintf(stderr, "Error: %s\n", tksmseq_last_error(ctx)); break; }
            read_index += n; total_reads += n;
            memmove(buf.data(), buf.data() + cut, have - cut);
            have -= cut;
        }
        fclose(in);
        wb.close(); wp.close();
        tksmseq_destroy(ctx);
        if (!status) fprintf(stderr, "Sequencing: %llu reads\n", (unsigned long long)total_reads);
        return status;
    }
};

Sequencer_module::Sequencer_module(int argc, char** argv) : pimpl{std::make_unique<impl>(argc, argv)} {}
Sequencer_module::~Sequencer_module() = default;
int Sequencer_module::run() { return pimpl->run(); }

extern "C" int tksmseq_sequence_main(int argc, char** argv) { return Sequencer_module{argc, argv}.run(); }
