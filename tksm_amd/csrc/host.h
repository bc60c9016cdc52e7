// host.h -- host-side data structures shared by the C-ABI implementation, the loaders and the CLI.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace tkh {

struct ErrorModelHost {
    int type = -1, k = 0, max_alts = 0;
    std::vector<uint32_t> cdf;
    std::vector<uint64_t> alts;
    std::vector<uint8_t> nalts;
};

struct QScoreModelHost {
    int n_slots = 0, kmer_size = 1;
    bool guide_direct = false;          // the device guide table carries q itself for single-entry buckets (api.cpp)
    std::vector<uint64_t> keys;
    std::vector<uint32_t> row_off, row_cnt, cdf_pool;
    std::vector<uint8_t> q_pool;
};

struct IdentityHost {
    bool set = false, constant = true;
    double mean = 0, stdev = 0, max_identity = 0, value = 0, beta_a = 0, beta_b = 0;
    std::vector<double> qtab;
};

// reads sorted by length, ties in read order (what a stable comparison sort gives): two counting passes over 16-bit digits -- a
// million reads in a few milliseconds instead of a tenth of a second
inline void order_by_length(const std::vector<uint32_t>& len, std::vector<uint32_t>& order) {
    const size_t n = len.size();
    order.resize(n);
    std::vector<uint32_t> tmp(n);
    std::vector<uint32_t> cnt(65537);
    auto pass = [&](int shift, const uint32_t* src, uint32_t* dst, bool first) {
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (size_t i = 0; i < n; i++) cnt[((len[first ? i : src[i]] >> shift) & 0xffffu) + 1]++;
        for (size_t d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; i++) { const uint32_t r = first ? (uint32_t)i : src[i]; dst[cnt[(len[r] >> shift) & 0xffffu]++] = r; }
    };
    uint32_t mx = 0;
    for (uint32_t v : len) mx = std::max(mx, v);
    if (mx < 65536u) { pass(0, nullptr, order.data(), true); return; }
    pass(0, nullptr, tmp.data(), true);
    pass(16, tmp.data(), order.data(), false);
}

// host image of a molecule batch in the binary layout of include/tksmseq.h
struct BatchHost {
    std::vector<uint32_t> reads;       // [n][2]
    std::vector<uint32_t> intervals;   // [n][4] (no sentinel)
    std::vector<uint32_t> mods;        // [n][2]
    std::vector<uint64_t> literals;    // [n][2]
    std::vector<uint8_t> literal_pool;
    std::vector<uint32_t> ids;         // [n][2]
    std::vector<uint8_t> id_pool;
    // what the C++ modules of the reference see beyond Seq (PCR, truncation, the MDF writer):
    std::vector<uint32_t> dup;         // [n] bit 31: copy of a depth > 1 molecule, bits 0..30: its index (unroll naming id_i, src/mdf.h:97-105)
    std::vector<uint32_t> comments;    // [n][2] {offset, length} of the header's comment field in comment_pool
    std::vector<char> comment_pool;
};

// truncation model (src/truncate.cpp:77-227; written by py/truncate_kde.py:298-320): KDE_mtx = 2-D empirical distribution of the
// truncation length given the molecule's size, end_mtx = histogram of the share of the truncation that hits the 3' end
struct TrcModelHost {
    std::vector<long long> xlab, ylab;
    std::vector<double> cdf;           // [ny][nx + 1] cumulative sums of row i's first min(i + 1, nx) entries, padded with the last value
    std::vector<int> row_n;            // [ny]
    bool have_sider = false;
    std::vector<double> slab, scdf;    // [ns], [ns + 1]
};
bool load_trc_model(const std::string& path, TrcModelHost& m, std::string& err);

// tail-noise model (KDE_noise_generator, py/tksm_badread.py:886-962): the length sampler's grid and the base chain
struct TailModelHost {
    bool enabled = false;
    std::vector<double> lx, ly;        // tail lengths; fragment-length labels (non-decreasing)
    std::vector<double> cdf;           // [ly][lx] running sums of the normalised grid rows (CustomDist.__init__, :980-986)
    double cum[16] = {};               // running sums of the 4 transition rows (random.choices)
    double ratio = 0.0;
    uint8_t bases[4] = {'A', 'G', 'T', 'C'};
};

bool read_text_file(const std::string& path, std::string& out, std::string& err);
std::string resolve_model(const std::string& name, const char* kind);
void cdf_thresholds(const std::vector<double>& probs, bool residual_to_one, std::vector<uint32_t>& out);
bool load_error_model(const std::string& name_or_path, ErrorModelHost& m, std::string& err);
bool load_qscore_model(const std::string& name_or_path, QScoreModelHost& m, std::string& err);
// name_or_path "no_noise" disables the model
bool load_tail_model(const std::string& name_or_path, TailModelHost& m, std::string& err);
bool make_tail_model(const double* lx, size_t n_lx, const double* ly, size_t n_ly, const double* grid, const double* trans16,
                     double ratio, const uint8_t* bases4, TailModelHost& m, std::string& err);
bool make_identity(double mean, double max, double stdev, IdentityHost& id, std::string& err);

// FASTA (py/sequence.py:168-186): calls sink(name, sequence) per record, in file order
struct FastaRecord { std::string name, seq; };
bool read_fasta(const std::string& path, std::vector<FastaRecord>& out, std::string& err);

// MDF text (py/sequence.py:197-221) -> binary batch; contig_id(name) returns -1 for literals
struct ContigLookup { virtual int find(const std::string& name) const = 0; virtual ~ContigLookup() {} };
bool parse_mdf(const char* text, uint64_t len, const ContigLookup& contigs, BatchHost& out, std::string& err);
// the same on n_threads host threads (pieces cut at molecule headers)
bool parse_mdf_mt(const char* text, uint64_t len, const ContigLookup& contigs, BatchHost& out, std::string& err, int n_threads);
bool model_available(const std::string& name, const char* kind);   // resolve_model() finds a file

}  // namespace tkh
