// mdf_kernels.h -- device side of the molecule-description transforms upstream of Seq in BASELINE config 5:
// PCR amplification (src/pcr.cpp:22-89) and truncation (src/truncate.cpp:23-65, :77-227, :322-351).
// Both read a molecule batch in the binary layout of include/tksmseq.h and write a new one, on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace tk {

constexpr int PCR_MAX_CYCLES = 56;        // path masks are 56-bit (the RNG counter carries them)
constexpr int PCR_MAX_MUT = 32;           // mutations of one copy event (rate x length is ~0.3 for Taq on 1 kb)

struct PcrParams {
    uint64_t seed;
    int cycles;
    double efficiency, rate;              // rate = 4 * error_rate / 3 (src/pcr.cpp:36)
    double drop;                          // target / ((1 + efficiency)^cycles * molecules)
    double q[PCR_MAX_CYCLES + 1];         // q[t]: P(no molecule is emitted in the subtree of an existing copy made in cycle t)
    double A[PCR_MAX_CYCLES + 2];         // A[t]: P(none of the copies made in cycles t .. cycles-1 from one template leads to an emission)
};

// input batch plus what PCR / truncation need beyond the Seq view
struct MolView {
    BatchView B;
    const uint32_t* dup;                  // [n_reads] bit 31: the molecule had depth > 1, bits 0..30: its index among the copies (MDF unroll, src/mdf.h:97-105); may be null
    uint64_t n_intervals, n_mods;
    const uint32_t* keep;                 // optional [n_kept] molecule indices to amplify (more than 2 x target molecules: src/pcr.cpp:217-220)
    uint64_t n_kept;
};

// output tables (sized by the count pass)
struct MolOut {
    uint32_t* reads; uint32_t* intervals; uint32_t* mods; uint32_t* ids; uint8_t* idpool;
};

// PCR: pass 1 counts the emitted copies of every template; pass 2 lists them (template, path mask) with their sizes;
// pass 3 writes the molecules
hipError_t launch_pcr_count(const MolView& m, const PcrParams& p, uint64_t* n_out /* [n_kept] */, uint32_t* status, hipStream_t s);
hipError_t launch_pcr_list(const MolView& m, const PcrParams& p, const uint64_t* out_off /* [n_kept + 1] */, uint32_t* node_mol,
                           uint64_t* node_mask, uint64_t* node_ivls, uint64_t* node_mods, uint64_t* node_idlen, hipStream_t s);
hipError_t launch_pcr_write(const MolView& m, const PcrParams& p, uint64_t n_nodes, const uint32_t* node_mol, const uint64_t* node_mask,
                            const uint64_t* ivl_off, const uint64_t* mod_off, const uint64_t* id_off, const MolOut& o, hipStream_t s);

// truncation
struct TrcParams {
    uint64_t seed;
    int mode;                             // 0 normal(mu, sigma), 1 lognormal(mu, sigma), 2 KDE model
    double mu, sigma;
    int min_len;                          // truncate()'s min_val (100)
    // KDE model (custom_distribution2D / custom_distribution, src/truncate.cpp:77-227)
    int nx, ny;                           // x labels (bins), y labels (rows)
    const long long* xlab; const long long* ylab;
    const double* cdf;                    // [ny][nx + 1] running sums of row i's first i + 1 entries (the rest repeat the last)
    const int* row_n;                     // [ny] entries of row i that belong to its distribution (i + 1, at most nx)
    int have_sider, ns;                   // end_mtx: which share of the truncation goes to the 3' end
    const double* slab; const double* scdf;   // [ns] labels, [ns + 1] running sums
    int always_end, models_length;
};
// per molecule: the kept part [cut5, total - cut3) of its bases in segment order, plus what the TR comment shows
hipError_t launch_trc_plan(const MolView& m, const TrcParams& p, uint64_t first_index, uint32_t* keep_from, uint32_t* keep_to,
                           double* tr_len, double* tr_side, uint64_t* n_ivls, uint64_t* n_mods, uint64_t* n_idlen, hipStream_t s);
hipError_t launch_trc_write(const MolView& m, const uint32_t* keep_from, const uint32_t* keep_to, const uint64_t* ivl_off,
                            const uint64_t* mod_off, const uint64_t* id_off, const MolOut& o, hipStream_t s);

}  // namespace tk
