// mdf_kernels.hip -- PCR amplification and truncation of molecule descriptions on the device (gfx950).
//
// Reference behaviour restated (file:line into vpc-ccg/tksm):
//   PCR::do_pcr / perform        src/pcr.cpp:40-89     branching amplification, per-copy substitutions, capture by sequencing
//   molecule_descriptor::add_error  src/interval.h:866-874  molecule position -> (segment, offset)
//   truncate()                   src/truncate.cpp:23-65   keep the first L bases in segment order
//   einterval::truncate          src/interval.h:708-735   cut segment: new bounds, substitutions re-based and filtered
//   custom_distribution / 2D     src/truncate.cpp:77-227  empirical samplers of the KDE truncation model
//   truncate_transformer(_kde)   src/truncate.cpp:322-351 3' truncation, then 5' truncation of the flipped molecule
// Integer / byte work, one LANE per molecule: the tables of a molecule are a few dozen bytes, the work per molecule is a short
// serial walk (tree of copies; list of segments).  The reference draws from a sequential Mersenne Twister; here every
// decision has its own Philox counter (template molecule, path of copy cycles, purpose), so the result does not depend on
// the order molecules are processed in, and the CPU oracle (oracle/mdf_ops_oracle.py) reproduces it bit for bit.
#include "mdf_kernels.h"

namespace tk {

#define DEV __device__ __forceinline__

struct Ph4m { uint32_t x, y, z, w; };
enum { ST_PCR_PICK = 16, ST_PCR_EMIT = 17, ST_PCR_CHILD = 18, ST_PCR_MUT = 19, ST_TRC_LEN = 24, ST_TRC_SIDE = 25 };

DEV Ph4m philox_raw(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Ph4m{c0, c1, c2, c3};
}
// a decision about the copy with path `mask` of template molecule u
DEV Ph4m philox_node(uint64_t seed, uint32_t u, uint64_t mask, uint32_t stream, uint32_t n) {
    return philox_raw(seed, u, (uint32_t)mask, stream | ((uint32_t)(mask >> 32) << 8), n);
}
// per-molecule streams (same counter layout as the Seq kernels: read index, stream, n)
DEV Ph4m philox_mol(uint64_t seed, uint64_t g, uint32_t stream, uint32_t n) {
    return philox_raw(seed, (uint32_t)g, (uint32_t)(g >> 32), stream, n);
}
DEV double u01(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }

// nominal size of a segment (ginterval::size, end - start; the Seq kernels clamp to the contig, PCR / Trc do not)
DEV uint32_t seg_size(const uint32_t* iv) { return iv[2] > iv[1] ? iv[2] - iv[1] : 0u; }
DEV uint32_t mol_size(const BatchView& B, uint32_t r) {
    const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    uint32_t t = 0;
    for (uint32_t i = 0; i < ic; i++) t += seg_size(B.intervals + 4ull * (ib + i));
    return t;
}
DEV uint32_t mol_mods(const BatchView& B, uint32_t r) {
    const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    return (B.intervals[4ull * (ib + ic) + 3] & 0x7fffffffu) - (B.intervals[4ull * ib + 3] & 0x7fffffffu);
}
DEV int ndig(uint32_t v) { int d = 1; while (v >= 10) { v /= 10; d++; } return d; }
DEV int put_dec(uint8_t* o, uint32_t v) { const int d = ndig(v); for (int i = d - 1; i >= 0; i--) { o[i] = (uint8_t)('0' + v % 10); v /= 10; } return d; }

// ------------------------------------------------------------------------------------------------
// PCR.  The reference walks the whole tree of copies of a template (every copy made in cycle s is a template in the cycles
// after s: src/pcr.cpp:62-64) and lets each copy be captured by the sequencing with probability drop_ratio (:59).  The tree has
// (1 + efficiency)^cycles nodes, of which a fraction drop_ratio is written; here only the branches that lead to a written
// copy are walked: P(no copy is written in the subtree of an existing copy made in cycle t) = q[t] is known in closed
// form (PcrParams), so "the copy exists AND its subtree writes something" is decided first, and inside such a subtree
// the events {this copy is written, the copy made from it in cycle t leads to a written copy} are drawn one after the other
// conditioned on at least one of them happening.  The distribution of the written set (with its ancestry, hence shared
// substitutions) is the reference's; the cost is proportional to what is written.
// visit(): calls out(mask) for every written copy of template u, in the reference's order (a copy, then the copies made
// from it cycle by cycle: depth first).
// ------------------------------------------------------------------------------------------------
template <class F>
DEV void pcr_walk(const PcrParams& P, uint32_t u, F&& out) {
    // explicit stack: node mask, next cycle to try, "an emission has already happened in this subtree"
    unsigned long long smask[PCR_MAX_CYCLES + 1]; int snext[PCR_MAX_CYCLES + 1]; bool ssat[PCR_MAX_CYCLES + 1];
    int sp = 0;
    smask[0] = 0ull; snext[0] = 0; ssat[0] = true;                    // the template itself: nothing required of its subtree
    while (sp >= 0) {
        const unsigned long long R = smask[sp];
        const int t = snext[sp];
        if (t >= P.cycles) { sp--; continue; }
        snext[sp] = t + 1;
        const double pm = P.efficiency * (1.0 - P.q[t]);                // the copy made in cycle t exists and leads to an emission
        const double p = ssat[sp] ? pm : pm / (1.0 - P.A[t]);
        const unsigned long long C = R | (1ull << t);
        if (!(u01(philox_node(P.seed, u, C, ST_PCR_CHILD, 0).x) < p)) continue;
        ssat[sp] = true;
        // enter the copy: is it written itself?
        const double pe = P.drop / (1.0 - P.q[t]);                      // given that its subtree writes something
        const bool emit = u01(philox_node(P.seed, u, C, ST_PCR_EMIT, 0).x) < pe;
        if (emit) out(C);
        sp++;
        smask[sp] = C; snext[sp] = t + 1; ssat[sp] = emit;
    }
}

// substitutions of the copy event that made node `mask` (src/pcr.cpp:44-56): count = floor(rate * size) + Bernoulli(fraction),
// distinct positions (std::sample: in increasing order), bases from "ACTG"
DEV int pcr_mutations(const PcrParams& P, uint32_t u, unsigned long long mask, uint32_t size, uint32_t* pos, uint8_t* base) {
    const double expected = P.rate * (double)size;
    int cnt = (int)expected;
    cnt += u01(philox_node(P.seed, u, mask, ST_PCR_MUT, 0).x) < (expected - (double)cnt) ? 1 : 0;
    cnt = min(min(cnt, PCR_MAX_MUT), (int)size);
    uint32_t attempt = 1;
    for (int j = 0; j < cnt; j++) {
        for (;;) {
            const Ph4m w = philox_node(P.seed, u, mask, ST_PCR_MUT, attempt++);
            const uint32_t p = __umulhi(w.x, size);
            bool dup = false;
            for (int q = 0; q < j; q++) dup |= pos[q] == p;
            if (dup) continue;
            // insertion sort by position
            int q = j;
            while (q > 0 && pos[q - 1] > p) { pos[q] = pos[q - 1]; base[q] = base[q - 1]; q--; }
            pos[q] = p; base[q] = (uint8_t)((0x47544341u >> (8 * (w.y & 3u))) & 0xffu);   // "ACTG"
            break;
        }
    }
    return cnt;
}

DEV uint32_t pcr_template(const MolView& M, uint64_t i) { return M.keep ? M.keep[i] : (uint32_t)i; }

__global__ void k_pcr_count(MolView M, PcrParams P, uint64_t* __restrict__ n_out, uint32_t* __restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.n_kept) return;
    const uint32_t u = pcr_template(M, i);
    unsigned long long n = 0;
    pcr_walk(P, u, [&](unsigned long long) { n++; });
    n_out[i] = n;
    if ((double)mol_size(M.B, u) * P.rate >= (double)PCR_MAX_MUT) atomicOr(status, 1u);
}

// id of a copy: template id [+ "_" + index among the unrolled copies] + "." + cycle for every copy event on its path
DEV uint32_t pcr_id_len(const MolView& M, uint32_t u, unsigned long long mask) {
    uint32_t n = M.B.ids[2 * u + 1];
    if (M.dup && (M.dup[u] >> 31)) n += 1 + ndig(M.dup[u] & 0x7fffffffu);
    for (unsigned long long m = mask; m; m &= m - 1) n += 1 + ndig((uint32_t)__builtin_ctzll(m));
    return n;
}

__global__ void k_pcr_list(MolView M, PcrParams P, const uint64_t* __restrict__ out_off, uint32_t* __restrict__ node_mol,
                           uint64_t* __restrict__ node_mask, uint64_t* __restrict__ node_ivls, uint64_t* __restrict__ node_mods,
                           uint64_t* __restrict__ node_idlen) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.n_kept) return;
    const uint32_t u = pcr_template(M, i);
    const uint32_t size = mol_size(M.B, u), base_mods = mol_mods(M.B, u), ic = M.B.reads[2 * u + 1];
    uint64_t at = out_off[i];
    pcr_walk(P, u, [&](unsigned long long mask) {
        // substitutions accumulated along the path: one count per copy event (prefix of the path)
        uint32_t nm = base_mods;
        unsigned long long pre = 0ull;
        for (unsigned long long m = mask; m; m &= m - 1) {
            pre |= m & (~m + 1ull);
            const double expected = P.rate * (double)size;
            int cnt = (int)expected;
            cnt += u01(philox_node(P.seed, u, pre, ST_PCR_MUT, 0).x) < (expected - (double)cnt) ? 1 : 0;
            nm += (uint32_t)min(min(cnt, PCR_MAX_MUT), (int)size);
        }
        node_mol[at] = u; node_mask[at] = mask; node_ivls[at] = ic; node_mods[at] = nm; node_idlen[at] = pcr_id_len(M, u, mask);
        at++;
    });
}

__global__ void k_pcr_write(MolView M, PcrParams P, uint64_t n_nodes, const uint32_t* __restrict__ node_mol,
                            const uint64_t* __restrict__ node_mask, const uint64_t* __restrict__ ivl_off,
                            const uint64_t* __restrict__ mod_off, const uint64_t* __restrict__ id_off, MolOut O) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_nodes) return;
    const uint32_t u = node_mol[j];
    const unsigned long long mask = node_mask[j];
    const BatchView& B = M.B;
    const uint32_t ib = B.reads[2 * u], ic = B.reads[2 * u + 1];
    const uint32_t size = mol_size(B, u);
    const uint64_t io = ivl_off[j];
    uint64_t mo = mod_off[j];
    O.reads[2 * j] = (uint32_t)io; O.reads[2 * j + 1] = ic;
    // id
    {
        uint8_t* d = O.idpool + id_off[j];
        const uint32_t so = B.ids[2 * u], sl = B.ids[2 * u + 1];
        uint32_t k = 0;
        for (; k < sl; k++) d[k] = B.idpool[so + k];
        if (M.dup && (M.dup[u] >> 31)) { d[k++] = '_'; k += (uint32_t)put_dec(d + k, M.dup[u] & 0x7fffffffu); }
        for (unsigned long long m = mask; m; m &= m - 1) { d[k++] = '.'; k += (uint32_t)put_dec(d + k, (uint32_t)__builtin_ctzll(m)); }
        O.ids[2 * j] = (uint32_t)id_off[j]; O.ids[2 * j + 1] = k;
    }
    // segments with their substitutions: the template's own first, then those of every copy event on the path, oldest first
    // (do_pcr appends to the copy it was handed: src/pcr.cpp:52-56)
    uint32_t cum = 0;
    for (uint32_t i = 0; i < ic; i++) {
        const uint32_t* iv = B.intervals + 4ull * (ib + i);
        const uint32_t sz = seg_size(iv);
        uint32_t* ov = O.intervals + 4ull * (io + i);
        ov[0] = iv[0]; ov[1] = iv[1]; ov[2] = iv[2]; ov[3] = (uint32_t)mo | (iv[3] & 0x80000000u);
        const uint32_t mb = iv[3] & 0x7fffffffu, me = iv[7] & 0x7fffffffu;
        for (uint32_t q = mb; q < me; q++) { O.mods[2 * mo] = B.mods[2ull * q]; O.mods[2 * mo + 1] = B.mods[2ull * q + 1]; mo++; }
        unsigned long long pre = 0ull;
        for (unsigned long long m = mask; m; m &= m - 1) {
            pre |= m & (~m + 1ull);
            uint32_t pos[PCR_MAX_MUT]; uint8_t base[PCR_MAX_MUT];
            const int cnt = pcr_mutations(P, u, pre, size, pos, base);
            for (int q = 0; q < cnt; q++)
                // add_error: the segment whose cumulative size first exceeds the position (empty segments are skipped)
                if (pos[q] >= cum && pos[q] < cum + sz) { O.mods[2 * mo] = pos[q] - cum; O.mods[2 * mo + 1] = base[q]; mo++; }
        }
        cum += sz;
    }
}

// ------------------------------------------------------------------------------------------------
// truncation
// ------------------------------------------------------------------------------------------------
// double -> int as the reference's implicit conversion at the call truncate(md, <double>) does it (toward zero); clamped
DEV int to_int(double v) { return v >= 2147483647.0 ? 2147483647 : (v <= -2147483648.0 ? (-2147483647 - 1) : (int)v); }

// truncate() as a window computation: what [0, size) shrinks to when the first L bases are kept.  Returns the new size;
// cut = false when the call changes nothing (L == size, or fewer bases than L).
DEV int trc_keep(int size, int L, int min_val, bool& cut) {
    cut = false;
    if (L == size) return size;
    if (min_val > L) L = min_val;
    if (size < L) return size;
    cut = true;                                                        // (L == size here: a cut that removes nothing)
    return L;
}

// first index with cdf[idx] >= u (std::lower_bound), cdf has n entries
DEV int lower_bound_d(const double* cdf, int n, double u) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

// custom_distribution<double, long>::operator()(g, u) on row `row` of the 2-D model: bin by the row's cumulative sums,
// then a uniform integer in [previous label (0 for the first bin), label] (src/truncate.cpp:128-134, :107-114)
DEV double trc_row_draw(const TrcParams& T, int row, double u, uint32_t w) {
    const double* cdf = T.cdf + (size_t)row * (T.nx + 1);
    int bin = lower_bound_d(cdf, T.row_n[row] + 1, u) - 1;
    bin = max(0, min(bin, T.nx - 1));
    const long long lo = bin == 0 ? 0ll : T.xlab[bin - 1], hi = T.xlab[bin];
    const unsigned long long span = (unsigned long long)(hi - lo) + 1ull;
    return (double)(lo + (long long)(((unsigned long long)w * span) >> 32));
}

__global__ void k_trc_plan(MolView M, TrcParams T, uint64_t first_index, uint32_t* __restrict__ keep_from, uint32_t* __restrict__ keep_to,
                           double* __restrict__ tr_len, double* __restrict__ tr_side, uint64_t* __restrict__ n_ivls,
                           uint64_t* __restrict__ n_mods, uint64_t* __restrict__ n_idlen) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M.B.n_reads) return;
    const BatchView& B = M.B;
    const int size = (int)mol_size(B, (uint32_t)r);
    const uint64_t g = first_index + r;
    int w0 = 0, w1 = size;
    bool cut3 = false, cut5 = false;
    double tl = 0.0, side = 1.0;
    if (T.mode != 2) {
        // normal / lognormal post-truncation length (src/truncate.cpp:335-345): Box-Muller on one Philox draw
        const Ph4m w = philox_mol(T.seed, g, ST_TRC_LEN, 0);
        const double u1 = ((double)w.x + 1.0) * (1.0 / 4294967296.0), u2 = u01(w.y);
        const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = T.mu + T.sigma * z;
        if (T.mode == 1) v = exp(v);
        tl = v;
        w1 = trc_keep(size, to_int(v), T.min_len, cut3);
    } else {
        // KDE model (src/truncate.cpp:322-351): truncation length from the row nearest to the molecule's size, averaged with the
        // next row's draw at the same quantile; share of the 3' end from the end-ratio histogram
        const Ph4m w = philox_mol(T.seed, g, ST_TRC_LEN, 0);
        int d = 0;
        {
            int lo = 0, hi = T.ny;                                     // lower_bound(y labels, size)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (T.ylab[mid] < (long long)size) lo = mid + 1; else hi = mid; }
            d = min(lo, T.ny - 1);
            if (lo < T.ny && d > 0 && llabs(T.ylab[d] - (long long)size) > llabs(T.ylab[d - 1] - (long long)size)) d--;
        }
        const double u = u01(w.x);
        double val = trc_row_draw(T, d, u, w.y);
        if (d + 1 < T.ny) val = (val + trc_row_draw(T, d + 1, u, w.z)) / 2.0;
        tl = T.models_length ? (double)size - val : val;
        if (T.always_end && !T.have_sider) side = 1.0;
        else {
            const Ph4m s = philox_mol(T.seed, g, ST_TRC_SIDE, 0);
            int bin = lower_bound_d(T.scdf, T.ns + 1, u01(s.x)) - 1;
            bin = max(0, min(bin, T.ns - 1));
            const double lo = bin == 0 ? 0.0 : T.slab[bin - 1], hi = T.slab[bin];
            side = lo + (hi - lo) * u01(s.y);
        }
        w1 = trc_keep(size, to_int((double)size - tl * side), T.min_len, cut3);
        const int s1 = w1;
        const int l2 = trc_keep(s1, to_int((double)s1 - tl * (1.0 - side)), T.min_len, cut5);
        w0 = s1 - l2;
    }
    keep_from[r] = (uint32_t)w0 | (cut5 ? 0x80000000u : 0u);
    keep_to[r] = (uint32_t)w1 | (cut3 ? 0x80000000u : 0u);
    tr_len[r] = tl; tr_side[r] = side;
    // sizes of the truncated molecule
    const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    uint64_t ni = 0, nm = 0;
    int c = 0;
    for (uint32_t i = 0; i < ic; i++) {
        const uint32_t* iv = B.intervals + 4ull * (ib + i);
        const int sz = (int)seg_size(iv);
        const int lo = max(w0, c), hi = min(w1, c + sz);
        const bool keep = sz > 0 ? hi > lo : ((!cut3 || c < w1) && (!cut5 || c > w0));
        if (keep) {
            ni++;
            const bool minus = iv[3] >> 31;
            const int fx = minus ? c + sz - hi : lo - c, fy = minus ? c + sz - lo : hi - c;     // forward offsets kept
            const uint32_t mb = iv[3] & 0x7fffffffu, me = iv[7] & 0x7fffffffu;
            for (uint32_t q = mb; q < me; q++) { const int p = (int)B.mods[2ull * q]; nm += (sz == 0 || (p >= fx && p < fy)) ? 1 : 0; }
        }
        c += sz;
    }
    n_ivls[r] = ni; n_mods[r] = nm;
    // the molecules come out of the MDF reader unrolled (stream_mdf(..., true), src/mdf.h:97-105): copies of a depth > 1
    // molecule are named id_0, id_1, ...
    n_idlen[r] = B.ids[2 * r + 1] + ((M.dup && (M.dup[r] >> 31)) ? 1u + (uint32_t)ndig(M.dup[r] & 0x7fffffffu) : 0u);
}

__global__ void k_trc_write(MolView M, const uint32_t* __restrict__ keep_from, const uint32_t* __restrict__ keep_to,
                            const uint64_t* __restrict__ ivl_off, const uint64_t* __restrict__ mod_off,
                            const uint64_t* __restrict__ id_off, MolOut O) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M.B.n_reads) return;
    const BatchView& B = M.B;
    const int w0 = (int)(keep_from[r] & 0x7fffffffu), w1 = (int)(keep_to[r] & 0x7fffffffu);
    const bool cut5 = keep_from[r] >> 31, cut3 = keep_to[r] >> 31;
    const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    uint64_t io = ivl_off[r], mo = mod_off[r];
    O.reads[2 * r] = (uint32_t)io; O.reads[2 * r + 1] = (uint32_t)(ivl_off[r + 1] - io);
    {
        uint8_t* d = O.idpool + id_off[r];
        const uint32_t so = B.ids[2 * r], sl = B.ids[2 * r + 1];
        uint32_t k = 0;
        for (; k < sl; k++) d[k] = B.idpool[so + k];
        if (M.dup && (M.dup[r] >> 31)) { d[k++] = '_'; k += (uint32_t)put_dec(d + k, M.dup[r] & 0x7fffffffu); }
        O.ids[2 * r] = (uint32_t)id_off[r]; O.ids[2 * r + 1] = k;
    }
    int c = 0;
    for (uint32_t i = 0; i < ic; i++) {
        const uint32_t* iv = B.intervals + 4ull * (ib + i);
        const int sz = (int)seg_size(iv);
        const int lo = max(w0, c), hi = min(w1, c + sz);
        const bool keep = sz > 0 ? hi > lo : ((!cut3 || c < w1) && (!cut5 || c > w0));
        if (keep) {
            const bool minus = iv[3] >> 31;
            const int fx = minus ? c + sz - hi : lo - c, fy = minus ? c + sz - lo : hi - c;
            uint32_t* ov = O.intervals + 4ull * io;
            ov[0] = iv[0];
            ov[1] = sz > 0 ? iv[1] + (uint32_t)fx : iv[1];
            ov[2] = sz > 0 ? iv[1] + (uint32_t)fy : iv[2];
            ov[3] = (uint32_t)mo | (iv[3] & 0x80000000u);
            const uint32_t mb = iv[3] & 0x7fffffffu, me = iv[7] & 0x7fffffffu;
            // truncate() sorts the substitutions of THE cut segment -- the first whose end reaches the kept length, even when
            // the cut falls on its boundary -- in the 3' pass and, on the flipped molecule, in the 5' pass
            const int szk = min(c + sz, w1) - c;                      // size after the 3' pass
            const bool was_cut = sz > 0 && ((cut3 && c < w1 && c + sz >= w1) || (cut5 && c <= w0 && w0 < c + szk));
            const uint64_t m_first = mo;
            for (uint32_t q = mb; q < me; q++) {
                const int p = (int)B.mods[2ull * q];
                if (sz == 0 || (p >= fx && p < fy)) { O.mods[2 * mo] = (uint32_t)(sz == 0 ? p : p - fx); O.mods[2 * mo + 1] = B.mods[2ull * q + 1]; mo++; }
            }
            if (was_cut) {
                // einterval::truncate sorts the substitutions of a cut segment by position (stable here)
                for (uint64_t a = m_first + 1; a < mo; a++) {
                    const uint32_t kp = O.mods[2 * a], kb = O.mods[2 * a + 1];
                    uint64_t b = a;
                    while (b > m_first && O.mods[2 * (b - 1)] > kp) { O.mods[2 * b] = O.mods[2 * (b - 1)]; O.mods[2 * b + 1] = O.mods[2 * (b - 1) + 1]; b--; }
                    O.mods[2 * b] = kp; O.mods[2 * b + 1] = kb;
                }
            }
            io++;
        }
        c += sz;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline unsigned nblk(uint64_t n) { return (unsigned)((n + 127) / 128); }

hipError_t launch_pcr_count(const MolView& m, const PcrParams& p, uint64_t* n_out, uint32_t* status, hipStream_t s) {
    if (!m.n_kept) return hipSuccess;
    hipLaunchKernelGGL(k_pcr_count, dim3(nblk(m.n_kept)), dim3(128), 0, s, m, p, n_out, status);
    return hipGetLastError();
}
hipError_t launch_pcr_list(const MolView& m, const PcrParams& p, const uint64_t* out_off, uint32_t* node_mol, uint64_t* node_mask,
                           uint64_t* node_ivls, uint64_t* node_mods, uint64_t* node_idlen, hipStream_t s) {
    if (!m.n_kept) return hipSuccess;
    hipLaunchKernelGGL(k_pcr_list, dim3(nblk(m.n_kept)), dim3(128), 0, s, m, p, out_off, node_mol, node_mask, node_ivls, node_mods, node_idlen);
    return hipGetLastError();
}
hipError_t launch_pcr_write(const MolView& m, const PcrParams& p, uint64_t n_nodes, const uint32_t* node_mol, const uint64_t* node_mask,
                            const uint64_t* ivl_off, const uint64_t* mod_off, const uint64_t* id_off, const MolOut& o, hipStream_t s) {
    if (!n_nodes) return hipSuccess;
    hipLaunchKernelGGL(k_pcr_write, dim3(nblk(n_nodes)), dim3(128), 0, s, m, p, n_nodes, node_mol, node_mask, ivl_off, mod_off, id_off, o);
    return hipGetLastError();
}
hipError_t launch_trc_plan(const MolView& m, const TrcParams& p, uint64_t first_index, uint32_t* keep_from, uint32_t* keep_to, double* tr_len,
                           double* tr_side, uint64_t* n_ivls, uint64_t* n_mods, uint64_t* n_idlen, hipStream_t s) {
    if (!m.B.n_reads) return hipSuccess;
    hipLaunchKernelGGL(k_trc_plan, dim3(nblk(m.B.n_reads)), dim3(128), 0, s, m, p, first_index, keep_from, keep_to, tr_len, tr_side, n_ivls, n_mods, n_idlen);
    return hipGetLastError();
}
hipError_t launch_trc_write(const MolView& m, const uint32_t* keep_from, const uint32_t* keep_to, const uint64_t* ivl_off,
                            const uint64_t* mod_off, const uint64_t* id_off, const MolOut& o, hipStream_t s) {
    if (!m.B.n_reads) return hipSuccess;
    hipLaunchKernelGGL(k_trc_write, dim3(nblk(m.B.n_reads)), dim3(128), 0, s, m, keep_from, keep_to, ivl_off, mod_off, id_off, o);
    return hipGetLastError();
}

}  // namespace tk
