// kernels.hip -- hand-written gfx950 kernels of the TKSM Seq hot path.
//
// Two implementations of the Badread path share the stage code below:
//   * the fast pipeline: k_init, then rounds of k_loop (one LANE per read: error loop up to the next identity re-estimation;
//     k_loopw, one wave per read, when few reads are left) and k_alnf (one lane per alignment job: the window decoded from the
//     read's slot codes and aligned, bit-parallel); the last few thousand reads of a batch -- and, from round 0 on a stream of
//     their own, the reads predicted to need several times the median read's visits -- finish in the straggler kernel
//     (k_loopw<true>: every remaining visit of a read on one wave, alignments by band_align); then the last visit k_err (one wave
//     per read) -- ACGT reads;
//   * k_simulate: one wavefront owns one read from splice to finished sequence/qualities, alignment done across the
//     wave -- byte-exact for any alphabet; the exact fallback and the --perfect path.
// Reference behaviour restated per stage (file:line into vpc-ccg/tksm):
//   S0 pack        py/sequence.py:168-194  (FASTA text -> contig strings; here 2 bit/base + byte blocks)
//   S1 splice      py/sequence.py:303-313, :224-239
//   S2 identity    py/tksm_badread.py:741-745
//   S3 errors      py/tksm_badread.py:333-432, :119-144, :199-213
//   S4 identity re-estimation   py/tksm_badread.py:405-432 (edlib -> banded NW on the wave)
//   S5 q-scores    py/tksm_badread.py:607-655, :584-598
//   S6 trim/format py/tksm_badread.py:434-451, py/sequence.py:252-288
// Integer/byte work throughout: no MFMA.  fp64 is used only for the scalar identity bookkeeping
// and is compiled with -ffp-contract=off so it matches the CPU oracle bit for bit.
#include "kernels.h"
#include <type_traits>

namespace tk {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
#define DEV __device__ __forceinline__

constexpr int BIG = 1 << 28;
DEV bool is_inf(int v) { return v >= (1 << 27); }

struct Ph4 { uint32_t x, y, z, w; };
enum { ST_ID = 0, ST_PAD = 1, ST_IDENT = 2, ST_DRAW = 3, ST_ALNPOS = 4, ST_QUAL = 5, ST_TAIL = 6 };

DEV Ph4 philox(uint64_t seed, uint64_t read, uint32_t stream, uint32_t n) {
    uint32_t c0 = (uint32_t)read, c1 = (uint32_t)(read >> 32), c2 = stream, c3 = n;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32 -> 64 product per word (v_mad_u64_u32) instead of a high and a low multiply
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Ph4{c0, c1, c2, c3};
}

DEV uint8_t base_char(int code) { return (uint8_t)((0x54474341u >> (8 * (code & 3))) & 0xff); }
DEV int code_of(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
DEV uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
DEV uint8_t comp(uint8_t c) {
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
}

// Wave-wide inclusive scans by data-parallel-primitive moves (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then the last lane
// of row 0 / 2 to row 1 / 3 and lane 31 to the upper half): 6 steps of ~2 instructions, no LDS crossbar (ds_bpermute: ~70 cycles
// of latency a step, 6 steps a scan -- the column step of band_align is a chain of two scans).  A lane without a source keeps
// `old` = the operation's identity.
template <int CTRL, int ROW_MASK> DEV int dpp_mov(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;    // lane l takes lane l + 1 / lane l - 1 (the first / last lane keeps `old`)
#define TKSM_DPP_SCAN(OP, IDENT, x)                         \
    x = OP(x, dpp_mov<0x111, 0xf>(IDENT, x));               \
    x = OP(x, dpp_mov<0x112, 0xf>(IDENT, x));               \
    x = OP(x, dpp_mov<0x114, 0xf>(IDENT, x));               \
    x = OP(x, dpp_mov<0x118, 0xf>(IDENT, x));               \
    x = OP(x, dpp_mov<0x142, 0xa>(IDENT, x));               \
    x = OP(x, dpp_mov<0x143, 0xc>(IDENT, x));
DEV int scan_min_incl(int x, int) { TKSM_DPP_SCAN(min, 0x7fffffff, x) return x; }
DEV int scan_max_incl(int x, int) { TKSM_DPP_SCAN(max, (int)0x80000000, x) return x; }
DEV int add_i32(int a, int b) { return a + b; }
DEV int scan_add_incl(int x, int) { TKSM_DPP_SCAN(add_i32, 0, x) return x; }
DEV uint32_t scan_umax_incl(uint32_t x) {
#define TKSM_UMAX_STEP(C, R) x = max(x, (uint32_t)dpp_mov<C, R>(0, (int)x));
    TKSM_UMAX_STEP(0x111, 0xf) TKSM_UMAX_STEP(0x112, 0xf) TKSM_UMAX_STEP(0x114, 0xf) TKSM_UMAX_STEP(0x118, 0xf) TKSM_UMAX_STEP(0x142, 0xa) TKSM_UMAX_STEP(0x143, 0xc)
#undef TKSM_UMAX_STEP
    return x;
}
// exclusive prefix sum over the wave of a small value (0..7) from three ballots: no cross-lane shuffles
DEV int prefix_small(int v, int& total) {
    const unsigned long long b0 = __ballot(v & 1), b1 = __ballot(v & 2), b2 = __ballot(v & 4);
    const int e0 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u));
    const int e1 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
    const int e2 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
    total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    return e0 + 2 * e1 + 4 * e2;
}

DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

DEV uint32_t lo32(unsigned long long v) { return (uint32_t)v; }
DEV uint32_t hi32(unsigned long long v) { return (uint32_t)(v >> 32); }
DEV unsigned long long mk64(uint32_t hi, uint32_t lo) { return ((unsigned long long)hi << 32) | lo; }
// ({hi, lo} >> s) & 0xffffffff for s in 0..31: one v_alignbit_b32
DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
// Any boolean function of three words in one instruction (V_BITOP3_B32, new in gfx950): the truth table is the function applied
// to the constants TA, TB, TC below, e.g. bool3<BOOL3(TA | ~(TB | TC))>(x, y, z) = x | ~(y | z).
constexpr uint32_t TA = 0xF0u, TB = 0xCCu, TC = 0xAAu;
#define BOOL3(expr) ((uint32_t)((expr) & 0xFFu))
template <uint32_t TT> DEV uint32_t bool3(uint32_t x, uint32_t y, uint32_t z) { return __builtin_amdgcn_bitop3_b32(x, y, z, TT); }
template <uint32_t TT> DEV unsigned long long bool3(unsigned long long x, unsigned long long y, unsigned long long z) {
    return mk64(bool3<TT>(hi32(x), hi32(y), hi32(z)), bool3<TT>(lo32(x), lo32(y), lo32(z)));
}
// bits [s, s + 64) of the 128-bit value {x1, x0}, s in 0..63
DEV unsigned long long funnel128(unsigned long long x0, unsigned long long x1, int s) {
    const bool a = (s & 32) != 0;
    const uint32_t b = (uint32_t)s & 31u;
    const uint32_t y0 = a ? hi32(x0) : lo32(x0), y1 = a ? lo32(x1) : hi32(x0), y2 = a ? hi32(x1) : lo32(x1);
    return mk64(alignbit(y2, y1, b), alignbit(y1, y0, b));
}

// fp64 helpers of the error loop's bookkeeping: the IEEE (correctly rounded) sequences the compiler emits for `sqrt(x)`
// and `a / b`, without the range scaling (operands here are between 1e-3 and 1e5) and with the part that depends only
// on the divisor computed once per read.  Bit-identical to the compiler's expansion, hence to the CPU oracle.
DEV double sqrt_inrange(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
DEV double rcp_refined(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double r = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, r, y);
    r = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, r, y);
}
DEV double div_inrange(double a, double b, double yb) {
    const double q = a * yb;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, yb, q);
}

DEV uint8_t ref_base(const RefView& R, uint64_t g) {
    uint32_t ex = R.blocktab[g >> BLOCK_SHIFT];
    if (ex != NO_BLOCK) return R.pool[((uint64_t)ex << BLOCK_SHIFT) + (g & ((1u << BLOCK_SHIFT) - 1))];
    uint32_t w = R.packed[g >> 4];
    return base_char((int)(w >> ((g & 15) * 2)));
}

// (k_simulate's own LDS slot codes; the fast pipeline's st_nb: 0x1000 | planar symbol while pristine, bit 15 set once changed)
// slot code (u16) of new_fragment_bases[p]: 0 = pristine (the original byte).  Otherwise
// bit15 = 1, bits 14..12 = length (0..5), bits 11..10 = 1 + index of the symbol that is the
// ORIGINAL byte (0 = none), bits 9..0 = 2-bit base codes, symbol x at bits 2x.
DEV int slot_len(uint32_t code) { return code ? (int)((code >> 12) & 7) : 1; }
DEV uint8_t slot_sym(uint32_t code, int x, uint8_t orig) {
    if (!code) return orig;
    if ((int)((code >> 10) & 3) == x + 1) return orig;
    return base_char((int)(code >> (2 * x)));
}

// ------------------------------------------------------------------------------------------------
// S0: ASCII -> 2 bit/base (+ per-4096-base-block exception flags), then byte copies of flagged blocks
// ------------------------------------------------------------------------------------------------
__global__ void k_pack(const uint8_t* __restrict__ ascii, uint64_t n, uint64_t gstart, uint32_t* __restrict__ packed,
                       uint32_t* __restrict__ blockflag) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // word index within the contig
    uint64_t nw = (n + 15) >> 4;
    for (; w < nw; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t b0 = w << 4;
        uint32_t word = 0; bool exc = false;
        if (b0 + 16 <= n && ((uintptr_t)(ascii + b0) & 15) == 0) {
            uint4 v = *reinterpret_cast<const uint4*>(ascii + b0);
            uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                int c = code_of(upper((uint8_t)(q[i >> 2] >> (8 * (i & 3)))));
                exc |= c < 0; word |= (uint32_t)(c & 3) << (2 * i);
            }
        } else {
            for (int i = 0; i < 16 && b0 + i < n; i++) {
                int c = code_of(upper(ascii[b0 + i]));
                exc |= c < 0; word |= (uint32_t)(c & 3) << (2 * i);
            }
        }
        packed[(gstart >> 4) + w] = word;
        if (exc) blockflag[(gstart + b0) >> BLOCK_SHIFT] = 1;
    }
}

__global__ void k_fill_pool(const uint8_t* __restrict__ ascii, uint64_t n, uint64_t gstart,
                            const uint32_t* __restrict__ blocktab, uint8_t* __restrict__ pool) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t g = gstart + i;
        uint32_t ex = blocktab[g >> BLOCK_SHIFT];
        if (ex != NO_BLOCK) pool[((uint64_t)ex << BLOCK_SHIFT) + (g & ((1u << BLOCK_SHIFT) - 1))] = upper(ascii[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// interval geometry shared by the length pre-pass and the splice
// ------------------------------------------------------------------------------------------------
struct Ivl { uint64_t gbase; uint32_t s, len; bool literal, minus; uint32_t mod_begin, mod_end; };

DEV Ivl load_interval(const BatchView& B, const RefView& R, uint32_t idx) {
    const uint4 v = *reinterpret_cast<const uint4*>(B.intervals + 4ull * idx);
    uint32_t next_mod = B.intervals[4ull * (idx + 1) + 3] & 0x7fffffffu;
    Ivl iv;
    iv.literal = v.x >> 31; iv.minus = v.w >> 31;
    iv.mod_begin = v.w & 0x7fffffffu; iv.mod_end = next_mod;
    uint64_t clen, base;
    if (iv.literal) { uint32_t li = v.x & 0x7fffffffu; base = B.literals[2ull * li]; clen = B.literals[2ull * li + 1]; }
    else { base = R.contigs[2ull * v.x]; clen = R.contigs[2ull * v.x + 1]; }
    uint64_t s = min((uint64_t)v.y, clen), e = min((uint64_t)v.z, clen);   // python slice clamp
    iv.s = (uint32_t)s; iv.len = e > s ? (uint32_t)(e - s) : 0; iv.gbase = base;
    return iv;
}

// four 2-bit base codes (bits 0..7 of v) -> four ASCII bytes "ACGT"[code], first code in the low byte
DEV uint32_t ascii4(uint32_t v) {
    v &= 0xffu;
    uint32_t t = (v | (v << 12)) & 0x000f000fu;
    t = (t | (t << 6)) & 0x03030303u;
    const uint32_t b0 = t & 0x01010101u, b1 = (t >> 1) & 0x01010101u;
    return 0x41414141u + 2u * b0 + 6u * b1 + 11u * (b0 & b1);     // A 0x41, C +2, G +6, T +2+6+11
}

// 16 bases (fewer at the end) of one interval, in output order: piece t0 of the slice that starts at forward position g.
// From the packed reference: two words, a funnel shift, for the minus strand a bit reversal + complement (= both bits
// inverted), four codes -> four ASCII bytes at a time.  Literal segments and reference blocks that hold other symbols
// than ACGT: bytewise.  Writes exactly the piece's bytes (a neighbouring interval may be written in the same pass).
DEV void splice_piece(const BatchView& B, const RefView& R, uint64_t g, uint32_t len, bool literal, bool minus, uint32_t t0, uint8_t* dstp) {
    const uint32_t n = min(16u, len - t0);
    const uint64_t g0 = g + (minus ? len - t0 - n : t0);              // forward positions [g0, g0 + n)
    uint32_t d[4];
    bool fast = false;
    if (!literal) {
        const uint32_t bt0 = R.blocktab[g0 >> BLOCK_SHIFT], bt1 = R.blocktab[(g0 + n - 1) >> BLOCK_SHIFT];
        const uint32_t w0 = R.packed[g0 >> 4], w1 = R.packed[(g0 >> 4) + 1];      // (the buffer has a spare line at its end)
        fast = bt0 == NO_BLOCK && bt1 == NO_BLOCK;
        uint32_t x = (uint32_t)((((unsigned long long)w1 << 32) | w0) >> (2 * (g0 & 15)));          // n codes, first at bit 0
        if (minus) {
            // last base first; after the reversal the n codes sit at the top
            x = __builtin_bitreverse32(~x);
            x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
            x >>= 2 * (16 - n);
        }
        d[0] = ascii4(x); d[1] = ascii4(x >> 8); d[2] = ascii4(x >> 16); d[3] = ascii4(x >> 24);
    }
    if (!fast) {
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t src = minus ? len - 1 - (t0 + j) : t0 + j;
            const uint8_t b = literal ? upper(B.litpool[g + src]) : ref_base(R, g + src);
            dstp[j] = minus ? comp(b) : b;
        }
    } else if (n == 16) __builtin_memcpy(dstp, d, 16);
    else {
        unsigned long long lo8 = ((unsigned long long)d[1] << 32) | d[0], hi8 = ((unsigned long long)d[3] << 32) | d[2];
        for (uint32_t j = 0; j < n; j++) {
            dstp[j] = (uint8_t)lo8;
            lo8 = (lo8 >> 8) | (hi8 << 56); hi8 >>= 8;
        }
    }
}

// tail noise length of one read (KDE_noise_generator.noise_seq / Custom2Dist.__call__ / CustomDist.__call__,
// py/tksm_badread.py:919-926, :1023-1033, :988-991): nothing with probability 1 - ratio; else the row of the first label
// >= the fragment length (past the last label: the last row and the factor len(ly) / ly[-1], as the reference has it),
// inverse-CDF pick of a length, truncated product.  The host reads the lengths back to size the batch.
__global__ void k_tail_lengths(BatchView B, RefView R, TailView T, uint64_t seed, uint64_t first_read, uint64_t stride,
                               uint32_t* __restrict__ tail_len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n_reads) return;
    const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    uint64_t total = 0;
    for (uint32_t i = 0; i < ic; i++) total += load_interval(B, R, ib + i).len;
    const Ph4 w = philox(seed, first_read + r * stride, ST_TAIL, 0);
    const double two32 = 1.0 / 4294967296.0;
    uint32_t x = 0;
    if (!((double)w.x * two32 > T.ratio)) {
        const double y = (double)total;
        int lo = 0, hi = T.n_ly;                        // np.searchsorted(ly, y), side left
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (T.ly[mid] < y) lo = mid + 1; else hi = mid; }
        int pos = lo;
        if (pos < T.n_ly - 1 && fabs(T.ly[pos] - y) > fabs(T.ly[pos + 1] - y)) pos++;
        double mult = 1.0;
        if (pos >= T.n_ly) { mult = (double)pos / T.ly[T.n_ly - 1]; pos = T.n_ly - 1; }
        const double* cdf = T.cdf + (size_t)pos * T.n_lx;
        const double val = (double)w.y * two32;
        lo = 0; hi = T.n_lx - 1;                        // first entry >= val (the last one if rounding left none)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] >= val) hi = mid; else lo = mid + 1; }
        const double v = T.lx[lo] * mult;
        if (v >= 1.0) x = v > 1e9 ? 1000000000u : (uint32_t)v;
    }
    tail_len[r] = x;
}

// the tail's bases (noise_seq, py/tksm_badread.py:927-933): a 4-state chain, first state uniform, one weighted step per
// base.  Each lane turns its step's uniform into the map state -> next state; a wave scan composes the maps.
DEV void tail_fill(const TailChain* TC, uint64_t seed, uint64_t g, uint8_t* dst, int x, int lane) {
    if (x <= 0) return;
    const double two32 = 1.0 / 4294967296.0;
    int state = (int)(philox(seed, g, ST_TAIL, 0).z >> 30);
    const uint32_t bases = TC->bases;
    for (int t0 = 0; t0 < x; t0 += 64) {
        const int t = t0 + lane;
        uint32_t map = 0xE4u;                           // identity for the lanes past the end
        if (t < x) {
            const Ph4 w = philox(seed, g, ST_TAIL, 1u + (uint32_t)(t >> 2));
            const uint32_t u = (t & 3) == 0 ? w.x : (t & 3) == 1 ? w.y : (t & 3) == 2 ? w.z : w.w;
            map = 0u;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double v = (double)u * two32 * TC->cum[4 * s + 3];
                const uint32_t nx = (uint32_t)(TC->cum[4 * s] <= v) + (uint32_t)(TC->cum[4 * s + 1] <= v) + (uint32_t)(TC->cum[4 * s + 2] <= v);
                map |= nx << (2 * s);
            }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)map, o, 64);      // the earlier steps
            if (lane >= o) {
                uint32_t c = 0u;
#pragma unroll
                for (int s = 0; s < 4; s++) c |= ((map >> (2 * ((y >> (2 * s)) & 3u))) & 3u) << (2 * s);
                map = c;
            }
        }
        const int mine = (int)((map >> (2 * state)) & 3u);
        if (t < x) dst[t] = (uint8_t)(bases >> (8 * mine));
        state = __shfl(mine, 63, 64);
    }
}

__global__ void k_read_lengths(BatchView B, RefView R, int k, int cap_num, int cap_den, int cap_add,
                               const uint32_t* __restrict__ tail_len,
                               uint32_t* __restrict__ raw_len, uint64_t* __restrict__ slot_cap,
                               uint32_t* __restrict__ status) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n_reads) return;
    uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    uint64_t total = tail_len ? tail_len[r] : 0;
    for (uint32_t i = 0; i < ic; i++) total += load_interval(B, R, ib + i).len;
    raw_len[r] = (uint32_t)total;
    uint64_t cap = (total + 2 * (uint64_t)k) * cap_num / cap_den + cap_add;
    cap = (cap + 15) & ~15ull;
    slot_cap[r] = 2 * cap;       // seq | qual
    status[r] = 0;
}

// ------------------------------------------------------------------------------------------------
// S4/S5: guided banded global alignment on one wave (the byte-exact path; also the slow path for
// reads with non-ACGT bytes).  Column-major: column j (1-based, joined new base N[j-1]) belongs to
// slot owner[j-1] and owns the 64 fragment rows t_j .. t_j+63 (1-based), t_j = max(1, owner+1-31);
// lane b = row t_j + b.  Above the window: unreachable.  Below: value of the bottom cell + distance,
// predecessor up.  Row 0 (H[0][j] = j) is the real boundary while t_j == 1.
// MODE 0: predecessor preference up, left, diagonal (query = fragment).
// MODE 1: left, up, diagonal (query = new sequence; q-score cigar).
// stat = matches << 16 | columns of the preferred optimal path.
// TRACE: per column j the masks {up ok, left ok} are stored at trace[2*j], trace[2*j+1] (u64).
// ------------------------------------------------------------------------------------------------
struct AlnOut { int dist; uint32_t stat; };

template <int MODE, bool TRACE, class OT>
DEV AlnOut band_align(const uint8_t* F, int n, const uint8_t* N, const OT* owner, int m, int lane,
                      unsigned long long* trace) {
    // One column per step; the step is a chain (previous column -> vertical propagation = a min-scan -> predecessor -> the
    // statistics of the preferred path = a segmented copy), so its length is its latency: the column's symbol, window and fragment
    // bytes are loaded one / two columns ahead, the previous column's cells come by a one-lane shift of the wave when the window
    // moved by 0 or 1 rows (all but the columns behind a deleted slot), the scans are data-parallel-primitive moves, and the
    // segmented copy is ONE unsigned max-scan of {source lane, value} packed into a word.  stat (matches << 16 | columns) is only
    // produced without TRACE (the callers with a trace count along their walk), for windows of <= 1023 rows (the re-estimation
    // windows have <= 1000): the value then fits the 26 bits the packing leaves it.
    // Unreachable cells carry BIG (is_inf: >= 2^27); BIG + a few thousand is still unreachable and never equals a finite value, so
    // the sums below need no guards, and every selection is a bitwise one (no branches inside the chain).
    int tp = 1;
    int H = 1 + lane;                         // column 0: H[i][0] = i
    uint32_t st = (uint32_t)(1 + lane);
    auto top_at = [&](int j) { return max(1, (int)owner[j - 1] + 1 - 31); };
    auto frag_at = [&](int t) { const int b = (int)F[max(min(t + lane, n) - 1, 0)]; return t + lane <= n ? b : 257; };
    int t1 = m >= 1 ? top_at(1) : 1, nc1 = m >= 1 ? (int)N[0] : 0;
    int fc1 = frag_at(t1);
    int t2 = m >= 2 ? top_at(2) : 1, nc2 = m >= 2 ? (int)N[1] : 0;
    for (int j = 1; j <= m; j++) {
        const int t = __builtin_amdgcn_readfirstlane(t1);
        const int nc = nc1, fc = fc1;
        t1 = t2; nc1 = nc2;
        fc1 = frag_at(t1);
        { const int j2 = min(j + 2, m); t2 = top_at(j2); nc2 = (int)N[j2 - 1]; }
        const int sh = t - tp;                                  // (wave-uniform)
        const bool valid = t + lane <= n;
        const int match = fc == nc ? 1 : 0;
        // previous column at rows i-1 (diag) and i (left)
        int hd, hl; uint32_t sd, sl;
        if (sh == 1) {
            // diag = the same lane; left = the lane above, for lane 63 the virtual cell below the previous window
            hd = H; sd = st;
            hl = dpp_mov<DPP_WAVE_SHL1, 0xf>(H + 1, H);
            sl = (uint32_t)dpp_mov<DPP_WAVE_SHL1, 0xf>((int)(st + 1u), (int)st);
        } else if (sh == 0) {
            // left = the same lane; diag = the lane below, for lane 0 the boundary row 0 (H[0][j-1] = j-1) while the window is at row 1
            hl = H; sl = st;
            hd = dpp_mov<DPP_WAVE_SHR1, 0xf>(tp == 1 ? j - 1 : BIG, H);
            sd = (uint32_t)dpp_mov<DPP_WAVE_SHR1, 0xf>(j - 1, (int)st);
        } else {
            const int ld = lane + sh - 1, ll = lane + sh;
            const int hbot = __builtin_amdgcn_readlane(H, 63); const uint32_t sbot = (uint32_t)__builtin_amdgcn_readlane((int)st, 63);
            hd = __shfl(H, ld & 63, 64); sd = __shfl(st, ld & 63, 64);
            hl = __shfl(H, ll & 63, 64); sl = __shfl(st, ll & 63, 64);
            if (ld > 63) { hd = hbot + (ld - 63); sd = sbot + (uint32_t)(ld - 63); }
            if (ll > 63) { hl = hbot + (ll - 63); sl = sbot + (uint32_t)(ll - 63); }
        }
        const int vd = hd + 1 - match, vl = hl + 1;
        const int vu0 = (lane == 0 && t == 1) ? j + 1 : BIG;      // boundary row 0 above lane 0
        const int tmin = valid ? min(min(vd, vl), vu0) : BIG;
        const int x = scan_min_incl(tmin - lane, lane) + lane;
        const bool ok = valid & !is_inf(x);
        const int h = ok ? x : BIG;
        // the cell above: lane 0 has the boundary row 0 (value j) while the window is at row 1
        const int hup = dpp_mov<DPP_WAVE_SHR1, 0xf>(t == 1 ? j : BIG, h);
        const bool upok = ok & (hup + 1 == h);
        const bool leftok = ok & (vl == h);
        if (!TRACE) {
            const bool up_taken = MODE == 0 ? upok : (upok & !leftok), left_taken = MODE == 0 ? (leftok & !upok) : leftok;
            const bool base = !up_taken | (lane == 0);
            const uint32_t sb = up_taken ? (uint32_t)(j + 1) : (left_taken ? sl + 1u : sd + ((uint32_t)match << 16) + 1u);
            // a run of up moves copies the statistics of the cell below the run, + 1 column per row: the nearest base lane at or
            // below each lane and its value, by one max-scan of lane << 26 | value
            const uint32_t km = scan_umax_incl(base ? ((uint32_t)lane << 26) | (sb & 0x3ffffffu) : 0u);
            st = base ? sb : (km & 0x3ffffffu) + (uint32_t)(lane - (int)(km >> 26));
        }
        H = h; tp = t;
        if (TRACE) {
            const unsigned long long um = __ballot(upok), lm = __ballot(leftok);
            if (lane == 0) { trace[2 * j] = um; trace[2 * j + 1] = lm; }
        }
    }
    AlnOut o;
    const int bf = n - tp;
    if (bf < 0 || bf > 63) { o.dist = BIG; o.stat = 0; return o; }
    o.dist = __builtin_amdgcn_readlane(H, bf); o.stat = TRACE ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)st, bf);
    return o;
}

// ------------------------------------------------------------------------------------------------
// Unbanded alignment, for the rare window whose optimal path the guided band cannot hold (the end cell outside the last
// window: a homopolymer that lost more than 32 bases at the end of the read, ...).  Same recurrence and predecessor
// preference as band_align, every row: 64 rows per pass over the columns, lane = row, the last row of a pass is the
// boundary of the next; per (pass, column) the masks {up ok, left ok}; then the walk from (n, m).  Restates full_align of
// the oracle (oracle/tksm_oracle.c), which the specification prescribes for exactly this case.
// Returns false when the pool cannot hold the masks (the caller reports status bit 2).
// ------------------------------------------------------------------------------------------------
template <int MODE>
DEV bool full_align_wave(const uint8_t* F, int n, const uint8_t* N, int m, int lane, const SimBuffers& O, int& mt_out, int& cols_out,
                         uint8_t* popd) {
    const int nblk = (n + 63) / 64;
    const size_t stride = (size_t)(m + 1) * 2;                          // u64 words per pass
    const unsigned long long bytes = (((unsigned long long)nblk * stride * 8ull + 2ull * (unsigned long long)(m + 64) * 4ull) + 255ull) & ~255ull;
    unsigned long long off = 0;
    if (lane == 0) off = atomicAdd(O.full_pool_used, bytes);
    off = (unsigned long long)__shfl((long long)off, 0, 64);
    if (!O.full_pool || off + bytes > O.full_pool_bytes) return false;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(O.full_pool + off);
    int* hprev = reinterpret_cast<int*>(masks + (size_t)nblk * stride);
    int* hcur = hprev + (m + 64);
    for (int j = lane; j <= m; j += 64) hprev[j] = j;                    // row 0: H[0][j] = j
    wave_sync();
    for (int b = 0; b < nblk; b++) {
        const int i = 64 * b + lane + 1;
        const bool valid = i <= n;
        const int fc = valid ? (int)F[i - 1] : 257;
        int H = valid ? i : BIG;                                         // column 0: H[i][0] = i
        if (lane == 63) hcur[0] = H;
        int chunk = hprev[min(lane, m)];                                 // boundary values of columns 0..63
        int outv = 0;
        for (int j = 1; j <= m; j++) {
            const int top_d = __shfl(chunk, (j - 1) & 63, 64);           // H[64b][j-1]
            if ((j & 63) == 0) chunk = hprev[min(j + lane, m)];
            const int top_u = __shfl(chunk, j & 63, 64);                 // H[64b][j]
            const int match = fc == (int)N[j - 1];
            int hd = __shfl_up(H, 1, 64);
            if (lane == 0) hd = top_d;
            const int vd = !is_inf(hd) ? hd + 1 - match : BIG;
            const int vl = !is_inf(H) ? H + 1 : BIG;
            const int vu0 = lane == 0 ? top_u + 1 : BIG;
            const int tmin = valid ? min(min(vd, vl), vu0) : BIG;
            const int x = scan_min_incl(tmin - lane, lane) + lane;       // vertical moves inside the pass
            const int h = (valid && !is_inf(x)) ? x : BIG;
            int hup = __shfl_up(h, 1, 64);
            if (lane == 0) hup = top_u;
            const bool upok = valid && hup + 1 == h;
            const bool leftok = valid && vl == h;
            const unsigned long long um = __ballot(upok), lm = __ballot(leftok);
            if (lane == 0) { masks[(size_t)b * stride + 2 * (size_t)j] = um; masks[(size_t)b * stride + 2 * (size_t)j + 1] = lm; }
            const int last = __shfl(h, 63, 64);                          // row 64(b+1): boundary of the next pass
            if (lane == (j & 63)) outv = last;
            if ((j & 63) == 63 || j == m) { const int j0 = j & ~63; if (j0 + lane <= j && j0 + lane >= 1) hcur[j0 + lane] = outv; }
            H = h;
        }
        wave_sync();
        int* tswap = hprev; hprev = hcur; hcur = tswap;
    }
    // walk from (n, m)
    int i = n, j = m, mt = 0, cols = 0, dpend = 0;
    while (i > 0 || j > 0) {
        int mv;                                                          // 0 up, 1 left, 2 diagonal
        if (j == 0) mv = 0;
        else if (i == 0) mv = 1;
        else {
            const size_t at = (size_t)((i - 1) >> 6) * stride + 2 * (size_t)j;
            const unsigned long long um = masks[at], lm = masks[at + 1];
            const bool up = (um >> ((i - 1) & 63)) & 1ull, left = (lm >> ((i - 1) & 63)) & 1ull;
            mv = MODE == 0 ? (up ? 0 : (left ? 1 : 2)) : (left ? 1 : (up ? 0 : 2));
        }
        cols++;
        if (mv == 1) {
            if (popd && lane == 0) popd[j - 1] = (uint8_t)(2 | (min(dpend, 63) << 2));
            dpend = 0; j--;
        } else if (mv == 0) {
            dpend++; i--;
        } else {
            const bool eq = F[i - 1] == N[j - 1];
            mt += eq;
            if (popd && lane == 0) popd[j - 1] = (uint8_t)((eq ? 0 : 1) | (min(dpend, 63) << 2));
            dpend = 0; i--; j--;
        }
    }
    mt_out = mt; cols_out = cols;
    // give the memory back if nobody allocated after this wave (the usual case: these alignments are rare)
    if (lane == 0) atomicCAS(O.full_pool_used, off + bytes, off);
    return true;
}

// joins slots [p0, p0+n) into N, owner[j] = window-relative slot of joined base j; returns the joined length
// (may exceed ncap: nothing is written past ncap and the caller flags the overflow).
template <class OT>
DEV int join_window(const uint8_t* frag, const uint16_t* nb, int p0, int n, uint8_t* N, OT* owner, int ncap, int lane) {
    int base = 0;
    for (int q = 0; q < n; q += 64) {
        const int p = q + lane;
        uint32_t code = 0; int len = 0; uint8_t orig = 0;
        if (p < n) { code = nb[p0 + p]; len = slot_len(code); orig = frag[p0 + p]; }
        int total;
        const int off = base + prefix_small(len, total);
        if (off + len <= ncap)
            for (int x2 = 0; x2 < len; x2++) { N[off + x2] = slot_sym(code, x2, orig); owner[off + x2] = (OT)p; }
        base += total;
    }
    return base;
}

DEV uint64_t qs_hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

// '{:.2f}'.format(identity * 100) as an integer number of hundredths, correctly rounded
DEV long long pct_hundredths(double identity) {
    double e = identity * 100.0;
    double r = rint(e * 100.0);
    double err = fma(e, 100.0, -r);
    if (err > 0.5) r += 1.0; else if (err < -0.5) r -= 1.0;
    else if (err == 0.5) { if (fmod(r, 2.0) != 0.0) r += 1.0; }
    else if (err == -0.5) { if (fmod(r, 2.0) != 0.0) r -= 1.0; }
    return (long long)r;
}
DEV int ndigits(unsigned long long v) { int d = 1; while (v >= 10) { v /= 10; d++; } return d; }

// ------------------------------------------------------------------------------------------------
// S1..S6 main kernel: persistent waves pull reads from a global counter.
// LDS per wave: frag[lcap] | nb[lcap] (u16) | N[ncap] | popd[ncap] | owner[ncap] (u16)
// ------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];

// BIG: the working set (and the final alignment's masks) of a wave live in HBM instead of LDS -- for the rare molecule that
// needs this exact kernel (a non-ACGT byte, an alignment outside the band representation) and is longer than LDS holds
// (~18 kb); slot indices are 32-bit there.
template <bool BIG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_simulate(BatchView B, RefView R, ErrModelView EM,
                                                                 QsModelView QM, IdentView IM, SimParams P,
                                                                 SimBuffers O) {
    using OT = typename std::conditional<BIG, uint32_t, uint16_t>::type;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int glc = BIG ? P.lcap : P.s_lcap, gnc = BIG ? P.ncap : P.s_ncap;      // capacity for the padded fragment / the joined sequence
    const int wpw = blockDim.x >> 6;
    const size_t wave_id = (size_t)blockIdx.x * wpw + wave;
    uint8_t* frag = BIG ? O.big_scratch + wave_id * O.big_per_wave : lds_raw + (size_t)wave * (glc * 3 + gnc * 4);
    uint16_t* nb = reinterpret_cast<uint16_t*>(frag + glc);
    uint8_t* N = frag + 3 * (size_t)glc;
    uint8_t* popd = N + gnc;
    OT* owner = reinterpret_cast<OT*>(popd + gnc);
    unsigned long long* trace = BIG ? O.big_trace + wave_id * (size_t)(2 * (gnc + 2))
                                    : reinterpret_cast<unsigned long long*>(O.trace) + wave_id * (size_t)(P.trace_words / 2);
    const int k = EM.k;

    for (;;) {
        unsigned long long rr = 0;
        if (lane == 0) rr = atomicAdd(O.work_counter, 1ull);
        rr = __shfl((long long)rr, 0, 64);
        if (rr >= O.n_work) break;                // every wave reaches this exit
        const uint64_t r = O.read_list ? (uint64_t)O.read_list[rr] : rr;
        const uint64_t g = P.first_read + r * P.stride;
        const int raw_len = (int)O.raw_len[r];             // spliced bases + tail noise
        const int tail = O.tail_len ? (int)O.tail_len[r] : 0;
        const int L = raw_len + 2 * k;
        const uint64_t slot = O.slot_off[r];
        const int cap = (int)((O.slot_off[r + 1] - slot) >> 1);
        uint8_t* out_seq = O.scratch + slot;
        uint8_t* out_qual = out_seq + cap;
        uint32_t status = 0;
        if (BIG && lane == 0) O.status[r] &= ~8u;          // (this launch takes the reads the LDS-resident one flagged)
        if (L > glc) {
            // longer than this kernel's LDS-resident working set allows (the fast pipeline takes such reads as long as
            // they are plain ACGT): reported to the host
            if (lane == 0) { O.status[r] |= 8u; O.out_len[r] = 0; O.rec_len[r] = 0; O.identity[r] = 0.0; }
            continue;
        }

        // ---- S1 splice (py/sequence.py:303-313) into frag[k .. k+raw_len)
        {
            const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
            int o = k;
            for (uint32_t ii = 0; ii < ic; ii++) {
                const Ivl iv = load_interval(B, R, ib + ii);
                const int len = (int)iv.len;
                for (int t = lane; t < len; t += 64) {
                    const uint32_t src = iv.minus ? iv.s + (uint32_t)(len - 1 - t) : iv.s + (uint32_t)t;
                    uint8_t b = iv.literal ? upper(B.litpool[iv.gbase + src]) : ref_base(R, iv.gbase + src);
                    frag[o + t] = iv.minus ? comp(b) : b;
                }
                wave_sync();
                // modifications: position relative to the slice, applied before the strand flip,
                // later entries overwrite earlier ones (py/sequence.py:229-239) -> serial order
                for (uint32_t mi = iv.mod_begin; mi < iv.mod_end; mi++) {
                    const uint32_t mp = B.mods[2ull * mi], mc = B.mods[2ull * mi + 1];
                    if (mp >= (uint32_t)len) { status |= 2; continue; }
                    if (lane == 0) frag[o + (iv.minus ? len - 1 - (int)mp : (int)mp)] = iv.minus ? comp((uint8_t)mc) : (uint8_t)mc;
                }
                o += len;
            }
            if (tail) tail_fill(O.tail_chain, P.seed, g, frag + o, tail, lane);     // py/tksm_badread.py:335-339
        }
        double identity = 1.0;
        int out_len = 0;
        int st_draws = 0, st_changes = 0, st_aligns = 0, st_newlen = 0, st_strim = 0, st_etrim = 0;
        double errors = 0.0, target = 1.0;

        if (P.mode == 0) {
            // perfect (py/sequence.py:261-270): the error-free sequence itself
            wave_sync();
            for (int t = lane; t < raw_len; t += 64) out_seq[t] = frag[k + t];
            out_len = raw_len;
        } else {
            // ---- :334-341 pad with k random bases each side
            {
                const Ph4 pad = philox(P.seed, g, ST_PAD, 0);
                if (lane < k) {
                    frag[lane] = base_char((int)(pad.x >> (2 * lane)));
                    frag[k + raw_len + lane] = base_char((int)(pad.y >> (2 * lane)));
                }
            }
            for (int p = lane; p < L; p += 64) nb[p] = 0;
            // ---- S2 target identity (py/tksm_badread.py:741-745)
            if (IM.constant) target = IM.value;
            else {
                const uint32_t u = philox(P.seed, g, ST_IDENT, 0).x;
                const uint32_t idx = u >> 16;
                const double fr = (double)(u & 0xffffu) * (1.0 / 65536.0);
                const double qa = IM.qtab[idx], qb = IM.qtab[idx + 1];
                target = IM.value * (qa + (qb - qa) * fr);
            }
            wave_sync();

            // ---- S3 error insertion loop (py/tksm_badread.py:348-432)
            const double frag_len = (double)L;
            const int max_kmer_index = L - 1 - k;
            int change_count = 0;
            uint32_t n_base = 0, aln_no = 0;
            const long long loop_limit = 100ll * L;   // iterations with loop_count <= loop_limit run
            bool done = false;
            // the stop rules are evaluated at the top of every iteration; state only changes when a
            // draw is applied, so they are re-evaluated after each applied draw.
            if ((double)change_count > 0.9 * frag_len || 1.0 - errors / frag_len <= target) done = true;
            while (!done) {
                // 64 candidate draws, lane l = draw n_base + l  (loop_count = n + 1)
                const uint32_t n = n_base + (uint32_t)lane;
                const bool live = (long long)n + 1 <= loop_limit;
                const Ph4 d = philox(P.seed, g, ST_DRAW, n);
                const int i = (int)__umulhi(d.x, (uint32_t)(max_kmer_index + 1));
                int kind = 0;            // 0 no-op, 1 model alternative, 2 random change
                uint64_t alt = 0;
                if (live) {
                    int kidx = 0; bool valid = true;
                    for (int jj = 0; jj < k; jj++) { const int cc = code_of(frag[i + jj]); valid &= cc >= 0; kidx = (kidx << 2) | (cc & 3); }
                    if (EM.type == 0 || !valid) kind = 2;
                    else {
                        const uint32_t* cdf = EM.cdf + (size_t)kidx * EM.max_alts;
                        const int na = EM.nalts[kidx];
                        int a = 0;
                        while (a < na && !(d.y < cdf[a])) a++;
                        if (a == na) kind = 2;
                        else { alt = EM.alts[(size_t)kidx * EM.max_alts + a]; kind = (alt >> 63) ? 0 : 1; }
                    }
                    if (kind == 2) {
                        const uint32_t type = __umulhi(d.z, 3u), pos = __umulhi(d.w, (uint32_t)k);
                        const uint32_t base4 = d.w & 3u, side = (d.w >> 2) & 1u;
                        const uint32_t r3 = (((d.z & 0xffffu) * 3u) >> 16) + 1u;
                        alt = type | (pos << 2) | (base4 << 8) | (side << 10) | (r3 << 12);
                    }
                }
                unsigned long long mask = __ballot(live && kind != 0);
                const unsigned long long dead = __ballot(!live);
                while (mask) {
                    const int src = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int ai = __shfl(i, src, 64);
                    const int akind = __shfl(kind, src, 64);
                    const uint64_t aalt = ((uint64_t)(uint32_t)__shfl((int)(alt >> 32), src, 64) << 32) |
                                          (uint32_t)__shfl((int)(uint32_t)alt, src, 64);
                    const double est = 1.0 - errors / frag_len;
                    int boff = 0;
                    for (int jj = 0; jj < k; jj++) {
                        const int p = ai + jj;
                        const uint8_t orig = frag[p];
                        uint32_t enc; int len; bool differs;
                        if (akind == 1) {
                            len = (int)((aalt >> (3 * jj)) & 7);
                            const uint32_t codes = (uint32_t)((aalt >> (24 + 2 * boff)) & ((1u << (2 * len)) - 1u));
                            boff += len;
                            differs = !(len == 1 && base_char((int)codes) == orig);
                            enc = 0x8000u | ((uint32_t)len << 12) | codes;
                        } else {
                            const int type = (int)(aalt & 3), pos = (int)((aalt >> 2) & 15);
                            const uint32_t base4 = (uint32_t)((aalt >> 8) & 3), side = (uint32_t)((aalt >> 10) & 1);
                            const int r3 = (int)((aalt >> 12) & 3);
                            if (jj != pos) continue;
                            differs = true;
                            if (type == 0) {
                                const int cc = code_of(orig);
                                len = 1; enc = 0x8000u | (1u << 12) | (uint32_t)(cc < 0 ? (int)base4 : ((cc + r3) & 3));
                            } else if (type == 1) {
                                len = 2;
                                enc = side ? (0x8000u | (2u << 12) | (1u << 10) | (base4 << 2))    // orig + random
                                           : (0x8000u | (2u << 12) | (2u << 10) | base4);          // random + orig
                            } else { len = 0; enc = 0x8000u; }
                        }
                        if (!differs || nb[p] != 0) continue;
                        if (lane == 0) nb[p] = (uint16_t)enc;
                        change_count++;
                        const int new_errors = len < 2 ? 1 : len - 1;
                        errors += (double)new_errors * (est * sqrt(est));
                        if (change_count % 25 == 0) {           // ALIGNMENT_INTERVAL
                            wave_sync();
                            st_aligns++;
                            int p0 = 0, nrows = L;
                            if (L > 1000) {                     // ALIGNMENT_SIZE: random 1000-base window
                                const uint32_t w = philox(P.seed, g, ST_ALNPOS, aln_no).x;
                                p0 = (int)__umulhi(w, (uint32_t)(L - 1000 + 1));
                                nrows = 1000;
                            }
                            const int m = join_window(frag, nb, p0, nrows, N, owner, gnc, lane);
                            wave_sync();
                            if (m > gnc) status |= 1;
                            else {
                                const AlnOut a = band_align<0, false>(frag + p0, nrows, N, owner, m, lane, nullptr);
                                int cols = (int)(a.stat & 0xffffu), mt = (int)(a.stat >> 16);
                                if (is_inf(a.dist)) status |= full_align_wave<0>(frag + p0, nrows, N, m, lane, O, mt, cols, nullptr) ? 16u : 4u;
                                const double ident = cols ? (double)mt / (double)cols : 0.0;
                                if (L <= 1000) errors = (1.0 - ident) * frag_len;
                                else {
                                    const double estimated = (1.0 - ident) * frag_len;
                                    const double weight = 1000.0 / frag_len;
                                    errors = estimated * weight + errors * (1.0 - weight);
                                }
                            }
                            aln_no++;
                        }
                    }
                    wave_sync();
                    if ((double)change_count > 0.9 * frag_len || 1.0 - errors / frag_len <= target) {
                        done = true; st_draws = (int)n_base + src + 1; break;
                    }
                }
                if (!done) {
                    if (dead) { done = true; st_draws = (int)loop_limit; }
                    else n_base += 64;
                }
            }
            st_changes = change_count;

            // ---- :434-437 trims and the joined sequence
            wave_sync();
            int start_trim = 0, end_trim = 0;
            {
                int v1 = lane < k ? slot_len(nb[lane]) : 0, v2 = lane < k ? slot_len(nb[L - k + lane]) : 0;
                start_trim = __shfl(scan_add_incl(v1, lane), 63, 64);
                end_trim = __shfl(scan_add_incl(v2, lane), 63, 64);
            }
            const int m = join_window(frag, nb, 0, L, N, owner, min(gnc, cap), lane);
            wave_sync();
            st_newlen = m; st_strim = start_trim; st_etrim = end_trim;
            int lo = start_trim, hi = end_trim == 0 ? 0 : m - end_trim;   // seq[start_trim:-end_trim]
            lo = min(lo, m); hi = max(hi, lo);
            if (m > min(gnc, cap)) { status |= 1; lo = hi = 0; }
            out_len = hi - lo;
            if (P.compute_q && m > 0 && !(status & 1)) {
                // ---- S5 q-scores (py/tksm_badread.py:607-655): align read vs fragment with path
                const AlnOut a = band_align<1, true>(frag, L, N, owner, m, lane, trace);
                wave_sync();
                int mt = 0, cols = 0;
                const bool banded = !is_inf(a.dist);
                if (!banded) status |= full_align_wave<1>(frag, L, N, m, lane, O, mt, cols, popd) ? 16u : 4u;
                if (banded) {
                    int rr2 = L, j = m, dpend = 0;
                    while (rr2 > 0 || j > 0) {
                        int mv;                                  // 0 up, 1 left, 2 diagonal
                        if (j == 0) mv = 0;
                        else if (rr2 == 0) mv = 1;
                        else {
                            const int t = max(1, (int)owner[j - 1] + 1 - 31);
                            const int bb = rr2 - t;
                            if (bb > 63) mv = 0;                 // virtual cell below the window
                            else {
                                const unsigned long long um = trace[2 * j], lm = trace[2 * j + 1];
                                mv = ((lm >> bb) & 1ull) ? 1 : (((um >> bb) & 1ull) ? 0 : 2);
                            }
                        }
                        if (cols > L + m) { status |= 4; break; }   // cannot happen with a consistent trace
                        cols++;
                        if (mv == 1) {                           // read-only base: 'I'
                            if (lane == 0) popd[j - 1] = (uint8_t)(2 | (min(dpend, 63) << 2));
                            dpend = 0; j--;
                        } else if (mv == 0) {                    // fragment-only base: 'D'
                            dpend++; rr2--;
                        } else {
                            const bool eq = frag[rr2 - 1] == N[j - 1];
                            mt += eq;
                            if (lane == 0) popd[j - 1] = (uint8_t)((eq ? 0 : 1) | (min(dpend, 63) << 2));
                            dpend = 0; rr2--; j--;
                        }
                    }
                }
                identity = cols ? (double)mt / (double)cols : 0.0;
                wave_sync();
                const int margins = (QM.kmer_size - 1) / 2;
                const uint32_t hmask = (uint32_t)QM.n_slots - 1u;
                for (int i2 = lo + lane; i2 < hi; i2 += 64) {
                    int s0 = i2 - margins, e0 = i2 + margins;
                    while (s0 < 0 || e0 >= m) { s0++; e0--; }
                    int row = -1;
                    for (;;) {                                   // get_qscore :584-598
                        uint64_t key = 0; int len = 0; bool ok = true;
                        for (int x2 = s0; x2 <= e0; x2++) {
                            if (x2 > s0) {
                                const int dd = popd[x2 - 1] >> 2;
                                if (len + dd > 29) { ok = false; break; }
                                key |= ((1ull << (2 * dd)) - 1ull) << (2 * len); len += dd;
                            }
                            if (len >= 29) { ok = false; break; }
                            key |= (uint64_t)(popd[x2] & 3) << (2 * len); len++;
                        }
                        if (ok) {
                            key |= (uint64_t)len << 58;
                            uint32_t s = (uint32_t)qs_hash(key) & hmask;
                            for (;;) {
                                const uint64_t kk = QM.keys[s];
                                if (kk == key) { row = (int)s; break; }
                                if (kk == 0) break;
                                s = (s + 1) & hmask;
                            }
                        }
                        if (row >= 0 || s0 == e0) break;
                        s0++; e0--;
                    }
                    uint8_t q = 0;
                    if (row >= 0) {
                        const uint32_t w = philox(P.seed, g, ST_QUAL, (uint32_t)i2).x;
                        const uint32_t off = QM.row_off[row], cnt = QM.row_cnt[row];
                        uint32_t a = 0;
                        while (a + 1 < cnt && !(w < QM.cdf_pool[off + a])) a++;
                        q = QM.q_pool[off + a];
                    }
                    out_qual[i2 - lo] = (uint8_t)(q + 33);
                }
            } else {
                identity = 1.0 - errors / frag_len;             // :442-445
            }
            for (int t = lo + lane; t < hi; t += 64) out_seq[t - lo] = N[t];
        }

        // ---- per-read results; record length of py/sequence.py:252-288
        if (P.quirk_perfect) identity = 1.0;
        if (lane == 0) {
            const int efl = P.quirk_perfect ? out_len : raw_len - tail;
            const long long h = pct_hundredths(identity);
            const uint32_t idl = B.ids[2 * r + 1];
            // '@' uuid(36) ' length=' n ' error_free_length=' n ' read_identity=' x.xx '% molecule_id=' id '\n'
            uint64_t rec = 1 + 36 + 8 + ndigits((unsigned)out_len) + 19 + ndigits((unsigned)efl) + 15 +
                           ndigits((unsigned long long)(h / 100)) + 3 + 14 + idl + 1;
            rec += (uint64_t)out_len + 1;
            if (P.fastq) rec += 2 + (uint64_t)out_len + 1;
            O.out_len[r] = (uint32_t)out_len;
            O.identity[r] = identity;
            O.rec_len[r] = rec;
            O.status[r] |= status;
            if (O.istats) {
                int32_t* s = O.istats + 16 * r;
                s[0] = st_draws; s[1] = st_changes; s[2] = st_aligns; s[3] = L; s[4] = st_newlen;
                s[5] = st_strim; s[6] = st_etrim; s[7] = (int32_t)status;
                O.dstats[2 * r] = errors; O.dstats[2 * r + 1] = target;
            }
        }
        wave_sync();
    }
}

// ================================================================================================
// Fast Badread pipeline: k_init -> rounds of { k_loop / k_loopw (error loop up to the next identity
// re-estimation) -> k_alnf (windows decoded and aligned, bit-parallel banded, one LANE per alignment) }
// -> k_qjobs + k_alnf (q-score alignments) -> k_err (q-scores, trims, output).
// Reads whose fragment holds a non-ACGT byte (or whose alignment leaves the band representation)
// are routed to the byte-exact wave-wide path (k_simulate over slow_list).  Same specification,
// same results, bit for bit.
// ================================================================================================
// first alternative a with w < cdf[a] (cumulative, non-decreasing), or na if none.  Device rows are padded to
// 32 thresholds (128 B, filled with 0xffffffff): ~81 % of the draws are settled by the first threshold alone
// (the k-mer itself), only the rest fetch the row, with independent 16-byte loads.
DEV int cdf_pick(const uint32_t* cdf32, uint32_t pself, uint32_t ptot, int na, uint32_t w) {
    if (w < pself) return 0;                 // the k-mer itself
    if (!(w < ptot)) return na;              // residual mass: random change
    const uint4* c4 = reinterpret_cast<const uint4*>(cdf32);
    int a = 0;
    {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = c4[q];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a += !(w < v[q].x) ? 1 : 0; a += !(w < v[q].y) ? 1 : 0; a += !(w < v[q].z) ? 1 : 0; a += !(w < v[q].w) ? 1 : 0;
        }
    }
    if (a == 16) {                           // beyond the 16 most likely alternatives: second half of the row
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = c4[4 + q];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a += !(w < v[q].x) ? 1 : 0; a += !(w < v[q].y) ? 1 : 0; a += !(w < v[q].z) ? 1 : 0; a += !(w < v[q].w) ? 1 : 0;
        }
    }
    return min(a, na);
}

DEV void go_slow(const FastBuffers& FB, uint64_t r, int lane, int cause) {
    if (lane == 0) {
        FB.state[r].slow = 1;
        atomicAdd(&FB.counters[4 + cause], 1u);            // diagnostics: 0/3 alignment left the band, 1/2 window shift > 15
        const uint32_t idx = atomicAdd(&FB.counters[2], 1u);
        FB.slow_list[idx] = (uint32_t)r;
    }
}

// Per-read state rows are ragged: read r owns the 64-position blocks row64[r] .. row64[r + 1] - 1 of st_frag / st_nb (its padded
// fragment and at least one spare block), as many pairs of plane words + 8, and four packed words per block + 4 -- one 100 kb molecule
// among a million short ones costs its own row, not a million rows of its length.
DEV uint8_t* frag_row(const FastBuffers& FB, uint64_t r) { return FB.st_frag + (size_t)FB.row64[r] * 64; }
DEV uint16_t* nb_row(const FastBuffers& FB, uint64_t r) { return FB.st_nb + (size_t)FB.row64[r] * 64; }
DEV unsigned long long* planes_row(const FastBuffers& FB, uint64_t r) { return FB.st_fplanes + 2 * ((size_t)FB.row64[r] + 8 * r); }
DEV int planes_words(const FastBuffers& FB, uint64_t r) { return (int)(FB.row64[r + 1] - FB.row64[r]) + 8; }      // pairs {lo, hi}
DEV uint32_t* frag2_row(const FastBuffers& FB, uint64_t r) { return FB.st_frag2 + 4 * ((size_t)FB.row64[r] + r); }
DEV int frag2_words(const FastBuffers& FB, uint64_t r) { return 4 * (int)(FB.row64[r + 1] - FB.row64[r]) + 4; }

// a read's visits are ~ 0.14 x length x (1 - target identity): 8 bins per factor of two of that score (monotone; host and kernels
// only compare bins)
DEV uint32_t early_bin(int L, double target) {
    const float sc = (float)L * (float)fmax(0.0, 1.0 - target);
    return (uint32_t)min(255, max(0, (int)(8.0f * __log2f(1.0f + sc))));
}
// ---- k_init: splice (S1), flanks, target identity, 2-bit planes of the fragment, classification
// (seven waves per SIMD: 71 registers, no spills -- once the modification loop below is kept from unrolling; the compiler's own choice was 99
// registers, four waves: 7.35 -> 5 ms per step.  Eight waves -- 52 registers -- spill six since the literal segments have a path of their own)
__global__ __launch_bounds__(256, 7) void k_init(BatchView B, RefView R, ErrModelView EM, IdentView IM, SimParams P,
                                               SimBuffers O, FastBuffers FB) {
#ifndef TKSM_ABLATE
    __builtin_amdgcn_s_setprio(2);                            // (a latency-bound kernel beside the alignment kernel's always-ready waves: see k_loopw)
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpw = blockDim.x >> 6;
    const uint64_t r = (uint64_t)blockIdx.x * wpw + wave;
    if (r >= B.n_reads) return;
    uint8_t* frag = lds_raw + (size_t)wave * P.lcap;
    const int k = EM.k;
    const uint64_t g = P.first_read + r * P.stride;
    const int raw_len = (int)O.raw_len[r];
    const int L = raw_len + 2 * k;
    uint32_t status = 0;
    {
        const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
        int o = k;
        for (uint32_t ii = 0; ii < ic; ii++) {
            const Ivl iv = load_interval(B, R, ib + ii);
            const int len = (int)iv.len;
            if (iv.literal) {
                // literal segments (barcode, UMI, polyA: tens of bytes): a byte per lane -- as 16-byte pieces a single lane copied a barcode one
                // dependent byte load after the other, three literals of an scRNA-like molecule cost as much as the rest of the kernel
                const uint64_t lg = iv.gbase + iv.s;
                for (uint32_t t = (uint32_t)lane; t < (uint32_t)len; t += 64u) {
                    const uint8_t bch = upper(B.litpool[lg + (iv.minus ? (uint32_t)len - 1u - t : t)]);
                    frag[o + t] = iv.minus ? comp(bch) : bch;
                }
            } else
            for (uint32_t t0 = 16u * lane; t0 < (uint32_t)len; t0 += 1024u)
                splice_piece(B, R, iv.gbase + iv.s, (uint32_t)len, false, iv.minus, t0, frag + o + t0);
            wave_sync();
            // (not unrolled: with eight iterations' loads in flight the kernel took 99 registers -- four waves per SIMD -- instead of 80 -- six)
#pragma unroll 1
            for (uint32_t mi = iv.mod_begin; mi < iv.mod_end; mi++) {
                const uint32_t mp = B.mods[2ull * mi], mc = B.mods[2ull * mi + 1];
                if (mp >= (uint32_t)len) { status |= 2; continue; }
                if (lane == 0) frag[o + (iv.minus ? len - 1 - (int)mp : (int)mp)] = iv.minus ? comp((uint8_t)mc) : (uint8_t)mc;
            }
            o += len;
        }
        if (O.tail_len && O.tail_len[r]) tail_fill(O.tail_chain, P.seed, g, frag + o, (int)O.tail_len[r], lane);
    }
    {
        const Ph4 pad = philox(P.seed, g, ST_PAD, 0);
        if (lane < k) {
            frag[lane] = base_char((int)(pad.x >> (2 * lane)));
            frag[k + raw_len + lane] = base_char((int)(pad.y >> (2 * lane)));
        }
    }
    double target;
    if (IM.constant) target = IM.value;
    else {
        const uint32_t u = philox(P.seed, g, ST_IDENT, 0).x;
        const uint32_t idx = u >> 16;
        const double fr = (double)(u & 0xffffu) * (1.0 / 65536.0);
        const double qa = IM.qtab[idx], qb = IM.qtab[idx + 1];
        target = IM.value * (qa + (qb - qa) * fr);
    }
    wave_sync();
    uint8_t* gfrag = frag_row(FB, r);
    unsigned long long* fpl = planes_row(FB, r);
    const int fw_r = planes_words(FB, r), fw2_r = frag2_words(FB, r);
    // the fragment goes to HBM as one 2-bit code per byte ("ACGT" -> 0..3: bits 1 and 2 of the letter, xor-ed), in 8-byte
    // pieces (the slot is a multiple of 8 long; bytes past L are never used).  Reads with other letters never use it:
    // they take the wave-wide kernel, which splices its own fragment.
    for (int t = 8 * lane; t < L; t += 512) {
        uint2 c = *reinterpret_cast<const uint2*>(frag + t);
        c.x = ((c.x >> 1) ^ (c.x >> 2)) & 0x03030303u; c.y = ((c.y >> 1) ^ (c.y >> 2)) & 0x03030303u;
        *reinterpret_cast<uint2*>(gfrag + t) = c;
    }
    // the slot codes start out pristine: length 1, the original base as the slot's only symbol, bit 15 (changed) clear -- so that the
    // alignment kernel decodes every slot the same way (k_alnf); 8 slots (16 bytes) per lane and step
    {
        uint16_t* gnb0 = nb_row(FB, r);
        for (int t = 8 * lane; t < L; t += 512) {
            uint2 c = *reinterpret_cast<const uint2*>(frag + t);
            c.x = ((c.x >> 1) ^ (c.x >> 2)) & 0x03030303u; c.y = ((c.y >> 1) ^ (c.y >> 2)) & 0x03030303u;
            auto two = [](uint32_t b0, uint32_t b1) {            // two slot codes from two 2-bit base codes: 0x1000 | low bit | high bit << 5
                return (0x1000u | (b0 & 1u) | ((b0 >> 1) << 5)) | ((0x1000u | (b1 & 1u) | ((b1 >> 1) << 5)) << 16);
            };
            uint4 o;
            o.x = two(c.x & 3u, (c.x >> 8) & 3u); o.y = two((c.x >> 16) & 3u, (c.x >> 24) & 3u);
            o.z = two(c.y & 3u, (c.y >> 8) & 3u); o.w = two((c.y >> 16) & 3u, (c.y >> 24) & 3u);
            if (t + 8 > L) {                                      // slots past the fragment stay zero
                const int keep = L - t;
                uint32_t* ow = &o.x;
                for (int x2 = 0; x2 < 8; x2++) if (x2 >= keep) ow[x2 >> 1] &= (x2 & 1) ? 0x0000ffffu : 0xffff0000u;
            }
            *reinterpret_cast<uint4*>(gnb0 + t) = o;
        }
    }
    // ... and once more packed, 16 bases per word with the first base in the top bits, for the error loop (k_loop cuts a
    // k-mer's table index out of two consecutive words); bases past L are zero
    {
        uint32_t* f2 = frag2_row(FB, r);
        for (int w = lane; w < fw2_r; w += 64) {
            uint32_t word = 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int t = 16 * w + 4 * q;
                uint32_t c = t < P.lcap ? *reinterpret_cast<const uint32_t*>(frag + t) : 0u;
                if (t + 4 > L) c = t >= L ? 0u : c & (0xffffffffu >> (8 * (t + 4 - L)));
                const uint32_t codes4 = ((c >> 1) ^ (c >> 2)) & 0x03030303u;
                word |= ((codes4 * 0x40100401u) >> 24) << (24 - 8 * q);
            }
            f2[w] = word;
        }
    }
    // code planes, 64 positions per pair of words: lane q keeps the pair of word q, one 16-byte store per lane then;
    // the words past the fragment are zero (the alignment's window runs into them)
    bool dirty = false;
    const int nw = (L + 63) >> 6;
    for (int q0 = 0; q0 < fw_r; q0 += 64) {
        unsigned long long mylo = 0ull, myhi = 0ull;
        for (int q = q0; q < min(q0 + 64, nw); q++) {
            const int p = q * 64 + lane;
            const bool valid = p < L;
            const uint8_t c = valid ? frag[p] : (uint8_t)'A';
            const int code = code_of(c);
            dirty |= code < 0;
            const unsigned long long lo = __ballot(valid && (code & 1)), hi = __ballot(valid && (code & 2));
            if (lane == q - q0) { mylo = lo; myhi = hi; }
        }
        if (q0 + lane < fw_r) { ulonglong2 v; v.x = mylo; v.y = myhi; *reinterpret_cast<ulonglong2*>(fpl + 2 * (q0 + lane)) = v; }
    }
    const bool slow = __ballot(dirty) != 0ull;
    if (lane == 0) {
        ReadState S;
        S.errors = 0.0; S.target = target; S.est = 0.0; S.change_count = 0; S.n_base = 0; S.aln_no = 0;
        S.resume_src = -1; S.resume_j = 0; S.stage = 0; S.pending = 0; S.slow = slow ? 1 : 0; S.early = 0;
        S.st_draws = 0; S.st_aligns = 0; S.job = 0; S.raw_len = raw_len; S.res_mt = 0; S.res_cols = 0; S.res_fail = 0; S.pad3 = 0;
        FB.state[r] = S;
        O.status[r] |= status;
        if (slow) { const uint32_t idx = atomicAdd(&FB.counters[2], 1u); FB.slow_list[idx] = (uint32_t)r; }
        else if (FB.early_hist) atomicAdd(&FB.early_hist[(blockIdx.x & 63u) * 256u + early_bin(L, target)], 1u);    // (64 copies: a batch's reads fall into a dozen bins)
    }
}

// per-read results shared by both stages of k_err (py/sequence.py:252-288 record length)
DEV void finish_read(const BatchView& B, const SimParams& P, const SimBuffers& O, uint64_t r, int raw_len, int out_len,
                     double identity, uint32_t status, int st_draws, int st_changes, int st_aligns, int L, int st_newlen,
                     int st_strim, int st_etrim, double errors, double target, int lane) {
    if (lane != 0) return;
    if (O.tail_len && !P.quirk_perfect) raw_len -= (int)O.tail_len[r];      // error_free_length = len(raw_seq), py/sequence.py:254
    const long long h = pct_hundredths(identity);
    const uint32_t idl = B.ids[2 * r + 1];
    uint64_t rec = 1 + 36 + 8 + ndigits((unsigned)out_len) + 19 + ndigits((unsigned)raw_len) + 15 +
                   ndigits((unsigned long long)(h / 100)) + 3 + 14 + idl + 1;
    rec += (uint64_t)out_len + 1;
    if (P.fastq) rec += 2 + (uint64_t)out_len + 1;
    O.out_len[r] = (uint32_t)out_len;
    O.identity[r] = identity;
    O.rec_len[r] = rec;
    O.status[r] |= status;
    if (O.istats) {
        int32_t* s = O.istats + 16 * r;
        s[0] = st_draws; s[1] = st_changes; s[2] = st_aligns; s[3] = L; s[4] = st_newlen;
        s[5] = st_strim; s[6] = st_etrim; s[7] = (int32_t)status;
        O.dstats[2 * r] = errors; O.dstats[2 * r + 1] = target;
    }
}

// A draw as the fast pipeline carries it: the alternative's eight 16-bit slot encodings (length << 12 | symbols, planar: low
// bits of the up to 5 symbols in bits 0..4, high bits in bits 5..9; bit
// 15 = the slot differs from the original base), ErrModelView::alts_enc for a model alternative, random_change_enc for
// add_one_random_change.  A slot is applied if it differs and the position is still pristine (py/tksm_badread.py:378-390).
DEV uint32_t draw_slot(const uint4& A, int jj) {
    const uint32_t w = (jj >> 1) == 0 ? A.x : (jj >> 1) == 1 ? A.y : (jj >> 1) == 2 ? A.z : A.w;
    return (w >> (16 * (jj & 1))) & 0xffffu;
}

// add_one_random_change (py/tksm_badread.py:199-213) on the k-mer with index kidx (first base in the high bits): every
// slot keeps its base except slot pos: substitution by the r3-th next base, insertion of base4 before / after, deletion
DEV uint32_t planar1(uint32_t a) { return (a & 1u) | ((a >> 1) << 5); }                                   // one symbol
DEV uint32_t planar2(uint32_t a, uint32_t b) { return (a & 1u) | ((b & 1u) << 1) | ((a >> 1) << 5) | ((b >> 1) << 6); }   // two symbols
DEV uint4 random_change_enc(int kidx, int k, uint32_t type, uint32_t pos, uint32_t base4, uint32_t side, uint32_t r3) {
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t kc = j < k ? (uint32_t)(kidx >> (2 * (k - 1 - j))) & 3u : 0u;
        uint32_t v = (1u << 12) | planar1(kc);
        if ((uint32_t)j == pos)
            v = type == 0 ? 0x8000u | (1u << 12) | planar1((kc + r3) & 3u)
              : type == 1 ? 0x8000u | (2u << 12) | (side ? planar2(kc, base4) : planar2(base4, kc))
                          : 0x8000u;
        e[j] = j < k ? v : 0u;
    }
    return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

// all k slots of a lane's own draw: am = applied mask, lens = 3 bits per slot, e = 16-bit encodings (bit 15 set)
struct SlotEval { uint32_t am, lens; uint32_t e[4]; };
DEV SlotEval eval_draw(const uint16_t* nb, int k, int i, const uint4& A, bool acc) {
    SlotEval r;
    r.am = 0; r.lens = 0;
    r.e[0] = A.x | 0x80008000u; r.e[1] = A.y | 0x80008000u; r.e[2] = A.z | 0x80008000u; r.e[3] = A.w | 0x80008000u;
    if (acc) {
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            if (jj < k) {
                const uint32_t e = draw_slot(A, jj);
                if ((e >> 15) && (nb[i + jj] & 0x8000u) == 0) r.am |= 1u << jj;
                r.lens |= ((e >> 12) & 7u) << (3 * jj);
            }
        }
    }
    return r;
}

// joined length of slots [p0, p0+n)
DEV int joined_len(const uint16_t* nb, int p0, int n, int lane) {
    int m = 0;
    for (int q = 0; q < n; q += 64) {
        const int p = q + lane;
        const int len = p < n ? slot_len(nb[p0 + p]) : 0;
        int total;
        (void)prefix_small(len, total);
        m += total;
    }
    return m;
}

// Walks the slots [0, n) once, slot per lane, and writes the joined bases [lo, hi) to out_seq (the last visit).
DEV void join_out(const uint8_t* frag, const uint16_t* nb, int n, uint8_t* out_seq, int lo, int hi, int lane) {
    int base = 0;
    for (int q = 0; q < n; q += 64) {
        const int p = q + lane;
        uint32_t code = 0; int len = 0; uint8_t orig = 0;
        if (p < n) { code = nb[p]; len = slot_len(code); orig = frag[p]; }
        int total;
        const int off = base + prefix_small(len, total);
        // the slot's symbols, 2 bits each, planar: the original base of a pristine slot, else the stored codes
        const uint32_t syms = code ? code & 0x3ffu : planar1((uint32_t)orig);
        for (int x2 = 0; x2 < len; x2++) { const int c = off + x2; if (c >= lo && c < hi) out_seq[c - lo] = base_char((int)(((syms >> x2) & 1u) | (((syms >> (5 + x2)) & 1u) << 1))); }
        base += total;
    }
}

// ---- k_loop: the error loop (py/tksm_badread.py:351-403), one LANE per read.
// A read's loop is a serial chain: draw -> k-mer -> alternative -> (19 % of the draws) slots applied one by one, each
// adding est^1.5-weighted errors that decide when the loop stops.  A wave cannot make one chain faster, but it can run 64
// of them: every lane owns one read from its current draw to its next re-estimation point (every 25th applied change) or to
// the end of its loop, in plain per-lane code; the wave leaves when all of its reads have got there.
// A draw's table reads depend on each other (k-mer -> thresholds -> alternative), and a lane cannot hide its own latency, so
// draws are taken four at a time: the four k-mers, then the four first-level threshold entries, then -- only for the draws
// that change something -- the 8 thresholds of the segment the draw falls in and the slot codes under the k-mer, then the
// alternatives' slot encodings: three rounds of loads per four draws.  The draws are then applied in order; whatever
// follows a stop (re-estimation point, end of the loop) is dropped and drawn again on the next visit.
// Per lane in LDS: the first Wl words of the padded fragment at 2 bits per base (longer fragments read the rest from HBM); the
// slot codes stay in HBM (read and written only by the draws that change something).
// A read that stops at a re-estimation point gets an alignment job here (id, meta record: k_alnf decodes and aligns its window
// it; pending = 1); one whose loop has ended waits in stage 3 for its q-score job and its last visit (k_qjobs, k_err).
#ifndef TKSM_LOOP_N                          // (diagnostic builds of tools/exp_ab.sh vary them)
#define TKSM_LOOP_N 16
#define TKSM_LOOP_K 3
#endif
constexpr int LOOP_N = TKSM_LOOP_N, LOOP_K = TKSM_LOOP_K;      // draws per pass of k_loop; changing draws applied per pass
__global__ __launch_bounds__(64) void k_loop(ErrModelView EM, SimParams P, FastBuffers FB, const uint32_t* __restrict__ order,
                                              uint32_t begin, uint32_t count, int Wl, int from_jobs, uint32_t c0, uint32_t c1) {
#ifndef TKSM_ABLATE
    __builtin_amdgcn_s_setprio(2);                            // (a latency-bound kernel beside the alignment kernel's always-ready waves: see k_loopw)
#endif
    uint32_t* lf = reinterpret_cast<uint32_t*>(lds_raw);      // [Wl][64]: word w of lane l at w * 64 + l (conflict-free)
    const int lane = threadIdx.x;
    const uint32_t widx = blockIdx.x * 64u + (uint32_t)lane;
    bool act = widx < count;
    uint32_t r = 0, rc = 0;                                   // the read and its range of the sorted order
    if (act) {
        if (!from_jobs) { r = order[begin + widx]; rc = (begin + widx) / FB.rs; }
        else {
            const uint32_t target = FB.prefix[c0] + widx;
            uint32_t lo2 = c0, hi2 = c1 - 1;
            while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2 + 1) >> 1; if (FB.prefix[mid] <= target) lo2 = mid; else hi2 = mid - 1; }
            r = FB.prev_meta[4ull * (FB.base_prev[lo2] + (target - FB.prefix[lo2]))];
            rc = lo2;
        }
    }
    ReadState* sp = FB.state + r;
    ReadState S{};
    if (act) S = *sp;
    act = act && S.stage == 0 && !S.slow && !S.early;
    const int k = EM.k;
    const int L = S.raw_len + 2 * k;
    const uint32_t* f2 = frag2_row(FB, r);
    // rows are zero beyond the fragment; Wl is a multiple of 4 <= fw2; unconditional loads, four in flight (idle lanes read
    // read 0's row)
    for (int w = 0; w < Wl; w += 16) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const uint4*>(f2 + min(w + 4 * q, Wl - 4));
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ww = min(w + 4 * q, Wl - 4);
            lf[(ww + 0) * 64 + lane] = v[q].x; lf[(ww + 1) * 64 + lane] = v[q].y; lf[(ww + 2) * 64 + lane] = v[q].z; lf[(ww + 3) * 64 + lane] = v[q].w;
        }
    }
    uint16_t* gnb = nb_row(FB, r);
    const uint64_t g = P.first_read + (uint64_t)r * P.stride;
    const double frag_len = (double)L, target = S.target;
    const double rcp_len = rcp_refined(act ? frag_len : 1.0);
    double errors = S.errors;
    int change_count = S.change_count;
    uint32_t n = S.n_base, aln_no = S.aln_no;
    int resume_j = S.resume_j, st_draws = S.st_draws;
    double est_keep = S.est;
    if (act && S.pending == 1) {                               // apply the re-estimation result (py/tksm_badread.py:412-432)
        const uint32_t mt = S.res_mt, cols = S.res_cols;
        if (S.res_fail) {
            // the alignment left the band representation: the read takes the byte-exact wave-wide kernel
            sp->slow = 1;
            atomicAdd(&FB.counters[4], 1u);
            FB.slow_list[atomicAdd(&FB.counters[2], 1u)] = r;
            act = false;
        } else {
            const double ident = cols ? (double)mt / (double)cols : 0.0;
            if (L <= 1000) errors = (1.0 - ident) * frag_len;
            else {
                const double estimated = (1.0 - ident) * frag_len;
                const double weight = 1000.0 / frag_len;
                errors = estimated * weight + errors * (1.0 - weight);
            }
            aln_no++;
        }
    }
    const uint32_t kmer_range = (uint32_t)(L - k);            // max_kmer_index + 1
    const long long loop_limit = 100ll * L;
    const uint32_t kmask = (1u << (2 * k)) - 1u;
    int cc25 = change_count % 25;
    enum { RUN = 0, NEED_ALN = 1, DONE = 2, IDLE = 3 };
    int st = act ? RUN : IDLE;
#ifdef TKSM_ABLATE
    if (P.ablate == 33) st = IDLE;                              // timing experiment: prologue only
#endif
    // ---- passes of LOOP_N draws.  Phase 1, all LOOP_N draws: generator, k-mer, the k-mer's own threshold t0 (one round of gathers
    // for all of them) -> which draws change something (~19 %).  Phase 2, the first LOOP_K of those only: the generator once more
    // (its other words), then segment thresholds + the slot codes under the k-mer, then the alternative -- the rounds of dependent
    // gathers that go to the read's own rows and the big tables are paid once per pass, not once per four draws.  Phase 3 applies
    // them in draw order; the no-op draws in between only advance the draw counter (the stop rules move with applied changes
    // alone).  Draws behind the LOOP_K-th changing one, a re-estimation point or the end of the loop are dropped and drawn again.
    // (the k-mer index of a chosen draw is cut out of the fragment words again in phase 2: a second array for the indices would make
    // the wave's LDS 22 KB -- seven waves per CU; at 20 KB it is eight, two on every SIMD)
    uint32_t* ldi = lf + (size_t)Wl * 64;                     // [LOOP_N][64] draw positions of the pass
    while (__ballot(st == RUN) != 0ull) {
        if (st != RUN) continue;
        const uint32_t navail = (uint32_t)max(0ll, min((long long)LOOP_N, loop_limit - (long long)n));     // draws of this pass below the loop limit
        uint32_t mask = 0u;                                    // draws of the pass that change something
        {
            uint32_t dwv[LOOP_N], t0v[LOOP_N];
#pragma unroll
            for (int b = 0; b < LOOP_N; b++) {
                const Ph4 d = philox(P.seed, g, ST_DRAW, n + (uint32_t)b);
                const int di1 = (int)__umulhi(d.x, kmer_range);
                dwv[b] = d.y;
                const int w = di1 >> 4, o = di1 & 15;
#ifdef TKSM_ABLATE
                // timing experiments (results are wrong): 35 / 36: every k-mer from the words in LDS; 34 / 36: no first-level threshold gather
                const bool abl_lds = P.ablate == 35 || P.ablate == 36, abl_t0 = P.ablate == 34 || P.ablate == 36;
                const uint32_t hi = (abl_lds || w < Wl) ? lf[min(w, Wl - 1) * 64 + lane] : f2[w], lo = (abl_lds || w + 1 < Wl) ? lf[min(w + 1, Wl - 1) * 64 + lane] : f2[w + 1];
#else
                const uint32_t hi = w < Wl ? lf[w * 64 + lane] : f2[w], lo = w + 1 < Wl ? lf[(w + 1) * 64 + lane] : f2[w + 1];
#endif
                const uint32_t kx = (uint32_t)(mk64(hi, lo) >> (64 - 2 * o - 2 * k)) & kmask;
                ldi[b * 64 + lane] = (uint32_t)di1;
#ifdef TKSM_ABLATE
                t0v[b] = abl_t0 ? 0xCF000000u : EM.pt0[kx];
#else
                t0v[b] = EM.pt0[kx];
#endif
            }
#pragma unroll
            for (int b = 0; b < LOOP_N; b++) {
                const bool chg = EM.type == 0 || !EM.alt0_noop || !(dwv[b] < t0v[b]);
                mask |= (chg && (uint32_t)b < navail) ? 1u << b : 0u;
            }
        }
        // ---- the first LOOP_K changing draws (relative index LOOP_N: none)
        int di[LOOP_K], kidx[LOOP_K];
        uint32_t bj[LOOP_K], dw[LOOP_K];
        bool rest;                                             // more changing draws than LOOP_K: the pass ends behind the last chosen one
        {
            uint32_t mm = mask;
#pragma unroll
            for (int j = 0; j < LOOP_K; j++) {
                bj[j] = mm ? (uint32_t)__builtin_ctz(mm) : (uint32_t)LOOP_N;
                mm &= mm - 1u;
                const uint32_t bb = min(bj[j], (uint32_t)LOOP_N - 1u);
                const Ph4 d = philox(P.seed, g, ST_DRAW, n + bb);
                dw[j] = d.y;
                di[j] = (int)ldi[bb * 64 + lane];
                const int w = di[j] >> 4, o = di[j] & 15;
                const uint32_t hi = w < Wl ? lf[w * 64 + lane] : f2[w], lo = w + 1 < Wl ? lf[(w + 1) * 64 + lane] : f2[w + 1];
                kidx[j] = (int)((uint32_t)(mk64(hi, lo) >> (64 - 2 * o - 2 * k)) & kmask);
            }
            rest = mm != 0u;
        }
        // ---- first-level thresholds {t0, t8, t16, t24} again (the lines are in the cache), segment thresholds, slot codes.
        // Every load below is unconditional and straight-line (a load inside a divergent region is waited for at the region's end);
        // lanes that do not need one read a common dummy address (one cache line for the whole wave).
        int cls[LOOP_K];                                      // 0 no-op, 1 alternative a (thresholds needed), 2 random change, 3 alternative 0
        uint4 seg[LOOP_K];
        int nab[LOOP_K];
#pragma unroll
        for (int j = 0; j < LOOP_K; j++) { seg[j] = EM.pseg[kidx[j]]; nab[j] = EM.max_alts; }
        if (!EM.uniform_nalts) {
#pragma unroll
            for (int j = 0; j < LOOP_K; j++) nab[j] = (int)EM.nalts[kidx[j]];
        }
        int sbase[LOOP_K];
        uint4 th0[LOOP_K], th1[LOOP_K];
        struct __attribute__((packed, aligned(4))) W5 { uint32_t v[5]; };
        W5 nbw[LOOP_K];                                       // the slot codes under the k-mer: 10 u16 from the even position below i
#pragma unroll
        for (int j = 0; j < LOOP_K; j++) {
            const bool have = bj[j] < (uint32_t)LOOP_N;
            cls[j] = !have ? 0 : EM.type == 0 ? 2 : (dw[j] < seg[j].x ? (EM.alt0_noop ? 0 : 3) : 1);
            sbase[j] = 8 * ((dw[j] < seg[j].y ? 0 : 1) + (dw[j] < seg[j].z ? 0 : 1) + (dw[j] < seg[j].w ? 0 : 1));
            const uint4* c4 = reinterpret_cast<const uint4*>(EM.cdf32 + (cls[j] == 1 ? (size_t)kidx[j] * 32 + sbase[j] : (size_t)0));
            th0[j] = c4[0]; th1[j] = c4[1];
            const uint16_t* gp = cls[j] != 0 ? gnb + (di[j] & ~1) : FB.st_nb;
            nbw[j] = *reinterpret_cast<const W5*>(gp);
        }
        // ---- the alternatives
        uint4 alt[LOOP_K];
#pragma unroll
        for (int j = 0; j < LOOP_K; j++) {
            size_t at = 0;
            if (cls[j] == 1 || cls[j] == 3) {
                const int na = nab[j];
                int a2 = 0;
                if (cls[j] == 1) {
                    const uint32_t w = dw[j];
                    a2 = sbase[j];
                    a2 += !(w < th0[j].x) ? 1 : 0; a2 += !(w < th0[j].y) ? 1 : 0; a2 += !(w < th0[j].z) ? 1 : 0; a2 += !(w < th0[j].w) ? 1 : 0;
                    a2 += !(w < th1[j].x) ? 1 : 0; a2 += !(w < th1[j].y) ? 1 : 0; a2 += !(w < th1[j].z) ? 1 : 0; a2 += !(w < th1[j].w) ? 1 : 0;
                    a2 = min(a2, na);
                }
                if (a2 == na) cls[j] = 2;                     // residual mass: add_one_random_change
                else if (a2 == 0 && EM.alt0_noop) cls[j] = 0;
                else at = (size_t)kidx[j] * EM.max_alts + a2;
            }
            alt[j] = EM.alts_enc[at];
        }
        // ---- apply in draw order.  `n` is the draw whose iteration comes next; n0 the pass's first draw; the pass holds the draws
        // n0 .. n0 + end - 1
        const uint32_t n0 = n;
        const uint32_t end = rest ? bj[LOOP_K - 1] + 1u : navail;
        uint32_t wrote = 0u;                                  // chosen draws of this pass that wrote slots
        // the top of an iteration (:353-367); false: the loop has ended.  The estimate only moves when a change is applied, so one
        // look covers a run of draws that change nothing and the changing draw behind them
        auto top_of_iteration = [&](double& est) -> bool {
            est = 1.0 - div_inrange(errors, frag_len, rcp_len);
            if ((double)change_count > 0.9 * frag_len || est <= target) { st = DONE; st_draws = (int)n; return false; }
            if ((long long)n + 1 > loop_limit) { st = DONE; st_draws = (int)loop_limit; return false; }
            return true;
        };
#pragma unroll
        for (int j = 0; j <= LOOP_K; j++) {
            if (st != RUN) break;
            double est = est_keep;
            if (j == LOOP_K || bj[j] >= (uint32_t)LOOP_N) {
                // behind the last chosen draw: the rest of the pass changes nothing
                if (n - n0 < end) { if (!top_of_iteration(est)) break; n = n0 + end; }
                if (!rest && navail < (uint32_t)LOOP_N) (void)top_of_iteration(est);      // the next iteration is the one the loop limit ends
                break;
            }
            if (resume_j == 0 && !top_of_iteration(est)) break;
            n = n0 + bj[j];                                   // (the draws in front of this one change nothing)
            if (cls[j] != 0) {
                uint4 A = alt[j];
                if (cls[j] == 2) {
                    // add_one_random_change (:199-213): one slot of the k-mer changes -- substitution by the r3-th next base,
                    // insertion of base4 before / after, deletion; the other slots keep their base (encoding 0: never applied)
                    // (the draw's other two words: generated again here -- the residual mass is rare, and six registers held for it
                    // across the gathers were what kept the kernel from a fourth wave per SIMD)
                    const Ph4 d2 = philox(P.seed, g, ST_DRAW, n0 + bj[j]);
                    const uint32_t dzj = d2.z, dvj = d2.w;
                    const uint32_t type = __umulhi(dzj, 3u), pos = __umulhi(dvj, (uint32_t)k);
                    const uint32_t base4 = dvj & 3u, side = (dvj >> 2) & 1u;
                    const uint32_t r3 = (((dzj & 0xffffu) * 3u) >> 16) + 1u;
                    const uint32_t kc = ((uint32_t)kidx[j] >> (2 * (k - 1 - (int)pos))) & 3u;
                    const uint32_t v = type == 0 ? 0x8000u | (1u << 12) | planar1((kc + r3) & 3u)
                                     : type == 1 ? 0x8000u | (2u << 12) | (side ? planar2(kc, base4) : planar2(base4, kc))
                                                 : 0x8000u;
                    const uint32_t w = v << (16 * (pos & 1u)), which = pos >> 1;
                    A = make_uint4(which == 0 ? w : 0u, which == 1 ? w : 0u, which == 2 ? w : 0u, which == 3 ? w : 0u);
                }
                // the slots that differ from the original base (bit 15 of their encodings), from the resume point on
                uint32_t dm = ((A.x >> 15) & 1u) | ((A.x >> 31) << 1) | (((A.y >> 15) & 1u) << 2) | ((A.y >> 31) << 3) |
                              (((A.z >> 15) & 1u) << 4) | ((A.z >> 31) << 5) | (((A.w >> 15) & 1u) << 6) | ((A.w >> 31) << 7);
                dm &= ((1u << k) - 1u) & ~((1u << resume_j) - 1u);
                if (dm) {
                    // slot codes read before an overlapping earlier draw of this pass wrote: read them again (rare)
                    bool stale = false;
#pragma unroll
                    for (int j2 = 0; j2 < LOOP_K; j2++) if (j2 < j) stale |= ((wrote >> j2) & 1u) && abs(di[j] - di[j2]) < k;
                    if (stale) nbw[j] = *reinterpret_cast<const W5*>(gnb + (di[j] & ~1));
                    // in slot order (:378-403): applied if the position is still pristine
                    const double f15 = est * sqrt_inrange(est);
                    const int odd = di[j] & 1;
                    int stop_at = -1;
                    while (dm) {
                        const int jj = __builtin_ctz(dm);
                        dm &= dm - 1u;
                        const int ent = jj + odd, wi = ent >> 1;
                        const uint32_t cw2 = wi == 0 ? nbw[j].v[0] : wi == 1 ? nbw[j].v[1] : wi == 2 ? nbw[j].v[2] : wi == 3 ? nbw[j].v[3] : nbw[j].v[4];
                        if (((cw2 >> (16 * (ent & 1))) & 0x8000u) == 0u) {              // still pristine (bit 15: changed)
                            const uint32_t e = draw_slot(A, jj);
                            gnb[di[j] + jj] = (uint16_t)(e | 0x8000u);
                            wrote |= 1u << j;
                            change_count++;
                            const int len_j = (int)((e >> 12) & 7u);
                            errors += (double)(len_j < 2 ? 1 : len_j - 1) * f15;
                            if (++cc25 == 25) { cc25 = 0; stop_at = jj; dm = 0u; }       // ALIGNMENT_INTERVAL
                        }
                    }
                    if (stop_at >= 0) {
                        st = NEED_ALN;
                        if (stop_at + 1 < k) { resume_j = stop_at + 1; est_keep = est; }    // the rest of this draw follows the alignment
                        else { resume_j = 0; n++; }
                        break;
                    }
                }
            }
            resume_j = 0;
            n++;
        }
    }
    // ---- a read at a re-estimation point gets an alignment job: id from its range's counter (one atomic per range and wave),
    // meta record for k_alnf.  Window: the whole fragment, or a random 1000-base window of a
    // longer one (py/tksm_badread.py:405-432).
    int st_aligns = S.st_aligns;
    uint32_t job = 0;
    {
        bool need = st == NEED_ALN;
        unsigned long long todo = __ballot(need);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t lrc = (uint32_t)__shfl((int)rc, leader, 64);
            const unsigned long long grp = __ballot(need && rc == lrc);
            uint32_t base = 0;
            if (lane == leader) base = FB.base_cur[lrc] + atomicAdd(&FB.job_cnt[lrc * 32u], (uint32_t)__popcll(grp));
            base = (uint32_t)__shfl((int)base, leader, 64);
            if (need && rc == lrc) { job = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull)); need = false; }
            todo &= ~grp;
        }
    }
    if (st == NEED_ALN) {
        st_aligns++;
        uint32_t p0 = 0, nrows = (uint32_t)L;
        if (L > 1000) {
            p0 = __umulhi(philox(P.seed, g, ST_ALNPOS, aln_no).x, (uint32_t)(L - 1000 + 1));
            nrows = 1000u;
        }
        *reinterpret_cast<uint4*>(FB.job_meta + 4ull * job) = make_uint4(r, p0, nrows, 0u);
    }
    if (st == NEED_ALN || st == DONE) {
        sp->errors = errors; sp->est = est_keep; sp->change_count = change_count; sp->n_base = n; sp->aln_no = aln_no;
        sp->resume_src = -1; sp->resume_j = (int16_t)resume_j; sp->st_draws = st_draws; sp->st_aligns = st_aligns;
        sp->pending = st == NEED_ALN ? 1 : 0;
        sp->job = job;
        // a read whose loop has ended waits (stage 3): trims, q-score alignment and output of all reads run together after
        // the last regular round, at full occupancy (k_err)
        sp->stage = st == NEED_ALN ? 0 : 3;
        if (st == DONE) {
            atomicAdd(&FB.defer_cnt[rc * 32u], 1u);
            FB.defer_list[atomicAdd(&FB.counters[1], 1u)] = make_uint2(r, rc);
        }
    }
}

// ---- k_loopw: the same visit of the error loop as k_loop, one WAVE per read -- for the rounds in which few reads are left.
// A lane-per-read visit is a chain of ~40 batches of dependent table reads (1 - 2 ms whatever the number of reads); here the 64
// lanes take 64 consecutive draws at once (generator, k-mer, thresholds, alternative: three rounds of loads for all of them), and
// the draws that change something (~12 of 64) are then applied in draw order by the whole wave, on the read's slot codes staged
// in LDS (every applied slot is also written through to HBM).  What follows a stop (re-estimation point, end of the loop) is
// dropped and drawn again on the next visit, exactly as in k_loop: same draws, same order, same state records.
// TAIL (k_tail, the last rounds of a batch): the wave does not hand its read back at a re-estimation point -- it aligns the window
// itself (band_align, one lane per band row: ~0.1 us per column against the ~0.45 us of a lone lane-per-job wave) and goes on
// with the read's next visit, until the read's loop has ended: the stragglers of a batch (reads with a low target identity need
// two to three times the visits of the average read) finish in ONE launch instead of one host round trip + three latency-bound
// launches per visit.  Same draws, same windows, same preference: band_align is the alignment of the byte-exact kernel.  A window
// with more columns than the wave's LDS holds (TAIL_WCAP) takes the regular route (job, k_alnf) for that visit.
constexpr int TAIL_WCAP = 2048, TAIL_FCAP = 1024;
__host__ __device__ inline size_t tail_bitmap_bytes(int lcap) { return ((size_t)((lcap + 63) / 64) * 8 + 16 + 15) & ~(size_t)15; }
DEV int join_window_planar(const uint16_t* gnb, int p0, int n, uint8_t* N, uint16_t* owner, int ncap, int lane) {
    // the window's slot codes come from the read's row in HBM, all groups of 64 requested before the first is used (n <= 1024)
    uint32_t codes[16];
#pragma unroll
    for (int q = 0; q < 16; q++) codes[q] = 64 * q + lane < n ? (uint32_t)gnb[p0 + 64 * q + lane] : 0u;
    int base = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        if (64 * q >= n) break;
        const int p = 64 * q + lane;
        const uint32_t code = codes[q];
        const int len = (int)((code >> 12) & 7u);              // (0 past the window)
        int total;
        const int off = base + prefix_small(len, total);
        if (off + len <= ncap)
            for (int x2 = 0; x2 < len; x2++) {
                N[off + x2] = (uint8_t)(((code >> x2) & 1u) | (((code >> (5 + x2)) & 1u) << 1));     // 2-bit code, as st_frag has the fragment (band_align compares for equality only)
                owner[off + x2] = (uint16_t)p;
            }
        base += total;
    }
    return base;
}
template <bool TAIL>
__global__ __launch_bounds__(64) void k_loopw(ErrModelView EM, SimParams P, FastBuffers FB, const uint32_t* __restrict__ order,
                                               uint32_t begin, uint32_t count, int from_jobs, uint32_t c0, uint32_t c1, int lcap, int wcap) {
    uint16_t* nbl = reinterpret_cast<uint16_t*>(lds_raw);     // [lcap] slot codes of the read
    // TAIL: a bit per slot ("changed") instead of the codes -- the wave's LDS does not grow with the fragment, so the stragglers of
    // a batch of long molecules all get a wave at once -- | [TAIL_FCAP] fragment bytes of the window | N [TAIL_WCAP] | owner [TAIL_WCAP] u16
    uint32_t* bml = reinterpret_cast<uint32_t*>(lds_raw);
    uint8_t* Fw = lds_raw + tail_bitmap_bytes(lcap);
    uint8_t* Nw = Fw + TAIL_FCAP;
    uint16_t* ownw = reinterpret_cast<uint16_t*>(Nw + TAIL_WCAP);
    const int lane = threadIdx.x;
    const uint32_t widx = blockIdx.x;
    if (widx >= count) return;
    uint32_t r, rc;                                           // the read and its range of the sorted order (wave-uniform)
    // a straggler wave is a chain of dependent instructions that decides when the batch ends; beside the bulk kernels' waves (four per
    // SIMD, each ready every cycle) it would get a fifth of the issue slots it can use: highest wave priority
#ifndef TKSM_ABLATE                                           // (the diagnostic build runs them at the default priority: lognormal lengths, three contexts: 8.3 -> 8.0 M reads/s)
    // Wave priorities (s_setprio), measured with three contexts in flight on one box: the stragglers at 3 (skewed lengths 8.0 -> 8.3 M
    // reads/s), the other latency-bound kernels -- k_loop, k_loopw, k_err, k_init, k_emit, the full-width alignment passes -- at 2
    // beside the 14-row alignment pass at the default 0: 13.35 -> 13.7 M reads/s (3 instead of 2: no better)
    if (TAIL) __builtin_amdgcn_s_setprio(3);
#ifndef TKSM_ABLATE
    if (!TAIL) __builtin_amdgcn_s_setprio(2);
#endif
#endif
    const bool early_mode = TAIL && from_jobs == 3;           // the predicted stragglers, from their list (side stream, from round 0 on)
    if (early_mode) {
        if (widx >= FB.counters[27]) return;
        const uint2 e = FB.early_list[widx];
        r = e.x; rc = e.y;
    } else if (!from_jobs) { r = order[begin + widx]; rc = (begin + widx) / FB.rs; }
    else {
        const uint32_t target = FB.prefix[c0] + widx;
        uint32_t lo2 = c0, hi2 = c1 - 1;
        while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2 + 1) >> 1; if (FB.prefix[mid] <= target) lo2 = mid; else hi2 = mid - 1; }
        r = FB.prev_meta[4ull * (FB.base_prev[lo2] + (target - FB.prefix[lo2]))];
        rc = lo2;
    }
    r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r); rc = (uint32_t)__builtin_amdgcn_readfirstlane((int)rc);
    ReadState* sp = FB.state + r;
    const ReadState S = *sp;
    if (S.stage != 0 || S.slow || (S.early && !early_mode)) return;
    // an early read never has a job, and its kernel runs beside the rounds (which read the slow list's counter): what it hands to the
    // exact kernel goes on a list of its own, merged after the side kernel has ended
    auto to_exact_kernel = [&]() {
        if (!early_mode) { go_slow(FB, r, lane, 0); return; }
        if (lane == 0) { sp->slow = 1; FB.early_slow[atomicAdd(&FB.counters[26], 1u)] = r; }
    };
    const int k = EM.k;
    const int L = S.raw_len + 2 * k;
    const uint32_t* f2 = frag2_row(FB, r);
    uint16_t* gnb = nb_row(FB, r);
    if (!TAIL) {
        for (int t = 2 * lane; t < L; t += 128) *reinterpret_cast<uint32_t*>(nbl + t) = *reinterpret_cast<const uint32_t*>(gnb + t);   // (rows are padded to 8 slots)
    } else {
        // 32 slots per lane and step (rows are padded to 64 slots, zero past the fragment): bit 15 of each code
        for (int t = 32 * lane; t < L; t += 2048) {
            const uint4* c4 = reinterpret_cast<const uint4*>(gnb + t);
            uint32_t word = 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = c4[q];
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) word |= (((w4[e] >> 15) & 1u) | ((w4[e] >> 31) << 1)) << (8 * q + 2 * e);
            }
            bml[t >> 5] = word;
        }
    }
    wave_sync();
    const uint64_t g = P.first_read + (uint64_t)r * P.stride;
    const double frag_len = (double)L, target = S.target;
    const double rcp_len = rcp_refined(frag_len);
    double errors = S.errors;
    int change_count = S.change_count;
    uint32_t n = S.n_base, aln_no = S.aln_no;
    int resume_j = S.resume_j, st_draws = S.st_draws;
    double est_keep = S.est;
    if (S.pending == 1) {                                      // apply the re-estimation result (py/tksm_badread.py:412-432)
        if (S.res_fail) {
            // the alignment left the band representation: the read takes the byte-exact wave-wide kernel
            if (lane == 0) {
                sp->slow = 1;
                atomicAdd(&FB.counters[4], 1u);
                FB.slow_list[atomicAdd(&FB.counters[2], 1u)] = r;
            }
            return;
        }
        const uint32_t mt = S.res_mt, cols = S.res_cols;
        const double ident = cols ? (double)mt / (double)cols : 0.0;
        if (L <= 1000) errors = (1.0 - ident) * frag_len;
        else {
            const double estimated = (1.0 - ident) * frag_len;
            const double weight = 1000.0 / frag_len;
            errors = estimated * weight + errors * (1.0 - weight);
        }
        aln_no++;
    }
    const uint32_t kmer_range = (uint32_t)(L - k);            // max_kmer_index + 1
    const long long loop_limit = 100ll * L;
    const uint32_t kmask = (1u << (2 * k)) - 1u;
    int cc25 = change_count % 25;
    enum { RUN = 0, NEED_ALN = 1, DONE = 2 };
    int st = RUN;
    int st_aligns = S.st_aligns;
    const uint8_t* gfrag = frag_row(FB, r);
    if (TAIL && L <= 1000) { for (int t = lane; t < L; t += 64) Fw[t] = gfrag[t]; }     // (a longer fragment: the window of each alignment)
    for (;;) {                                               // (TAIL: one turn per visit of the read; else a single turn)
        while (st == RUN) {
            double est_cur = est_keep;
            if (resume_j == 0) {
                // stop rules at the top of an iteration (:353-367)
                est_cur = 1.0 - div_inrange(errors, frag_len, rcp_len);
                if ((double)change_count > 0.9 * frag_len || est_cur <= target) { st = DONE; st_draws = (int)n; break; }
            }
            // ---- 64 draws: lane l takes draw n + l
            const uint32_t nl = n + (uint32_t)lane;
            const bool live = (long long)nl + 1 <= loop_limit;
            const Ph4 d = philox(P.seed, g, ST_DRAW, nl);
            const int di = (int)__umulhi(d.x, kmer_range);
            const int w = di >> 4, o = di & 15;
            const int kidx = (int)((uint32_t)(mk64(f2[w], f2[w + 1]) >> (64 - 2 * o - 2 * k)) & kmask);
            const uint4 seg = EM.pseg[kidx];
            const int na = EM.uniform_nalts ? EM.max_alts : (int)EM.nalts[kidx];
            int cls = EM.type == 0 ? 2 : (d.y < seg.x ? (EM.alt0_noop ? 0 : 3) : 1);     // 0 no-op, 1 alternative a, 2 random change, 3 alternative 0
            const int sbase = 8 * ((d.y < seg.y ? 0 : 1) + (d.y < seg.z ? 0 : 1) + (d.y < seg.w ? 0 : 1));
            const uint4* c4 = reinterpret_cast<const uint4*>(EM.cdf32 + (cls == 1 ? (size_t)kidx * 32 + sbase : (size_t)0));
            const uint4 th0 = c4[0], th1 = c4[1];
            size_t at = 0;
            if (cls == 1 || cls == 3) {
                int a = 0;
                if (cls == 1) {
                    const uint32_t wv = d.y;
                    a = sbase;
                    a += !(wv < th0.x) ? 1 : 0; a += !(wv < th0.y) ? 1 : 0; a += !(wv < th0.z) ? 1 : 0; a += !(wv < th0.w) ? 1 : 0;
                    a += !(wv < th1.x) ? 1 : 0; a += !(wv < th1.y) ? 1 : 0; a += !(wv < th1.z) ? 1 : 0; a += !(wv < th1.w) ? 1 : 0;
                    a = min(a, na);
                }
                if (a == na) cls = 2;                             // residual mass: add_one_random_change
                else if (a == 0 && EM.alt0_noop) cls = 0;
                else at = (size_t)kidx * EM.max_alts + a;
            }
            uint4 A = EM.alts_enc[at];
            if (cls == 2) {
                // add_one_random_change (:199-213), as in k_loop
                const uint32_t type = __umulhi(d.z, 3u), pos = __umulhi(d.w, (uint32_t)k);
                const uint32_t base4 = d.w & 3u, side = (d.w >> 2) & 1u;
                const uint32_t r3 = (((d.z & 0xffffu) * 3u) >> 16) + 1u;
                const uint32_t kc = ((uint32_t)kidx >> (2 * (k - 1 - (int)pos))) & 3u;
                const uint32_t v = type == 0 ? 0x8000u | (1u << 12) | planar1((kc + r3) & 3u)
                                 : type == 1 ? 0x8000u | (2u << 12) | (side ? planar2(kc, base4) : planar2(base4, kc))
                                             : 0x8000u;
                const uint32_t wv = v << (16 * (pos & 1u)), which = pos >> 1;
                A = make_uint4(which == 0 ? wv : 0u, which == 1 ? wv : 0u, which == 2 ? wv : 0u, which == 3 ? wv : 0u);
            }
            // the slots that differ from the original base (bit 15 of their encodings)
            uint32_t dm = ((A.x >> 15) & 1u) | ((A.x >> 31) << 1) | (((A.y >> 15) & 1u) << 2) | ((A.y >> 31) << 3) |
                          (((A.z >> 15) & 1u) << 4) | ((A.z >> 31) << 5) | (((A.w >> 15) & 1u) << 6) | ((A.w >> 31) << 7);
            dm &= (1u << k) - 1u;
            if (!live || cls == 0) dm = 0u;
            unsigned long long mask = __ballot(dm != 0u);
            const unsigned long long dead = __ballot(!live);
            // ---- the draws that may change something, in draw order
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                const int ai = __builtin_amdgcn_readlane(di, src);
                uint4 As;
                As.x = (uint32_t)__builtin_amdgcn_readlane((int)A.x, src); As.y = (uint32_t)__builtin_amdgcn_readlane((int)A.y, src);
                As.z = (uint32_t)__builtin_amdgcn_readlane((int)A.z, src); As.w = (uint32_t)__builtin_amdgcn_readlane((int)A.w, src);
                uint32_t dms = (uint32_t)__builtin_amdgcn_readlane((int)dm, src);
                double est = est_cur;
                if (resume_j > 0) { dms &= ~((1u << resume_j) - 1u); est = est_keep; }      // (the first draw of the visit: lane 0)
                // in slot order (:378-403): applied if the position is still pristine
                const double f15 = est * sqrt_inrange(est);
                int stop_at = -1;
                while (dms) {
                    const int jj = __builtin_ctz(dms);
                    dms &= dms - 1u;
                    const int ps = ai + jj;
                    if (TAIL ? ((bml[ps >> 5] >> (ps & 31)) & 1u) == 0u : (nbl[ps] & 0x8000u) == 0) {
                        const uint32_t e = draw_slot(As, jj);
                        if (lane == 0) {
                            if (TAIL) bml[ps >> 5] |= 1u << (ps & 31); else nbl[ps] = (uint16_t)(e | 0x8000u);
                            gnb[ps] = (uint16_t)(e | 0x8000u);
                        }
                        change_count++;
                        const int len_j = (int)((e >> 12) & 7u);
                        errors += (double)(len_j < 2 ? 1 : len_j - 1) * f15;
                        if (++cc25 == 25) { cc25 = 0; stop_at = jj; break; }                  // ALIGNMENT_INTERVAL
                    }
                }
                wave_sync();
                if (stop_at >= 0) {
                    st = NEED_ALN;
                    if (stop_at + 1 < k) { resume_j = stop_at + 1; est_keep = est; n += (uint32_t)src; }   // the rest of this draw follows the alignment
                    else { resume_j = 0; n += (uint32_t)src + 1u; }
                    break;
                }
                resume_j = 0;
                // the rules at the top of the next iteration
                est_cur = 1.0 - div_inrange(errors, frag_len, rcp_len);
                if ((double)change_count > 0.9 * frag_len || est_cur <= target) { st = DONE; st_draws = (int)n + src + 1; break; }
            }
            if (st != RUN) break;
            resume_j = 0;
            if (dead) { st = DONE; st_draws = (int)loop_limit; break; }
            n += 64u;
        }
        if (!TAIL || st != NEED_ALN) break;
        // ---- TAIL: the re-estimation alignment (py/tksm_badread.py:405-432) on this wave, then the next visit
        {
            int p0 = 0, nrows = L;
            if (L > 1000) {
                p0 = (int)__umulhi(philox(P.seed, g, ST_ALNPOS, aln_no).x, (uint32_t)(L - 1000 + 1));
                nrows = 1000;
                wave_sync();
                for (int t = lane; t < nrows; t += 64) Fw[t] = gfrag[p0 + t];
            }
            wave_sync();
            const int m = join_window_planar(gnb, p0, nrows, Nw, ownw, wcap, lane);
            wave_sync();
            if (m > wcap) {
                if (early_mode) { to_exact_kernel(); return; }    // (no job slots beside the rounds)
                break;                                        // the regular route for this visit: st stays NEED_ALN
            }
            const AlnOut a = band_align<0, false>(Fw, nrows, Nw, ownw, m, lane, nullptr);
            st_aligns++;
            if (is_inf(a.dist)) {                                  // outside the band representation: the byte-exact kernel takes the read
                to_exact_kernel();
                return;
            }
            const int cols = (int)(a.stat & 0xffffu), mt = (int)(a.stat >> 16);
            const double ident = cols ? (double)mt / (double)cols : 0.0;
            if (L <= 1000) errors = (1.0 - ident) * frag_len;
            else {
                const double estimated = (1.0 - ident) * frag_len;
                const double weight = 1000.0 / frag_len;
                errors = estimated * weight + errors * (1.0 - weight);
            }
            aln_no++;
            st = RUN;
        }
    }
    // ---- as at the end of k_loop: a read at a re-estimation point gets an alignment job, one whose loop has ended waits (stage 3)
    if (lane == 0) {
        uint32_t job = 0;
        if (st == NEED_ALN) {
            job = FB.base_cur[rc] + atomicAdd(&FB.job_cnt[rc * 32u], 1u);
            st_aligns++;
            uint32_t p0 = 0, nrows = (uint32_t)L;
            if (L > 1000) {
                p0 = __umulhi(philox(P.seed, g, ST_ALNPOS, aln_no).x, (uint32_t)(L - 1000 + 1));
                nrows = 1000u;
            }
            *reinterpret_cast<uint4*>(FB.job_meta + 4ull * job) = make_uint4(r, p0, nrows, 0u);
        }
        sp->errors = errors; sp->est = est_keep; sp->change_count = change_count; sp->n_base = n; sp->aln_no = aln_no;
        sp->resume_src = -1; sp->resume_j = (int16_t)resume_j; sp->st_draws = st_draws; sp->st_aligns = st_aligns;
        sp->pending = st == NEED_ALN ? 1 : 0;
        sp->job = job;
        sp->stage = st == NEED_ALN ? 0 : 3;
        if (st == DONE) {
            atomicAdd(&FB.defer_cnt[rc * 32u], 1u);
            FB.defer_list[atomicAdd(&FB.counters[1], 1u)] = make_uint2(r, rc);
        }
    }
}

// ---- k_err: the last visit of a read, one wave per read: trims, q-score lookups from the ops of its q-score alignment (S5),
// output sequence, record sizes.  LDS per wave: frag[lcap] | nb[lcap] u16 | aux[ncap + 128] (the alignment's op bytes).
// STATE_IN_HBM (long reads): the fragment and its slot codes are read where they are in HBM -- their LDS footprint would leave
// a handful of waves per CU; only the aux area is in LDS.
template <bool STATE_IN_HBM>
__global__ __launch_bounds__(256, 8) void k_err(BatchView B, ErrModelView EM, QsModelView QM, SimParams P, SimBuffers O,
                                              FastBuffers FB, const uint32_t* __restrict__ order, uint32_t begin, uint32_t count,
                                              int lds_lcap, int lds_ncap, int from_jobs, uint32_t c0, uint32_t c1) {
    // one launch per length bucket: reads order[begin .. begin+count) share the LDS geometry (lds_lcap, lds_ncap),
    // so a batch with a few long molecules does not cost everyone its occupancy
    // wave-uniform values are pinned to scalar registers (readfirstlane): the vector file is the occupancy limit
#ifndef TKSM_ABLATE
    __builtin_amdgcn_s_setprio(2);                            // (a latency-bound kernel beside the alignment kernel's always-ready waves: see k_loopw)
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpw = blockDim.x >> 6;
    const uint32_t widx = blockIdx.x * (uint32_t)wpw + (uint32_t)wave;
    if (widx >= count) return;
    uint64_t r; uint32_t pos;
    if (!from_jobs) { pos = begin + widx; r = (uint64_t)__builtin_amdgcn_readfirstlane((int)order[pos]) & 0xffffffffull; }   // round 0: every read, in sorted order
    else if (from_jobs == 2) {
        // the reads whose q-score alignment was deferred (below), all at once after the last regular round
        const uint2 e = FB.defer_list[widx];
        r = (uint64_t)__builtin_amdgcn_readfirstlane((int)e.x) & 0xffffffffull;
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y) * FB.rs;
    } else {
        // later rounds: one wave per job of the previous round (= per read still running); a dispatched wave costs
        // ~1-2 ns even if it returns at once, and most reads are finished long before the last round
        const uint32_t target = FB.prefix[c0] + widx;
        uint32_t lo2 = c0, hi2 = c1 - 1;
        while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2 + 1) >> 1; if (FB.prefix[mid] <= target) lo2 = mid; else hi2 = mid - 1; }
        const uint32_t pjob = FB.base_prev[lo2] + (target - FB.prefix[lo2]);
        r = (uint64_t)__builtin_amdgcn_readfirstlane((int)FB.prev_meta[4ull * pjob]) & 0xffffffffull;
        pos = lo2 * FB.rs;                                        // any position inside the read's range
    }
    ReadState S = FB.state[r];
    if (S.stage == 2 || S.slow) return;
    if (S.stage == 0) return;                                  // still in its error loop (k_loop's business)
    uint16_t* gnb = nb_row(FB, r);
    const int per_wave = STATE_IN_HBM ? lds_ncap + 128 : lds_lcap * 3 + lds_ncap + 128;
    uint8_t* lds_wave = lds_raw + (size_t)wave * per_wave;
    uint8_t* frag = STATE_IN_HBM ? frag_row(FB, r) : lds_wave;
    uint16_t* nb = STATE_IN_HBM ? gnb : reinterpret_cast<uint16_t*>(lds_wave + lds_lcap);
    uint8_t* aux = STATE_IN_HBM ? lds_wave : lds_wave + 3 * (size_t)lds_lcap;
    const int k = EM.k;
    const uint64_t g = P.first_read + r * P.stride;
    const int raw_len = __builtin_amdgcn_readfirstlane(S.raw_len);
    const int L = raw_len + 2 * k;
    const uint64_t slot = O.slot_off[r];
    const int cap = __builtin_amdgcn_readfirstlane((int)((O.slot_off[r + 1] - slot) >> 1));
    uint8_t* out_seq = O.scratch + slot;
    uint8_t* out_qual = out_seq + cap;
    if (!STATE_IN_HBM) {
        const uint8_t* gfrag0 = frag_row(FB, r);
        for (int t = lane * 4; t < L; t += 256) *reinterpret_cast<uint32_t*>(frag + t) = *reinterpret_cast<const uint32_t*>(gfrag0 + t);
        for (int t = lane * 2; t < L; t += 128) *reinterpret_cast<uint32_t*>(nb + t) = *reinterpret_cast<const uint32_t*>(gnb + t);
    }
    wave_sync();
    uint32_t status = 0;
    const double frag_len = (double)L;
    const double errors = S.errors;
    const double target = S.target;
    const int change_count = __builtin_amdgcn_readfirstlane(S.change_count), st_draws = __builtin_amdgcn_readfirstlane(S.st_draws);
    int st_aligns = __builtin_amdgcn_readfirstlane(S.st_aligns);
    double identity = 1.0;

    // ---- :434-437 trims and the joined sequence (both stages)
    int start_trim, end_trim;
    {
        int v1 = lane < k ? slot_len(nb[lane]) : 0, v2 = lane < k ? slot_len(nb[L - k + lane]) : 0;
        (void)prefix_small(v1, start_trim);
        (void)prefix_small(v2, end_trim);
    }
    const int jcap = min(lds_ncap, cap);
    const int m = joined_len(nb, 0, L, lane);
    int lo = start_trim, hi = end_trim == 0 ? 0 : m - end_trim;
    lo = min(lo, m); hi = max(hi, lo);
    if (m > jcap) { status |= 1; lo = hi = 0; }
    const int out_len = hi - lo;
    const bool want_q = P.compute_q && m > 0 && !(status & 1);
    // (a read with q-scores comes here twice at most: never in stage 3 -- k_qjobs turned that into stage 1 -- so what follows is
    // either the last visit after the q-score alignment, or the only visit of a run without q-scores)
    if (want_q) {
        // ---- S5 q-scores from the alignment k_aln left in job_popd (py/tksm_badread.py:607-655)
        const uint32_t mt = S.res_mt, cols = S.res_cols, fail = S.res_fail;
        if (fail) { go_slow(FB, r, lane, 3); return; }
        identity = cols ? (double)mt / (double)cols : 0.0;
        const uint32_t rc1 = pos / FB.rs;
        const uint8_t* gp = FB.prev_popd + FB.geo_prev[rc1].popd_off + (size_t)(S.job - FB.base_prev[rc1]) * FB.geo_prev[rc1].ncap;
        wave_sync();
        uint8_t* popd = aux + 16;               // 16 bytes in front, 12 behind: the 9-byte windows are read as whole words
        for (int t = lane; t < m; t += 64) popd[t] = gp[t];
        wave_sync();
        const int margins = (QM.kmer_size - 1) / 2;
        const uint32_t hmask = (uint32_t)QM.n_slots - 1u;
        for (int i2 = lo + lane; i2 < hi; i2 += 64) {
#ifdef TKSM_ABLATE
            if (P.ablate == 40) { out_qual[i2 - lo] = 40; continue; }      // timing experiment: no lookups at all
#endif
            const int d = max(0, max(margins - i2, i2 + margins - (m - 1)));     // window shrunk symmetrically at the ends
            int row = -1; uint32_t roff = 0, rcnt = 0;
            auto probe = [&](uint64_t key) {
                uint32_t sl = (uint32_t)qs_hash(key) & hmask;
#ifdef TKSM_ABLATE
                if (P.ablate == 41) { row = (int)(sl & 1023u); roff = 0u; rcnt = 1u; return; }    // timing experiment: no table probe
#endif
                for (;;) {
                    const uint4 e = QM.ent[sl];                   // {key lo, key hi, row offset, row count}
                    const uint64_t kk = ((uint64_t)e.y << 32) | e.x;
                    if (kk == key) { row = (int)sl; roff = e.z; rcnt = e.w; break; }
                    if (kk == 0) break;
                    sl = (sl + 1) & hmask;
                }
            };
            if (margins <= 4) {
                // the windows [i2 - h, i2 + h], h = 0 .. 4, nest: their keys are built from the centre outwards in one
                // pass over the 9 bytes around i2 (three aligned LDS words), then tried from the widest one down
                // (a miss strips one symbol each side, get_qscore, py/tksm_badread.py:596-597)
                const int ob = 16 + i2 - 4, sh = ob & 3;
                const uint32_t* aw = reinterpret_cast<const uint32_t*>(aux) + (ob >> 2);
                const uint32_t w0 = aw[0], w1 = aw[1], w2 = aw[2];
                const uint32_t b0 = __builtin_amdgcn_alignbyte(w1, w0, sh), b1 = __builtin_amdgcn_alignbyte(w2, w1, sh), b2 = w2 >> (8 * sh);
                auto byte_at = [&](int j) -> uint32_t { return j < 4 ? (b0 >> (8 * j)) & 255u : j < 8 ? (b1 >> (8 * (j - 4))) & 255u : b2 & 255u; };
                uint64_t K[5]; int ln[5];
                K[0] = byte_at(4) & 3u; ln[0] = 1;
#pragma unroll
                for (int h = 1; h <= 4; h++) {
                    const uint32_t bl = byte_at(4 - h);
                    const int ddl = (int)(bl >> 2), ddr = (int)(byte_at(3 + h) >> 2);
                    ln[h] = ln[h - 1] + 2 + ddl + ddr;
                    const bool fits = ln[h] <= 29;                 // longer keys do not exist (and would not fit 58 bits)
                    const int sl_ = fits ? 2 * (1 + ddl) : 0, sr_ = fits ? 2 * (1 + ddl + ln[h - 1]) : 0, so_ = fits ? 2 * (ln[h] - 1) : 0;
                    const int dl2 = fits ? 2 * ddl : 0, dr2 = fits ? 2 * ddr : 0;
                    K[h] = (uint64_t)(bl & 3u) | (((1ull << dl2) - 1ull) << 2) | (K[h - 1] << sl_) | (((1ull << dr2) - 1ull) << sr_) |
                           ((uint64_t)(byte_at(4 + h) & 3u) << so_);
                }
                const int hmax = margins - d;
                bool pend = true;
#pragma unroll
                for (int h = 4; h >= 0; h--) {
                    const bool tryit = pend && h <= hmax;
                    if (__ballot(tryit) == 0ull) continue;
                    if (tryit && ln[h] <= 29) { probe(K[h] | ((uint64_t)ln[h] << 58)); pend = row < 0; }
#ifdef TKSM_ABLATE
                    if (P.ablate == 44) { if (row < 0) row = (int)(K[h] & 1023u); pend = false; }     // timing experiment: one level only
                    if (P.ablate == 45 && h == 3) { if (row < 0) row = (int)(K[h] & 1023u); pend = false; }   // ... two levels
#endif
                }
            } else {
                int s0 = i2 - margins + d, e0 = i2 + margins - d;
                for (;;) {
                    uint64_t key = 0; int len = 0; bool ok = true;
                    for (int x2 = s0; x2 <= e0; x2++) {
                        if (x2 > s0) {
                            const int dd = popd[x2 - 1] >> 2;
                            if (len + dd > 29) { ok = false; break; }
                            key |= ((1ull << (2 * dd)) - 1ull) << (2 * len); len += dd;
                        }
                        if (len >= 29) { ok = false; break; }
                        key |= (uint64_t)(popd[x2] & 3) << (2 * len); len++;
                    }
                    if (ok) probe(key | ((uint64_t)len << 58));
                    if (row >= 0 || s0 == e0) break;
                    s0++; e0--;
                }
            }
            uint8_t q = 0;
#ifdef TKSM_ABLATE
            if (P.ablate == 42) { out_qual[i2 - lo] = (uint8_t)(33 + (row & 31)); continue; }   // timing experiment: no draw, no row lookup
            if (P.ablate == 43 && row >= 0) {                                                     // ... a row lookup without the generator
                const uint32_t w = (uint32_t)i2 * 2654435761u;
                const uint32_t a2 = QM.guide[(size_t)row * 64 + (w >> 26)] & 0x7fu;
                out_qual[i2 - lo] = (uint8_t)(33 + QM.pairs[roff + min(a2, rcnt - 1u)].y); continue;
            }
#endif
            if (row >= 0) {
                const uint32_t w = philox(P.seed, g, ST_QUAL, (uint32_t)i2).x;
                // first a with w < cdf[a], else the last entry; the per-row guide table gives the first candidate
                // for the 64-quantile bucket of w, the scan from there ends within a step or two
                uint32_t a2 = QM.guide[(size_t)row * 64 + (w >> 26)];
                if (QM.guide_direct && (a2 & 0x80u)) q = (uint8_t)(a2 & 0x7fu);       // every draw of this bucket picks the same entry
                else {
                    uint2 pr = QM.pairs[roff + a2];
                    while (a2 + 1 < rcnt && !(w < pr.x)) { a2++; pr = QM.pairs[roff + a2]; }
                    q = (uint8_t)pr.y;
                }
            }
            out_qual[i2 - lo] = (uint8_t)(q + 33);
        }
    } else {
        identity = 1.0 - errors / frag_len;
    }
    join_out(frag, nb, L, out_seq, lo, hi, lane);
    if (P.quirk_perfect) identity = 1.0;
    finish_read(B, P, O, r, P.quirk_perfect ? out_len : raw_len, out_len, identity, status, st_draws, change_count, st_aligns, L, m,
                start_trim, end_trim, errors, target, lane);
    if (lane == 0) FB.state[r].stage = 2;
}

// ---- tail cut (diagnostic, TKSMSEQ_TAIL_CUT): reads still in the error loop are handed to the wave-wide kernel
__global__ void k_collect_unfinished(FastBuffers FB, uint64_t n_reads) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    ReadState S = FB.state[r];
    if (S.stage != 2 && !S.slow) {
        FB.state[r].slow = 1;
        const uint32_t idx = atomicAdd(&FB.counters[2], 1u);
        FB.slow_list[idx] = (uint32_t)r;
    }
}

// ---- Alignment of the re-estimation and q-score jobs: bit-parallel (Myers / Hyyro) banded global alignment, one LANE per job
// (k_alnf below).  Column j of the joined sequence owns fragment rows t_j .. t_j+63 (bit b = row t_j + b); t advances by the
// column's window shift.  Entering rows take vertical delta +1 (virtual cells below the previous window), the row above the
// window is unreachable (horizontal delta in = +1), and when the window does not move the top row can only be reached from the
// left (vertical delta forced to -1).  Per column the resolved predecessor of every cell is stored as 2 bits {w0, w1}: 0 up,
// 1 left, 2 diagonal mismatch, 3 diagonal match; the walk back from (n, m) yields matches / columns (identity) and, for q-score
// jobs, the per-read-position ops.  MODE: 0 = identity jobs of the error loop (preference up, left, diagonal; only matches /
// columns come back), 1 = q-score jobs (left, up, diagonal; per-position ops written).  All jobs of a launch have the same mode.
// Job geometry shared by the alignment kernels: `job0` is the first job id of the wave (all 64 lanes of a k_aln wave
// belong to one range), `job` the lane's own.
DEV uint32_t range_of_job(const FastBuffers& FB, uint32_t job0) {
    uint32_t rng = 0, hi2 = FB.n_ranges - 1;
    while (rng < hi2) { const uint32_t mid = (rng + hi2 + 1) >> 1; if (FB.base_cur[mid] <= job0) rng = mid; else hi2 = mid - 1; }
    return rng;
}
// ---- k_qjobs: after the last regular round every read waits in stage 3; with q-scores each of them gets one more alignment
// job, its whole new sequence against its whole fragment with the path kept (get_qscores, py/tksm_badread.py:611-613): job ids
// per range as in k_loop, aligned by k_alnf, looked up by the last visit of k_err.
__global__ __launch_bounds__(64) void k_qjobs(FastBuffers FB, int k, uint32_t count) {
    const int lane = threadIdx.x;
    const uint32_t widx = blockIdx.x * 64u + (uint32_t)lane;
    bool need = widx < count;
    uint32_t r = 0, rc = 0, job = 0;
    if (need) { const uint2 e = FB.defer_list[widx]; r = e.x; rc = e.y; }
    need = need && FB.state[r].stage == 3 && !FB.state[r].slow;
    const bool mine = need;
    unsigned long long todo = __ballot(need);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t lrc = (uint32_t)__shfl((int)rc, leader, 64);
        const unsigned long long grp = __ballot(need && rc == lrc);
        uint32_t base = 0;
        if (lane == leader) base = FB.base_cur[lrc] + atomicAdd(&FB.job_cnt[lrc * 32u], (uint32_t)__popcll(grp));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (need && rc == lrc) { job = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull)); need = false; }
        todo &= ~grp;
    }
    if (mine) {
        ReadState* sp = FB.state + r;
        const uint32_t L = (uint32_t)(sp->raw_len + 2 * k);
        *reinterpret_cast<uint4*>(FB.job_meta + 4ull * job) = make_uint4(r, 0u, L | 0x80000000u, 0u);
        sp->resume_src = -1; sp->pending = 1; sp->stage = 1; sp->job = job;
    }
}

DEV int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
// the lanes that want it append their job to a list (wave-aggregated)
DEV void list_append(uint32_t* list, uint32_t* counter, bool want, uint32_t job, int lane) {
    const unsigned long long wm = __ballot(want);
    if (!wm) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(counter, (uint32_t)__popcll(wm));
    base = __shfl(base, 0, 64);
    if (want) list[base + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull))] = job;
}


// ------------------------------------------------------------------------------------------------
// k_alnf: window packing FUSED into the alignment (round 3; rounds 1 - 2 packed the windows into block records with a kernel of
// their own, k_job, and aligned column by column: 16.25 B of HBM traffic per column against 10.25 B here, two kernels' worth of
// vector instructions against one -- DESIGN.md section 4.1).
// The lanes of a wave walk their windows SLOT BY SLOT in lockstep: in iteration s every lane decodes slot s of its own window
// straight from the read's slot codes (16 codes = 32 bytes per 16 iterations, unconditional, prefetched a pass ahead; a pristine
// slot carries its original base as its only symbol, so every slot decodes the same way) and pushes the slot's 0 .. 5 columns
// into per-lane bit queues (symbol low / high bit, and a unary slot stream: a 0 per slot boundary, a 1 per column); then it pops
// ONE column, if it has one, and runs the bit-parallel column step on it.  The window shift of a column follows from the slot it
// belongs to (top = max(1, slot + 1 - 31)), which the pop reads off the unary stream -- no shift queue.  Lanes whose windows hold
// more columns than slots fall behind by their backlog (a reflected random walk: a few columns); passes of 16 pop-only
// iterations drain them when a backlog passes BACKLOG_DRAIN columns and after the last slot.  The 32 .. 64 fragment rows below
// the window wait in a 64-bit reservoir per plane, refilled by 32 rows every 32 slots (the same rows for all lanes: unconditional
// loads).  The loop body is branch-free but for the pop's `if`, and every memory operation is unconditional.
// Predecessor codes are stored PER ITERATION (the iteration index is wave-uniform, the column index is not): one u32 per
// iteration = 14 band rows x 2 bits + the column's window shift (4 bits; 15 + codes 0 = no column in this iteration; a job with
// a shift the field cannot hold, >= 15, is redone with all rows whatever its entries say: the lane's largest shift is kept beside them),
// 16 iterations per 64 bytes, the pieces of a wave's 64 jobs interleaved ([wave][line][16-byte piece][lane]),
// so the walk back needs nothing but its code lines.  Stored rows: st .. st + 13 around the generative row, which sits at bit 31
// once the window moves and climbs from bit 0 while it is still clamped at row 1 (st = clamp(iteration - 7, 0, 24)).  14 rows
// (offsets -7 .. +6 from the generative row) miss 1.9 % of the bulk and 5.4 % of the polyA-tailed alignments (tools/band_rows.py
// on the CPU oracle; 16 rows: 0.9 % / 3.1 %, 8 rows: 36 %: co-optimal paths take the deletions of a homopolymer run at its end);
// those are redone with all 64 rows stored (two u64 per iteration, the shifts as a byte each behind the job's code lines) from the
// redo list; a path that leaves even those (or a shift above 31 rows) sends the read to the exact wave-wide kernel.
// The walk: every iteration's column is left exactly once, so all lanes walk back in LOCKSTEP over the iterations (an entry
// without a column is skipped by its lane, a lane joins at its own last column); inside a column the run of up moves is one
// find-first-bit on the column's "not up" bits.  Only the stored-row index `bs` of the current cell is tracked at iterations
// >= 32 (matches and diagonal moves are counted, columns = n + m - diagonals); below (window possibly clamped at row 1, st
// ramps) row and window position are tracked as well: there the path may reach row 0, after which only left moves remain.
constexpr uint32_t ENT_EMPTY = 0xF0000000u;
constexpr int BACKLOG_DRAIN = 20;
struct AlnJobF {
    bool act; int p0, n;
    const uint16_t* nb;                 // the read's slot codes (start of its row)
    const ulonglong2* fp; int wlast;    // the read's fragment planes, {lo, hi} per 64 positions; index of the last pair
    unsigned long long* popd8;          // q-score jobs: the job's op bytes
};
struct AlnResF { uint32_t mt, cols; int m; bool fail, needfull, overflow; uint32_t why; };

template <int MODE, int ROWS>
DEV AlnResF aln_fused(const AlnJobF& J, uint4* trl, int tg, int cl, uint32_t ls, int mcap) {      // (tg, cl, ls: the same for every lane of the wave)
    static_assert(ROWS == 14 || ROWS == 64, "14 stored rows, or all of them");
    constexpr int NC = ROWS == 14 ? 16 : 4;                       // iterations per 64-byte line of codes
    constexpr int ST = ROWS == 14 ? 24 : 0;                       // first stored band row once the window moves
    constexpr int RAMP0 = 31 - ST;
    const bool act = J.act;
    const int n = act ? J.n : 0;
    const int base = J.p0 & ~1, skip = J.p0 - base;               // slot codes are fetched from an even position
    // wave-uniform by construction, and TOLD so (scalar registers): the loop below and the walk then leave by uniform branches only.
    // Left as vector values (the compiler cannot know that a shuffle reduction or a per-lane load of the wave's range geometry is
    // uniform) they made every exit of the pass loop a divergent one: a page of EXEC bookkeeping per pass -- and, in builds forced
    // below the kernel's natural register count, spill code around that bookkeeping that gave WRONG results (tools/spill_probe.py)
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max(act ? n + skip : 0));
    tg = __builtin_amdgcn_readfirstlane(tg); cl = __builtin_amdgcn_readfirstlane(cl);
    // ---- fragment planes as a stream aligned to `base` (x = position - base): word j = x in [64 j, 64 j + 64)
    const int wb = base >> 6, bsh = base & 63;
    auto fpw = [&](int w) { return J.fp[min(max(w, 0), J.wlast)]; };
    auto aligned = [&](const ulonglong2& a, const ulonglong2& b) { ulonglong2 v; v.x = funnel128(a.x, b.x, bsh); v.y = funnel128(a.y, b.y, bsh); return v; };
    ulonglong2 Wn;                                                // word jc + 1 (jc = s0 >> 6), fetched when the slots reach it (once per 64
                                                                  // iterations: its latency is the other waves' to hide, its registers are not held)
    unsigned long long A, B;
    {
        const ulonglong2 r0 = fpw(wb), r1 = fpw(wb + 1), r2 = fpw(wb + 2);
        const ulonglong2 W0 = aligned(r0, r1);
        Wn = aligned(r1, r2);
        A = ~funnel128(W0.x, Wn.x, skip); B = ~funnel128(W0.y, Wn.y, skip);
    }
    int jc = 0;
    // the window's 64 rows (x in [skip, skip + 64)) and the reservoir of the rows below it (x from 64 + skip on: `ev` of them valid);
    // planes kept COMPLEMENTED: Eq = (~A ^ cl) & (~B ^ ch) saves two inversions per column.  At the end of the 32 slots from s0 the reservoir takes x in [64 + s0, 96 + s0) --
    // half a word of the aligned stream, the same for every lane (the first time a lane with skip = 1 drops the row its window
    // has already) -- so that the rows a column can need (x <= slot + 32) are always there.
    unsigned long long EA = 0ull, EB = 0ull;
    int app = 0;                                                  // rows appended to the reservoir so far; t - 1 of window + reservoir are used up: ev = app - (t - 1)
    unsigned long long Pv = ~0ull, Mv = 0ull;
    int t = 1, t32 = 1;
    // ---- queues
    uint32_t qlo = 0u, qhi = 0u;
    unsigned long long U = 0ull;
    int npend = 0, upos = 0, pcol = -1 - skip, col = 0, col32 = 0, shmax = 0;
    bool bad = false;
    // ---- slot codes: 16 slots (32 bytes) at a time, the next 16 in flight
    struct __attribute__((packed, aligned(4))) U4b { uint32_t x, y, z, w; };
    uint32_t cw[8], cwn[8];
    auto load_codes = [&](int s, uint32_t (&d)[8]) {
        const int src = (s <= n + skip) ? base + s : base;        // (lanes past their window re-read its start)
        const U4b* cp = reinterpret_cast<const U4b*>(J.nb + src);
        const U4b c0 = cp[0], c1 = cp[1];
        d[0] = c0.x; d[1] = c0.y; d[2] = c0.z; d[3] = c0.w; d[4] = c1.x; d[5] = c1.y; d[6] = c1.z; d[7] = c1.w;
    };
    load_codes(0, cw);
    constexpr int HS = 16;                                        // slots (iterations) per pass of the loop below
    constexpr int LPH = HS / NC;                                  // lines of codes per pass
    uint32_t tw[16], shw[4] = {0u, 0u, 0u, 0u};
    int ql = 0;                                                   // lines written (wave-uniform)
    bool ovf = false;
    int s0 = 0;
    for (;;) {
        const bool more = s0 < nmax;
        const bool drain = __ballot(npend > (more ? BACKLOG_DRAIN : 0)) != 0ull;
        if (!more && !drain) break;
        if (ql + LPH > (ROWS == 64 ? cl : tg - 1)) { ovf = true; break; }             // the job's lines are used up (drain passes of an insertion-heavy window)
        if (!drain) load_codes(s0 + HS, cwn);
        const int inc = drain ? 0 : 1;
        // (idle lanes write the job's spare line: with the lines of fewer than 64 jobs interleaved -- the full-width pool -- theirs
        // are another lane's)
        uint4* const dl = trl + (size_t)(act ? ql : tg - 1) * ls;
        const int it0 = __builtin_amdgcn_readfirstlane(ql * NC);  // the pass's first iteration (scalar)
        int mxn = 0, mxu = 0;                                     // the queues' fill before each push of the pass
#pragma unroll
        for (int q = 0; q < HS; q++) {
            // ---- push slot s0 + q
            const int p = s0 + q - skip;
            const bool on = !drain && act && (uint32_t)p < (uint32_t)n;
            const uint32_t code = (cw[q >> 1] >> (16 * (q & 1))) & 0xffffu;
            const int len = on ? (int)((code >> 12) & 7u) : 0;    // (a pristine slot: length 1, its symbol the original base -- k_init)
            const uint32_t m5 = (1u << len) - 1u;
            const uint32_t lo5 = code & m5, hi5 = (code >> 5) & m5;
            mxn = max(mxn, npend); mxu = max(mxu, upos);          // (capacity checked once per pass, below)
            qlo |= lo5 << npend; qhi |= hi5 << npend;
            U |= (unsigned long long)(m5 << 1) << upos;
            upos += len + inc; npend += len;
            // ---- pop: slot boundaries in front of the next column, then the column itself
            const uint32_t zz = U ? (uint32_t)__builtin_ctzll(U) : 64u;
            const int zc = (int)min(zz, (uint32_t)upos);
            pcol += zc;
            const bool have = npend > 0;
            const int used = zc + (have ? 1 : 0);
            U >>= used; upos -= used;
            uint32_t ent0 = ENT_EMPTY, ent1 = 0u, ent2 = 0u, ent3 = 0u, shb = 0xffu;     // (all rows: the entry's four words; shift byte 0xff = no column)
            if (have) {
                const int tn = max(1, pcol - 30);
                const uint32_t sh = (uint32_t)(tn - t);
                t = tn;
                shmax = max(shmax, (int)sh);                     // (14 stored rows: the entry's shift field holds 0 .. 14; a job with a larger one is redone with all rows)
                const uint32_t cl = (uint32_t)__builtin_amdgcn_sbfe((int)qlo, 0u, 1u), ch = (uint32_t)__builtin_amdgcn_sbfe((int)qhi, 0u, 1u);
                qlo >>= 1; qhi >>= 1; npend--; col++;
                const bool g = t > 1;
                const uint32_t f = (sh == 0u && g) ? 1u : 0u;     // window did not move: top row only from the left
                Pv = mk64(alignbit(~0u, hi32(Pv), sh), alignbit(hi32(Pv), lo32(Pv), sh) & ~f);
                Mv = mk64(hi32(Mv) >> sh, alignbit(hi32(Mv), lo32(Mv), sh) | f);
                // {EA : A} and {EB : B} move down by sh rows, IN PLACE and from the low word up (the compiler, left to itself, computes the
                // four new words of a plane into fresh registers and copies them back behind the branch: four v_mov_b64 per column)
#ifdef TKSM_NO_INLINE_ASM                                        // (diagnostic builds of tools/spill_probe.sh: the same arithmetic in plain C++)
                {
                    const unsigned long long nA = sh >= 64u ? 0ull : (A >> sh) | (sh ? EA << (64u - sh) : 0ull), nB = sh >= 64u ? 0ull : (B >> sh) | (sh ? EB << (64u - sh) : 0ull);
                    EA = sh >= 64u ? 0ull : EA >> sh; EB = sh >= 64u ? 0ull : EB >> sh; A = nA; B = nB;
                }
#else
                {
                    uint32_t a0 = lo32(A), a1 = hi32(A), a2 = lo32(EA), a3 = hi32(EA), b0 = lo32(B), b1 = hi32(B), b2 = lo32(EB), b3 = hi32(EB);
                    asm("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a0) : "v"(a1), "v"(sh));
                    asm("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a1) : "v"(a2), "v"(sh));
                    asm("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a2) : "v"(a3), "v"(sh));
                    asm("v_lshrrev_b32 %0, %1, %0" : "+v"(a3) : "v"(sh));
                    asm("v_alignbit_b32 %0, %1, %0, %2" : "+v"(b0) : "v"(b1), "v"(sh));
                    asm("v_alignbit_b32 %0, %1, %0, %2" : "+v"(b1) : "v"(b2), "v"(sh));
                    asm("v_alignbit_b32 %0, %1, %0, %2" : "+v"(b2) : "v"(b3), "v"(sh));
                    asm("v_lshrrev_b32 %0, %1, %0" : "+v"(b3) : "v"(sh));
                    A = mk64(a1, a0); EA = mk64(a3, a2); B = mk64(b1, b0); EB = mk64(b3, b2);
                }
#endif
                const unsigned long long clm = mk64(cl, cl), chm = mk64(ch, ch);
                const unsigned long long Eq = bool3<BOOL3(TA & (TB ^ TC))>(A ^ clm, B, chm);
                const unsigned long long Xv = Eq | Mv;
                const unsigned long long Xh = bool3<BOOL3((TA ^ TB) | TC)>((Eq & Pv) + Pv, Pv, Eq);
                const unsigned long long Ph = bool3<BOOL3(TA | ~(TB | TC))>(Mv, Xh, Pv);
                const unsigned long long Mh = Pv & Xh;
                const unsigned long long D0 = Xh | Mv;
                const unsigned long long Phs = mk64(alignbit(hi32(Ph), lo32(Ph), 31u), (lo32(Ph) << 1) | 1u);
                const unsigned long long Mhs = mk64(alignbit(hi32(Mh), lo32(Mh), 31u), lo32(Mh) << 1);
                Pv = bool3<BOOL3(TA | ~(TB | TC))>(Mhs, Xv, Phs);
                Mv = Phs & Xv;
                if constexpr (ROWS == 14) {
                    // stored rows st .. st + 13, st = clamp(iteration - RAMP0, 0, ST): a function of the iteration alone
                    // (on the scalar unit: the compiler fuses min(max()) into a vector v_med3_i32 and reads it back)
                    int st_s;
#ifdef TKSM_NO_INLINE_ASM
                    st_s = min(max(it0 + q, RAMP0), 31);
#else
                    asm("s_max_i32 %0, %1, %2\n\ts_min_i32 %0, %0, 31" : "=&s"(st_s) : "s"(it0 + q), "i"(RAMP0) : "scc");
#endif
                    const uint32_t st = (uint32_t)(st_s - RAMP0);
                    // the two code bits of the stored rows only: the 32-bit windows [st, st + 32) of Pv, Ph and D0 first, the logic on those.
                    // ("row above the window unreachable" -- bit 0 of Pv cleared when t > 1 -- cannot touch a stored row: bit 0 is stored
                    // only while st == 0, i.e. in the first RAMP0 + 1 iterations, whose columns belong to slots < 32: t is still 1 there)
                    const uint32_t PvW = alignbit(hi32(Pv), lo32(Pv), st), PhW = alignbit(hi32(Ph), lo32(Ph), st), D0W = alignbit(hi32(D0), lo32(D0), st);
                    const uint32_t c1 = bool3<BOOL3(~(TA | TB))>(PvW, PhW, 0u);
                    const uint32_t c0 = MODE ? bool3<BOOL3(TB | (~TA & TC))>(PvW, PhW, D0W) : bool3<BOOL3(~TA & (TB | TC))>(PvW, PhW, D0W);
                    const uint32_t c1s = c1 << 14;
                    const uint32_t c01 = ((c0 & 0x3fffu) | (c1s & ~0x3fffu)) & 0x0fffffffu;      // (a bit-field insert and an and-or)
                    ent0 = c01 | (min(sh, 15u) << 28);            // (sh >= 15: whatever the entry looks like, shmax sends the job to the full-width pass)
                } else {
                    const unsigned long long upv = Pv & ~(g ? 1ull : 0ull);
                    const unsigned long long w1 = bool3<BOOL3(~(TA | TB))>(upv, Ph, 0ull);
                    const unsigned long long w0 = MODE ? bool3<BOOL3(TB | (~TA & TC))>(upv, Ph, D0) : bool3<BOOL3(~TA & (TB | TC))>(upv, Ph, D0);
                    ent0 = lo32(w0); ent1 = hi32(w0); ent2 = lo32(w1); ent3 = hi32(w1);
                    shb = min(sh, 254u);
                }
            }
            if constexpr (ROWS == 14) {
                tw[q & 7] = ent0;
                if ((q & 7) == 7) {                               // half a line at a time
                    // (the four 16-byte pieces of a lane's line lie 64 pieces apart -- [wave][line][piece][lane] --: one store instruction
                    // of the wave writes 1 KB of consecutive bytes, whole cache lines, instead of a quarter of each of 64 lines)
                    uint4* d = dl + 128 * (q >> 3);
                    d[0] = make_uint4(tw[0], tw[1], tw[2], tw[3]); d[64] = make_uint4(tw[4], tw[5], tw[6], tw[7]);
                }
            } else {
                tw[4 * (q & 3)] = ent0; tw[4 * (q & 3) + 1] = ent1; tw[4 * (q & 3) + 2] = ent2; tw[4 * (q & 3) + 3] = ent3;
                shw[q >> 2] = (q & 3) == 0 ? shb : shw[q >> 2] | (shb << (8 * (q & 3)));
                if ((q & 3) == 3) {
                    uint4* d = dl + (act ? (size_t)(q >> 2) * ls : (size_t)0);
                    d[0] = make_uint4(tw[0], tw[1], tw[2], tw[3]); d[1] = make_uint4(tw[4], tw[5], tw[6], tw[7]);
                    d[2] = make_uint4(tw[8], tw[9], tw[10], tw[11]); d[3] = make_uint4(tw[12], tw[13], tw[14], tw[15]);
                }
            }
        }
        if constexpr (ROWS == 64) {
            // the pass's 16 shift bytes: one uint4 behind the job's code lines (line cl + pass / 4)
            const int pi = ql / LPH;
            uint4* d = act ? trl + (size_t)(cl + (pi >> 2)) * ls + (pi & 3) : trl + (size_t)(tg - 1) * ls;
            *d = make_uint4(shw[0], shw[1], shw[2], shw[3]);
        }
        bad |= mxn > 26 || mxu > 57;                              // a push would not have fitted the 32-bit symbol queues / the 64-bit slot stream
        ql += LPH;
        if (ql * NC == 32) { t32 = t; col32 = col; }              // window position / columns after iteration 31 (the walk's clamped start)
        if (!drain) {
            if (s0 & 16) {
                // the reservoir takes the 32 rows x in [64 + s0 - 16, 96 + s0 - 16): a half of the next aligned word
                const int ev = app - (t - 1);
                bad |= ev > 32 && s0 + HS - skip < n;              // (a lane past its last slot takes nothing from the reservoir any more)
                const bool upper = ((s0 - 16) & 32) != 0;
                const bool first = s0 == 16;
                const uint32_t drop = first ? (uint32_t)skip : 0u;
                const uint32_t na = ~(upper ? hi32(Wn.x) : lo32(Wn.x)) >> drop, nb2 = ~(upper ? hi32(Wn.y) : lo32(Wn.y)) >> drop;
                EA |= (unsigned long long)na << (ev & 63); EB |= (unsigned long long)nb2 << (ev & 63);
                app += 32 - (int)drop;
                if (upper) {                                      // the plane words follow the slot position
                    jc++;
                    Wn = aligned(fpw(wb + jc + 1), fpw(wb + jc + 2));
                }
            }
#pragma unroll
            for (int i2 = 0; i2 < 8; i2++) cw[i2] = cwn[i2];
            s0 += HS;
        }
    }
    const int m = col;
    ovf = ovf || m > mcap;                                        // (the job's op bytes / the read's output slot hold mcap columns)
    // ---- walk back from (n, m): all lanes in lockstep over the ITERATIONS; an entry without a column is skipped by its lane, a
    // lane joins at its own last column
    const uint32_t why = (bad ? 1u : 0u) | (shmax > 31 ? 8u : 0u) | (shmax > 14 ? 16u : 0u) | ((act && m > 0 && (n - t > 63 || n - t < 0)) ? 32u : 0u);
    bad = bad || (ROWS == 64 && shmax > 31);                      // (the 14-row pass only hands such a job on)
    bool fail = act && m > 0 && (n - t > 63 || n - t < 0 || bad), needfull = ROWS == 14 && act && shmax > 14;
    bool live = act && !fail && m > 0 && !ovf && !needfull;
    uint32_t mt = 0, dg = 0;
    int bs = n - t - ST;
    int i = n, tt = t;                                            // used below iteration 32 only
    int c = m - 1;                                                // the lane's current column (q-score jobs: where its op byte goes)
    unsigned long long acc = 0ull;
    auto walk_ent = [&](int it, auto ramp, auto lo, auto hi, uint32_t shc, bool is_col) {
        constexpr bool RAMP = decltype(ramp)::value;
        if (live && is_col) {
            if (RAMP) bs = i - tt - (min(max(it, RAMP0), 31) - RAMP0);
            uint32_t run, extra = 0u;
            if constexpr (ROWS == 64) { extra = (uint32_t)max(bs - 63, 0); bs -= (int)extra; }      // virtual cells below the band: up
            if constexpr (ROWS == 64) {
                const unsigned long long y = (lo | hi) << ((63 - bs) & 63);
                run = y ? (uint32_t)__builtin_clzll(y) : 64u;
            } else {
                const uint32_t y = (lo | hi) << ((31 - bs) & 31); // (the fields above bit bs are shifted out)
                run = y ? (uint32_t)__builtin_clz(y) : 32u;
            }
            bool ok = (uint32_t)bs < (uint32_t)ROWS;
            bool zero = false;
            if (RAMP) {
                run = min(run, (uint32_t)(bs + 1));               // (the shift filled the word with "up" bits below bit 0)
                if (i == 0) { run = 0u; extra = 0u; }
                zero = (uint32_t)i == run + extra || i == 0;      // the path reaches (or is in) row 0: this column and all before it are left moves
                ok |= i == 0;
            }
            const int bs2 = bs - (int)run;
            ok &= bs2 >= 0 || zero;
            const uint32_t lb = zero ? 1u : (uint32_t)(lo >> (bs2 & (ROWS == 64 ? 63 : 31))) & 1u;
            const uint32_t hb = zero ? 0u : (uint32_t)(hi >> (bs2 & (ROWS == 64 ? 63 : 31))) & 1u;     // lb | hb << 1: 1 left, 2 diagonal mismatch, 3 diagonal match
            needfull |= !ok;
            live = ok && !zero;
            mt += hb & lb;
            dg += hb;
            if (MODE) {
                const uint32_t op = hb ? (lb ^ 1u) : 2u;          // 0 match, 1 mismatch, 2 read-only base; the run = fragment-only bases in front
                acc |= (unsigned long long)(op | (min(run + extra, 63u) << 2)) << (8 * (c & 7));
                if (zero) {
                    // only left moves remain: the rest of this group and every group before it
                    acc |= 0x0202020202020202ull & ((1ull << (8 * (c & 7))) - 1ull);
                    J.popd8[c >> 3] = acc; acc = 0ull;
                    for (int b2 = (c >> 3) - 1; b2 >= 0; b2--) J.popd8[b2] = 0x0202020202020202ull;
                } else if ((c & 7) == 0) { J.popd8[c >> 3] = acc; acc = 0ull; }
                c--;
            }
            if (RAMP) { i -= (int)(run + extra + hb); tt -= (int)shc; }
            else bs = bs2 - (int)hb + (int)shc;
        }
    };
    uint4 la0, la1, la2, la3, lb0, lb1, lb2, lb3;
    auto load_line = [&](int q, uint4& l0, uint4& l1, uint4& l2, uint4& l3) {
        const uint4* p = trl + (size_t)max(q, 0) * ls;
        if constexpr (ROWS == 14) { l0 = p[0]; l1 = p[64]; l2 = p[128]; l3 = p[192]; }
        else { l0 = p[0]; l1 = p[1]; l2 = p[2]; l3 = p[3]; }
    };
    uint4 sw = make_uint4(0u, 0u, 0u, 0u), swn = sw;               // all rows: the shift bytes of the current pass, of the pass below it
    auto load_shifts = [&](int pi) { return trl[(size_t)(cl + (max(pi, 0) >> 2)) * ls + (max(pi, 0) & 3)]; };
    auto walk_line = [&](int q, const uint4& l0, const uint4& l1, const uint4& l2, const uint4& l3) {
        const uint32_t w[16] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w, l3.x, l3.y, l3.z, l3.w};
        const int cq = q * NC;
        uint32_t swq = 0u;
        if constexpr (ROWS == 64) {
            if ((q & 3) == 3) { sw = swn; swn = load_shifts((q >> 2) - 1); }
            swq = (q & 3) == 0 ? sw.x : (q & 3) == 1 ? sw.y : (q & 3) == 2 ? sw.z : sw.w;
        }
        auto ents_of_line = [&](auto ramp) {
#pragma unroll
            for (int e = NC - 1; e >= 0; e--) {
                if constexpr (ROWS == 14) {
                    const uint32_t v = w[e];
                    walk_ent(cq + e, ramp, v, v >> 14, v >> 28, v != ENT_EMPTY);
                } else {
                    const uint32_t shc = (swq >> (8 * e)) & 0xffu;
                    walk_ent(cq + e, ramp, mk64(w[4 * e + 1], w[4 * e]), mk64(w[4 * e + 3], w[4 * e + 2]), shc, shc != 0xffu);
                }
            }
        };
        if (cq >= 32) ents_of_line(std::false_type{});
        else {
            if (cq == 32 - NC && m > col32) { tt = t32; i = bs + t32 + ST; }      // leaving the iterations where only bs is tracked
            ents_of_line(std::true_type{});
        }
    };
    if (ql > 0) {
        const int topq = ql - 1;
        if constexpr (ROWS == 64) swn = load_shifts(topq >> 2);
        load_line(topq, la0, la1, la2, la3);
        for (int q = topq; q >= 0; q -= 2) {
            load_line(q - 1, lb0, lb1, lb2, lb3);
            walk_line(q, la0, la1, la2, la3);
            if (q > 0) {
                load_line(q - 2, la0, la1, la2, la3);
                walk_line(q - 1, lb0, lb1, lb2, lb3);
            }
        }
    }
    AlnResF R;
    R.mt = mt; R.cols = (uint32_t)(n + m) - dg; R.m = m;
    R.fail = fail || (ROWS == 64 && needfull); R.needfull = ROWS != 64 && needfull; R.overflow = act && ovf;
    R.why = why | (needfull ? 64u : 0u) | ((uint32_t)min(max(bs + 8, 0), 255) << 8);
    return R;
}

DEV void load_job_f(const FastBuffers& FB, uint32_t job, uint32_t rng, bool act, AlnJobF& J, uint32_t& r, int& mcap) {
    J.act = act;
    r = 0;
    J.p0 = 0; J.n = 0;
    if (act) {
        const uint4 meta = *reinterpret_cast<const uint4*>(FB.job_meta + 4ull * job);
        r = meta.x; J.p0 = (int)meta.y; J.n = (int)(meta.z & 0x7fffffffu);
    }
    const RangeGeo G = FB.geo_cur[rng];
    const uint32_t rel = job - FB.base_cur[rng];
    J.nb = nb_row(FB, r);
    J.fp = reinterpret_cast<const ulonglong2*>(planes_row(FB, r));
    J.wlast = planes_words(FB, r) - 1;
    J.popd8 = reinterpret_cast<unsigned long long*>(FB.job_popd + G.popd_off + (size_t)rel * G.ncap);
    mcap = (int)G.ncap;
}
// a window that outgrew its rows (insertion-heavy read): the host reruns the batch with larger slots
DEV void job_overflow(const FastBuffers& FB, const SimBuffers& O, uint32_t r) {
    O.status[r] |= 1u; O.out_len[r] = 0; O.rec_len[r] = 0; O.identity[r] = 0.0;
    FB.state[r].stage = 2;
}
DEV void store_result_f(const FastBuffers& FB, uint32_t r, const AlnResF& R) {
    ReadState* st = FB.state + r;
    st->res_mt = R.mt; st->res_cols = R.cols; st->res_fail = (R.fail || R.needfull) ? 1u : 0u;
}

// Alignment passes of a round (launch_alnf): pass 1 = every job with 14 stored rows (ROWS 14, LIST false); pass 2 = the jobs whose
// path left them (counters[10] of them in redo_list) with all 64 rows, lines in the full-width pool: a fixed grid whose waves loop over
// the list.  Rounds with few jobs are bound by the latency of one lane's pass: all their jobs go straight to the 64-row version
// (LIST false; counters[3] allocates pool lines per wave).
// (register budget: the 14-row pass needs ~131 vector registers -- 3 waves per SIMD; forced into 128 it spills, and a spill inside
// the pop's divergent region cost correct results once: never again below its natural size)
constexpr int ALNF_WAVES = 4;
constexpr int LONG_QJOB = 3000;         // slots; q-score jobs above it skip the 14-row pass
#ifdef TKSM_ALNF_NUM_VGPR                                        // (diagnostic builds of tools/spill_probe.sh: a register cap without a change of occupancy)
#define ALNF_VGPR_CAP __attribute__((amdgpu_num_vgpr(TKSM_ALNF_NUM_VGPR)))
#else
#define ALNF_VGPR_CAP
#endif
template <int MODE, int ROWS, bool LIST>
__global__ __launch_bounds__(64, ROWS == 64 ? 3 : ALNF_WAVES) ALNF_VGPR_CAP void k_alnf(SimParams P, FastBuffers FB, SimBuffers O, uint32_t n_jobs) {
#ifndef TKSM_ABLATE
    if (ROWS == 64) __builtin_amdgcn_s_setprio(2);            // (the full-width passes are a few latency-bound waves)
#endif
    const int lane = threadIdx.x;
    if (!LIST) {
        const uint32_t job0 = blockIdx.x * 64u, job = job0 + (uint32_t)lane;
        const uint32_t rng = range_of_job(FB, job0);
        const uint32_t rbase = FB.base_cur[rng];
        const uint32_t in_rng = FB.job_cnt[rng * 32u];
        if (in_rng <= job0 - rbase) return;                           // whole wave beyond the range's job count
        AlnJobF J;
        uint32_t r;
        bool act = job < n_jobs && job - rbase < in_rng;
        uint4* trl;
        int tg, mcap;
        bool norow = false;
        uint32_t ls = 256u;
        if (ROWS == 64) {
            // pool lines for the wave's jobs only (long molecules: a job's lines are megabytes)
            const unsigned long long wm = __ballot(act);
            const uint32_t na = (uint32_t)__popcll(wm);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&FB.counters[3], na);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base + na > FB.full_rows) { norow = act; act = false; base = 0; }    // no pool lines left: reported as failures (the wave-wide kernel takes the reads)
            tg = (int)FB.full_tg;
            ls = 4u * max(na, 1u);
            trl = reinterpret_cast<uint4*>(FB.trace_full) + ((size_t)base * FB.full_tg + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull)) % max(na, 1u)) * 4;
        } else {
            const RangeGeo G = FB.geo_cur[rng];
            const uint32_t rel = job - rbase;
            tg = (int)G.tstride;
            trl = reinterpret_cast<uint4*>(FB.trace) + (G.trace_off + (size_t)(rel >> 6) * G.tstride * 64) * 4 + (rel & 63u);
        }
        load_job_f(FB, job, rng, act || norow, J, r, mcap);
        // a q-score alignment covers the whole read: one over thousands of columns leaves the 14 stored rows somewhere with near
        // certainty (1.9 % per 1000 columns), and its pass is as long as its window (0.45 us per iteration: the whole launch waits
        // for the longest read) -- such a job goes straight to the list of the full-width pass
        const bool straight_to_list = MODE == 1 && ROWS == 14 && act && J.n > LONG_QJOB;
        J.act = act && !straight_to_list;
        AlnResF R = aln_fused<MODE, ROWS>(J, trl, tg, (int)FB.full_cl, ls, mcap);
        if (straight_to_list) { R.fail = false; R.needfull = true; R.overflow = false; }
        if (norow) { R.fail = true; R.needfull = false; R.overflow = false; }
        if (act && R.overflow) { job_overflow(FB, O, r); return; }
        if (ROWS != 64) list_append(FB.redo_list, FB.counters + 10, act && R.needfull, job, lane);
        if ((act || norow) && !R.needfull) store_result_f(FB, r, R);
        if (act && R.fail) { atomicAdd(&FB.counters[12], 1u); atomicOr(&FB.counters[13], R.why & 255u); FB.counters[14] = R.why; FB.counters[15] = (uint32_t)J.n | ((uint32_t)R.m << 16);
              for (int b = 0; b < 8; b++) if ((R.why >> b) & 1u) atomicAdd(&FB.counters[16 + b], 1u); if (MODE) atomicAdd(&FB.counters[24], 1u); if (LIST) atomicAdd(&FB.counters[25], 1u); }
    } else {
        static_assert(!LIST || ROWS == 64, "the list holds the jobs of the full-width pass");
        const uint32_t n_list = FB.counters[10];
        if (lane == 0 && blockIdx.x == 0 && n_list) { atomicAdd(&FB.counters[8], n_list); atomicAdd(&FB.counters[9], (n_list + 63u) / 64u); }   // diagnostics
        const uint32_t* list = FB.redo_list;
        const int tg = (int)FB.full_tg;
        uint4* trl = reinterpret_cast<uint4*>(FB.trace_full) + ((size_t)blockIdx.x * tg * 64 + (uint32_t)lane) * 4;
        for (uint32_t base = blockIdx.x * 64u; base < n_list; base += gridDim.x * 64u) {     // wave-uniform: every wave ends
            const uint32_t idx = base + (uint32_t)lane;
            const bool act = idx < n_list;
            const uint32_t job = list[min(idx, n_list - 1u)];                               // (idle lanes shadow the last job)
            const uint32_t rng = range_of_job(FB, job);
            AlnJobF J;
            uint32_t r;
            int mcap;
            load_job_f(FB, job, rng, act, J, r, mcap);
            const AlnResF R = aln_fused<MODE, ROWS>(J, trl, tg, (int)FB.full_cl, 256u, mcap);
            if (act && R.overflow) job_overflow(FB, O, r);
            else if (act) store_result_f(FB, r, R);
            if (act && R.fail) { atomicAdd(&FB.counters[12], 1u); atomicOr(&FB.counters[13], R.why & 255u); FB.counters[14] = R.why; FB.counters[15] = (uint32_t)J.n | ((uint32_t)R.m << 16);
              for (int b = 0; b < 8; b++) if ((R.why >> b) & 1u) atomicAdd(&FB.counters[16 + b], 1u); if (MODE) atomicAdd(&FB.counters[24], 1u); if (LIST) atomicAdd(&FB.counters[25], 1u); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// S6 emit: one wave formats one record at its scanned offset
// ------------------------------------------------------------------------------------------------
DEV int put_str(uint8_t* o, const char* s) { int n = 0; while (s[n]) { o[n] = (uint8_t)s[n]; n++; } return n; }
DEV int put_u(uint8_t* o, unsigned long long v) {
    int d = ndigits(v);
    for (int i = d - 1; i >= 0; i--) { o[i] = (uint8_t)('0' + v % 10); v /= 10; }
    return d;
}

// record header up to "molecule_id=" (py/sequence.py:252-288): "@<uuid> length=.. error_free_length=.. read_identity=..% molecule_id="
DEV int format_header(uint8_t* hdr, const SimParams& P, uint64_t g, uint32_t out_len, uint32_t raw_len, double identity) {
    int kx = 0;
    hdr[kx++] = P.fastq ? '@' : '>';
    const Ph4 id = philox(P.seed, g, ST_ID, 0);
    const uint32_t w[4] = {id.x, id.y, id.z, id.w};
    int nib = 0;
    for (int a = 0; a < 4; a++)
        for (int b = 7; b >= 0; b--) {
            if (nib == 8 || nib == 12 || nib == 16 || nib == 20) hdr[kx++] = '-';
            const uint32_t v = (w[a] >> (4 * b)) & 15u;
            hdr[kx++] = (uint8_t)(v < 10 ? '0' + v : 'a' + v - 10); nib++;
        }
    kx += put_str(hdr + kx, " length="); kx += put_u(hdr + kx, out_len);
    kx += put_str(hdr + kx, " error_free_length="); kx += put_u(hdr + kx, raw_len);
    kx += put_str(hdr + kx, " read_identity=");
    const long long h = pct_hundredths(identity);
    kx += put_u(hdr + kx, (unsigned long long)(h / 100));
    hdr[kx++] = '.'; hdr[kx++] = (uint8_t)('0' + (h % 100) / 10); hdr[kx++] = (uint8_t)('0' + h % 10);
    kx += put_str(hdr + kx, "% molecule_id=");
    return kx;
}

// a record's image in LDS (origin aligned like its place in the output, mod 16) -> the output: whole aligned 16-byte
// pieces, 1 KB per store instruction; the first and the last piece, shared with the neighbouring records, byte by byte
// (lanes 0 and 1)
DEV void flush_image(const uint8_t* img, int a, uint32_t rec_len, uint8_t* gout /* record start - a */, int lane) {
    const uint32_t end = (uint32_t)a + rec_len;
    const uint32_t c0 = a ? 1u : 0u, c1 = end >> 4;          // whole pieces: [c0, c1)
    for (uint32_t c = c0 + lane; c < c1; c += 64) *reinterpret_cast<uint4*>(gout + 16u * c) = *reinterpret_cast<const uint4*>(img + 16u * c);
    if (lane < 2) {
        const uint32_t lo = lane == 0 ? (uint32_t)a : max(16u * c1, (uint32_t)a), hi = lane == 0 ? (a ? min(16u, end) : 0u) : end;
        for (uint32_t t = lo; t < hi; t++) gout[t] = img[t];
    }
}

constexpr int EMIT_IMG = 6144;           // LDS image of a Badread record (bytes per wave); longer records go out bytewise

// Record formatting of the Badread path (py/sequence.py:242-258, :273-300): header, sequence and quality line of a read
// are assembled as an image in LDS (16 bytes per lane from the read's scratch slot) and leave in aligned 16-byte pieces,
// like k_perfect's records.
// (six waves per SIMD: what the workgroup's 25 KB of LDS allow.  Left to itself the compiler takes 110 registers -- four waves -- for a kernel
// that waits for its loads: 3.2 -> 1.5 ms per step, + 1.5 % on the bench)
__global__ __launch_bounds__(256, 6) void k_emit(BatchView B, SimParams P, SimBuffers O, const uint64_t* __restrict__ rec_off,
                                               uint8_t* __restrict__ records) {
    __shared__ __attribute__((aligned(16))) uint8_t img_all[WAVES_PER_WG][EMIT_IMG + 32];
    __shared__ uint8_t hdr_all[WAVES_PER_WG][160];
#ifndef TKSM_ABLATE
    __builtin_amdgcn_s_setprio(2);                            // (a latency-bound kernel beside the alignment kernel's always-ready waves: see k_loopw)
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t r = (uint64_t)blockIdx.x * WAVES_PER_WG + wave;
    if (r >= B.n_reads) return;
    uint8_t* hdr = hdr_all[wave];
    const uint64_t g = P.first_read + r * P.stride;
    const uint32_t out_len = O.out_len[r], raw_len = P.quirk_perfect ? O.out_len[r] : O.raw_len[r] - (O.tail_len ? O.tail_len[r] : 0u);
    int hl = 0;
    if (lane == 0) hl = format_header(hdr, P, g, out_len, raw_len, O.identity[r]);
    hl = __shfl(hl, 0, 64);
    wave_sync();
    const uint64_t off = rec_off[r], rec_len = rec_off[r + 1] - off;
    const uint32_t ido = B.ids[2 * r], idl = B.ids[2 * r + 1];
    const uint8_t* seq = O.scratch + O.slot_off[r];
    const uint64_t cap = (O.slot_off[r + 1] - O.slot_off[r]) >> 1;
    const bool real_q = P.mode == 1 && P.compute_q && !P.quirk_perfect;
    const uint8_t* qual = seq + cap;
    const int a = (int)(reinterpret_cast<uintptr_t>(records + off) & 15);
    if (rec_len + (uint64_t)a > (uint64_t)EMIT_IMG) {
        uint8_t* dst = records + off;
        for (int t = lane; t < hl; t += 64) dst[t] = hdr[t];
        dst += hl;
        for (uint32_t t = lane; t < idl; t += 64) dst[t] = B.idpool[ido + t];
        dst += idl;
        if (lane == 0) dst[0] = '\n';
        dst += 1;
        for (uint32_t t = lane; t < out_len; t += 64) dst[t] = seq[t];
        dst += out_len;
        if (lane == 0) dst[0] = '\n';
        dst += 1;
        if (P.fastq) {
            if (lane == 0) { dst[0] = '+'; dst[1] = '\n'; }
            dst += 2;
            for (uint32_t t = lane; t < out_len; t += 64) dst[t] = real_q ? qual[t] : (uint8_t)'K';
            dst += out_len;
            if (lane == 0) dst[0] = '\n';
        }
        return;
    }
    uint8_t* img = img_all[wave];
    for (int t = lane; t < hl; t += 64) img[a + t] = hdr[t];
    for (uint32_t t = lane; t < idl; t += 64) img[a + hl + t] = B.idpool[ido + t];
    if (lane == 0) img[a + hl + idl] = '\n';
    // sequence and qualities: whole 16-byte pieces of the read's (16-byte aligned, padded) scratch slot; a last piece may
    // run over its line's end, so what follows a line is written after it
    uint8_t* simg = img + a + hl + idl + 1;
    for (uint32_t t0 = 16u * lane; t0 < out_len; t0 += 1024u) {
        const uint4 v = *reinterpret_cast<const uint4*>(seq + t0);
        __builtin_memcpy(simg + t0, &v, 16);
    }
    wave_sync();
    if (lane == 0) { simg[out_len] = '\n'; if (P.fastq) { simg[out_len + 1] = '+'; simg[out_len + 2] = '\n'; } }
    if (P.fastq) {
        uint8_t* qimg = simg + out_len + 3;
        for (uint32_t t0 = 16u * lane; t0 < out_len; t0 += 1024u) {
            uint4 v = make_uint4(0x4b4b4b4bu, 0x4b4b4b4bu, 0x4b4b4b4bu, 0x4b4b4b4bu);
            if (real_q) v = *reinterpret_cast<const uint4*>(qual + t0);
            __builtin_memcpy(qimg + t0, &v, 16);
        }
        wave_sync();
        if (lane == 0) qimg[out_len] = '\n';
    }
    wave_sync();
    flush_image(img, a, (uint32_t)rec_len, records + off - a, lane);
}

// ------------------------------------------------------------------------------------------------
// --perfect (py/sequence.py:261-270, :303-313): the error-free sequence goes from the packed reference straight into
// its record -- no per-read working set, no intermediate copy.  k_perfect_lengths gives every record its length
// (then a scan), k_perfect writes header, bases and the constant quality line.
// ------------------------------------------------------------------------------------------------
__global__ void k_perfect_lengths(BatchView B, RefView R, SimParams P, SimBuffers O) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n_reads) return;
    const uint32_t L = O.raw_len[r], idl = B.ids[2 * r + 1];
    // a substitution outside its slice is an error of the input (the reference raises IndexError)
    uint32_t status = 0;
    const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    for (uint32_t ii = 0; ii < ic; ii++) {
        const Ivl iv = load_interval(B, R, ib + ii);
        for (uint32_t mi = iv.mod_begin; mi < iv.mod_end; mi++)
            if (B.mods[2ull * mi] >= iv.len) status |= 2;
    }
    if (status) O.status[r] |= status;
    uint64_t rec = 1 + 36 + 8 + ndigits(L) + 19 + ndigits(L) + 15 + 3 + 3 + 14 + idl + 1;   // identity "100.00"
    rec += (uint64_t)L + 1;
    if (P.fastq) rec += 2 + (uint64_t)L + 1;
    O.out_len[r] = L; O.identity[r] = 1.0; O.rec_len[r] = rec;
}

// One record, byte by byte (any length; also what the image path below falls back to when a record does not fit).
DEV void perfect_record_bytewise(const BatchView& B, const RefView& R, const SimParams& P, uint64_t r, uint32_t L, uint8_t* hdr,
                                 uint8_t* dst, int lane) {
    int hl = 0;
    wave_sync();
    if (lane == 0) hl = format_header(hdr, P, P.first_read + r * P.stride, L, L, 1.0);
    hl = __shfl(hl, 0, 64);
    wave_sync();
    for (int t = lane; t < hl; t += 64) dst[t] = hdr[t];
    dst += hl;
    const uint32_t ido = B.ids[2 * r], idl = B.ids[2 * r + 1];
    for (uint32_t t = lane; t < idl; t += 64) dst[t] = B.idpool[ido + t];
    dst += idl;
    if (lane == 0) dst[0] = '\n';
    dst += 1;
    // splice: slices of the contigs / literals, substitutions applied before the strand flip, later ones win
    const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
    for (uint32_t ii = 0; ii < ic; ii++) {
        const Ivl iv = load_interval(B, R, ib + ii);
        const uint32_t len = iv.len;
        for (uint32_t t = lane; t < len; t += 64) {
            const uint32_t src = iv.minus ? len - 1 - t : t;             // position in the slice
            uint8_t b = iv.literal ? upper(B.litpool[iv.gbase + iv.s + src]) : ref_base(R, iv.gbase + iv.s + src);
            for (uint32_t mi = iv.mod_begin; mi < iv.mod_end; mi++)
                if (B.mods[2ull * mi] == src) b = (uint8_t)B.mods[2ull * mi + 1];
            dst[t] = iv.minus ? comp(b) : b;
        }
        dst += len;
    }
    if (lane == 0) dst[0] = '\n';
    dst += 1;
    if (P.fastq) {
        if (lane == 0) { dst[0] = '+'; dst[1] = '\n'; }
        dst += 2;
        for (uint32_t t = lane; t < L; t += 64) dst[t] = (uint8_t)'K';
        dst += L;
        if (lane == 0) dst[0] = '\n';
    }
}

__constant__ char PERFECT_TEXT[] = " length= error_free_length= read_identity=100.00% molecule_id=";
constexpr int PT_LEN = 8, PT_EFL = 19, PT_REST = 35;
constexpr int PERFECT_IMG_MAX = 8192;    // largest LDS image of a record (bytes) in the first launch
constexpr int PERFECT_IMG_LONG = 36864;  // ... and in the second one, for the long records of a batch; longer ones are written bytewise
constexpr int PERFECT_IVLS = 32;         // intervals of a read held in the wave's LDS table; reads with more are written bytewise
struct PIvl { unsigned long long g; uint32_t len, o, pc, flags, mod_begin, mod_end; };   // flags: bit 0 literal, bit 1 minus
int perfect_lds_bytes(int img_bytes) { return WAVES_PER_WG * (img_bytes + 32 + PERFECT_IVLS * (int)sizeof(PIvl) + 160); }

// --perfect records at memory speed.  A wave assembles its record as an image in LDS whose origin has the alignment of
// the record's place in the output (mod 16), then copies it out in whole aligned 16-byte pieces: 1 KB per store
// instruction, only the first and last piece of a record (shared with its neighbours) go out byte by byte.  The read's
// intervals are loaded by one lane each; then every lane takes pieces of 16 bases of any interval, so that all reference
// reads of a record are in flight together: two packed words, a funnel shift, for the minus strand a bit reversal and
// complement, then four codes -> four ASCII bytes by byte-parallel arithmetic.  Literal segments and reference blocks that
// hold other symbols than ACGT are copied bytewise; substitutions are written over the image afterwards, in order.
#ifndef PERFECT_WAVES
#define PERFECT_WAVES 5
#endif
__global__ __launch_bounds__(256, PERFECT_WAVES) void k_perfect(BatchView B, RefView R, SimParams P, SimBuffers O, const uint64_t* __restrict__ rec_off,
                                                     uint8_t* __restrict__ records, int img_bytes, int skip_below, int last) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform: the read index and its metadata stay scalar
    const int per_wave = img_bytes + 32 + PERFECT_IVLS * (int)sizeof(PIvl) + 160;
    uint8_t* img = lds_raw + (size_t)wave * per_wave;
    PIvl* tab = reinterpret_cast<PIvl*>(img + img_bytes + 32);
    uint8_t* hdr = img + img_bytes + 32 + PERFECT_IVLS * sizeof(PIvl);
    const uint64_t n_waves = (uint64_t)gridDim.x * WAVES_PER_WG;
    for (uint64_t r = (uint64_t)blockIdx.x * WAVES_PER_WG + wave; r < B.n_reads; r += n_waves) {
        const uint32_t L = O.raw_len[r];
        const uint64_t off = rec_off[r], rec_len = rec_off[r + 1] - off;
        const uint32_t ib = B.reads[2 * r], ic = B.reads[2 * r + 1];
        const uint32_t ido = B.ids[2 * r], idl = B.ids[2 * r + 1];
        const int a = (int)(reinterpret_cast<uintptr_t>(records + off) & 15);
        // a batch with long records is written by two launches: the first one with small images (many waves per CU)
        // leaves the records that do not fit to the second one with large images
        const bool few = ic <= (uint32_t)PERFECT_IVLS;
        if (few && rec_len + (uint64_t)a <= (uint64_t)skip_below) continue;
        if (!few || rec_len + (uint64_t)a > (uint64_t)img_bytes) {
            if (last) perfect_record_bytewise(B, R, P, r, L, hdr, records + off, lane);
            continue;
        }
        wave_sync();                                      // the previous record's image and table have been read
        // ---- the read's intervals, one per lane: offsets of their bases and of their 16-base pieces
        uint32_t total_pieces;
        {
            Ivl iv{}; uint32_t len_i = 0, np_i = 0;
            if ((uint32_t)lane < ic) { iv = load_interval(B, R, ib + lane); len_i = iv.len; np_i = (len_i + 15) >> 4; }
            const uint32_t o_incl = (uint32_t)scan_add_incl((int)len_i, lane), pc_incl = (uint32_t)scan_add_incl((int)np_i, lane);
            total_pieces = (uint32_t)__shfl((int)pc_incl, 63, 64);
            if ((uint32_t)lane < ic) {
                PIvl e;
                e.g = iv.gbase + iv.s; e.len = len_i; e.o = o_incl - len_i; e.pc = pc_incl - np_i;
                e.flags = (iv.literal ? 1u : 0u) | (iv.minus ? 2u : 0u); e.mod_begin = iv.mod_begin; e.mod_end = iv.mod_end;
                tab[lane] = e;
            }
        }
        // ---- header (format_header, one character per lane), molecule id, newline
        // the digits of L, first one in the low byte (a record that fits an image has at most 5)
        int nd = 0; unsigned long long dpack = 0;
        for (uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)L);; v /= 10) { dpack = (dpack << 8) | ('0' + v % 10); nd++; if (v < 10) break; }
        const int hl = 1 + 36 + PT_LEN + nd + PT_EFL + nd + PT_REST;
        {
            const Ph4 id = philox(P.seed, P.first_read + r * P.stride, ST_ID, 0);
            for (int t = lane; t < hl; t += 64) {
                uint32_t ch;
                if (t == 0) ch = P.fastq ? '@' : '>';
                else if (t <= 36) {
                    const int u = t - 1;
                    if (u == 8 || u == 13 || u == 18 || u == 23) ch = '-';
                    else {
                        const int nb = u - (u > 8) - (u > 13) - (u > 18) - (u > 23);
                        const uint32_t w = (nb >> 3) == 0 ? id.x : (nb >> 3) == 1 ? id.y : (nb >> 3) == 2 ? id.z : id.w;
                        const uint32_t v = (w >> (4 * (7 - (nb & 7)))) & 15u;
                        ch = v < 10 ? '0' + v : 'a' + v - 10;
                    }
                } else {
                    const int u = t - 37;
                    int dg = -1;                           // >= 0: this position is digit dg (from the left) of L
                    if (u < PT_LEN) ch = (uint8_t)PERFECT_TEXT[u];
                    else if (u < PT_LEN + nd) { dg = u - PT_LEN; ch = 0; }
                    else if (u < PT_LEN + nd + PT_EFL) ch = (uint8_t)PERFECT_TEXT[u - nd];
                    else if (u < PT_LEN + nd + PT_EFL + nd) { dg = u - (PT_LEN + nd + PT_EFL); ch = 0; }
                    else ch = (uint8_t)PERFECT_TEXT[u - 2 * nd];
                    if (dg >= 0) ch = (uint32_t)(dpack >> (8 * dg)) & 255u;
                }
                img[a + t] = (uint8_t)ch;
            }
            for (uint32_t t = lane; t < idl; t += 64) img[a + hl + t] = B.idpool[ido + t];
            if (lane == 0) img[a + hl + idl] = '\n';
        }
        wave_sync();
        // ---- bases: piece w of the read = piece w - pc of the last interval whose first piece is <= w
        uint8_t* bimg = img + a + hl + idl + 1;
        for (uint32_t w = lane; w < total_pieces; w += 64) {
            uint32_t i = 0;
            for (uint32_t q = 1; q < ic; q++) i += tab[q].pc <= w ? 1u : 0u;
            const PIvl e = tab[i];
            splice_piece(B, R, e.g, e.len, e.flags & 1u, e.flags & 2u, 16u * (w - e.pc), bimg + e.o + 16u * (w - e.pc));
        }
        wave_sync();
        // ---- substitutions: before the strand flip, later entries win (py/sequence.py:229-239); one lane per interval
        if ((uint32_t)lane < ic) {
            const PIvl e = tab[lane];
            for (uint32_t mi = e.mod_begin; mi < e.mod_end; mi++) {
                const uint32_t mp = B.mods[2ull * mi], mc = B.mods[2ull * mi + 1];
                if (mp < e.len) bimg[e.o + ((e.flags & 2u) ? e.len - 1 - mp : mp)] = (e.flags & 2u) ? comp((uint8_t)mc) : (uint8_t)mc;
            }
        }
        // ---- separator and quality line
        if (lane == 0) {
            bimg[L] = '\n';
            if (P.fastq) { bimg[L + 1] = '+'; bimg[L + 2] = '\n'; bimg[L + 3 + L] = '\n'; }
        }
        if (P.fastq) {
            uint8_t* kq = bimg + L + 3;
            // whole words of 'K' between the unaligned ends
            const uint32_t head = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(kq) & 15)) & 15);
            for (uint32_t t = lane; t < min(head, L); t += 64) kq[t] = 'K';
            if (L > head) {
                const uint32_t body = (L - head) & ~15u;
                for (uint32_t t = 16u * lane; t < body; t += 1024u)
                    *reinterpret_cast<uint4*>(kq + head + t) = make_uint4(0x4b4b4b4bu, 0x4b4b4b4bu, 0x4b4b4b4bu, 0x4b4b4b4bu);
                for (uint32_t t = head + body + lane; t < L; t += 64) kq[t] = 'K';
            }
        }
        wave_sync();
        // ---- out: image bytes [a, a + rec_len) -> records[off ..)
        flush_image(img, a, (uint32_t)rec_len, records + off - a, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// S7: interleave P per-rank record streams into global read order
// ------------------------------------------------------------------------------------------------
struct PtrPack { const void* p[16]; uint64_t n[16]; };

__global__ void k_interleave_lens(int P_, PtrPack offs, uint64_t n_total, uint64_t* __restrict__ lens) {
    uint64_t gi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n_total) return;
    const int p = (int)(gi % P_); const uint64_t i = gi / P_;
    const uint64_t* o = (const uint64_t*)offs.p[p];
    lens[gi] = o[i + 1] - o[i];
}

__global__ __launch_bounds__(256) void k_interleave_copy(int P_, PtrPack streams, PtrPack offs, uint64_t n_total,
                                                          const uint64_t* __restrict__ dst_off, uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gi = (uint64_t)blockIdx.x * WAVES_PER_WG + wave;
    if (gi >= n_total) return;
    const int p = (int)(gi % P_); const uint64_t i = gi / P_;
    const uint64_t* o = (const uint64_t*)offs.p[p];
    const uint8_t* s = (const uint8_t*)streams.p[p] + o[i];
    const uint64_t len = o[i + 1] - o[i];
    uint8_t* d = dst + dst_off[gi];
    // 16 bytes per lane and step (neither side is aligned: records have any length), the last bytes one at a time
    struct __attribute__((packed, aligned(1))) U16 { uint32_t x, y, z, w; };
    const uint64_t body = len & ~15ull;
    for (uint64_t t = 16ull * lane; t < body; t += 1024) *reinterpret_cast<U16*>(d + t) = *reinterpret_cast<const U16*>(s + t);
    if (body + lane < len) d[body + lane] = s[body + lane];
}

// ------------------------------------------------------------------------------------------------
// exclusive scan (u64), three-phase: block sums -> recursive scan -> apply.  2048 values per block.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_T = 256, SCAN_V = 8, SCAN_B = SCAN_T * SCAN_V;

__global__ __launch_bounds__(SCAN_T) void k_scan_sums(const uint64_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ sums) {
    __shared__ uint64_t sh[SCAN_T];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_B + (uint64_t)threadIdx.x * SCAN_V;
    uint64_t a = 0;
    for (int i = 0; i < SCAN_V; i++) if (base + i < n) a += in[base + i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = SCAN_T / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(SCAN_T) void k_scan_apply(const uint64_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ block_off,
                                                        uint64_t* __restrict__ out) {
    __shared__ uint64_t sh[SCAN_T];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_B + (uint64_t)threadIdx.x * SCAN_V;
    uint64_t v[SCAN_V], a = 0;
    for (int i = 0; i < SCAN_V; i++) { v[i] = base + i < n ? in[base + i] : 0; a += v[i]; }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        uint64_t y = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += y;
        __syncthreads();
    }
    uint64_t run = (block_off ? block_off[blockIdx.x] : 0) + sh[threadIdx.x] - a;
    for (int i = 0; i < SCAN_V; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (base <= n - 1 && n - 1 < base + SCAN_V) out[n] = run;   // total after the last element
}

__global__ void k_set_u64(uint64_t* p, uint64_t v) { *p = v; }

__global__ void k_sum_u32(const uint32_t* __restrict__ in, uint64_t n, unsigned long long* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long a = 0;
    for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) a += in[i];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down((long long)a, o, 64);
    if ((threadIdx.x & 63) == 0 && a) atomicAdd(out, a);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t n, unsigned threads, unsigned cap = 256 * 16) {
    uint64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

hipError_t launch_pack(const uint8_t* ascii, uint64_t n, uint64_t gstart, uint32_t* packed, uint32_t* blockflag, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_pack, dim3(grid_for((n + 15) >> 4, 256)), dim3(256), 0, s, ascii, n, gstart, packed, blockflag);
    return hipGetLastError();
}
hipError_t launch_fill_pool(const uint8_t* ascii, uint64_t n, uint64_t gstart, const uint32_t* blocktab, uint8_t* pool, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_fill_pool, dim3(grid_for(n, 256)), dim3(256), 0, s, ascii, n, gstart, blocktab, pool);
    return hipGetLastError();
}
hipError_t launch_read_lengths(const BatchView& b, const RefView& r, int k, int cap_num, int cap_den, int cap_add,
                               const uint32_t* tail_len, uint32_t* raw_len, uint64_t* slot_cap, uint32_t* status, hipStream_t s) {
    if (!b.n_reads) return hipSuccess;
    hipLaunchKernelGGL(k_read_lengths, dim3((unsigned)((b.n_reads + 255) / 256)), dim3(256), 0, s, b, r, k, cap_num, cap_den,
                       cap_add, tail_len, raw_len, slot_cap, status);
    return hipGetLastError();
}
hipError_t launch_tail_lengths(const BatchView& b, const RefView& r, const TailView& t, uint64_t seed, uint64_t first_read,
                               uint64_t stride, uint32_t* tail_len, hipStream_t s) {
    if (!b.n_reads) return hipSuccess;
    hipLaunchKernelGGL(k_tail_lengths, dim3((unsigned)((b.n_reads + 255) / 256)), dim3(256), 0, s, b, r, t, seed, first_read, stride,
                       tail_len);
    return hipGetLastError();
}
int simulate_lds_bytes(int lcap, int ncap, int wpw) { return wpw * (lcap * 3 + ncap * 4); }
int simulate_max_wgs(int lds_bytes) {
    int per_cu = lds_bytes > 0 ? (160 * 1024) / lds_bytes : 8;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    return per_cu;
}
hipError_t launch_simulate(const BatchView& b, const RefView& r, const ErrModelView& em, const QsModelView& qm,
                           const IdentView& im, const SimParams& p, const SimBuffers& o, int n_wgs, int wpw, hipStream_t s) {
    if (!b.n_reads) return hipSuccess;
    const int lds = simulate_lds_bytes(p.s_lcap, p.s_ncap, wpw);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_simulate<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_simulate<false>, dim3(n_wgs), dim3(64 * wpw), lds, s, b, r, em, qm, im, p, o);
    return hipGetLastError();
}
// the same with the waves' working sets in o.big_scratch / o.big_trace (molecules beyond the LDS-resident limit); one wave per workgroup
size_t simulate_big_bytes(int lcap, int ncap) { return ((size_t)lcap * 3 + (size_t)ncap * 6 + 255) & ~(size_t)255; }
hipError_t launch_simulate_big(const BatchView& b, const RefView& r, const ErrModelView& em, const QsModelView& qm,
                               const IdentView& im, const SimParams& p, const SimBuffers& o, int n_waves, hipStream_t s) {
    if (!b.n_reads || !n_waves) return hipSuccess;
    hipLaunchKernelGGL(k_simulate<true>, dim3(n_waves), dim3(64), 0, s, b, r, em, qm, im, p, o);
    return hipGetLastError();
}
hipError_t launch_init(const BatchView& b, const RefView& r, const ErrModelView& em, const IdentView& im, const SimParams& p,
                       const SimBuffers& o, const FastBuffers& fb, int wpw, hipStream_t s) {
    if (!b.n_reads) return hipSuccess;
    const int lds = wpw * p.lcap;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_init), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_init, dim3((unsigned)((b.n_reads + wpw - 1) / wpw)), dim3(64 * wpw), lds, s, b, r, em, im, p, o, fb);
    return hipGetLastError();
}
int err_lds_bytes(int lcap, int ncap, int wpw, bool state_in_hbm) { return wpw * ((state_in_hbm ? 0 : lcap * 3) + ncap + 128); }
hipError_t launch_err(const BatchView& b, const ErrModelView& em, const QsModelView& qm, const SimParams& p, const SimBuffers& o,
                      const FastBuffers& fb, const uint32_t* order, uint32_t begin, uint32_t count, int lds_lcap, int lds_ncap,
                      int from_jobs, uint32_t c0, uint32_t c1, int wpw, bool state_in_hbm, hipStream_t s) {
    if (!count) return hipSuccess;
    const int lds = err_lds_bytes(lds_lcap, lds_ncap, wpw, state_in_hbm);
    auto kern = state_in_hbm ? k_err<true> : k_err<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((count + wpw - 1) / wpw), dim3(64 * wpw), lds, s, b, em, qm, p, o, fb, order, begin, count, lds_lcap,
                       lds_ncap, from_jobs, c0, c1);
    return hipGetLastError();
}
// fragment words per lane kept in LDS: at most LOOP_WL_MAX words (512 bases, 8 KB per wave); the words beyond come from HBM
// (L2).  Measured with three batches in flight (bench.py): 80 words / 6 draws at a time (20 KB, 228 VGPRs: 2 waves per SIMD)
// 8.11 M reads/s; 48 / 4: 8.29 M; 32 / 4 (164 VGPRs: 3 waves per SIMD): 8.42 M; no LDS at all: 8.25 M; 2 draws at a time: 6.8 M
// round 3 (16-draw passes): what the first phase pays for is its gather instructions, and the k-mers past the part in LDS are 22 % of
// the kernel (tools/ablate_loop.sh); 64 words (1024 bases, 22 KB per wave, 7 waves per CU): 31.2 -> 29.3 ms per 1.31 M reads, 96: 32.0
constexpr int LOOP_WL_MAX = 64;
int loop_lds_words(int lcap) {
    static const int wl_max = [] { const char* e = getenv("TKSMSEQ_LOOP_WL"); return e ? std::max(4, atoi(e) & ~3) : LOOP_WL_MAX; }();
    return std::min(wl_max, (((lcap + 15) / 16 + 1) + 3) & ~3);
}
hipError_t launch_loop(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, const uint32_t* order, uint32_t begin,
                       uint32_t count, int lcap, int from_jobs, uint32_t c0, uint32_t c1, hipStream_t s) {
    if (!count) return hipSuccess;
    const int Wl = loop_lds_words(lcap);
    hipLaunchKernelGGL(k_loop, dim3((count + 63) / 64), dim3(64), (size_t)Wl * 256 + (size_t)LOOP_N * 64 * 4, s, em, p, fb, order, begin, count, Wl, from_jobs, c0, c1);
    return hipGetLastError();
}
hipError_t launch_loopw(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, const uint32_t* order, uint32_t begin,
                        uint32_t count, int lcap, int from_jobs, uint32_t c0, uint32_t c1, hipStream_t s) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(k_loopw<false>, dim3(count), dim3(64), (size_t)lcap * 2 + 16, s, em, p, fb, order, begin, count, from_jobs, c0, c1, lcap, 0);
    return hipGetLastError();
}
size_t tail_lds_bytes(int lcap) { return tail_bitmap_bytes(lcap) + TAIL_FCAP + TAIL_WCAP * 3; }
// the reads of score bin >= min_bin, in sorted order, become early reads: flag (the regular kernels pass them by) and list entry
__global__ void k_mark_early(FastBuffers FB, const uint32_t* __restrict__ order, uint64_t n_reads, int k, uint32_t min_bin) {
    const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n_reads) return;
    const uint32_t r = order[pos];
    ReadState* sp = FB.state + r;
    if (sp->stage != 0 || sp->slow) return;
    if (early_bin(sp->raw_len + 2 * k, sp->target) < min_bin) return;
    sp->early = 1;
    FB.early_list[atomicAdd(&FB.counters[27], 1u)] = make_uint2(r, (uint32_t)(pos / FB.rs));
}
__global__ void k_merge_early_slow(FastBuffers FB) {
    const uint32_t n = FB.counters[26];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) FB.slow_list[FB.counters[2] + i] = FB.early_slow[i];
    __syncthreads();
    if (threadIdx.x == 0) { FB.counters[2] += n; FB.counters[4] += n; FB.counters[26] = 0u; }
}
hipError_t launch_mark_early(const FastBuffers& fb, const uint32_t* order, uint64_t n_reads, int k, uint32_t min_bin, hipStream_t s) {
    hipLaunchKernelGGL(k_mark_early, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, fb, order, n_reads, k, min_bin);
    return hipGetLastError();
}
hipError_t launch_tail_early(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, uint32_t max_count, int lcap, int wcap, hipStream_t s) {
    if (!max_count) return hipSuccess;
    hipLaunchKernelGGL(k_loopw<true>, dim3(max_count), dim3(64), tail_lds_bytes(lcap), s, em, p, fb, nullptr, 0u, max_count, 3, 0u, 0u, lcap, std::min(std::max(wcap, 1), TAIL_WCAP));
    return hipGetLastError();
}
hipError_t launch_merge_early_slow(const FastBuffers& fb, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_early_slow, dim3(1), dim3(256), 0, s, fb);
    return hipGetLastError();
}
hipError_t launch_tail(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, const uint32_t* order, uint32_t begin,
                       uint32_t count, int lcap, int from_jobs, uint32_t c0, uint32_t c1, int wcap, hipStream_t s) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(k_loopw<true>, dim3(count), dim3(64), tail_lds_bytes(lcap), s, em, p, fb, order, begin, count, from_jobs, c0, c1, lcap, std::min(std::max(wcap, 1), TAIL_WCAP));
    return hipGetLastError();
}
hipError_t launch_qjobs(const FastBuffers& fb, int k, uint32_t count, hipStream_t s) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(k_qjobs, dim3((count + 63) / 64), dim3(64), 0, s, fb, k, count);
    return hipGetLastError();
}
// start of a round: this round's per-range job counts and the alignment passes' two counters (rows of the full-width pool, jobs
// handed to the second pass), in one launch
__global__ void k_round_reset(uint32_t* __restrict__ job_cnt, uint32_t n_words, uint32_t* __restrict__ counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) job_cnt[i] = 0u;
    if (i == 0) { counters[3] = 0u; counters[10] = 0u; }
}
hipError_t launch_round_reset(const FastBuffers& fb, hipStream_t s) {
    const uint32_t n_words = fb.n_ranges * 32u;
    hipLaunchKernelGGL(k_round_reset, dim3((n_words + 255) / 256), dim3(256), 0, s, fb.job_cnt, n_words, fb.counters);
    return hipGetLastError();
}
hipError_t launch_collect_unfinished(const FastBuffers& fb, uint64_t n_reads, hipStream_t s) {
    if (!n_reads) return hipSuccess;
    hipLaunchKernelGGL(k_collect_unfinished, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, fb, n_reads);
    return hipGetLastError();
}
hipError_t launch_alnf(const SimParams& p, const FastBuffers& fb, const SimBuffers& o, uint32_t n_jobs, bool full_only, int mode, unsigned lds_pad, hipStream_t s) {
    if (!n_jobs) return hipSuccess;
    const uint32_t waves = (n_jobs + 63) / 64;
    if (full_only) {
        if (mode) hipLaunchKernelGGL((k_alnf<1, 64, false>), dim3(waves), dim3(64), 0, s, p, fb, o, n_jobs);
        else hipLaunchKernelGGL((k_alnf<0, 64, false>), dim3(waves), dim3(64), 0, s, p, fb, o, n_jobs);
        return hipGetLastError();
    }
    const uint32_t g2 = std::max<uint32_t>(1u, std::min<uint32_t>((waves + 7) / 8, std::max<uint32_t>(1u, fb.full_rows / 64)));
    if (mode) {
        hipLaunchKernelGGL((k_alnf<1, 14, false>), dim3(waves), dim3(64), lds_pad, s, p, fb, o, n_jobs);
        hipLaunchKernelGGL((k_alnf<1, 64, true>), dim3(g2), dim3(64), 0, s, p, fb, o, n_jobs);
    } else {
        hipLaunchKernelGGL((k_alnf<0, 14, false>), dim3(waves), dim3(64), lds_pad, s, p, fb, o, n_jobs);
        hipLaunchKernelGGL((k_alnf<0, 64, true>), dim3(g2), dim3(64), 0, s, p, fb, o, n_jobs);
    }
    return hipGetLastError();
}
hipError_t launch_emit(const BatchView& b, const SimParams& p, const SimBuffers& o, const uint64_t* rec_off, uint8_t* records, hipStream_t s) {
    if (!b.n_reads) return hipSuccess;
    hipLaunchKernelGGL(k_emit, dim3((unsigned)((b.n_reads + WAVES_PER_WG - 1) / WAVES_PER_WG)), dim3(256), 0, s, b, p, o, rec_off, records);
    return hipGetLastError();
}
hipError_t launch_perfect_lengths(const BatchView& b, const RefView& r, const SimParams& p, const SimBuffers& o, hipStream_t s) {
    if (!b.n_reads) return hipSuccess;
    hipLaunchKernelGGL(k_perfect_lengths, dim3((unsigned)((b.n_reads + 255) / 256)), dim3(256), 0, s, b, r, p, o);
    return hipGetLastError();
}
hipError_t launch_perfect(const BatchView& b, const RefView& r, const SimParams& p, const SimBuffers& o, const uint64_t* rec_off, uint8_t* records,
                          uint32_t max_raw, int n_cus, hipStream_t s) {
    if (!b.n_reads) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_perfect), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    // LDS image per wave: sized for the batch's longest record (header and id: a few hundred bytes), 8 KB at most in the
    // first launch; longer records get a second launch with images up to 36 KB (one workgroup per CU)
    const uint64_t need = ((p.fastq ? 2ull : 1ull) * max_raw + 512 + 255) & ~255ull;
    const int img[2] = {(int)std::min<uint64_t>(PERFECT_IMG_MAX, need), (int)std::min<uint64_t>(PERFECT_IMG_LONG, need)};
    const int n_launch = need > (uint64_t)PERFECT_IMG_MAX ? 2 : 1;
    for (int k = 0; k < n_launch; k++) {
        const int lds = perfect_lds_bytes(img[k]);
        // a persistent grid: exactly the workgroups that are resident at once (registers and LDS decide)
        int wgs_per_cu = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs_per_cu, reinterpret_cast<const void*>(k_perfect), 64 * WAVES_PER_WG, (size_t)lds);
        if (e != hipSuccess) return e;
        wgs_per_cu = std::max(1, wgs_per_cu);
        const uint64_t want = (b.n_reads + WAVES_PER_WG - 1) / WAVES_PER_WG;
        hipLaunchKernelGGL(k_perfect, dim3((unsigned)std::min<uint64_t>(want, (uint64_t)n_cus * wgs_per_cu)), dim3(64 * WAVES_PER_WG), lds, s, b, r, p, o,
                           rec_off, records, img[k], k ? img[0] : 0, k == n_launch - 1 ? 1 : 0);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_interleave_lens(int n_ranks, const uint64_t* const* offsets, const uint64_t* n_per_rank, uint64_t n_total,
                                  uint64_t* lens, hipStream_t s) {
    if (!n_total) return hipSuccess;
    PtrPack o{};
    for (int i = 0; i < n_ranks; i++) { o.p[i] = offsets[i]; o.n[i] = n_per_rank[i]; }
    hipLaunchKernelGGL(k_interleave_lens, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, s, n_ranks, o, n_total, lens);
    return hipGetLastError();
}
hipError_t launch_interleave_copy(int n_ranks, const uint8_t* const* streams, const uint64_t* const* offsets, uint64_t n_total,
                                  const uint64_t* dst_off, uint8_t* dst, hipStream_t s) {
    if (!n_total) return hipSuccess;
    PtrPack st{}, o{};
    for (int i = 0; i < n_ranks; i++) { st.p[i] = streams[i]; o.p[i] = offsets[i]; }
    hipLaunchKernelGGL(k_interleave_copy, dim3((unsigned)((n_total + WAVES_PER_WG - 1) / WAVES_PER_WG)), dim3(256), 0, s, n_ranks,
                       st, o, n_total, dst_off, dst);
    return hipGetLastError();
}

size_t scan_temp_bytes(uint64_t n) {
    size_t total = 0;
    while (n > 1) { uint64_t nb = (n + SCAN_B - 1) / SCAN_B; total += (nb + 1) * sizeof(uint64_t) * 2; n = nb; if (nb == 1) break; }
    return total + 64;
}
hipError_t launch_scan(const uint64_t* in, uint64_t* out, uint64_t n, void* temp, size_t temp_bytes, hipStream_t s) {
    if (n == 0) { hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, s, out, 0ull); return hipGetLastError(); }
    const uint64_t nb = (n + SCAN_B - 1) / SCAN_B;
    if (nb == 1) {
        hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(SCAN_T), 0, s, in, n, (const uint64_t*)nullptr, out);
        return hipGetLastError();
    }
    uint64_t* sums = (uint64_t*)temp;
    uint64_t* sums_scan = sums + (nb + 1);
    const size_t used = (nb + 1) * sizeof(uint64_t) * 2;
    if (used > temp_bytes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, n, sums);
    hipError_t e = launch_scan(sums, sums_scan, nb, (uint8_t*)temp + used, temp_bytes - used, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, n, (const uint64_t*)sums_scan, out);
    return hipGetLastError();
}
hipError_t launch_sum_u32(const uint32_t* in, uint64_t n, unsigned long long* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || !n) return e;
    hipLaunchKernelGGL(k_sum_u32, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, in, n, out);
    return hipGetLastError();
}

}  // namespace tk
