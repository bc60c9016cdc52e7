/*
 * oracle/tksm_oracle.c -- CPU restatement of TKSM's Seq hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (tksm_amd/csrc) must not link or call it.
 *
 * It restates, in plain scalar C, the algorithm of the reference's Python path
 * (all citations are into /root/reference):
 *   splice            py/sequence.py:303-313  (mdf_to_seq), :229-239 (apply_modifications),
 *                     :224-226 (reverse_complement)
 *   error insertion   py/tksm_badread.py:324-451 (SIMULATE_PY.sequence_fragment),
 *                     :119-144 (add_errors_to_kmer), :199-213 (add_one_random_change)
 *   identity          py/tksm_badread.py:245-257 (identity_from_edlib_cigar)
 *   q-scores          py/tksm_badread.py:607-655 (get_qscores), :584-598 (get_qscore)
 *   records           py/sequence.py:242-288 (badread/perfect/formatters)
 *   edlib             third-party python-edlib (env.yaml:8, unpinned, NOT in /root/reference):
 *                     global unit-cost alignment with path; restated here from its published
 *                     algorithm (NW, traceback preferring query-only 'I', then target-only 'D',
 *                     then diagonal).  PARITY UNPINNED for edlib tie-breaking.
 *
 * Randomness: the reference is unseeded.  The restatement replaces the Mersenne
 * Twister call sequence by a counter-based Philox4x32-10 keyed by (seed, read index, stream, n)
 * so that results do not depend on batch/GPU partitioning.  It is therefore distribution-
 * equivalent to the Python reference (pinned by tests/golden histograms) and bit-exact with the
 * HIP path (same counters, same arithmetic).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

/* ------------------------------------------------------------------ Philox4x32-10 */
typedef struct { uint32_t v[4]; } ph4;

static inline ph4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    ph4 o = {{c0, c1, c2, c3}};
    return o;
}

enum { ST_ID = 0, ST_PAD = 1, ST_IDENT = 2, ST_DRAW = 3, ST_ALNPOS = 4, ST_QUAL = 5, ST_TAIL = 6 };

static inline ph4 rng(uint64_t seed, uint64_t read, uint32_t stream, uint32_t n) {
    return philox4x32_10((uint32_t)read, (uint32_t)(read >> 32), stream, n,
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

void oracle_philox(uint64_t seed, uint64_t read, uint32_t stream, uint32_t n, uint32_t* out4) {
    ph4 p = rng(seed, read, stream, n);
    memcpy(out4, p.v, 16);
}

/* ------------------------------------------------------------------ exact NW with edlib-like path
 * q = query (rows), t = target (cols).  ops: '=' 'X' 'I' (query only) 'D' (target only), written
 * in forward order.  Traceback from (n,m): prefer up ('I'), then left ('D'), then diagonal.
 * (edlib: obtainAlignmentTraceback; call sites py/tksm_badread.py:170,:409,:422,:613.)
 * Returns edit distance, or -1 on allocation failure. */
int oracle_nw_path(const uint8_t* q, int n, const uint8_t* t, int m, uint8_t* ops, int* nops) {
    size_t W = (size_t)m + 1;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1) * W);
    if (!H) return -1;
    for (int j = 0; j <= m; j++) H[j] = j;
    for (int i = 1; i <= n; i++) {
        int32_t* r = H + (size_t)i * W;
        const int32_t* p = r - W;
        r[0] = i;
        uint8_t qc = q[i - 1];
        for (int j = 1; j <= m; j++) {
            int32_t d = p[j - 1] + (qc != t[j - 1]);
            int32_t u = p[j] + 1, l = r[j - 1] + 1;
            int32_t v = d < u ? d : u;
            r[j] = v < l ? v : l;
        }
    }
    int dist = H[(size_t)n * W + m];
    if (ops) {
        int i = n, j = m, k = 0;
        while (i > 0 || j > 0) {
            int32_t cur = H[(size_t)i * W + j];
            if (i > 0 && H[(size_t)(i - 1) * W + j] + 1 == cur) { ops[k++] = 'I'; i--; }
            else if (j > 0 && H[(size_t)i * W + j - 1] + 1 == cur) { ops[k++] = 'D'; j--; }
            else { ops[k++] = (H[(size_t)(i - 1) * W + j - 1] == cur) ? '=' : 'X'; i--; j--; }
        }
        for (int a = 0, b = k - 1; a < b; a++, b--) { uint8_t x = ops[a]; ops[a] = ops[b]; ops[b] = x; }
        *nops = k;
    }
    free(H);
    return dist;
}

/* ------------------------------------------------------------------ splice (py/sequence.py:303-313) */
static inline uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
static inline uint8_t comp(uint8_t c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 default: return c; }  /* 'N'->'N', anything else unchanged (py/sequence.py:225) */
}

/* One interval: contig bytes [0,clen), python slice [start:end] (clamped, start/end >= 0),
 * upper-cased, mods applied (pos relative to slice start, char verbatim), then reverse
 * complement if strand != '+'.  Returns bytes written, or -1 if a mod position is out of range
 * (the reference raises IndexError there). */
int64_t oracle_splice_interval(const uint8_t* contig, int64_t clen, int64_t start, int64_t end,
                               int plus, const int64_t* mod_pos, const uint8_t* mod_chr, int nmods,
                               uint8_t* out) {
    if (start > clen) start = clen;
    if (end > clen) end = clen;
    int64_t len = end - start;
    if (len < 0) len = 0;
    for (int64_t i = 0; i < len; i++) out[i] = up(contig[start + i]);
    for (int k = 0; k < nmods; k++) {
        if (mod_pos[k] < 0 || mod_pos[k] >= len) return -1;
        out[mod_pos[k]] = mod_chr[k];
    }
    if (!plus) {
        for (int64_t a = 0, b = len - 1; a < b; a++, b--) {
            uint8_t x = comp(out[a]); out[a] = comp(out[b]); out[b] = x;
        }
        if (len & 1) out[len / 2] = comp(out[len / 2]);
    }
    return len;
}

/* ------------------------------------------------------------------ models (tables built by oracle/pyoracle.py) */
typedef struct {
    int32_t type;      /* 0 = "random" (k = 1), 1 = k-mer model */
    int32_t k;
    int32_t max_alts;  /* row stride A */
    int32_t pad;
    const uint32_t* cdf;   /* [4^k][A] cumulative thresholds: alt a chosen iff w < cdf[a] (first such a) */
    const uint64_t* alts;  /* [4^k][A] packed aligned alternatives, see ALT_* below */
    const uint8_t* nalts;  /* [4^k] */
} err_model;

/* packed alternative: bits [0,24): 8 x 3-bit slot lengths; bits [24,62): bases, 2 bit each, in
 * concatenation order; bit 63: alternative equals the k-mer (no change). */
#define ALT_NOOP (1ull << 63)
static inline int alt_slot_len(uint64_t a, int j) { return (int)((a >> (3 * j)) & 7); }
static inline int alt_base(uint64_t a, int b) { return (int)((a >> (24 + 2 * b)) & 3); }

typedef struct {
    int32_t n_slots;       /* hash table size (power of two) */
    int32_t kmer_size;     /* max non-D key length (9 for shipped models) */
    const uint64_t* keys;  /* [n_slots] 0 = empty; else encoded cigar key */
    const uint32_t* row_off;   /* [n_slots] offset into pools */
    const uint32_t* row_cnt;   /* [n_slots] */
    const uint32_t* cdf_pool;  /* cumulative thresholds */
    const uint8_t* q_pool;     /* q values */
} qs_model;

/* key encoding: ops '='=0 'X'=1 'I'=2 'D'=3, first op in the lowest 2 bits of the digit string,
 * length in the top 6 bits:  key = (len << 58) | sum(op_i << 2i), len <= 29.  0 is never a
 * valid key because len >= 1. */
static inline uint64_t qs_hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
static int qs_find(const qs_model* m, uint64_t key) {
    uint32_t mask = (uint32_t)m->n_slots - 1, s = (uint32_t)qs_hash(key) & mask;
    while (m->keys[s]) {
        if (m->keys[s] == key) return (int)s;
        s = (s + 1) & mask;
    }
    return -1;
}
uint64_t oracle_qs_hash(uint64_t k) { return qs_hash(k); }

/* ------------------------------------------------------------------ fragment state */
typedef struct {
    int len;            /* padded fragment length */
    uint8_t* frag;      /* original padded fragment */
    uint8_t* slen;      /* new_fragment_bases[p] length (0..6) */
    uint8_t (*sb)[8];   /* new_fragment_bases[p] bytes */
    uint8_t* changed;
} fstate;

static const char BASES[4] = {'A', 'C', 'G', 'T'};
static inline int code_of(uint8_t c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

/* guided banded alignment of the fragment window F[0,n) (rows) against the joined new bases N[0,m) (columns),
 * column by column.  Column j (1-based) belongs to slot owner[j-1] (0-based row of the slot that emitted that
 * base; the generative path); it owns the 64 rows t_j .. t_j+63 (1-based), t_j = max(1, owner+1 - 31).
 * Cells above a column's window are unreachable (INF); cells below it are "virtual": value of the window's bottom
 * cell + distance, predecessor = up.  Row 0 (H[0][j] = j) is the real boundary while t_j == 1.
 * mode 0: query = fragment (loop re-estimation, py/tksm_badread.py:409,:422): predecessor preference
 *         up (fragment-only, 'I'), left (new-only, 'D'), diagonal.
 * mode 1: query = new sequence (q-scores, py/tksm_badread.py:613): preference left (read-only, 'I' in that
 *         cigar), up (fragment-only, 'D'), diagonal.
 * Returns the distance and matches / columns of the preferred optimal path; if trace != NULL stores per column
 * j (1..m) at trace[j*64 + b] bit0 = up ok, bit1 = left ok and the window top in ttop[j]. */
#define BW 32
#define INF 0x3fffffff
#ifdef BAND_STATS
/* diagnostic build only (tools/band_rows.py): every cell also carries the range of row offsets from the generative row
 * (i - g_j) along its preferred path; band_row_hist[lo + 64][hi + 64] counts the finished alignments of mode 0 */
typedef struct { int32_t h, m, c; int16_t lo, hi; } cell;
static int64_t band_row_hist[128][128];
void oracle_band_row_hist(int64_t* out, int reset) { memcpy(out, band_row_hist, sizeof(band_row_hist)); if (reset) memset(band_row_hist, 0, sizeof(band_row_hist)); }
#define BS_INIT(x) do { (x).lo = 0; (x).hi = 0; } while (0)
#define BS_TAKE(dst, src, off) do { int o_ = (off); (dst).lo = (int16_t)((src).lo < o_ ? (src).lo : o_); (dst).hi = (int16_t)((src).hi > o_ ? (src).hi : o_); } while (0)
#else
typedef struct { int32_t h, m, c; } cell;
#define BS_INIT(x) do { } while (0)
#define BS_TAKE(dst, src, off) do { } while (0)
#endif

static inline cell col_cell(const cell* col, int t, int i, int jcol) {
    /* value of row i (1-based, i >= 1) in a finished column whose window starts at row t; jcol = column index */
    cell r;
    if (i == 0) { BS_INIT(r); if (t == 1) { r.h = jcol; r.m = 0; r.c = jcol; } else r.h = INF; return r; }
    int b = i - t;
    if (b < 0) { r.h = INF; return r; }
    if (b <= 63) return col[b];
    r = col[63];
    if (r.h < INF) { r.h += b - 63; r.c += b - 63; }
    return r;
}

static int band_align(const uint8_t* F, int n, const uint8_t* N, int m, const uint32_t* owner /* m */,
                      int mode, int* out_match, int* out_cols, uint8_t* trace, int32_t* ttop) {
    cell prev[64], cur[64];
    int tp = 1;
    for (int b = 0; b < 64; b++) { prev[b].h = 1 + b; prev[b].m = 0; prev[b].c = 1 + b; BS_INIT(prev[b]); }
    for (int j = 1; j <= m; j++) {
        int g = (int)owner[j - 1] + 1;
        int t = g - (BW - 1) > 1 ? g - (BW - 1) : 1;
        uint8_t nc = N[j - 1];
        if (ttop) ttop[j] = t;
        for (int b = 0; b < 64; b++) {
            int i = t + b;
            if (i > n) { cur[b].h = INF; if (trace) trace[(size_t)j * 64 + b] = 0; continue; }
            cell up;
            if (b >= 1) up = cur[b - 1];
            else if (t == 1) { up.h = j; up.m = 0; up.c = j; BS_INIT(up); }
            else { up.h = INF; BS_INIT(up); }
            cell dg = col_cell(prev, tp, i - 1, j - 1);
            cell lf = col_cell(prev, tp, i, j - 1);
            int match = F[i - 1] == nc;
            int32_t vd = dg.h >= INF ? INF : dg.h + (match ? 0 : 1);
            int32_t vu = up.h >= INF ? INF : up.h + 1;
            int32_t vl = lf.h >= INF ? INF : lf.h + 1;
            int32_t h = vd < vu ? vd : vu;
            if (vl < h) h = vl;
            cur[b].h = h;
            if (h >= INF) { if (trace) trace[(size_t)j * 64 + b] = 0; continue; }
            int upok = (vu == h), leftok = (vl == h);
            if (trace) trace[(size_t)j * 64 + b] = (uint8_t)(upok | (leftok << 1));
            int take = mode == 0 ? (upok ? 0 : (leftok ? 1 : 2)) : (leftok ? 1 : (upok ? 0 : 2));
            if (take == 0) { cur[b].m = up.m; cur[b].c = up.c + 1; BS_TAKE(cur[b], up, i - g); }
            else if (take == 1) { cur[b].m = lf.m; cur[b].c = lf.c + 1; BS_TAKE(cur[b], lf, i - g); }
            else { cur[b].m = dg.m + match; cur[b].c = dg.c + 1; BS_TAKE(cur[b], dg, i - g); }
        }
        memcpy(prev, cur, sizeof(cur));
        tp = t;
    }
    cell f = col_cell(prev, tp, n, m);
    if (n - tp > 63 || f.h >= INF) return -1;
#ifdef BAND_STATS
    if (mode == 0) {
        int lo = f.lo < -64 ? -64 : f.lo, hi = f.hi > 63 ? 63 : f.hi;
        __atomic_fetch_add(&band_row_hist[lo + 64][hi + 64], 1, __ATOMIC_RELAXED);
    }
#endif
    *out_match = f.m; *out_cols = f.c;
    return f.h;
}

/* full-matrix version of the same recurrence and preference (no band) -- used by tests to show
 * that the band never changes a result on the test corpus. */
static int full_align(const uint8_t* F, int n, const uint8_t* N, int m, int mode,
                      int* out_match, int* out_cols, uint8_t* ops, int* nops) {
    size_t W = (size_t)m + 1;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1) * W);
    if (!H) return -1;
    for (int j = 0; j <= m; j++) H[j] = j;
    for (int i = 1; i <= n; i++) {
        int32_t* r = H + (size_t)i * W; const int32_t* p = r - W;
        r[0] = i;
        for (int j = 1; j <= m; j++) {
            int32_t d = p[j - 1] + (F[i - 1] != N[j - 1]);
            int32_t u = p[j] + 1, l = r[j - 1] + 1;
            int32_t v = d < u ? d : u;
            r[j] = v < l ? v : l;
        }
    }
    int i = n, j = m, mt = 0, cols = 0, k = 0;
    while (i > 0 || j > 0) {
        int32_t curv = H[(size_t)i * W + j];
        int upok = i > 0 && H[(size_t)(i - 1) * W + j] + 1 == curv;
        int leftok = j > 0 && H[(size_t)i * W + j - 1] + 1 == curv;
        int take;
        if (mode == 0) take = upok ? 0 : (leftok ? 1 : 2);
        else take = leftok ? 1 : (upok ? 0 : 2);
        cols++;
        if (take == 0) { if (ops) ops[k++] = (mode == 0) ? 'I' : 'D'; i--; }
        else if (take == 1) { if (ops) ops[k++] = (mode == 0) ? 'D' : 'I'; j--; }
        else {
            int eq = H[(size_t)(i - 1) * W + j - 1] == curv;
            mt += eq; if (ops) ops[k++] = eq ? '=' : 'X'; i--; j--;
        }
    }
    if (ops) {
        for (int a = 0, b = k - 1; a < b; a++, b--) { uint8_t x = ops[a]; ops[a] = ops[b]; ops[b] = x; }
        *nops = k;
    }
    int dist = H[(size_t)n * W + m];
    free(H);
    *out_match = mt; *out_cols = cols;
    return dist;
}

/* ------------------------------------------------------------------ TEST-ONLY variants of the q-score alignment
 * (tests/test_edlib_hole.py, tools/edlib_hole.py): how far do the q-scores and the printed identity move when co-optimal
 * paths are chosen differently?  The shipped specification keeps edlib's TRACEBACK order at every size; the real edlib
 * (python-edlib, not in the reference tree) switches to Hirschberg's divide and conquer once its alignment data reach
 * 1 MB -- (2 * 8 + 4) * ceil(query / 64) * target + 2 * 4 * target bytes -- i.e. for the q-score alignment
 * (py/tksm_badread.py:611-613: query = read, target = fragment) of every read above ~1.77 kb.  Restated from edlib's
 * published source (obtainAlignment / obtainAlignmentHirschberg): the target is cut in half, the forward scores of the
 * left half's last column and the reverse scores of the right half's first column are summed, the FIRST query row
 * 0 .. n-2 (0-based cell row; then row -1, then row n-1) whose sum equals the distance is where the path crosses, and
 * both parts recurse through the same size rule.
 *   variant 0  shipped: guided band, traceback order read-only ('I'), fragment-only ('D'), diagonal
 *   variant 1  the opposite indel preference (fragment-only first), unbanded
 *   variant 2  edlib as published: traceback below the 1 MB rule, Hirschberg above it, unbanded
 *   variant 3  the shipped order, unbanded (the control for 1 and 2)
 * Cigar letters as edlib.align(read, fragment) writes them: 'I' read-only, 'D' fragment-only. */
static int g_qalign_variant = 0;
void oracle_set_qscore_alignment_variant(int v) { g_qalign_variant = v; }

/* q = read (edlib's query, rows), t = fragment (edlib's target, columns); pref 0: up ('I') before left ('D'), 1: left first */
static void nw_trace_pref(const uint8_t* q, int n, const uint8_t* t, int m, int pref, uint8_t* ops, int* nops) {
    size_t W = (size_t)m + 1;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1) * W);
    for (int j = 0; j <= m; j++) H[j] = j;
    for (int i = 1; i <= n; i++) {
        int32_t* r = H + (size_t)i * W; const int32_t* p = r - W;
        r[0] = i;
        for (int j = 1; j <= m; j++) {
            int32_t d = p[j - 1] + (q[i - 1] != t[j - 1]), u = p[j] + 1, l = r[j - 1] + 1;
            int32_t v = d < u ? d : u;
            r[j] = v < l ? v : l;
        }
    }
    int i = n, j = m, k = 0;
    while (i > 0 || j > 0) {
        int32_t cur = H[(size_t)i * W + j];
        int upok = i > 0 && H[(size_t)(i - 1) * W + j] + 1 == cur;
        int leftok = j > 0 && H[(size_t)i * W + j - 1] + 1 == cur;
        int take = pref == 0 ? (upok ? 0 : (leftok ? 1 : 2)) : (leftok ? 1 : (upok ? 0 : 2));
        if (take == 0) { ops[k++] = 'I'; i--; }
        else if (take == 1) { ops[k++] = 'D'; j--; }
        else { ops[k++] = (H[(size_t)(i - 1) * W + j - 1] == cur) ? '=' : 'X'; i--; j--; }
    }
    for (int a = 0, b = k - 1; a < b; a++, b--) { uint8_t x = ops[a]; ops[a] = ops[b]; ops[b] = x; }
    *nops = k;
    free(H);
}

/* out[i] = edit distance of q[0, i) and t[0, m) (rev: of the last i query bytes and the last m target bytes), i = 0 .. n */
static void nw_last_column(const uint8_t* q, int n, const uint8_t* t, int m, int rev, int32_t* out) {
    for (int i = 0; i <= n; i++) out[i] = i;
    for (int j = 1; j <= m; j++) {
        uint8_t tc = rev ? t[m - j] : t[j - 1];
        int32_t diag = out[0];
        out[0] = j;
        for (int i = 1; i <= n; i++) {
            uint8_t qc = rev ? q[n - i] : q[i - 1];
            int32_t d = diag + (qc != tc), u = out[i - 1] + 1, l = out[i] + 1;
            diag = out[i];
            int32_t v = d < u ? d : u;
            out[i] = v < l ? v : l;
        }
    }
}

static void edlib_like_path(const uint8_t* q, int n, const uint8_t* t, int m, uint8_t* ops, int* nops, int* n_splits) {
    if (n == 0 || m == 0) {                                    /* edlib: one sequence empty */
        for (int a = 0; a < n + m; a++) ops[a] = n == 0 ? 'D' : 'I';
        *nops = n + m;
        return;
    }
    long long blocks = (n + 63) / 64;
    long long data = (2ll * 8 + 4) * blocks * m + 2ll * 4 * m;
    if (data < 1024 * 1024) { nw_trace_pref(q, n, t, m, 0, ops, nops); return; }
    (*n_splits)++;
    int lw = m / 2, rw = m - lw;
    int32_t* fwd = (int32_t*)malloc(sizeof(int32_t) * ((size_t)n + 1) * 2);
    int32_t* bwd = fwd + n + 1;
    nw_last_column(q, n, t, lw, 0, fwd);                      /* fwd[i]: q[0, i) vs t[0, lw) */
    nw_last_column(q, n, t + lw, rw, 1, bwd);                 /* bwd[x]: last x of q vs t[lw, m) */
    int32_t best = INF;
    for (int i = 0; i <= n; i++) { int32_t v = fwd[i] + bwd[n - i]; if (v < best) best = v; }
    int cut = -1;
    for (int i = 1; i <= n - 1 && cut < 0; i++) if (fwd[i] + bwd[n - i] == best) cut = i;    /* cell rows 0 .. n-2 */
    if (cut < 0 && fwd[0] + bwd[n] == best) cut = 0;          /* row -1 */
    if (cut < 0) cut = n;                                     /* row n-1 */
    free(fwd);
    int k1 = 0, k2 = 0;
    edlib_like_path(q, cut, t, lw, ops, &k1, n_splits);
    edlib_like_path(q + cut, n - cut, t + lw, rw, ops + k1, &k2, n_splits);
    *nops = k1 + k2;
}

/* join new bases of window [p0,p0+n) into buf; owner[j] = window-relative slot of joined base j; returns joined length */
static int join_window(const fstate* s, int p0, int n, uint8_t* buf, uint32_t* owner) {
    int m = 0;
    for (int r = 0; r < n; r++)
        for (int b = 0; b < s->slen[p0 + r]; b++) { owner[m] = (uint32_t)r; buf[m++] = s->sb[p0 + r][b]; }
    return m;
}

/* statistics the tests histogram; mirrors the probe counters of SURVEY.md section 6 */
typedef struct {
    int32_t n_draws, n_noop, n_kmers_applied, change_count, n_aligns, n_random_change;
    int32_t n_sub, n_ins_slots, n_del, ins_bases, frag_len, new_len, start_trim, end_trim;
    int32_t band_fail, pad0;
    double errors, target_identity;
} frag_stats;

/* identity sampler (py/tksm_badread.py:741-745): constant, or max * Q(u) with Q the Beta(a,b)
 * quantile function tabulated on 65537 points (linear interpolation). */
typedef struct {
    int32_t constant; int32_t pad;
    double value;        /* constant identity, or max_identity */
    const double* qtab;  /* 65537 quantiles of Beta(a,b) */
} ident_model;

double oracle_target_identity(const ident_model* im, uint64_t seed, uint64_t read) {
    if (im->constant) return im->value;
    uint32_t u = rng(seed, read, ST_IDENT, 0).v[0];
    uint32_t idx = u >> 16; double fr = (double)(u & 0xffffu) * (1.0 / 65536.0);
    double a = im->qtab[idx], b = im->qtab[idx + 1];
    double q = a + (b - a) * fr;
    return im->value * q;
}

/* ------------------------------------------------------------------ tail noise
 * TAIL_NOISE_MODEL_PY.KDE_noise_generator.noise_seq (py/tksm_badread.py:919-933), Custom2Dist.__call__ (:1023-1033),
 * CustomDist (:975-991).  The noise is appended to the fragment before the k-base pads (:334-341); with the
 * default `no_noise` model it is empty.
 *   with probability 1 - ratio: nothing;
 *   row   = np.searchsorted(ly, frag_len) (the :1025-1027 adjustment only moves for unsorted labels), past the
 *           last label: last row and the quirk factor len(ly) / ly[-1] (:1029);
 *   x     = int(lx[np.searchsorted(cdf_row, uniform)] * factor), cdf_row[i] = pdf[i] / sum + cdf_row[i-1];
 *   chain = first state uniform over 4, then x steps of random.choices(range(4), trans[state]) (bisect_right on the
 *           running sums scaled by their total), emitting bases[state] after each step.
 * Philox stream ST_TAIL: counter 0 = {ratio draw, length draw, first state, -}; step t uses word t & 3 of counter
 * 1 + (t >> 2).  Uniforms are word * 2^-32. */
typedef struct {
    int32_t n_lx, n_ly;
    const double* lx;     /* [n_lx] tail lengths */
    const double* ly;     /* [n_ly] fragment-length labels */
    const double* grid;   /* [n_ly][n_lx] densities */
    double trans[16];     /* [4][4] transition weights */
    double ratio;
    uint8_t bases[4]; uint8_t pad[4];
} tail_model;

int oracle_tail_length(const tail_model* tm, int frag_len, uint64_t seed, uint64_t read) {
    const ph4 w = rng(seed, read, ST_TAIL, 0);
    const double two32 = 1.0 / 4294967296.0;
    if ((double)w.v[0] * two32 > tm->ratio) return 0;
    int pos = 0;
    while (pos < tm->n_ly && tm->ly[pos] < (double)frag_len) pos++;
    if (pos < tm->n_ly - 1 && fabs(tm->ly[pos] - (double)frag_len) > fabs(tm->ly[pos + 1] - (double)frag_len)) pos++;
    double mult = 1.0;
    if (pos >= tm->n_ly) { mult = (double)pos / tm->ly[tm->n_ly - 1]; pos = tm->n_ly - 1; }
    const double* pdf = tm->grid + (size_t)pos * tm->n_lx;
    double sum = 0.0;
    for (int i = 0; i < tm->n_lx; i++) sum += pdf[i];
    const double val = (double)w.v[1] * two32;
    double c = 0.0; int p2 = tm->n_lx - 1;
    for (int i = 0; i < tm->n_lx; i++) { c = pdf[i] / sum + c; if (c >= val) { p2 = i; break; } }
    const double x = tm->lx[p2] * mult;
    if (!(x >= 1.0)) return 0;
    return x > 1e9 ? 1000000000 : (int)x;
}

int oracle_tail_noise(const tail_model* tm, int frag_len, uint64_t seed, uint64_t read, uint8_t* out, int cap) {
    const int x = oracle_tail_length(tm, frag_len, seed, read);
    if (x > cap) return -1;
    const double two32 = 1.0 / 4294967296.0;
    int state = (int)(rng(seed, read, ST_TAIL, 0).v[2] >> 30);
    double cum[16];
    for (int s = 0; s < 4; s++) { double c = 0.0; for (int j = 0; j < 4; j++) { c += tm->trans[4 * s + j]; cum[4 * s + j] = c; } }
    ph4 w = {{0, 0, 0, 0}};
    for (int t = 0; t < x; t++) {
        if ((t & 3) == 0) w = rng(seed, read, ST_TAIL, 1u + (uint32_t)(t >> 2));
        const double v = (double)w.v[t & 3] * two32 * cum[4 * state + 3];
        int nx = 0;
        for (int j = 0; j < 3; j++) nx += cum[4 * state + j] <= v;
        state = nx;
        out[t] = tm->bases[state];
    }
    return x;
}

/* py/tksm_badread.py:324-451.  raw = error-free sequence (bytes).  Outputs the UNTRIMMED new
 * sequence/quals plus trims, exactly like the reference's locals, then the caller trims.
 * use_full: 1 = unbanded DP everywhere (slow, test only). */
int oracle_sequence_fragment(const uint8_t* raw, int raw_len, double target_identity,
                             const err_model* em, const qs_model* qm, int compute_q,
                             uint64_t seed, uint64_t read, int use_full,
                             uint8_t* out_seq, uint8_t* out_qual, int* out_len, double* out_identity,
                             frag_stats* st) {
    int k = em->k;
    fstate s;
    s.len = raw_len + 2 * k;
    int L = s.len;
    s.frag = (uint8_t*)malloc((size_t)L);
    s.slen = (uint8_t*)malloc((size_t)L);
    s.sb = (uint8_t(*)[8])malloc((size_t)L * 8);
    s.changed = (uint8_t*)calloc((size_t)L, 1);
    uint8_t* joined = (uint8_t*)malloc((size_t)L * 6 + 16);
    uint32_t* cen = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)L * 6 + 16));   /* owner of each joined base */
    memset(st, 0, sizeof(*st));
    /* :334-341 pad with k random bases each side (the caller has appended the tail noise to raw) */
    ph4 pad = rng(seed, read, ST_PAD, 0);
    for (int j = 0; j < k; j++) {
        s.frag[j] = (uint8_t)BASES[(pad.v[0] >> (2 * j)) & 3];
        s.frag[k + raw_len + j] = (uint8_t)BASES[(pad.v[1] >> (2 * j)) & 3];
    }
    memcpy(s.frag + k, raw, (size_t)raw_len);
    for (int p = 0; p < L; p++) { s.slen[p] = 1; s.sb[p][0] = s.frag[p]; }

    double errors = 0.0, frag_len = (double)L;
    int change_count = 0; int64_t loop_count = 0;
    int max_kmer_index = L - 1 - k;
    uint32_t n = 0, aln_no = 0;
    for (;;) {
        loop_count++;
        if (loop_count > 100 * (int64_t)L) break;
        if ((double)change_count > 0.9 * frag_len) break;
        double est = 1.0 - errors / frag_len;
        if (est <= target_identity) break;
        ph4 d = rng(seed, read, ST_DRAW, n); n++;
        st->n_draws++;
        int i = (int)mulhi32(d.v[0], (uint32_t)(max_kmer_index + 1));
        /* add_errors_to_kmer :119-144 */
        uint8_t nlen[8]; uint8_t nb[8][8];
        int kidx = 0, valid = 1;
        for (int j = 0; j < k; j++) { int c = code_of(s.frag[i + j]); if (c < 0) valid = 0; kidx = (kidx << 2) | (c & 3); }
        int random_change = 0;
        if (em->type == 0 || !valid) random_change = 1;
        else {
            const uint32_t* cdf = em->cdf + (size_t)kidx * em->max_alts;
            int na = em->nalts[kidx], a = 0;
            while (a < na && !(d.v[1] < cdf[a])) a++;
            if (a == na) random_change = 1;
            else {
                uint64_t alt = em->alts[(size_t)kidx * em->max_alts + a];
                if (alt & ALT_NOOP) { st->n_noop++; continue; }   /* :375-376 */
                int b = 0;
                for (int j = 0; j < k; j++) {
                    nlen[j] = (uint8_t)alt_slot_len(alt, j);
                    for (int x = 0; x < nlen[j]; x++) nb[j][x] = (uint8_t)BASES[alt_base(alt, b++)];
                }
            }
        }
        if (random_change) {                                        /* :199-213 */
            st->n_random_change++;
            for (int j = 0; j < k; j++) { nlen[j] = 1; nb[j][0] = s.frag[i + j]; }
            int type = (int)mulhi32(d.v[2], 3);
            int pos = (int)mulhi32(d.v[3], (uint32_t)k);
            int base4 = (int)(d.v[3] & 3), side = (int)((d.v[3] >> 2) & 1);
            int r3 = (int)((((d.v[2] & 0xffffu) * 3u) >> 16) + 1);
            if (type == 0) {
                int c = code_of(nb[pos][0]);
                nb[pos][0] = (uint8_t)BASES[c < 0 ? base4 : ((c + r3) & 3)];
            } else if (type == 1) {
                nlen[pos] = 2;
                if (side) { nb[pos][1] = (uint8_t)BASES[base4]; }                    /* base + random */
                else { nb[pos][1] = nb[pos][0]; nb[pos][0] = (uint8_t)BASES[base4]; } /* random + base */
            } else nlen[pos] = 0;
        }
        st->n_kmers_applied++;
        for (int j = 0; j < k; j++) {                               /* :378-432 */
            int p = i + j;
            int differs = !(nlen[j] == 1 && nb[j][0] == s.frag[p]);
            if (!differs || s.changed[p]) continue;
            s.changed[p] = 1; s.slen[p] = nlen[j];
            for (int x = 0; x < nlen[j]; x++) s.sb[p][x] = nb[j][x];
            change_count++;
            int new_errors = nlen[j] < 2 ? 1 : nlen[j] - 1;
            if (nlen[j] == 0) st->n_del++; else if (nlen[j] == 1) st->n_sub++; else { st->n_ins_slots++; st->ins_bases += nlen[j] - 1; }
            errors += (double)new_errors * (est * sqrt(est));       /* est ** 1.5 */
            if (change_count % 25 == 0) {                           /* ALIGNMENT_INTERVAL */
                int mt = 0, cols = 0, dist;
                st->n_aligns++;
                if (L <= 1000) {                                    /* ALIGNMENT_SIZE */
                    int m = join_window(&s, 0, L, joined, cen);
                    dist = use_full ? full_align(s.frag, L, joined, m, 0, &mt, &cols, NULL, NULL)
                                    : band_align(s.frag, L, joined, m, cen, 0, &mt, &cols, NULL, NULL);
                    if (dist < 0) { st->band_fail++; dist = full_align(s.frag, L, joined, m, 0, &mt, &cols, NULL, NULL); }
                    double ident = cols ? (double)mt / (double)cols : 0.0;
                    errors = (1.0 - ident) * frag_len;
                } else {
                    uint32_t w = rng(seed, read, ST_ALNPOS, aln_no).v[0];
                    int pos = (int)mulhi32(w, (uint32_t)(L - 1000 + 1));
                    int m = join_window(&s, pos, 1000, joined, cen);
                    dist = use_full ? full_align(s.frag + pos, 1000, joined, m, 0, &mt, &cols, NULL, NULL)
                                    : band_align(s.frag + pos, 1000, joined, m, cen, 0, &mt, &cols, NULL, NULL);
                    if (dist < 0) { st->band_fail++; dist = full_align(s.frag + pos, 1000, joined, m, 0, &mt, &cols, NULL, NULL); }
                    double ident = cols ? (double)mt / (double)cols : 0.0;
                    double estimated = (1.0 - ident) * frag_len;
                    double weight = 1000.0 / frag_len;
                    errors = estimated * weight + errors * (1.0 - weight);
                }
                aln_no++;
            }
        }
    }
    st->change_count = change_count; st->errors = errors; st->frag_len = L; st->target_identity = target_identity;
    /* :434-437 */
    int start_trim = 0, end_trim = 0;
    for (int j = 0; j < k; j++) { start_trim += s.slen[j]; end_trim += s.slen[L - k + j]; }
    int m = join_window(&s, 0, L, joined, cen);
    st->new_len = m; st->start_trim = start_trim; st->end_trim = end_trim;
    double actual_identity;
    uint8_t* qual = (uint8_t*)malloc((size_t)m + 1);
    if (compute_q && m > 0) {
        /* get_qscores :607-655; cigar of edlib.align(seq, frag): 'I' = read-only base, 'D' = fragment-only */
        uint8_t* ops = (uint8_t*)malloc((size_t)L + (size_t)m + 8);
        int nops = 0, mt = 0, cols = 0;
        if (g_qalign_variant) {                                   /* test-only: see oracle_set_qscore_alignment_variant */
            int splits = 0;
            if (g_qalign_variant == 1) nw_trace_pref(joined, m, s.frag, L, 1, ops, &nops);
            else if (g_qalign_variant == 2) edlib_like_path(joined, m, s.frag, L, ops, &nops, &splits);
            else nw_trace_pref(joined, m, s.frag, L, 0, ops, &nops);
            for (int a = 0; a < nops; a++) mt += ops[a] == '=';
            cols = nops;
            st->pad0 = splits;
        } else if (use_full) full_align(s.frag, L, joined, m, 1, &mt, &cols, ops, &nops);
        else {
            uint8_t* trace = (uint8_t*)malloc((size_t)(m + 1) * 64);
            int32_t* ttop = (int32_t*)malloc(sizeof(int32_t) * ((size_t)m + 1));
            int dist = band_align(s.frag, L, joined, m, cen, 1, &mt, &cols, trace, ttop);
            if (dist < 0) { st->band_fail++; full_align(s.frag, L, joined, m, 1, &mt, &cols, ops, &nops); }
            else {
                int r = L, j = m, kk = 0;
                while (r > 0 || j > 0) {
                    int mv;                                  /* 0 up, 1 left, 2 diag */
                    if (j == 0) mv = 0;
                    else if (r == 0) mv = 1;
                    else {
                        int bb = r - ttop[j];
                        if (bb > 63) mv = 0;                 /* virtual cell below the window */
                        else { uint8_t tb = trace[(size_t)j * 64 + bb]; mv = (tb & 2) ? 1 : ((tb & 1) ? 0 : 2); }
                    }
                    if (mv == 1) { ops[kk++] = 'I'; j--; }
                    else if (mv == 0) { ops[kk++] = 'D'; r--; }
                    else { ops[kk++] = (s.frag[r - 1] == joined[j - 1]) ? '=' : 'X'; r--; j--; }
                }
                for (int a = 0, b = kk - 1; a < b; a++, b--) { uint8_t x = ops[a]; ops[a] = ops[b]; ops[b] = x; }
                nops = kk;
            }
            free(ttop);
            free(trace);
        }
        actual_identity = cols ? (double)mt / (double)cols : 0.0;
        /* per read position: op and number of D columns between it and the previous read position */
        uint8_t* pop = (uint8_t*)malloc((size_t)m); uint32_t* dbef = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)m);
        { int rp = 0; uint32_t dp = 0;
          for (int a = 0; a < nops; a++) {
              if (ops[a] == 'D') dp++;
              else { pop[rp] = ops[a] == '=' ? 0 : (ops[a] == 'X' ? 1 : 2); dbef[rp] = dp; dp = 0; rp++; }
          } }
        int margins = (qm->kmer_size - 1) / 2;
        for (int i2 = 0; i2 < m; i2++) {
            int s0 = i2 - margins, e0 = i2 + margins;
            while (s0 < 0 || e0 >= m) { s0++; e0--; }
            int row = -1;
            for (;;) {                                              /* get_qscore :584-598 */
                uint64_t key = 0; int len = 0, ok = 1;
                for (int x = s0; x <= e0 && ok; x++) {
                    if (x > s0) for (uint32_t z = 0; z < dbef[x]; z++) { if (len >= 29) { ok = 0; break; } key |= 3ull << (2 * len); len++; }
                    if (!ok || len >= 29) { ok = 0; break; }
                    key |= (uint64_t)pop[x] << (2 * len); len++;
                }
                if (ok) { key |= (uint64_t)len << 58; row = qs_find(qm, key); }
                if (row >= 0 || s0 == e0) break;
                s0++; e0--;
            }
            uint8_t q = 0;
            if (row >= 0) {
                uint32_t w = rng(seed, read, ST_QUAL, (uint32_t)i2).v[0];
                const uint32_t* cdf = qm->cdf_pool + qm->row_off[row];
                uint32_t cnt = qm->row_cnt[row], a = 0;
                while (a + 1 < cnt && !(w < cdf[a])) a++;
                q = qm->q_pool[qm->row_off[row] + a];
            }
            qual[i2] = (uint8_t)(q + 33);
        }
        free(pop); free(dbef); free(ops);
    } else {
        memset(qual, 'K', (size_t)m);
        actual_identity = 1.0 - errors / frag_len;
    }
    /* :448-449  seq[start_trim:-end_trim] -- end_trim == 0 gives an empty slice (reference quirk) */
    int lo = start_trim, hi = (end_trim == 0) ? 0 : m - end_trim;
    if (lo > m) lo = m;
    if (hi < lo) hi = lo;
    *out_len = hi - lo;
    memcpy(out_seq, joined + lo, (size_t)(hi - lo));
    memcpy(out_qual, qual + lo, (size_t)(hi - lo));
    *out_identity = actual_identity;
    free(qual); free(cen); free(joined); free(s.changed); free(s.sb); free(s.slen); free(s.frag);
    return 0;
}

/* ------------------------------------------------------------------ record formatting */
/* '{:.2f}'.format(identity*100) -- correctly rounded like CPython's float formatting */
static int64_t pct_hundredths(double identity) {
    double e = identity * 100.0;
    double r = nearbyint(e * 100.0);
    double err = fma(e, 100.0, -r);      /* exact residual of e*100 - r */
    if (err > 0.5) r += 1.0; else if (err < -0.5) r -= 1.0;
    else if (err == 0.5) { if (fmod(r, 2.0) != 0.0) r += 1.0; }
    else if (err == -0.5) { if (fmod(r, 2.0) != 0.0) r -= 1.0; }
    return (int64_t)r;
}
int64_t oracle_pct_hundredths(double identity) { return pct_hundredths(identity); }

static int put_uuid(uint8_t* o, uint64_t seed, uint64_t read) {
    ph4 p = rng(seed, read, ST_ID, 0);
    static const char hx[] = "0123456789abcdef";
    int k = 0, nib = 0;
    for (int w = 0; w < 4; w++)
        for (int b = 7; b >= 0; b--) {
            if (nib == 8 || nib == 12 || nib == 16 || nib == 20) o[k++] = '-';
            o[k++] = (uint8_t)hx[(p.v[w] >> (4 * b)) & 15]; nib++;
        }
    return k; /* 36 */
}

/* py/sequence.py:252-288.  fastq != 0: '@id info\nSEQ\n+\nQUAL\n', else '>id info\nSEQ\n'. */
int64_t oracle_format_record(uint8_t* o, int fastq, uint64_t seed, uint64_t read,
                             const uint8_t* seq, const uint8_t* qual, int64_t len, int64_t error_free_len,
                             double identity, const uint8_t* mol_id, int mol_id_len) {
    int64_t k = 0;
    o[k++] = fastq ? '@' : '>';
    k += put_uuid(o + k, seed, read);
    int64_t h = pct_hundredths(identity);
    k += sprintf((char*)o + k, " length=%lld error_free_length=%lld read_identity=%lld.%02lld%% molecule_id=",
                 (long long)len, (long long)error_free_len, (long long)(h / 100), (long long)(h % 100));
    memcpy(o + k, mol_id, (size_t)mol_id_len); k += mol_id_len;
    o[k++] = '\n';
    memcpy(o + k, seq, (size_t)len); k += len; o[k++] = '\n';
    if (fastq) { o[k++] = '+'; o[k++] = '\n'; memcpy(o + k, qual, (size_t)len); k += len; o[k++] = '\n'; }
    return k;
}
