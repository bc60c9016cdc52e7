// kernels.h -- device-side views and launchers of the Seq hot path (see DESIGN.md for layouts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tk {

constexpr uint32_t NO_BLOCK = 0xFFFFFFFFu;   // blocktab entry: block is pure 2-bit
constexpr int BLOCK_SHIFT = 12;              // 4096-base exception blocks
constexpr int WAVES_PER_WG = 4;

struct RefView {
    const uint32_t* packed;     // 16 bases per word, base g at bits 2*(g&15)
    const uint32_t* blocktab;   // [total_blocks] NO_BLOCK or index into pool
    const uint8_t* pool;        // [n_pool_blocks][4096] upper-cased bytes
    const uint64_t* contigs;    // [n][2] {gstart (block aligned), len}
    uint32_t n_contigs;
};

struct BatchView {
    const uint32_t* reads;      // [n_reads][2] {ivl_begin, ivl_count}
    const uint32_t* intervals;  // [n_intervals+1][4]
    const uint32_t* mods;       // [n_mods][2]
    const uint64_t* literals;   // [n_literals][2]
    const uint8_t* litpool;
    const uint32_t* ids;        // [n_reads][2]
    const uint8_t* idpool;
    uint64_t n_reads;
    uint32_t n_literals;
};

struct ErrModelView {
    int type, k, max_alts;
    int alt0_noop;         // alternative 0 of every row is the k-mer itself (true for Badread model files)
    int uniform_nalts;     // every k-mer row has exactly max_alts alternatives: nalts need not be read
    const uint32_t* cdf;
    const uint64_t* alts;
    const uint8_t* nalts;
    const uint2* pself2;    // [4^k] {first, last} threshold of every row (cache-resident, 128 KB)
    const uint32_t* cdf32;  // [4^k][32] the same thresholds padded to 128-byte rows (max_alts <= 32 only)
    const uint4* pseg;      // [4^k] thresholds 0, 8, 16 and 24 of every row: a draw's first-level lookup in k_loop (k-mer itself /
                            // which 8-threshold segment of cdf32 to read)
    const uint32_t* pt0;    // [4^k] threshold 0 alone (64 KB): what EVERY draw of k_loop looks at first (four times as many entries per cache line as pseg)
    const uint4* alts_enc;  // [4^k][max_alts] the alternatives as the fast pipeline applies them: eight 16-bit slot
                            // encodings (length << 12 | 2-bit codes), bit 15 = the slot differs from the k-mer's base
};

struct QsModelView {
    int n_slots, kmer_size;
    const uint64_t* keys;
    const uint32_t* row_off;
    const uint32_t* row_cnt;
    const uint32_t* cdf_pool;
    const uint8_t* q_pool;
    // the same model in the layout k_err reads: one 16-byte hash entry {key, row offset, row count}, rows as
    // {threshold, q} pairs, and a 64-bucket guide per hash slot: the first candidate index for the quantile bucket of the draw -- or,
    // with guide_direct, 0x80 | q where every draw of the bucket picks the same entry (no row read at all)
    const uint4* ent;
    const uint2* pairs;
    const uint8_t* guide;
    int guide_direct;
};

struct IdentView {
    int constant;
    double value;
    const double* qtab;
};

// tail noise (TAIL_NOISE_MODEL_PY.KDE_noise_generator, py/tksm_badread.py:886-1050): length sampler tables and the base chain
struct TailView {
    int n_lx, n_ly;
    double ratio;
    const double* lx;    // [n_lx] tail lengths
    const double* ly;    // [n_ly] fragment-length labels
    const double* cdf;   // [n_ly][n_lx] running sums of the normalised rows
};
struct TailChain {
    double cum[16];      // running sums of the 4 transition rows
    uint32_t bases;      // the 4 output bytes, state 0 in the low byte
    uint32_t pad;
};

struct SimParams {
    uint64_t seed, first_read, stride;
    int mode, compute_q, fastq, quirk_perfect;
    int lcap;          // LDS capacity for the padded fragment (bytes, multiple of 4)
    int ncap;          // LDS capacity for the joined new sequence (bytes, multiple of 4)
    int trace_words;   // u32 words per wave in the traceback scratch
    int s_lcap, s_ncap; // LDS geometry of the wave-wide kernel (k_simulate); reads beyond it get status 8
    int ablate;        // diagnostic builds only
    int cap_num, cap_den, cap_add;   // per-read scratch capacity = (raw+2k)*num/den + add, 16-aligned
};

struct SimBuffers {
    const uint32_t* raw_len;     // [n_reads]
    const uint64_t* slot_off;    // [n_reads] byte offset of the read's scratch slot (seq | qual)
    uint8_t* scratch;
    uint32_t* out_len;           // [n_reads]
    double* identity;            // [n_reads]
    uint64_t* rec_len;           // [n_reads] formatted record length
    uint32_t* status;            // [n_reads] 0 ok, bit0 overflow, bit1 bad mod position, bit2 alignment fallback out of memory, bit3 too long for the wave-wide kernel, bit4 (informational) an alignment took the unbanded fallback
    uint32_t* trace;             // [n_waves][trace_words]
    unsigned long long* work_counter;
    int32_t* istats;             // optional [n_reads][16]
    double* dstats;              // optional [n_reads][2]
    // k_simulate only: memory for the unbanded alignment of the rare window the guided band cannot hold (bump allocator)
    uint8_t* full_pool; unsigned long long full_pool_bytes; unsigned long long* full_pool_used;
    const uint32_t* tail_len;    // optional [n_reads]: tail-noise bases at the end of the read's raw_len (else none)
    const TailChain* tail_chain;
    // k_simulate<BIG> only: per-wave working sets and final-alignment masks in HBM (molecules beyond the LDS-resident limit)
    uint8_t* big_scratch; size_t big_per_wave; unsigned long long* big_trace;
    const uint32_t* read_list;   // k_simulate only: optional list of reads to process (slow path), else all
    uint64_t n_work;             // k_simulate only: number of work items (list length or n_reads)
};

// per-read state of the fast Badread pipeline between k_err rounds
struct ReadState {
    double errors, target, est;
    int32_t change_count;
    uint32_t n_base, aln_no;
    int16_t resume_src, resume_j;   // resume_j > 0: draw n_base was interrupted by a re-estimation after its slot resume_j - 1 (resume_src unused)
    uint8_t stage;                  // 0 error loop, 1 waiting for the q-score alignment, 2 done, 3 error loop done, q-score alignment deferred,
                                    // 4 error loop done in this round (k_loop), trims / q-score job / output still to do (k_err)
    uint8_t pending, slow, early;   // pending: 1 an alignment result waits to be applied, 2 k_loop asks k_err for an alignment job;
                                    // early: the read's error loop runs on a wave of its own from round 0 on (predicted straggler: k_mark_early)
    int32_t st_draws, st_aligns;
    uint32_t job;
    int32_t raw_len;
    // result of the read's last alignment, written by k_aln: one record load brings everything k_err needs
    uint32_t res_mt, res_cols, res_fail, pad3;
};

// Where the jobs of one range of the sorted read order live in this round's job set.  Rows are as long as the longest
// read of the range needs (a batch's longest read is twice its mean), and ranges are packed one after the other.
struct RangeGeo {
    uint64_t trace_off;   // 64-byte lines into trace
    uint64_t jc_off;      // (unused since the block records went: round 3)
    uint64_t popd_off;    // bytes into job_popd
    uint64_t unused;
    uint32_t tstride;     // lines of predecessor codes per job (16 columns each; the last one is spare)
    uint32_t cw;          // (unused)
    uint32_t ncap;        // joined-window capacity of the range (multiple of 16)
    uint32_t pad;
};

struct FastBuffers {
    ReadState* state;                 // [n_reads]
    const uint32_t* row64;            // [n_reads + 1] first 64-position block of every read's state rows (ragged; kernels.hip frag_row ...)
    uint8_t* st_frag;                 // [blocks][64] 2-bit codes of the padded fragments, one per byte
    uint16_t* st_nb;                  // [blocks][64] slot codes
    unsigned long long* st_fplanes;   // [blocks + 8 n_reads][2] 2-bit planes of the padded fragments, {lo, hi} word pairs
    uint32_t* st_frag2;               // [4 blocks + 4 n_reads] the padded fragments, 16 bases per word, first base in the top bits: what the
                                      // error loop (k_loop, one LANE per read) keeps in LDS and cuts its k-mers from
    uint32_t* job_meta;               // [n_reads][4] {read, p0, n | mode << 31, -}
    uint8_t* job_popd;                // [n_reads][ncap] per read position: op | D-run << 2 (q-score jobs)
    void* trace;                      // predecessor codes of the first alignment pass (14 band rows + the column's shift: 4 bytes per iteration):
                                      // 64-byte lines of 16 iterations, per range [wave][line][lane] (RangeGeo::trace_off / tstride)
    uint32_t* redo_list;              // jobs whose path left the stored rows of pass 1 (counters[10] of them): 64-row pass
    void* trace_full;                 // pool of the passes that store all 64 rows (16 bytes per iteration, 4 per line; shift bytes behind them): [wave][full_tg lines][lane]; counters[3] allocates
    uint32_t full_rows, full_tg;      // jobs the full-width pool holds (multiple of 64), lines per job (4 columns each)
    uint32_t full_cl;                 // fused alignment (k_alnf): of a job's full_tg lines the first full_cl hold codes, the shift bytes follow, the last one is spare
    uint32_t* counters;               // [2] reads on the slow list, [3] rows taken from the full-width pool, [4..9] diagnostics, [10] jobs on redo_list
    uint32_t* job_cnt;                // [n_ranges] jobs of this round per range of rs reads, one counter per 128 B
    uint32_t rs, n_ranges;
    // previous round's job set (double buffered): its jobs are the list of reads that are still running
    const uint32_t* prev_meta; const uint8_t* prev_popd;
    const uint32_t* prefix;           // [n_ranges + 1] exclusive prefix sum of the previous round's per-range job counts
    // first job id of every range in this round's / the previous round's job set.  Round 0: c * rs; later rounds: the
    // previous round's counts rounded up to whole waves and packed -- a round's jobs, records and predecessor columns
    // stay dense as the reads finish (fewer pages touched, no empty waves)
    const uint32_t* base_cur; const uint32_t* base_prev;
    const RangeGeo* geo_cur; const RangeGeo* geo_prev;   // [n_ranges]
    uint32_t* slow_list;              // [n_reads]
    // predicted stragglers (reads whose length x (1 - target identity) says they need several times the visits of the batch's median
    // read): their error loops run from round 0 on, a wave each, on a side stream underneath the regular rounds
    uint32_t* early_hist;             // [64][256] reads per score bin (k_init; 64 copies against contention), bin = 8 log2(1 + length x (1 - target))
    uint2* early_list;                // {read, range}; counters[27] counts
    uint32_t* early_slow;             // early reads that need the exact kernel; counters[26] counts (merged into slow_list after the side kernel)
    // reads longer than defer_len wait (stage 3) with their q-score alignment until the regular rounds are over
    uint2* defer_list;                // [n_reads] {read, range}; counters[1] counts
    uint32_t* defer_cnt;              // [n_ranges] per range, one counter per 128 B
    int defer_len;
};

hipError_t launch_pack(const uint8_t* ascii, uint64_t n, uint64_t gstart, uint32_t* packed, uint32_t* blockflag,
                       hipStream_t s);
hipError_t launch_fill_pool(const uint8_t* ascii, uint64_t n, uint64_t gstart, const uint32_t* blocktab,
                            uint8_t* pool, hipStream_t s);
hipError_t launch_read_lengths(const BatchView& b, const RefView& r, int k, int cap_num, int cap_den, int cap_add,
                               const uint32_t* tail_len, uint32_t* raw_len, uint64_t* slot_cap, uint32_t* status, hipStream_t s);
hipError_t launch_tail_lengths(const BatchView& B, const RefView& R, const TailView& T, uint64_t seed, uint64_t first_read,
                               uint64_t stride, uint32_t* tail_len, hipStream_t s);
hipError_t launch_simulate(const BatchView& b, const RefView& r, const ErrModelView& em, const QsModelView& qm,
                           const IdentView& im, const SimParams& p, const SimBuffers& o, int n_wgs,
                           int waves_per_wg, hipStream_t s);
size_t simulate_big_bytes(int lcap, int ncap);
hipError_t launch_simulate_big(const BatchView& b, const RefView& r, const ErrModelView& em, const QsModelView& qm,
                               const IdentView& im, const SimParams& p, const SimBuffers& o, int n_waves, hipStream_t s);
hipError_t launch_init(const BatchView& b, const RefView& r, const ErrModelView& em, const IdentView& im, const SimParams& p,
                       const SimBuffers& o, const FastBuffers& fb, int waves_per_wg, hipStream_t s);
int err_lds_bytes(int lcap, int ncap, int waves_per_wg, bool state_in_hbm);
// error loop, one lane per read: reads order[begin .. begin+count) (from_jobs = 0) or the reads of the previous round's jobs
// of ranges [c0, c1) (from_jobs = 1); words = fragment words per lane held in LDS (0: fragments stay in HBM)
int loop_lds_words(int lcap);
// the same visit with one wave per read (rounds with few reads left; the read's slot codes are staged in LDS: lcap <= 32768)
hipError_t launch_loopw(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, const uint32_t* order, uint32_t begin,
                        uint32_t count, int lcap, int from_jobs, uint32_t c0, uint32_t c1, hipStream_t s);
// the same visits, and every later visit of the read, on one wave (k_loopw<true>: the alignments by band_align on the wave)
hipError_t launch_tail(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, const uint32_t* order, uint32_t begin,
                        uint32_t count, int lcap, int from_jobs, uint32_t c0, uint32_t c1, int wcap, hipStream_t s);
size_t tail_lds_bytes(int lcap);
hipError_t launch_mark_early(const FastBuffers& fb, const uint32_t* order, uint64_t n_reads, int k, uint32_t min_bin, hipStream_t s);
hipError_t launch_tail_early(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, uint32_t max_count, int lcap, int wcap, hipStream_t s);
hipError_t launch_merge_early_slow(const FastBuffers& fb, hipStream_t s);
hipError_t launch_loop(const ErrModelView& em, const SimParams& p, const FastBuffers& fb, const uint32_t* order, uint32_t begin,
                       uint32_t count, int lcap, int from_jobs, uint32_t c0, uint32_t c1, hipStream_t s);
hipError_t launch_err(const BatchView& b, const ErrModelView& em, const QsModelView& qm, const SimParams& p, const SimBuffers& o,
                      const FastBuffers& fb, const uint32_t* order, uint32_t begin, uint32_t count, int lds_lcap, int lds_ncap,
                      int from_jobs, uint32_t c0, uint32_t c1, int waves_per_wg, bool state_in_hbm, hipStream_t s);
// q-score jobs for the first `count` reads of fb.defer_list (after the last regular round)
hipError_t launch_qjobs(const FastBuffers& fb, int k, uint32_t count, hipStream_t s);
hipError_t launch_round_reset(const FastBuffers& fb, hipStream_t s);
hipError_t launch_collect_unfinished(const FastBuffers& fb, uint64_t n_reads, hipStream_t s);
// this round's alignment jobs (ids below n_jobs; per-range counts in fb.job_cnt): windows decoded from the slot codes and aligned, one
// lane per job.  mode: 0 = the round's jobs are identity re-estimations, 1 = q-score alignments (all jobs of a round have one mode)
hipError_t launch_alnf(const SimParams& p, const FastBuffers& fb, const SimBuffers& o, uint32_t n_jobs, bool full_only, int mode, unsigned lds_pad, hipStream_t s);
hipError_t launch_perfect_lengths(const BatchView& b, const RefView& r, const SimParams& p, const SimBuffers& o, hipStream_t s);
hipError_t launch_perfect(const BatchView& b, const RefView& r, const SimParams& p, const SimBuffers& o, const uint64_t* rec_off, uint8_t* records,
                          uint32_t max_raw, int n_cus, hipStream_t s);
hipError_t launch_emit(const BatchView& b, const SimParams& p, const SimBuffers& o, const uint64_t* rec_off,
                       uint8_t* records, hipStream_t s);
hipError_t launch_interleave_lens(int n_ranks, const uint64_t* const* offsets, const uint64_t* n_per_rank,
                                  uint64_t n_total, uint64_t* lens, hipStream_t s);
hipError_t launch_interleave_copy(int n_ranks, const uint8_t* const* streams, const uint64_t* const* offsets,
                                  uint64_t n_total, const uint64_t* dst_off, uint8_t* dst, hipStream_t s);
// exclusive scan of n u64 values (+ total in out[n]); temp storage managed by the caller through
// scan_temp_bytes().
size_t scan_temp_bytes(uint64_t n);
hipError_t launch_scan(const uint64_t* in, uint64_t* out, uint64_t n, void* temp, size_t temp_bytes, hipStream_t s);
hipError_t launch_sum_u32(const uint32_t* in, uint64_t n, unsigned long long* out, hipStream_t s);
int simulate_lds_bytes(int lcap, int ncap, int waves_per_wg);
int simulate_max_wgs(int lds_bytes);

}  // namespace tk
