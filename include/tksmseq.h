/*
 * tksmseq.h -- C-ABI of the MI355X-native TKSM `Seq` hot path (libtksmseq.so).
 *
 * This is the drop-in boundary for the path BASELINE.json names: TKSM's Seq exit module,
 *   src/sequence.cpp:21-57  (Sequencer_module::impl::run -> embedded Python)
 *   py/sequence.py:323-376  (main block: load reference + models, per-molecule loop, write)
 *   py/tksm_badread.py      (Badread identity / error / q-score model)
 * Every entry point below names the reference code it replaces (file:line into vpc-ccg/tksm).
 * Plain pointers and sizes only; no C++ or torch types; no exceptions cross this boundary.
 * All functions return TKSMSEQ_OK (0) or a TKSMSEQ_E* code; tksmseq_last_error() gives the text.
 * A context is used from one host thread at a time and owns one HIP device + stream.  Contexts are independent:
 * several may be driven from different threads on the same device (each on its own stream), and that is how batches
 * are streamed at full rate -- the kernels of concurrent runs fill each other's gaps (instruction-bound error loop next
 * to memory-bound alignment, the latency-bound last rounds of one batch underneath the bulk of the next).
 *
 * There is no CPU fallback: every compute entry point runs HIP kernels on gfx950 and fails with
 * TKSMSEQ_EDEVICE when no device is usable.
 *
 * Environment.  Results never depend on any of these: they choose between kernel variants that produce the same bytes (the GPU tests
 * use them to reach every variant) or print diagnostics.  Read when a context is created unless noted.
 *   TKSM_MODELS            colon list of model directories searched after the built-in one (src/sequence.cpp:38-52, py/sequence.py:17-31)
 *   TKSMSEQ_BUILTIN_MODELS the built-in model directory (default: models/ next to the library)
 *   TKSMSEQ_VERBOSE=1|2    per-run statistics on stderr (rounds, slow-path reads, full-width redo jobs, alignment fall-backs by reason);
 *                          2: per-batch stage times as well (also read by the CLI)
 *   TKSMSEQ_FORCE_SLOW=1   every read through the exact wave-wide kernel (k_simulate) instead of the fast pipeline
 *   TKSMSEQ_SMALL_ALN=N    rounds with at most N alignment jobs store all 64 band rows in one launch (default 131072; 0: always the
 *                          14-row pass + redo list)
 *   TKSMSEQ_SMALL_ROUND=N  (round 1's launch grouping; kept for the tests) rounds below N reads are launched merged (default 16384)
 *   TKSMSEQ_WAVE_LOOP=N    rounds with at most N reads left run the error loop one wave per read (k_loopw; default 16384, 0: never)
 *   TKSMSEQ_TAIL_WAVE=N    once at most N reads are left (and each can have a wave of its own at once) they finish in ONE launch that runs
 *                          every remaining visit of a read on one wave, alignments included (k_loopw<true>; default 4096, 0: never)
 *   TKSMSEQ_LOOP_WL=W      words (16 bases each) of a read's packed fragment that k_loop keeps in LDS per lane (default 64, multiple of 4)
 *   TKSMSEQ_EARLY_TAIL=N   at most N reads whose length x (1 - target identity) exceeds 4 x the batch's median get their straggler waves at
 *                          round 0, on a stream of their own underneath the regular rounds (default 1024, at most 4096; 0: never)
 *   TKSMSEQ_TAIL_WCAP=C    columns of a window the straggler kernel aligns on its wave (default and maximum 2048; a wider window takes the
 *                          regular route for that visit; the tests force that with a small value)
 *   TKSMSEQ_ALN_LDS_PAD=B  bytes of LDS the 14-row alignment pass asks for without using them: caps its waves per CU (default 0)
 *   TKSMSEQ_HBM_STATE_LEN=L fragments longer than L are edited in HBM by the last visit instead of being staged in LDS (default 2304)
 *   TKSMSEQ_DEFER_LEN=L    reads longer than L wait with their q-score alignment until the regular rounds are over (default 0: all)
 *   TKSMSEQ_BUCKETS=N      length buckets of the last visit's launches (default 16)
 *   TKSMSEQ_TAIL_CUT=N     diagnostic: once fewer than N reads are left, they finish in the wave-wide kernel (default 0: off)
 *   TKSMSEQ_FULL_POOL_MB=M memory for the unbanded alignment fallback of the wave-wide kernel (default 1024)
 *   TKSMSEQ_STATS_FILE=P   (CLI) `tksm sequence` writes one JSON object with the run's stage clocks to P: reads, batches, MDF bytes in, record
 *                          bytes out, seconds of set-up (devices, references, models), of streaming (first chunk read -> last record byte
 *                          written) and, summed over the threads of a stage, of parsing, running, device-side staging copies, waiting for
 *                          device-to-host pieces, write calls, waiting for the writer, reading + counting (bench.py's end-to-end leg)
 *   TKSMSEQ_PIECE_BYTES=B  (CLI) size of the page-locked pieces a batch's records pass through (default 64 MB; the tests use 4 KB)
 *   TKSMSEQ_ABLATE=N       only in the diagnostic build (`make ablate`, -DTKSM_ABLATE): timing experiments on the last visit's q-score loop
 *                          (40 - 45, tools/ablate_err.sh) and k_loop's prologue (33)
 *   GPU_MAX_HW_QUEUES      (HIP runtime) the CLI and bench.py set 16 when unset: a hardware queue per context in flight -- INTEGRATION.md
 */
#ifndef TKSMSEQ_H
#define TKSMSEQ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TKSMSEQ_OK 0
#define TKSMSEQ_EINVAL 1    /* bad argument / malformed input (reference: Python exception -> exit 1) */
#define TKSMSEQ_EIO 2       /* file could not be read / written */
#define TKSMSEQ_EDEVICE 3   /* HIP error or no device */
#define TKSMSEQ_ENOMEM 4
#define TKSMSEQ_ESTATE 5    /* call order violated (e.g. run before models are set) */
#define TKSMSEQ_ELIMIT 6    /* input exceeds a documented limit of this build (molecule too long) */

typedef struct tksmseq_ctx tksmseq_ctx;
typedef struct tksmseq_batch tksmseq_batch;

/* ---- lifetime ------------------------------------------------------------------------------
 * Replaces Py_Initialize + module globals (src/python_runner.h:44-73, py/sequence.py:323-345). */
int tksmseq_create(int device, tksmseq_ctx** out);
void tksmseq_destroy(tksmseq_ctx* ctx);
const char* tksmseq_last_error(const tksmseq_ctx* ctx);   /* ctx may be NULL: last create() error */
const char* tksmseq_version(void);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = library stream. */
int tksmseq_set_stream(tksmseq_ctx* ctx, void* hip_stream);
/* A second context on the same device that SHARES src's packed reference and model tables (read-only views; it gets
 * its own stream and work buffers).  Replaces what the reference's worker processes inherit by fork from the module
 * globals (multiprocessing.Pool, py/sequence.py:354-366; reference_seqs / error_model / qscore_model, :336-345).
 * Destroy clones before src.  Loading another model or contig into a clone gives that clone a private copy. */
int tksmseq_clone(const tksmseq_ctx* src, tksmseq_ctx** out);
/* Page-locked host memory for tksmseq_result_download destinations (a pageable destination halves the copy rate). */
int tksmseq_host_alloc(uint64_t bytes, void** out);
void tksmseq_host_free(void* p);
int tksmseq_synchronize(tksmseq_ctx* ctx);
/* Device memory of the caller's own (on ctx's device) and copies out of it on ctx's stream: what a writer needs to keep a batch's
 * records (tksmseq_result_copy_device into such a buffer) while the context that made them runs its next batch.  async != 0:
 * returns once the copy is queued (tksmseq_synchronize before the bytes are read). */
int tksmseq_device_alloc(tksmseq_ctx* ctx, uint64_t bytes, void** out);
void tksmseq_device_free(tksmseq_ctx* ctx, void* p);
int tksmseq_copy_to_host(tksmseq_ctx* ctx, void* dst_host, const void* src_device, uint64_t bytes, int async);

/* ---- reference genome (S0) -------------------------------------------------------------------
 * get_reference_seqs / generate_fasta, py/sequence.py:168-194: name = header up to the first
 * space; later contigs with the same name replace earlier ones.  The library packs to 2 bit/base
 * on the device (upper-cased; 4096-base blocks holding any non-ACGT byte are kept as bytes). */
int tksmseq_reference_add_fasta(tksmseq_ctx* ctx, const char* path /* .fa / .fa.gz */);
/* ascii: host pointer (on_device = 0) or device pointer (on_device = 1); caller keeps ownership. */
int tksmseq_reference_add_contig(tksmseq_ctx* ctx, const char* name, const uint8_t* ascii, uint64_t len,
                                 int on_device);
int tksmseq_reference_contig_id(const tksmseq_ctx* ctx, const char* name);   /* -1 if absent */
int tksmseq_reference_info(const tksmseq_ctx* ctx, uint64_t* n_contigs, uint64_t* total_bases,
                           uint64_t* device_bytes);

/* ---- models ----------------------------------------------------------------------------------
 * ErrorModel.__init__/load_from_file + align_kmers, py/tksm_badread.py:76-117, :146-197.
 * QScoreModel.__init__/load_from_file/random/ideal, py/tksm_badread.py:464-582.
 * name_or_path: "random" (both), "ideal" (q-score), a model name resolved through $TKSM_MODELS
 * (colon list, <dir>/badread/<name>.{error,qscore}.gz, py/sequence.py:17-31) or a file path. */
int tksmseq_load_error_model(tksmseq_ctx* ctx, const char* name_or_path);
int tksmseq_load_qscore_model(tksmseq_ctx* ctx, const char* name_or_path);
/* Tail noise, TAIL_NOISE_MODEL_PY.KDE_noise_generator (py/tksm_badread.py:886-962; sampled at :335, appended to the
 * fragment before the k-base pads at :336-341).  name_or_path: "no_noise" (the default: nothing is appended), a name
 * resolved through $TKSM_MODELS (<dir>/badread/<name>.tail.gz) or a JSON[.gz] file in the layout
 * KDE_noise_generator.save writes (:935-942).  The tail's bases count towards `length=` but not towards
 * `error_free_length=` (py/sequence.py:253-254).  --perfect output never carries a tail. */
int tksmseq_load_tail_model(tksmseq_ctx* ctx, const char* name_or_path);
typedef struct tksmseq_tail_model {
    uint32_t n_lx, n_ly;
    const double* lx;      /* [n_lx] tail lengths (Custom2Dist.lx) */
    const double* ly;      /* [n_ly] fragment-length labels, sorted (Custom2Dist.ly) */
    const double* grid;    /* [n_ly][n_lx] densities, >= 0, every row with a positive sum */
    double trans[16];      /* [4][4] transition weights of the base chain (transition_matrix[1]) */
    double ratio;          /* probability that a read gets a tail */
    uint8_t bases[4];      /* output byte of each chain state (default "AGTC") */
    uint8_t pad[4];
} tksmseq_tail_model;
/* the same from tables in memory; NULL switches the tail off */
int tksmseq_set_tail_model(tksmseq_ctx* ctx, const tksmseq_tail_model* model);
/* Identities.__init__ + beta_parameters, py/tksm_badread.py:703-757 (percent units, as the CLI). */
int tksmseq_set_identity(tksmseq_ctx* ctx, double mean, double max, double stdev);

/* Table read-back (host copies) so tests can compare the packed layouts with the oracle's.
 * Pass NULL for an array to query sizes only. */
int tksmseq_get_error_model(const tksmseq_ctx* ctx, int32_t* type, int32_t* k, int32_t* max_alts,
                            uint32_t* cdf, uint64_t* alts, uint8_t* nalts);
int tksmseq_get_qscore_model(const tksmseq_ctx* ctx, int32_t* n_slots, int32_t* kmer_size, uint64_t* pool_len,
                             uint64_t* keys, uint32_t* row_off, uint32_t* row_cnt, uint32_t* cdf_pool,
                             uint8_t* q_pool);
int tksmseq_get_identity(const tksmseq_ctx* ctx, int32_t* constant, double* value, double* beta_a,
                         double* beta_b, double* qtab /* [65537] or NULL */);

/* ---- molecule batches (MDF) ------------------------------------------------------------------
 * mdf_generator, py/sequence.py:197-221: header '+id\tdepth\tcomment', interval lines with exactly
 * 5 tab fields 'contig\tstart\tend\tstrand\tmods'; a molecule is emitted `depth` times.  A contig
 * name absent from the reference is a literal sequence (py/sequence.py:307).
 * Binary layout (what the kernels read; "algorithmic bytes" 8 + 16 S + 8 M per read):
 *   reads      [n_reads]      {u32 ivl_begin, u32 ivl_count}
 *   intervals  [n_intervals]  {u32 contig (bit31: literal index), u32 start, u32 end,
 *                              u32 mod_begin | strand_minus << 31}
 *   mods       [n_mods]       {u32 pos, u32 chr}
 *   literals   [n_literals]   {u64 off, u64 len} into literal_pool
 *   ids        [n_reads]      {u32 off, u32 len} into id_pool                                   */
typedef struct {
    uint64_t n_reads, n_intervals, n_mods, n_literals, literal_bytes, id_bytes;
    const uint32_t* reads;        /* [n_reads][2] */
    const uint32_t* intervals;    /* [n_intervals][4]; mods of interval i are [mod_begin_i, mod_begin_{i+1}) */
    const uint32_t* mods;         /* [n_mods][2] */
    const uint64_t* literals;     /* [n_literals][2] */
    const uint8_t* literal_pool;
    const uint32_t* ids;          /* [n_reads][2] */
    const uint8_t* id_pool;
} tksmseq_batch_desc;

/* Copies host arrays to the device (validated first). */
int tksmseq_batch_create(tksmseq_ctx* ctx, const tksmseq_batch_desc* host_desc, tksmseq_batch** out);
/* Parses MDF text (whole file or a chunk ending at a molecule boundary) and uploads it. */
int tksmseq_batch_from_mdf_text(tksmseq_ctx* ctx, const char* text, uint64_t len, tksmseq_batch** out);
/* The same for the MDF -> MDF modules (tksmseq_pcr, tksmseq_truncate), which the reference runs without a FASTA: contig names
 * the context does not know stay what they are (the writer prints them back), substitution positions are not checked. */
int tksmseq_molecules_from_mdf_text(tksmseq_ctx* ctx, const char* text, uint64_t len, tksmseq_batch** out);
int tksmseq_batch_info(const tksmseq_batch* b, uint64_t* n_reads, uint64_t* n_intervals, uint64_t* n_mods);
/* Releases a batch.  `ctx`: the context that last ran / transformed the batch -- its stream is drained first, then the batch's device
 * tables go back to the library's per-process cache of device blocks (not to the driver: hipFree waits for the whole device), from
 * which the next batch of about that size takes them.  A caller that used the batch on SEVERAL contexts frees it through the last one
 * after the others have been synchronised (tksmseq_synchronize).  ctx may be NULL once every context has been destroyed. */
void tksmseq_batch_free(tksmseq_ctx* ctx, tksmseq_batch* b);

/* ---- the hot path ----------------------------------------------------------------------------
 * One call = the body of the reference's per-molecule loop for every read of the batch:
 * mdf_to_seq (py/sequence.py:303-320) -> perfect (:261-270) or badread (:242-258) ->
 * sequence_fragment / get_qscores (py/tksm_badread.py:324-451, :607-655) -> fastq/fasta
 * formatter (py/sequence.py:273-288).  Output records are concatenated in read order.
 * Randomness is counter-based: it depends only on (seed, global read index), where
 * global index = first_read_index + i * read_index_stride for read i of the batch. */
#define TKSMSEQ_MODE_PERFECT 0
#define TKSMSEQ_MODE_BADREAD 1
typedef struct {
    uint64_t seed;
    uint64_t first_read_index;
    uint64_t read_index_stride;   /* 0 is treated as 1 */
    int32_t mode;                 /* TKSMSEQ_MODE_* */
    int32_t fastq;                /* 1: '@id info\nSEQ\n+\nQUAL\n'   0: '>id info\nSEQ\n' */
    int32_t compute_qual;         /* badread only: 0 = all 'K' (--skip-qual-compute) */
    int32_t collect_stats;        /* 1: fill the per-read debug statistics (tests) */
    int32_t perfect_of_badread;   /* badread only: format the badread sequence the way perfect() does (quals 'K',
                                     error_free_length = length, identity 100.00%).  This is what the reference
                                     writes to --perfect when -o is given too (py/sequence.py:317-319 rebinds
                                     `seq`); the CLI uses it to reproduce that behaviour. */
    int32_t reserved;
} tksmseq_run_params;

typedef struct {
    const void* records;          /* device pointer, records_bytes bytes */
    const void* record_offsets;   /* device u64[n_reads + 1] */
    uint64_t records_bytes;
    uint64_t n_reads;
    uint64_t bases_in;            /* error-free bases (sum of spliced lengths) */
    uint64_t bases_out;           /* emitted bases */
    float kernel_ms[8];           /* device time of this call by HIP events on its stream, 0 if timing is off: [0] lengths + scan,
                                     [1] the simulate stage as a whole, [2] record offsets, [3] k_emit / k_perfect, [4] everything;
                                     Badread: summed launch durations of [5] the error-loop kernels (k_loop, k_loopw, the straggler launch), [6] the
                                     alignment kernel k_alnf (all passes), [7] the launches around k_qjobs (rounds 1 - 2: k_job); the rest of [1]:
                                     k_init, the final-stage k_err, host gaps between rounds */
} tksmseq_result;

int tksmseq_run(tksmseq_ctx* ctx, const tksmseq_batch* batch, const tksmseq_run_params* params,
                tksmseq_result* result);
/* Caller-provided device buffer for the record stream (e.g. a torch tensor); NULL restores the
 * library-owned buffer.  tksmseq_run fails with TKSMSEQ_ENOMEM if it is too small. */
int tksmseq_set_output_buffer(tksmseq_ctx* ctx, void* device_ptr, uint64_t capacity);
/* Host threads used to parse MDF text in tksmseq_batch_from_mdf_text (default 1).  The reference's parallel knob is
 * -t/--threads: multiprocessing.Pool(args.threads) over molecules, py/sequence.py:360-368; here the device does the per-molecule
 * work and the threads go to the text -> binary batch conversion. */
int tksmseq_set_host_threads(tksmseq_ctx* ctx, int n);
/* 1 if `name` resolves to a model file of `kind` ("error", "qscore", "tail") in the built-in directory or $TKSM_MODELS
 * (set_tksm_models_dicts, py/sequence.py:17-31): the CLI's default-model rule "nanopore2020 if discoverable else random"
 * (py/sequence.py:86-107). */
int tksmseq_model_available(const char* name, const char* kind);
/* Host only, no context, callable from any thread: parses a model file ("error" / "qscore") or computes the identity quantile
 * table now and keeps the result for the process, so that a later tksmseq_load_*_model / tksmseq_set_identity with the same
 * file / parameters copies it instead of parsing again.  The CLI calls these on threads of their own while the device is set up
 * and the reference packed (the reference's Python loads its models serially at import, py/sequence.py:323-345). */
int tksmseq_prefetch_model(const char* name_or_path, const char* kind);
int tksmseq_prefetch_identity(double mean, double max, double stdev);
int tksmseq_set_timing(tksmseq_ctx* ctx, int enable);   /* hipEvent per stage, read via result.kernel_ms */
/* Diagnostics of the last Badread tksmseq_run on this context (no reference counterpart: the lane-per-alignment kernels fall back
 * to an exact wave-wide kernel for what their fixed-size queues / stored rows cannot hold, with the same results -- so a defect that
 * only shows as fall-backs is invisible to parity tests; tests/test_gpu_parity.py watches these counts instead).  out[16]:
 *  [0] rounds of the host loop            [1] reads finished by the exact wave-wide kernel (non-ACGT bytes, band exits, fall-backs)
 *  [2] predicted stragglers (own stream)   [3] alignment jobs of the rounds whose first pass stores 14 rows
 *  [4] of those, jobs redone at full width (path left the stored rows)       [5] fused-alignment fall-backs (job given up by k_alnf)
 *  [6] the fall-backs' reasons, or-ed (bit 0 column-queue / reservoir overflow, 3 window shift > 31, 4 shift > 14 in the 14-row
 *      pass, 5 end cell outside the band, 6 walk left the stored rows)        [7] fall-backs among q-score jobs
 *  [8] fall-backs in the full-width list pass                                 [9] alignment jobs of all rounds
 *  [10] reads whose alignment left the 64-row band (exact kernel)            [11..15] reserved (0) */
int tksmseq_run_diagnostics(tksmseq_ctx* ctx, uint32_t* out);
/* Copies the last result to host memory (records: records_bytes, offsets: n_reads + 1). */
int tksmseq_result_download(tksmseq_ctx* ctx, uint8_t* records, uint64_t* offsets);
/* A slice [offset, offset + bytes) of the last result's record stream to host memory -- for callers that stream a large result
 * through a small page-locked buffer (tksmseq_host_alloc) instead of holding it whole.  async != 0: returns once the copy is
 * queued on the context's stream (tksmseq_synchronize before the bytes are read). */
int tksmseq_result_download_range(tksmseq_ctx* ctx, uint8_t* dst, uint64_t offset, uint64_t bytes, int async);
/* Device-to-device copies of the last result into caller buffers (either may be NULL): records_bytes bytes and
 * n_reads + 1 u64 offsets.  Asynchronous on the context's stream. */
int tksmseq_result_copy_device(tksmseq_ctx* ctx, void* records_dst, void* offsets_dst);
/* Per-read debug statistics of the last badread run with collect_stats = 1: int32[n_reads][16]
 * {n_draws, change_count, n_aligns, frag_len, new_len, start_trim, end_trim, status, ...} +
 * double[n_reads][2] {errors, target_identity}. */
int tksmseq_stats_download(tksmseq_ctx* ctx, int32_t* istats, double* dstats);

/* ---- multi-GPU record ordering (S7) ----------------------------------------------------------
 * Interleaves P per-rank record streams (rank p holds global reads p, p+P, p+2P, ...) into global
 * read order on the device: dst gets sum(len) bytes.  Used after an RCCL gather on rank 0.
 * streams[p] / offsets[p] are device pointers (u64 offsets[n_p + 1]). */
int tksmseq_interleave_records(tksmseq_ctx* ctx, int n_ranks, const void* const* streams,
                               const void* const* offsets, const uint64_t* n_reads_per_rank,
                               void* dst, uint64_t dst_capacity, uint64_t* dst_bytes);

/* ---- the module entry point ------------------------------------------------------------------
 * int Sequencer_module::run() (src/sequence.cpp:30-54, src/pimpl.h:5-9) with the CLI of
 * py/sequence.py:34-165; argv[0] is "sequence" as in src/tksm.cpp:164-166.  Returns the process
 * exit code (0 ok, 1 argument / input error). */
/* ---- molecule-description transforms upstream of Seq (BASELINE config 5), on the device ----------------------------------
 * Both take a batch and return a new one (free with tksmseq_batch_free) that tksmseq_run accepts directly: the molecule
 * tables never leave the device between PCR, truncation and sequencing.  Molecules are taken depth-unrolled, as every
 * C++ module of the reference reads them (stream_mdf(..., true), src/mdf.h:97-105): copies of a depth > 1 molecule
 * become id_0, id_1, ...  Results depend only on (seed, molecule index), not on batching or the device.
 *
 * tksmseq_pcr replaces PCR::perform / do_pcr (src/pcr.cpp:40-89; module src/pcr.cpp:91-260): every cycle copies each
 * molecule present with probability `efficiency`; a copy gets floor(4/3 error_rate x size) (+1 with the fractional
 * probability) substitutions at distinct positions, bases uniform in "ACTG", on top of its template's; each copy is
 * written with probability target_count / ((1 + efficiency)^cycles x molecules); id = template id + "." + cycle.
 * More than 2 x target_count input molecules: 2 x target_count of them are used, a uniformly random ORDERED subset as :217-220's
 * std::shuffle + resize gives (the molecules with the smallest Philox keys, in key order); their copies are written in that order. */
/* flags of both transforms.  TKSMSEQ_MOL_NO_COMMENTS: the output batch carries no header comments.  Comments ("truncated=...", "TR=...",
 * the template's own) are host-side text per molecule; a caller whose next step is tksmseq_run -- which never reads them, like the
 * reference's Seq (mdf_generator, py/sequence.py:206-213, drops the comment column) -- saves that work: the chained `tksm sequence
 * --pcr-... --truncate-...` sets it, `tksm pcr` / `tksm truncate` (MDF text out) do not. */
#define TKSMSEQ_MOL_NO_COMMENTS 1
typedef struct {
    uint64_t seed;
    uint64_t target_count;        /* --molecule-count */
    int32_t cycles;               /* --cycles (at most 56) */
    int32_t flags;                /* TKSMSEQ_MOL_NO_COMMENTS or 0 */
    double error_rate;            /* --error-rate, before the 4/3 adjustment of src/pcr.cpp:36 */
    double efficiency;            /* --efficiency */
    /* the copies of the templates at positions template_begin <= i < template_end of the PROCESSING ORDER only (0, 0: all of them):
     * input order, or the subsample's key order when more than 2 x target_count molecules come in.  The drop ratio and the subsample
     * are those of the WHOLE input either way, and a copy depends on (seed, its template, its path of cycles) alone, so the outputs
     * of consecutive slices, one after the other, are the output of the whole: how `tksm pcr` streams 200 M molecules through
     * bounded memory and spreads them over several devices */
    uint64_t template_begin, template_end;
} tksmseq_pcr_params;
/* written copies per template, by position in the processing order (counts[n_reads of the batch]; positions beyond the subsample:
 * 0): what a caller needs to number the molecules of a slice's output before the slices before it have been made */
int tksmseq_pcr_template_counts(tksmseq_ctx* ctx, const tksmseq_batch* in, const tksmseq_pcr_params* params, uint64_t* counts);
int tksmseq_pcr_preset(const char* name, double* error_rate, double* efficiency);   /* -x/--preset, src/pcr.cpp:136-140 */
int tksmseq_pcr(tksmseq_ctx* ctx, const tksmseq_batch* in, const tksmseq_pcr_params* params, tksmseq_batch** out);

/* tksmseq_truncate replaces truncate() + truncate_transformer / truncate_transformer_kde (src/truncate.cpp:23-65, :322-351):
 * NORMAL / LOGNORMAL: keep the first (int)X bases in segment order, X ~ Normal(mu, sigma) / exp(Normal(mu, sigma)), at least
 * 100 (min_val); KDE: truncation length from the 2-D model of <kde_model_path> (custom_distribution2D, :163-203), split
 * between the 3' end and the 5' end by the model's end-ratio histogram (or all at the 3' end with always_end when the
 * model has none).  A cut segment keeps its substitutions re-based, sorted by position (einterval::truncate). */
#define TKSMSEQ_TRC_NORMAL 0
#define TKSMSEQ_TRC_LOGNORMAL 1
#define TKSMSEQ_TRC_KDE 2
typedef struct {
    uint64_t seed;
    uint64_t first_molecule_index;   /* index of the batch's first molecule in the whole input (RNG key) */
    int32_t mode;                    /* TKSMSEQ_TRC_* */
    int32_t always_end;              /* --always-end */
    int32_t kde_models_length;       /* --kde-models-length */
    int32_t flags;                   /* TKSMSEQ_MOL_NO_COMMENTS or 0 */
    double mu, sigma;                /* --normal / --lognormal */
    const char* kde_model_path;      /* --kde-model */
} tksmseq_trc_params;
int tksmseq_truncate(tksmseq_ctx* ctx, const tksmseq_batch* in, const tksmseq_trc_params* params, tksmseq_batch** out);

/* The batch as MDF text, the way molecule_descriptor::operator<< writes it (src/interval.h:898-905): "+id<TAB>depth<TAB>comment",
 * then "chr<TAB>start<TAB>end<TAB>strand<TAB>pos<base>,..." per segment; depth 1 per molecule; comments re-serialised key-sorted
 * like dump_comment (:880-890).  *text is malloc'ed: release with tksmseq_text_free. */
int tksmseq_batch_to_mdf_text(tksmseq_ctx* ctx, const tksmseq_batch* b, char** text, uint64_t* len);
void tksmseq_text_free(char* text);

/* The `tksm pcr` and `tksm truncate` modules (PCR_module / Truncate_module: src/pcr.cpp:91-260, src/truncate.cpp:236-451) on
 * top of the functions above: same flags, MDF file in, MDF file out; argv[0] is the module name. */
int tksmseq_pcr_main(int argc, char** argv);
int tksmseq_truncate_main(int argc, char** argv);
int tksmseq_sequence_main(int argc, char** argv);

#ifdef __cplusplus
}
#endif
#endif
