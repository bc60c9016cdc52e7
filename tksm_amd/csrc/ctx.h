// ctx.h -- internal: the context and batch objects behind the opaque handles of include/tksmseq.h (shared by api.cpp and
// mdf_ops.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tksmseq.h"
#include "host.h"
#include "kernels.h"

using namespace tkh;

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return TKSMSEQ_EDEVICE;                                                               \
        }                                                                                         \
    } while (0)

// (diagnostics, TKSMSEQ_VERBOSE: time this thread has spent in hipMalloc)
inline double& alloc_seconds() { static thread_local double v = 0.0; return v; }
inline unsigned& alloc_calls() { static thread_local unsigned v = 0; return v; }

// ---- Cache of device allocations, per process and device, for the buffers that live for ONE batch: the tables of a molecule batch and
// the temporaries of PCR / truncation.  hipFree waits until the whole device is idle -- every stream of every context: ~100 ms while
// three batches are in flight -- and a streaming run frees (and allocates) such buffers once per batch: the workers of `tksm sequence`
// spent a third of their cycle in tksmseq_batch_free (profiles/r04_e2e_stream.log).  A pooled buffer goes back to the cache instead
// (after the stream that used it has drained -- the caller's duty, see DevBuf::pool_stream) and the next request of about its size takes
// it.  Bounded: beyond CACHE_LIMIT bytes per device a block is really freed; everything is freed when the last context goes.
struct DevCache {
    static constexpr int MAX_DEV = 16;
    static constexpr size_t CACHE_LIMIT = 12ull << 30;
    std::mutex m;
    std::multimap<size_t, void*> blocks[MAX_DEV];
    size_t cached[MAX_DEV] = {};
    int live_contexts = 0;
    static DevCache& get() { static DevCache c; return c; }
    // sizes in classes of 1/8 of a power of two, so that consecutive batches (a few percent apart) reuse each other's blocks
    static size_t size_class(size_t bytes) {
        if (bytes < 4096) return 4096;
        int top = 63 - __builtin_clzll((unsigned long long)bytes);
        const size_t step = (size_t)1 << (top > 3 ? top - 3 : 0);
        return (bytes + step - 1) & ~(step - 1);
    }
    void* take(int dev, size_t cap) {                        // a cached block of exactly this class, or nullptr
        if (dev < 0 || dev >= MAX_DEV) return nullptr;
        std::lock_guard<std::mutex> l(m);
        auto it = blocks[dev].find(cap);
        if (it == blocks[dev].end()) return nullptr;
        void* p = it->second;
        blocks[dev].erase(it);
        cached[dev] -= cap;
        return p;
    }
    void give(int dev, void* p, size_t cap) {
        if (dev >= 0 && dev < MAX_DEV) {
            std::lock_guard<std::mutex> l(m);
            if (live_contexts > 0 && cached[dev] + cap <= CACHE_LIMIT) { blocks[dev].emplace(cap, p); cached[dev] += cap; return; }
        }
        (void)hipFree(p);
    }
    void context_created() { std::lock_guard<std::mutex> l(m); live_contexts++; }
    void context_destroyed() {                               // the last one out frees the cache
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> l(m);
            if (--live_contexts > 0) return;
            for (int d = 0; d < MAX_DEV; d++) { for (auto& kv : blocks[d]) drop.push_back(kv.second); blocks[d].clear(); cached[d] = 0; }
        }
        for (void* p : drop) (void)hipFree(p);
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool owned = true;                     // false: a view of another context's buffer (tksmseq_clone), read-only
    // pooled: allocations come from / go back to DevCache.  Whoever lets go of a pooled buffer must know that no queued work still
    // uses it: pool_stream, if set, is drained first (the temporaries of a call); the tables of a batch are released by
    // tksmseq_batch_free, which drains the stream of the context that ran the batch.
    bool pooled = false;
    hipStream_t pool_stream = nullptr;
    int dev = -1;
    void release() {
        if (p && owned) {
            if (pooled) { if (pool_stream) (void)hipStreamSynchronize(pool_stream); DevCache::get().give(dev, p, cap); }
            else (void)hipFree(p);
        }
        p = nullptr; cap = 0;
    }
    ~DevBuf() { release(); }
    void borrow(const DevBuf& o) { release(); p = o.p; cap = o.cap; owned = false; }
    hipError_t ensure(size_t bytes, bool keep = false, hipStream_t s = nullptr) {
        if (bytes <= cap && owned) return hipSuccess;
        // a borrowed buffer is never written: the first write access replaces it by a private copy
        // large work buffers get 1/8 of headroom: consecutive batches of a stream differ by a few percent, and
        // re-allocating tens of GB costs more than a batch
        size_t ncap = std::max(bytes > (64u << 20) ? bytes + bytes / 8 : bytes, owned ? cap + cap / 2 : cap);
        ncap = (ncap + 255) & ~(size_t)255;
        void* np = nullptr;
        int ndev = dev;
        if (pooled) {
            ncap = DevCache::size_class(std::max<size_t>(bytes, 1));
            if (hipGetDevice(&ndev) != hipSuccess) ndev = -1;
            np = DevCache::get().take(ndev, ncap);
        }
        if (!np) {
            const auto t_alloc = std::chrono::steady_clock::now();
            hipError_t e = hipMalloc(&np, ncap);
            alloc_seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_alloc).count();
            alloc_calls()++;
            if (e != hipSuccess) return e;
        }
        if (keep && p && cap) {
            hipError_t e = hipMemcpyAsync(np, p, std::min(cap, ncap), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { if (pooled) DevCache::get().give(ndev, np, ncap); else (void)hipFree(np); return e; }
        }
        const bool was_owned = owned;
        if (was_owned) release(); else { p = nullptr; cap = 0; }
        p = np; cap = ncap; owned = true; dev = ndev;
        return hipSuccess;
    }
    template <class T> T* as() const { return (T*)p; }
};

struct tksmseq_batch {
    uint64_t n_reads = 0, n_intervals = 0, n_mods = 0, n_literals = 0;
    DevBuf reads, intervals, mods, literals, litpool, ids, idpool;
    tksmseq_batch() { for (DevBuf* b : {&reads, &intervals, &mods, &literals, &litpool, &ids, &idpool, &d_dup, &d_order, &d_tail}) b->pooled = true; }
    // what PCR / truncation / the MDF writer need beyond Seq (batches parsed from MDF text only)
    std::vector<uint32_t> h_dup;         // [n_reads] bit 31: copy of a depth > 1 molecule, low bits: its index (unroll naming)
    DevBuf d_dup;
    std::vector<uint32_t> h_comments;    // [n_reads][2] header comment of every read's molecule
    std::vector<char> h_comment_pool;
    std::vector<uint32_t> raw_len;       // host copy, for sizing
    std::vector<uint32_t> order;         // reads sorted by raw length (bucketed k_err launches)
    DevBuf d_order;
    uint32_t max_raw = 0;
    uint64_t total_raw = 0;
    // tail noise: raw_len / max_raw / order describe splice + tail while a tail model applies to the run
    std::vector<uint32_t> splice_len;    // the spliced lengths (filled the first time a tail is added)
    DevBuf d_tail;                       // [n_reads] tail lengths of the run keyed below
    bool tail_on = false;
    uint64_t tail_key[4] = {};           // seed, first read index, stride, model version
    // cached scratch sizing, keyed by (k, cap_num, cap_den, cap_add)
    int cache_k = -1, cache_num = 0, cache_den = 0, cache_add = 0;
    uint64_t cache_scratch = 0;
};

struct tksmseq_ctx : ContigLookup {
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    // wave-wide kernel for the reads that cannot take the fast pipeline, underneath the rounds: a few streams, each with
    // its own slice of the per-wave trace buffer, so that launches for reads found in different rounds overlap
    static constexpr int N_SIDE = 4, SIDE_WAVES = 512;
    hipStream_t side[N_SIDE] = {};
    hipEvent_t side_start = nullptr, side_done[N_SIDE] = {};
    bool side_used[N_SIDE] = {};
    // predicted stragglers: their straggler-kernel launch runs on a stream of its own from round 0 on (api.cpp)
    hipStream_t early_stream = nullptr;
    hipEvent_t early_start = nullptr, early_done = nullptr;
    // the wide alignment passes on a stream of their own priority (TKSMSEQ_ALN_STREAM_PRIORITY; api.cpp)
    hipStream_t aln_stream = nullptr;
    hipEvent_t aln_start = nullptr, aln_done = nullptr;
    int aln_prio_set = 0, aln_prio = 0;
    uint32_t early_tail = 1024;           // at most this many reads (and only from a batch with a long tail of predicted visits); 0: never
    DevBuf f_early;
    bool own_stream = false;
    std::string err;

    // reference
    std::vector<std::string> contig_names;
    std::unordered_map<std::string, int> contig_index;
    std::vector<uint64_t> contigs;        // {gstart, len}
    uint64_t total_alloc = 0;             // bases allocated in the global coordinate (block aligned)
    uint64_t total_bases = 0;
    uint32_t pool_blocks = 0;
    DevBuf d_packed, d_blocktab, d_pool, d_contigs, d_stage;

    // models
    ErrorModelHost em; QScoreModelHost qm; IdentityHost idm;
    bool em_uniform = false, em_alt0 = false;
    TailModelHost tail; uint64_t tail_version = 0;
    DevBuf d_tail_lx, d_tail_ly, d_tail_cdf, d_tail_chain;
    DevBuf d_pself, d_pseg, d_pt0, d_cdf32, d_cdf, d_alts, d_altenc, d_nalts, d_qkeys, d_qoff, d_qcnt, d_qcdf, d_qq, d_qtab, d_qent, d_qpairs, d_qguide;

    // per-run work buffers
    DevBuf w_rawlen, w_slotcap, w_slotoff, w_outlen, w_ident, w_reclen, w_recoff, w_status, w_scan, w_trace, w_counter,
        w_scratch, w_records, w_istats, w_dstats, w_sums, w_fullpool, w_biglist, w_bigscratch, w_bigtrace;
    unsigned long long full_pool_bytes = 1ull << 30;
    // fast Badread pipeline state (see kernels.h FastBuffers)
    DevBuf f_state, f_frag, f_nb, f_fplanes, f_jmeta[2], f_redo, f_jpopd[2], f_trace, f_tracefull, f_slow, f_defer, f_defercnt, f_frag2, f_row64;
    std::vector<uint32_t> h_row64;        // host copy of the state-row offsets of the current run
    // what the host and the device exchange in every round, in one copy each way: f_round = {job counts of the even rounds,
    // counters, job counts of the odd rounds} -> h_round; h_geo = {prefix, bases, range geometry} -> f_geoall (page-locked)
    DevBuf f_round, f_geoall;
    uint32_t* h_round = nullptr; size_t h_round_bytes = 0;
    uint8_t* h_geo = nullptr; size_t h_geo_bytes = 0;
    bool force_slow = false;
    uint32_t tail_cut = 0;   // > 0: hand the last reads of a batch to the wave-wide kernel (diagnostic)
    int tail_wcap = 2048;                 // columns of a re-estimation window the straggler kernel aligns itself (its LDS holds 2048; smaller: tests)
    uint32_t tail_wave = 4096;            // rounds with at most this many reads left: one launch that runs every remaining visit of a read on a wave of its own (k_loopw<true>); 0: never
    uint32_t wave_loop = 16384;           // rounds with at most this many reads left run the error loop one wave per read (k_loopw)
    unsigned aln_lds_pad = 0;             // LDS the first alignment pass asks for without using it: caps its waves per CU (kernels.hip launch_aln)
    uint32_t small_round = 16384, small_aln = 131072;   // rounds with fewer reads are latency-bound: merged launches, full-width alignment
    uint32_t n_buckets = 16;
    int defer_len = 0;          // reads longer than this align for their q-scores after the regular rounds, all together
    int hbm_state_len = 2304;   // fragments longer than this are edited in HBM instead of being staged in LDS every round
    std::vector<hipEvent_t> evpool;
    uint32_t last_rounds = 0, last_slow = 0, last_early = 0;
    uint32_t last_diag[16] = {};      // tksmseq_run_diagnostics
    void* user_out = nullptr; uint64_t user_out_cap = 0;
    bool timing = false;
    int host_threads = 1;       // host threads for MDF parsing (tksmseq_set_host_threads)
    hipEvent_t ev[6] = {};
    tksmseq_result last{};
    bool have_last = false, have_stats = false;

    int find(const std::string& name) const override {
        auto it = contig_index.find(name);
        return it == contig_index.end() ? -1 : it->second;
    }
};

// a batch whose tables were written on the device (PCR, truncation): lengths, sorted order and sizing on the host
int finalize_device_batch(tksmseq_ctx* ctx, tksmseq_batch* b);
