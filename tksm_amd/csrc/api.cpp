// api.cpp -- implementation of the C-ABI in include/tksmseq.h (host orchestration of the HIP path).
// No CPU fallback exists here: every compute call launches the kernels of kernels.hip.
#include "../../include/tksmseq.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>

#include "ctx.h"

static thread_local std::string g_create_error;


template <class T>
static int upload(tksmseq_ctx* ctx, DevBuf& b, const std::vector<T>& v) {
    HIPCHK(ctx, b.ensure(v.size() * sizeof(T) + 16));
    if (!v.empty()) HIPCHK(ctx, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

extern "C" {

const char* tksmseq_version(void) { return "tksm-amd seq 0.1 (gfx950)"; }

int tksmseq_create(int device, tksmseq_ctx** out) {
    if (!out) return TKSMSEQ_EINVAL;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = "no HIP device available (this library has no CPU fallback)"; return TKSMSEQ_EDEVICE; }
    if (device < 0 || device >= n) { g_create_error = "device index out of range"; return TKSMSEQ_EINVAL; }
    e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return TKSMSEQ_EDEVICE; }
    std::unique_ptr<tksmseq_ctx> c(new tksmseq_ctx());
    c->device = device;
    if (const char* fs = getenv("TKSMSEQ_FORCE_SLOW")) c->force_slow = fs[0] == '1';
    if (const char* tc = getenv("TKSMSEQ_TAIL_CUT")) c->tail_cut = (uint32_t)atoi(tc);
    if (const char* tc = getenv("TKSMSEQ_SMALL_ROUND")) c->small_round = (uint32_t)atoi(tc);
    if (const char* tc = getenv("TKSMSEQ_SMALL_ALN")) c->small_aln = (uint32_t)atoi(tc);
    if (const char* tc = getenv("TKSMSEQ_WAVE_LOOP")) c->wave_loop = (uint32_t)atoi(tc);
    if (const char* tc = getenv("TKSMSEQ_TAIL_WAVE")) c->tail_wave = (uint32_t)atoi(tc);
    if (const char* tc = getenv("TKSMSEQ_TAIL_WCAP")) c->tail_wcap = atoi(tc);
    if (const char* tc = getenv("TKSMSEQ_EARLY_TAIL")) c->early_tail = (uint32_t)std::min(4096, std::max(0, atoi(tc)));
    if (const char* tc = getenv("TKSMSEQ_ALN_STREAM_PRIORITY")) { c->aln_prio_set = 1; c->aln_prio = atoi(tc); }
    if (const char* tc = getenv("TKSMSEQ_ALN_LDS_PAD")) c->aln_lds_pad = (unsigned)std::min(60000, std::max(0, atoi(tc)));
    if (const char* hl = getenv("TKSMSEQ_HBM_STATE_LEN")) c->hbm_state_len = atoi(hl);
    if (const char* dl = getenv("TKSMSEQ_DEFER_LEN")) c->defer_len = atoi(dl);
    if (const char* fp = getenv("TKSMSEQ_FULL_POOL_MB")) c->full_pool_bytes = (unsigned long long)atoll(fp) << 20;
    if (const char* nbk = getenv("TKSMSEQ_BUCKETS")) c->n_buckets = (uint32_t)std::max(1, atoi(nbk));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cus = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return TKSMSEQ_EDEVICE; }
    c->own_stream = true;
    for (auto& ev : c->ev) (void)hipEventCreate(&ev);
    *out = c.release();
    DevCache::get().context_created();
    return TKSMSEQ_OK;
}

int tksmseq_clone(const tksmseq_ctx* src, tksmseq_ctx** out) {
    if (!src || !out) return TKSMSEQ_EINVAL;
    int rc = tksmseq_create(src->device, out);
    if (rc != TKSMSEQ_OK) return rc;
    tksmseq_ctx* c = *out;
    c->contig_names = src->contig_names; c->contig_index = src->contig_index; c->contigs = src->contigs;
    c->total_alloc = src->total_alloc; c->total_bases = src->total_bases; c->pool_blocks = src->pool_blocks;
    c->d_packed.borrow(src->d_packed); c->d_blocktab.borrow(src->d_blocktab); c->d_pool.borrow(src->d_pool); c->d_contigs.borrow(src->d_contigs);
    c->em = src->em; c->qm = src->qm; c->idm = src->idm; c->em_uniform = src->em_uniform; c->em_alt0 = src->em_alt0;
    c->d_pself.borrow(src->d_pself); c->d_pseg.borrow(src->d_pseg); c->d_pt0.borrow(src->d_pt0); c->d_cdf32.borrow(src->d_cdf32); c->d_cdf.borrow(src->d_cdf); c->d_alts.borrow(src->d_alts); c->d_altenc.borrow(src->d_altenc);
    c->d_nalts.borrow(src->d_nalts); c->d_qkeys.borrow(src->d_qkeys); c->d_qoff.borrow(src->d_qoff); c->d_qcnt.borrow(src->d_qcnt);
    c->d_qcdf.borrow(src->d_qcdf); c->d_qq.borrow(src->d_qq); c->d_qtab.borrow(src->d_qtab); c->d_qent.borrow(src->d_qent);
    c->d_qpairs.borrow(src->d_qpairs); c->d_qguide.borrow(src->d_qguide);
    c->tail = src->tail; c->tail_version = src->tail_version; c->host_threads = src->host_threads;
    c->d_tail_lx.borrow(src->d_tail_lx); c->d_tail_ly.borrow(src->d_tail_ly); c->d_tail_cdf.borrow(src->d_tail_cdf); c->d_tail_chain.borrow(src->d_tail_chain);
    return TKSMSEQ_OK;
}

int tksmseq_host_alloc(uint64_t bytes, void** out) {
    if (!out) return TKSMSEQ_EINVAL;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? TKSMSEQ_OK : TKSMSEQ_ENOMEM;
}

void tksmseq_host_free(void* p) { if (p) (void)hipHostFree(p); }

int tksmseq_device_alloc(tksmseq_ctx* ctx, uint64_t bytes, void** out) {
    if (!ctx || !out) return TKSMSEQ_EINVAL;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(out, bytes ? bytes : 1));
    return TKSMSEQ_OK;
}
void tksmseq_device_free(tksmseq_ctx* ctx, void* p) { if (ctx && p) { (void)hipSetDevice(ctx->device); (void)hipFree(p); } }
int tksmseq_copy_to_host(tksmseq_ctx* ctx, void* dst_host, const void* src_device, uint64_t bytes, int async) {
    if (!ctx || ((!dst_host || !src_device) && bytes)) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (!async) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

void tksmseq_destroy(tksmseq_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->evpool) (void)hipEventDestroy(ev);
    for (auto& st : ctx->side) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (ctx->early_stream) { (void)hipStreamSynchronize(ctx->early_stream); (void)hipStreamDestroy(ctx->early_stream); }
    if (ctx->aln_stream) { (void)hipStreamSynchronize(ctx->aln_stream); (void)hipStreamDestroy(ctx->aln_stream); }
    if (ctx->aln_start) (void)hipEventDestroy(ctx->aln_start);
    if (ctx->aln_done) (void)hipEventDestroy(ctx->aln_done);
    if (ctx->early_start) (void)hipEventDestroy(ctx->early_start);
    if (ctx->early_done) (void)hipEventDestroy(ctx->early_done);
    for (auto& ev : ctx->side_done) if (ev) (void)hipEventDestroy(ev);
    if (ctx->side_start) (void)hipEventDestroy(ctx->side_start);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->h_round) (void)hipHostFree(ctx->h_round);
    if (ctx->h_geo) (void)hipHostFree(ctx->h_geo);
    delete ctx;
    DevCache::get().context_destroyed();                     // (the last context of the process frees the cached batch buffers)
}

const char* tksmseq_last_error(const tksmseq_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int tksmseq_set_stream(tksmseq_ctx* ctx, void* s) {
    if (!ctx) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (s) { ctx->stream = (hipStream_t)s; ctx->own_stream = false; }
    else { HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    return TKSMSEQ_OK;
}

int tksmseq_synchronize(tksmseq_ctx* ctx) {
    if (!ctx) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

int tksmseq_set_host_threads(tksmseq_ctx* ctx, int n) { if (!ctx || n < 1) return TKSMSEQ_EINVAL; ctx->host_threads = std::min(n, 64); return TKSMSEQ_OK; }

int tksmseq_model_available(const char* name, const char* kind) { return (name && kind && model_available(name, kind)) ? 1 : 0; }

int tksmseq_set_timing(tksmseq_ctx* ctx, int enable) { if (!ctx) return TKSMSEQ_EINVAL; ctx->timing = enable != 0; return TKSMSEQ_OK; }

// ------------------------------------------------------------------------------------------- reference
int tksmseq_reference_add_contig(tksmseq_ctx* ctx, const char* name, const uint8_t* ascii, uint64_t len, int on_device) {
    if (!ctx || !name || (!ascii && len)) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t BLK = 1ull << tk::BLOCK_SHIFT;
    const uint64_t gstart = ctx->total_alloc;
    const uint64_t nblk = (len + BLK - 1) / BLK;
    if (gstart + nblk * BLK > (1ull << 44)) { ctx->err = "reference larger than 2^44 bases"; return TKSMSEQ_ELIMIT; }
    HIPCHK(ctx, ctx->d_packed.ensure(((gstart + nblk * BLK) >> 4) * 4 + 64, true, ctx->stream));
    HIPCHK(ctx, ctx->d_blocktab.ensure(((gstart >> tk::BLOCK_SHIFT) + nblk) * 4 + 64, true, ctx->stream));
    uint32_t* blocktab = ctx->d_blocktab.as<uint32_t>() + (gstart >> tk::BLOCK_SHIFT);
    const uint64_t CH = 64ull << 20;   // staging chunk (multiple of the block size)
    std::vector<uint32_t> flags;
    for (uint64_t off = 0; off < len; off += CH) {
        const uint64_t n = std::min(CH, len - off);
        const uint64_t cb = (n + BLK - 1) / BLK;
        const uint8_t* dsrc;
        if (on_device) dsrc = ascii + off;
        else {
            HIPCHK(ctx, ctx->d_stage.ensure(CH));
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage.p, ascii + off, n, hipMemcpyHostToDevice, ctx->stream));
            dsrc = ctx->d_stage.as<uint8_t>();
        }
        uint32_t* bt = blocktab + (off >> tk::BLOCK_SHIFT);
        HIPCHK(ctx, hipMemsetAsync(bt, 0, cb * 4, ctx->stream));
        HIPCHK(ctx, tk::launch_pack(dsrc, n, gstart + off, ctx->d_packed.as<uint32_t>(), ctx->d_blocktab.as<uint32_t>(), ctx->stream));
        flags.resize(cb);
        HIPCHK(ctx, hipMemcpyAsync(flags.data(), bt, cb * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t newblocks = 0;
        for (auto& f : flags) { if (f) { f = ctx->pool_blocks + newblocks; newblocks++; } else f = tk::NO_BLOCK; }
        HIPCHK(ctx, hipMemcpyAsync(bt, flags.data(), cb * 4, hipMemcpyHostToDevice, ctx->stream));
        if (newblocks) {
            HIPCHK(ctx, ctx->d_pool.ensure((uint64_t)(ctx->pool_blocks + newblocks) * BLK, true, ctx->stream));
            HIPCHK(ctx, tk::launch_fill_pool(dsrc, n, gstart + off, ctx->d_blocktab.as<uint32_t>(), ctx->d_pool.as<uint8_t>(), ctx->stream));
            ctx->pool_blocks += newblocks;
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    // later contigs with the same name replace earlier ones (dict.update, py/sequence.py:193)
    std::string nm(name);
    auto it = ctx->contig_index.find(nm);
    if (it == ctx->contig_index.end()) {
        ctx->contig_index[nm] = (int)ctx->contig_names.size();
        ctx->contig_names.push_back(nm);
        ctx->contigs.push_back(gstart); ctx->contigs.push_back(len);
    } else {
        ctx->total_bases -= ctx->contigs[2 * it->second + 1];
        ctx->contigs[2 * it->second] = gstart; ctx->contigs[2 * it->second + 1] = len;
    }
    ctx->total_alloc = gstart + nblk * BLK;
    ctx->total_bases += len;
    HIPCHK(ctx, ctx->d_contigs.ensure(ctx->contigs.size() * 8 + 16));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_contigs.p, ctx->contigs.data(), ctx->contigs.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

int tksmseq_reference_add_fasta(tksmseq_ctx* ctx, const char* path) {
    if (!ctx || !path) return TKSMSEQ_EINVAL;
    std::vector<FastaRecord> recs;
    if (!read_fasta(path, recs, ctx->err)) return TKSMSEQ_EIO;
    for (auto& r : recs) {
        int rc = tksmseq_reference_add_contig(ctx, r.name.c_str(), (const uint8_t*)r.seq.data(), r.seq.size(), 0);
        if (rc) return rc;
    }
    return TKSMSEQ_OK;
}

int tksmseq_reference_contig_id(const tksmseq_ctx* ctx, const char* name) { return (!ctx || !name) ? -1 : ctx->find(name); }

int tksmseq_reference_info(const tksmseq_ctx* ctx, uint64_t* n_contigs, uint64_t* total_bases, uint64_t* device_bytes) {
    if (!ctx) return TKSMSEQ_EINVAL;
    if (n_contigs) *n_contigs = ctx->contig_names.size();
    if (total_bases) *total_bases = ctx->total_bases;
    if (device_bytes) *device_bytes = (ctx->total_alloc >> 2) + (ctx->total_alloc >> tk::BLOCK_SHIFT) * 4 + ((uint64_t)ctx->pool_blocks << tk::BLOCK_SHIFT);
    return TKSMSEQ_OK;
}

// ------------------------------------------------------------------------------------------- models
int tksmseq_load_error_model(tksmseq_ctx* ctx, const char* name_or_path) {
    if (!ctx || !name_or_path) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ErrorModelHost m;
    if (!load_error_model(name_or_path, m, ctx->err)) return TKSMSEQ_EIO;
    ctx->em = std::move(m);
    ctx->em_alt0 = ctx->em.type == 1;
    for (size_t i = 0; i < ctx->em.nalts.size() && ctx->em_alt0; i++)
        if (ctx->em.nalts[i] && !(ctx->em.alts[i * (size_t)ctx->em.max_alts] >> 63)) ctx->em_alt0 = false;
    ctx->em_uniform = ctx->em.type == 1;
    for (uint8_t v : ctx->em.nalts) if ((int)v != ctx->em.max_alts) { ctx->em_uniform = false; break; }
    int rc;
    if ((rc = upload(ctx, ctx->d_cdf, ctx->em.cdf))) return rc;
    {
        if (ctx->em.max_alts > 32) { ctx->err = "error models with more than 32 alternatives per k-mer are not supported"; return TKSMSEQ_ELIMIT; }
        const size_t nk = ctx->em.nalts.size(), A = (size_t)ctx->em.max_alts;
        std::vector<uint32_t> c32(nk * 32, 0xFFFFFFFFu);
        for (size_t i = 0; i < nk; i++) for (size_t a = 0; a < A; a++) c32[i * 32 + a] = ctx->em.cdf[i * A + a];
        if (ctx->em.type == 0) std::fill(c32.begin(), c32.end(), 0u);
        if ((rc = upload(ctx, ctx->d_cdf32, c32))) return rc;
        std::vector<uint32_t> ps(nk * 2);
        for (size_t i = 0; i < nk; i++) { ps[2 * i] = c32[i * 32]; ps[2 * i + 1] = ctx->em.nalts[i] ? c32[i * 32 + ctx->em.nalts[i] - 1] : 0u; }
        if ((rc = upload(ctx, ctx->d_pself, ps))) return rc;
        std::vector<uint32_t> sg(nk * 4);                     // thresholds 0, 8, 16, 24 of every row (kernels.h ErrModelView::pseg)
        for (size_t i = 0; i < nk; i++) for (int q = 0; q < 4; q++) sg[4 * i + q] = c32[i * 32 + 8 * q];
        if ((rc = upload(ctx, ctx->d_pseg, sg))) return rc;
        std::vector<uint32_t> t0(nk);
        for (size_t i = 0; i < nk; i++) t0[i] = c32[i * 32];
        if ((rc = upload(ctx, ctx->d_pt0, t0))) return rc;
    }
    if ((rc = upload(ctx, ctx->d_alts, ctx->em.alts))) return rc;
    {
        // the alternatives once more, as the fast pipeline applies them (kernels.h ErrModelView::alts_enc): per slot the
        // 16-bit encoding it would write, bit 15 telling whether the slot differs from the k-mer's own base
        const size_t nk = ctx->em.nalts.size(), A = (size_t)ctx->em.max_alts;
        const int k = ctx->em.k;
        std::vector<uint32_t> enc(nk * A * 4, 0u);
        for (size_t i = 0; i < nk && ctx->em.type == 1; i++)
            for (size_t a = 0; a < A; a++) {
                const uint64_t alt = ctx->em.alts[i * A + a];
                uint32_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int boff = 0;
                for (int j = 0; j < k; j++) {
                    const uint32_t kc = (uint32_t)(i >> (2 * (k - 1 - j))) & 3u;
                    if (alt >> 63) { e[j] = (1u << 12) | (kc & 1u) | ((kc >> 1) << 5); continue; }          // the k-mer itself
                    const uint32_t len = (uint32_t)(alt >> (3 * j)) & 7u;
                    const uint32_t codes = (uint32_t)(alt >> (24 + 2 * boff)) & ((1u << (2 * len)) - 1u);
                    boff += (int)len;
                    // symbols planar: low bits of the (up to 5) symbols in bits 0..4, high bits in bits 5..9 (what k_job queues)
                    uint32_t planar = 0;
                    for (uint32_t x = 0; x < len && x < 5; x++) planar |= (((codes >> (2 * x)) & 1u) << x) | (((codes >> (2 * x + 1)) & 1u) << (5 + x));
                    e[j] = ((len == 1 && codes == kc) ? 0u : 0x8000u) | (len << 12) | planar;
                }
                for (int q = 0; q < 4; q++) enc[(i * A + a) * 4 + q] = e[2 * q] | (e[2 * q + 1] << 16);
            }
        if ((rc = upload(ctx, ctx->d_altenc, enc))) return rc;
    }
    return upload(ctx, ctx->d_nalts, ctx->em.nalts);
}

int tksmseq_load_qscore_model(tksmseq_ctx* ctx, const char* name_or_path) {
    if (!ctx || !name_or_path) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    QScoreModelHost m;
    if (!load_qscore_model(name_or_path, m, ctx->err)) return TKSMSEQ_EIO;
    ctx->qm = std::move(m);
    int rc;
    if ((rc = upload(ctx, ctx->d_qkeys, ctx->qm.keys))) return rc;
    if ((rc = upload(ctx, ctx->d_qoff, ctx->qm.row_off))) return rc;
    if ((rc = upload(ctx, ctx->d_qcnt, ctx->qm.row_cnt))) return rc;
    if ((rc = upload(ctx, ctx->d_qcdf, ctx->qm.cdf_pool))) return rc;
    if ((rc = upload(ctx, ctx->d_qq, ctx->qm.q_pool))) return rc;
    {
        const QScoreModelHost& m2 = ctx->qm;
        const size_t ns = (size_t)m2.n_slots;
        std::vector<uint32_t> ent(ns * 4, 0), pairs(m2.q_pool.size() * 2, 0);
        std::vector<uint8_t> guide(ns * 64, 0);
        // (the flag bit needs candidate indices and q values below 128: every model we know of; otherwise the plain guide)
        bool direct = true;
        for (size_t s2 = 0; s2 < ns; s2++) if (m2.keys[s2] && m2.row_cnt[s2] > 128) direct = false;
        for (uint8_t qv : m2.q_pool) if (qv > 127) direct = false;
        for (size_t i = 0; i < m2.q_pool.size(); i++) { pairs[2 * i] = m2.cdf_pool[i]; pairs[2 * i + 1] = m2.q_pool[i]; }
        for (size_t s2 = 0; s2 < ns; s2++) {
            ent[4 * s2] = (uint32_t)m2.keys[s2]; ent[4 * s2 + 1] = (uint32_t)(m2.keys[s2] >> 32);
            ent[4 * s2 + 2] = m2.row_off[s2]; ent[4 * s2 + 3] = m2.row_cnt[s2];
            if (!m2.keys[s2]) continue;
            const uint32_t off = m2.row_off[s2], cnt = m2.row_cnt[s2];
            if (cnt > 255) { ctx->err = "q-score rows with more than 255 entries are not supported"; return TKSMSEQ_ELIMIT; }
            uint32_t a = 0;
            for (uint32_t bkt = 0; bkt < 64; bkt++) {
                // entries whose threshold is <= the smallest draw of the bucket can never be chosen in it
                const uint32_t wmin = bkt << 26, wmax = wmin | 0x3ffffffu;
                while (a + 1 < cnt && m2.cdf_pool[off + a] <= wmin) a++;
                // ... and if the largest draw of the bucket stops at the same entry, the bucket IS that entry's q: no row read
                uint32_t ah = a;
                while (ah + 1 < cnt && m2.cdf_pool[off + ah] <= wmax) ah++;
                guide[s2 * 64 + bkt] = (direct && ah == a) ? (uint8_t)(0x80u | m2.q_pool[off + a]) : (uint8_t)a;
            }
        }
        ctx->qm.guide_direct = direct;
        if ((rc = upload(ctx, ctx->d_qent, ent))) return rc;
        if ((rc = upload(ctx, ctx->d_qpairs, pairs))) return rc;
        return upload(ctx, ctx->d_qguide, guide);
    }
}

static int install_tail_model(tksmseq_ctx* ctx, TailModelHost&& m) {
    ctx->tail = std::move(m);
    ctx->tail_version++;
    if (!ctx->tail.enabled) return TKSMSEQ_OK;
    int rc;
    if ((rc = upload(ctx, ctx->d_tail_lx, ctx->tail.lx))) return rc;
    if ((rc = upload(ctx, ctx->d_tail_ly, ctx->tail.ly))) return rc;
    if ((rc = upload(ctx, ctx->d_tail_cdf, ctx->tail.cdf))) return rc;
    std::vector<tk::TailChain> ch(1);
    memcpy(ch[0].cum, ctx->tail.cum, sizeof(ch[0].cum));
    ch[0].bases = (uint32_t)ctx->tail.bases[0] | ((uint32_t)ctx->tail.bases[1] << 8) | ((uint32_t)ctx->tail.bases[2] << 16) | ((uint32_t)ctx->tail.bases[3] << 24);
    ch[0].pad = 0;
    return upload(ctx, ctx->d_tail_chain, ch);
}

int tksmseq_load_tail_model(tksmseq_ctx* ctx, const char* name_or_path) {
    if (!ctx || !name_or_path) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TailModelHost m;
    if (!load_tail_model(name_or_path, m, ctx->err)) return TKSMSEQ_EINVAL;
    return install_tail_model(ctx, std::move(m));
}

int tksmseq_set_tail_model(tksmseq_ctx* ctx, const tksmseq_tail_model* t) {
    if (!ctx) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TailModelHost m;
    if (t) {
        if (!t->lx || !t->ly || !t->grid) { ctx->err = "tail model: null table"; return TKSMSEQ_EINVAL; }
        if (!make_tail_model(t->lx, t->n_lx, t->ly, t->n_ly, t->grid, t->trans, t->ratio, t->bases, m, ctx->err)) return TKSMSEQ_EINVAL;
    }
    return install_tail_model(ctx, std::move(m));
}

int tksmseq_set_identity(tksmseq_ctx* ctx, double mean, double max, double stdev) {
    if (!ctx) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    IdentityHost id;
    if (!make_identity(mean, max, stdev, id, ctx->err)) return TKSMSEQ_EINVAL;
    id.set = true;
    ctx->idm = std::move(id);
    if (!ctx->idm.constant) return upload(ctx, ctx->d_qtab, ctx->idm.qtab);
    return TKSMSEQ_OK;
}

int tksmseq_prefetch_model(const char* name_or_path, const char* kind) {
    if (!name_or_path || !kind) return TKSMSEQ_EINVAL;
    std::string err;
    // (the loader's message: tksmseq_last_error(NULL) on the calling thread)
    if (!strcmp(kind, "error")) { ErrorModelHost m; if (load_error_model(name_or_path, m, err)) return TKSMSEQ_OK; g_create_error = err; return TKSMSEQ_EIO; }
    if (!strcmp(kind, "qscore")) { QScoreModelHost m; if (load_qscore_model(name_or_path, m, err)) return TKSMSEQ_OK; g_create_error = err; return TKSMSEQ_EIO; }
    g_create_error = std::string("unknown model kind: ") + kind;
    return TKSMSEQ_EINVAL;
}

int tksmseq_prefetch_identity(double mean, double max, double stdev) {
    IdentityHost id; std::string err;
    return make_identity(mean, max, stdev, id, err) ? TKSMSEQ_OK : TKSMSEQ_EINVAL;
}

int tksmseq_get_error_model(const tksmseq_ctx* ctx, int32_t* type, int32_t* k, int32_t* max_alts, uint32_t* cdf, uint64_t* alts, uint8_t* nalts) {
    if (!ctx || ctx->em.type < 0) return TKSMSEQ_ESTATE;
    if (type) *type = ctx->em.type;
    if (k) *k = ctx->em.k;
    if (max_alts) *max_alts = ctx->em.max_alts;
    if (cdf) memcpy(cdf, ctx->em.cdf.data(), ctx->em.cdf.size() * 4);
    if (alts) memcpy(alts, ctx->em.alts.data(), ctx->em.alts.size() * 8);
    if (nalts) memcpy(nalts, ctx->em.nalts.data(), ctx->em.nalts.size());
    return TKSMSEQ_OK;
}

int tksmseq_get_qscore_model(const tksmseq_ctx* ctx, int32_t* n_slots, int32_t* kmer_size, uint64_t* pool_len, uint64_t* keys,
                             uint32_t* row_off, uint32_t* row_cnt, uint32_t* cdf_pool, uint8_t* q_pool) {
    if (!ctx || ctx->qm.n_slots == 0) return TKSMSEQ_ESTATE;
    if (n_slots) *n_slots = ctx->qm.n_slots;
    if (kmer_size) *kmer_size = ctx->qm.kmer_size;
    if (pool_len) *pool_len = ctx->qm.q_pool.size();
    if (keys) memcpy(keys, ctx->qm.keys.data(), ctx->qm.keys.size() * 8);
    if (row_off) memcpy(row_off, ctx->qm.row_off.data(), ctx->qm.row_off.size() * 4);
    if (row_cnt) memcpy(row_cnt, ctx->qm.row_cnt.data(), ctx->qm.row_cnt.size() * 4);
    if (cdf_pool) memcpy(cdf_pool, ctx->qm.cdf_pool.data(), ctx->qm.cdf_pool.size() * 4);
    if (q_pool) memcpy(q_pool, ctx->qm.q_pool.data(), ctx->qm.q_pool.size());
    return TKSMSEQ_OK;
}

int tksmseq_get_identity(const tksmseq_ctx* ctx, int32_t* constant, double* value, double* beta_a, double* beta_b, double* qtab) {
    if (!ctx || !ctx->idm.set) return TKSMSEQ_ESTATE;
    if (constant) *constant = ctx->idm.constant ? 1 : 0;
    if (value) *value = ctx->idm.value;
    if (beta_a) *beta_a = ctx->idm.beta_a;
    if (beta_b) *beta_b = ctx->idm.beta_b;
    if (qtab && !ctx->idm.constant) memcpy(qtab, ctx->idm.qtab.data(), 65537 * sizeof(double));
    return TKSMSEQ_OK;
}

// ------------------------------------------------------------------------------------------- batches
static int verbose_level() { const char* v = getenv("TKSMSEQ_VERBOSE"); return v ? std::max(1, atoi(v)) : 0; }
static int batch_from_host(tksmseq_ctx* ctx, const tksmseq_batch_desc* d, tksmseq_batch** out, bool check_mods = true) {
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (d->n_intervals >= 0x7fffffffull || d->n_mods >= 0x7fffffffull || d->n_reads >= 0xffffffffull) {
        ctx->err = "batch too large (split it: < 2^31 intervals/mods per batch)"; return TKSMSEQ_ELIMIT;
    }
    std::unique_ptr<tksmseq_batch> b(new tksmseq_batch());
    b->n_reads = d->n_reads; b->n_intervals = d->n_intervals; b->n_mods = d->n_mods; b->n_literals = d->n_literals;
    // validate + host-side raw lengths (python slice clamp, py/sequence.py:307)
    std::vector<uint32_t> ilen(d->n_intervals);
    const uint64_t nc = ctx->contig_names.size();
    uint32_t prev_mod = 0;
    for (uint64_t i = 0; i < d->n_intervals; i++) {
        const uint32_t* iv = d->intervals + 4 * i;
        uint64_t clen;
        if (iv[0] >> 31) {
            uint32_t li = iv[0] & 0x7fffffffu;
            if (li >= d->n_literals) { ctx->err = "interval refers to a literal that does not exist"; return TKSMSEQ_EINVAL; }
            clen = d->literals[2 * li + 1];
            if (d->literals[2 * li] + clen > d->literal_bytes) { ctx->err = "literal outside the literal pool"; return TKSMSEQ_EINVAL; }
        } else {
            if (iv[0] >= nc) { ctx->err = "interval refers to a contig that does not exist"; return TKSMSEQ_EINVAL; }
            clen = ctx->contigs[2 * (uint64_t)iv[0] + 1];
        }
        const uint64_t s = std::min<uint64_t>(iv[1], clen), e = std::min<uint64_t>(iv[2], clen);
        ilen[i] = e > s ? (uint32_t)(e - s) : 0;
        const uint32_t mb = iv[3] & 0x7fffffffu;
        if (mb < prev_mod || mb > d->n_mods) { ctx->err = "interval modification offsets are not monotone"; return TKSMSEQ_EINVAL; }
        prev_mod = mb;
    }
    // the reference raises IndexError for a modification outside its slice (py/sequence.py:238)
    for (uint64_t i = 0; i < d->n_intervals && check_mods; i++) {
        const uint32_t mb = d->intervals[4 * i + 3] & 0x7fffffffu;
        const uint32_t me = i + 1 < d->n_intervals ? (d->intervals[4 * (i + 1) + 3] & 0x7fffffffu) : (uint32_t)d->n_mods;
        for (uint32_t m = mb; m < me; m++)
            if (d->mods[2 * (uint64_t)m] >= ilen[i]) { ctx->err = "modification position outside its interval (the reference raises IndexError)"; return TKSMSEQ_EINVAL; }
    }
    b->raw_len.resize(d->n_reads);
    for (uint64_t r = 0; r < d->n_reads; r++) {
        const uint32_t ib = d->reads[2 * r], ic = d->reads[2 * r + 1];
        if ((uint64_t)ib + ic > d->n_intervals) { ctx->err = "read refers to intervals that do not exist"; return TKSMSEQ_EINVAL; }
        uint64_t t = 0;
        for (uint32_t i = 0; i < ic; i++) t += ilen[ib + i];
        if (t > 0x7fffff00ull) { ctx->err = "molecule longer than 2^31 bases"; return TKSMSEQ_ELIMIT; }
        b->raw_len[r] = (uint32_t)t;
        b->max_raw = std::max(b->max_raw, (uint32_t)t);
        b->total_raw += t;
        if ((uint64_t)d->ids[2 * r] + d->ids[2 * r + 1] > d->id_bytes) { ctx->err = "molecule id outside the id pool"; return TKSMSEQ_EINVAL; }
    }
    order_by_length(b->raw_len, b->order);
    auto up = [&](DevBuf& buf, const void* src, size_t bytes) -> int {
        HIPCHK(ctx, buf.ensure(bytes + 64));
        if (bytes) HIPCHK(ctx, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return TKSMSEQ_OK;
    };
    int rc;
    if ((rc = up(b->reads, d->reads, d->n_reads * 8))) return rc;
    // intervals + sentinel carrying n_mods
    std::vector<uint32_t> iv(d->intervals, d->intervals + 4 * d->n_intervals);
    iv.push_back(0); iv.push_back(0); iv.push_back(0); iv.push_back((uint32_t)d->n_mods);
    if ((rc = up(b->intervals, iv.data(), iv.size() * 4))) return rc;
    if ((rc = up(b->mods, d->mods, d->n_mods * 8))) return rc;
    if ((rc = up(b->literals, d->literals, d->n_literals * 16))) return rc;
    if ((rc = up(b->litpool, d->literal_pool, d->literal_bytes))) return rc;
    if ((rc = up(b->ids, d->ids, d->n_reads * 8))) return rc;
    if ((rc = up(b->idpool, d->id_pool, d->id_bytes))) return rc;
    if ((rc = up(b->d_order, b->order.data(), b->order.size() * 4))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out = b.release();
    return TKSMSEQ_OK;
}

int tksmseq_batch_create(tksmseq_ctx* ctx, const tksmseq_batch_desc* d, tksmseq_batch** out) {
    if (!ctx || !d || !out) return TKSMSEQ_EINVAL;
    return batch_from_host(ctx, d, out);
}

static int batch_from_text(tksmseq_ctx* ctx, const char* text, uint64_t len, tksmseq_batch** out, bool check_mods);

int tksmseq_batch_from_mdf_text(tksmseq_ctx* ctx, const char* text, uint64_t len, tksmseq_batch** out) { return batch_from_text(ctx, text, len, out, true); }

// For the MDF -> MDF modules (PCR, truncation), which run without a reference: contig names the context does not know are
// kept as literal names, and substitution positions are not checked against slices that cannot be taken here.
int tksmseq_molecules_from_mdf_text(tksmseq_ctx* ctx, const char* text, uint64_t len, tksmseq_batch** out) { return batch_from_text(ctx, text, len, out, false); }

static int batch_from_text(tksmseq_ctx* ctx, const char* text, uint64_t len, tksmseq_batch** out, bool check_mods) {
    if (!ctx || (!text && len) || !out) return TKSMSEQ_EINVAL;
    BatchHost h;
    const auto t_text = std::chrono::steady_clock::now();
    if (!parse_mdf_mt(text, len, *ctx, h, ctx->err, ctx->host_threads)) return TKSMSEQ_EINVAL;
    const double s_text = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_text).count();
    const double a0 = alloc_seconds();
    tksmseq_batch_desc d{};
    d.n_reads = h.reads.size() / 2; d.n_intervals = h.intervals.size() / 4; d.n_mods = h.mods.size() / 2;
    d.n_literals = h.literals.size() / 2; d.literal_bytes = h.literal_pool.size(); d.id_bytes = h.id_pool.size();
    d.reads = h.reads.data(); d.intervals = h.intervals.data(); d.mods = h.mods.data(); d.literals = h.literals.data();
    d.literal_pool = h.literal_pool.data(); d.ids = h.ids.data(); d.id_pool = h.id_pool.data();
    const int rc = batch_from_host(ctx, &d, out, check_mods);
    if (rc != TKSMSEQ_OK) return rc;
    if (verbose_level() >= 2)
        fprintf(stderr, "[tksmseq] batch from %.0f MB of MDF text: parse %.3f s, tables + upload %.3f s (of which device allocation %.3f s)\n", len / 1e6, s_text,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_text).count() - s_text, alloc_seconds() - a0);
    // kept for PCR / truncation / the MDF writer: which reads are copies of a depth > 1 molecule, and the header comments
    tksmseq_batch* b = *out;
    bool any_dup = false;
    for (uint32_t v : h.dup) any_dup |= (v >> 31) != 0;
    if (any_dup) {
        b->h_dup = std::move(h.dup);
        HIPCHK(ctx, b->d_dup.ensure(b->h_dup.size() * 4 + 16));
        HIPCHK(ctx, hipMemcpyAsync(b->d_dup.p, b->h_dup.data(), b->h_dup.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    b->h_comments = std::move(h.comments); b->h_comment_pool = std::move(h.comment_pool);
    return TKSMSEQ_OK;
}

int tksmseq_batch_info(const tksmseq_batch* b, uint64_t* n_reads, uint64_t* n_intervals, uint64_t* n_mods) {
    if (!b) return TKSMSEQ_EINVAL;
    if (n_reads) *n_reads = b->n_reads;
    if (n_intervals) *n_intervals = b->n_intervals;
    if (n_mods) *n_mods = b->n_mods;
    return TKSMSEQ_OK;
}

void tksmseq_batch_free(tksmseq_ctx* ctx, tksmseq_batch* b) {
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    const auto t_free = std::chrono::steady_clock::now();
    delete b;
    if (b && verbose_level() >= 2) fprintf(stderr, "[tksmseq] batch freed in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_free).count());
}

// ------------------------------------------------------------------------------------------- run
int tksmseq_set_output_buffer(tksmseq_ctx* ctx, void* p, uint64_t cap) {
    if (!ctx) return TKSMSEQ_EINVAL;
    ctx->user_out = p; ctx->user_out_cap = p ? cap : 0;
    return TKSMSEQ_OK;
}

static tk::BatchView batch_view(const tksmseq_batch* b) {
    return tk::BatchView{b->reads.as<uint32_t>(), b->intervals.as<uint32_t>(), b->mods.as<uint32_t>(), b->literals.as<uint64_t>(),
                         b->litpool.as<uint8_t>(), b->ids.as<uint32_t>(), b->idpool.as<uint8_t>(), b->n_reads, (uint32_t)b->n_literals};
}
static tk::RefView ref_view(const tksmseq_ctx* ctx) {
    return tk::RefView{ctx->d_packed.as<uint32_t>(), ctx->d_blocktab.as<uint32_t>(), ctx->d_pool.as<uint8_t>(),
                       ctx->d_contigs.as<uint64_t>(), (uint32_t)ctx->contig_names.size()};
}

// Tail noise (py/tksm_badread.py:335-339) lengthens the fragment before the error loop, and everything that is sized or
// ordered by length on the host follows: the lengths are drawn on the device (they depend on the run's seed and read
// indices only), read back, and the batch's lengths, maximum and sorted order are rebuilt for this run.
static int apply_tail(tksmseq_ctx* ctx, tksmseq_batch* b, const tksmseq_run_params* p) {
    const bool want = p->mode == TKSMSEQ_MODE_BADREAD && ctx->tail.enabled && b->n_reads > 0;
    const uint64_t key[4] = {p->seed, p->first_read_index, p->read_index_stride ? p->read_index_stride : 1, ctx->tail_version};
    if (want == b->tail_on && (!want || !memcmp(key, b->tail_key, sizeof(key)))) return TKSMSEQ_OK;
    if (b->splice_len.empty()) b->splice_len = b->raw_len;
    const uint64_t n = b->n_reads;
    if (want) {
        HIPCHK(ctx, b->d_tail.ensure(n * 4 + 16));
        tk::TailView T{(int)ctx->tail.lx.size(), (int)ctx->tail.ly.size(), ctx->tail.ratio, ctx->d_tail_lx.as<double>(),
                       ctx->d_tail_ly.as<double>(), ctx->d_tail_cdf.as<double>()};
        HIPCHK(ctx, tk::launch_tail_lengths(batch_view(b), ref_view(ctx), T, key[0], key[1], key[2], b->d_tail.as<uint32_t>(), ctx->stream));
        std::vector<uint32_t> tl(n);
        HIPCHK(ctx, hipMemcpyAsync(tl.data(), b->d_tail.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (uint64_t r = 0; r < n; r++) {
            const uint64_t t = (uint64_t)b->splice_len[r] + tl[r];
            if (t > 0x7fffff00ull) { ctx->err = "molecule plus tail noise longer than 2^31 bases"; return TKSMSEQ_ELIMIT; }
            b->raw_len[r] = (uint32_t)t;
        }
    } else b->raw_len = b->splice_len;
    b->max_raw = 0;
    for (uint32_t v : b->raw_len) b->max_raw = std::max(b->max_raw, v);
    order_by_length(b->raw_len, b->order);
    HIPCHK(ctx, hipMemcpyAsync(b->d_order.p, b->order.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    b->cache_k = -1;                      // the cached scratch size was for the old lengths
    b->tail_on = want;
    memcpy(b->tail_key, key, sizeof(key));
    return TKSMSEQ_OK;
}

static int run_once(tksmseq_ctx* ctx, tksmseq_batch* b, const tksmseq_run_params* p, int cap_num, int cap_den, int cap_add,
                    tksmseq_result* res, bool* overflow) {
    const int vlevel = verbose_level();
    const auto t_entry = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (vlevel >= 2) fprintf(stderr, "[tksmseq] run: %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count());
    };
    *overflow = false;
    const uint64_t n = b->n_reads;
    hipStream_t s = ctx->stream;
    const bool badread = p->mode == TKSMSEQ_MODE_BADREAD;
    const int k = badread ? ctx->em.k : 0;
    auto align16 = [](uint64_t v) { return (v + 15) & ~15ull; };
    auto capf = [&](uint64_t raw) { return align16((raw + 2 * (uint64_t)k) * cap_num / cap_den + cap_add); };
    if (b->cache_k != k || b->cache_num != cap_num || b->cache_den != cap_den || b->cache_add != cap_add) {
        uint64_t t = 0;
        for (uint32_t rl : b->raw_len) t += 2 * capf(rl);
        b->cache_scratch = t; b->cache_k = k; b->cache_num = cap_num; b->cache_den = cap_den; b->cache_add = cap_add;
    }
    // LDS geometry from the longest molecule of the batch
    const int lcap = (int)((b->max_raw + 2 * k + 7) & ~7u);   // multiple of 8: 64-bit LDS words follow 3 * lcap bytes
    const int ncap = badread ? (int)capf(b->max_raw) : 4;
    const bool direct = !badread && !ctx->force_slow && !p->collect_stats;   // --perfect: packed reference -> records, no working set
    // the wave-wide kernel keeps a read's whole working set in LDS: 3 L + 4 x capacity bytes.  Longer molecules can
    // still take the fast pipeline (fragment state in HBM); only if one of them needs the wave-wide kernel (non-ACGT
    // bytes, an alignment outside the band representation) the run fails with TKSMSEQ_ELIMIT.
    int s_lcap = lcap, s_ncap = ncap;
    if (badread && !ctx->force_slow && tk::simulate_lds_bytes(s_lcap, s_ncap, 1) > 160 * 1024) {
        while (s_lcap > 64 && tk::simulate_lds_bytes(s_lcap, (int)capf((uint64_t)(s_lcap - 2 * k)), 1) > 160 * 1024) s_lcap -= 64;
        s_ncap = (int)capf((uint64_t)(s_lcap - 2 * k));
    }
    int wpw = tk::WAVES_PER_WG;
    while (wpw > 1 && tk::simulate_lds_bytes(s_lcap, s_ncap, wpw) > 160 * 1024) wpw >>= 1;
    const int lds = tk::simulate_lds_bytes(s_lcap, s_ncap, wpw);
    // Badread mode: the fast pipeline keeps one joined window (1.5 x the fragment) of a read's last visit in LDS: ~100 kb
    if (!direct && (lds > 160 * 1024 || (badread && !ctx->force_slow ? lcap > 100000 : (ncap >= 65000 || lcap >= 65000)))) {
        ctx->err = "molecule of " + std::to_string(b->max_raw) + " bases exceeds the limit of this build (Badread mode: 100 000 bases)";
        return TKSMSEQ_ELIMIT;
    }
    int wgs_per_cu = std::min(std::min(32 / wpw, 16), std::max(1, (160 * 1024) / std::max(lds, 1)));
    const uint64_t want = (n + wpw - 1) / wpw;
    const int n_wgs = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->n_cus * wgs_per_cu));
    const int trace_words = (s_ncap + 2) * 4;   // {up mask, left mask} u64 per column of the final alignment

    HIPCHK(ctx, ctx->w_rawlen.ensure(n * 4 + 16));
    HIPCHK(ctx, ctx->w_slotcap.ensure(n * 8 + 16));
    HIPCHK(ctx, ctx->w_slotoff.ensure((n + 1) * 8 + 16));
    HIPCHK(ctx, ctx->w_outlen.ensure(n * 4 + 16));
    HIPCHK(ctx, ctx->w_ident.ensure(n * 8 + 16));
    HIPCHK(ctx, ctx->w_reclen.ensure(n * 8 + 16));
    HIPCHK(ctx, ctx->w_recoff.ensure((n + 1) * 8 + 16));
    HIPCHK(ctx, ctx->w_status.ensure(n * 4 + 16));
    HIPCHK(ctx, ctx->w_scan.ensure(tk::scan_temp_bytes(n) + 64));
    HIPCHK(ctx, ctx->w_trace.ensure(((size_t)n_wgs * wpw + (size_t)tksmseq_ctx::N_SIDE * tksmseq_ctx::SIDE_WAVES) * trace_words * 4 + 64));
    HIPCHK(ctx, ctx->w_counter.ensure(8192));
    HIPCHK(ctx, ctx->w_sums.ensure(64));
    HIPCHK(ctx, ctx->w_scratch.ensure(b->cache_scratch + 64));
    if (p->collect_stats) {
        HIPCHK(ctx, ctx->w_istats.ensure(n * 64 + 16));
        HIPCHK(ctx, ctx->w_dstats.ensure(n * 16 + 16));
        HIPCHK(ctx, hipMemsetAsync(ctx->w_istats.p, 0, n * 64, s));
        HIPCHK(ctx, hipMemsetAsync(ctx->w_dstats.p, 0, n * 16, s));
    }

    const tk::BatchView B = batch_view(b);
    const tk::RefView R = ref_view(ctx);
    tk::ErrModelView EM{ctx->em.type, k, ctx->em.max_alts, ctx->em_alt0 ? 1 : 0, ctx->em_uniform ? 1 : 0, ctx->d_cdf.as<uint32_t>(), ctx->d_alts.as<uint64_t>(), ctx->d_nalts.as<uint8_t>(), ctx->d_pself.as<uint2>(), ctx->d_cdf32.as<uint32_t>(), ctx->d_pseg.as<uint4>(), ctx->d_pt0.as<uint32_t>(), ctx->d_altenc.as<uint4>()};
    tk::QsModelView QM{ctx->qm.n_slots, ctx->qm.kmer_size, ctx->d_qkeys.as<uint64_t>(), ctx->d_qoff.as<uint32_t>(),
                       ctx->d_qcnt.as<uint32_t>(), ctx->d_qcdf.as<uint32_t>(), ctx->d_qq.as<uint8_t>(), ctx->d_qent.as<uint4>(),
                       ctx->d_qpairs.as<uint2>(), ctx->d_qguide.as<uint8_t>(), ctx->qm.guide_direct ? 1 : 0};
    tk::IdentView IM{ctx->idm.constant ? 1 : 0, ctx->idm.value, ctx->d_qtab.as<double>()};
    tk::SimParams P{};
    P.seed = p->seed; P.first_read = p->first_read_index; P.stride = p->read_index_stride ? p->read_index_stride : 1;
    P.mode = badread ? 1 : 0; P.fastq = p->fastq ? 1 : 0;
    P.quirk_perfect = (badread && p->perfect_of_badread) ? 1 : 0;
    P.compute_q = (badread && p->compute_qual && p->fastq && !P.quirk_perfect) ? 1 : 0;
#ifdef TKSM_ABLATE
    P.ablate = getenv("TKSMSEQ_ABLATE") ? atoi(getenv("TKSMSEQ_ABLATE")) : 0;     // diagnostic build only (make ablate)
#endif
    P.lcap = lcap; P.ncap = ncap; P.s_lcap = s_lcap; P.s_ncap = s_ncap; P.trace_words = trace_words; P.cap_num = cap_num; P.cap_den = cap_den; P.cap_add = cap_add;
    tk::SimBuffers O{};
    O.raw_len = ctx->w_rawlen.as<uint32_t>(); O.slot_off = ctx->w_slotoff.as<uint64_t>(); O.scratch = ctx->w_scratch.as<uint8_t>();
    O.out_len = ctx->w_outlen.as<uint32_t>(); O.identity = ctx->w_ident.as<double>(); O.rec_len = ctx->w_reclen.as<uint64_t>();
    O.status = ctx->w_status.as<uint32_t>(); O.trace = ctx->w_trace.as<uint32_t>();
    O.work_counter = ctx->w_counter.as<unsigned long long>();
    O.tail_len = (badread && b->tail_on) ? b->d_tail.as<uint32_t>() : nullptr;
    O.tail_chain = ctx->d_tail_chain.as<tk::TailChain>();
    if (badread) {
        // memory for the unbanded alignments of the wave-wide kernel (rare: kernels.hip, full_align_wave)
        HIPCHK(ctx, ctx->w_fullpool.ensure(ctx->full_pool_bytes));
        O.full_pool = ctx->w_fullpool.as<uint8_t>(); O.full_pool_bytes = ctx->full_pool_bytes;
        O.full_pool_used = ctx->w_counter.as<unsigned long long>() + 1023;      // zeroed with the work counters
    }
    O.istats = p->collect_stats ? ctx->w_istats.as<int32_t>() : nullptr;
    O.dstats = p->collect_stats ? ctx->w_dstats.as<double>() : nullptr;

    const bool T = ctx->timing;
    O.read_list = nullptr; O.n_work = n;
    float ms_loop = 0, ms_aln = 0, ms_job = 0;
    if (T) HIPCHK(ctx, hipEventRecord(ctx->ev[0], s));
    HIPCHK(ctx, hipMemsetAsync(ctx->w_counter.p, 0, 8192, s));
    HIPCHK(ctx, tk::launch_read_lengths(B, R, k, cap_num, cap_den, cap_add, O.tail_len, ctx->w_rawlen.as<uint32_t>(), ctx->w_slotcap.as<uint64_t>(),
                                        ctx->w_status.as<uint32_t>(), s));
    HIPCHK(ctx, tk::launch_scan(ctx->w_slotcap.as<uint64_t>(), ctx->w_slotoff.as<uint64_t>(), n, ctx->w_scan.p, ctx->w_scan.cap, s));
    if (T) HIPCHK(ctx, hipEventRecord(ctx->ev[1], s));
    const bool fast = badread && !ctx->force_slow && n > 0;
    if (direct) {
        HIPCHK(ctx, tk::launch_perfect_lengths(B, R, P, O, s));
    } else if (!fast) {
        HIPCHK(ctx, tk::launch_simulate(B, R, EM, QM, IM, P, O, n_wgs, wpw, s));
    } else {
        // ---- fast pipeline: k_init, then rounds of k_err (wave per read) + k_aln (lane per alignment)
        tk::FastBuffers FB{};
        // job-id ranges: ~256 ranges of rs (multiple of 64) consecutive reads of the sorted order
        FB.rs = (uint32_t)((((n + 255) / 256) + 63) & ~63ull);
        FB.n_ranges = (uint32_t)((n + FB.rs - 1) / FB.rs);
        const uint64_t jcap = (uint64_t)FB.n_ranges * FB.rs;     // job slots (>= n)
        // rows of a range's jobs are sized by the range's longest read
        std::vector<uint32_t> r_ncap(FB.n_ranges);
        std::vector<uint32_t> r_tg(FB.n_ranges);              // 64-byte lines of predecessor codes per job (16 columns each, one spare)
        uint64_t tot_trace = 0, tot_popd = 0;
        for (uint32_t c = 0; c < FB.n_ranges; c++) {
            const uint64_t last = std::min<uint64_t>((uint64_t)(c + 1) * FB.rs, n) - 1;
            r_ncap[c] = (uint32_t)capf(b->raw_len[b->order[last]]);
            r_tg[c] = ((r_ncap[c] + 31) & ~31u) / 16 + 1;
            tot_trace += (uint64_t)FB.rs * r_tg[c]; tot_popd += (uint64_t)FB.rs * r_ncap[c];
        }
        HIPCHK(ctx, ctx->f_state.ensure(n * sizeof(tk::ReadState) + 64));
        // ragged per-read state rows: whole 64-position blocks, the padded fragment + at least one spare block
        ctx->h_row64.resize(n + 1);
        uint64_t nblk = 0;
        for (uint64_t r2 = 0; r2 < n; r2++) { ctx->h_row64[r2] = (uint32_t)nblk; nblk += ((uint64_t)b->raw_len[r2] + 2 * k + 63) / 64 + 1; }
        ctx->h_row64[n] = (uint32_t)nblk;
        if (nblk >= (1ull << 32)) { ctx->err = "batch too large (split it)"; return TKSMSEQ_ELIMIT; }
        HIPCHK(ctx, ctx->f_row64.ensure((n + 1) * 4 + 64));
        HIPCHK(ctx, hipMemcpyAsync(ctx->f_row64.p, ctx->h_row64.data(), (n + 1) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, ctx->f_frag.ensure(nblk * 64 + 256));
        HIPCHK(ctx, ctx->f_nb.ensure(nblk * 128 + 256));
        HIPCHK(ctx, ctx->f_fplanes.ensure((nblk + 8 * n) * 16 + 256));
        HIPCHK(ctx, ctx->f_frag2.ensure((4 * nblk + 4 * n) * 4 + 1024));
        for (int z = 0; z < 2; z++) {
            HIPCHK(ctx, ctx->f_jmeta[z].ensure(jcap * 16 + 64));
            if (z == 0) {
                // (one set: only the meta records and the counts of the previous round are read again)
                HIPCHK(ctx, ctx->f_jpopd[z].ensure(tot_popd + 64));
            }
        }
        // the round's exchange with the host, one copy each way (ctx.h): {job counts even | counters | job counts odd} down,
        // {prefix, bases | range geometry} up, through page-locked host memory
        const size_t nrb = (size_t)FB.n_ranges * 128;                               // bytes of one set of job counts
        const size_t round_bytes = 2 * nrb + 1024;
        const size_t geo_off = (((size_t)(FB.n_ranges + 1) * 12 + 63) & ~(size_t)63);   // range geometry behind {prefix, base_prev, base_cur}
        const size_t geo_bytes = geo_off + (size_t)FB.n_ranges * 2 * sizeof(tk::RangeGeo);
        HIPCHK(ctx, ctx->f_round.ensure(round_bytes + 64));
        HIPCHK(ctx, ctx->f_geoall.ensure(geo_bytes + 64));
        if (ctx->h_round_bytes < round_bytes) {
            if (ctx->h_round) (void)hipHostFree(ctx->h_round);
            ctx->h_round = nullptr; ctx->h_round_bytes = 0;
            HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_round, round_bytes * 2, hipHostMallocDefault));
            ctx->h_round_bytes = round_bytes * 2;
        }
        if (ctx->h_geo_bytes < geo_bytes) {
            if (ctx->h_geo) (void)hipHostFree(ctx->h_geo);
            ctx->h_geo = nullptr; ctx->h_geo_bytes = 0;
            HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_geo, geo_bytes * 2, hipHostMallocDefault));
            ctx->h_geo_bytes = geo_bytes * 2;
        }
        uint8_t* const d_round = ctx->f_round.as<uint8_t>();
        HIPCHK(ctx, ctx->f_trace.ensure(tot_trace * 64 + 64));                     // predecessor codes of the first alignment pass
        HIPCHK(ctx, ctx->f_redo.ensure(jcap * 4 + 64));
        // pool of full-width rows: as many as a round can ask for, at most 4 GB (homopolymer-rich batches need many)
        // code lines (4 iterations each; whole passes of 16 iterations, some room for drain passes), one uint4 of shift bytes per pass, a spare line
        FB.full_cl = ((((uint32_t)ncap + 31) & ~31u) / 4 + 16 + 3) & ~3u;
        FB.full_tg = FB.full_cl + (FB.full_cl / 4 + 3) / 4 + 1;
        FB.full_rows = (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(jcap, (4ull << 30) / ((uint64_t)FB.full_tg * 64)) & ~63ull);
        HIPCHK(ctx, ctx->f_tracefull.ensure((size_t)FB.full_rows * FB.full_tg * 64 + 64));
        HIPCHK(ctx, ctx->f_slow.ensure(n * 4 + 64));
        FB.state = ctx->f_state.as<tk::ReadState>(); FB.st_frag = ctx->f_frag.as<uint8_t>(); FB.st_nb = ctx->f_nb.as<uint16_t>();
        FB.st_fplanes = ctx->f_fplanes.as<unsigned long long>(); FB.st_frag2 = ctx->f_frag2.as<uint32_t>(); FB.row64 = ctx->f_row64.as<uint32_t>();
        FB.trace = ctx->f_trace.p;
        FB.redo_list = ctx->f_redo.as<uint32_t>();
        FB.trace_full = ctx->f_tracefull.p; FB.counters = reinterpret_cast<uint32_t*>(d_round + nrb);
        FB.slow_list = ctx->f_slow.as<uint32_t>();
        // predicted stragglers (below): histogram of the reads' scores, their list, what they hand to the exact kernel
        constexpr uint32_t EARLY_CAP = 4096;
        const bool early_on = ctx->early_tail > 0 && ctx->tail_cut == 0 && ctx->tail_wave > 0 && tk::tail_lds_bytes(lcap) <= 65536 && n >= 16ull * ctx->early_tail;
        FB.early_hist = nullptr; FB.early_list = nullptr; FB.early_slow = nullptr;
        if (early_on) {
            HIPCHK(ctx, ctx->f_early.ensure(65536 + (size_t)EARLY_CAP * 12 + 64));
            FB.early_hist = ctx->f_early.as<uint32_t>();
            FB.early_list = reinterpret_cast<uint2*>(ctx->f_early.as<uint8_t>() + 65536);
            FB.early_slow = reinterpret_cast<uint32_t*>(ctx->f_early.as<uint8_t>() + 65536 + (size_t)EARLY_CAP * 8);
            HIPCHK(ctx, hipMemsetAsync(ctx->f_early.p, 0, 65536, s));
        }
        HIPCHK(ctx, ctx->f_defer.ensure(n * 8 + 64));
        HIPCHK(ctx, ctx->f_defercnt.ensure((size_t)FB.n_ranges * 128 + 64));
        FB.defer_list = ctx->f_defer.as<uint2>(); FB.defer_cnt = ctx->f_defercnt.as<uint32_t>(); FB.defer_len = ctx->defer_len;
        HIPCHK(ctx, hipMemsetAsync(ctx->f_defercnt.p, 0, (size_t)FB.n_ranges * 128, s));
        FB.prefix = ctx->f_geoall.as<uint32_t>(); FB.base_prev = FB.prefix + (FB.n_ranges + 1); FB.base_cur = FB.prefix + 2 * (FB.n_ranges + 1);
        FB.geo_cur = reinterpret_cast<tk::RangeGeo*>(ctx->f_geoall.as<uint8_t>() + geo_off); FB.geo_prev = FB.geo_cur + FB.n_ranges;
        auto select_set = [&](uint32_t round) {
            const int z = round & 1, y = z ^ 1;
            FB.job_meta = ctx->f_jmeta[z].as<uint32_t>();
            FB.job_popd = ctx->f_jpopd[0].as<uint8_t>(); FB.job_cnt = reinterpret_cast<uint32_t*>(d_round + (z ? nrb + 1024 : 0));
            FB.prev_meta = ctx->f_jmeta[y].as<uint32_t>(); FB.prev_popd = ctx->f_jpopd[0].as<uint8_t>();
        };
        select_set(0);
        // host copy of {prefix, base_prev, base_cur}, uploaded before every round
        const size_t nr1 = FB.n_ranges + 1;
        // (page-locked: the copy of a round has run by the time the host writes the next round's values -- after that round's
        // synchronisation -- so one buffer is enough)
        HIPCHK(ctx, hipStreamSynchronize(s));                                      // (an earlier run's last copy)
        memset(ctx->h_geo, 0, geo_bytes);
        uint32_t* hprefix = reinterpret_cast<uint32_t*>(ctx->h_geo);
        uint32_t* hbase_prev = hprefix + nr1;
        uint32_t* hbase_cur = hprefix + 2 * nr1;
        for (uint32_t c = 0; c <= FB.n_ranges; c++) hbase_cur[c] = hbase_prev[c] = c * FB.rs;
        // where the rows of every range start in this round's (first half) and the previous round's (second half) job set
        tk::RangeGeo* hrg = reinterpret_cast<tk::RangeGeo*>(ctx->h_geo + geo_off);
        auto place_ranges = [&]() {
            uint64_t ot = 0, oj = 0, op = 0;
            for (uint32_t c = 0; c < FB.n_ranges; c++) {
                hrg[FB.n_ranges + c] = hrg[c];
                const uint64_t slots = hbase_cur[c + 1] - hbase_cur[c];
                tk::RangeGeo g{};
                g.trace_off = ot; g.jc_off = oj; g.popd_off = op;
                g.tstride = r_tg[c]; g.ncap = r_ncap[c];
                hrg[c] = g;
                ot += slots * g.tstride; op += slots * g.ncap;
            }
            return hipMemcpyAsync(ctx->f_geoall.p, ctx->h_geo, geo_bytes, hipMemcpyHostToDevice, s);      // {prefix, bases} go along
        };
        HIPCHK(ctx, place_ranges());
        uint32_t* const cnt = ctx->h_round + nrb / 4;                              // the counters' place in the host copy of f_round
        const uint32_t* hcnt = ctx->h_round;                                       // this round's job counts (set below)
        // length buckets over the sorted read order: each bucket gets its own LDS geometry
        struct Bucket { uint32_t begin, count; int lcap, ncap, wpw; bool hbm; };
        std::vector<Bucket> buckets;
        {
            const uint32_t minr = b->raw_len[b->order.front()], maxr = b->raw_len[b->order.back()];
            const uint32_t step = std::max<uint32_t>(128, ((maxr - minr) / ctx->n_buckets + 63) & ~63u);
            uint64_t i0 = 0;
            while (i0 < n) {
                const uint32_t lim = (b->raw_len[b->order[i0]] / step + 1) * step;
                uint64_t i1 = i0;
                while (i1 < n && b->raw_len[b->order[i1]] < lim) i1++;
                const uint32_t mx = b->raw_len[b->order[i1 - 1]];
                Bucket bk;
                bk.begin = (uint32_t)i0; bk.count = (uint32_t)(i1 - i0);
                bk.lcap = (int)((mx + 2 * k + 7) & ~7u); bk.ncap = (int)capf(mx);
                bk.hbm = bk.lcap > ctx->hbm_state_len;            // long reads: fragment state edited in HBM (kernels.hip, k_err)
                bk.wpw = tk::WAVES_PER_WG;
                while (bk.wpw > 1 && tk::err_lds_bytes(bk.lcap, bk.ncap, bk.wpw, bk.hbm) > 64 * 1024) bk.wpw >>= 1;
                buckets.push_back(bk);
                i0 = i1;
            }
        }
        size_t evi = 0;
        auto tick = [&]() -> int {
            if (!T) return 0;
            if (evi >= ctx->evpool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return 1; ctx->evpool.push_back(e); }
            return hipEventRecord(ctx->evpool[evi++], s) == hipSuccess ? 0 : 1;
        };
        std::vector<int> kinds;   // kernel kind between event i and i+1: 0 other (k_init, k_err, wave-wide kernel), 1 k_loop, 2 k_aln, 3 k_job, -1 host gap
        HIPCHK(ctx, hipMemsetAsync(ctx->f_nb.p, 0, nblk * 128, s));
        HIPCHK(ctx, hipMemsetAsync(ctx->f_round.p, 0, round_bytes, s));
        if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
        mark("buffers ready");
        HIPCHK(ctx, tk::launch_init(B, R, EM, IM, P, O, FB, lcap * tk::WAVES_PER_WG <= 150 * 1024 ? tk::WAVES_PER_WG : (lcap * 2 <= 150 * 1024 ? 2 : 1), s));
        if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
        kinds.push_back(0);
        cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
        // reads with non-ACGT bytes are known after k_init: their wave-wide kernel (latency-bound, a few waves) starts
        // now on a second stream and runs underneath the rounds
        // ... and so does the kernel of every read that leaves the fast pipeline later (an alignment the band
        // representation cannot hold: about one read in two million): launched as soon as the host sees it
        uint32_t n_side = 0, side_launches = 0;
        bool late_flushed = false;
        for (bool& u : ctx->side_used) u = false;
        // whatever way this function is left (an error return in the middle of the rounds included), no kernel of the side streams
        // may still be running on the context's buffers when the caller reuses or frees them
        struct SideGuard {
            tksmseq_ctx* c;
            ~SideGuard() {
                for (int k2 = 0; k2 < tksmseq_ctx::N_SIDE; k2++) if (c->side_used[k2] && c->side[k2]) (void)hipStreamSynchronize(c->side[k2]);
                if (c->early_stream) (void)hipStreamSynchronize(c->early_stream);
                if (c->aln_stream) (void)hipStreamSynchronize(c->aln_stream);
            }
        } side_guard{ctx};
        auto launch_side = [&](uint32_t upto) -> int {
            if (upto <= n_side || side_launches + 2 >= 1024) return TKSMSEQ_OK;
            const int k2 = (int)(side_launches % tksmseq_ctx::N_SIDE);
            if (!ctx->side_start) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->side_start, hipEventDisableTiming));
            if (!ctx->side[k2]) {
                HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side[k2], hipStreamNonBlocking));
                HIPCHK(ctx, hipEventCreateWithFlags(&ctx->side_done[k2], hipEventDisableTiming));
            }
            tk::SimBuffers O2 = O;
            O2.read_list = ctx->f_slow.as<uint32_t>() + n_side; O2.n_work = upto - n_side;
            O2.work_counter = ctx->w_counter.as<unsigned long long>() + 1 + side_launches;     // zeroed at the start of the run
            O2.trace = O.trace + ((size_t)n_wgs * wpw + (size_t)k2 * tksmseq_ctx::SIDE_WAVES) * trace_words;
            HIPCHK(ctx, hipEventRecord(ctx->side_start, s));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->side[k2], ctx->side_start, 0));
            const uint64_t want2 = (O2.n_work + wpw - 1) / wpw;
            const int wgs2 = (int)std::max<uint64_t>(1, std::min<uint64_t>(want2, (uint64_t)(tksmseq_ctx::SIDE_WAVES / wpw)));
            HIPCHK(ctx, tk::launch_simulate(B, R, EM, QM, IM, P, O2, wgs2, wpw, ctx->side[k2]));
            HIPCHK(ctx, hipEventRecord(ctx->side_done[k2], ctx->side[k2]));
            ctx->side_used[k2] = true;
            n_side = upto; side_launches++;
            return TKSMSEQ_OK;
        };
        HIPCHK(ctx, hipMemcpyAsync(cnt, FB.counters, 64, hipMemcpyDeviceToHost, s));
        std::vector<uint32_t> early_copies(early_on ? 64 * 256 : 0);
        if (early_on) HIPCHK(ctx, hipMemcpyAsync(early_copies.data(), FB.early_hist, early_copies.size() * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        mark("fragments spliced (k_init)");
        { const int rc2 = launch_side(cnt[2]); if (rc2) return rc2; }
        // ---- predicted stragglers.  A read's visits are ~ 0.14 x length x (1 - target identity), known now.  In a batch whose
        // distribution of that score has a long tail (skewed lengths), the reads at its end set the number of rounds and the length
        // of the straggler launch: the top early_tail of them -- those that need > 4 x the median read's visits -- get their waves at
        // once, on a stream of their own, and run underneath the regular rounds (which pass them by).
        bool early_active = false, early_joined = false;
        uint32_t n_early = 0;
        if (early_on) {
            uint32_t early_hist[256] = {};
            for (size_t i = 0; i < early_copies.size(); i++) early_hist[i & 255] += early_copies[i];
            uint64_t total = 0; for (uint32_t v : early_hist) total += v;
            uint64_t acc = 0; int median_bin = 0;
            for (int bb = 0; bb < 256; bb++) { acc += early_hist[bb]; if (2 * acc >= total) { median_bin = bb; break; } }
            int min_bin = 256; uint64_t top = 0;
            while (min_bin > median_bin + 16 && top + early_hist[min_bin - 1] <= ctx->early_tail) { min_bin--; top += early_hist[min_bin]; }   // 8 bins per factor of two: 16 bins = 4 x
            if (top > 0) {
                if (!ctx->early_stream) {
                    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->early_stream, hipStreamNonBlocking));
                    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->early_start, hipEventDisableTiming));
                    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->early_done, hipEventDisableTiming));
                }
                n_early = (uint32_t)top;
                HIPCHK(ctx, tk::launch_mark_early(FB, b->d_order.as<uint32_t>(), n, k, (uint32_t)min_bin, s));
                HIPCHK(ctx, hipEventRecord(ctx->early_start, s));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->early_stream, ctx->early_start, 0));
                HIPCHK(ctx, tk::launch_tail_early(EM, P, FB, n_early, lcap, ctx->tail_wcap, ctx->early_stream));
                HIPCHK(ctx, hipEventRecord(ctx->early_done, ctx->early_stream));
                early_active = true;
            }
        }
        ctx->last_early = n_early;
        uint32_t rounds = 0, n_deferred = 0;
        uint64_t jobs_all = 0, jobs_14 = 0;                  // alignment jobs launched (diagnostics)
        bool revive = false, revived = false;
        // waves of the straggler kernel the device holds at once (its LDS per wave grows with the longest fragment of the batch): it
        // takes over when every read that is left gets a wave of its own at once
        const uint64_t tail_slots = (uint64_t)ctx->n_cus * std::min<uint64_t>(16, (160u * 1024u) / tk::tail_lds_bytes(lcap));
        for (;; rounds++) {
            select_set(rounds);
            hcnt = ctx->h_round + ((rounds & 1) ? (nrb + 1024) / 4 : 0);
            HIPCHK(ctx, tk::launch_round_reset(FB, s));                  // this round's job counts, the alignment passes' counters
            if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
            kinds.push_back(-1);
            bool regular = false;
            const bool qround = revive && P.compute_q;                 // this round's jobs are the q-score alignments
            if (revive) {
                // every read's error loop has ended.  With q-scores: one more alignment job per read, the whole new sequence
                // against the whole fragment (k_qjobs + k_job, then k_aln below); without: their output, in this one round
                if (P.compute_q) {
                    HIPCHK(ctx, tk::launch_qjobs(FB, k, n_deferred, s));
                } else {
                    const Bucket& bk = buckets.back();
                    HIPCHK(ctx, tk::launch_err(B, EM, QM, P, O, FB, b->d_order.as<uint32_t>(), 0, n_deferred, bk.lcap, bk.ncap, 2, 0, FB.n_ranges, bk.wpw, bk.hbm, s));
                }
                revive = false;
            } else if (revived) {
                // last visits: q-score lookups and output, one wave per q-score job; ranges are chunks of the sorted order,
                // so a bucket is a run of ranges
                size_t bi = 0;
                uint32_t c = 0;
                while (c < FB.n_ranges) {
                    const uint32_t last_pos = (uint32_t)std::min<uint64_t>((uint64_t)(c + 1) * FB.rs, n) - 1;
                    while (bi + 1 < buckets.size() && last_pos >= buckets[bi].begin + buckets[bi].count) bi++;
                    uint32_t c1 = c + 1;
                    while (c1 < FB.n_ranges) {
                        const uint32_t lp = (uint32_t)std::min<uint64_t>((uint64_t)(c1 + 1) * FB.rs, n) - 1;
                        if (lp >= buckets[bi].begin + buckets[bi].count) break;
                        c1++;
                    }
                    const uint32_t cntw = hprefix[c1] - hprefix[c];
                    if (cntw)
                        HIPCHK(ctx, tk::launch_err(B, EM, QM, P, O, FB, b->d_order.as<uint32_t>(), 0, cntw, buckets[bi].lcap, buckets[bi].ncap, 1, c, c1, buckets[bi].wpw, buckets[bi].hbm, s));
                    c = c1;
                }
            } else {
                // the error loops of all reads that are still running, one lane each (round 0: every read, in sorted order;
                // later: the reads of the previous round's jobs), then this round's jobs packed for k_aln, one lane each
                if (rounds == 0) HIPCHK(ctx, tk::launch_loop(EM, P, FB, b->d_order.as<uint32_t>(), 0, (uint32_t)n, lcap, 0, 0, 0, s));
                else if (hprefix[FB.n_ranges] <= std::min<uint64_t>(ctx->tail_wave, tail_slots) && tk::tail_lds_bytes(lcap) <= 65536)    // the stragglers: every remaining visit in this launch
                    HIPCHK(ctx, tk::launch_tail(EM, P, FB, b->d_order.as<uint32_t>(), 0, hprefix[FB.n_ranges], lcap, 1, 0, FB.n_ranges, ctx->tail_wcap, s));
                else if (hprefix[FB.n_ranges] <= ctx->wave_loop && lcap <= 32768)        // few reads left: a wave each (latency)
                    HIPCHK(ctx, tk::launch_loopw(EM, P, FB, b->d_order.as<uint32_t>(), 0, hprefix[FB.n_ranges], lcap, 1, 0, FB.n_ranges, s));
                else HIPCHK(ctx, tk::launch_loop(EM, P, FB, b->d_order.as<uint32_t>(), 0, hprefix[FB.n_ranges], lcap, 1, 0, FB.n_ranges, s));
                if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
                kinds.push_back(1);
                regular = true;
            }
            if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
            kinds.push_back(regular ? 3 : 0);
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_round, ctx->f_round.p, round_bytes, hipMemcpyDeviceToHost, s));     // counters + job counts
            HIPCHK(ctx, hipStreamSynchronize(s));
            cnt[0] = 0;
            for (uint32_t c = 0; c < FB.n_ranges; c++) { hprefix[c] = cnt[0]; cnt[0] += hcnt[(size_t)c * 32]; }
            hprefix[FB.n_ranges] = cnt[0];
            // reads that left the fast pipeline in this round: a launch of the wave-wide kernel costs the latency of its
            // slowest read (25-45 ms), so they are collected while the rounds are busy and flushed in batches
            // (128 at a time, once more when the rounds become latency-bound; what comes after that waits for the end)
            const bool late = cnt[0] * 16ull < n;
            // -- and only onto a side stream that has finished its previous launch, unless a lot is waiting: many small
            // launches in a row on one stream each cost the full latency
            const uint32_t pending = cnt[2] - n_side;
            const int k_next = (int)(side_launches % tksmseq_ctx::N_SIDE);
            const bool stream_idle = !ctx->side_used[k_next] || hipEventQuery(ctx->side_done[k_next]) == hipSuccess;
            if ((pending >= 128 && (stream_idle || pending >= 2048)) || (pending && late && !late_flushed)) { const int rc2 = launch_side(cnt[2]); if (rc2) return rc2; }
            late_flushed = late_flushed || late;
            if (cnt[0] == 0 && early_active && !early_joined) {
                // the regular rounds are over: wait for the early reads' kernel, take over what it left for the exact kernel, and look
                // at the counters again (deferred reads, slow list)
                HIPCHK(ctx, hipEventSynchronize(ctx->early_done));
                HIPCHK(ctx, tk::launch_merge_early_slow(FB, s));
                HIPCHK(ctx, hipMemcpyAsync(cnt, FB.counters, 64, hipMemcpyDeviceToHost, s));
                HIPCHK(ctx, hipStreamSynchronize(s));
                cnt[0] = 0;
                early_joined = true;
            }
            if (cnt[0] == 0) {
                if (cnt[1] == 0 || revived) break;
                // every other read is done: job slots for the deferred reads (per-range counts), then their rounds
                revived = revive = true; n_deferred = cnt[1];
                std::vector<uint32_t> hd((size_t)FB.n_ranges * 32);
                HIPCHK(ctx, hipMemcpy(hd.data(), ctx->f_defercnt.p, hd.size() * 4, hipMemcpyDeviceToHost));
                uint32_t acc = 0;
                for (uint32_t c = 0; c < FB.n_ranges; c++) { hbase_prev[c] = hbase_cur[c]; hbase_cur[c] = acc; acc += (hd[(size_t)c * 32] + 63) & ~63u; hprefix[c] = 0; }
                hbase_prev[FB.n_ranges] = hbase_cur[FB.n_ranges]; hbase_cur[FB.n_ranges] = acc; hprefix[FB.n_ranges] = 0;
                HIPCHK(ctx, place_ranges());
                continue;
            }
#ifdef TKSM_ABLATE
            if (P.ablate >= 1 && P.ablate <= 9) break;          // k_err returned early: the reads would never finish
#endif
            if (cnt[0] < ctx->tail_cut && cnt[0] * 64ull < n) {
                // tail: every further round costs a full alignment latency for a handful of reads; finish the
                // stragglers in one launch of the wave-wide kernel instead (same results: it recomputes them)
                HIPCHK(ctx, tk::launch_collect_unfinished(FB, n, s));
                HIPCHK(ctx, hipMemcpyAsync(cnt, FB.counters, 64, hipMemcpyDeviceToHost, s));
                HIPCHK(ctx, hipStreamSynchronize(s));
                break;
            }
            if (rounds > 100000) { ctx->err = "internal: error loop did not terminate"; return TKSMSEQ_EDEVICE; }
            if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
            kinds.push_back(-1);
            {
                const uint32_t n_jobs = hbase_cur[FB.n_ranges - 1] + ((hcnt[(size_t)(FB.n_ranges - 1) * 32] + 63) & ~63u);
                const bool full_only = cnt[0] <= std::min(ctx->small_aln, FB.full_rows);
                jobs_all += cnt[0]; if (!full_only) jobs_14 += cnt[0];
                hipStream_t as = s;
                if (ctx->aln_prio_set && !full_only) {
                    if (!ctx->aln_stream) {
                        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->aln_stream, hipStreamNonBlocking, ctx->aln_prio));
                        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->aln_start, hipEventDisableTiming));
                        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->aln_done, hipEventDisableTiming));
                    }
                    HIPCHK(ctx, hipEventRecord(ctx->aln_start, s));
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->aln_stream, ctx->aln_start, 0));
                    as = ctx->aln_stream;
                }
                HIPCHK(ctx, tk::launch_alnf(P, FB, O, n_jobs, full_only, qround ? 1 : 0, ctx->aln_lds_pad, as));
                if (as != s) {
                    HIPCHK(ctx, hipEventRecord(ctx->aln_done, as));
                    HIPCHK(ctx, hipStreamWaitEvent(s, ctx->aln_done, 0));
                }
            }
            if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
            kinds.push_back(2);
            // next round: its jobs are packed by this round's counts (a read has at most one job per round)
            {
                uint32_t acc = 0;
                for (uint32_t c = 0; c < FB.n_ranges; c++) { hbase_prev[c] = hbase_cur[c]; hbase_cur[c] = acc; acc += (hcnt[(size_t)c * 32] + 63) & ~63u; }
                hbase_prev[FB.n_ranges] = hbase_cur[FB.n_ranges]; hbase_cur[FB.n_ranges] = acc;
                HIPCHK(ctx, place_ranges());
            }
        }
        ctx->last_rounds = rounds; ctx->last_slow = cnt[2];
        mark("rounds done");
        {
            // (the counters came down with the last round's copy; nothing that counts has run since)
            uint32_t* d = ctx->last_diag;
            memset(d, 0, sizeof(ctx->last_diag));
            d[0] = rounds; d[1] = cnt[2]; d[2] = n_early; d[3] = (uint32_t)std::min<uint64_t>(jobs_14, 0xffffffffu); d[4] = cnt[8]; d[5] = cnt[12];
            d[6] = cnt[13]; d[7] = cnt[24]; d[8] = cnt[25]; d[9] = (uint32_t)std::min<uint64_t>(jobs_all, 0xffffffffu); d[10] = cnt[4] + cnt[7];
        }
        if (getenv("TKSMSEQ_VERBOSE")) {
            uint32_t cc[32];
            HIPCHK(ctx, hipMemcpy(cc, FB.counters, 128, hipMemcpyDeviceToHost));
            fprintf(stderr, "[tksmseq] this thread so far: %u device allocations, %.3f s in hipMalloc\n", alloc_calls(), alloc_seconds());
            fprintf(stderr, "[tksmseq] reads %llu rounds %u slow-path reads %u (band exit %u/%u, shift %u/%u), full-width redo: %u jobs in %u waves; predicted stragglers on their own stream: %u\n",
                    (unsigned long long)n, rounds, cnt[2], cc[4], cc[7], cc[5], cc[6], cc[8], cc[9], n_early);
            fprintf(stderr, "[tksmseq] fused alignment failures: %u, reasons or-ed 0x%x, last 0x%x (n %u, m %u)\n", cc[12], cc[13], cc[14], cc[15] & 0xffffu, cc[15] >> 16);
            fprintf(stderr, "[tksmseq]   per reason: queue / reservoir overflow %u - - shift>31 %u shift>14 %u end cell %u walk %u | q-score jobs %u, list pass %u\n", cc[16], cc[19], cc[20], cc[21], cc[22], cc[24], cc[25]);
        }
        for (int k2 = 0; k2 < tksmseq_ctx::N_SIDE; k2++)
            if (ctx->side_used[k2]) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->side_done[k2], 0));
        if (cnt[2] > n_side) {
            // reads that left the fast pipeline later (alignment outside the band representation, tail cut): byte-exact
            // wave-wide path
            O.read_list = ctx->f_slow.as<uint32_t>() + n_side; O.n_work = cnt[2] - n_side;
            const uint64_t want2 = (cnt[2] - n_side + wpw - 1) / wpw;
            const int n_wgs2 = (int)std::max<uint64_t>(1, std::min<uint64_t>(want2, (uint64_t)n_wgs));
            if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
            kinds.push_back(-1);
            HIPCHK(ctx, tk::launch_simulate(B, R, EM, QM, IM, P, O, n_wgs2, wpw, s));
            if (tick()) { ctx->err = "event"; return TKSMSEQ_EDEVICE; }
            kinds.push_back(0);
        }
        if (T) {
            HIPCHK(ctx, hipStreamSynchronize(s));
            for (size_t i = 0; i + 1 < evi && i < kinds.size(); i++) {
                float ms = 0; (void)hipEventElapsedTime(&ms, ctx->evpool[i], ctx->evpool[i + 1]);
                if (kinds[i] == 1) ms_loop += ms; else if (kinds[i] == 2) ms_aln += ms; else if (kinds[i] == 3) ms_job += ms;
            }
        }
    }
    if (T) HIPCHK(ctx, hipEventRecord(ctx->ev[2], s));
    unsigned long long* sums = ctx->w_sums.as<unsigned long long>();
    unsigned long long hs[2] = {0, 0}; uint64_t total = 0;
    for (int pass = 0;; pass++) {
        HIPCHK(ctx, tk::launch_scan(ctx->w_reclen.as<uint64_t>(), ctx->w_recoff.as<uint64_t>(), n, ctx->w_scan.p, ctx->w_scan.cap, s));
        HIPCHK(ctx, tk::launch_sum_u32(ctx->w_status.as<uint32_t>(), n, sums, s));
        HIPCHK(ctx, tk::launch_sum_u32(ctx->w_outlen.as<uint32_t>(), n, sums + 1, s));
        if (T && pass == 0) HIPCHK(ctx, hipEventRecord(ctx->ev[3], s));
        HIPCHK(ctx, hipMemcpyAsync(hs, sums, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(&total, ctx->w_recoff.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        if (!hs[0]) break;
        std::vector<uint32_t> st(n);
        HIPCHK(ctx, hipMemcpy(st.data(), ctx->w_status.p, n * 4, hipMemcpyDeviceToHost));
        uint32_t any = 0; uint64_t first = 0;
        for (uint64_t i = 0; i < n; i++) if (st[i]) { if (!any) first = i; any |= st[i]; }
        if ((any & 16) && getenv("TKSMSEQ_VERBOSE")) {
            std::string l;
            int shown = 0;
            for (uint64_t i = 0; i < n && shown < 16; i++) if (st[i] & 16) { l += " " + std::to_string(i); shown++; }
            fprintf(stderr, "[tksmseq] reads with an unbanded alignment (first %d):%s\n", shown, l.c_str());
        }
        if (any & 2) { ctx->err = "modification position outside its interval at read " + std::to_string(first); return TKSMSEQ_EINVAL; }
        if (any & 4) { ctx->err = "out of memory for the unbanded alignment fallback at read " + std::to_string(first) + " (TKSMSEQ_FULL_POOL_MB)"; return TKSMSEQ_ENOMEM; }
        if ((any & 8) && pass == 0 && badread) {
            // molecules that need the exact wave-wide kernel (a non-ACGT byte, an alignment outside the band representation)
            // and are longer than its LDS-resident working set: the same kernel with the working sets in HBM, then the
            // sums and record offsets once more
            std::vector<uint32_t> big;
            for (uint64_t i = 0; i < n; i++) if (st[i] & 8) big.push_back((uint32_t)i);
            const int n_waves = (int)std::min<size_t>(big.size(), 64);
            const size_t per_wave = tk::simulate_big_bytes(lcap, ncap);
            HIPCHK(ctx, ctx->w_biglist.ensure(big.size() * 4 + 16));
            HIPCHK(ctx, ctx->w_bigscratch.ensure(per_wave * n_waves + 64));
            HIPCHK(ctx, ctx->w_bigtrace.ensure((size_t)n_waves * 2 * (ncap + 2) * 8 + 64));
            HIPCHK(ctx, hipMemcpyAsync(ctx->w_biglist.p, big.data(), big.size() * 4, hipMemcpyHostToDevice, s));
            HIPCHK(ctx, hipMemsetAsync(ctx->w_counter.p, 0, 8, s));
            tk::SimBuffers O3 = O;
            O3.read_list = ctx->w_biglist.as<uint32_t>(); O3.n_work = big.size();
            O3.work_counter = ctx->w_counter.as<unsigned long long>();
            O3.big_scratch = ctx->w_bigscratch.as<uint8_t>(); O3.big_per_wave = per_wave; O3.big_trace = ctx->w_bigtrace.as<unsigned long long>();
            HIPCHK(ctx, tk::launch_simulate_big(B, R, EM, QM, IM, P, O3, n_waves, s));
            if (getenv("TKSMSEQ_VERBOSE")) fprintf(stderr, "[tksmseq] %zu molecules beyond the LDS-resident limit took the exact kernel with HBM working sets\n", big.size());
            continue;
        }
        if (any & 8) {
            uint64_t f8 = 0;
            for (uint64_t i = 0; i < n; i++) if (st[i] & 8) { f8 = i; break; }
            ctx->err = "read " + std::to_string(f8) + " (" + std::to_string(b->raw_len[f8]) + " bases) exceeds the limits of the exact wave-wide kernel";
            return TKSMSEQ_ELIMIT;
        }
        if (any & 1) { *overflow = true; return TKSMSEQ_OK; }
        break;
    }
    uint8_t* records;
    if (ctx->user_out) {
        if (total > ctx->user_out_cap) { ctx->err = "caller-provided output buffer too small: need " + std::to_string(total) + " bytes"; return TKSMSEQ_ENOMEM; }
        records = (uint8_t*)ctx->user_out;
    } else {
        HIPCHK(ctx, ctx->w_records.ensure(total + 64));
        records = ctx->w_records.as<uint8_t>();
    }
    if (direct) HIPCHK(ctx, tk::launch_perfect(B, R, P, O, ctx->w_recoff.as<uint64_t>(), records, b->max_raw, ctx->n_cus, s));
    else HIPCHK(ctx, tk::launch_emit(B, P, O, ctx->w_recoff.as<uint64_t>(), records, s));
    if (T) {
        HIPCHK(ctx, hipEventRecord(ctx->ev[4], s));
        HIPCHK(ctx, hipEventSynchronize(ctx->ev[4]));
        for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&res->kernel_ms[i], ctx->ev[i], ctx->ev[i + 1]);
        (void)hipEventElapsedTime(&res->kernel_ms[4], ctx->ev[0], ctx->ev[4]);
        res->kernel_ms[5] = ms_loop; res->kernel_ms[6] = ms_aln; res->kernel_ms[7] = ms_job;
    }
    res->records = records; res->record_offsets = ctx->w_recoff.p; res->records_bytes = total; res->n_reads = n;
    res->bases_in = b->total_raw; res->bases_out = hs[1];
    return TKSMSEQ_OK;
}

int tksmseq_run(tksmseq_ctx* ctx, const tksmseq_batch* batch, const tksmseq_run_params* p, tksmseq_result* result) {
    if (!ctx || !batch || !p || !result) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(result, 0, sizeof(*result));
    if (p->mode != TKSMSEQ_MODE_PERFECT && p->mode != TKSMSEQ_MODE_BADREAD) { ctx->err = "unknown mode"; return TKSMSEQ_EINVAL; }
    if (p->mode == TKSMSEQ_MODE_BADREAD) {
        if (ctx->em.type < 0) { ctx->err = "no error model loaded"; return TKSMSEQ_ESTATE; }
        if (!ctx->idm.set) { ctx->err = "identity distribution not set"; return TKSMSEQ_ESTATE; }
        if (p->compute_qual && p->fastq && ctx->qm.n_slots == 0) { ctx->err = "no q-score model loaded"; return TKSMSEQ_ESTATE; }
    }
    tksmseq_batch* b = const_cast<tksmseq_batch*>(batch);
    bool overflow = false;
    memset(ctx->last_diag, 0, sizeof(ctx->last_diag));
    int rc = apply_tail(ctx, b, p);
    if (rc != TKSMSEQ_OK) return rc;
    rc = run_once(ctx, b, p, 3, 2, 64, result, &overflow);
    if (rc == TKSMSEQ_OK && overflow) {
        // insertion-heavy reads outgrew the default 1.5x slot: rerun with the worst-case factor
        rc = run_once(ctx, b, p, 6, 1, 64, result, &overflow);
        if (rc == TKSMSEQ_OK && overflow) { ctx->err = "internal: output slot overflow at the worst-case factor"; rc = TKSMSEQ_EDEVICE; }
    }
    if (rc == TKSMSEQ_OK) { ctx->last = *result; ctx->have_last = true; ctx->have_stats = p->collect_stats != 0; }
    return rc;
}

int tksmseq_run_diagnostics(tksmseq_ctx* ctx, uint32_t* out) {
    if (!ctx || !out) return TKSMSEQ_EINVAL;
    if (!ctx->have_last) return TKSMSEQ_ESTATE;
    memcpy(out, ctx->last_diag, sizeof(ctx->last_diag));
    return TKSMSEQ_OK;
}

int tksmseq_result_download(tksmseq_ctx* ctx, uint8_t* records, uint64_t* offsets) {
    if (!ctx || !ctx->have_last) return TKSMSEQ_ESTATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (records && ctx->last.records_bytes) HIPCHK(ctx, hipMemcpyAsync(records, ctx->last.records, ctx->last.records_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (offsets) HIPCHK(ctx, hipMemcpyAsync(offsets, ctx->last.record_offsets, (ctx->last.n_reads + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

int tksmseq_result_download_range(tksmseq_ctx* ctx, uint8_t* dst, uint64_t offset, uint64_t bytes, int async) {
    if (!ctx || !ctx->have_last || (!dst && bytes)) return TKSMSEQ_ESTATE;
    if (offset > ctx->last.records_bytes || bytes > ctx->last.records_bytes - offset) { ctx->err = "record range outside the last result"; return TKSMSEQ_EINVAL; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst, (const uint8_t*)ctx->last.records + offset, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (!async) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

int tksmseq_result_copy_device(tksmseq_ctx* ctx, void* records_dst, void* offsets_dst) {
    if (!ctx || !ctx->have_last) return TKSMSEQ_ESTATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (records_dst && ctx->last.records_bytes && records_dst != ctx->last.records)
        HIPCHK(ctx, hipMemcpyAsync(records_dst, ctx->last.records, ctx->last.records_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (offsets_dst) HIPCHK(ctx, hipMemcpyAsync(offsets_dst, ctx->last.record_offsets, (ctx->last.n_reads + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return TKSMSEQ_OK;
}

int tksmseq_stats_download(tksmseq_ctx* ctx, int32_t* istats, double* dstats) {
    if (!ctx || !ctx->have_last || !ctx->have_stats) return TKSMSEQ_ESTATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (istats) HIPCHK(ctx, hipMemcpyAsync(istats, ctx->w_istats.p, ctx->last.n_reads * 64, hipMemcpyDeviceToHost, ctx->stream));
    if (dstats) HIPCHK(ctx, hipMemcpyAsync(dstats, ctx->w_dstats.p, ctx->last.n_reads * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

// ------------------------------------------------------------------------------------------- interleave
int tksmseq_interleave_records(tksmseq_ctx* ctx, int n_ranks, const void* const* streams, const void* const* offsets,
                               const uint64_t* n_per_rank, void* dst, uint64_t dst_capacity, uint64_t* dst_bytes) {
    if (!ctx || n_ranks < 1 || n_ranks > 16 || !streams || !offsets || !n_per_rank || !dst) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    uint64_t n_total = 0;
    for (int i = 0; i < n_ranks; i++) {
        // round-robin sharding: rank p holds ceil((N - p) / P) reads
        n_total += n_per_rank[i];
        if (i && n_per_rank[i] > n_per_rank[i - 1]) { ctx->err = "per-rank read counts are not a round-robin split"; return TKSMSEQ_EINVAL; }
    }
    if (n_per_rank[0] - n_per_rank[n_ranks - 1] > 1) { ctx->err = "per-rank read counts are not a round-robin split"; return TKSMSEQ_EINVAL; }
    HIPCHK(ctx, ctx->w_reclen.ensure(n_total * 8 + 16));
    HIPCHK(ctx, ctx->w_slotoff.ensure((n_total + 1) * 8 + 16));
    HIPCHK(ctx, ctx->w_scan.ensure(tk::scan_temp_bytes(n_total) + 64));
    HIPCHK(ctx, tk::launch_interleave_lens(n_ranks, (const uint64_t* const*)offsets, n_per_rank, n_total, ctx->w_reclen.as<uint64_t>(), ctx->stream));
    HIPCHK(ctx, tk::launch_scan(ctx->w_reclen.as<uint64_t>(), ctx->w_slotoff.as<uint64_t>(), n_total, ctx->w_scan.p, ctx->w_scan.cap, ctx->stream));
    uint64_t total = 0;
    HIPCHK(ctx, hipMemcpyAsync(&total, ctx->w_slotoff.as<uint64_t>() + n_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (total > dst_capacity) { ctx->err = "interleave destination too small"; return TKSMSEQ_ENOMEM; }
    HIPCHK(ctx, tk::launch_interleave_copy(n_ranks, (const uint8_t* const*)streams, (const uint64_t* const*)offsets, n_total,
                                           ctx->w_slotoff.as<uint64_t>(), (uint8_t*)dst, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (dst_bytes) *dst_bytes = total;
    return TKSMSEQ_OK;
}

}  // extern "C"
