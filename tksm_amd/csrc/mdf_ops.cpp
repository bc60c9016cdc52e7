// mdf_ops.cpp -- host side of the molecule-description transforms upstream of Seq (BASELINE config 5) and of the MDF writer:
//   tksmseq_pcr            src/pcr.cpp:22-89, :138 (presets), :215-229 (whole input in memory, at most 2 x target templates)
//   tksmseq_truncate       src/truncate.cpp:23-65, :77-227, :322-351, :362-404
//   tksmseq_batch_to_mdf_text   molecule_descriptor::operator<<, src/interval.h:898-905 (+ dump_comment :880-890)
// The molecule tables stay on the device from one transform to the next and into tksmseq_run; only sizes, per-read lengths
// (which the host needs to size and order a Seq batch) and, for the text writer, the tables themselves come back.
#include <charconv>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

#include "ctx.h"
#include <thread>
#include "mdf_kernels.h"

namespace {

struct Ph4h { uint32_t x, y, z, w; };
Ph4h philox_host(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Ph4h{c0, c1, c2, c3};
}

tk::BatchView view_of(const tksmseq_batch* b) {
    return tk::BatchView{b->reads.as<uint32_t>(), b->intervals.as<uint32_t>(), b->mods.as<uint32_t>(), b->literals.as<uint64_t>(),
                         b->litpool.as<uint8_t>(), b->ids.as<uint32_t>(), b->idpool.as<uint8_t>(), b->n_reads, (uint32_t)b->n_literals};
}

int scan_to(tksmseq_ctx* ctx, DevBuf& in, DevBuf& out, uint64_t n, uint64_t* total) {
    HIPCHK(ctx, out.ensure((n + 1) * 8 + 16));
    HIPCHK(ctx, ctx->w_scan.ensure(tk::scan_temp_bytes(n) + 64));
    HIPCHK(ctx, tk::launch_scan(in.as<uint64_t>(), out.as<uint64_t>(), n, ctx->w_scan.p, ctx->w_scan.cap, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(total, out.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

// literals (contigs whose name is their sequence) are shared by reference: the output batch gets its own copy of the tables
int copy_literals(tksmseq_ctx* ctx, const tksmseq_batch* in, tksmseq_batch* out) {
    out->n_literals = in->n_literals;
    HIPCHK(ctx, out->literals.ensure(in->n_literals * 16 + 64));
    HIPCHK(ctx, out->litpool.ensure(in->litpool.cap + 64));
    if (in->n_literals) HIPCHK(ctx, hipMemcpyAsync(out->literals.p, in->literals.p, in->n_literals * 16, hipMemcpyDeviceToDevice, ctx->stream));
    if (in->litpool.cap) HIPCHK(ctx, hipMemcpyAsync(out->litpool.p, in->litpool.p, in->litpool.cap, hipMemcpyDeviceToDevice, ctx->stream));
    return TKSMSEQ_OK;
}

// std::map<string, vector<string>> round trip of a header comment (molecule_descriptor::comment / dump_comment,
// src/interval.h:809-830, :880-890), with extra values appended
std::string normalize_comment(const char* c, size_t n, const std::vector<std::pair<std::string, std::string>>& extra) {
    std::map<std::string, std::vector<std::string>> meta;
    size_t a = 0;
    while (a < n) {
        size_t b = a;
        while (b < n && c[b] != ';') b++;
        if (b > a) {
            const std::string f(c + a, b - a);
            const size_t eq = f.find('=');
            if (eq == std::string::npos) meta[f].push_back(".");
            else {
                size_t k1 = f.find_first_not_of('='), k2 = f.find('=', k1 == std::string::npos ? 0 : k1);
                const std::string key = k1 == std::string::npos ? std::string() : f.substr(k1, k2 - k1);
                size_t v = k2 == std::string::npos ? f.size() : k2;
                while (v < f.size() && f[v] == '=') v++;
                size_t ve = f.find('=', v);
                const std::string vals = f.substr(v, ve == std::string::npos ? std::string::npos : ve - v);
                size_t p = 0;
                while (p < vals.size()) {
                    size_t q = vals.find(',', p);
                    if (q == std::string::npos) q = vals.size();
                    if (q > p) meta[key].push_back(vals.substr(p, q - p));
                    p = q + 1;
                }
            }
        }
        a = b + 1;
    }
    for (auto& kv : extra) meta[kv.first].push_back(kv.second);
    std::string out;
    for (auto& kv : meta) {
        if (kv.second.empty()) continue;
        out += kv.first;
        if (kv.second[0] != ".") {
            out += '=';
            for (size_t i = 0; i < kv.second.size(); i++) { if (i) out += ','; out += kv.second[i]; }
        }
        out += ';';
    }
    return out;
}

// fmt's "{}" of a double: the shortest digits that round-trip, in FIXED notation while the decimal exponent is in [-4, 16) and in
// exponent notation outside (like Python's repr; std::to_chars alone would switch to "1e+05" as soon as that is shorter)
std::string fmt_double(double v) {
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string sci(buf, r.ptr);                          // d[.ddd]e[+-]XX: read the exponent off it
    const size_t e = sci.find('e');
    const int ex = e == std::string::npos ? 0 : atoi(sci.c_str() + e + 1);
    if (!std::isfinite(v) || ex < -4 || ex >= 16) {
        r = std::to_chars(buf, buf + sizeof buf, v);
        return std::string(buf, r.ptr);
    }
    r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

}  // namespace

// lengths (python-slice clamped, as Seq sees them), their maximum and sum, and the order sorted by length: what
// batch_from_host computes on the host for an uploaded batch
int finalize_device_batch(tksmseq_ctx* ctx, tksmseq_batch* b) {
    const uint64_t n = b->n_reads;
    b->raw_len.assign(n, 0); b->order.resize(n); b->max_raw = 0; b->total_raw = 0;
    b->splice_len.clear(); b->tail_on = false; b->cache_k = -1;
    HIPCHK(ctx, b->d_order.ensure(n * 4 + 64));
    if (!n) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return TKSMSEQ_OK; }      // (the callers' copies from stack / local buffers have run)
    HIPCHK(ctx, ctx->w_rawlen.ensure(n * 4 + 16));
    HIPCHK(ctx, ctx->w_slotcap.ensure(n * 8 + 16));
    HIPCHK(ctx, ctx->w_status.ensure(n * 4 + 16));
    const tk::RefView R{ctx->d_packed.as<uint32_t>(), ctx->d_blocktab.as<uint32_t>(), ctx->d_pool.as<uint8_t>(), ctx->d_contigs.as<uint64_t>(),
                        (uint32_t)ctx->contig_names.size()};
    HIPCHK(ctx, tk::launch_read_lengths(view_of(b), R, 0, 1, 1, 0, nullptr, ctx->w_rawlen.as<uint32_t>(), ctx->w_slotcap.as<uint64_t>(),
                                        ctx->w_status.as<uint32_t>(), ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->raw_len.data(), ctx->w_rawlen.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t r = 0; r < n; r++) { b->max_raw = std::max(b->max_raw, b->raw_len[r]); b->total_raw += b->raw_len[r]; b->order[r] = (uint32_t)r; }
    order_by_length(b->raw_len, b->order);
    HIPCHK(ctx, hipMemcpyAsync(b->d_order.p, b->order.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return TKSMSEQ_OK;
}

extern "C" {

int tksmseq_pcr_preset(const char* name, double* error_rate, double* efficiency) {
    // Cha & Thilly 1993 as listed in src/pcr.cpp:136-140
    static const struct { const char* n; double er, ef; } P[] = {
        {"Taq-setting1", 2 * std::pow(0.1, 4), 0.88}, {"Taq-setting2", 7.2 * std::pow(0.1, 5), 0.36}, {"Klenow", 1.3 * std::pow(0.1, 4), 0.80},
        {"T7", 3.4 * std::pow(0.1, 5), 0.90},          {"T4", 3.0 * std::pow(0.1, 6), 0.56},          {"Vent", 4.5 * std::pow(0.1, 5), 0.70}};
    if (!name) return TKSMSEQ_EINVAL;
    for (auto& p : P)
        if (!strcmp(p.n, name)) { if (error_rate) *error_rate = p.er; if (efficiency) *efficiency = p.ef; return TKSMSEQ_OK; }
    return TKSMSEQ_EINVAL;
}

// The templates of a PCR call and its kernel parameters, in PROCESSING ORDER.  Templates: every (depth-unrolled) molecule in input order,
// or 2 x target of them when there are more (src/pcr.cpp:217-220 shuffles and cuts: a uniformly random ordered subset; here: the
// 2 x target molecules with the smallest Philox keys, in key order); of those, positions [template_begin, template_end) of that order
// when the caller asks for a slice.  keep empty = every molecule, in input order.
static int pcr_setup(tksmseq_ctx* ctx, const tksmseq_batch* in, const tksmseq_pcr_params* p, std::vector<uint32_t>& keep, uint64_t& n_local, tk::PcrParams& P) {
    if (p->cycles < 0 || p->cycles > tk::PCR_MAX_CYCLES) { ctx->err = "PCR: between 0 and " + std::to_string(tk::PCR_MAX_CYCLES) + " cycles are supported"; return TKSMSEQ_ELIMIT; }
    if (!(p->efficiency >= 0.0) || !(p->error_rate >= 0.0)) { ctx->err = "PCR: efficiency and error rate must be non-negative"; return TKSMSEQ_EINVAL; }
    const uint64_t n = in->n_reads;
    const bool sliced = p->template_begin != 0 || p->template_end != 0;
    if (sliced && (p->template_begin > p->template_end || p->template_end > n)) { ctx->err = "PCR: template slice outside the batch"; return TKSMSEQ_EINVAL; }
    keep.clear();
    uint64_t n_kept = n;
    if (n > 2 * p->target_count) {
        n_kept = 2 * p->target_count;
        std::vector<std::pair<uint64_t, uint32_t>> key(n);
        for (uint64_t u = 0; u < n; u++) { const Ph4h w = philox_host(p->seed, (uint32_t)u, 0u, 16u, 0u); key[u] = {((uint64_t)w.x << 32) | w.y, (uint32_t)u}; }
        // std::shuffle + resize (src/pcr.cpp:217-220) = a uniformly random ORDERED subset: the 2 x target molecules with the smallest
        // keys, in the order of their keys -- the order in which their copies are written
        std::nth_element(key.begin(), key.begin() + (ptrdiff_t)n_kept, key.end());
        std::sort(key.begin(), key.begin() + (ptrdiff_t)n_kept);
        keep.resize(n_kept);
        for (uint64_t i = 0; i < n_kept; i++) keep[i] = key[i].second;
    }
    n_local = n_kept;
    if (sliced) {
        if (keep.empty()) {
            if (p->template_begin != 0 || p->template_end != n) { keep.resize(p->template_end - p->template_begin); for (size_t i = 0; i < keep.size(); i++) keep[i] = (uint32_t)(p->template_begin + i); }
        } else {
            // (a slice is a range of POSITIONS in the processing order: with the subsample that is the order of the keys)
            const uint64_t lo = std::min<uint64_t>(p->template_begin, n_kept), hi = std::min<uint64_t>(p->template_end, n_kept);
            keep = std::vector<uint32_t>(keep.begin() + (ptrdiff_t)lo, keep.begin() + (ptrdiff_t)hi);
        }
        n_local = (keep.empty() && p->template_begin == 0 && p->template_end == n) ? n : keep.size();
        if (n_local == 0) keep.assign(1, 0u);                     // (an empty slice: a list that is not "every molecule")
    }
    P = tk::PcrParams{};
    P.seed = p->seed; P.cycles = p->cycles; P.efficiency = p->efficiency; P.rate = (4 * p->error_rate) / 3;
    // drop ratio (src/pcr.cpp:68-76) of the whole input, then the probabilities that a subtree of copies emits nothing
    const double expected_after = std::pow(1 + p->efficiency, p->cycles) * (double)n_kept;
    P.drop = expected_after > 0.0 ? (double)p->target_count / expected_after : 0.0;
    if (P.drop > 1.0) P.drop = 1.0;
    const int c = p->cycles;
    P.A[c] = 1.0; P.A[c + 1] = 1.0;
    for (int t = c - 1; t >= 0; t--) {
        P.q[t] = (1.0 - P.drop) * P.A[t + 1];
        P.A[t] = P.A[t + 1] * (1.0 - P.efficiency * (1.0 - P.q[t]));
    }
    P.q[c] = 1.0;
    return TKSMSEQ_OK;
}

int tksmseq_pcr_template_counts(tksmseq_ctx* ctx, const tksmseq_batch* in, const tksmseq_pcr_params* p, uint64_t* counts) {
    if (!ctx || !in || !p || (!counts && in->n_reads)) return TKSMSEQ_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<uint32_t> keep;
    uint64_t n_kept = 0;
    tk::PcrParams P{};
    tksmseq_pcr_params whole = *p;
    whole.template_begin = whole.template_end = 0;
    int rc = pcr_setup(ctx, in, &whole, keep, n_kept, P);
    if (rc) return rc;
    DevBuf d_keep, d_cnt, d_status;
    for (DevBuf* pb_ : {&d_keep, &d_cnt, &d_status}) { pb_->pooled = true; pb_->pool_stream = s; }   // (per-call temporaries: DevCache, ctx.h)
    if (!keep.empty()) { HIPCHK(ctx, d_keep.ensure(n_kept * 4 + 16)); HIPCHK(ctx, hipMemcpyAsync(d_keep.p, keep.data(), n_kept * 4, hipMemcpyHostToDevice, s)); }
    tk::MolView M{view_of(in), in->d_dup.p ? in->d_dup.as<uint32_t>() : nullptr, in->n_intervals, in->n_mods, keep.empty() ? nullptr : d_keep.as<uint32_t>(), n_kept};
    HIPCHK(ctx, d_cnt.ensure(n_kept * 8 + 16));
    HIPCHK(ctx, d_status.ensure(64));
    HIPCHK(ctx, hipMemsetAsync(d_status.p, 0, 64, s));
    HIPCHK(ctx, tk::launch_pcr_count(M, P, d_cnt.as<uint64_t>(), d_status.as<uint32_t>(), s));
    std::vector<uint64_t> h(n_kept);
    uint32_t st = 0;
    if (n_kept) HIPCHK(ctx, hipMemcpyAsync(h.data(), d_cnt.p, n_kept * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(&st, d_status.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    if (st & 1u) { ctx->err = "PCR: more than " + std::to_string(tk::PCR_MAX_MUT) + " substitutions per copy (error rate x molecule length) are not supported"; return TKSMSEQ_ELIMIT; }
    std::fill(counts, counts + in->n_reads, 0ull);
    for (uint64_t i = 0; i < n_kept; i++) counts[i] = h[i];          // by position in the processing order (beyond n_kept: 0)
    return TKSMSEQ_OK;
}

int tksmseq_pcr(tksmseq_ctx* ctx, const tksmseq_batch* in, const tksmseq_pcr_params* p, tksmseq_batch** out) {
    if (!ctx || !in || !p || !out) return TKSMSEQ_EINVAL;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<uint32_t> keep;
    uint64_t n_kept = 0;
    tk::PcrParams P{};
    {
        const int rc0 = pcr_setup(ctx, in, p, keep, n_kept, P);
        if (rc0) return rc0;
    }

    DevBuf d_keep, d_cnt, d_off, d_status, n_mol, n_mask, n_ivl, n_mod, n_idl, o_ivl, o_mod, o_id;
    for (DevBuf* pb_ : {&d_keep, &d_cnt, &d_off, &d_status, &n_mol, &n_mask, &n_ivl, &n_mod, &n_idl, &o_ivl, &o_mod, &o_id}) { pb_->pooled = true; pb_->pool_stream = s; }   // (per-call temporaries: DevCache, ctx.h)
    if (!keep.empty()) { HIPCHK(ctx, d_keep.ensure(n_kept * 4 + 16)); HIPCHK(ctx, hipMemcpyAsync(d_keep.p, keep.data(), n_kept * 4, hipMemcpyHostToDevice, s)); }
    tk::MolView M{view_of(in), in->d_dup.p ? in->d_dup.as<uint32_t>() : nullptr, in->n_intervals, in->n_mods,
                  keep.empty() ? nullptr : d_keep.as<uint32_t>(), n_kept};
    HIPCHK(ctx, d_cnt.ensure(n_kept * 8 + 16));
    HIPCHK(ctx, d_status.ensure(64));
    HIPCHK(ctx, hipMemsetAsync(d_status.p, 0, 64, s));
    HIPCHK(ctx, tk::launch_pcr_count(M, P, d_cnt.as<uint64_t>(), d_status.as<uint32_t>(), s));
    uint64_t n_nodes = 0;
    int rc = scan_to(ctx, d_cnt, d_off, n_kept, &n_nodes);
    if (rc) return rc;
    uint32_t st = 0;
    HIPCHK(ctx, hipMemcpy(&st, d_status.p, 4, hipMemcpyDeviceToHost));
    if (st & 1u) { ctx->err = "PCR: more than " + std::to_string(tk::PCR_MAX_MUT) + " substitutions per copy (error rate x molecule length) are not supported"; return TKSMSEQ_ELIMIT; }
    if (n_nodes >= 0xffffffffull) { ctx->err = "PCR: more than 2^32 output molecules in one call"; return TKSMSEQ_ELIMIT; }
    HIPCHK(ctx, n_mol.ensure(n_nodes * 4 + 16));
    for (DevBuf* b : {&n_mask, &n_ivl, &n_mod, &n_idl}) HIPCHK(ctx, b->ensure(n_nodes * 8 + 16));
    HIPCHK(ctx, tk::launch_pcr_list(M, P, d_off.as<uint64_t>(), n_mol.as<uint32_t>(), n_mask.as<uint64_t>(), n_ivl.as<uint64_t>(), n_mod.as<uint64_t>(),
                                    n_idl.as<uint64_t>(), s));
    uint64_t t_ivl = 0, t_mod = 0, t_id = 0;
    if ((rc = scan_to(ctx, n_ivl, o_ivl, n_nodes, &t_ivl)) || (rc = scan_to(ctx, n_mod, o_mod, n_nodes, &t_mod)) || (rc = scan_to(ctx, n_idl, o_id, n_nodes, &t_id))) return rc;
    if (t_ivl >= 0x7fffffffull || t_mod >= 0x7fffffffull || t_id >= 0xffffffffull) { ctx->err = "PCR: output batch too large (split the input)"; return TKSMSEQ_ELIMIT; }
    std::unique_ptr<tksmseq_batch> b(new tksmseq_batch());
    b->n_reads = n_nodes; b->n_intervals = t_ivl; b->n_mods = t_mod;
    HIPCHK(ctx, b->reads.ensure(n_nodes * 8 + 64));
    HIPCHK(ctx, b->intervals.ensure((t_ivl + 1) * 16 + 64));
    HIPCHK(ctx, b->mods.ensure(t_mod * 8 + 64));
    HIPCHK(ctx, b->ids.ensure(n_nodes * 8 + 64));
    HIPCHK(ctx, b->idpool.ensure(t_id + 64));
    if ((rc = copy_literals(ctx, in, b.get()))) return rc;
    tk::MolOut O{b->reads.as<uint32_t>(), b->intervals.as<uint32_t>(), b->mods.as<uint32_t>(), b->ids.as<uint32_t>(), b->idpool.as<uint8_t>()};
    HIPCHK(ctx, tk::launch_pcr_write(M, P, n_nodes, n_mol.as<uint32_t>(), n_mask.as<uint64_t>(), o_ivl.as<uint64_t>(), o_mod.as<uint64_t>(),
                                     o_id.as<uint64_t>(), O, s));
    const uint32_t sentinel[4] = {0u, 0u, 0u, (uint32_t)t_mod};          // the interval after the last carries n_mods
    HIPCHK(ctx, hipMemcpyAsync(b->intervals.as<uint32_t>() + 4 * t_ivl, sentinel, 16, hipMemcpyHostToDevice, s));
    // comments follow the template (re-serialised the way the reference's reader / writer pair does)
    if (!in->h_comments.empty() && !(p->flags & TKSMSEQ_MOL_NO_COMMENTS)) {
        std::vector<uint32_t> mol(n_nodes);
        HIPCHK(ctx, hipMemcpyAsync(mol.data(), n_mol.p, n_nodes * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        b->h_comments.reserve(2 * n_nodes);
        uint32_t last = 0xffffffffu, off = 0, len = 0;
        for (uint64_t j = 0; j < n_nodes; j++) {
            const uint32_t u = mol[j];
            if (u != last) {
                const std::string c = normalize_comment(in->h_comment_pool.data() + in->h_comments[2 * (size_t)u], in->h_comments[2 * (size_t)u + 1], {});
                if (b->h_comment_pool.size() + c.size() >= 0xffffffffull) { ctx->err = "PCR: more than 4 GB of header comments in one batch (use template slices)"; return TKSMSEQ_ELIMIT; }
                off = (uint32_t)b->h_comment_pool.size(); len = (uint32_t)c.size();
                b->h_comment_pool.insert(b->h_comment_pool.end(), c.begin(), c.end());
                last = u;
            }
            b->h_comments.push_back(off); b->h_comments.push_back(len);
        }
    }
    if ((rc = finalize_device_batch(ctx, b.get()))) return rc;
    *out = b.release();
    return TKSMSEQ_OK;
}

int tksmseq_truncate(tksmseq_ctx* ctx, const tksmseq_batch* in, const tksmseq_trc_params* p, tksmseq_batch** out) {
    if (!ctx || !in || !p || !out) return TKSMSEQ_EINVAL;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = in->n_reads;
    tk::TrcParams T{};
    T.seed = p->seed; T.mode = p->mode; T.mu = p->mu; T.sigma = p->sigma; T.min_len = 100;      // truncate()'s default min_val
    T.always_end = p->always_end ? 1 : 0; T.models_length = p->kde_models_length ? 1 : 0;
    DevBuf d_x, d_y, d_cdf, d_rn, d_sl, d_sc;
    for (DevBuf* pb_ : {&d_x, &d_y, &d_cdf, &d_rn, &d_sl, &d_sc}) { pb_->pooled = true; pb_->pool_stream = s; }   // (per-call temporaries: DevCache, ctx.h)
    TrcModelHost tm;
    auto upv = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        HIPCHK(ctx, b.ensure(bytes + 64));
        if (bytes) HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
        return TKSMSEQ_OK;
    };
    int rc;
    if (p->mode == TKSMSEQ_TRC_KDE) {
        if (!p->kde_model_path) { ctx->err = "truncate: the KDE mode needs a model file"; return TKSMSEQ_EINVAL; }
        if (!load_trc_model(p->kde_model_path, tm, ctx->err)) return TKSMSEQ_EIO;
        if (!p->always_end && !tm.have_sider) { ctx->err = "truncate: the model has no end_mtx (use --always-end)"; return TKSMSEQ_EINVAL; }
        T.nx = (int)tm.xlab.size(); T.ny = (int)tm.ylab.size(); T.have_sider = tm.have_sider ? 1 : 0; T.ns = (int)tm.slab.size();
        if ((rc = upv(d_x, tm.xlab.data(), tm.xlab.size() * 8)) || (rc = upv(d_y, tm.ylab.data(), tm.ylab.size() * 8)) ||
            (rc = upv(d_cdf, tm.cdf.data(), tm.cdf.size() * 8)) || (rc = upv(d_rn, tm.row_n.data(), tm.row_n.size() * 4)) ||
            (rc = upv(d_sl, tm.slab.data(), tm.slab.size() * 8)) || (rc = upv(d_sc, tm.scdf.data(), tm.scdf.size() * 8))) return rc;
        T.xlab = d_x.as<long long>(); T.ylab = d_y.as<long long>(); T.cdf = d_cdf.as<double>(); T.row_n = d_rn.as<int>();
        T.slab = d_sl.as<double>(); T.scdf = d_sc.as<double>();
    } else if (p->mode != TKSMSEQ_TRC_NORMAL && p->mode != TKSMSEQ_TRC_LOGNORMAL) { ctx->err = "truncate: unknown mode"; return TKSMSEQ_EINVAL; }
    else if (!std::isfinite(p->mu) || !std::isfinite(p->sigma)) { ctx->err = "truncate: mu and sigma must be finite"; return TKSMSEQ_EINVAL; }
    tk::MolView M{view_of(in), in->d_dup.p ? in->d_dup.as<uint32_t>() : nullptr, in->n_intervals, in->n_mods, nullptr, n};
    DevBuf kf, kt, tl, ts, n_ivl, n_mod, n_idl, o_ivl, o_mod, o_id;
    for (DevBuf* pb_ : {&kf, &kt, &tl, &ts, &n_ivl, &n_mod, &n_idl, &o_ivl, &o_mod, &o_id}) { pb_->pooled = true; pb_->pool_stream = s; }   // (per-call temporaries: DevCache, ctx.h)
    HIPCHK(ctx, kf.ensure(n * 4 + 16)); HIPCHK(ctx, kt.ensure(n * 4 + 16));
    for (DevBuf* b : {&tl, &ts, &n_ivl, &n_mod, &n_idl}) HIPCHK(ctx, b->ensure(n * 8 + 16));
    HIPCHK(ctx, tk::launch_trc_plan(M, T, p->first_molecule_index, kf.as<uint32_t>(), kt.as<uint32_t>(), tl.as<double>(), ts.as<double>(),
                                    n_ivl.as<uint64_t>(), n_mod.as<uint64_t>(), n_idl.as<uint64_t>(), s));
    uint64_t t_ivl = 0, t_mod = 0, t_id = 0;
    if ((rc = scan_to(ctx, n_ivl, o_ivl, n, &t_ivl)) || (rc = scan_to(ctx, n_mod, o_mod, n, &t_mod)) || (rc = scan_to(ctx, n_idl, o_id, n, &t_id))) return rc;
    std::unique_ptr<tksmseq_batch> b(new tksmseq_batch());
    b->n_reads = n; b->n_intervals = t_ivl; b->n_mods = t_mod;
    HIPCHK(ctx, b->reads.ensure(n * 8 + 64));
    HIPCHK(ctx, b->intervals.ensure((t_ivl + 1) * 16 + 64));
    HIPCHK(ctx, b->mods.ensure(t_mod * 8 + 64));
    HIPCHK(ctx, b->ids.ensure(n * 8 + 64));
    HIPCHK(ctx, b->idpool.ensure(t_id + 64));
    if ((rc = copy_literals(ctx, in, b.get()))) return rc;
    tk::MolOut O{b->reads.as<uint32_t>(), b->intervals.as<uint32_t>(), b->mods.as<uint32_t>(), b->ids.as<uint32_t>(), b->idpool.as<uint8_t>()};
    HIPCHK(ctx, tk::launch_trc_write(M, kf.as<uint32_t>(), kt.as<uint32_t>(), o_ivl.as<uint64_t>(), o_mod.as<uint64_t>(), o_id.as<uint64_t>(), O, s));
    const uint32_t sentinel[4] = {0u, 0u, 0u, (uint32_t)t_mod};
    HIPCHK(ctx, hipMemcpyAsync(b->intervals.as<uint32_t>() + 4 * t_ivl, sentinel, 16, hipMemcpyHostToDevice, s));
    // comments: the template's, plus truncated=chr:start-end,... for what was cut away and (KDE mode) TR=<length>,<3' share>
    // (src/truncate.cpp:54-60, :343).  Needs the input tables on the host.
    if (!in->h_comments.empty() && !(p->flags & TKSMSEQ_MOL_NO_COMMENTS)) {
        std::vector<uint32_t> reads(2 * n), ivs(4 * (in->n_intervals + 1)), hkf(n), hkt(n);
        std::vector<double> htl(n), hts(n);
        std::vector<uint64_t> lits(2 * in->n_literals);
        std::vector<char> lpool(in->litpool.cap);
        HIPCHK(ctx, hipMemcpyAsync(reads.data(), in->reads.p, n * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(ivs.data(), in->intervals.p, (in->n_intervals + 1) * 16, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(hkf.data(), kf.p, n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(hkt.data(), kt.p, n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(htl.data(), tl.p, n * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(hts.data(), ts.p, n * 8, hipMemcpyDeviceToHost, s));
        if (in->n_literals) HIPCHK(ctx, hipMemcpyAsync(lits.data(), in->literals.p, in->n_literals * 16, hipMemcpyDeviceToHost, s));
        if (!lpool.empty()) HIPCHK(ctx, hipMemcpyAsync(lpool.data(), in->litpool.p, lpool.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        auto chr_of = [&](uint32_t c) -> std::string {
            if (c >> 31) { const uint32_t li = c & 0x7fffffffu; return std::string(lpool.data() + lits[2 * (size_t)li], (size_t)lits[2 * (size_t)li + 1]); }
            return ctx->contig_names[c];
        };
        for (uint64_t r = 0; r < n; r++) {
            std::vector<std::pair<std::string, std::string>> extra;
            const int w0 = (int)(hkf[r] & 0x7fffffffu), w1 = (int)(hkt[r] & 0x7fffffffu);
            const bool cut5 = hkf[r] >> 31, cut3 = hkt[r] >> 31;
            const uint32_t ib = reads[2 * r], ic = reads[2 * r + 1];
            auto piece = [&](const uint32_t* iv, long long a, long long bb) { extra.push_back({"truncated", chr_of(iv[0]) + ":" + std::to_string(a) + "-" + std::to_string(bb)}); };
            if (cut3) {
                // 3' cut in segment order: the cut segment's removed part, then the dropped segments (src/truncate.cpp:41-58)
                int c = 0; bool found = false;
                for (uint32_t i = 0; i < ic; i++) {
                    const uint32_t* iv = ivs.data() + 4 * (size_t)(ib + i);
                    const int sz = iv[2] > iv[1] ? (int)(iv[2] - iv[1]) : 0;
                    if (!found && c + sz >= w1) {
                        const int keepn = w1 - c;
                        if (iv[3] >> 31) piece(iv, iv[1], (long long)iv[2] - keepn); else piece(iv, (long long)iv[1] + keepn, iv[2]);
                        found = true;
                    } else if (found) piece(iv, iv[1], iv[2]);
                    c += sz;
                }
            }
            if (cut5) {
                // 5' cut = the same on the flipped molecule (after the first cut): segments from the last kept one backwards
                int cend = 0; std::vector<std::pair<uint32_t, int>> segs;      // (interval index, start coordinate) of the once-truncated molecule
                { int c = 0; for (uint32_t i = 0; i < ic; i++) { const uint32_t* iv = ivs.data() + 4 * (size_t)(ib + i); const int sz = iv[2] > iv[1] ? (int)(iv[2] - iv[1]) : 0; if (c < w1 || (sz == 0 && !cut3)) segs.push_back({i, c}); c += sz; } cend = std::min(c, w1); }
                const int L2 = cend - w0;
                int c = 0; bool found = false;                                   // c: bases of the flipped molecule before the segment
                for (size_t q = segs.size(); q-- > 0;) {
                    const uint32_t* iv = ivs.data() + 4 * (size_t)(ib + segs[q].first);
                    long long st = iv[1], en = iv[2];
                    int sz = en > st ? (int)(en - st) : 0;
                    const bool minus = iv[3] >> 31;
                    if (segs[q].second + sz > w1) {                              // already cut at its 3' side
                        const int keepn = w1 - segs[q].second;
                        if (minus) st = en - keepn; else en = st + keepn;
                        sz = keepn;
                    }
                    if (!found && c + sz >= L2) {
                        const int keepn = L2 - c;
                        // the flipped segment has the opposite strand: flipped plus (original minus) keeps [st, st + keep)
                        if (minus) extra.push_back({"truncated", chr_of(iv[0]) + ":" + std::to_string(st + keepn) + "-" + std::to_string(en)});
                        else extra.push_back({"truncated", chr_of(iv[0]) + ":" + std::to_string(st) + "-" + std::to_string(en - keepn)});
                        found = true;
                    } else if (found) extra.push_back({"truncated", chr_of(iv[0]) + ":" + std::to_string(st) + "-" + std::to_string(en)});
                    c += sz;
                }
            }
            if (p->mode == TKSMSEQ_TRC_KDE) {
                char sr[32]; snprintf(sr, sizeof sr, "%.2f", hts[r]);
                extra.push_back({"TR", fmt_double(htl[r]) + "," + sr});
            }
            const std::string cmt = normalize_comment(in->h_comment_pool.data() + in->h_comments[2 * r], in->h_comments[2 * r + 1], extra);
            if (b->h_comment_pool.size() + cmt.size() >= 0xffffffffull) { ctx->err = "truncate: more than 4 GB of header comments in one batch (split the input)"; return TKSMSEQ_ELIMIT; }
            b->h_comments.push_back((uint32_t)b->h_comment_pool.size()); b->h_comments.push_back((uint32_t)cmt.size());
            b->h_comment_pool.insert(b->h_comment_pool.end(), cmt.begin(), cmt.end());
        }
    }
    if ((rc = finalize_device_batch(ctx, b.get()))) return rc;
    *out = b.release();
    return TKSMSEQ_OK;
}

int tksmseq_batch_to_mdf_text(tksmseq_ctx* ctx, const tksmseq_batch* b, char** text, uint64_t* len) {
    if (!ctx || !b || !text || !len) return TKSMSEQ_EINVAL;
    *text = nullptr; *len = 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = b->n_reads;
    std::vector<uint32_t> reads(2 * n), ivs(4 * (b->n_intervals + 1)), mods(2 * b->n_mods), ids(2 * n), dup;
    std::vector<uint64_t> lits(2 * b->n_literals);
    std::vector<char> lpool(b->litpool.cap), idpool(b->idpool.cap);
    HIPCHK(ctx, hipMemcpyAsync(reads.data(), b->reads.p, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(ivs.data(), b->intervals.p, (b->n_intervals + 1) * 16, hipMemcpyDeviceToHost, s));
    if (b->n_mods) HIPCHK(ctx, hipMemcpyAsync(mods.data(), b->mods.p, b->n_mods * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(ids.data(), b->ids.p, n * 8, hipMemcpyDeviceToHost, s));
    if (b->n_literals) HIPCHK(ctx, hipMemcpyAsync(lits.data(), b->literals.p, b->n_literals * 16, hipMemcpyDeviceToHost, s));
    if (!lpool.empty()) HIPCHK(ctx, hipMemcpyAsync(lpool.data(), b->litpool.p, lpool.size(), hipMemcpyDeviceToHost, s));
    if (!idpool.empty()) HIPCHK(ctx, hipMemcpyAsync(idpool.data(), b->idpool.p, idpool.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    // One molecule per read, depth 1: what every C++ module of the reference writes after reading with unroll = true
    // (copies of a depth > 1 molecule are named id_0, id_1, ...: src/mdf.h:97-105).  print_tsv: "+id<TAB>depth<TAB>comment".
    // The reads are formatted in contiguous shares, one per host thread (tksmseq_set_host_threads), and the shares copied into
    // the result side by side.
    auto put_u32 = [](std::string& o, uint32_t v) {
        char tmp[10]; int k = 10;
        do { tmp[--k] = (char)('0' + v % 10u); v /= 10u; } while (v);
        o.append(tmp + k, (size_t)(10 - k));
    };
    auto format = [&](uint64_t r0, uint64_t r1, std::string& out) {
        out.reserve((r1 - r0) * 112);
        for (uint64_t r = r0; r < r1; r++) {
            out += '+';
            out.append(idpool.data() + ids[2 * r], ids[2 * r + 1]);
            if (!b->h_dup.empty() && (b->h_dup[r] >> 31)) { out += '_'; put_u32(out, b->h_dup[r] & 0x7fffffffu); }
            out += "\t1\t";
            if (!b->h_comments.empty()) out += normalize_comment(b->h_comment_pool.data() + b->h_comments[2 * r], b->h_comments[2 * r + 1], {});
            out += '\n';
            const uint32_t ib = reads[2 * r], ic = reads[2 * r + 1];
            for (uint32_t i = 0; i < ic; i++) {
                const uint32_t* iv = ivs.data() + 4 * (size_t)(ib + i);
                if (iv[0] >> 31) { const uint32_t li = iv[0] & 0x7fffffffu; out.append(lpool.data() + lits[2 * (size_t)li], (size_t)lits[2 * (size_t)li + 1]); }
                else out += ctx->contig_names[iv[0]];
                out += '\t'; put_u32(out, iv[1]); out += '\t'; put_u32(out, iv[2]); out += '\t';
                out += (iv[3] >> 31) ? '-' : '+';
                out += '\t';
                const uint32_t mb = iv[3] & 0x7fffffffu, me = iv[7] & 0x7fffffffu;
                for (uint32_t q = mb; q < me; q++) { if (q > mb) out += ','; put_u32(out, mods[2 * (size_t)q]); out += (char)mods[2 * (size_t)q + 1]; }
                out += '\n';
            }
        }
    };
    const uint64_t nt = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, ctx->host_threads), n / 1024 + 1));
    std::vector<std::string> parts(nt);
    {
        std::vector<std::thread> th;
        for (uint64_t t = 1; t < nt; t++) th.emplace_back([&, t]() { format(n * t / nt, n * (t + 1) / nt, parts[t]); });
        format(0, n / nt, parts[0]);
        for (auto& x : th) x.join();
    }
    uint64_t total = 0;
    std::vector<uint64_t> at(nt);
    for (uint64_t t = 0; t < nt; t++) { at[t] = total; total += parts[t].size(); }
    char* buf = (char*)malloc(total + 1);
    if (!buf) return TKSMSEQ_ENOMEM;
    {
        std::vector<std::thread> th;
        for (uint64_t t = 1; t < nt; t++) th.emplace_back([&, t]() { memcpy(buf + at[t], parts[t].data(), parts[t].size()); });
        memcpy(buf, parts[0].data(), parts[0].size());
        for (auto& x : th) x.join();
    }
    buf[total] = 0;
    *text = buf; *len = total;
    return TKSMSEQ_OK;
}

void tksmseq_text_free(char* text) { free(text); }

}  // extern "C"
