// hostio.cpp -- FASTA and MDF text readers (the wire formats either side of the kernels).
//
//   read_fasta   restates generate_fasta, py/sequence.py:168-186 (name = header up to the first
//                space, lines concatenated verbatim); C++ statement of the same job: src/fasta.h:28-60
//   parse_mdf    restates mdf_generator, py/sequence.py:197-221 and apply_modifications' token
//                grammar, py/sequence.py:229-239 ("<pos><char>" comma separated); the writer side is
//                src/interval.h:898-905 (always 5 tab fields)
#include "host.h"

#include <thread>

#include <cstdlib>
#include <cstring>

namespace tkh {

bool read_fasta(const std::string& path, std::vector<FastaRecord>& out, std::string& err) {
    std::string text;
    if (!read_text_file(path, text, err)) return false;
    size_t a = 0;
    FastaRecord cur;
    bool have = false;
    while (a < text.size()) {
        size_t b = text.find('\n', a);
        if (b == std::string::npos) b = text.size();
        if (b == a) { err = "empty line in FASTA " + path; return false; }   // reference: l[0] IndexError
        if (text[a] == '>') {
            // the reference only flushes when it has collected sequence lines (py/sequence.py:178-183):
            // a header directly following a header replaces the name
            if (have && !cur.seq.empty()) { out.push_back(cur); cur = FastaRecord(); }
            size_t sp = text.find(' ', a + 1);
            if (sp == std::string::npos || sp > b) sp = b;
            cur.name = text.substr(a + 1, sp - a - 1);
            cur.seq.clear();
            have = true;
        } else {
            cur.seq.append(text, a, b - a);
        }
        a = b + 1;
    }
    out.push_back(cur);     // the reference yields the last record unconditionally (py/sequence.py:186)
    return true;
}

static bool parse_i64(const char* s, const char* e, long long& v) {
    if (s == e) return false;
    bool neg = false;
    if (*s == '-' || *s == '+') { neg = *s == '-'; s++; if (s == e) return false; }
    v = 0;
    for (; s < e; s++) {
        if (*s < '0' || *s > '9') return false;
        v = v * 10 + (*s - '0');
        if (v > (1ll << 40)) return false;
    }
    if (neg) v = -v;
    return true;
}

bool parse_mdf(const char* text, uint64_t len, const ContigLookup& contigs, BatchHost& out, std::string& err) {
    out = BatchHost();
    std::unordered_map<std::string, uint32_t> literal_ids;
    const char* p = text; const char* end = text + len;
    bool have = false;
    long long depth = 0;
    uint32_t ivl_begin = 0, id_off = 0, id_len = 0, cm_off = 0, cm_len = 0;
    uint64_t line_no = 0;
    auto flush = [&]() {
        if (!have) return;
        const uint32_t cnt = (uint32_t)(out.intervals.size() / 4) - ivl_begin;
        for (long long d = 0; d < depth; d++) {
            out.reads.push_back(ivl_begin); out.reads.push_back(cnt);
            out.ids.push_back(id_off); out.ids.push_back(id_len);
            out.dup.push_back(depth > 1 ? (0x80000000u | (uint32_t)d) : 0u);
            out.comments.push_back(cm_off); out.comments.push_back(cm_len);
        }
    };
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        line_no++;
        if (le == p) { err = "MDF line " + std::to_string(line_no) + ": empty line"; return false; }
        // split on tabs
        const char* f[8]; const char* fe[8]; int nf = 0;
        const char* s = p;
        for (;;) {
            const char* t = (const char*)memchr(s, '\t', (size_t)(le - s));
            if (nf < 8) { f[nf] = s; fe[nf] = t ? t : le; }
            nf++;
            if (!t) break;
            s = t + 1;
        }
        if (*p == '+') {
            flush();
            if (nf < 2 || !parse_i64(f[1], fe[1], depth)) { err = "MDF line " + std::to_string(line_no) + ": bad molecule header"; return false; }
            have = true;
            ivl_begin = (uint32_t)(out.intervals.size() / 4);
            id_off = (uint32_t)out.id_pool.size(); id_len = (uint32_t)(fe[0] - f[0] - 1);
            out.id_pool.insert(out.id_pool.end(), f[0] + 1, fe[0]);
            cm_off = (uint32_t)out.comment_pool.size(); cm_len = nf >= 3 ? (uint32_t)(fe[2] - f[2]) : 0u;
            if (nf >= 3) out.comment_pool.insert(out.comment_pool.end(), f[2], fe[2]);
        } else {
            if (nf != 5) { err = "MDF line " + std::to_string(line_no) + ": interval lines need exactly 5 tab-separated fields"; return false; }
            if (!have) { err = "MDF line " + std::to_string(line_no) + ": interval before the first molecule header"; return false; }
            long long st, en;
            if (!parse_i64(f[1], fe[1], st) || !parse_i64(f[2], fe[2], en)) { err = "MDF line " + std::to_string(line_no) + ": bad interval coordinates"; return false; }
            if (st < 0 || en < 0 || st > 0xFFFFFFFFll || en > 0xFFFFFFFFll) { err = "MDF line " + std::to_string(line_no) + ": interval coordinate out of the supported range"; return false; }
            std::string name(f[0], fe[0]);
            uint32_t contig;
            int cid = contigs.find(name);
            if (cid >= 0) contig = (uint32_t)cid;
            else {
                auto it = literal_ids.find(name);
                if (it == literal_ids.end()) {
                    uint32_t li = (uint32_t)(out.literals.size() / 2);
                    out.literals.push_back(out.literal_pool.size()); out.literals.push_back(name.size());
                    out.literal_pool.insert(out.literal_pool.end(), name.begin(), name.end());
                    it = literal_ids.emplace(name, li).first;
                }
                contig = 0x80000000u | it->second;
            }
            const bool minus = !(fe[3] - f[3] == 1 && *f[3] == '+');   // strand == "+" else reverse complement
            const uint32_t mod_begin = (uint32_t)(out.mods.size() / 2);
            // modifications: "<pos><char>" joined by ','
            const char* m = f[4];
            if (m < fe[4]) {
                for (;;) {
                    const char* c = (const char*)memchr(m, ',', (size_t)(fe[4] - m));
                    const char* te = c ? c : fe[4];
                    long long pos;
                    if (te - m < 2 || !parse_i64(m, te - 1, pos) || pos < 0) { err = "MDF line " + std::to_string(line_no) + ": bad modification token"; return false; }
                    out.mods.push_back((uint32_t)pos); out.mods.push_back((uint8_t)te[-1]);
                    if (!c) break;
                    m = c + 1;
                }
            }
            out.intervals.push_back(contig); out.intervals.push_back((uint32_t)st); out.intervals.push_back((uint32_t)en);
            out.intervals.push_back(mod_begin | (minus ? 0x80000000u : 0u));
        }
        p = nl ? nl + 1 : end;
    }
    flush();
    if (out.id_pool.size() >= 0xffffffffull || out.comment_pool.size() >= 0xffffffffull) { err = "batch too large (split it: < 4 GB of ids / header comments per batch)"; return false; }
    return true;
}

// parse_mdf on several threads: the text is cut at molecule headers into one piece per thread, the pieces are parsed
// side by side and their tables appended with the offsets fixed up.  A literal contig that occurs in two pieces is kept
// twice (harmless); error messages count lines from the start of the whole text.
bool parse_mdf_mt(const char* text, uint64_t len, const ContigLookup& contigs, BatchHost& out, std::string& err, int n_threads) {
    if (n_threads <= 1 || len < (1u << 20)) return parse_mdf(text, len, contigs, out, err);
    std::vector<uint64_t> cut{0};
    for (int t = 1; t < n_threads; t++) {
        uint64_t at = len * (uint64_t)t / (uint64_t)n_threads;
        if (at <= cut.back()) continue;
        const char* q = text + at;
        const char* end = text + len;
        while (q < end) {                                         // the next line that starts a molecule
            const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q));
            if (!nl || nl + 1 >= end) { q = end; break; }
            q = nl + 1;
            if (*q == '+') break;
        }
        if (q < end && (uint64_t)(q - text) > cut.back()) cut.push_back((uint64_t)(q - text));
    }
    cut.push_back(len);
    const size_t np = cut.size() - 1;
    if (np <= 1) return parse_mdf(text, len, contigs, out, err);
    std::vector<BatchHost> parts(np);
    std::vector<std::string> errs(np);
    std::vector<char> ok(np, 0);
    std::vector<std::thread> th;
    for (size_t i = 0; i < np; i++)
        th.emplace_back([&, i]() { ok[i] = parse_mdf(text + cut[i], cut[i + 1] - cut[i], contigs, parts[i], errs[i]) ? 1 : 0; });
    for (auto& t : th) t.join();
    for (size_t i = 0; i < np; i++)
        if (!ok[i]) {
            // the piece counted its own lines: add the lines before it
            uint64_t before = 0;
            for (const char* q = text; q < text + cut[i]; q++) before += *q == '\n';
            err = errs[i];
            const std::string tag = "MDF line ";
            if (err.compare(0, tag.size(), tag) == 0) {
                size_t e = tag.size();
                unsigned long long ln = 0;
                while (e < err.size() && err[e] >= '0' && err[e] <= '9') ln = ln * 10 + (unsigned)(err[e++] - '0');
                err = tag + std::to_string(ln + before) + err.substr(e);
            }
            return false;
        }
    out = BatchHost();
    size_t nr = 0, ni = 0, nm = 0, nl = 0, nlp = 0, nip = 0, ncp = 0;
    for (auto& b : parts) { nr += b.reads.size(); ni += b.intervals.size(); nm += b.mods.size(); nl += b.literals.size(); nlp += b.literal_pool.size(); nip += b.id_pool.size(); ncp += b.comment_pool.size(); }
    // (ids and header comments are addressed with 32-bit offsets)
    if (ni / 4 >= 0x7fffffffull || nm / 2 >= 0x7fffffffull || nip >= 0xffffffffull || ncp >= 0xffffffffull) { err = "batch too large (split it: < 2^31 intervals/mods, < 4 GB of ids / header comments per batch)"; return false; }
    out.reads.reserve(nr); out.ids.reserve(nr); out.intervals.reserve(ni); out.mods.reserve(nm); out.literals.reserve(nl);
    out.literal_pool.reserve(nlp); out.id_pool.reserve(nip);
    for (auto& b : parts) {
        const uint32_t io = (uint32_t)(out.intervals.size() / 4), mo = (uint32_t)(out.mods.size() / 2), lo = (uint32_t)(out.literals.size() / 2);
        const uint64_t lpo = out.literal_pool.size();
        const uint32_t ido = (uint32_t)out.id_pool.size();
        for (size_t q = 0; q < b.reads.size(); q += 2) { out.reads.push_back(b.reads[q] + io); out.reads.push_back(b.reads[q + 1]); }
        for (size_t q = 0; q < b.ids.size(); q += 2) { out.ids.push_back(b.ids[q] + ido); out.ids.push_back(b.ids[q + 1]); }
        const uint32_t cmo = (uint32_t)out.comment_pool.size();
        for (size_t q = 0; q < b.comments.size(); q += 2) { out.comments.push_back(b.comments[q] + cmo); out.comments.push_back(b.comments[q + 1]); }
        out.comment_pool.insert(out.comment_pool.end(), b.comment_pool.begin(), b.comment_pool.end());
        out.dup.insert(out.dup.end(), b.dup.begin(), b.dup.end());
        for (size_t q = 0; q < b.intervals.size(); q += 4) {
            const uint32_t c = b.intervals[q];
            out.intervals.push_back((c >> 31) ? (0x80000000u | ((c & 0x7fffffffu) + lo)) : c);
            out.intervals.push_back(b.intervals[q + 1]); out.intervals.push_back(b.intervals[q + 2]);
            const uint32_t w = b.intervals[q + 3];
            out.intervals.push_back((w & 0x80000000u) | ((w & 0x7fffffffu) + mo));
        }
        out.mods.insert(out.mods.end(), b.mods.begin(), b.mods.end());
        for (size_t q = 0; q < b.literals.size(); q += 2) { out.literals.push_back(b.literals[q] + lpo); out.literals.push_back(b.literals[q + 1]); }
        out.literal_pool.insert(out.literal_pool.end(), b.literal_pool.begin(), b.literal_pool.end());
        out.id_pool.insert(out.id_pool.end(), b.id_pool.begin(), b.id_pool.end());
        b = BatchHost();
    }
    return true;
}

}  // namespace tkh
