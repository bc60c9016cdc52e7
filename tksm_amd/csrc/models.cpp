// models.cpp -- host-side loaders for the Badread model files and the identity distribution.
//
// Restates (file:line into vpc-ccg/tksm):
//   ErrorModel.load_from_file            py/tksm_badread.py:91-117
//   align_kmers                          py/tksm_badread.py:146-197  (edlib -> unit-cost NW with
//                                        path; traceback prefers query-only, target-only, diagonal)
//   QScoreModel.load_from_file/random/ideal   py/tksm_badread.py:487-582
//   Identities / beta_parameters         py/tksm_badread.py:703-757
//   KDE_noise_generator.load, Custom2Dist / CustomDist tables   py/tksm_badread.py:944-962, :975-1021
//   model-name lookup through $TKSM_MODELS    py/sequence.py:17-31, src/sequence.cpp:38-52
// and produces the packed device layouts described in DESIGN.md (identical, bit for bit, to the
// tables oracle/pyoracle.py builds independently).
#include "host.h"

#include <zlib.h>
#include <dlfcn.h>
#include <sys/stat.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>

namespace tkh {

static bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

// reads a whole (optionally gzip-compressed) text file; gzread handles plain files transparently
bool read_text_file(const std::string& path, std::string& out, std::string& err) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    gzbuffer(f, 1 << 20);
    out.clear();
    std::vector<char> buf(1 << 22);
    for (;;) {
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { err = "read error in " + path; gzclose(f); return false; }
        if (n == 0) break;
        out.append(buf.data(), (size_t)n);
    }
    gzclose(f);
    return true;
}

static std::string library_dir() {
    Dl_info info;
    if (dladdr((void*)&library_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.rfind('/');
        return s == std::string::npos ? "." : p.substr(0, s);
    }
    return ".";
}

// <dir>/badread/<name>.<kind>.gz, first hit in [built-in models dir, $TKSM_MODELS...] wins
// (src/sequence.cpp:38-52 prepends the compile-time path; py/sequence.py:21 lets earlier dirs shadow later).
std::string resolve_model(const std::string& name, const char* kind) {
    if (file_exists(name)) return name;
    std::vector<std::string> dirs;
    if (const char* bi = getenv("TKSMSEQ_BUILTIN_MODELS")) dirs.push_back(bi);
    dirs.push_back(library_dir() + "/../models");
    dirs.push_back(library_dir() + "/models");
    if (const char* env = getenv("TKSM_MODELS")) {
        std::string v = env; size_t a = 0;
        while (a <= v.size()) { size_t b = v.find(':', a); if (b == std::string::npos) b = v.size(); if (b > a) dirs.push_back(v.substr(a, b - a)); a = b + 1; }
    }
    for (auto& d : dirs) {
        std::string p = d + "/badread/" + name + "." + kind + ".gz";
        if (file_exists(p)) return p;
    }
    return name;   // fall through: treated as a path (the reference does the same, py/tksm_badread.py:88-89)
}

bool model_available(const std::string& name, const char* kind) { return file_exists(resolve_model(name, kind)); }

// ---------------------------------------------------------------------------------------------
// tiny global alignment with path for align_kmers (strings of length <= ~12)
// ---------------------------------------------------------------------------------------------
static std::string nw_ops(const std::string& q, const std::string& t) {
    const int n = (int)q.size(), m = (int)t.size();
    std::vector<int> H((size_t)(n + 1) * (m + 1));
    auto at = [&](int i, int j) -> int& { return H[(size_t)i * (m + 1) + j]; };
    for (int j = 0; j <= m; j++) at(0, j) = j;
    for (int i = 1; i <= n; i++) {
        at(i, 0) = i;
        for (int j = 1; j <= m; j++) {
            int d = at(i - 1, j - 1) + (q[i - 1] != t[j - 1]), u = at(i - 1, j) + 1, l = at(i, j - 1) + 1;
            at(i, j) = std::min(d, std::min(u, l));
        }
    }
    std::string ops;
    int i = n, j = m;
    while (i > 0 || j > 0) {
        int cur = at(i, j);
        if (i > 0 && at(i - 1, j) + 1 == cur) { ops.push_back('I'); i--; }
        else if (j > 0 && at(i, j - 1) + 1 == cur) { ops.push_back('D'); j--; }
        else { ops.push_back(at(i - 1, j - 1) == cur ? '=' : 'X'); i--; j--; }
    }
    return std::string(ops.rbegin(), ops.rend());
}

static bool align_kmers(const std::string& kmer_in, const std::string& alt_in, std::vector<std::string>& result) {
    if (kmer_in.size() <= 2 || alt_in.size() <= 1) return false;
    if (kmer_in.front() != alt_in.front() || kmer_in.back() != alt_in.back()) return false;
    const size_t k = kmer_in.size();
    result.assign(k, std::string());
    result[0] = std::string(1, kmer_in.front());
    result[k - 1] = std::string(1, kmer_in.back());
    std::vector<bool> set(k, false);
    std::string kmer = kmer_in.substr(1, k - 2), alt = alt_in.substr(1, alt_in.size() - 2);
    std::string ops = alt.empty() ? std::string(kmer.size(), 'D') : nw_ops(alt, kmer);
    size_t kp = 0, ap = 0;
    for (char c : ops) {
        if (c == '=' || c == 'X') { result[kp + 1] = std::string(1, alt[ap]); ap++; kp++; }
        else if (c == 'D') { result[kp + 1] = ""; kp++; }
        else { result[kp] += alt[ap]; ap++; }
    }
    if (result[0].size() == 2) {           // insertion landed on the first base: shift it to the second
        std::string ins(1, result[0][1]);
        result[0] = result[0].substr(0, 1);
        result[1] = ins + result[1];
    }
    return true;
}

static int base_code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

void cdf_thresholds(const std::vector<double>& probs, bool residual_to_one, std::vector<uint32_t>& out) {
    double s = 0.0;
    for (double p : probs) s += p;
    const double total = (residual_to_one && s < 1.0) ? 1.0 : s;
    double cum = 0.0;
    for (double p : probs) {
        cum += p;
        double x = cum / total * 4294967296.0;
        out.push_back(x >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)x);
    }
}

static void split(const std::string& s, char sep, std::vector<std::string>& out) {
    out.clear();
    size_t a = 0;
    for (;;) { size_t b = s.find(sep, a); if (b == std::string::npos) { out.push_back(s.substr(a)); break; } out.push_back(s.substr(a, b - a)); a = b + 1; }
}
static std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

static bool parse_error_model(const std::string& name_or_path, ErrorModelHost& m, std::string& err) {
    m = ErrorModelHost();
    if (name_or_path == "random") {        // py/tksm_badread.py:80-83
        m.type = 0; m.k = 1; m.max_alts = 1;
        m.cdf.assign(4, 0); m.alts.assign(4, 0); m.nalts.assign(4, 0);
        return true;
    }
    std::string text;
    if (!read_text_file(resolve_model(name_or_path, "error"), text, err)) return false;
    struct Row { std::string kmer; std::vector<uint64_t> alts; std::vector<uint32_t> thr; };
    std::vector<Row> rows;
    size_t a = 0; int k = -1; size_t A = 0;
    std::vector<std::string> parts, kv, slots;
    while (a < text.size()) {
        size_t b = text.find('\n', a);
        if (b == std::string::npos) b = text.size();
        std::string line = strip(text.substr(a, b - a));
        a = b + 1;
        if (line.empty()) continue;
        split(line, ';', parts);
        Row row; std::vector<double> probs;
        for (auto& part : parts) {
            if (part.empty()) continue;
            split(part, ',', kv);
            if (kv.size() < 2) { err = "malformed error model line"; return false; }
            if (row.kmer.empty() && row.alts.empty()) {
                row.kmer = kv[0];
                if (k < 0) k = (int)row.kmer.size();
                if ((int)row.kmer.size() != k) { err = "error model k-mers differ in size"; return false; }
                if (k > 8) { err = "error model k-mer longer than 8 is not supported"; return false; }
            }
            if (!align_kmers(row.kmer, kv[0], slots)) { err = "cannot align alternative " + kv[0] + " to " + row.kmer; return false; }
            uint64_t v = 0; int nbases = 0;
            for (size_t j = 0; j < slots.size(); j++) {
                if (slots[j].size() > 5) { err = "alternative slot longer than 5 bases (the slot codes hold 5 symbols)"; return false; }
                v |= (uint64_t)slots[j].size() << (3 * j);
                for (char ch : slots[j]) {
                    int c = base_code(ch);
                    if (c < 0 || nbases >= 19) { err = "alternative not representable: " + kv[0]; return false; }
                    v |= (uint64_t)c << (24 + 2 * nbases); nbases++;
                }
            }
            if (kv[0] == row.kmer) v |= 1ull << 63;
            row.alts.push_back(v);
            char* endp = nullptr;
            probs.push_back(strtod(kv[1].c_str(), &endp));
        }
        if (row.alts.empty()) continue;
        cdf_thresholds(probs, true, row.thr);
        A = std::max(A, row.alts.size());
        rows.push_back(std::move(row));
    }
    if (k < 3) { err = "error model has no usable k-mers"; return false; }
    if (A > 255) { err = "too many alternatives per k-mer"; return false; }
    const size_t n = (size_t)1 << (2 * k);
    m.type = 1; m.k = k; m.max_alts = (int)A;
    m.cdf.assign(n * A, 0); m.alts.assign(n * A, 0); m.nalts.assign(n, 0);
    for (auto& row : rows) {
        size_t idx = 0;
        for (char ch : row.kmer) { int c = base_code(ch); if (c < 0) { err = "non-ACGT k-mer in error model"; return false; } idx = idx * 4 + (size_t)c; }
        m.nalts[idx] = (uint8_t)row.alts.size();
        for (size_t j = 0; j < A; j++) {
            m.alts[idx * A + j] = j < row.alts.size() ? row.alts[j] : 0;
            m.cdf[idx * A + j] = j < row.thr.size() ? row.thr[j] : row.thr.back();
        }
    }
    return true;
}

static uint64_t qs_hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

static bool encode_key(const std::string& cigar, uint64_t& key) {
    if (cigar.empty() || cigar.size() > 29) return false;
    key = 0;
    for (size_t i = 0; i < cigar.size(); i++) {
        int op = cigar[i] == '=' ? 0 : cigar[i] == 'X' ? 1 : cigar[i] == 'I' ? 2 : cigar[i] == 'D' ? 3 : -1;
        if (op < 0) return false;
        key |= (uint64_t)op << (2 * i);
    }
    key |= (uint64_t)cigar.size() << 58;
    return true;
}

static bool parse_qscore_model(const std::string& name_or_path, QScoreModelHost& m, std::string& err) {
    m = QScoreModelHost();
    struct Row { std::string cigar; std::vector<int> scores; std::vector<double> probs; };
    std::vector<Row> rows;
    m.kmer_size = 1;
    auto uniform = [&](const std::string& c, int lo, int hi) {
        Row r; r.cigar = c;
        int cnt = hi - lo + 1;
        for (int q = lo; q <= hi; q++) { r.scores.push_back(q); r.probs.push_back(1.0 / cnt); }
        rows.push_back(r);
    };
    if (name_or_path == "random") {          // py/tksm_badread.py:487-497
        uniform("=", 1, 20); uniform("X", 1, 20); uniform("I", 1, 20);
    } else if (name_or_path == "ideal") {    // py/tksm_badread.py:499-544
        m.kmer_size = 9;
        uniform("X", 1, 3); uniform("I", 1, 3); uniform("=", 4, 7); uniform("===", 8, 20); uniform("=====", 21, 30);
        uniform("=======", 31, 40); uniform("=========", 41, 50);
    } else {
        std::string text;
        if (!read_text_file(resolve_model(name_or_path, "qscore"), text, err)) return false;
        size_t a = 0;
        std::vector<std::string> parts, sp, qp;
        while (a < text.size()) {
            size_t b = text.find('\n', a);
            if (b == std::string::npos) b = text.size();
            std::string line = strip(text.substr(a, b - a));
            a = b + 1;
            if (line.empty()) continue;
            split(line, ';', parts);
            if (parts[0] == "overall") continue;
            if (parts.size() < 3) { err = name_or_path + " does not seem to be a valid qscore model file"; return false; }
            Row r; r.cigar = parts[0];
            int kk = 0;
            for (char c : r.cigar) kk += c != 'D';
            m.kmer_size = std::max(m.kmer_size, kk);
            split(parts[2], ',', sp);
            for (auto& x : sp) {
                if (x.empty()) continue;
                split(x, ':', qp);
                if (qp.size() < 2) { err = name_or_path + " does not seem to be a valid qscore model file"; return false; }
                r.scores.push_back(atoi(qp[0].c_str()));
                r.probs.push_back(strtod(qp[1].c_str(), nullptr));
            }
            // a later line with the same cigar replaces the earlier one (dict semantics)
            bool replaced = false;
            for (auto& e : rows) if (e.cigar == r.cigar) { e = r; replaced = true; break; }
            if (!replaced) rows.push_back(std::move(r));
        }
    }
    bool has_eq = false, has_x = false, has_i = false;
    for (auto& r : rows) { has_eq |= r.cigar == "="; has_x |= r.cigar == "X"; has_i |= r.cigar == "I"; }
    if (!(has_eq && has_x && has_i)) { err = "qscore model lacks one of the 1-mer cigars =, X, I"; return false; }
    size_t n_slots = 1;
    while (n_slots < 2 * rows.size()) n_slots *= 2;
    m.n_slots = (int)n_slots;
    m.keys.assign(n_slots, 0); m.row_off.assign(n_slots, 0); m.row_cnt.assign(n_slots, 0);
    for (auto& r : rows) {
        uint64_t key;
        if (!encode_key(r.cigar, key)) { err = "q-score key not representable (longer than 29 ops): " + r.cigar; return false; }
        size_t s = (size_t)(qs_hash(key) & (n_slots - 1));
        while (m.keys[s] != 0) s = (s + 1) & (n_slots - 1);
        m.keys[s] = key;
        m.row_off[s] = (uint32_t)m.q_pool.size();
        m.row_cnt[s] = (uint32_t)r.scores.size();
        cdf_thresholds(r.probs, false, m.cdf_pool);
        for (int q : r.scores) m.q_pool.push_back((uint8_t)q);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// tail-noise model: JSON {lx, ly, grid, begin, trans, ratio, bases} as KDE_noise_generator.save writes it
// (py/tksm_badread.py:935-942); `begin` is read by the reference and never used (:912, :927)
// ---------------------------------------------------------------------------------------------
namespace {
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0.0; bool b = false; std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* key) const { for (auto& kv : obj) if (kv.first == key) return &kv.second; return nullptr; }
};
struct JParser {
    const char* p; const char* e; std::string err;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool str(std::string& out) {
        if (p >= e || *p != '"') return fail("expected a string");
        p++; out.clear();
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) return fail("bad escape");
                switch (*p) { case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                              case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                              case 'u': if (e - p < 5) return fail("bad escape"); out += (char)strtol(std::string(p + 1, 4).c_str(), nullptr, 16); p += 4; break;
                              default: out += *p; }
                p++;
            } else out += *p++;
        }
        if (p >= e) return fail("unterminated string");
        p++;
        return true;
    }
    bool value(JVal& v, int depth) {
        if (depth > 32) return fail("nesting too deep");
        ws();
        if (p >= e) return fail("unexpected end");
        if (*p == '{') {
            v.kind = JVal::Obj; p++; ws();
            if (p < e && *p == '}') { p++; return true; }
            for (;;) {
                ws(); std::string k; if (!str(k)) return false;
                ws(); if (p >= e || *p != ':') return fail("expected ':'");
                p++;
                v.obj.emplace_back(k, JVal());
                if (!value(v.obj.back().second, depth + 1)) return false;
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = JVal::Arr; p++; ws();
            if (p < e && *p == ']') { p++; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back(), depth + 1)) return false;
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v.kind = JVal::Str; return str(v.str); }
        if (e - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JVal::Bool; v.b = true; p += 4; return true; }
        if (e - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JVal::Bool; p += 5; return true; }
        if (e - p >= 4 && !strncmp(p, "null", 4)) { p += 4; return true; }
        if (e - p >= 3 && !strncmp(p, "NaN", 3)) { v.kind = JVal::Num; v.num = NAN; p += 3; return true; }
        char* end = nullptr;
        v.num = strtod(p, &end);                       // the text is NUL-terminated (std::string)
        if (end == p) return fail("unexpected character");
        v.kind = JVal::Num; p = end;
        return true;
    }
};
bool num_list(const JVal* v, std::vector<double>& out) {
    if (!v || v->kind != JVal::Arr) return false;
    for (auto& x : v->arr) { if (x.kind != JVal::Num) return false; out.push_back(x.num); }
    return true;
}
}  // namespace

bool make_tail_model(const double* lx, size_t n_lx, const double* ly, size_t n_ly, const double* grid, const double* trans16,
                     double ratio, const uint8_t* bases4, TailModelHost& m, std::string& err) {
    if (!n_lx || !n_ly) { err = "tail model: empty lx or ly"; return false; }
    if (n_lx > (1u << 24) || n_ly > (1u << 24)) { err = "tail model: grid too large"; return false; }
    m = TailModelHost();
    m.lx.assign(lx, lx + n_lx); m.ly.assign(ly, ly + n_ly);
    for (size_t i = 0; i < n_lx; i++) if (!std::isfinite(lx[i])) { err = "tail model: lx is not finite"; return false; }
    for (size_t i = 0; i < n_ly; i++)
        if (!std::isfinite(ly[i]) || (i && ly[i] < ly[i - 1])) { err = "tail model: ly must be finite and sorted (np.searchsorted)"; return false; }
    if (!(ly[n_ly - 1] != 0.0)) { err = "tail model: last ly label is zero"; return false; }
    m.cdf.resize(n_lx * n_ly);
    for (size_t r = 0; r < n_ly; r++) {
        const double* pdf = grid + r * n_lx;
        double sum = 0.0;
        for (size_t i = 0; i < n_lx; i++) {
            if (!(pdf[i] >= 0.0) || !std::isfinite(pdf[i])) { err = "tail model: grid entries must be finite and >= 0"; return false; }
            sum += pdf[i];
        }
        if (!(sum > 0.0)) { err = "tail model: a grid row sums to zero"; return false; }
        double c = 0.0;
        for (size_t i = 0; i < n_lx; i++) { c = pdf[i] / sum + c; m.cdf[r * n_lx + i] = c; }
    }
    for (int s = 0; s < 4; s++) {
        double c = 0.0;
        for (int j = 0; j < 4; j++) {
            const double w = trans16[4 * s + j];
            if (!(w >= 0.0) || !std::isfinite(w)) { err = "tail model: transition weights must be finite and >= 0"; return false; }
            c += w; m.cum[4 * s + j] = c;
        }
        if (!(c > 0.0)) { err = "tail model: Total of weights must be greater than zero"; return false; }   // random.choices
    }
    if (!std::isfinite(ratio)) { err = "tail model: ratio is not finite"; return false; }
    m.ratio = ratio;
    memcpy(m.bases, bases4, 4);
    m.enabled = true;
    return true;
}

bool load_tail_model(const std::string& name_or_path, TailModelHost& m, std::string& err) {
    if (name_or_path == "no_noise") { m = TailModelHost(); return true; }      // Mock_noise_generator, :964-972
    std::string text;
    if (!read_text_file(resolve_model(name_or_path, "tail"), text, err)) return false;
    JParser jp{text.c_str(), text.c_str() + text.size(), {}};
    JVal root;
    if (!jp.value(root, 0) || root.kind != JVal::Obj) { err = "tail model: not a JSON object" + (jp.err.empty() ? std::string() : " (" + jp.err + ")"); return false; }
    std::vector<double> lx, ly, grid, trans;
    if (!num_list(root.get("lx"), lx) || !num_list(root.get("ly"), ly)) { err = "tail model: lx / ly missing or not numeric lists"; return false; }
    const JVal* g = root.get("grid");
    if (!g || g->kind != JVal::Arr || g->arr.size() != ly.size()) { err = "tail model: grid must hold one row per ly label"; return false; }
    for (auto& row : g->arr) {
        const size_t before = grid.size();
        if (!num_list(&row, grid) || grid.size() - before != lx.size()) { err = "tail model: grid rows must hold one number per lx entry"; return false; }
    }
    const JVal* t = root.get("trans");
    if (!t || t->kind != JVal::Arr || t->arr.size() != 4) { err = "tail model: trans must be 4 x 4"; return false; }
    for (auto& row : t->arr) {
        const size_t before = trans.size();
        if (!num_list(&row, trans) || trans.size() - before != 4) { err = "tail model: trans must be 4 x 4"; return false; }
    }
    const JVal* r = root.get("ratio");
    if (!r || r->kind != JVal::Num) { err = "tail model: ratio missing"; return false; }
    const JVal* bs = root.get("bases");
    uint8_t bases[4];
    if (!bs || bs->kind != JVal::Arr || bs->arr.size() != 4) { err = "tail model: bases must list 4 symbols"; return false; }
    for (int i = 0; i < 4; i++) {
        if (bs->arr[i].kind != JVal::Str || bs->arr[i].str.size() != 1) { err = "tail model: bases must be single characters"; return false; }
        bases[i] = (uint8_t)bs->arr[i].str[0];
    }
    if (!root.get("begin")) { err = "tail model: begin missing"; return false; }               // KeyError in the reference (:958)
    return make_tail_model(lx.data(), lx.size(), ly.data(), ly.size(), grid.data(), trans.data(), r->num, bases, m, err);
}

// ---------------------------------------------------------------------------------------------
// truncation model: JSON list [{name: "KDE_mtx", shape: [w, h], data: [...], labels: [x labels..., y labels...]},
// {name: "end_mtx", data, labels}] (py/truncate_kde.py:298-320), read the way custom_distribution2D / custom_distribution do
// (src/truncate.cpp:88-101, :163-177): row i of the 2-D model uses the first i + 1 of its w entries; cumulative sums start at 0
// ---------------------------------------------------------------------------------------------
bool load_trc_model(const std::string& path, TrcModelHost& m, std::string& err) {
    std::string text;
    if (!read_text_file(path, text, err)) return false;
    JParser jp{text.c_str(), text.c_str() + text.size(), {}};
    JVal root;
    if (!jp.value(root, 0) || root.kind != JVal::Arr) { err = "truncation model: not a JSON list" + (jp.err.empty() ? std::string() : " (" + jp.err + ")"); return false; }
    m = TrcModelHost();
    bool have_kde = false;
    for (auto& part : root.arr) {
        if (part.kind != JVal::Obj) continue;
        const JVal* name = part.get("name");
        if (!name || name->kind != JVal::Str) continue;
        std::vector<double> data, labels;
        if (!num_list(part.get("data"), data) || !num_list(part.get("labels"), labels)) { err = "truncation model: data / labels missing in " + name->str; return false; }
        if (name->str == "KDE_mtx" && !have_kde) {
            std::vector<double> shape;
            if (!num_list(part.get("shape"), shape) || shape.size() != 2) { err = "truncation model: KDE_mtx needs a 2-entry shape"; return false; }
            const size_t w = (size_t)shape[0], h = (size_t)shape[1];
            if (!w || !h || labels.size() < w + h || data.size() < w * h) { err = "truncation model: KDE_mtx shape, labels and data disagree"; return false; }
            for (size_t i = 0; i < w; i++) m.xlab.push_back((long long)labels[i]);
            for (size_t i = 0; i < h; i++) m.ylab.push_back((long long)labels[w + i]);
            m.cdf.assign(h * (w + 1), 0.0);
            m.row_n.resize(h);
            for (size_t i = 0; i < h; i++) {
                const size_t n = std::min(i + 1, w);
                double sum = 0.0;
                for (size_t q = 0; q < n; q++) sum += data[i * w + q];
                double* c = m.cdf.data() + i * (w + 1);
                c[0] = 0.0;
                for (size_t q = 0; q < n; q++) c[q + 1] = data[i * w + q] / sum + c[q];
                for (size_t q = n + 1; q <= w; q++) c[q] = c[n];
                m.row_n[i] = (int)n;
            }
            have_kde = true;
        } else if (name->str == "end_mtx" && !m.have_sider) {
            if (data.empty() || labels.size() < data.size()) { err = "truncation model: end_mtx data and labels disagree"; return false; }
            double sum = 0.0;
            for (double d : data) sum += d;
            m.scdf.assign(1, 0.0);
            for (double d : data) m.scdf.push_back(d / sum + m.scdf.back());
            m.slab.assign(labels.begin(), labels.begin() + (ptrdiff_t)data.size());
            m.have_sider = true;
        }
    }
    if (!have_kde) { err = "KDE matrix not found"; return false; }      // src/truncate.cpp:383-386
    return true;
}

// ---------------------------------------------------------------------------------------------
// identity distribution: max * Beta(a, b), tabulated as 65537 quantiles
// ---------------------------------------------------------------------------------------------
static double betacf(double a, double b, double x) {     // continued fraction of the incomplete beta (Lentz)
    const double tiny = 1e-300, eps = 1e-16;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0, c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 10000; m++) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        double del = d * c; h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}
static double betainc(double a, double b, double x) {    // regularized I_x(a, b)
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    double lbt = lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x);
    if (x < (a + 1.0) / (a + b + 2.0)) return exp(lbt) * betacf(a, b, x) / a;
    return 1.0 - exp(lbt) * betacf(b, a, 1.0 - x) / b;
}
static double beta_pdf(double a, double b, double x) {
    if (x <= 0.0 || x >= 1.0) return 0.0;
    return exp(lgamma(a + b) - lgamma(a) - lgamma(b) + (a - 1.0) * log(x) + (b - 1.0) * log1p(-x));
}

static bool compute_identity(double mean, double max, double stdev, IdentityHost& id, std::string& err) {
    id = IdentityHost();
    id.mean = mean / 100.0; id.stdev = stdev / 100.0; id.max_identity = max / 100.0;
    if (id.mean == id.max_identity) { id.constant = true; id.value = id.mean; return true; }
    if (id.stdev == 0.0) { id.max_identity = id.mean; id.constant = true; id.value = id.mean; return true; }
    // beta_parameters, py/tksm_badread.py:747-757 (percent units, exactly as the reference calls it)
    const double u = mean, s = stdev, m = max;
    id.beta_a = (((1 - (u / m)) / ((s / m) * (s / m))) - (m / u)) * ((u / m) * (u / m));
    id.beta_b = id.beta_a * ((m / u) - 1);
    if (id.beta_a < 0.0 || id.beta_b < 0.0) {
        err = "Error: invalid beta parameters for identity distribution - trying increasing the maximum identity or "
              "reducing the standard deviation";
        return false;
    }
    id.constant = false; id.value = id.max_identity;
    id.qtab.assign(65537, 0.0);
    const double a = id.beta_a, b = id.beta_b;
    id.qtab[0] = 0.0; id.qtab[65536] = 1.0;
    double x = 0.5 * std::min(1.0, a / (a + b));
    for (int i = 1; i < 65536; i++) {
        const double p = (double)i / 65536.0;
        double lo = id.qtab[i - 1], hi = 1.0;
        if (x <= lo || x >= hi) x = 0.5 * (lo + hi);
        for (int it = 0; it < 200; it++) {
            const double f = betainc(a, b, x) - p;
            if (f > 0) hi = x; else lo = x;
            const double d = beta_pdf(a, b, x);
            double nx = d > 0 ? x - f / d : 0.5 * (lo + hi);
            if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
            if (fabs(nx - x) <= 1e-15 * fabs(x) || hi - lo <= 1e-16) { x = nx; break; }
            x = nx;
        }
        id.qtab[i] = x;
    }
    return true;
}


// ---- parsed models are kept for the life of the process, keyed by file (path, size, modification time) or parameters: a model
// parsed once -- e.g. ahead of its use, on a thread of its own while the device is being set up and the reference packed
// (sequencer_module.cpp) -- is copied, not parsed again (0.1 - 0.2 s each for the shipped models and the identity table).
namespace {
template <class T> struct Memo {
    std::mutex m;
    std::map<std::string, std::shared_ptr<const T>> map;
    bool get(const std::string& key, T& out) {
        std::lock_guard<std::mutex> l(m);
        auto it = map.find(key);
        if (it == map.end()) return false;
        out = *it->second;
        return true;
    }
    void put(const std::string& key, const T& v) { std::lock_guard<std::mutex> l(m); map[key] = std::make_shared<const T>(v); }
};
std::string file_key(const std::string& name_or_path, const char* kind) {
    const std::string path = resolve_model(name_or_path, kind);
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return std::string();
    return path + '|' + std::to_string((long long)st.st_size) + '|' + std::to_string((long long)st.st_mtim.tv_sec) + '.' + std::to_string((long long)st.st_mtim.tv_nsec);
}
Memo<ErrorModelHost> error_models;
Memo<QScoreModelHost> qscore_models;
Memo<IdentityHost> identities;
}  // namespace

bool load_error_model(const std::string& name_or_path, ErrorModelHost& m, std::string& err) {
    const std::string key = name_or_path == "random" ? std::string() : file_key(name_or_path, "error");
    if (!key.empty() && error_models.get(key, m)) return true;
    if (!parse_error_model(name_or_path, m, err)) return false;
    if (!key.empty()) error_models.put(key, m);
    return true;
}

bool load_qscore_model(const std::string& name_or_path, QScoreModelHost& m, std::string& err) {
    const std::string key = (name_or_path == "random" || name_or_path == "ideal") ? std::string() : file_key(name_or_path, "qscore");
    if (!key.empty() && qscore_models.get(key, m)) return true;
    if (!parse_qscore_model(name_or_path, m, err)) return false;
    if (!key.empty()) qscore_models.put(key, m);
    return true;
}

bool make_identity(double mean, double max, double stdev, IdentityHost& id, std::string& err) {
    char key[96];
    snprintf(key, sizeof key, "%a|%a|%a", mean, max, stdev);
    if (identities.get(key, id)) return true;
    if (!compute_identity(mean, max, stdev, id, err)) return false;
    identities.put(key, id);
    return true;
}

}  // namespace tkh
