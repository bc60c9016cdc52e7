// module_log.h -- what the tksm modules here share: --verbosity / --log-file (src/module.h:95-122, src/util.h:94-120) and the
// --devices list.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace tkmod {

// levels DEBUG < INFO < WARN < ERROR < OFF; the file is "stderr", "stdout" or a path
struct Logger {
    enum Level { DEBUG = 0, INFO = 1, WARN = 2, ERROR = 3, OFF = 4 };
    int level = INFO; FILE* f = stderr; bool own = false; std::mutex m; std::string module = "sequence";
    static int parse(const std::string& v) {
        static const char* names[] = {"DEBUG", "INFO", "WARN", "ERROR", "OFF"};
        for (int i = 0; i < 5; i++) if (v == names[i]) return i;
        return -1;
    }
    bool open(const std::string& path) {
        if (path == "stderr") { f = stderr; return true; }
        if (path == "stdout") { f = stdout; return true; }
        FILE* g = fopen(path.c_str(), "a");
        if (!g) return false;
        f = g; own = true;
        return true;
    }
    void log(int lv, const char* fmt, ...) __attribute__((format(printf, 3, 4))) {
        if (lv < level || level == OFF) return;
        static const char* tag[] = {"DBG", "INF", "WRN", "ERR"};
        std::lock_guard<std::mutex> l(m);
        fprintf(f, "[%s %s] ", module.c_str(), tag[lv]);
        va_list ap; va_start(ap, fmt); vfprintf(f, fmt, ap); va_end(ap);
        fputc('\n', f); fflush(f);
    }
    ~Logger() { if (own) fclose(f); }
};

// cxxopts (the C++ modules' parser, src/module.h:75-104) accepts `--opt=value` next to `--opt value`: the former is rewritten into the
// latter before a module's own loop reads the command line.  `store` owns the strings `out` points into.
inline void split_equals(int argc, char** argv, std::vector<std::string>& store, std::vector<char*>& out) {
    store.clear(); out.clear();
    for (int i = 0; i < argc; i++) {
        const std::string t = argv[i];
        const size_t eq = t.find('=');
        if (i > 0 && t.size() > 2 && t[0] == '-' && t[1] == '-' && eq != std::string::npos) { store.push_back(t.substr(0, eq)); store.push_back(t.substr(eq + 1)); }
        else store.push_back(t);
    }
    for (auto& t : store) out.push_back(const_cast<char*>(t.c_str()));
}

// "D[,D...]": one entry per device group (an entry may repeat); false: not a list of non-negative integers
inline bool parse_device_list(const char* v, std::vector<int>& out) {
    out.clear();
    const char* q = v;
    for (;;) {
        char* e = nullptr;
        const long d = strtol(q, &e, 10);
        if (e == q || d < 0 || (*e && *e != ',')) return false;
        out.push_back((int)d);
        if (!*e) return true;
        q = e + 1;
    }
}

// molecules a piece of MDF text holds once depth is unrolled = reads Seq makes of it: the depth column of every molecule header
// (mdf_generator, py/sequence.py:206-213; stream_mdf unroll, src/mdf.h:97-105)
inline uint64_t count_reads(const char* p, size_t len) {
    uint64_t n = 0;
    const char* end = p + len;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (*p == '+') {
            const char* t = (const char*)memchr(p, '\t', (size_t)(le - p));
            long long d = 0;
            if (t) { const char* q = t + 1; bool neg = false; if (q < le && *q == '-') { neg = true; q++; } while (q < le && *q >= '0' && *q <= '9') d = d * 10 + (*q++ - '0'); if (neg) d = 0; }
            n += (uint64_t)d;
        }
        p = nl ? nl + 1 : end;
    }
    return n;
}

// Reads MDF text in pieces of about `bytes` that hold whole molecules (cut at the last "\n+").  next(): false at the end of the input.
// the last molecule boundary ("\n+") of buf[0, have), searched backwards down to `floor` (positions below it are known to hold none):
// the index of the '+', or 0 if there is none
inline size_t last_molecule_boundary(const char* buf, size_t have, size_t floor) {
    size_t p = have;
    const size_t lo = floor > 1 ? floor : 1;
    while (p > lo && !(buf[p - 1] == '\n' && p < have && buf[p] == '+')) p--;
    return p > lo ? p : 0;
}

struct ChunkReader {
    FILE* in = nullptr; uint64_t bytes = 64ull << 20; std::vector<char> buf; size_t have = 0; bool eof = false;
    size_t floor = 0;                                        // buf[0, floor) holds no boundary: a molecule larger than `bytes` is scanned once, not once per refill
    bool next(std::vector<char>& out) {
        while (!eof || have) {
            buf.resize(have + bytes);
            const size_t got = eof ? 0 : fread(buf.data() + have, 1, bytes, in);
            if (got < bytes) eof = true;
            have += got;
            size_t cut = have;
            if (!eof) {
                cut = last_molecule_boundary(buf.data(), have, floor);
                if (!cut) { floor = have ? have - 1 : 0; buf.resize(have); continue; }   // no boundary yet: read more
            }
            if (cut == 0) return false;
            out.assign(buf.begin(), buf.begin() + (ptrdiff_t)cut);
            memmove(buf.data(), buf.data() + cut, have - cut);
            have -= cut;
            floor = have ? have - 1 : 0;                      // (the cut was the LAST boundary: what is left holds none)
            return true;
        }
        return false;
    }
};

}  // namespace tkmod
