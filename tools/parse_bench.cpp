// Host-only timing of the MDF parser (hostio.cpp): whole parse_mdf_mt at several thread counts, and its parallel phase alone.
//   g++ -O2 -std=c++17 -I tksm_amd/csrc -o /tmp/parse_bench tools/parse_bench.cpp tksm_amd/csrc/hostio.cpp tksm_amd/csrc/models.cpp -lz -lpthread
//   /tmp/parse_bench mols.mdf
#include "host.h"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <thread>
using namespace tkh;
struct CL : ContigLookup {
    std::unordered_map<std::string, int> ix;
    int find(const std::string& n) const override { auto it = ix.find(n); return it == ix.end() ? -1 : it->second; }
};
static double since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary); std::stringstream ss; ss << f.rdbuf(); const std::string text = ss.str();
    CL cl; for (int c = 0; c < 64; c++) cl.ix["chr" + std::to_string(c + 1)] = c;
    for (int nt : {1, 2, 4, 8, 16}) for (int rep = 0; rep < 2; rep++) {
        BatchHost h; std::string err;
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = parse_mdf_mt(text.data(), text.size(), cl, h, err, nt);
        const double dt = since(t0);
        printf("parse_mdf_mt, %2d threads: ok=%d %.3f s, %.0f MB/s, %zu reads\n", nt, ok, dt, text.size() / dt / 1e6, h.reads.size() / 2);
    }
    for (int np : {8, 16}) {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th; std::vector<BatchHost> hs(np);
        std::vector<size_t> cuts(np + 1); cuts[0] = 0; cuts[np] = text.size();
        for (int i = 1; i < np; i++) { size_t at = text.size() * i / np; while (!(text[at - 1] == '\n' && text[at] == '+')) at++; cuts[i] = at; }
        for (int i = 0; i < np; i++) th.emplace_back([&, i]() { std::string e; parse_mdf(text.data() + cuts[i], cuts[i + 1] - cuts[i], cl, hs[i], e); });
        for (auto& t : th) t.join();
        printf("%d pieces side by side (no merge): %.3f s\n", np, since(t0));
    }
    { BatchHost h; std::string e; const auto t0 = std::chrono::steady_clock::now(); parse_mdf(text.data(), text.size() / 8, cl, h, e);
      printf("one eighth alone: %.3f s\n", since(t0)); }
    return 0;
}
