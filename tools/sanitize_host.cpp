#include "host.h"
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <sstream>
using namespace tkh;
struct CL : ContigLookup {
    std::unordered_map<std::string, int> ix;
    int find(const std::string& n) const override { auto it = ix.find(n); return it == ix.end() ? -1 : it->second; }
};
int main(int argc, char** argv) {
    CL cl; for (int c = 0; c < 4; c++) cl.ix["chr" + std::to_string(c + 1)] = c;
    std::ifstream f(argv[1], std::ios::binary); std::stringstream ss; ss << f.rdbuf(); std::string text = ss.str();
    text.resize(std::min<size_t>(text.size(), 8u << 20));
    text.resize(text.rfind("\n+") + 1);
    for (int nt : {1, 5}) { BatchHost h; std::string e; bool ok = parse_mdf_mt(text.data(), text.size(), cl, h, e, nt); printf("threads %d ok=%d reads=%zu\n", nt, ok, h.reads.size() / 2); }
    const char* bad[] = {"", "\n", "+m\t1\t\nchr1\t0\t5\t+\n", "chr1\t0\t5\t+\t\n", "+m\tx\t\n", "+m\t1\t\nchr1\t-1\t5\t+\t\n", "+m\t1\t\nchr1\t0\t5\t+\t3\n",
                         "+m\t1\t\nchr1\t0\t5\t+\t3A,\n", "+m\t2\tc=1;\nAAAA\t0\t4\t-\t0C,3G\n+n\t0\t\nchr2\t1\t2\t+\t", "+", "+\t", "+m\t99999999999999999999\t\n"};
    for (const char* b : bad) { BatchHost h; std::string e; bool ok = parse_mdf(b, strlen(b), cl, h, e); printf("ok=%d reads=%zu err=%s\n", ok, h.reads.size() / 2, e.c_str()); }
    // models
    { ErrorModelHost m; std::string e; printf("error model %d\n", load_error_model(argv[2], m, e)); }
    { QScoreModelHost m; std::string e; printf("qscore model %d\n", load_qscore_model(argv[3], m, e)); }
    { IdentityHost id; std::string e; printf("identity %d\n", make_identity(84, 99, 5.5, id, e)); }
    { std::vector<uint32_t> len = {5, 3, 70000, 3, 0}, ord; order_by_length(len, ord); printf("order %u %u %u %u %u\n", ord[0], ord[1], ord[2], ord[3], ord[4]); }
    return 0;
}
