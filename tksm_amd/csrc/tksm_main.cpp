// tksm_main.cpp -- minimal dispatcher with the reference's calling convention (src/tksm.cpp:118-200):
// `tksm sequence [args]` constructs the module with (argc - 1, argv + 1) and returns run().
// This build provides the Seq exit module and the two modules upstream of it in BASELINE config 5 (pcr, truncate); every other
// module name is reported as unknown.
#include <cstdio>
#include <cstring>

#include "../../include/tksmseq.h"
#include "sequencer_module.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s sequence [options]\n", argv[0]); return 1; }
    if (!strcmp(argv[1], "sequence")) return Sequencer_module{argc - 1, argv + 1}.run();
    if (!strcmp(argv[1], "pcr")) return tksmseq_pcr_main(argc - 1, argv + 1);
    if (!strcmp(argv[1], "truncate")) return tksmseq_truncate_main(argc - 1, argv + 1);
    if (!strcmp(argv[1], "list")) { printf("sequence\npcr\ntruncate\n"); return 0; }
    fprintf(stderr, "Unknown kisim: %s (this build provides `sequence`, `pcr` and `truncate`)\n", argv[1]);
    return 1;
}
