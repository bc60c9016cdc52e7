"""The spill episode of round 3 (DESIGN.md / HISTORY.md): does a k_alnf forced below its natural register count still compute the same?

    TKSMSEQ_LIB=<variant library> python tools/spill_probe.py        (GPU box; tools/spill_probe.sh builds the variants and compares)

Runs one 131 072-read bulk batch with the 14-row pass + redo list in every round (TKSMSEQ_SMALL_ALN=0) and once more with every round at
full width, prints a digest of the records and the fall-back counters of tksmseq_run_diagnostics."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (its ROCm runtime before libtksmseq.so)

torch.cuda.is_available()
from tksm_amd import synthetic  # noqa: E402
from tksm_amd.sequence import Sequencer  # noqa: E402

MODELS = os.path.join(ROOT, "tksm_amd", "models", "badread")


def main():
    rs = np.random.RandomState(11)
    lens = [4_000_000] * 4
    contigs = [rs.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() for n in lens]
    n = 131072
    for kind in ("bulk", "scrna"):
        m = synthetic.make_molecules(rs, lens, n, 1000, 200, kind=kind)
        for small_aln in ("0", str(1 << 30)):
            os.environ["TKSMSEQ_SMALL_ALN"] = small_aln
            s = Sequencer(0)
            for c, seq in enumerate(contigs):
                s.add_contig(f"chr{c + 1}", seq)
            s.set_identity(84.0, 99.0, 5.5)
            s.load_error_model(os.path.join(MODELS, "nanopore2020.error.gz"))
            s.load_qscore_model(os.path.join(MODELS, "nanopore2020.qscore.gz"))
            b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
            rec, off = s.run(b, seed=5).download()
            dg = s.run_diagnostics()
            dump = os.environ.get("SPILL_DUMP")
            if dump:                                          # reference records of this configuration (the shipped library), or a comparison with them
                path = f"{dump}_{kind}_{'0' if small_aln == '0' else 'all'}.bin"
                if os.path.exists(path):
                    ref = open(path, "rb").read()
                    roff = np.load(path + ".off.npy")
                    hd = sq = ql = ln = 0
                    first = []
                    for i in range(n):
                        a, b2 = rec[int(off[i]):int(off[i + 1])], ref[int(roff[i]):int(roff[i + 1])]
                        if a != b2:
                            la, lb = a.split(b"\n"), b2.split(b"\n")
                            hd += la[0] != lb[0]; sq += la[1] != lb[1]; ql += la[3] != lb[3]; ln += len(la[1]) != len(lb[1])
                            if len(first) < 3:
                                first.append((i, la[0][38:90], lb[0][38:90]))
                    print(f"   vs the shipped library: header differs in {hd} reads, sequence in {sq} (length in {ln}), qualities in {ql}; first: {first}")
                else:
                    open(path, "wb").write(rec)
                    np.save(path + ".off.npy", off)
            print(f"{kind} small_aln={'0' if small_aln == '0' else 'all'} sha256={hashlib.sha256(rec).hexdigest()[:16]} fallbacks={dg['fallbacks']} reasons=0x{dg['fallback_reasons']:x} "
                  f"(q-score jobs {dg['fallbacks_qscore_jobs']}, list pass {dg['fallbacks_list_pass']}) exact_kernel_reads={dg['exact_kernel_reads']} "
                  f"redone={dg['jobs_redone_full_width']}/{dg['jobs_14_row_rounds']} rounds={dg['rounds']}", flush=True)
            s.close()


if __name__ == "__main__":
    main()
