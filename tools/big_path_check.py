"""diagnostic: a long molecule with N bases (exact wave-wide kernel, working set in HBM: k_simulate<BIG>) against the oracle"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); os.chdir(ROOT)
long_len = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
import pyoracle as po
from tksm_amd.sequence import Sequencer
rs = np.random.RandomState(5)
big = rs.choice(np.frombuffer(b"ACGT", np.uint8), 120_000).tobytes().decode()
text = f"+longn\t1\t\nbig\t100\t{100 + long_len}\t+\t300N,301N,{long_len // 2}N\n+short\t1\t\nbig\t7\t907\t-\t\n"
s = Sequencer(0)
s.add_contig("big", big.encode())
md = os.path.join("tksm_amd", "models", "badread")
s.set_identity(84.0, 99.0, 5.5); s.load_error_model(os.path.join(md, "nanopore2020.error.gz")); s.load_qscore_model(os.path.join(md, "nanopore2020.qscore.gz"))
t = time.time(); recs = s.run(s.batch_from_mdf(text), target="badread", fastq=True, compute_qual=True, seed=11, collect_stats=True); out = recs.records(); print(f"gpu {time.time() - t:.1f} s", flush=True)
ist, dst = recs.stats()
print("status words:", [int(x) for x in ist[:, 7]], flush=True)
em = po.ErrorModel(os.path.join(md, "nanopore2020.error.gz")); qm = po.QScoreModel(os.path.join(md, "nanopore2020.qscore.gz"))
ident = po.Identities(84.0, 5.5, 99.0)
for i, (mid, ivs) in enumerate(po.mdf_generator(text.splitlines(keepends=True))):
    want, st = po.badread_record(True, 11, i, po.splice({"big": big}, ivs), ident, em, qm, True, mid)
    a, b = out[i].split(b"\n"), want.split(b"\n")
    print(mid, "header", a[0] == b[0], "sequence", a[1] == b[1], "quality", a[3] == b[3], "oracle band_fail", st.band_fail, flush=True)
    if a[3] != b[3]:
        d = [j for j in range(min(len(a[3]), len(b[3]))) if a[3][j] != b[3][j]]
        print("  quality positions differing:", len(d), "first", d[:10], "last", d[-5:], "of", len(a[3]))
