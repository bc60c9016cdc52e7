"""diagnostic: a batch of many short molecules and one very long one (the state rows are ragged: it must cost neither the HBM of
a batch of long molecules nor wrong records) -- timing, memory, and the oracle's records for the outlier and a sample"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); os.chdir(ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
long_len = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
import pyoracle as po
import torch
from tksm_amd import synthetic
from tksm_amd.sequence import Sequencer
rs = np.random.RandomState(3)
lens = [2_000_000] * 4
contigs = [rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes() for L in lens]
m = synthetic.make_molecules(rs, lens, n, 1000, 200)
names = [f"chr{c+1}" for c in range(4)]
text = synthetic.mdf_text(m, names) + f"+outlier\t1\t\nchr2\t1000\t{1000 + long_len}\t-\t\n"
s = Sequencer(0)
for nm, c in zip(names, contigs): s.add_contig(nm, c)
md = os.path.join("tksm_amd", "models", "badread")
s.set_identity(84.0, 99.0, 5.5); s.load_error_model(os.path.join(md, "nanopore2020.error.gz")); s.load_qscore_model(os.path.join(md, "nanopore2020.qscore.gz"))
b = s.batch_from_mdf(text)
for rep in range(2):
    t = time.time(); r = s.run(b, target="badread", fastq=True, compute_qual=True, seed=9); s.synchronize(); dt = time.time() - t
    free_b, total_b = torch.cuda.mem_get_info()
    print(f"run {rep}: {n + 1} reads in {dt * 1e3:.0f} ms; HBM in use {(total_b - free_b) / 2**30:.1f} GiB", flush=True)
recs = r.records()
ref = {nm: c.decode() for nm, c in zip(names, contigs)}
em = po.ErrorModel(os.path.join(md, "nanopore2020.error.gz")); qm = po.QScoreModel(os.path.join(md, "nanopore2020.qscore.gz"))
ident = po.Identities(84.0, 5.5, 99.0)
mols = list(po.mdf_generator(text.splitlines(keepends=True)))
bad = 0
for i in list(range(0, n, max(1, n // 40))) + [n]:
    want = po.badread_record(True, 9, i, po.splice(ref, mols[i][1]), ident, em, qm, True, mols[i][0])[0]
    bad += want != recs[i]
print("records compared with the oracle:", len(range(0, n, max(1, n // 40))) + 1, "mismatching:", bad, "; outlier record bytes", len(recs[n]))
