"""diagnostic: what the first tksmseq_run of a context costs (a tiny batch first, then a large one; then a clone)"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from tksm_amd import synthetic
from tksm_amd.sequence import Sequencer
dev = torch.device('cuda', 0)
s = Sequencer(0)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
for c in range(4):
    s.add_contig(f"chr{c+1}", lut[torch.randint(0, 4, (16_000_000,), dtype=torch.uint8, device=dev).long()])
m_ = os.path.join('tksm_amd', 'models', 'badread')
s.set_identity(84.0, 99.0, 5.5); s.load_error_model(os.path.join(m_, 'nanopore2020.error.gz')); s.load_qscore_model(os.path.join(m_, 'nanopore2020.qscore.gz'))
rs = np.random.RandomState(2)
def batch(seq, n):
    m = synthetic.make_molecules(rs, [16_000_000] * 4, n, 1000, 200)
    return seq.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
def timed(seq, b, tag):
    t = time.time(); r = seq.run(b, target='badread', fastq=True, compute_qual=True, seed=42, first_read_index=0); seq.synchronize(); print(f"{tag}: {(time.time()-t)*1e3:.0f} ms", flush=True)
small, big = batch(s, 1000), batch(s, 880000)
timed(s, small, "ctx0 first run, 1 k reads"); timed(s, small, "ctx0 second run, 1 k reads")
timed(s, big, "ctx0 first big run, 880 k reads"); timed(s, big, "ctx0 second big run")
c1 = s.clone()
big1 = batch(c1, 880000)
timed(c1, big1, "clone first run, 880 k reads"); timed(c1, big1, "clone second run")
# three fresh contexts, first large runs at the same time (as the CLI's workers do)
import threading
ctxs = [s.clone() for _ in range(3)]
bs = [batch(c, 880000) for c in ctxs]
def w(i):
    for rep in range(2):
        t = time.time(); ctxs[i].run(bs[i], target='badread', fastq=True, compute_qual=True, seed=42, first_read_index=0); ctxs[i].synchronize()
        print(f"concurrent ctx {i} run {rep}: {(time.time()-t)*1e3:.0f} ms", flush=True)
ts = [threading.Thread(target=w, args=(i,)) for i in range(3)]
t0 = time.time()
for t in ts: t.start()
for t in ts: t.join()
print(f"all: {(time.time()-t0)*1e3:.0f} ms")
