"""diagnostic: find the reads of the scRNA-like synthetic workload (polyA mean 40) that need the unbanded alignment, and
dump their error-free sequences so that they can be replayed as literal molecules (tests/golden/unbanded_reads.json)"""
import sys, os, json, re, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from tksm_amd import synthetic
from tksm_amd.sequence import Sequencer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
rs = np.random.RandomState(2)
lens = [2_000_000] * 4
contigs = [rs.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() for n in lens]
s = Sequencer(0)
for c, seq in enumerate(contigs):
    s.add_contig(f"chr{c + 1}", seq)
m_ = os.path.join('tksm_amd', 'models', 'badread')
s.set_identity(84.0, 99.0, 5.5); s.load_error_model(os.path.join(m_, 'nanopore2020.error.gz')); s.load_qscore_model(os.path.join(m_, 'nanopore2020.qscore.gz'))
m = synthetic.make_molecules(rs, lens, B, 1000, 200, kind="scrna", polya_mean=float(os.environ.get("POLYA_MEAN", "40")))
b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
r = s.run(b, target='badread', fastq=True, compute_qual=True, seed=42, first_read_index=0, collect_stats=True)
ist, dst = r.stats()
idx = [int(i) for i in np.nonzero(ist[:, 7] & 16)[0]]
print("reads with an unbanded alignment:", idx[:40], flush=True)
per = s.run(b, target='perfect', fastq=False, seed=42).records()
out = [{"index": i, "sequence": per[i].split(b"\n")[1].decode()} for i in idx[:8]]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/unbanded_reads.json", "w"))
print("dumped", len(out))
