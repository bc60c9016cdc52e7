"""diagnostic: throughput of the device-side PCR amplification and truncation (BASELINE config 5: 20 cycles), and of the Seq
path on their output"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from tksm_amd import synthetic
from tksm_amd.sequence import Sequencer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
target = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
dev = torch.device('cuda', 0)
s = Sequencer(0)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
for c in range(4):
    s.add_contig(f"chr{c+1}", lut[torch.randint(0, 4, (16_000_000,), dtype=torch.uint8, device=dev).long()])
md = os.path.join('tksm_amd', 'models', 'badread')
s.set_identity(84.0, 99.0, 5.5); s.load_error_model(os.path.join(md, 'nanopore2020.error.gz')); s.load_qscore_model(os.path.join(md, 'nanopore2020.qscore.gz'))
rs = np.random.RandomState(4)
m = synthetic.make_molecules(rs, [16_000_000] * 4, n, 1000, 200)
b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
for rep in range(2):
    t = time.time(); pb = s.pcr(b, 20, target, preset="Taq-setting1", seed=7 + rep); s.synchronize(); dt = time.time() - t
    print(f"pcr: {n} templates, 20 cycles -> {pb.n_reads} molecules in {dt * 1e3:.1f} ms = {pb.n_reads / dt / 1e6:.1f} M molecules/s", flush=True)
for rep in range(2):
    t = time.time(); tb = s.truncate(pb, lognormal=(6.9, 0.5), seed=9 + rep); s.synchronize(); dt = time.time() - t
    print(f"truncate (lognormal): {pb.n_reads} -> {tb.n_reads} molecules in {dt * 1e3:.1f} ms = {pb.n_reads / dt / 1e6:.1f} M molecules/s", flush=True)
for rep in range(2):
    t = time.time(); r = s.run(tb, target='badread', fastq=True, compute_qual=True, seed=42); s.synchronize(); dt = time.time() - t
    print(f"sequence: {tb.n_reads} molecules in {dt * 1e3:.1f} ms = {tb.n_reads / dt / 1e6:.2f} M reads/s", flush=True)
