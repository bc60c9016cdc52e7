"""diagnostic: hipMalloc latency while another context keeps the GPU busy"""
import sys, os, time, threading, ctypes, numpy as np
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
m = synthetic.make_molecules(rs, [16_000_000] * 4, 880000, 1000, 200)
b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
s.run(b, target='badread', fastq=True, compute_qual=True, seed=42, first_read_index=0); s.synchronize()
stop = False
def load():
    while not stop:
        s.run(b, target='badread', fastq=True, compute_qual=True, seed=42, first_read_index=0); s.synchronize()
t = threading.Thread(target=load); t.start()
time.sleep(0.3)
hip = ctypes.CDLL("libamdhip64.so")
for size_mb in (1, 64, 1024, 8192, 1024, 64):
    ts = []
    ps = []
    for rep in range(4):
        p = ctypes.c_void_p()
        t0 = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size_mb << 20)); ts.append((time.time() - t0) * 1e3); ps.append(p)
    print(f"hipMalloc {size_mb} MiB under load: " + " ".join(f"{x:.1f}" for x in ts) + " ms", flush=True)
    for p in ps:
        t0 = time.time(); hip.hipFree(p); ts.append((time.time() - t0) * 1e3)
    print("   hipFree: " + " ".join(f"{x:.1f}" for x in ts[4:]) + " ms", flush=True)
stop = True; t.join()
