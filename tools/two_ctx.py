"""experiment: aggregate throughput of two contexts driven from two threads (kernels of both streams overlap)"""
import sys, os, time, threading, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from tksm_amd import synthetic
from tksm_amd.sequence import Sequencer
dev = torch.device('cuda', 0)
B = int(sys.argv[1]); NCTX = int(sys.argv[2]); IT = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
gen = [lut[torch.randint(0, 4, (16_000_000,), dtype=torch.uint8, device=dev).long()] for _ in range(4)]
m_ = os.path.join('tksm_amd', 'models', 'badread')
ctxs = []
streams = []
for k in range(NCTX):
    st = torch.cuda.Stream(device=dev); streams.append(st)
    s = Sequencer(0, stream=st.cuda_stream)
    for c in range(4): s.add_contig(f"chr{c+1}", gen[c])
    s.set_identity(84.0, 99.0, 5.5); s.load_error_model(os.path.join(m_, 'nanopore2020.error.gz')); s.load_qscore_model(os.path.join(m_, 'nanopore2020.qscore.gz'))
    rs = np.random.RandomState(2 + k)
    m = synthetic.make_molecules(rs, [16_000_000] * 4, B, 1000, 200)
    b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
    ctxs.append((s, b))
def work(s, b, it):
    for i in range(it): s.run(b, target='badread', fastq=True, compute_qual=True, seed=42, first_read_index=i * B)
for s, b in ctxs: work(s, b, 1)
torch.cuda.synchronize()
t = time.time()
th = [threading.Thread(target=work, args=(s, b, IT)) for s, b in ctxs]
[x.start() for x in th]; [x.join() for x in th]
torch.cuda.synchronize()
dt = time.time() - t
print(f"B={B} ctx={NCTX}: {NCTX * IT * B / dt:.0f} reads/s aggregate ({dt*1e3/IT:.1f} ms per step-set)")
