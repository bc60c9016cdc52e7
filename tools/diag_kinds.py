import sys, os, numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from tksm_amd import synthetic
from tksm_amd.sequence import Sequencer
dev=torch.device('cuda',0)
s=Sequencer(0)
lut=torch.tensor(list(b"ACGT"),dtype=torch.uint8,device=dev)
for c in range(4):
    codes=torch.randint(0,4,(16_000_000,),dtype=torch.uint8,device=dev)
    s.add_contig(f"chr{c+1}", lut[codes.long()])
m_=os.path.join('tksm_amd','models','badread')
s.set_identity(84.0,99.0,5.5); s.load_error_model(os.path.join(m_,'nanopore2020.error.gz')); s.load_qscore_model(os.path.join(m_,'nanopore2020.qscore.gz'))
for kind in ('bulk','scrna'):
    rs=np.random.RandomState(2)
    m=synthetic.make_molecules(rs,[16_000_000]*4,1703936,1000,200,kind=kind)
    b=s.batch_from_arrays(m["reads"],m["intervals"],m["mods"],m["literals"],m["literal_pool"],m["ids"],m["id_pool"])
    r=s.run(b,target='badread',fastq=True,compute_qual=True,seed=42)
    print(kind, s.run_diagnostics(), flush=True)
    b.free()
