import sys, os, time, faulthandler, numpy as np
faulthandler.dump_traceback_later(90, exit=True)
_t0 = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
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
B=int(sys.argv[1]) if len(sys.argv)>1 else 65536
rs=np.random.RandomState(2)
KIND=sys.argv[2] if len(sys.argv)>2 else 'bulk'
MEAN=int(sys.argv[3]) if len(sys.argv)>3 else 1000
SIG=float(sys.argv[4]) if len(sys.argv)>4 else None
PA=float(os.environ.get('POLYA_MEAN','15'))
m=synthetic.make_molecules(rs,[16_000_000]*4,B,MEAN,MEAN//5,kind=KIND,lognormal_sigma=SIG,polya_mean=PA)
print('lengths: mean %.0f max %d' % (m['raw_len'].mean(), m['raw_len'].max()), flush=True)
b=s.batch_from_arrays(m["reads"],m["intervals"],m["mods"],m["literals"],m["literal_pool"],m["ids"],m["id_pool"])
s.set_timing(True)
print('setup %.1f s' % (time.time() - _t0), flush=True)
for it in range(3):
    t=time.time(); r=s.run(b,target='badread',fastq=True,compute_qual=True,seed=42,first_read_index=it*B); dt=time.time()-t
    faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(90, exit=True)
    print('wall %.1f ms'%(dt*1e3), 'reads/s %.0f'%(B/dt), 'ms: lens %.2f sim %.2f scan %.2f emit %.2f total %.2f | loop %.2f aln %.2f job %.2f'%tuple(r.kernel_ms))
