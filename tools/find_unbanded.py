"""diagnostic: which reads of a reproducible polyA-tailed corpus take the unbanded alignment fallback (TKSMSEQ_VERBOSE=1)"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from tksm_amd.sequence import Sequencer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rs = np.random.RandomState(77)
ref = rs.choice(np.frombuffer(b"ACGT", np.uint8), 200_000).tobytes()
s = Sequencer(0)
s.add_contig("chrT", ref)
m_ = os.path.join('tksm_amd', 'models', 'badread')
s.set_identity(84.0, 99.0, 5.5); s.load_error_model(os.path.join(m_, 'nanopore2020.error.gz')); s.load_qscore_model(os.path.join(m_, 'nanopore2020.qscore.gz'))
lines = []
for i in range(n):
    st = int(rs.randint(1000, 150_000)); ln = int(rs.randint(150, 400)); pa = int(rs.randint(30, 120))
    lines.append(f"+pa{i}\t1\t\nchrT\t{st}\t{st + ln}\t{'+-'[i & 1]}\t\n{'A' * pa}\t0\t{pa}\t+\t\n")
r = s.run(s.batch_from_mdf("".join(lines)), target='badread', fastq=True, compute_qual=True, seed=20240517)
print("done", r.n_reads)
