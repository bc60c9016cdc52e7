#!/bin/bash
# `--devices 0,0` (two context groups, six batches in flight) against `--devices 0` on a 2 M-molecule file: same bytes (md5)
set -e
cd "$(dirname "$0")/.."
d=/tmp/devchk; mkdir -p $d
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from tksm_amd import synthetic
rs = np.random.RandomState(3)
lens = [8_000_000] * 4
with open('/tmp/devchk/ref.fa', 'w') as f:
    for c, L in enumerate(lens):
        s = rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode()
        f.write(f">chr{c+1}\n"); f.write("\n".join(s[i:i+80] for i in range(0, L, 80))); f.write("\n")
m = synthetic.make_molecules(rs, lens, 2_000_000, 1000, 200)
open('/tmp/devchk/mols.mdf', 'w').write(synthetic.mdf_text(m, [f"chr{c+1}" for c in range(4)]))
PY
export TKSM_MODELS=$PWD/tksm_amd/models
for dev in 0 0,0; do
  t0=$(date +%s%N)
  tksm_amd/tksm sequence -i $d/mols.mdf -r $d/ref.fa -o $d/out_$dev.fastq --devices $dev -t 8 --batch-bytes 16777216 --verbosity ERROR
  echo "devices $dev: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall"
  md5sum $d/out_$dev.fastq
done
rm -rf $d
