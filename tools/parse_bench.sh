#!/bin/bash
# usage: bash tools/parse_bench.sh   (host only; generates 0.88 M molecules of MDF text under /tmp)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from tksm_amd import synthetic
rs = np.random.RandomState(1)
m = synthetic.make_molecules(rs, [8_000_000] * 4, 880000, 1000, 200)
open('/tmp/parse_bench.mdf', 'w').write(synthetic.mdf_text(m, [f"chr{c+1}" for c in range(4)]))
PY
g++ -O2 -std=c++17 -I tksm_amd/csrc -o /tmp/parse_bench tools/parse_bench.cpp tksm_amd/csrc/hostio.cpp tksm_amd/csrc/models.cpp -lz -lpthread
/tmp/parse_bench /tmp/parse_bench.mdf
