#!/bin/bash
# AddressSanitizer + UBSan over the host-only code of the library (MDF parser incl. malformed inputs and the multi-threaded split,
# model loaders, identity table, read ordering).  CPU only; the MDF comes from tools/parse_bench.sh (/tmp/parse_bench.mdf).
set -e
cd "$(dirname "$0")/.."
[ -f /tmp/parse_bench.mdf ] || python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from tksm_amd import synthetic
rs = np.random.RandomState(1)
m = synthetic.make_molecules(rs, [8_000_000] * 4, 120000, 1000, 200)
open('/tmp/parse_bench.mdf', 'w').write(synthetic.mdf_text(m, [f"chr{c+1}" for c in range(4)]))
PY
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I tksm_amd/csrc -o /tmp/sanitize_host tools/sanitize_host.cpp \
    tksm_amd/csrc/hostio.cpp tksm_amd/csrc/models.cpp -lz -lpthread
/tmp/sanitize_host /tmp/parse_bench.mdf tksm_amd/models/badread/nanopore2020.error.gz tksm_amd/models/badread/nanopore2020.qscore.gz
