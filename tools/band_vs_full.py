#!/usr/bin/env python3
"""Guided 64-row band vs exact (unbanded) alignment, at scale, on the CPU oracle.

The band is part of the specification: this tool measures how often it changes anything relative to
the exact global alignment the reference's edlib call computes.  Every read is simulated twice with the same seed -- banded
(the specification) and with `use_full=1` (every alignment unbanded) -- and sequence, realised identity, per-read error
estimate and qualities are compared.  Output: one summary line per workload (profiles/r02_band_vs_full.log).
usage: python tools/band_vs_full.py [reads per workload]"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
MODELS = os.path.join(ROOT, "tksm_amd", "models", "badread")
_W = {}


def init(model):
    import pyoracle as po
    _W.update(po=po, em=po.ErrorModel(os.path.join(MODELS, model + ".error.gz")), qm=po.QScoreModel(os.path.join(MODELS, model + ".qscore.gz")),
              ident=po.Identities(84.0, 5.5, 99.0))


def work(job):
    kind, lo, hi = job
    po = _W["po"]
    rs = np.random.RandomState(1000003 * {"bulk": 1, "scrna": 2, "lognormal": 3}[kind] + lo)      # (a fixed number per workload: str hashes differ from process to process)
    out = dict(n=0, seq=0, ident=0, errors=0, qual=0, qual_pos=0, bases=0, band_fail=0, maxd=0.0)
    for r in range(lo, hi):
        if kind == "bulk":
            L = max(200, int(round(rs.normal(1000, 200))))
            raw = bytes(rs.choice(list(b"ACGT"), L).tolist())
        elif kind == "scrna":
            L = max(200, int(round(rs.normal(1000, 200))))
            pa = int(np.clip(round(rs.normal(15, 7.5)), 0, 5000))
            raw = bytes(rs.choice(list(b"ACGT"), L + 26).tolist()) + b"A" * pa
        else:                                               # transcript-like lengths, lognormal with a tail to 16 kb
            L = int(np.clip(round(1000 * np.exp(rs.normal(0.0, 0.6))), 200, 16000))
            raw = bytes(rs.choice(list(b"ACGT"), L).tolist())
        read = 77_000_000 + r
        tgt = _W["ident"].get_identity(5, read)
        a = po.sequence_fragment(raw, tgt, _W["em"], _W["qm"], True, 5, read)
        b = po.sequence_fragment(raw, tgt, _W["em"], _W["qm"], True, 5, read, use_full=True)
        out["n"] += 1
        out["seq"] += a[0] != b[0]
        out["ident"] += a[2] != b[2]
        out["errors"] += a[3].errors != b[3].errors
        out["band_fail"] += a[3].band_fail
        out["maxd"] = max(out["maxd"], abs(a[2] - b[2]))
        if a[0] == b[0]:
            d = sum(x != y for x, y in zip(a[1], b[1]))
            out["qual"] += d > 0
            out["qual_pos"] += d
            out["bases"] += len(a[1])
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    procs = max(1, min(8, len(os.sched_getaffinity(0))))
    for kind, cnt in (("bulk", n), ("scrna", n), ("lognormal", max(1000, n // 4))):
        t0 = time.time()
        step = max(10, cnt // (procs * 8))
        with Pool(procs, initializer=init, initargs=("nanopore2020",)) as p:
            res = p.map(work, [(kind, lo, min(cnt, lo + step)) for lo in range(0, cnt, step)], chunksize=1)
        tot = {k: (max(r[k] for r in res) if k == "maxd" else sum(r[k] for r in res)) for k in res[0]}
        print(f"{kind}: {tot['n']} reads, banded vs unbanded -- sequence differs {tot['seq']}, realised identity differs {tot['ident']} "
              f"(largest |delta| {tot['maxd']:.2e}), error estimate differs {tot['errors']}, reads with a differing quality {tot['qual']} "
              f"({tot['qual_pos']} of {tot['bases']} positions), band failures (unbanded fallback taken by the specification) {tot['band_fail']}; "
              f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
