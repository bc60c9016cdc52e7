#!/usr/bin/env python3
"""Which band rows does the preferred path of an identity re-estimation alignment visit?  (diagnostic, CPU oracle)

Builds the oracle with -DBAND_STATS (every DP cell carries the range of row offsets from the generative row along its preferred
path), simulates reads of the bench workloads and prints, for candidate windows of stored rows, the share of alignments whose
path leaves the window (= what k_aln's first pass would hand to the full-width pass).
usage: python tools/band_rows.py [reads per workload]"""
import ctypes as C
import os
import subprocess
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = "/tmp/tksm_band_stats/libtksm_oracle.so"
MODELS = os.path.join(ROOT, "tksm_amd", "models", "badread")
_W = {}


def init():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    lib = C.CDLL(SO)
    for name in ("oracle_sequence_fragment", "oracle_target_identity"):      # same prototypes, the instrumented build
        getattr(lib, name).argtypes = getattr(po._lib, name).argtypes
        getattr(lib, name).restype = getattr(po._lib, name).restype
    po._lib = lib
    _W.update(po=po, lib=lib, em=po.ErrorModel(os.path.join(MODELS, "nanopore2020.error.gz")), qm=po.QScoreModel(os.path.join(MODELS, "nanopore2020.qscore.gz")),
              ident=po.Identities(84.0, 5.5, 99.0))


def work(job):
    kind, lo, hi = job
    po = _W["po"]
    rs = np.random.RandomState(7 * {"bulk": 1, "scrna": 2}[kind] + lo)
    for r in range(lo, hi):
        L = max(200, int(round(rs.normal(1000, 200))))
        raw = bytes(rs.choice(list(b"ACGT"), L + (26 if kind == "scrna" else 0)).tolist())
        if kind == "scrna":
            raw += b"A" * int(np.clip(round(rs.normal(15, 7.5)), 0, 5000))
        read = 5_000_000 + r
        po.sequence_fragment(raw, _W["ident"].get_identity(5, read), _W["em"], _W["qm"], False, 5, read)
    h = np.zeros((128, 128), np.int64)
    _W["lib"].oracle_band_row_hist(h.ctypes.data_as(C.c_void_p), 1)
    return h


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-DBAND_STATS", "-shared", "-w", "-o", SO,
                           os.path.join(ROOT, "oracle", "tksm_oracle.c"), "-lm"])
    procs = max(1, min(8, len(os.sched_getaffinity(0))))
    for kind in ("bulk", "scrna"):
        step = max(10, n // (procs * 4))
        with Pool(procs, initializer=init) as p:
            h = sum(p.map(work, [(kind, lo, min(n, lo + step)) for lo in range(0, n, step)], chunksize=1))
        tot = h.sum()
        print(f"{kind}: {n} reads, {tot} alignments")
        for rows in (6, 8, 10, 12, 13, 14, 16, 20, 24, 32):
            best = None
            for first in range(-rows + 1, 1):                   # window = offsets first .. first + rows - 1 (must hold offset 0)
                inside = h[64 + first:, :64 + first + rows].sum()    # lo >= first and hi <= first + rows - 1
                miss = 1.0 - inside / tot
                if best is None or miss < best[0]:
                    best = (miss, first)
            print(f"  {rows:2d} stored rows: best window = offsets {best[1]:+d} .. {best[1] + rows - 1:+d} (band rows {31 + best[1]} .. {31 + best[1] + rows - 1}), "
                  f"path leaves it in {100 * best[0]:.2f} % of the alignments")


if __name__ == "__main__":
    main()
