#!/usr/bin/env python3
"""Every-record golden digests of bench-size batches (run in the build container; ~10 min per workload on 8 cores).

    python tests/golden/make_bench_digests.py [bulk] [scrna] [pcr] [lognormal]

For each workload of tests/test_gpu_parity.py::test_bench_size_batch_every_record_by_digest -- one batch of bench.py's default size
(1 703 936 molecules; bulk = BASELINE config 2, scrna = config 3's barcode / UMI / polyA literals, pcr = config 5's substitution-heavy
molecules) or 1 048 576 molecules with lognormal lengths to 16 kb, the same seeded genome and molecule generator as that test -- the CPU oracle (oracle/tksm_oracle.c through oracle/pyoracle.py) computes every Badread FASTQ
record and this script stores one SHA-256 per block of 4 096 consecutive records (416 digests per workload) in
tests/golden/bench_digests_<kind>.json.  The GPU test hashes the device output the same way: all 1 703 936 records are compared
without a second of oracle time on the GPU box.

The oracle is test infrastructure; nothing here touches the product library.
"""
import hashlib
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
BLOCK = 4096
SEED = 9                      # the run's seed in the test
GEN_SEED = 23                 # genome + molecule generator seed in the test
N = 1_703_936
LENS = [8_000_000] * 4
NAMES = [f"chr{c + 1}" for c in range(4)]
_S = {}


# name -> (generator kind, molecules, lognormal sigma): the parameter sets of the GPU test
WORKLOADS = {"bulk": ("bulk", N, None), "scrna": ("scrna", N, None), "pcr": ("pcr", N, None), "lognormal": ("bulk", 1_048_576, 0.6)}


def workload(name):
    """the genome and molecules of the test, from the same generator calls in the same order"""
    from tksm_amd import synthetic
    kind, n, sigma = WORKLOADS[name]
    rs = np.random.RandomState(GEN_SEED)
    ref = {nm: rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode() for nm, L in zip(NAMES, LENS)}
    m = synthetic.make_molecules(rs, LENS, n, 1000, 200, kind=kind, lognormal_sigma=sigma)
    return ref, m


def block_records(po, ref, m, lo, hi, ident, em, qm):
    """the oracle's records of reads lo .. hi-1 (global read index = position in the batch)"""
    from tksm_amd import synthetic
    text = synthetic.mdf_text(synthetic.take(m, np.arange(lo, hi)), NAMES)
    out = []
    for g, (mid, ivs) in zip(range(lo, hi), po.mdf_generator(text.splitlines(keepends=True))):
        out.append(po.badread_record(True, SEED, g, po.splice(ref, ivs), ident, em, qm, True, mid)[0])
    return out


def _worker(blk):
    lo, hi = blk * BLOCK, min(_S["n"], (blk + 1) * BLOCK)
    recs = block_records(_S["po"], _S["ref"], _S["m"], lo, hi, _S["ident"], _S["em"], _S["qm"])
    h = hashlib.sha256()
    nbytes = 0
    for r in recs:
        h.update(r)
        nbytes += len(r)
    return blk, h.hexdigest(), nbytes


def main():
    kinds = [a for a in sys.argv[1:] if a in WORKLOADS] or list(WORKLOADS)
    procs = int(os.environ.get("DIGEST_PROCS", str(max(1, len(os.sched_getaffinity(0)) - 1))))
    import pyoracle as po
    models = os.path.join(ROOT, "tksm_amd", "models", "badread")
    em = po.ErrorModel(os.path.join(models, "nanopore2020.error.gz"))
    qm = po.QScoreModel(os.path.join(models, "nanopore2020.qscore.gz"))
    ident = po.Identities(84.0, 5.5, 99.0)
    for kind in kinds:
        ref, m = workload(kind)
        n = WORKLOADS[kind][1]
        _S.update(po=po, ref=ref, m=m, n=n, ident=ident, em=em, qm=qm)
        nblk = (n + BLOCK - 1) // BLOCK
        digests, total = [None] * nblk, 0
        t0 = time.time()
        with Pool(procs) as p:
            for k, (blk, hx, nb) in enumerate(p.imap_unordered(_worker, range(nblk))):
                digests[blk] = hx
                total += nb
                if k % 32 == 0:
                    print(f"{kind}: {k + 1}/{nblk} blocks, {time.time() - t0:.0f} s", flush=True)
        out = {"kind": kind, "generator": {"kind": WORKLOADS[kind][0], "lognormal_sigma": WORKLOADS[kind][2]}, "n": n, "block_records": BLOCK, "run_seed": SEED, "generator_seed": GEN_SEED,
               "genome": "4 x 8 Mb uniform ACGT (np.random.RandomState(23))", "models": "nanopore2020 error + qscore, identity 84,99,5.5",
               "records": "Badread FASTQ with computed qualities, first_read_index 0, stride 1", "total_bytes": total,
               "made_by": "tests/golden/make_bench_digests.py (oracle/tksm_oracle.c via pyoracle)", "sha256": digests}
        with open(os.path.join(ROOT, "tests", "golden", f"bench_digests_{kind}.json"), "w") as f:
            json.dump(out, f, indent=0)
        print(f"{kind}: {n} records, {total} bytes, {time.time() - t0:.0f} s on {procs} processes", flush=True)


if __name__ == "__main__":
    main()
