"""Soak parity: many synthetic reads through the HIP path and through the CPU oracle, records compared by digest.

    python tools/soak_parity.py [n=200000] [kind=bulk|scrna|pcr] [mean=1000] [sigma=0 (lognormal)] [tail=0|1] [seed=42]
    (environment: SOAK_MODEL = nanopore2020 | nanopore2018 | pacbio2016, SOAK_IDENT = mean,max,stdev)

The oracle runs first in forked workers (before this process touches the GPU), ~7 k reads/s on 16 cores.
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
_S = {}


def worker(args):
    lo, hi = args
    po = _S["po"]
    out = []
    fails = 0
    for i in range(lo, hi):
        raw = po.splice(_S["ref"], _S["mols"][i][1])
        rec, st = po.badread_record(True, _S["seed"], i, raw, _S["ident"], _S["em"], _S["qm"], True, _S["mols"][i][0],
                                    tail_model=_S["tail"])
        fails += st.band_fail
        out.append(hashlib.md5(rec).digest())
    return out, fails


def main():
    a = sys.argv[1:]
    n = int(a[0]) if len(a) > 0 else 200000
    kind = a[1] if len(a) > 1 else "bulk"
    mean = int(a[2]) if len(a) > 2 else 1000
    sigma = float(a[3]) if len(a) > 3 else 0.0
    tail = int(a[4]) if len(a) > 4 else 0
    seed = int(a[5]) if len(a) > 5 else 42
    import pyoracle as po
    from multiprocessing import Pool
    from tksm_amd import synthetic
    rs = np.random.RandomState(seed)
    lens = [1_000_000] * 24
    ref = {f"chr{i + 1}": rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode() for i, L in enumerate(lens)}
    m = synthetic.make_molecules(rs, lens, n, mean, mean * 0.2, kind=kind, lognormal_sigma=sigma or None)
    text = synthetic.mdf_text(m, list(ref))
    mols = list(po.mdf_generator(text.splitlines(keepends=True)))
    models = os.path.join(ROOT, "tksm_amd", "models", "badread")
    model = os.environ.get("SOAK_MODEL", "nanopore2020")                       # a shipped model pair by name
    imean, imax, isd = (float(x) for x in os.environ.get("SOAK_IDENT", "84,99,5.5").split(","))   # identity mean,max,stdev
    tail_path = os.path.join(ROOT, "tests", "golden", "tail_model_synth.json")
    _S.update(po=po, ref=ref, mols=mols, seed=seed, em=po.ErrorModel(os.path.join(models, f"{model}.error.gz")),
              qm=po.QScoreModel(os.path.join(models, f"{model}.qscore.gz")), ident=po.Identities(imean, isd, imax),
              tail=po.TailModel(tail_path) if tail else None)
    cores = min(16, len(os.sched_getaffinity(0)))
    step = 2000
    chunks = [(i, min(n, i + step)) for i in range(0, n, step)]
    t0 = time.time()
    want, fails = [], 0
    with Pool(cores) as p:
        for k, (d, f) in enumerate(p.imap(worker, chunks)):
            want.extend(d); fails += f
            if k % 10 == 0:
                print(f"oracle {len(want)}/{n} reads, {time.time() - t0:.0f} s", flush=True)
    print(f"oracle: {n} reads in {time.time() - t0:.1f} s on {cores} processes; unbanded fallbacks {fails}", flush=True)

    import torch  # noqa: F401  (its ROCm runtime has to be loaded before libtksmseq.so)
    torch.cuda.is_available()
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    for name, seq in ref.items():
        s.add_contig(name, seq.encode())
    s.set_identity(imean, imax, isd)
    s.load_error_model(os.path.join(models, f"{model}.error.gz"))
    s.load_qscore_model(os.path.join(models, f"{model}.qscore.gz"))
    if tail:
        s.load_tail_model(tail_path)
    b = s.batch_from_mdf(text)
    t0 = time.time()
    recs = s.run(b, target="badread", fastq=True, compute_qual=True, seed=seed).records()
    print(f"gpu: {len(recs)} records in {time.time() - t0:.2f} s (incl. download)", flush=True)
    bad = [i for i in range(n) if hashlib.md5(recs[i]).digest() != want[i]]
    print(f"RESULT n={n} kind={kind} mean={mean} sigma={sigma} tail={tail} seed={seed} model={model} identity={imean},{imax},{isd}: mismatching records {len(bad)}"
          + (f", first {bad[:5]}" if bad else ""), flush=True)
    s.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
