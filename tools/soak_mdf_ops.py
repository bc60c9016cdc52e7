"""Soak parity of the MDF modules on the device: PCR and truncation of many molecules, MDF text compared with the Python oracle
(oracle/mdf_ops_oracle.py), molecule by molecule.

    python tools/soak_mdf_ops.py [templates=20000] [target=200000] [seed=5]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
from multiprocessing import Pool
import mdf_ops_oracle as mo
from test_mdf_ops import _mdf

_S = {}
def trc_chunk(a):
    lo, hi, seed, kw = a
    return mo.write_mdf([mo.trc_spec(md, g, seed, **kw) for g, md in zip(range(lo, hi), _S["mols"][lo:hi])])

def main():
    a = sys.argv[1:]
    nt = int(a[0]) if len(a) > 0 else 20000
    target = int(a[1]) if len(a) > 1 else 200000
    seed = int(a[2]) if len(a) > 2 else 5
    text = _mdf(np.random.RandomState(seed), nt)
    t0 = time.time()
    tmpl = mo.stream_mdf(text, unroll=True)
    want_pcr = mo.write_mdf(mo.pcr_spec(tmpl, 8, 0.85, 5e-4, target, seed))
    print(f"oracle pcr: {want_pcr.count(chr(10) + '+') + 1} molecules in {time.time() - t0:.0f} s", flush=True)
    _S["mols"] = mo.stream_mdf(want_pcr, unroll=True)
    n = len(_S["mols"])
    kws = [dict(normal=(400.0, 150.0)), dict(lognormal=(6.2, 0.5))]
    want_trc = []
    with Pool(min(16, len(os.sched_getaffinity(0)))) as p:
        for kw in kws:
            t0 = time.time()
            want_trc.append("".join(p.map(trc_chunk, [(lo, min(n, lo + 4000), seed + 1, kw) for lo in range(0, n, 4000)])))
            print(f"oracle truncate {kw}: {n} molecules in {time.time() - t0:.0f} s", flush=True)
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    rs = np.random.RandomState(21)
    for c in (1, 2):
        s.add_contig(f"chr{c}", rs.choice(np.frombuffer(b"ACGT", np.uint8), 60_000).tobytes())
    s.set_host_threads(8)
    b = s.batch_from_mdf(text)
    out = s.pcr(b, 8, target, error_rate=5e-4, efficiency=0.85, seed=seed)
    got = s.to_mdf_text(out)
    bad = 0 if got == want_pcr else sum(1 for x, y in zip(got.split("\n+"), want_pcr.split("\n+")) if x != y) + abs(got.count("\n+") - want_pcr.count("\n+"))
    print(f"RESULT pcr templates={nt} molecules={out.n_reads}: mismatching molecules {bad}", flush=True)
    total_bad = bad
    for kw, want in zip(kws, want_trc):
        o2 = s.truncate(out, seed=seed + 1, **kw)
        g2 = s.to_mdf_text(o2)
        bad = 0 if g2 == want else sum(1 for x, y in zip(g2.split("\n+"), want.split("\n+")) if x != y) + abs(g2.count("\n+") - want.count("\n+"))
        print(f"RESULT truncate {kw} molecules={o2.n_reads}: mismatching molecules {bad}", flush=True)
        total_bad += bad
        o2.free()
    out.free(); b.free(); s.close()
    sys.exit(1 if total_bad else 0)

if __name__ == "__main__":
    main()
