"""How much do co-optimal path choices of the q-score alignment matter?  ("quantify the edlib hole".)

    python tools/edlib_hole.py [n_3000=3000] [n_9000=600] [procs]        -> profiles/r04_edlib_hole.log (copy the output)

The specification (oracle + kernels) keeps edlib's TRACEBACK order (read-only 'I', fragment-only 'D', diagonal) at every size; the real
edlib -- python-edlib, not in the reference tree, not installable here -- switches to Hirschberg's divide and conquer for the q-score
alignment (py/tksm_badread.py:611-613) of every read above ~1.77 kb.  The oracle's test-only variants (oracle/tksm_oracle.c,
oracle_set_qscore_alignment_variant) run the SAME reads -- same error loop, same new sequence: only the q-score alignment's path
differs -- with (1) the opposite indel preference, (2) edlib as published incl. its Hirschberg branch, (3) the shipped order without
the band (control), and this script measures what moves: reads and positions whose quality differs, the realised identity
(KS distance, largest difference), and the q-score histograms per alignment op ('=', 'X', 'I' of a fixed ruler alignment of the read
against its molecule; total variation distance).  Gates of tests/test_oracle_golden.py: KS D <= 0.02, TV <= 0.01 (+ sampling noise).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
MODELS = os.path.join(ROOT, "tksm_amd", "models", "badread")
VARIANTS = {0: "shipped (band, I / D / diagonal)", 3: "shipped order, no band", 1: "opposite indel preference", 2: "edlib as published (Hirschberg above 1 MB)"}
_W = {}


def init(model="nanopore2020"):
    import pyoracle
    _W["po"] = pyoracle
    _W["em"] = pyoracle.ErrorModel(os.path.join(MODELS, model + ".error.gz"))
    _W["qm"] = pyoracle.QScoreModel(os.path.join(MODELS, model + ".qscore.gz"))
    _W["ident"] = pyoracle.Identities(84.0, 5.5, 99.0)


def work(job):
    """reads lo .. hi-1 of length class L through every variant"""
    from stats_common import qscore_hist
    L, lo, hi = job
    po, em, qm, ident = _W["po"], _W["em"], _W["qm"], _W["ident"]
    out = {v: {"identity": [], "qh": np.zeros((3, 94), np.int64), "reads_differ": 0, "pos_differ": 0, "splits": 0, "max_dident": 0.0} for v in VARIANTS}
    positions = 0
    for r in range(lo, hi):
        rs = np.random.RandomState((L * 1000003 + r) % (2 ** 32))
        raw = bytes(rs.choice(list(b"ACGT"), L).tolist())
        read = 77_000_000 + L * 100_000 + r
        tgt = ident.get_identity(4321, read)
        base = None
        cig = None
        for v in VARIANTS:
            po.set_qscore_alignment_variant(v)
            seq, qual, idt, st = po.sequence_fragment(raw, tgt, em, qm, True, 4321, read)
            if v == 0:
                base = (seq, qual, idt)
                assert st.band_fail == 0
                _, cig = po.nw_cigar(seq, raw) if len(seq) else (0, "")
                positions += len(seq)
            assert seq == base[0], "the variants must not touch the sequence"
            o = out[v]
            o["identity"].append(idt)
            if len(seq):
                o["qh"] += qscore_hist(cig, qual)
            nd = sum(a != b for a, b in zip(qual, base[1]))
            o["pos_differ"] += nd
            o["reads_differ"] += 1 if (nd or idt != base[2]) else 0
            o["max_dident"] = max(o["max_dident"], abs(idt - base[2]))
            o["splits"] += st.pad0
    po.set_qscore_alignment_variant(0)
    return out, positions, hi - lo


def measure(L, n, procs):
    """-> {variant: distances vs the shipped variant} for n reads of L bases"""
    from multiprocessing import Pool
    from scipy.stats import ks_2samp
    step = max(1, n // (procs * 6))
    jobs = [(L, lo, min(n, lo + step)) for lo in range(0, n, step)]
    with Pool(procs, initializer=init) as p:
        res = p.map(work, jobs, chunksize=1)
    positions = sum(r[1] for r in res)
    tv = lambda a, b: 0.5 * np.abs(a / max(1.0, a.sum()) - b / max(1.0, b.sum())).sum()
    idb = np.concatenate([np.asarray(r[0][0]["identity"]) for r in res])
    qhb = np.sum([r[0][0]["qh"] for r in res], axis=0).astype(float)
    rep = {}
    for v in VARIANTS:
        if v == 0:
            continue
        idv = np.concatenate([np.asarray(r[0][v]["identity"]) for r in res])
        qhv = np.sum([r[0][v]["qh"] for r in res], axis=0).astype(float)
        rep[v] = {"reads": n, "positions": positions, "reads_differ": sum(r[0][v]["reads_differ"] for r in res),
                  "positions_differ": sum(r[0][v]["pos_differ"] for r in res), "ks_identity": float(ks_2samp(idv, idb).statistic),
                  "max_abs_identity_difference": max(r[0][v]["max_dident"] for r in res),
                  "mean_abs_identity_difference": float(np.abs(idv - idb).mean()),
                  "tv_qhist": {op: float(tv(qhv[i], qhb[i])) for i, op in enumerate("=XI")},
                  "hirschberg_splits_per_read": sum(r[0][v]["splits"] for r in res) / n}
    return rep


def main():
    a = sys.argv[1:]
    n3 = int(a[0]) if len(a) > 0 else 3000
    n9 = int(a[1]) if len(a) > 1 else 600
    procs = int(a[2]) if len(a) > 2 else max(1, len(os.sched_getaffinity(0)) - 1)
    for L, n in ((1000, min(n3, 2000)), (3000, n3), (9000, n9)):
        if n <= 0:
            continue
        t0 = time.time()
        rep = measure(L, n, procs)
        print(f"--- {n} reads of {L} bases, nanopore2020, identity 84,99,5.5 ({time.time() - t0:.0f} s on {procs} processes); distances from the shipped variant", flush=True)
        for v, d in rep.items():
            print(f"  {VARIANTS[v]:46s} reads that differ {d['reads_differ']:5d} / {d['reads']}, positions {d['positions_differ']:7d} / {d['positions']} "
                  f"({d['positions_differ'] / max(1, d['positions']):.2e}); identity: KS D {d['ks_identity']:.4f}, largest |difference| {d['max_abs_identity_difference']:.2e}, "
                  f"mean {d['mean_abs_identity_difference']:.2e}; q-score histograms TV '=' {d['tv_qhist']['=']:.5f} 'X' {d['tv_qhist']['X']:.5f} 'I' {d['tv_qhist']['I']:.5f}; "
                  f"Hirschberg splits per read {d['hirschberg_splits_per_read']:.2f}", flush=True)


if __name__ == "__main__":
    main()
