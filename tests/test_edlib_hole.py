"""The one published branch of edlib that the specification does not take: Hirschberg's divide and conquer for alignments whose data
pass 1 MB -- the q-score alignment (py/tksm_badread.py:611-613) of every read above ~1.77 kb.  python-edlib is not in the reference tree
and cannot be installed here, so the branch is restated from edlib's published source as a TEST-ONLY variant of the oracle
(oracle/tksm_oracle.c, oracle_set_qscore_alignment_variant) and MEASURED against the shipped traceback order on the same reads: the error
loop and the new sequence are untouched by construction (asserted), what can move is the path of the q-score alignment -- hence the
q-scores and the printed identity.  tools/edlib_hole.py is the full-size run (profiles/r04_edlib_hole.log); this is a slice of it with
the gates of the distribution tests: KS D <= 0.02 on the realised identity, total variation <= 0.01 (+ sampling noise) on the q-score
histograms per alignment op."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hirschberg_split_follows_the_published_rule(po):
    """the variant itself: below edlib's 1 MB rule it IS the traceback (no split, same qualities as the unbanded shipped order); above,
    it splits the fragment in halves recursively (3 kb: once; 9 kb: 7 times) and still yields an optimal path (same distance: the
    realised identity moves by < 1e-3)"""
    em = po.ErrorModel(os.path.join(ROOT, "tksm_amd", "models", "badread", "nanopore2020.error.gz"))
    qm = po.QScoreModel(os.path.join(ROOT, "tksm_amd", "models", "badread", "nanopore2020.qscore.gz"))
    rs = np.random.RandomState(3)
    try:
        for L, splits in ((1000, 0), (3000, 1), (9000, 7)):
            raw = bytes(rs.choice(list(b"ACGT"), L).tolist())
            res = {}
            for v in (0, 3, 2):
                po.set_qscore_alignment_variant(v)
                seq, qual, idt, st = po.sequence_fragment(raw, 0.88, em, qm, True, 17, 1000 + L)
                res[v] = (seq, qual, idt, st.pad0)
            assert res[0][:3] == res[3][:3]                                  # the band does not change the shipped order's result
            assert res[2][0] == res[0][0] and res[2][3] == splits, (L, res[2][3])
            assert abs(res[2][2] - res[0][2]) < 1e-3
            if splits == 0:
                assert res[2][1] == res[0][1]
    finally:
        po.set_qscore_alignment_variant(0)


def test_path_choice_of_the_qscore_alignment_stays_inside_the_distribution_gates():
    import edlib_hole as eh
    procs = max(1, min(8, len(os.sched_getaffinity(0))))
    gate_tv = lambda n_pos: 0.01 + np.sqrt(60.0 / (np.pi * max(1.0, n_pos)))          # ~60 occupied q-score bins
    for L, n in ((3000, 160), (9000, 32)):
        rep = eh.measure(L, n, procs)
        ctl, opp, hir = rep[3], rep[1], rep[2]
        assert ctl["reads_differ"] == 0 and ctl["positions_differ"] == 0            # control: the band is not what is measured
        assert hir["hirschberg_splits_per_read"] >= 1.0
        # edlib as published: a handful of positions per ten thousand, nothing a histogram sees
        assert hir["positions_differ"] <= 2e-3 * hir["positions"], hir
        # (KS of two samples of n values moves in steps of 1 / n: at this size the gate is the step, the full-size run is in profiles/)
        assert hir["ks_identity"] <= max(0.02, 1.5 / n) and hir["max_abs_identity_difference"] < 1e-3, hir
        for op, share in (("=", 0.85), ("X", 0.06), ("I", 0.04)):
            assert hir["tv_qhist"][op] <= gate_tv(share * hir["positions"]), (L, op, hir)
        # the bound from the other side -- the OPPOSITE indel preference, which no edlib version has: ~1 % of the positions move, the
        # identity by < 1e-3, the histograms stay inside the same gates
        assert opp["positions_differ"] <= 3e-2 * opp["positions"] and opp["max_abs_identity_difference"] < 2e-3, opp
        assert opp["ks_identity"] <= max(0.05, 3.0 / n), opp
        for op, share in (("=", 0.85), ("X", 0.06), ("I", 0.04)):
            assert opp["tv_qhist"][op] <= gate_tv(share * opp["positions"]) + 0.01, (L, op, opp)
