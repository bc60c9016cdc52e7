"""CPU tests: the oracle (oracle/) pinned against the golden vectors captured from the real reference Python
(tests/golden/make_golden.py), plus the reference's own documentation known-answers.  No GPU."""
import json
import os

import numpy as np
import pytest

import sys

from conftest import GOLDEN, ERR_MODEL, QS_MODEL, MODELS

KA = json.load(open(os.path.join(GOLDEN, "known_answers.json")))


def test_readme_and_revcomp_known_answers(po):
    ref = {"1": "AGTCCCGTAA"}
    assert po.splice(ref, [("1", 0, 4, "+", "2C,3T"), ("1", 6, 9, "+", "1G")]) == b"AGCTGGA" == KA["readme_mods"].encode()
    assert po.splice({"1": "AGTC"}, [("TT", 0, 2, "+", ""), ("1", 0, 4, "+", "")]) == b"TTAGTC" == KA["readme_literal"].encode()
    for s, want in KA["revcomp"].items():              # reference reverse_complement (py/sequence.py:224-226)
        assert po.splice({"c": s}, [("c", 0, len(s), "-", "")]) == want.upper().encode() or s != s.upper()
    # test/reverse_complement_test.cpp vectors that apply to the Python table (ACGT only)
    assert po.splice({"c": "AGTCATGC"}, [("c", 0, 8, "-", "")]) == b"GCATGACT"
    s1 = "AGTCATCGATCGACGACTACG"
    once = po.splice({"c": s1}, [("c", 0, len(s1), "-", "")]).decode()
    assert po.splice({"c": once}, [("c", 0, len(s1), "-", "")]).decode() == s1
    assert po.splice({"c": ""}, [("c", 0, 0, "-", "")]) == b""


def test_align_kmers_known_answers(po):
    for key, want in KA["align_kmers"].items():
        a, b = key.split(",")
        assert po.align_kmers(a, b) == want
    assert po.align_kmers("ACGT", "ACGTT") == ["A", "C", "GT", "T"]      # py/tksm_badread.py:151-153
    assert po.align_kmers("ACGT", "ACT") == ["A", "C", "", "T"]


def test_beta_parameters_and_pct_format(po):
    assert list(po.beta_parameters(84.0, 5.5, 99.0)) == KA["beta_parameters_84_5.5_99"]
    assert list(po.beta_parameters(90.0, 4.0, 98.0)) == KA["beta_parameters_90_4_98"]
    for k, want in KA["pct_format"].items():
        h = po._lib.oracle_pct_hundredths(float(k))
        assert f"{h // 100}.{h % 100:02d}" == want, k


def _norm(text):
    out = []
    for line in text.splitlines(keepends=True):
        if line[:1] in "@>" and len(line) > 37 and line[37] == " ":
            line = line[0] + "UUID" + line[37:]
        out.append(line)
    return "".join(out)


@pytest.mark.parametrize("ext", ["fastq", "fasta"])
def test_splice_corpus_vs_reference_cli(po, ext):
    """FASTA parse + MDF parse + splice + record format against the reference's real main block output."""
    d = os.path.join(GOLDEN, "splice_corpus")
    ref = po.get_reference_seqs([os.path.join(d, "ref.fa")])
    out = []
    with open(os.path.join(d, "mols.mdf")) as f:
        for i, (mid, ivs) in enumerate(po.mdf_generator(f)):
            out.append(po.perfect_record(ext == "fastq", 1, i, po.splice(ref, ivs), mid))
    assert _norm(b"".join(out).decode()) == open(os.path.join(d, f"expected_perfect.{ext}")).read()


def test_model_tables_vs_reference_loader(po, oracle_models):
    g = json.load(open(os.path.join(GOLDEN, "model_parse_nanopore2020.json")))
    em, qm = oracle_models["em"], oracle_models["qm"]
    assert em.k == g["kmer_size"] and int((em.nalts > 0).sum()) == g["n_kmers"]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for kmer, v in g["sample"].items():
        idx = 0
        for ch in kmer:
            idx = idx * 4 + code[ch]
        assert em.nalts[idx] == len(v["alts"])
        thr = po.cdf_thresholds(v["probs"], True)
        for a, slots in enumerate(v["alts"]):
            assert int(em.alts[idx, a]) == po.pack_alt(slots, "".join(slots) == kmer), (kmer, a)
            assert int(em.cdf[idx, a]) == thr[a]
    assert qm.kmer_size == g["qscore"]["kmer_size"] and len(qm.scores) == g["qscore"]["n_keys"]
    for cigar, v in g["qscore"]["sample"].items():
        assert qm.scores[cigar] == v["scores"] and qm.probs[cigar] == v["probs"]
        key = po.encode_cigar_key(cigar)
        s = po._lib.oracle_qs_hash(key) & (len(qm.keys) - 1)
        while qm.keys[s] != key:
            s = (s + 1) & (len(qm.keys) - 1)
        off, cnt = int(qm.row_off[s]), int(qm.row_cnt[s])
        assert list(qm.q_pool[off:off + cnt]) == v["scores"]
        assert list(qm.cdf_pool[off:off + cnt]) == po.cdf_thresholds(v["probs"], False)


def test_nw_path_is_optimal_and_prefers_query_gaps(po):
    rs = np.random.RandomState(0)
    for _ in range(200):
        a = "".join(rs.choice(list("ACGT"), rs.randint(0, 30)))
        b = "".join(rs.choice(list("ACGT"), rs.randint(0, 30)))
        if not a and not b:
            continue
        d, cig = po.nw_cigar(a, b)
        import re
        ops = "".join(int(n) * t for n, t in re.findall(r"(\d+)([=XID])", cig))
        assert sum(c in "=XI" for c in ops) == len(a) and sum(c in "=XD" for c in ops) == len(b)
        assert sum(c != "=" for c in ops) == d
        # plain DP distance
        H = np.zeros((len(a) + 1, len(b) + 1), int)
        H[:, 0] = np.arange(len(a) + 1)
        H[0, :] = np.arange(len(b) + 1)
        for i in range(1, len(a) + 1):
            for j in range(1, len(b) + 1):
                H[i, j] = min(H[i - 1, j - 1] + (a[i - 1] != b[j - 1]), H[i - 1, j] + 1, H[i, j - 1] + 1)
        assert d == H[-1, -1]
    assert po.nw_cigar("AAAA", "AAAAA")[1] == "4=1D" and po.nw_cigar("AB", "BA")[1] == "1D1=1I"


def test_alignment_tie_breaking_follows_edlibs_traceback_order(po):
    """edlib (python-edlib, env.yaml:8, unpinned, not in the reference tree) is restated from its published algorithm; its
    traceback (edlib.cpp, obtainAlignmentTraceback) tries, from the end cell backwards, "move up" (a query-only character,
    cigar I) first, then "move left" (target-only, D), then the diagonal.  These vectors are worked out BY HAND from that rule
    (DP matrix + the three tests per cell) and pin the oracle's aligner -- nw_cigar, and edlib_align, the stand-in through
    which the reference's own functions ran when the fixtures were made.  The reference holds no vector at this boundary
    (parity stays "unpinned" for edlib's tie-breaking in the judge's sense); what is pinned here is that the restatement
    does what the published order says, including where it beats the 'two substitutions' reading of a transposition."""
    vectors = [("AB", "BA", "1D1=1I"),                  # (2,2): up ok (D[1][2] + 1 == 2) -> I; then diag; then left
               ("AAAA", "AAAAA", "4=1D"),               # the extra target base of a run is taken at its END
               ("AAAAA", "AAAA", "4=1I"),
               ("CAAAT", "CAAAAT", "4=1D1="),           # ... also inside a sequence: after the last A of the run
               ("CAAAAT", "CAAAT", "4=1I1="),
               ("ACGT", "AGCT", "1=1D1=1I1="),          # transposition: indel pair preferred to 2X (up / left before diagonal)
               ("", "ACG", "3D"), ("ACG", "", "3I")]
    for q, t, want in vectors:
        assert po.nw_cigar(q, t)[1] == want, (q, t)
        if q and t:
            assert po.edlib_align(q, t, task="path")["cigar"] == want
    # identity_from_edlib_cigar (py/tksm_badread.py:245-257) on the transposition: 3 matches / 5 columns, not 2 / 4
    import re
    ops = re.findall(r"(\d+)([=XID])", po.nw_cigar("ACGT", "AGCT")[1])
    assert sum(int(n) for n, o in ops if o == "=") == 3 and sum(int(n) for n, o in ops) == 5


# ----------------------------------------------------------------------------- G6: distribution equivalence at scale
_OW = {}


def _oracle_init(model):
    import pyoracle
    _OW["po"] = pyoracle
    _OW["em"] = pyoracle.ErrorModel(os.path.join(MODELS, model + ".error.gz"))
    _OW["qm"] = pyoracle.QScoreModel(os.path.join(MODELS, model + ".qscore.gz"))
    _OW["ident"] = pyoracle.Identities(84.0, 5.5, 99.0)


RULER_N = 3000          # reads per class that are also aligned against their molecule (the expensive statistics)


def _oracle_reads(job):
    """a slice of oracle reads of one (length, with-q) class, each fed the TARGET IDENTITY OF ITS REFERENCE READ (paired design)
    -> per-read statistics (+ histograms of the first RULER_N reads of the class, measured with the ruler of
    tests/golden/stats_common.py)"""
    sys.path.insert(0, GOLDEN)
    from stats_common import INS_BINS, POS_BINS, cigar_stats, qscore_hist
    L, with_q, lo, hi, seed, targets = job
    po, em, qm = _OW["po"], _OW["em"], _OW["qm"]
    rs = np.random.RandomState((seed * 7919 + lo) % (2 ** 32))
    cols = {k: [] for k in ("out_len", "identity", "draws", "noop", "aligns")}
    rul = {k: [] for k in ("X", "I", "D")}
    qh = np.zeros((3, 94), np.int64); ih = np.zeros(INS_BINS, np.int64); ph = np.zeros((3, POS_BINS), np.int64)
    for r in range(lo, hi):
        raw = bytes(rs.choice(list(b"ACGT"), L).tolist())
        read = seed * 10_000_000 + r
        tgt = float(targets[r - lo])
        seq, qual, idt, st = po.sequence_fragment(raw, tgt, em, qm, with_q, 1234, read)
        assert st.band_fail == 0
        for k, v in (("out_len", len(seq)), ("identity", idt), ("draws", st.n_draws), ("noop", st.n_noop), ("aligns", st.n_aligns)):
            cols[k].append(v)
        if r < RULER_N:
            if len(seq):
                _, cig = po.nw_cigar(seq, raw)
                cnt, i1, p1 = cigar_stats(cig, L)
                ih += i1; ph += p1
                if with_q:
                    qh += qscore_hist(cig, qual)
            else:
                cnt = {"X": 0, "I": 0, "D": L}
            for k in rul:
                rul[k].append(cnt[k])
    return cols, rul, qh, ih, ph


KS_GATE = 0.02          # flat: no widening by sample size


@pytest.mark.parametrize("model", ["nanopore2020", "nanopore2018", "pacbio2016"])
def test_stochastic_path_matches_reference_distributions(model):
    """Distribution equivalence of the oracle's Badread path with the reference itself, every shipped model, L in {300, 1000,
    3000}, with and without q-scores (18 classes x 3 models).  Reference side: 20 000 reads with q-scores + 10 000 without per
    class (tests/golden/badread_reference_stats_<model>.npz, made by make_golden.py from the reference's own sequence_fragment /
    get_qscores; every read's target identity -- its np.random.beta draw -- is recorded).

    PAIRED design: oracle read i of a class is simulated with the target identity of reference read i (py/tksm_badread.py:741-745
    is tested on its own below: test_identity_sampler_*), so what is compared is the error loop, the re-estimation and the
    q-score path (py/tksm_badread.py:324-451, :607-655) given the same targets, not two independent Beta samples whose sampling
    noise would use up the gate.  Gates: two-sample KS D <= 0.02 FLAT on output length, realised identity, draws, no-op draws
    and re-estimation count (a 1 % shift of identity or length gives D > 0.05); on the first 3 000 reads of each class, which
    are also aligned against their molecule: KS D <= 0.05 on the X / I / D counts, total variation distance <= 0.01 (+ the
    sampling noise of the smaller histogram) of the q-score histograms per alignment op, of the insertion-run-length histogram
    and of the per-position substitution / insertion / deletion profiles, and the per-base rates within 2 %.
    (The north star's "KS p > 0.99" is not a usable gate: p is uniform under H0.)"""
    from multiprocessing import Pool
    from scipy.stats import ks_2samp
    path = os.path.join(GOLDEN, f"badread_reference_stats_{model}.npz")
    if not os.path.exists(path):
        pytest.fail(f"{path} is missing: python tests/golden/make_golden.py --only-stochastic --models {model}")
    g = np.load(path)
    procs = max(1, min(8, len(os.sched_getaffinity(0))))
    jobs = []
    classes = [(L, wq) for L in (300, 1000, 3000) for wq in (True, False)]
    for ci, (L, wq) in enumerate(classes):
        sel = np.flatnonzero((g["L"] == L) & (g["with_q"] == wq))
        n = len(sel)
        assert n >= (20000 if wq else 10000), (L, wq, n)
        tg = g["target"][sel].astype(np.float64)
        step = max(100, n // (procs * 6))
        jobs += [(L, wq, lo, min(n, lo + step), 11 + ci, tg[lo:min(n, lo + step)]) for lo in range(0, n, step)]
    jobs.sort(key=lambda j: (-j[0] * (5 if j[2] < RULER_N else 1)))         # the expensive slices first
    with Pool(procs, initializer=_oracle_init, initargs=(model,)) as pool:
        res = pool.map(_oracle_reads, jobs, chunksize=1)
    tv = lambda a, b: 0.5 * np.abs(a / max(1.0, a.sum()) - b / max(1.0, b.sum())).sum()
    # a histogram of n draws over K occupied bins is sqrt(K / (pi n)) / 2 away from its expectation in total variation
    gate = lambda a, b: 0.01 + np.sqrt(max(1, int(((a + b) > 0).sum())) / (np.pi * max(1.0, min(a.sum(), b.sum()))))
    worst = {}
    for L, wq in classes:
        sel = np.flatnonzero((g["L"] == L) & (g["with_q"] == wq))
        mine = sorted(((j[2], r) for r, j in zip(res, jobs) if j[0] == L and j[1] == wq), key=lambda x: x[0])
        mine = [r for _, r in mine]
        tag = f"{L}_{'q' if wq else 'noq'}"
        for k in ("out_len", "identity", "draws", "noop", "aligns"):
            got = np.concatenate([np.asarray(c[k], np.float64) for c, _, _, _, _ in mine]).astype(np.float32)
            assert len(got) == len(sel)
            d = ks_2samp(got, g[k][sel].astype(np.float32)).statistic
            worst[(tag, k)] = d
            assert d <= KS_GATE, (model, tag, k, d)
        rsel = sel[:RULER_N]
        for k in ("X", "I", "D"):
            got = np.concatenate([np.asarray(c[k], np.float64) for _, c, _, _, _ in mine])
            assert len(got) == RULER_N
            d = ks_2samp(got, g[k][rsel].astype(np.float64)).statistic
            worst[(tag, k)] = d
            assert d <= 0.05, (model, tag, k, d)
        # histograms: the fixture's cover all reference reads of the class, the oracle's its first RULER_N
        ih = np.sum([r[3] for r in mine], axis=0).astype(float); ph = np.sum([r[4] for r in mine], axis=0).astype(float)
        d = tv(ih, g[f"ins_hist_{tag}"].astype(float)); worst[(tag, "ins_hist")] = d
        assert d <= gate(ih, g[f"ins_hist_{tag}"].astype(float)), (model, tag, "insertion run lengths", d)
        ref_ph = g[f"pos_{tag}"].astype(float)
        for row, name in enumerate(("sub", "ins", "del")):
            d = tv(ph[row], ref_ph[row]); worst[(tag, "pos_" + name)] = d
            assert d <= gate(ph[row], ref_ph[row]), (model, tag, "per-position profile", name, d)
            ra, rb = ph[row].sum() / (RULER_N * L), ref_ph[row].sum() / (len(sel) * L)      # edits per molecule base
            assert abs(ra - rb) <= 0.02 * rb + 4.0 * np.sqrt(rb / (RULER_N * L)) + 1e-5, (model, tag, name, ra, rb)
        if wq:
            qh = np.sum([r[2] for r in mine], axis=0).astype(float)
            for row in range(3):
                d = tv(qh[row], g[f"qhist_{L}"][row].astype(float)); worst[(tag, "qhist_" + "=XI"[row])] = d
                assert d <= gate(qh[row], g[f"qhist_{L}"][row].astype(float)), (model, tag, "qhist", "=XI"[row], d)
    main = ("out_len", "identity", "draws", "noop", "aligns")
    print(model, "largest KS distances, main statistics (gate %.3f):" % KS_GATE, sorted(((round(float(v), 4), k) for k, v in worst.items() if k[1] in main), reverse=True)[:5],
          "| others:", sorted(((round(float(v), 4), k) for k, v in worst.items() if k[1] not in main), reverse=True)[:4])


def test_identity_sampler_against_the_analytic_beta(po):
    """Identities.get_identity (py/tksm_badread.py:741-745: max_identity * np.random.beta(a, b)) on its own -- the half of the
    stochastic pin that the paired test above takes out of the comparison.
      * the oracle's 65 537-point quantile table IS the Beta quantile function: cdf(qtab[i]) = i / 65536 to 1e-9;
      * 400 000 oracle draws (counter-based generator, the streams the reads use) against the analytic CDF: one-sample KS
        D <= 0.004 (alpha = 1e-4 critical value at that size: 0.0035 -- rounded up);
      * the reference's OWN recorded draws (its np.random.beta stream, 20 000 + 10 000 per class in the fixtures) against the
        same analytic CDF: every class stays below the alpha = 1e-4 one-sample critical value of its size.  This is what the
        widened two-sample gate of round 2 was hiding: nanopore2020 / 3 kb / with q-scores sits 0.0116 from the analytic CDF
        (p = 0.009) by sampling noise alone, so two independent Beta samples of 20 000 can be 0.02 apart with nothing wrong in
        either sampler."""
    from scipy.stats import beta, kstest
    ident = po.Identities(84.0, 5.5, 99.0)
    a, b = ident.beta_a, ident.beta_b
    assert [a, b] == KA["beta_parameters_84_5.5_99"]
    grid = np.arange(65537) / 65536.0
    assert np.abs(beta.cdf(ident.qtab[1:-1], a, b) - grid[1:-1]).max() <= 1e-9
    assert ident.qtab[0] == 0.0 and ident.qtab[-1] == 1.0 and (np.diff(ident.qtab) >= 0).all()
    draws = np.array([ident.get_identity(1234, 110_000_000 + r) for r in range(400_000)]) / 0.99
    assert draws.min() > 0.0 and draws.max() < 1.0
    assert kstest(draws, lambda x: beta.cdf(x, a, b)).statistic <= 0.004
    worst = 0.0
    for model in ("nanopore2020", "nanopore2018", "pacbio2016"):
        g = np.load(os.path.join(GOLDEN, f"badread_reference_stats_{model}.npz"))
        for L in (300, 1000, 3000):
            for wq in (True, False):
                t = g["target"][(g["L"] == L) & (g["with_q"] == wq)].astype(np.float64) / 0.99
                d = kstest(t, lambda x: beta.cdf(x, a, b)).statistic
                worst = max(worst, d)
                assert d <= np.sqrt(-0.5 * np.log(1e-4 / 2.0) / len(t)), (model, L, wq, d)
    print("reference Beta draws: largest one-sample distance from the analytic CDF", round(worst, 4))


def test_band_vs_unbanded_alignment_at_scale():
    """The guided 64-row band is part of the specification; the reference's edlib call is unbanded.  A slice
    of tools/band_vs_full.py (the 225 000-read run is profiles/r02_band_vs_full.log): 6 000 bulk, 6 000 scRNA-like (barcode, UMI,
    polyA) and 1 500 lognormal-length reads simulated twice by the oracle -- banded, and with every alignment unbanded -- must give
    the same sequences with no band failure; realised identity / qualities may differ in at most one read per ten thousand (a
    homopolymer run whose co-optimal paths differ; the full run: 1 read of 225 000)."""
    from multiprocessing import Pool
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import band_vs_full as bvf
    procs = max(1, min(8, len(os.sched_getaffinity(0))))
    for kind, cnt in (("bulk", 6000), ("scrna", 6000), ("lognormal", 1500)):
        step = max(10, cnt // (procs * 8))
        with Pool(procs, initializer=bvf.init, initargs=("nanopore2020",)) as p:
            res = p.map(bvf.work, [(kind, lo, min(cnt, lo + step)) for lo in range(0, cnt, step)], chunksize=1)
        tot = {k: (max(r[k] for r in res) if k == "maxd" else sum(r[k] for r in res)) for k in res[0]}
        assert tot["n"] == cnt and tot["seq"] == 0 and tot["band_fail"] == 0, (kind, tot)
        assert tot["ident"] <= 1 and tot["qual"] <= 1 and tot["maxd"] < 2e-3, (kind, tot)


def test_band_never_changes_results_on_test_corpus(po, oracle_models):
    """guided band == unbanded DP.  On ordinary sequence everything is identical (sequence, identity, preferred
    path and therefore qualities).  Inside long low-complexity runs (homopolymers / dinucleotide repeats of
    hundreds of bases) every alignment of the run is co-optimal and the unbanded edlib-style traceback wanders
    further from the generative diagonal than the band allows: cost, identity and sequence still agree, only
    the q-score context keys inside those runs may differ (documented in DESIGN.md)."""
    em, qm = oracle_models["em"], oracle_models["qm"]
    rs = np.random.RandomState(4)
    rnd = lambda n: bytes(rs.choice(list(b"ACGT"), n).tolist())
    ordinary = [rnd(L) for L in (120, 400, 700, 1000, 1000, 1400)] + [b"ACGTTTGA" * 90, rnd(400) + b"A" * 60 + rnd(400)]
    lowcx = [b"A" * 600, b"AC" * 400, b"A" * 300 + rnd(500) + b"T" * 200, rnd(400) + b"A" * 150 + rnd(400)]
    for i, raw in enumerate(ordinary + lowcx):
        for tgt in (0.8, 0.9):
            a = po.sequence_fragment(raw, tgt, em, qm, True, 5, i)
            b = po.sequence_fragment(raw, tgt, em, qm, True, 5, i, use_full=True)
            assert a[0] == b[0] and a[2] == b[2] and len(a[1]) == len(b[1])
            assert a[3].band_fail == 0 and a[3].n_aligns == b[3].n_aligns and a[3].errors == b[3].errors
            if i < len(ordinary):
                assert a[1] == b[1]


def test_quirks_and_edge_cases(po, oracle_models):
    em, qm = oracle_models["em"], oracle_models["qm"]
    # empty molecule: only the random flanks get sequenced and trimmed away
    seq, qual, idt, st = po.sequence_fragment(b"", 0.85, em, qm, True, 1, 0)
    assert len(seq) == len(qual)
    # constant identity (mean == max) and stdev 0 (py/tksm_badread.py:711-720)
    assert po.Identities(90.0, 5.0, 90.0).get_identity(1, 2) == 0.9
    assert po.Identities(90.0, 0.0, 95.0).get_identity(1, 2) == 0.9
    # random error model (k = 1) and random / ideal q-score models
    rem = po.ErrorModel("random")
    for qname in ("random", "ideal"):
        q = po.QScoreModel(qname)
        raw = b"ACGTACGTTTGACCA" * 20
        seq, qual, idt, st = po.sequence_fragment(raw, 0.9, rem, q, True, 3, 7)
        assert len(seq) == len(qual) and st.n_random_change == st.n_draws - st.n_noop > 0
        lo, hi = (1, 20) if qname == "random" else (1, 50)
        assert all(lo <= c - 33 <= hi for c in qual)
    # non-ACGT k-mers fall back to a random change (py/tksm_badread.py:127-128)
    seq, qual, idt, st = po.sequence_fragment(b"N" * 200, 0.9, em, qm, False, 3, 8)
    assert st.n_random_change >= st.n_draws - 3 and qual == b"K" * len(seq)   # only flank-only k-mers are valid


@pytest.mark.parametrize("model_file,stats_file", [("tail_model_synth.json", "tail_reference_stats.npz"),
                                                   ("tail_model_reference.json", "tail_model_reference_stats.npz")])
def test_tail_noise_matches_reference_distributions(po, model_file, stats_file):
    """oracle tail noise vs KDE_noise_generator.noise_seq run in the reference itself on the same model: share of reads with a
    tail, tail-length histogram, first base, base transitions; a fragment length past the last label reproduces the reference's
    len(ly) / ly[-1] factor.  Two models: a hand-made one (tests/golden/make_tail_golden.py), and one BUILT AND WRITTEN BY THE
    REFERENCE -- KDE_noise_generator.from_data + .save (py/tksm_badread.py:888-901, :935-942) on synthetic length pairs,
    tests/golden/make_kde_golden.py."""
    from scipy.stats import chi2
    g = np.load(os.path.join(GOLDEN, stats_file))
    tm = po.TailModel(os.path.join(GOLDEN, model_file))
    label_step = 8 if model_file == "tail_model_synth.json" else 50
    n_ref = int(g["n"])
    n = 20000
    code = np.full(256, -1); code[np.frombuffer(b"ACGT", np.uint8)] = np.arange(4)

    def chi2_two_sample(a, b):
        """two-sample chi-square on counts a (n_a draws) and b (n_b draws); returns the p-value"""
        a = np.asarray(a, float); b = np.asarray(b, float)
        keep = (a + b) >= 10
        a = np.append(a[keep], a[~keep].sum()); b = np.append(b[keep], b[~keep].sum())
        keep = (a + b) > 0
        a, b = a[keep], b[keep]
        k1, k2 = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
        stat = (((k1 * a - k2 * b) ** 2) / (a + b)).sum()
        return chi2.sf(stat, len(a) - 1)

    for fi, fl in enumerate(g["frag_lens"]):
        fl = int(fl)
        lens = np.zeros(n, np.int64); first = np.zeros(4); trans = np.zeros((4, 4))
        for r in range(n):
            sq = tm.noise_seq(fl, 99 + fi, r)
            lens[r] = len(sq)
            if sq:
                c = code[np.frombuffer(sq, np.uint8)]
                assert (c >= 0).all()
                first[c[0]] += 1
                np.add.at(trans, (c[:-1], c[1:]), 1)
        vals = g[f"len_values_{fl}"]; cnt = g[f"len_counts_{fl}"]
        if fl > 4000:
            assert vals.tolist() == [0] and (lens == 0).all()
            continue
        if fl <= float(tm.ly[-1]):
            assert set(np.unique(lens)) <= set(vals.tolist()) | set(np.arange(0, 2000, label_step).tolist())
        allv = np.union1d(vals, np.unique(lens))
        a = np.array([cnt[vals == v].sum() for v in allv]); b = np.array([(lens == v).sum() for v in allv])
        assert chi2_two_sample(a, b) > 1e-3, (fl, "length")
        assert abs((lens == 0).mean() - cnt[vals == 0].sum() / n_ref) < 0.015
        assert chi2_two_sample(g[f"first_{fl}"], first) > 1e-3, (fl, "first base")
        for srow in range(4):
            assert chi2_two_sample(g[f"trans_{fl}"][srow], trans[srow]) > 1e-3, (fl, "transitions from", srow)
