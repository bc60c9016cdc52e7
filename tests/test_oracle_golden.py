"""CPU tests: the oracle (oracle/) pinned against the golden vectors captured from the real reference Python
(tests/golden/make_golden.py), plus the reference's own documentation known-answers.  No GPU."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ERR_MODEL, QS_MODEL

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


@pytest.mark.parametrize("L,n", [(300, 1200), (1000, 700), (3000, 160)])
def test_stochastic_path_matches_reference_distributions(po, oracle_models, L, n):
    """Distribution equivalence of the oracle's Badread path with the real reference (seeded fixtures of
    tests/golden/badread_reference_stats.npz): two-sample KS on per-read statistics.  Gate: KS D below the
    alpha = 0.001 critical value (the north star's 'p > 0.99' is not a usable threshold: p is uniform under H0)."""
    from scipy.stats import ks_2samp
    g = np.load(os.path.join(GOLDEN, "badread_reference_stats.npz"))
    sel = (g["L"] == L) & g["with_q"]
    em, qm = oracle_models["em"], oracle_models["qm"]
    ident = po.Identities(84.0, 5.5, 99.0)
    rs = np.random.RandomState(L)
    got = {k: [] for k in ("out_len", "identity", "target", "draws", "noop", "aligns", "X", "I", "D")}
    qh = np.zeros((3, 94), np.int64)
    import re
    for r in range(n):
        raw = bytes(rs.choice(list(b"ACGT"), L).tolist())
        read = 10_000_000 + L * 100_000 + r
        tgt = ident.get_identity(99, read)
        seq, qual, idt, st = po.sequence_fragment(raw, tgt, em, qm, True, 99, read)
        _, cig = po.nw_cigar(seq, raw)
        cnt = {"=": 0, "X": 0, "I": 0, "D": 0}
        pos = 0
        for m in re.finditer(r"(\d+)([=XID])", cig):
            k, t = int(m.group(1)), m.group(2)
            cnt[t] += k
            if t != "D":
                np.add.at(qh["=XI".index(t)], np.frombuffer(qual[pos:pos + k], np.uint8) - 33, 1)
                pos += k
        for k, v in (("out_len", len(seq)), ("identity", idt), ("target", tgt), ("draws", st.n_draws), ("noop", st.n_noop),
                     ("aligns", st.n_aligns), ("X", cnt["X"]), ("I", cnt["I"]), ("D", cnt["D"])):
            got[k].append(v)
        assert st.band_fail == 0
    m = int(sel.sum())
    crit = 1.95 * np.sqrt((n + m) / (n * m))          # KS critical value, alpha = 0.001
    for k in got:
        d = ks_2samp(got[k], g[k][sel]).statistic
        assert d < crit, (k, d, crit)
    # q-score histograms conditioned on the alignment op: total variation distance
    ref_qh = g[f"qhist_{L}"].astype(float)
    for row in range(3):
        a, b = qh[row] / max(1, qh[row].sum()), ref_qh[row] / max(1, ref_qh[row].sum())
        assert 0.5 * np.abs(a - b).sum() < 0.03, ("qhist", "=XI"[row])


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


def test_tail_noise_matches_reference_distributions(po):
    """oracle tail noise vs KDE_noise_generator.noise_seq run in the reference itself on the same synthetic model
    (tests/golden/make_tail_golden.py): share of reads with a tail, tail-length histogram, first base, base transitions;
    the fragment length past the last label reproduces the reference's len(ly) / ly[-1] factor (always empty here)."""
    from scipy.stats import chi2
    g = np.load(os.path.join(GOLDEN, "tail_reference_stats.npz"))
    tm = po.TailModel(os.path.join(GOLDEN, "tail_model_synth.json"))
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
        assert set(np.unique(lens)) <= set(vals.tolist()) | set(np.arange(0, 408, 8).tolist())
        allv = np.union1d(vals, np.unique(lens))
        a = np.array([cnt[vals == v].sum() for v in allv]); b = np.array([(lens == v).sum() for v in allv])
        assert chi2_two_sample(a, b) > 1e-3, (fl, "length")
        assert abs((lens == 0).mean() - cnt[vals == 0].sum() / n_ref) < 0.015
        assert chi2_two_sample(g[f"first_{fl}"], first) > 1e-3, (fl, "first base")
        for srow in range(4):
            assert chi2_two_sample(g[f"trans_{fl}"][srow], trans[srow]) > 1e-3, (fl, "transitions from", srow)
