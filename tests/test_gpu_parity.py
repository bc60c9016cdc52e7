"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs, and against the committed golden fixtures.  Bit-exact everywhere: the splice path is integer work,
and the stochastic path uses the same counter-based RNG and the same arithmetic as the oracle."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ERR_MODEL, QS_MODEL

pytestmark = pytest.mark.gpu
SEED = 1234


def _normalize(text):
    out = []
    for line in text.splitlines(keepends=True):
        if line[:1] in "@>" and len(line) > 37 and line[37] == " ":
            line = line[0] + "UUID" + line[37:]
        out.append(line)
    return "".join(out)


@pytest.fixture(scope="module")
def corpus_seqr():
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    s.get_reference_seqs([os.path.join(GOLDEN, "splice_corpus", "ref.fa")])
    yield s
    s.close()


@pytest.mark.parametrize("ext", ["fastq", "fasta"])
def test_splice_corpus_matches_reference_cli(corpus_seqr, ext):
    """perfect path vs the output of the reference's real main block (tests/golden/make_golden.py)."""
    mdf = open(os.path.join(GOLDEN, "splice_corpus", "mols.mdf")).read()
    b = corpus_seqr.batch_from_mdf(mdf)
    res = corpus_seqr.run(b, target="perfect", fastq=(ext == "fastq"), seed=SEED)
    rec, off = res.download()
    got = _normalize(rec.decode())
    want = open(os.path.join(GOLDEN, "splice_corpus", f"expected_perfect.{ext}")).read()
    assert got == want
    assert int(off[-1]) == len(rec) and res.n_reads == b.n_reads


def test_splice_corpus_matches_oracle_records(corpus_seqr, po):
    """same corpus, full records incl. the counter-based read ids, vs the C oracle."""
    mdf_path = os.path.join(GOLDEN, "splice_corpus", "mols.mdf")
    ref = po.get_reference_seqs([os.path.join(GOLDEN, "splice_corpus", "ref.fa")])
    b = corpus_seqr.batch_from_mdf(open(mdf_path).read())
    recs = corpus_seqr.run(b, target="perfect", fastq=True, seed=SEED, first_read_index=77).records()
    with open(mdf_path) as f:
        mols = list(po.mdf_generator(f))
    assert len(mols) == len(recs)
    for i, (mid, ivs) in enumerate(mols):
        seq = po.splice(ref, ivs)
        assert recs[i] == po.perfect_record(True, SEED, 77 + i, seq, mid), (i, mid)


def test_readme_known_answers():
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    s.add_contig("1", "AGTCCCGTAA")
    r = s.mdf_to_seq([("m1", [("1", 0, 4, "+", "2C,3T"), ("1", 6, 9, "+", "1G")])], target="perfect", fastq=False)
    assert r[0].split(b"\n")[1] == b"AGCTGGA"                  # README.md:233-251
    s.close()
    s = Sequencer(0)
    s.add_contig("1", "AGTC")
    r = s.mdf_to_seq([("m1", [("TT", 0, 2, "+", ""), ("1", 0, 4, "+", "")])], target="perfect", fastq=False)
    assert r[0].split(b"\n")[1] == b"TTAGTC"                   # README.md:253-270
    s.close()


def test_model_tables_match_oracle(seqr, oracle_models):
    em, qm = oracle_models["em"], oracle_models["qm"]
    t = seqr.error_model_tables()
    assert (t["k"], t["max_alts"], t["type"]) == (em.k, em.max_alts, em.type)
    assert np.array_equal(t["nalts"], em.nalts) and np.array_equal(t["cdf"], em.cdf) and np.array_equal(t["alts"], em.alts)
    q = seqr.qscore_model_tables()
    assert (q["n_slots"], q["kmer_size"]) == (len(qm.keys), qm.kmer_size)
    for k in ("keys", "row_off", "row_cnt", "cdf_pool", "q_pool"):
        assert np.array_equal(q[k], getattr(qm, k)), k


def test_identity_table_matches_scipy(seqr, po):
    from scipy.stats import beta
    t = seqr.identity_tables()
    a, b = po.beta_parameters(84.0, 5.5, 99.0)
    assert t["beta_a"] == a and t["beta_b"] == b and not t["constant"] and t["value"] == 0.99
    want = beta.ppf(np.arange(65537) / 65536.0, a, b)
    assert np.max(np.abs(t["qtab"] - want)) < 1e-10


def _random_genome_seqr(n_contigs=3, size=200_000, seed=5):
    from tksm_amd.sequence import Sequencer
    rs = np.random.RandomState(seed)
    s = Sequencer(0)
    ref = {}
    for c in range(n_contigs):
        seq = rs.choice(np.frombuffer(b"ACGT", np.uint8), size).tobytes()
        if c == 0:                       # N run + lower case + IUPAC so k-mers with non-ACGT bytes occur
            seq = seq[:5000] + b"N" * 300 + seq[5300:9000] + seq[9000:9500].lower() + b"RY" + seq[9502:]
        ref[f"chr{c + 1}"] = seq.decode()
        s.add_contig(f"chr{c + 1}", seq)
    return s, ref, rs


def _make_molecules(rs, ref, n, mean_len, literal=True):
    mols = []
    names = list(ref)
    for i in range(n):
        ivs = []
        total = max(40, int(rs.normal(mean_len, mean_len * 0.2)))
        k = int(rs.randint(1, 4))
        for j in range(k):
            ln = total // k
            c = names[rs.randint(len(names))]
            if i % 7 == 0 and c == "chr1":
                st = int(rs.randint(4800, 9600))       # touch the N run / lower-case / IUPAC region
            else:
                st = int(rs.randint(0, len(ref[c]) - ln))
            mods = "" if rs.rand() < 0.8 else f"{rs.randint(ln)}{'ACGTN'[rs.randint(5)]}"
            ivs.append((c, st, st + ln, "+-"[rs.randint(2)], mods))
        if literal and i % 5 == 0:
            ivs.append(("A" * int(rs.randint(5, 40)), 0, 40, "+", ""))
        mols.append((f"mol{i}", ivs))
    return mols


@pytest.mark.parametrize("path", ["fast", "slow"])
@pytest.mark.parametrize("mean_len,n,compute_q", [(300, 96, True), (1000, 64, True), (1000, 32, False), (2600, 24, True)])
def test_badread_bit_exact_vs_oracle(oracle_models, po, monkeypatch, mean_len, n, compute_q, path):
    """whole records of the stochastic path, GPU vs oracle, same (seed, read index).  mean_len 2600 exercises
    the random 1000-base window re-estimation (py/tksm_badread.py:417-432).  path: "fast" = k_err + bit-parallel
    k_aln (reads touching N / IUPAC bytes still take the wave-wide kernel), "slow" = wave-wide kernel for all."""
    monkeypatch.setenv("TKSMSEQ_FORCE_SLOW", "1" if path == "slow" else "0")
    s, ref, rs = _random_genome_seqr()
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    mols = _make_molecules(rs, ref, n, mean_len)
    text = "".join(f"+{m}\t1\t\n" + "".join(f"{c}\t{a}\t{b}\t{st}\t{md}\n" for c, a, b, st, md in ivs) for m, ivs in mols)
    batch = s.batch_from_mdf(text)
    res = s.run(batch, target="badread", fastq=True, compute_qual=compute_q, seed=SEED, first_read_index=1000, stride=3,
                collect_stats=True)
    recs = res.records()
    ist, dst = res.stats()
    idt = s.identity_tables()
    ident = po.Identities(84.0, 5.5, 99.0, qtab=idt["qtab"])
    em, qm = oracle_models["em"], oracle_models["qm"]
    for i, (mid, ivs) in enumerate(mols):
        raw = po.splice(ref, ivs)
        want, st = po.badread_record(True, SEED, 1000 + 3 * i, raw, ident, em, qm, compute_q, mid)
        assert st.band_fail == 0
        got_stats = (ist[i, 0], ist[i, 1], ist[i, 2], ist[i, 3], ist[i, 4], ist[i, 5], ist[i, 6])
        want_stats = (st.n_draws, st.change_count, st.n_aligns, st.frag_len, st.new_len, st.start_trim, st.end_trim)
        assert got_stats == want_stats, (i, got_stats, want_stats, dst[i], st.errors, st.target_identity)
        assert dst[i, 1] == st.target_identity and dst[i, 0] == st.errors
        assert recs[i] == want, (i, mid)
    s.close()
