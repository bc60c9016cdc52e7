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


@pytest.mark.parametrize("path", ["fast", "fast-hbm", "fast-small", "fast-tail", "fast-tail-narrow", "slow"])
@pytest.mark.parametrize("mean_len,n,compute_q", [(300, 96, True), (1000, 64, True), (1000, 32, False), (2600, 24, True), (9000, 6, True)])
def test_badread_bit_exact_vs_oracle(oracle_models, po, monkeypatch, mean_len, n, compute_q, path):
    """whole records of the stochastic path, GPU vs oracle, same (seed, read index).  mean_len 2600 exercises
    the random 1000-base window re-estimation (py/tksm_badread.py:417-432).  path: "fast" = what the large rounds of large batches
    run: k_loop (one lane per read) in every round, k_aln with 16 stored rows and its 64-row follow-up pass, k_err per length
    bucket (reads touching N / IUPAC bytes still take the wave-wide kernel); "fast-hbm" = the same with k_err's long-read variant for
    every length; "fast-small" = the latency-bound variants small rounds switch to (k_loopw: one wave per read; every alignment with
    all 64 rows stored; one k_err launch); "fast-tail" = the default knobs, with which a batch this small is handed to the straggler
    kernel after its first round (k_loopw<true>: every later visit of a read on one wave, alignments by band_align on the wave);
    "fast-tail-narrow" = the same with room for 700 columns on the wave, so that wider windows take the regular route (job, k_alnf)
    for that visit and the straggler kernel picks the read up again in the next round; "slow" = wave-wide kernel for all."""
    monkeypatch.setenv("TKSMSEQ_FORCE_SLOW", "1" if path == "slow" else "0")
    if not path.startswith("fast-tail"):
        monkeypatch.setenv("TKSMSEQ_TAIL_WAVE", "0")
    if path == "fast-tail-narrow":
        monkeypatch.setenv("TKSMSEQ_TAIL_WCAP", "700")
    if path in ("fast", "fast-hbm"):
        monkeypatch.setenv("TKSMSEQ_SMALL_ALN", "0"); monkeypatch.setenv("TKSMSEQ_SMALL_ROUND", "0"); monkeypatch.setenv("TKSMSEQ_TAIL_CUT", "0")
        monkeypatch.setenv("TKSMSEQ_WAVE_LOOP", "0")
    if path == "fast-hbm":          # the long-read variant of k_err (fragment state edited in HBM) for every length
        monkeypatch.setenv("TKSMSEQ_HBM_STATE_LEN", "0")
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
    ident = po.Identities(84.0, 5.5, 99.0)
    em, qm = oracle_models["em"], oracle_models["qm"]
    for i, (mid, ivs) in enumerate(mols):
        raw = po.splice(ref, ivs)
        want, st = po.badread_record(True, SEED, 1000 + 3 * i, raw, ident, em, qm, compute_q, mid)
        assert st.band_fail == 0
        got_stats = (ist[i, 0], ist[i, 1], ist[i, 2], ist[i, 3], ist[i, 4], ist[i, 5], ist[i, 6])
        want_stats = (st.n_draws, st.change_count, st.n_aligns, st.frag_len, st.new_len, st.start_trim, st.end_trim)
        assert got_stats == want_stats, (i, got_stats, want_stats, dst[i], st.errors, st.target_identity)
        # the oracle tabulates the Beta quantile function with scipy, the library with its own routine (1e-10 apart:
        # test_identity_table_matches_scipy): the target agrees to that, everything downstream of it bit for bit
        assert abs(dst[i, 1] - st.target_identity) <= 1e-9 and dst[i, 0] == st.errors
        assert recs[i] == want, (i, mid)
    s.close()


@pytest.mark.parametrize("compute_q,wcap", [(True, None), (False, None), (True, 700)])
def test_predicted_stragglers_on_their_own_stream_bit_exact_vs_oracle(oracle_models, po, monkeypatch, capfd, compute_q, wcap):
    """a batch with a long tail of predicted visits (length x (1 - target identity) > 4 x the batch's median): those reads get their
    straggler waves at round 0, on a stream of their own, and the regular rounds pass them by (api.cpp: predicted stragglers);
    records and per-read statistics equal the oracle's whatever stream ran the read.  wcap 700: the straggler waves have room for 700
    columns only, so every early read hands itself to the exact kernel at its first re-estimation window (1000 slots) -- through the
    early reads' own slow list, merged into the batch's when the host joins the side stream"""
    monkeypatch.setenv("TKSMSEQ_FORCE_SLOW", "0"); monkeypatch.setenv("TKSMSEQ_EARLY_TAIL", "8"); monkeypatch.setenv("TKSMSEQ_VERBOSE", "1")
    if wcap:
        monkeypatch.setenv("TKSMSEQ_TAIL_WCAP", str(wcap))
    s, ref, rs = _random_genome_seqr()
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    mols = _make_molecules(rs, ref, 150, 300) + _make_molecules(rs, ref, 6, 5000)
    mols = [(f"m{i}", ivs) for i, (_, ivs) in enumerate(mols)]
    text = "".join(f"+{m}\t1\t\n" + "".join(f"{c}\t{a}\t{b}\t{st}\t{md}\n" for c, a, b, st, md in ivs) for m, ivs in mols)
    batch = s.batch_from_mdf(text)
    capfd.readouterr()
    res = s.run(batch, target="badread", fastq=True, compute_qual=compute_q, seed=SEED, first_read_index=77, stride=1, collect_stats=True)
    err = capfd.readouterr().err
    recs = res.records()
    ist, dst = res.stats()
    import re
    got = re.search(r"predicted stragglers on their own stream: (\d+)", err)
    assert got and 1 <= int(got.group(1)) <= 8, err[-600:]
    slow = re.search(r"slow-path reads (\d+)", err)
    if wcap:
        assert slow and int(slow.group(1)) >= int(got.group(1)), err[-600:]     # the early reads went to the exact kernel
    ident = po.Identities(84.0, 5.5, 99.0)
    em, qm = oracle_models["em"], oracle_models["qm"]
    for i, (mid, ivs) in enumerate(mols):
        raw = po.splice(ref, ivs)
        want, st = po.badread_record(True, SEED, 77 + i, raw, ident, em, qm, compute_q, mid)
        assert (ist[i, 0], ist[i, 1], ist[i, 2]) == (st.n_draws, st.change_count, st.n_aligns), (i, ist[i, :8], st.n_draws, st.change_count, st.n_aligns)
        assert dst[i, 0] == st.errors
        assert recs[i] == want, (i, mid)
    s.close()


def test_cli_end_to_end_matches_goldens_and_oracle(tmp_path, po, oracle_models):
    """`tksm sequence` (the module boundary) on files: --perfect vs the reference CLI golden; -o vs the oracle with
    the CLI's seed; both outputs at once reproduce the reference's quirk (perfect file = badread sequence, quals K)."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    d = os.path.join(GOLDEN, "splice_corpus")
    out = tmp_path / "perfect.fastq"
    r = subprocess.run([exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "--perfect", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _normalize(out.read_text()) == open(os.path.join(d, "expected_perfect.fastq")).read()
    # small batches (--batch-bytes) exercise the chunked MDF reader; read ids continue across chunks
    bad, bad2, per2 = tmp_path / "bad.fq.gz", tmp_path / "bad2.fastq", tmp_path / "per2.fasta"
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    r = subprocess.run([exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "-o", str(bad),
                        "-s", "7", "--batch-bytes", "4096"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    import gzip
    got = gzip.open(bad, "rb").read()
    ref = po.get_reference_seqs([os.path.join(d, "ref.fa")])
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    s.set_identity(84.0, 99.0, 5.5)
    ident = po.Identities(84.0, 5.5, 99.0)
    s.close()
    want = []
    with open(os.path.join(d, "mols.mdf")) as f:
        for i, (mid, ivs) in enumerate(po.mdf_generator(f)):
            want.append(po.badread_record(True, 7, i, po.splice(ref, ivs), ident, oracle_models["em"], oracle_models["qm"], True, mid)[0])
    assert got == b"".join(want)
    r = subprocess.run([exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "-o", str(bad2),
                        "--perfect", str(per2), "-s", "7", "--skip-qual-compute"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    b2 = bad2.read_bytes().split(b"\n")
    p2 = per2.read_bytes().split(b"\n")
    assert b2[1::4][: len(p2[1::2])] == p2[1::2]                       # same (badread) sequences in the "perfect" file
    assert all(b"read_identity=100.00%" in h for h in p2[0::2] if h)
    # the streaming skeleton: one batch in flight gives the same bytes as several; a malformed line in a late batch
    # ends the run with exit code 1 and a message, whatever the other workers are doing
    one = tmp_path / "one.fastq"
    r = subprocess.run([exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "-o", str(one),
                        "-s", "7", "--batch-bytes", "4096", "--in-flight", "1"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and one.read_bytes() == got
    broken = tmp_path / "broken.mdf"
    text = open(os.path.join(d, "mols.mdf")).read()
    cut = text.rfind("\n+", 0, len(text) * 3 // 4)
    broken.write_text(text[:cut + 1] + "chr1\tnot-a-number\t10\t+\t\n" + text[cut + 1:])
    r = subprocess.run([exe, "sequence", "-i", str(broken), "-r", os.path.join(d, "ref.fa"), "-o", str(tmp_path / "x.fastq"),
                        "--batch-bytes", "4096"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 1 and "MDF line" in r.stderr


def test_skewed_lengths_every_scheduling_variant_gives_the_same_bytes(capfd):
    """40 000 reads with lognormal lengths (median 1 kb, to 16 kb): the default run -- predicted stragglers on their own stream from
    round 0 (several hundred of them here), the straggler kernel for the last reads, long q-score jobs straight to the full-width
    pass -- against the same batch with every read in the regular rounds only (no straggler kernel at all, lane-per-read loops and
    the 14-row + list alignment passes in every round): not a byte may differ."""
    from tksm_amd import synthetic
    from tksm_amd.sequence import Sequencer
    rs = np.random.RandomState(23)
    lens = [4_000_000] * 4
    contigs = [rs.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() for n in lens]
    m = synthetic.make_molecules(rs, lens, 40000, 1000, 200, lognormal_sigma=0.6)

    def run(env):
        with pytest.MonkeyPatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            mp.setenv("TKSMSEQ_VERBOSE", "1")                                # (read at run time; the other knobs when the context is created)
            sq = Sequencer(0)
            for c, seq in enumerate(contigs):
                sq.add_contig(f"chr{c + 1}", seq)
            sq.set_identity(84.0, 99.0, 5.5)
            sq.load_error_model(ERR_MODEL)
            sq.load_qscore_model(QS_MODEL)
            b = sq.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
            capfd.readouterr()
            rec, off = sq.run(b, seed=9).download()
            err = capfd.readouterr().err
            sq.close()
        return rec, err
    import re
    rec_default, err_default = run({})
    got = re.search(r"predicted stragglers on their own stream: (\d+)", err_default)
    assert got and int(got.group(1)) >= 100, err_default[-500:]
    rec_plain, err_plain = run({"TKSMSEQ_TAIL_WAVE": "0", "TKSMSEQ_EARLY_TAIL": "0", "TKSMSEQ_WAVE_LOOP": "0", "TKSMSEQ_SMALL_ALN": "0"})
    assert re.search(r"predicted stragglers on their own stream: 0\b", err_plain)
    assert rec_default == rec_plain
    rec_late, _ = run({"TKSMSEQ_EARLY_TAIL": "0"})                       # the straggler kernel for the last reads only
    assert rec_late == rec_default


def test_full_size_properties_and_shard_invariance():
    """size-independent properties at a bench-like size (131072 reads of ~1 kb): record structure, determinism, and
    rank-count invariance -- two round-robin shards interleaved on the device equal the single-GPU stream."""
    import torch
    from tksm_amd import synthetic
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    rs = np.random.RandomState(11)
    lens = [4_000_000] * 4
    contigs = [rs.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() for n in lens]

    def setup(sq):
        for c, seq in enumerate(contigs):
            sq.add_contig(f"chr{c + 1}", seq)
        sq.set_identity(84.0, 99.0, 5.5)
        sq.load_error_model(ERR_MODEL)
        sq.load_qscore_model(QS_MODEL)
    setup(s)
    n = 131072
    m = synthetic.make_molecules(rs, lens, n, 1000, 200)
    b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
    rec1, off1 = s.run(b, seed=5).download()
    res2 = s.run(b, seed=5)
    rec2, _ = res2.download()
    assert rec1 == rec2                                                   # deterministic
    # canary on what parity tests cannot see (tksmseq_run_diagnostics): a lane-per-alignment job that k_alnf gives up falls back
    # to the exact wave-wide kernel with the same bytes -- a miscompiled queue (the spill episode) shows only here
    dg = s.run_diagnostics()
    assert dg["fallbacks"] == 0 and dg["fallback_reasons"] == 0 and dg["exact_kernel_reads"] == 0 and dg["band_exits"] == 0, dg
    assert 15 <= dg["rounds"] <= 60 and dg["jobs_all_rounds"] > 5 * n, dg
    # (a batch of this size aligns at full width in every round -- TKSMSEQ_SMALL_ALN = 131072; the 14-row pass + redo list of the
    # bench-size batches runs here with the threshold at 0: same bytes, 1 - 3 % of its jobs redone, none given up)
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("TKSMSEQ_SMALL_ALN", "0")
        s4 = Sequencer(0)
    setup(s4)
    b4 = s4.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
    rec4, _ = s4.run(b4, seed=5).download()
    dg4 = s4.run_diagnostics()
    s4.close()
    assert rec4 == rec1
    assert dg4["fallbacks"] == 0 and dg4["fallback_reasons"] == 0 and dg4["exact_kernel_reads"] == 0, dg4
    assert dg4["jobs_14_row_rounds"] > 5 * n and 0.01 <= dg4["jobs_redone_full_width"] / dg4["jobs_14_row_rounds"] <= 0.03, dg4
    # slices of the record stream (tksmseq_result_download_range: what the CLI streams through its page-locked pieces)
    for lo, nb in ((0, 1), (12345, 70001), (len(rec1) - 5, 5), (len(rec1), 0)):
        assert res2.download_range(lo, nb) == rec1[lo:lo + nb]
    with pytest.raises(Exception):
        res2.download_range(len(rec1) - 3, 4)
    # the same batch with every alignment at full width, the error loop one lane per read in every round and no tail cut: the
    # 16-row predecessor codes + follow-up passes, k_loopw in the late rounds, the launch grouping and the tail hand-over to the
    # wave-wide kernel do not change a byte
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("TKSMSEQ_SMALL_ALN", str(1 << 30)); mp.setenv("TKSMSEQ_SMALL_ROUND", str(1 << 30)); mp.setenv("TKSMSEQ_TAIL_CUT", "0")
        mp.setenv("TKSMSEQ_WAVE_LOOP", "0"); mp.setenv("TKSMSEQ_TAIL_WAVE", "0")
        s3 = Sequencer(0)
    setup(s3)
    b3 = s3.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
    rec3, _ = s3.run(b3, seed=5).download()
    assert rec3 == rec1
    s3.close()
    assert len(off1) == n + 1 and int(off1[-1]) == len(rec1)
    lines = rec1.split(b"\n")
    assert len(lines) == 4 * n + 1
    hdr, seq, plus, qual = lines[0:-1:4], lines[1::4], lines[2::4], lines[3::4]
    assert all(p == b"+" for p in plus[:: 97])
    for k in range(0, n, 257):
        f = dict(x.split(b"=") for x in hdr[k].split(b" ")[1:])
        assert int(f[b"length"]) == len(seq[k]) == len(qual[k])
        assert int(f[b"error_free_length"]) == int(m["raw_len"][k])
        assert f[b"molecule_id"] == f"mol_{k}".encode() and 50.0 < float(f[b"read_identity"][:-1]) <= 100.0
        assert set(seq[k]) <= set(b"ACGT") and min(qual[k]) >= 33
    total_in = int(m["raw_len"].sum())
    total_out = sum(len(x) for x in seq)
    assert 0.93 < total_out / total_in < 0.97                             # nanopore2020: reads ~0.95 x the molecule
    # two shards (global read g -> rank g mod 2), interleaved back on the device
    parts = []
    for rank in range(2):
        sel = np.arange(rank, n, 2)
        # mods of the selected reads must stay contiguous per interval: rebuild the shard from the text form
        text = synthetic.mdf_text({**m, "reads": m["reads"][sel], "ids": m["ids"][sel]}, [f"chr{c + 1}" for c in range(4)])
        sb = s.batch_from_mdf(text)
        res = s.run(sb, seed=5, first_read_index=rank, stride=2)
        rec, off = res.download()
        parts.append((torch.tensor(list(rec), dtype=torch.uint8, device="cuda") if len(rec) < 1 else
                      torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(sel)))
    dst = torch.empty(len(rec1) + 64, dtype=torch.uint8, device="cuda")
    nbytes = s.interleave_records([p[0].data_ptr() for p in parts], [p[1].data_ptr() for p in parts], [p[2] for p in parts],
                                  dst.data_ptr(), dst.numel())
    assert nbytes == len(rec1) and bytes(dst[:nbytes].cpu().numpy().tobytes()) == rec1
    s.close()


@pytest.mark.parametrize("err,qs", [("random", "random"), ("random", "ideal"), ("nanopore2020", "ideal"), ("nanopore2020", "random")])
def test_builtin_models_bit_exact_vs_oracle(po, oracle_models, err, qs):
    """the special model names of the CLI (py/tksm_badread.py:80-83 'random' error model with k = 1; :487-544 'random' and
    'ideal' q-score models) through the same kernels"""
    s, ref, rs = _random_genome_seqr(n_contigs=2, size=60_000, seed=9)
    s.set_identity(88.0, 97.0, 4.0)
    s.load_error_model(ERR_MODEL if err == "nanopore2020" else err)
    s.load_qscore_model(qs)
    em = oracle_models["em"] if err == "nanopore2020" else po.ErrorModel(err)
    qm = po.QScoreModel(qs)
    mols = _make_molecules(rs, ref, 40, 500)
    text = "".join(f"+{m}\t1\t\n" + "".join(f"{c}\t{a}\t{b}\t{st}\t{md}\n" for c, a, b, st, md in ivs) for m, ivs in mols)
    recs = s.run(s.batch_from_mdf(text), target="badread", fastq=True, compute_qual=True, seed=77).records()
    ident = po.Identities(88.0, 4.0, 97.0)
    for i, (mid, ivs) in enumerate(mols):
        want, st = po.badread_record(True, 77, i, po.splice(ref, ivs), ident, em, qm, True, mid)
        assert recs[i] == want, (err, qs, i)
    s.close()


def test_edge_cases_empty_short_constant_identity_and_limits(po, oracle_models):
    """empty batch, empty / one-base / shorter-than-k molecules, depth 0 and depth 3, a constant identity
    (mean == max, py/tksm_badread.py:711-715), FASTA output (no q-scores), and the documented size limit"""
    from tksm_amd.sequence import Sequencer, TksmSeqError
    s, ref, rs = _random_genome_seqr(n_contigs=1, size=30_000, seed=2)
    s.set_identity(90.0, 90.0, 3.0)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    b0 = s.batch_from_mdf("")
    r0 = s.run(b0, target="badread")
    assert r0.n_reads == 0 and r0.records_bytes == 0
    mdf = ("+empty\t1\t\nchr1\t5\t5\t+\t\n" "+one\t1\t\nchr1\t7\t8\t-\t\n" "+short\t1\t\nchr1\t100\t104\t+\t2A\n"
           "+gone\t0\t\nchr1\t0\t50\t+\t\n" "+thrice\t3\tx=1;\nchr1\t200\t420\t-\t\n" "+lit\t1\t\nACGTN\t0\t5\t+\t\n")
    ident = po.Identities(90.0, 3.0, 90.0)
    mols = list(po.mdf_generator(mdf.splitlines(keepends=True)))
    assert [m[0] for m in mols] == ["empty", "one", "short", "thrice", "thrice", "thrice", "lit"]
    for fastq in (True, False):
        recs = s.run(s.batch_from_mdf(mdf), target="badread", fastq=fastq, compute_qual=True, seed=3).records()
        assert len(recs) == len(mols)
        for i, (mid, ivs) in enumerate(mols):
            want, _ = po.badread_record(fastq, 3, i, po.splice(ref, ivs), ident, oracle_models["em"], oracle_models["qm"], fastq, mid)
            assert recs[i] == want, (fastq, i, mid)
        per = s.run(s.batch_from_mdf(mdf), target="perfect", fastq=fastq, seed=3).records()
        for i, (mid, ivs) in enumerate(mols):
            assert per[i] == po.perfect_record(fastq, 3, i, po.splice(ref, ivs), mid)
    # malformed input is refused the way the reference crashes on it (ValueError / IndexError -> exit 1)
    for bad in ("chr1\t0\t5\t+\t\n", "+m\t1\t\nchr1\t0\t5\t+\n", "+m\t1\t\nchr1\t0\t5\t+\t9A\n", "+m\tx\t\n"):
        with pytest.raises(TksmSeqError):
            s.run(s.batch_from_mdf(bad), target="perfect")
    # documented limit: a molecule beyond the LDS-resident working set
    s_big = rs.choice(np.frombuffer(b"ACGT", np.uint8), 80_000).tobytes().decode()
    s.add_contig("big", s_big.encode())
    # Badread mode has no practical length limit any more (100 kb; the reference has none): a 60 kb ACGT molecule takes the fast
    # pipeline, a 30 kb one with N's (real genomes have N runs) the exact wave-wide kernel with its working set in HBM -- next to
    # ordinary molecules in the same batch, all bit-exact with the oracle
    em, qm = oracle_models["em"], oracle_models["qm"]
    mdf = ("+long60\t1\t\nbig\t100\t60100\t+\t\n" "+n30\t1\t\nbig\t5000\t35000\t-\t200N,201N,202N,15000N,29999N\n"
           "+short\t1\t\nbig\t7\t907\t+\t\n" "+n_short\t1\t\nbig\t40000\t41000\t+\t500N\n")
    ivs = [("long60", [("big", 100, 60100, "+", "")]), ("n30", [("big", 5000, 35000, "-", "200N,201N,202N,15000N,29999N")]),
           ("short", [("big", 7, 907, "+", "")]), ("n_short", [("big", 40000, 41000, "+", "500N")])]
    got = s.run(s.batch_from_mdf(mdf), target="badread", seed=SEED).records()
    for i, (mid, iv) in enumerate(ivs):
        assert got[i] == po.badread_record(True, SEED, i, po.splice({"big": s_big}, iv), ident, em, qm, True, mid)[0], mid
    s.add_contig("huge", "".join(np.random.RandomState(6).choice(list("ACGT"), 120_000)))
    with pytest.raises(TksmSeqError) as e:
        s.run(s.batch_from_mdf("+huge\t1\t\nhuge\t0\t110000\t+\t\n"), target="badread")
    assert e.value.code == 6 and "100 000" in str(e.value)
    # ... which the --perfect path does not have
    big = s.run(s.batch_from_mdf("+huge\t1\t\nbig\t5\t70005\t-\t17C\n"), target="perfect", seed=3).records()[0]
    assert big == po.perfect_record(True, 3, 0, po.splice({"big": s_big}, [("big", 5, 70005, "-", "17C")]), "huge")
    s.close()


def test_output_slot_overflow_triggers_rerun(tmp_path, po):
    """an insertion-heavy error model outgrows the default 1.5x output slot: the library reruns with the worst-case
    slot and still matches the oracle"""
    import gzip
    path = tmp_path / "ins.error.gz"
    rs = np.random.RandomState(0)
    with gzip.open(path, "wt") as f:
        for idx in range(4 ** 3):
            kmer = "".join("ACGT"[(idx >> (2 * (2 - j))) & 3] for j in range(3))
            alt = kmer[0] + kmer[1] + "ACGT"[rs.randint(4)] * 4 + kmer[2]      # 4-base insertion in the middle
            f.write(f"{kmer},0.300000;{alt},0.650000;\n")
    from tksm_amd.sequence import Sequencer
    s, ref, rs2 = _random_genome_seqr(n_contigs=1, size=20_000, seed=4)
    s.set_identity(55.0, 60.0, 1.0)
    s.load_error_model(str(path))
    s.load_qscore_model("random")
    em, qm = po.ErrorModel(str(path)), po.QScoreModel("random")
    assert em.k == 3
    mols = [(f"m{i}", [("chr1", 100 * i, 100 * i + 300, "+", "")]) for i in range(12)]
    text = "".join(f"+{m}\t1\t\n" + "".join(f"{c}\t{a}\t{b}\t{st}\t{md}\n" for c, a, b, st, md in ivs) for m, ivs in mols)
    recs = s.run(s.batch_from_mdf(text), target="badread", fastq=True, compute_qual=True, seed=8).records()
    ident = po.Identities(55.0, 1.0, 60.0)
    grew = 0
    for i, (mid, ivs) in enumerate(mols):
        want, st = po.badread_record(True, 8, i, po.splice(ref, ivs), ident, em, qm, True, mid)
        grew += st.new_len > 1.5 * st.frag_len + 64
        assert recs[i] == want, i
    assert grew > 0, "the model did not outgrow the default slot: the test does not exercise the rerun"
    s.close()


def test_clone_shares_reference_and_models_across_threads(oracle_models):
    """tksmseq_clone: contexts that share one packed reference and one set of model tables, driven from their own host
    threads and streams at the same time, produce the records of the source context run alone."""
    import threading
    import torch
    s, ref, rs = _random_genome_seqr()
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    mols = _make_molecules(rs, ref, 300, 600)
    text = "".join(f"+{m}\t1\t\n" + "".join(f"{c}\t{a}\t{b}\t{st}\t{md}\n" for c, a, b, st, md in ivs) for m, ivs in mols)
    want = s.run(s.batch_from_mdf(text), seed=SEED, first_read_index=7).download()[0]
    streams = [torch.cuda.Stream() for _ in range(3)]
    clones = [s.clone(stream=st.cuda_stream) for st in streams]
    got = [None] * len(clones)

    def work(i):
        torch.cuda.set_device(0)
        c = clones[i]
        b = c.batch_from_mdf(text)
        for _ in range(3):
            got[i] = c.run(b, seed=SEED, first_read_index=7).download()[0]
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(clones))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(g == want for g in got)
    # a model loaded into a clone is private to it
    clones[0].load_error_model("random")
    assert s.run(s.batch_from_mdf(text), seed=SEED, first_read_index=7).download()[0] == want
    assert clones[0].run(clones[0].batch_from_mdf(text), seed=SEED, first_read_index=7).download()[0] != want
    for c in clones:
        c.close()
    s.close()


@pytest.mark.parametrize("kind", ["bulk", "scrna", "long"])
def test_perfect_direct_kernel_equals_wave_wide_kernel(monkeypatch, kind):
    """--perfect: the direct kernel (packed reference -> records) and the wave-wide kernel (splice into LDS, emit) write
    the same bytes for 65 536 synthetic molecules (both strands, substitutions, literals, N / IUPAC / lower-case blocks).
    "long": skewed lengths up to 16 kb -- records beyond the first launch's LDS images go to the second launch."""
    from tksm_amd import synthetic
    from tksm_amd.sequence import Sequencer
    rs = np.random.RandomState(3)
    lens = [1_000_000] * 3
    contigs = []
    for n in lens:
        seq = bytearray(rs.choice(np.frombuffer(b"ACGTacgt", np.uint8), n).tobytes())
        for _ in range(20):
            p = int(rs.randint(0, n - 500))
            seq[p:p + int(rs.randint(1, 400))] = b"N" * 400 if rs.rand() < 0.5 else b"RYKM" * 100
        contigs.append(bytes(seq[:n]))
    if kind == "long":
        m = synthetic.make_molecules(rs, lens, 8192, 2500, 500, lognormal_sigma=0.8)
        assert (m["raw_len"] > 4000).sum() > 1000 and m["raw_len"].max() > 12000
    else:
        m = synthetic.make_molecules(rs, lens, 65536, 700, 200, kind=kind)
    out = {}
    for mode in ("direct", "wave"):
        monkeypatch.setenv("TKSMSEQ_FORCE_SLOW", "1" if mode == "wave" else "0")
        s = Sequencer(0)
        for c, seq in enumerate(contigs):
            s.add_contig(f"chr{c + 1}", seq)
        b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
        for fastq in (True, False):
            out[mode, fastq] = s.run(b, target="perfect", fastq=fastq, seed=9, first_read_index=5, stride=2).download()
        s.close()
    for fastq in (True, False):
        assert out["direct", fastq][0] == out["wave", fastq][0]
        assert (out["direct", fastq][1] == out["wave", fastq][1]).all()


@pytest.mark.parametrize("path", ["fast", "slow"])
def test_band_failure_falls_back_to_the_unbanded_alignment(oracle_models, po, monkeypatch, path):
    """three reads of a 262 144-read scRNA-like run (polyA ~ N(40, 20); tools/dump_unbanded_reads.py) in which the
    nanopore2020 model deletes so much of the polyA tail that the last window of the guided band misses the end cell.
    The specification (oracle: full_align) prescribes the unbanded alignment for such a window; the GPU path gives the
    oracle's records, band failures included.  Replayed as literal molecules at their original read indices."""
    import json
    monkeypatch.setenv("TKSMSEQ_FORCE_SLOW", "1" if path == "slow" else "0")
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    s.add_contig("chr1", b"ACGT" * 64)
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    ident = po.Identities(84.0, 5.5, 99.0)
    em, qm = oracle_models["em"], oracle_models["qm"]
    for x in json.load(open(os.path.join(GOLDEN, "unbanded_reads.json"))):
        seq = x["sequence"]
        text = f"+mol_{x['index']}\t1\t\n{seq}\t0\t{len(seq)}\t+\t\n"
        res = s.run(s.batch_from_mdf(text), target="badread", fastq=True, compute_qual=True, seed=42, first_read_index=x["index"],
                    collect_stats=True)
        want, st = po.badread_record(True, 42, x["index"], seq.encode(), ident, em, qm, True, f"mol_{x['index']}")
        assert st.band_fail > 0
        assert res.records()[0] == want
        assert res.stats()[0][0, 7] & 16                    # the run reports that it took the unbanded path
    s.close()


TAIL_MODEL = os.path.join(GOLDEN, "tail_model_synth.json")


def test_reference_written_tail_model_bit_exact_vs_oracle(oracle_models, po):
    """tests/golden/tail_model_reference.json was built and written by the reference's own KDE_noise_generator.from_data /
    .save (py/tksm_badread.py:888-901, :935-942; tests/golden/make_kde_golden.py).  Loaded as it is through
    tksmseq_load_tail_model: whole records equal the oracle's with the same file (incl. fragments past the last label)."""
    path = os.path.join(GOLDEN, "tail_model_reference.json")
    s, ref, rs = _random_genome_seqr()
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    s.load_tail_model(path)
    tm = po.TailModel(path)
    mols = _make_molecules(rs, ref, 160, 900) + _make_molecules(rs, ref, 16, 2300)
    text = "".join(f"+{m}\t1\t\n" + "".join(f"{c}\t{a}\t{b}\t{st}\t{md}\n" for c, a, b, st, md in ivs) for m, ivs in mols)
    batch = s.batch_from_mdf(text)
    ident = po.Identities(84.0, 5.5, 99.0)
    em, qm = oracle_models["em"], oracle_models["qm"]
    recs = s.run(batch, target="badread", fastq=True, compute_qual=True, seed=SEED, first_read_index=9).records()
    n_tail = 0
    for i, (mid, ivs) in enumerate(mols):
        raw = po.splice(ref, ivs)
        want, st = po.badread_record(True, SEED, 9 + i, raw, ident, em, qm, True, mid, tail_model=tm)
        assert recs[i] == want, (i, mid)
        n_tail += st.frag_len != len(raw) + 2 * em.k
    assert n_tail > 30                                         # ratio 0.40 of 176 reads
    s.close()


@pytest.mark.parametrize("path", ["fast", "slow"])
def test_tail_noise_bit_exact_vs_oracle(oracle_models, po, monkeypatch, path):
    """tail-noise model (KDE_noise_generator, py/tksm_badread.py:886-962; appended at :335-339): whole records against
    the oracle with the same model; error_free_length excludes the tail; switching the model off (or the seed) on the
    same batch rebuilds the batch's lengths and order; symbols outside ACGT in `bases` take the wave-wide kernel."""
    monkeypatch.setenv("TKSMSEQ_FORCE_SLOW", "1" if path == "slow" else "0")
    if path == "fast":
        monkeypatch.setenv("TKSMSEQ_SMALL_ALN", "0"); monkeypatch.setenv("TKSMSEQ_SMALL_ROUND", "0"); monkeypatch.setenv("TKSMSEQ_WAVE_LOOP", "0"); monkeypatch.setenv("TKSMSEQ_TAIL_WAVE", "0")
    import json
    s, ref, rs = _random_genome_seqr()
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    s.load_tail_model(TAIL_MODEL)
    tm = po.TailModel(TAIL_MODEL)
    mols = _make_molecules(rs, ref, 96, 700)
    text = "".join(f"+{m}\t1\t\n" + "".join(f"{c}\t{a}\t{b}\t{st}\t{md}\n" for c, a, b, st, md in ivs) for m, ivs in mols)
    batch = s.batch_from_mdf(text)
    ident = po.Identities(84.0, 5.5, 99.0)
    em, qm = oracle_models["em"], oracle_models["qm"]

    def check(seed, tail):
        recs = s.run(batch, target="badread", fastq=True, compute_qual=True, seed=seed, first_read_index=500, stride=2).records()
        n_tail = 0
        for i, (mid, ivs) in enumerate(mols):
            raw = po.splice(ref, ivs)
            want, st = po.badread_record(True, seed, 500 + 2 * i, raw, ident, em, qm, True, mid, tail_model=tail)
            assert st.band_fail == 0
            assert recs[i] == want, (i, mid)
            n_tail += st.frag_len != len(raw) + 2 * em.k
            assert f" error_free_length={len(raw)} ".encode() in recs[i]
        return n_tail

    assert check(SEED, tm) > 15                                # ratio 0.35 of 96 reads
    assert check(SEED + 1, tm) > 15                            # other seed, other tails: lengths rebuilt
    s.load_tail_model("no_noise")
    assert check(SEED, None) == 0
    # every read with a tail, and a state that emits 'N' (not a base the bit-parallel path can hold)
    dc = json.load(open(TAIL_MODEL))
    s.set_tail_model(dc["lx"], dc["ly"], dc["grid"], dc["trans"], 1.0, "AGTN")
    dc["ratio"] = 1.0; dc["bases"] = list("AGTN")
    assert check(SEED, po.TailModel(dc)) > 60
    # perfect output never carries a tail
    recs = s.run(batch, target="perfect", fastq=True, seed=SEED).records()
    for i, (mid, ivs) in enumerate(mols[:8]):
        assert recs[i] == po.perfect_record(True, SEED, i, po.splice(ref, ivs), mid)
    s.close()


def test_cli_tail_model_flag(tmp_path, po, oracle_models):
    """--badread-tail-model through the module boundary (py/sequence.py:109-113, :343-345), and its error path."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    d = os.path.join(GOLDEN, "splice_corpus")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    out = tmp_path / "tail.fastq"
    r = subprocess.run([exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "-o", str(out), "-s", "9",
                        "--badread-tail-model", TAIL_MODEL, "--batch-bytes", "4096"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    ref = po.get_reference_seqs([os.path.join(d, "ref.fa")])
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    s.set_identity(84.0, 99.0, 5.5)
    ident = po.Identities(84.0, 5.5, 99.0)
    s.close()
    tm = po.TailModel(TAIL_MODEL)
    want = []
    with open(os.path.join(d, "mols.mdf")) as f:
        for i, (mid, ivs) in enumerate(po.mdf_generator(f)):
            want.append(po.badread_record(True, 9, i, po.splice(ref, ivs), ident, oracle_models["em"], oracle_models["qm"], True, mid,
                                          tail_model=tm)[0])
    assert out.read_bytes() == b"".join(want)
    bad = tmp_path / "bad.json"
    bad.write_text('{"lx": [1, 2], "ly": [5], "grid": [[0, 0]], "trans": [[1,1,1,1]], "ratio": 0.5, "bases": ["A","G","T","C"], "begin": []}')
    r = subprocess.run([exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "-o", str(tmp_path / "x.fastq"),
                        "--badread-tail-model", str(bad)], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "tail model" in r.stderr


@pytest.mark.gpu
def test_cli_devices_list_threads_and_error_path(tmp_path, po, oracle_models):
    """--devices 0,0 (two device groups, here on the same GPU) with -t 4 parse threads writes the bytes of --devices 0 -t 1;
    an unwritable output with several batches in flight ends with exit code 1 instead of hanging (workers waiting for
    the writer are woken); --verbosity / --log-file are honoured."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    d = os.path.join(GOLDEN, "splice_corpus")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    base = [exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "-s", "11", "--batch-bytes", "4096"]
    one, two = tmp_path / "one.fastq", tmp_path / "two.fastq"
    r = subprocess.run(base + ["-o", str(one), "--devices", "0", "--in-flight", "1", "-t", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    logf = tmp_path / "run.log"
    r = subprocess.run(base + ["-o", str(two), "--devices", "0,0", "--in-flight", "2", "-t", "4", "--verbosity", "DEBUG", "--log-file", str(logf)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert one.read_bytes() == two.read_bytes() and len(one.read_bytes()) > 10000
    log = logf.read_text()
    assert "2 device group(s) x 2 contexts" in log and "Sequencing:" in log and "[sequence DBG]" in log
    r = subprocess.run(base + ["-o", str(tmp_path / "q.fastq"), "--verbosity", "OFF"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "Sequencing:" not in r.stderr
    # the default models are the nanopore2020 pair (py/sequence.py:86-107): same bytes as naming them
    named = tmp_path / "named.fastq"
    r = subprocess.run(base + ["-o", str(named), "--devices", "0", "--badread-error-model", "nanopore2020", "--badread-qscore-model", "nanopore2020"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and named.read_bytes() == one.read_bytes()
    # the other shipped models by name
    for model in ("nanopore2018", "pacbio2016"):
        o2 = tmp_path / f"{model}.fastq"
        r = subprocess.run(base + ["-o", str(o2), "--badread-error-model", model, "--badread-qscore-model", model], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        assert o2.read_bytes().count(b"\n@") > 100 and o2.read_bytes() != one.read_bytes()
    # regular files take a batch's records in page-locked pieces (64 MB; here 4 KB, so every batch is many pieces): same bytes,
    # also with a gzip output next to a plain one (gzip members are written whole, the plain file in pieces)
    small = dict(env, TKSMSEQ_PIECE_BYTES="4096")
    pieces = tmp_path / "pieces.fastq"
    r = subprocess.run(base[:-2] + ["--batch-bytes", "65536", "-o", str(pieces), "--in-flight", "3"], capture_output=True, text=True, env=small, timeout=300)
    assert r.returncode == 0, r.stderr
    assert pieces.read_bytes() == one.read_bytes()
    import gzip
    gz_b, plain_p = tmp_path / "both.fastq.gz", tmp_path / "both_perfect.fasta"
    r = subprocess.run(base + ["-o", str(gz_b), "--perfect", str(plain_p), "--in-flight", "3"], capture_output=True, text=True, env=small, timeout=300)
    assert r.returncode == 0, r.stderr
    assert gzip.decompress(gz_b.read_bytes()) == one.read_bytes()
    seqs = one.read_bytes().split(b"\n")[1::4]
    assert plain_p.read_bytes().split(b"\n")[1::2] == seqs          # the -o + --perfect quirk: the badread sequence (py/sequence.py:317-319)
    # a pipe (not seekable): the ordered writer thread instead of the workers' positional writes -- same bytes
    import threading
    fifo = tmp_path / "pipe.fastq"                     # (the extension decides the format)
    os.mkfifo(fifo)
    got = []
    rd = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()), daemon=True)
    rd.start()
    r = subprocess.run(base + ["-o", str(fifo), "--devices", "0", "--in-flight", "3"], capture_output=True, text=True, env=env, timeout=300)
    rd.join(timeout=60)
    assert r.returncode == 0, r.stderr
    assert got and got[0] == one.read_bytes()
    # error path: /dev/full fails on write; several small batches, three in flight
    r = subprocess.run(base + ["-o", "/dev/full", "--skip-qual-compute", "--in-flight", "3", "--batch-bytes", "2048"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 1 and "write failed" in r.stderr
    r = subprocess.run(base + ["-o", str(tmp_path / "x.fastq"), "--devices", "99"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 1 and "device index out of range" in r.stderr
    # ... and with the input coming through a pipe whose writer neither writes on nor closes (Snakemake's pipes, Snakefile:283-305): the
    # reader polls the descriptor and gives up once a worker has failed, instead of sleeping in read() until the producer goes away
    import time
    fin = tmp_path / "in.mdf.pipe"
    os.mkfifo(fin)
    hold = threading.Event()
    def feed():
        with open(fin, "wb") as w:
            w.write(open(os.path.join(d, "mols.mdf"), "rb").read())
            w.flush()
            hold.wait(120)                                  # keeps the write end open
    fw = threading.Thread(target=feed, daemon=True)
    fw.start()
    t0 = time.time()
    r = subprocess.run([exe, "sequence", "-i", str(fin), "-r", os.path.join(d, "ref.fa"), "-s", "11", "--batch-bytes", "2048", "-o", "/dev/full", "--skip-qual-compute"],
                       capture_output=True, text=True, env=env, timeout=100)
    took = time.time() - t0
    hold.set()
    fw.join(timeout=30)
    assert r.returncode == 1 and "write failed" in r.stderr and took < 60, (r.returncode, took, r.stderr[-300:])


@pytest.mark.gpu
def test_cli_devices_list_over_distinct_gpus(tmp_path):
    """--devices over DISTINCT device indices where the box has more than one GPU (reference and models loaded per device, batches
    handed to whichever group is free, MDF order restored): the bytes of --devices 0.  The analogue of the reference's
    Pool(args.threads).imap_unordered over molecules (py/sequence.py:360-368)."""
    import subprocess
    import torch
    from conftest import ROOT
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one GPU on this box (--devices 0,0 is covered by test_cli_devices_list_threads_and_error_path)")
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    d = os.path.join(GOLDEN, "splice_corpus")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    base = [exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "-s", "11", "--batch-bytes", "4096"]
    one, many = tmp_path / "one.fastq", tmp_path / "many.fastq"
    r = subprocess.run(base + ["-o", str(one), "--devices", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    devs = ",".join(str(i) for i in range(min(n_dev, 4)))
    r = subprocess.run(base + ["-o", str(many), "--devices", devs, "-t", "4"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert one.read_bytes() == many.read_bytes() and len(one.read_bytes()) > 10000


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("pcr", 2500), ("scrna", 3000)])
def test_config3_and_config5_workloads_bit_exact_vs_oracle(seqr, po, oracle_models, monkeypatch, kind, n):
    """The bench generator's other workloads through the HIP path against the oracle, record for record (Badread with q-scores
    and --perfect): `pcr` = substitution-heavy molecules as 20 PCR cycles leave them (BASELINE config 5: ~5 substitutions per kb,
    both strands, interval ends), `scrna` = barcode + UMI + polyA literal segments (config 3).  Run as the large rounds of a large
    batch are (k_loop in every round; 16 stored rows per alignment: 5 % of the polyA-tailed jobs go on to the 64-row pass)."""
    monkeypatch.setenv("TKSMSEQ_SMALL_ALN", "0"); monkeypatch.setenv("TKSMSEQ_SMALL_ROUND", "0"); monkeypatch.setenv("TKSMSEQ_WAVE_LOOP", "0"); monkeypatch.setenv("TKSMSEQ_TAIL_WAVE", "0")
    from tksm_amd import synthetic
    rs = np.random.RandomState(17)
    lens = [200_000, 150_000]
    ref = {f"g{i}": rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode() for i, L in enumerate(lens)}
    from tksm_amd.sequence import Sequencer
    s = Sequencer(0)
    for k, v in ref.items():
        s.add_contig(k, v)
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    m = synthetic.make_molecules(rs, lens, n, 900, 250, kind=kind)
    if kind == "pcr":
        assert len(m["mods"]) > 3.5 * n                    # ~ 4.8 substitutions per molecule
    b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
    bad = s.run(b, target="badread", fastq=True, compute_qual=True, seed=6).records()
    per = s.run(b, target="perfect", fastq=True, seed=6).records()
    ident = po.Identities(84.0, 5.5, 99.0)
    text = synthetic.mdf_text(m, list(ref))
    for i, (mid, ivs) in enumerate(po.mdf_generator(text.splitlines(keepends=True))):
        raw = po.splice(ref, ivs)
        assert per[i] == po.perfect_record(True, 6, i, raw, mid), (kind, i)
        assert bad[i] == po.badread_record(True, 6, i, raw, ident, oracle_models["em"], oracle_models["qm"], True, mid)[0], (kind, i)
    b.free(); s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,sigma", [("bulk", 1_703_936, None), ("scrna", 1_703_936, None), ("pcr", 1_703_936, None), ("bulk", 1_048_576, 0.6)])
def test_bench_size_batch_every_record_by_digest(po, oracle_models, kind, n, sigma):
    """BASELINE's full sizes: one batch of 1 703 936 molecules (bench.py's default: every launch shape, buffer size and 32-bit
    offset of the bench; bulk = config 2, scRNA-like with barcode / UMI / polyA literals = config 3) and one of 1 048 576
    molecules with lognormal lengths (median 1 kb, clipped at 16 kb: ragged state rows, the long-read buckets), default settings.
    `pcr` = config 5's substitution-heavy molecules.  EVERY record of every batch is compared with the oracle's:
    tests/golden/bench_digests_<workload>.json holds a SHA-256 per block of 4 096 records, computed by the oracle in the build container
    (tests/golden/make_bench_digests.py, same generator calls); the device output is hashed the same way.  On top: every 499th read
    (first and last included) byte for byte against the oracle run here, the record structure of the whole stream, and the same batch
    and sample through the --perfect kernel."""
    from tksm_amd import synthetic
    from tksm_amd.sequence import Sequencer
    rs = np.random.RandomState(23)
    lens = [8_000_000] * 4
    names = [f"chr{c + 1}" for c in range(4)]
    ref = {nm: rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode() for nm, L in zip(names, lens)}
    s = Sequencer(0)
    for k, v in ref.items():
        s.add_contig(k, v)
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    m = synthetic.make_molecules(rs, lens, n, 1000, 200, kind=kind, lognormal_sigma=sigma)
    b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
    rec, off = s.run(b, target="badread", fastq=True, compute_qual=True, seed=9).download()
    assert len(off) == n + 1 and int(off[-1]) == len(rec) and rec.count(b"\n") == 4 * n
    ident = po.Identities(84.0, 5.5, 99.0)
    tag = "lognormal" if sigma else kind
    assert os.path.exists(os.path.join(GOLDEN, f"bench_digests_{tag}.json")), "python tests/golden/make_bench_digests.py " + tag
    if True:
        import hashlib
        import json
        gold = json.load(open(os.path.join(GOLDEN, f"bench_digests_{tag}.json")))
        blk = gold["block_records"]
        assert (gold["n"], gold["run_seed"], gold["generator_seed"]) == (n, 9, 23) and len(gold["sha256"]) == (n + blk - 1) // blk
        assert gold.get("generator", {"kind": kind, "lognormal_sigma": sigma}) == {"kind": kind, "lognormal_sigma": sigma}
        view = memoryview(rec)
        bad = [k for k, want in enumerate(gold["sha256"])
               if hashlib.sha256(view[int(off[k * blk]):int(off[min(n, (k + 1) * blk)])]).hexdigest() != want]
        # A block that differs is looked at record by record.  The one legitimate cause: the oracle's Beta quantile table here is
        # scipy's, the device's is the library's own (equal to 1e-10, test_identity_table_matches_scipy) -- a read whose stop rule
        # `1 - errors / len <= target` falls into that gap (~1e-7 per read) flips.  Such a read must equal the oracle run with the
        # library's table and have a different target under the two tables; anything else fails.
        if bad:
            import sys
            sys.path.insert(0, os.path.join(GOLDEN))
            import make_bench_digests as mk
            lib_ident = po.Identities(84.0, 5.5, 99.0, qtab=s.identity_tables()["qtab"])
            flagged = []
            for k in bad[:8]:
                lo, hi = k * blk, min(n, (k + 1) * blk)
                mine = mk.block_records(po, ref, m, lo, hi, ident, oracle_models["em"], oracle_models["qm"])
                lib = mk.block_records(po, ref, m, lo, hi, lib_ident, oracle_models["em"], oracle_models["qm"])
                for g in range(lo, hi):
                    got = rec[int(off[g]):int(off[g + 1])]
                    assert got == lib[g - lo], (kind, g, "differs from the oracle with the library's own quantile table")
                    if got != mine[g - lo]:
                        assert ident.get_identity(9, g) != lib_ident.get_identity(9, g), (kind, g)
                        flagged.append(g)
            assert len(bad) <= 2 and flagged, (tag, bad[:10])
            import warnings
            warnings.warn(f"{kind}: reads {flagged} differ from the golden digests through the 1e-10 gap between the two Beta quantile tables")
    sel = np.unique(np.concatenate([np.arange(0, n, 499), [n - 1]]))
    text = synthetic.mdf_text({**m, "reads": m["reads"][sel], "ids": m["ids"][sel]}, names)
    mols = list(po.mdf_generator(text.splitlines(keepends=True)))
    assert len(mols) == len(sel)
    raws = []
    for g, (mid, ivs) in zip(sel, mols):
        raw = po.splice(ref, ivs)
        raws.append(raw)
        want = po.badread_record(True, 9, int(g), raw, ident, oracle_models["em"], oracle_models["qm"], True, mid)[0]
        assert rec[int(off[g]):int(off[g + 1])] == want, int(g)
    # the same batch through the direct --perfect kernel
    del rec
    prec, poff = s.run(b, target="perfect", fastq=True, seed=9).download()
    assert len(poff) == n + 1 and int(poff[-1]) == len(prec)
    for g, (mid, ivs), raw in zip(sel, mols, raws):
        assert prec[int(poff[g]):int(poff[g + 1])] == po.perfect_record(True, 9, int(g), raw, mid), int(g)
    b.free(); s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,ident", [("nanopore2018", (85.0, 95.0, 5.0)), ("pacbio2016", (85.0, 95.0, 5.0))])
def test_other_shipped_models_bit_exact_vs_oracle(po, model, ident):
    """The two other shipped model pairs (py/tksm_models/badread: nanopore2018, pacbio2016; their own k-mer tables, alternative
    counts and q-score rows) through the same kernels, record for record vs the oracle loading the same files: bulk and scRNA-like
    molecules, with q-scores and --perfect; the model tables as the device holds them equal the oracle's."""
    from conftest import MODELS
    from tksm_amd import synthetic
    from tksm_amd.sequence import Sequencer
    em_path, qs_path = os.path.join(MODELS, f"{model}.error.gz"), os.path.join(MODELS, f"{model}.qscore.gz")
    em, qm = po.ErrorModel(em_path), po.QScoreModel(qs_path)
    rs = np.random.RandomState(41)
    lens = [300_000, 200_000]
    ref = {f"g{i}": rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode() for i, L in enumerate(lens)}
    s = Sequencer(0)
    for k, v in ref.items():
        s.add_contig(k, v)
    s.set_identity(*ident)                                 # (mean, max, stdev)
    s.load_error_model(em_path)
    s.load_qscore_model(qs_path)
    idt = po.Identities(ident[0], ident[2], ident[1])
    for kind, n in (("bulk", 1500), ("scrna", 700)):
        m = synthetic.make_molecules(rs, lens, n, 900, 300, kind=kind)
        b = s.batch_from_arrays(m["reads"], m["intervals"], m["mods"], m["literals"], m["literal_pool"], m["ids"], m["id_pool"])
        bad = s.run(b, target="badread", fastq=True, compute_qual=True, seed=8).records()
        text = synthetic.mdf_text(m, list(ref))
        for i, (mid, ivs) in enumerate(po.mdf_generator(text.splitlines(keepends=True))):
            raw = po.splice(ref, ivs)
            assert bad[i] == po.badread_record(True, 8, i, raw, idt, em, qm, True, mid)[0], (model, kind, i)
        b.free()
    s.close()


@pytest.mark.gpu
def test_cli_gzipped_reference_and_two_reference_files(tmp_path):
    """generate_fasta / get_reference_seqs (py/sequence.py:168-194): a .fa.gz reference, and the contigs split over two -r files,
    give the bytes of the plain single file"""
    import gzip
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    d = os.path.join(GOLDEN, "splice_corpus")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    ref = open(os.path.join(d, "ref.fa"), "rb").read()
    gz = tmp_path / "ref.fa.gz"
    gz.write_bytes(gzip.compress(ref))
    recs = ref.split(b">")[1:]
    assert len(recs) >= 2
    half = len(recs) // 2
    a, b = tmp_path / "a.fa", tmp_path / "b.fasta"
    a.write_bytes(b"".join(b">" + r for r in recs[:half])); b.write_bytes(b"".join(b">" + r for r in recs[half:]))
    outs = {}
    for name, refs in (("plain", [os.path.join(d, "ref.fa")]), ("gz", [str(gz)]), ("two", [str(a), str(b)])):
        o = tmp_path / f"{name}.fastq"
        r = subprocess.run([exe, "sequence", "-i", os.path.join(d, "mols.mdf"), "-r"] + refs + ["--perfect", str(o), "-s", "3"],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        outs[name] = o.read_bytes()
    assert len(outs["plain"]) > 1000 and outs["gz"] == outs["plain"] and outs["two"] == outs["plain"]


@pytest.mark.gpu
def test_cli_reads_mdf_from_a_pipe_and_chains_with_the_mdf_modules(tmp_path):
    """MDF arriving on /dev/stdin (how a pipeline hands molecules from one module to the next) gives the bytes of the file input;
    `tksm truncate` -> `tksm sequence` over a pipe equals the two steps over a file."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    d = os.path.join(GOLDEN, "splice_corpus")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    mdf, ref = os.path.join(d, "mols.mdf"), os.path.join(d, "ref.fa")
    f1, f2 = tmp_path / "file.fastq", tmp_path / "pipe.fastq"
    r = subprocess.run([exe, "sequence", "-i", mdf, "-r", ref, "-o", str(f1), "-s", "4", "--batch-bytes", "4096"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(mdf, "rb") as fin:
        r = subprocess.run([exe, "sequence", "-i", "/dev/stdin", "-r", ref, "-o", str(f2), "-s", "4", "--batch-bytes", "4096"], stdin=fin, capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f2.read_bytes() == f1.read_bytes() and len(f1.read_bytes()) > 10000
    # truncate | sequence
    t_file, o_file, o_pipe = tmp_path / "trc.mdf", tmp_path / "trc_file.fastq", tmp_path / "trc_pipe.fastq"
    r = subprocess.run([exe, "truncate", "-i", mdf, "-o", str(t_file), "--normal", "300,50", "-s", "5"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "sequence", "-i", str(t_file), "-r", ref, "--perfect", str(o_file), "-s", "4"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    p1 = subprocess.Popen([exe, "truncate", "-i", mdf, "-o", "/dev/stdout", "--normal", "300,50", "-s", "5"], stdout=subprocess.PIPE, env=env)
    p2 = subprocess.run([exe, "sequence", "-i", "/dev/stdin", "-r", ref, "--perfect", str(o_pipe), "-s", "4"], stdin=p1.stdout, capture_output=True, env=env, timeout=300)
    p1.stdout.close()
    assert p1.wait(timeout=120) == 0 and p2.returncode == 0, p2.stderr
    assert o_pipe.read_bytes() == o_file.read_bytes() and len(o_file.read_bytes()) > 1000


@pytest.mark.gpu
def test_cli_stream_equals_one_api_batch_at_scale(tmp_path):
    """`tksm sequence` streaming 300 000 bulk molecules in ~20 batches (three in flight, four parse threads, records written behind the
    workers at their offsets) writes the bytes ONE tksmseq_run of the same molecules gives through the API -- which the digest tests tie to
    the oracle record for record: read numbering across batches, the multi-threaded parser, the staging buffers and the positional writes
    at scale (the reference's `for read_dict in mapper(...)` loop over the whole MDF, py/sequence.py:360-370)."""
    import subprocess
    from conftest import ROOT
    from tksm_amd import synthetic
    from tksm_amd.sequence import Sequencer
    rs = np.random.RandomState(77)
    lens = [3_000_000] * 3
    names = ["chrA", "chrB", "chrC"]
    ref = {nm: rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes() for nm, L in zip(names, lens)}
    with open(tmp_path / "ref.fa", "wb") as f:
        for nm in names:
            f.write(b">" + nm.encode() + b"\n" + ref[nm] + b"\n")
    n = 300_000
    m = synthetic.make_molecules(rs, lens, n, 1000, 200)
    text = synthetic.mdf_text(m, names)
    (tmp_path / "mols.mdf").write_text(text)
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    out = tmp_path / "out.fastq"
    r = subprocess.run([exe, "sequence", "-i", str(tmp_path / "mols.mdf"), "-r", str(tmp_path / "ref.fa"), "-o", str(out), "-s", "31", "-t", "4",
                        "--batch-bytes", str(1 << 20), "--verbosity", "ERROR"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    s = Sequencer(0)
    for nm in names:
        s.add_contig(nm, ref[nm])
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    b = s.batch_from_mdf(text)
    rec, off = s.run(b, target="badread", fastq=True, compute_qual=True, seed=31).download()
    b.free(); s.close()
    got = out.read_bytes()
    assert len(off) == n + 1 and len(got) == len(rec) and got == rec
