"""PCR amplification and truncation (BASELINE config 5) and the MDF writer.

CPU part: the oracle (oracle/mdf_ops_oracle.py) against the reference's own unit-test vectors for truncate()
(test/truncate_test.cpp:12-55), hand-derived known answers for the C++ reader / writer pair (unroll naming src/mdf.h:97-105,
operator<< src/interval.h:898-905, comment round trip :809-830 / :880-890), and the counter-based PCR specification against
the reference's full-tree recursion (distribution of the written copies).
GPU part (-m gpu): the HIP kernels through the C-ABI against the oracle, text for text, and the device pipeline
PCR -> truncation -> Seq against the oracle's Seq on the transformed molecules."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def mo():
    import mdf_ops_oracle
    return mdf_ops_oracle


def _mol(segs, mid="m", depth=1):
    return dict(id=mid, depth=depth, meta={}, segments=[dict(chr=c, start=s, end=e, plus=p, errors=list(er)) for c, s, e, p, er in segs])


# ------------------------------------------------------------------------------------------------ CPU: reference vectors
def test_truncate_reference_unit_test_vectors(mo):
    """test/truncate_test.cpp:12-55: four 100-base segments truncated to 200, then 150, then 50 (min_val 100)"""
    md = _mol([("1", 0, 100, True, []), ("1", 100, 200, True, []), ("1", 200, 300, True, []), ("1", 300, 400, True, [])])
    mo.truncate(md, 200, 100)
    assert mo.mol_size(md) == 200 and [(s["start"], s["end"]) for s in md["segments"]] == [(0, 100), (100, 200)]
    mo.truncate(md, 150, 100)
    assert mo.mol_size(md) == 150 and [(s["start"], s["end"]) for s in md["segments"]] == [(0, 100), (100, 150)]
    mo.truncate(md, 50, 100)
    assert mo.mol_size(md) == 100 and [(s["start"], s["end"]) for s in md["segments"]] == [(0, 100)]
    # minus-strand cut segment keeps its END (the 5' part of the molecule), substitutions re-based and filtered
    md = _mol([("1", 0, 100, True, [(5, "A")]), ("2", 1000, 1200, False, [(10, "C"), (150, "G"), (199, "T"), (150, "A")])])
    mo.truncate(md, 160, 100)
    assert [(s["chr"], s["start"], s["end"]) for s in md["segments"]] == [("1", 0, 100), ("2", 1140, 1200)]
    assert md["segments"][1]["errors"] == [(10, "G"), (10, "A"), (59, "T")]
    assert md["meta"]["truncated"] == ["2:1000-1140"]


def test_mdf_reader_writer_known_answers(mo):
    """Hand-derived from the C++ (a22): stream_mdf(unroll=true) names the copies of a depth > 1 molecule id_0.. with depth 1
    (src/mdf.h:97-105); operator<< writes 5 fields per segment (src/interval.h:898-905); the comment goes through a key-sorted
    map, a key without '=' prints bare, several values join with ',' (:809-830, :880-890); a 4-field segment line (no
    substitution column) is accepted by the C++ reader (:82-85)."""
    ka = json.load(open(os.path.join(GOLDEN, "mdf_writer_known_answers.json")))
    for case in ka["cases"]:
        assert mo.write_mdf(mo.stream_mdf(case["in"], unroll=True)) == case["out"], case["name"]


@pytest.mark.parametrize("cycles,eff,er,per_template,seed,n", [
    (3, 0.56, 3e-4, 2.0, 5, 3000), (6, 0.7, 4e-4, 3.0, 9, 3000),
    # BASELINE config 5's depth: 20 cycles.  The full tree has (1 + efficiency)^cycles nodes per template, so the depth the
    # pipeline is configured for is enumerated with the Taq-setting2 preset (src/pcr.cpp:136-140: efficiency 0.36, 470 nodes per
    # template), and the Taq-setting1 efficiency (0.88) at 12 cycles (1 950 nodes per template) -- where the closed-form branch
    # pruning (pcr_tables: q[t] -> 1, pm / (1 - A[t])) is furthest from the shallow cases above
    (20, 0.36, 7.2e-5, 3.0, 13, 800), (12, 0.88, 2e-4, 4.0, 17, 300), (20, 0.36, 2.5e-3, 0.5, 19, 800)])
def test_pcr_specification_has_the_reference_distribution(mo, cycles, eff, er, per_template, seed, n):
    """The branch-pruned, counter-based PCR (what the kernels run) against the reference's full-tree recursion
    (src/pcr.cpp:40-89): number of written copies per template, generation of a written copy, substitutions per written copy,
    and the share of written pairs of one template that share a substitution (ancestry) -- chi-square / z tests."""
    from scipy.stats import chi2
    for _ in (0,):
        mols = [_mol([("1", 0, 700, True, []), ("2", 50, 350, False, [])], mid=f"t{u}") for u in range(n)]
        target = int(n * per_template)
        ref = mo.pcr_reference(mols, cycles, eff, er, target, np.random.RandomState(seed))
        got = mo.pcr_spec(mols, cycles, eff, er, target, seed)

        def stats(out):
            per = np.zeros(n, int); gen = np.zeros(cycles + 1, int); nm = np.zeros(8, int)
            by_t = {}
            for md in out:
                parts = md["id"].split(".")
                u = int(parts[0][1:])
                per[u] += 1
                gen[len(parts) - 1] += 1
                errs = [(si, p, b) for si, s in enumerate(md["segments"]) for p, b in s["errors"]]
                nm[min(7, len(errs))] += 1
                by_t.setdefault(u, []).append(set(errs))
            share = tot = 0
            for lst in by_t.values():
                for a in range(len(lst)):
                    for b in range(a + 1, len(lst)):
                        tot += 1; share += bool(lst[a] & lst[b])
            return np.bincount(np.minimum(per, 12), minlength=13), gen, nm, share, tot

        def chi2_p(a, b):
            a, b = np.asarray(a, float), np.asarray(b, float)
            keep = (a + b) >= 10
            a, b = np.append(a[keep], a[~keep].sum()), np.append(b[keep], b[~keep].sum())
            keep = (a + b) > 0
            a, b = a[keep], b[keep]
            k1, k2 = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
            return chi2.sf((((k1 * a - k2 * b) ** 2) / (a + b)).sum(), len(a) - 1)
        sr, sg = stats(ref), stats(got)
        assert abs(len(ref) - len(got)) < 5 * np.sqrt(len(ref) + len(got)), (len(ref), len(got), target)
        for k, name in enumerate(("copies per template", "generation", "substitutions per copy")):
            assert chi2_p(sr[k], sg[k]) > 1e-3, (cycles, name, sr[k], sg[k])
        pr, pg = sr[3] / max(1, sr[4]), sg[3] / max(1, sg[4])
        se = np.sqrt(pr * (1 - pr) / max(1, sr[4]) + pg * (1 - pg) / max(1, sg[4]) + 1e-12)
        assert abs(pr - pg) < 5 * se + 1e-3, ("shared substitutions among sibling copies", pr, pg)
        # ids: template id + "." + strictly increasing cycles
        for md in got[:200]:
            steps = [int(x) for x in md["id"].split(".")[1:]]
            assert steps == sorted(set(steps)) and steps and steps[-1] < cycles


def test_pcr_subsample_is_a_random_ordered_subset_like_shuffle_and_resize(mo):
    """More than 2 x target templates: the reference shuffles the molecules and keeps the first 2 x target (src/pcr.cpp:217-220) -- a
    uniformly random ORDERED subset; its copies come out in that order.  The specification takes the 2 x target templates with the
    smallest counter-based keys, in key order: the template ids of the output follow that order (not input order), every kept
    template is distinct, over seeds every template is kept with probability 2 x target / n and first with probability 1 / n."""
    text = _mdf(np.random.RandomState(12), 60, mods=False)
    mols = mo.stream_mdf(text, unroll=True)
    n = len(mols)
    target = 8
    first, kept = np.zeros(n), np.zeros(n)
    index_of = {m["id"]: i for i, m in enumerate(mols)}
    unsorted = 0
    for seed in range(400):
        out = mo.pcr_spec(mols, 2, 1.0, 0.0, target, seed)
        order = []
        for m in out:
            root = m["id"].split(".")[0]                                          # a copy's id: template id + "." + cycle (+ ...)
            if not order or order[-1] != index_of[root]:
                order.append(index_of[root])
        assert len(set(order)) == len(order) <= 2 * target                      # a template's copies are contiguous, each template once
        unsorted += order != sorted(order)
        if order:
            first[order[0]] += 1
        for u in order:
            kept[u] += 1
        # slices of the processing order, one after the other, are the whole
        parts = sum((mo.pcr_spec(mols, 2, 1.0, 0.0, target, seed, only=(lo, lo + 5)) for lo in range(0, 2 * target, 5)), [])
        assert [m["id"] for m in parts] == [m["id"] for m in out]
    assert unsorted > 300                                                         # key order, not input order
    assert kept.min() > 0 and kept.max() < 3.0 * kept.mean()                      # every template gets its turn
    assert first.max() < 400 * 6.0 / n                                            # no template is favoured as the first one


def test_reference_written_kde_model_in_the_oracle(mo):
    """tests/golden/kde_truncation_model.json was written by the reference's own py/truncate_kde.py (main -> printModelJson,
    :298-320) from synthetic mappings (tests/golden/make_kde_golden.py).  The oracle's loader (custom_distribution2D / end_mtx,
    src/truncate.cpp:148-203, :362-381) must take it as it is: 30 x 30 grid transposed row by row, labels x[1:] + y[1:], 100
    end-ratio bins with float labels; truncation through it keeps every molecule within its size, cuts both ends by the drawn
    ratio and records TR=."""
    parts = json.load(open(os.path.join(GOLDEN, "kde_truncation_model.json")))
    assert [p["name"] for p in parts] == ["KDE_mtx", "end_mtx"] and parts[0]["shape"] == [30, 30] and len(parts[1]["data"]) == 100
    model = mo.TruncationModel(parts)
    assert model.x == list(range(100, 3001, 100)) == model.y and len(model.rows) == 30 and model.sider is not None
    assert [len(r.pdf) for r in model.rows] == list(range(1, 31))           # row i: truncation lengths up to the molecule size
    assert abs(model.sider.cdf[-1] - 1.0) < 1e-12 and model.sider.bins[0] == 0.01 and model.sider.bins[-1] == 1.0
    rs = np.random.RandomState(5)
    cut5 = cut3 = 0
    for g in range(400):
        size = int(rs.randint(300, 2800))
        md = _mol([("1", 1000, 1000 + size // 2, True, [(7, "A")]), ("2", 50, 50 + size - size // 2, False, [])], mid=f"m{g}")
        out = mo.trc_spec(md, g, 31, model=model)
        tl, side = out["meta"]["TR"][0].split(",")
        assert 100 <= mo.mol_size(out) <= size
        assert abs(mo.mol_size(out) - max(100.0, size - float(tl))) <= 2.0 or mo.mol_size(out) == 100
        cut3 += out["segments"][-1]["start"] != 50 or len(out["segments"]) == 1
        cut5 += out["segments"][0]["start"] != 1000
    assert cut3 > 100 and cut5 > 100                                         # both ends get their share (end_mtx)


# ------------------------------------------------------------------------------------------------ GPU
REF = {"chr1": None, "chr2": None}


def _genome(rs):
    return {f"chr{i + 1}": rs.choice(np.frombuffer(b"ACGT", np.uint8), 60_000).tobytes().decode() for i in range(2)}


def _mdf(rs, n, mods=True):
    lines = []
    for i in range(n):
        depth = 1 if rs.rand() < 0.8 else int(rs.randint(2, 4))
        cm = ["", "tid=ENST7;CB=ACGT;", "z;a=1,2;"][int(rs.randint(0, 3))]
        lines.append(f"+mol{i}\t{depth}\t{cm}\n")
        for _ in range(int(rs.randint(1, 5))):
            ln = int(rs.randint(1, 600))
            st = int(rs.randint(0, 59_000))
            md = ",".join(f"{int(rs.randint(0, ln))}{'ACGT'[int(rs.randint(0, 4))]}" for _ in range(int(rs.randint(0, 3)))) if mods else ""
            lines.append(f"chr{int(rs.randint(1, 3))}\t{st}\t{st + ln}\t{'+-'[int(rs.randint(0, 2))]}\t{md}\n")
        if rs.rand() < 0.3:
            pa = "A" * int(rs.randint(1, 30))
            lines.append(f"{pa}\t0\t{len(pa)}\t+\t\n")
        if rs.rand() < 0.05:
            lines.append("chr1\t500\t500\t+\t\n")                       # an empty segment
    return "".join(lines)


@pytest.fixture(scope="module")
def gseq():
    from tksm_amd.sequence import Sequencer
    rs = np.random.RandomState(21)
    ref = _genome(rs)
    s = Sequencer(0)
    for k, v in ref.items():
        s.add_contig(k, v)
    yield s, ref
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cycles,eff,er,target,seed", [(3, 0.56, 3e-6, 700, 3), (5, 0.88, 2e-4, 1500, 4), (12, 0.5, 1e-3, 900, 5),
                                                       (4, 0.9, 5e-3, 100, 6), (0, 0.9, 1e-3, 100, 7)])
def test_pcr_kernels_match_the_oracle(gseq, mo, cycles, eff, er, target, seed):
    """tksmseq_pcr -> MDF text == oracle pcr_spec, text for text: ids (unroll suffix + copy path), order (depth first), segments,
    substitutions in (template, oldest copy first, position) order; (4, .., target 100): more than 2 x target templates."""
    s, _ = gseq
    text = _mdf(np.random.RandomState(seed), 400)
    b = s.batch_from_mdf(text)
    out = s.pcr(b, cycles, target, error_rate=er, efficiency=eff, seed=seed)
    got = s.to_mdf_text(out)
    want = mo.write_mdf(mo.pcr_spec(mo.stream_mdf(text, unroll=True), cycles, eff, er, target, seed))
    assert got == want
    assert cycles == 0 or abs(out.n_reads - target) < 6 * np.sqrt(target) + 10
    if cycles == 4:
        # the subsample's processing order (key order, as the reference's shuffle + resize) in template slices: one after the other = the whole
        parts = []
        for lo in range(0, 2 * target, 64):
            o2 = s.pcr(b, cycles, target, error_rate=er, efficiency=eff, seed=seed, templates=(lo, min(2 * target, lo + 64)))
            parts.append(s.to_mdf_text(o2))
            o2.free()
        assert "".join(parts) == got
        ids = [l.split("\t")[0].split(".")[0] for l in got.splitlines() if l.startswith("+")]
        roots = [x for i, x in enumerate(ids) if i == 0 or ids[i - 1] != x]
        assert len(set(roots)) == len(roots) and roots != sorted(roots, key=lambda x: (int(x[4:].split("_")[0]), x))      # shuffled, each template once
        assert list(s.pcr_template_counts(b, cycles, target, er, eff, seed=seed)[2 * target:]) == [0] * (b.n_reads - 2 * target)
    out.free(); b.free()


def _pcr_slice(job):
    import mdf_ops_oracle as mo2
    text, cycles, eff, er, target, seed, lo, hi = job
    return mo2.write_mdf(mo2.pcr_spec(mo2.stream_mdf(text, unroll=True), cycles, eff, er, target, seed, only=(lo, hi)))


@pytest.mark.gpu
def test_pcr_kernels_at_the_configured_depth(gseq, mo):
    """BASELINE config 5: 20 cycles with the Taq-setting1 preset (src/pcr.cpp:136-140), 30 000 templates -> ~300 000 molecules,
    tksmseq_pcr -> MDF text == oracle pcr_spec, text for text (the oracle's copies computed in slices of the templates, side by
    side; ids carry up to 20 copy steps)."""
    from multiprocessing import Pool
    s, _ = gseq
    er, eff = mo.PRESETS["Taq-setting1"]
    cycles, target, seed = 20, 300_000, 11
    text = _mdf(np.random.RandomState(seed), 24_000)
    b = s.batch_from_mdf(text)
    out = s.pcr(b, cycles, target, error_rate=er, efficiency=eff, seed=seed)
    got = s.to_mdf_text(out)
    n_t = len(mo.stream_mdf(text, unroll=True))
    assert n_t <= 2 * target
    procs = max(1, min(16, len(os.sched_getaffinity(0))))
    step = (n_t + 4 * procs - 1) // (4 * procs)
    with Pool(procs) as pool:
        parts = pool.map(_pcr_slice, [(text, cycles, eff, er, target, seed, lo, min(n_t, lo + step)) for lo in range(0, n_t, step)], chunksize=1)
    want = "".join(parts)
    assert abs(out.n_reads - target) < 6 * np.sqrt(target) + 10
    assert max(len(l.split("\t")[0].split(".")) - 1 for l in got.splitlines() if l.startswith("+")) >= 12      # deep copy paths occur
    assert got == want
    out.free(); b.free()


@pytest.mark.gpu
def test_truncation_kernels_match_the_oracle(gseq, mo, tmp_path):
    """tksmseq_truncate -> MDF text == the oracle's literal truncate() / flip_molecule() sequence (the kernels compute the kept
    window arithmetically): normal, lognormal, and a KDE model in the format py/truncate_kde.py:298-320 writes, with and
    without the end-ratio histogram, --always-end, --kde-models-length; comments truncated= / TR= included."""
    s, _ = gseq
    text = _mdf(np.random.RandomState(8), 1500)
    mols = mo.stream_mdf(text, unroll=True)
    b = s.batch_from_mdf(text)
    for kw in (dict(normal=(400.0, 150.0)), dict(lognormal=(5.8, 0.6)), dict(normal=(50.0, 10.0)), dict(normal=(5000.0, 1.0))):
        out = s.truncate(b, seed=17, first_molecule_index=1000, **kw)
        want = mo.write_mdf([mo.trc_spec(md, 1000 + g, 17, **kw) for g, md in enumerate(mols)])
        assert s.to_mdf_text(out) == want, kw
        out.free()
    # a KDE model: x = truncation lengths, y = molecule sizes, lower-triangular weights; end ratios in [0, 1]
    rs = np.random.RandomState(3)
    w, h = 12, 12
    xl = [int(v) for v in np.arange(1, w + 1) * 150]
    yl = [int(v) for v in np.arange(1, h + 1) * 150]
    data = (rs.rand(h, w) + 0.05).ravel().tolist()
    parts = [dict(name="KDE_mtx", shape=[w, h], data=data, labels=xl + yl),
             dict(name="end_mtx", shape=[20], data=[int(v) for v in rs.randint(0, 50, 20)], labels=[float(v) for v in np.arange(1, 21) / 20.0])]
    for with_end, always_end, ml in ((True, False, False), (True, True, True), (False, True, False)):
        path = tmp_path / f"model_{with_end}_{always_end}.json"
        path.write_text(json.dumps(parts if with_end else parts[:1]))
        model = mo.TruncationModel(parts if with_end else parts[:1])
        out = s.truncate(b, kde_model=path, always_end=always_end, kde_models_length=ml, seed=23)
        want = mo.write_mdf([mo.trc_spec(md, g, 23, model=model, always_end=always_end, models_length=ml) for g, md in enumerate(mols)])
        got = s.to_mdf_text(out)
        assert got == want, (with_end, always_end, ml)
        out.free()
    with pytest.raises(Exception):
        s.truncate(b, kde_model=tmp_path / "model_False_True.json", always_end=False)       # no end_mtx and not --always-end
    # ... and the model file the REFERENCE wrote (py/truncate_kde.py printModelJson, tests/golden/make_kde_golden.py), as it is
    ref_model = os.path.join(GOLDEN, "kde_truncation_model.json")
    model = mo.TruncationModel(json.load(open(ref_model)))
    for always_end, ml in ((False, False), (False, True)):
        out = s.truncate(b, kde_model=ref_model, always_end=always_end, kde_models_length=ml, seed=29)
        want = mo.write_mdf([mo.trc_spec(md, g, 29, model=model, always_end=always_end, models_length=ml) for g, md in enumerate(mols)])
        assert s.to_mdf_text(out) == want, ("reference-written model", always_end, ml)
        out.free()
    b.free()


@pytest.mark.gpu
def test_config5_pipeline_pcr_truncation_seq_on_device(gseq, mo, po, oracle_models):
    """BASELINE config 5 in small: PCR (substitution-heavy copies, minus strands, interval ends) -> truncation -> Seq, molecule
    tables never leaving the device; Seq's records (perfect and Badread with q-scores) equal the oracle's Seq on the oracle's
    transformed molecules."""
    from conftest import ERR_MODEL, QS_MODEL
    s, ref = gseq
    text = _mdf(np.random.RandomState(31), 300)
    b0 = s.batch_from_mdf(text)
    b1 = s.pcr(b0, 8, 1200, error_rate=2.5e-3, efficiency=0.8, seed=2)           # ~ 6 substitutions per kb per copy
    b2 = s.truncate(b1, lognormal=(6.2, 0.5), seed=3)
    mols = mo.pcr_spec(mo.stream_mdf(text, unroll=True), 8, 0.8, 2.5e-3, 1200, 2)
    mols = [mo.trc_spec(md, g, 3, lognormal=(6.2, 0.5)) for g, md in enumerate(mols)]
    assert s.to_mdf_text(b2) == mo.write_mdf(mols)
    assert b2.n_mods > 2 * b2.n_reads
    s.set_identity(84.0, 99.0, 5.5)
    s.load_error_model(ERR_MODEL)
    s.load_qscore_model(QS_MODEL)
    perfect = s.run(b2, target="perfect", fastq=True, seed=5).records()
    bad = s.run(b2, target="badread", fastq=True, compute_qual=True, seed=5).records()
    ident = po.Identities(84.0, 5.5, 99.0)
    gen = list(po.mdf_generator(mo.write_mdf(mols).splitlines(keepends=True)))
    assert len(gen) == len(perfect) == len(bad)
    for i, (mid, ivs) in enumerate(gen):
        raw = po.splice(ref, ivs)
        assert perfect[i] == po.perfect_record(True, 5, i, raw, mid), i
        assert bad[i] == po.badread_record(True, 5, i, raw, ident, oracle_models["em"], oracle_models["qm"], True, mid)[0], i
    for x in (b2, b1, b0):
        x.free()


@pytest.mark.gpu
def test_mdf_parser_and_writer_do_not_depend_on_the_host_threads(gseq, mo):
    """20 000 molecules (1.4 MB of MDF text, a third of them with substitutions, depth 1 - 3): parsed and written back with 1 and
    with 7 host threads -- the same text, equal to the oracle's reader / writer pair; PCR of it as well."""
    s, _ = gseq
    text = _mdf(np.random.RandomState(31), 20000)
    assert len(text) > (1 << 20)
    want = mo.write_mdf(mo.stream_mdf(text, unroll=True))
    got = {}
    for nt in (1, 7):
        s.set_host_threads(nt)
        b = s.batch_from_mdf(text)
        got[nt] = s.to_mdf_text(b)
        out = s.pcr(b, 4, 30000, error_rate=1e-3, efficiency=0.8, seed=3)
        got[nt, "pcr"] = s.to_mdf_text(out)
        out.free(); b.free()
    s.set_host_threads(1)
    assert got[1] == want and got[7] == want
    assert got[7, "pcr"] == got[1, "pcr"] and got[1, "pcr"].count("\n+") > 20000


@pytest.mark.gpu
def test_device_batch_writer_reproduces_the_cpp_writer_known_answers(gseq):
    """tksmseq_batch_to_mdf_text on a parsed batch == the hand-derived output of the C++ reader / writer pair (cases that the
    Seq grammar, exactly 5 fields per segment line, accepts)"""
    s, _ = gseq
    ka = json.load(open(os.path.join(GOLDEN, "mdf_writer_known_answers.json")))
    for case in ka["cases"]:
        if not case["seq_grammar"]:
            continue
        b = s.batch_from_mdf(case["in"])
        assert s.to_mdf_text(b) == case["out"], case["name"]
        b.free()


@pytest.mark.gpu
def test_pcr_and_truncate_modules_on_files(mo, tmp_path):
    """`tksm pcr` / `tksm truncate` (PCR_module / Truncate_module surface: src/pcr.cpp:91-260, src/truncate.cpp:236-451): MDF
    file in, MDF file out, no reference needed; outputs equal the oracle; the reference's argument checks and exit codes."""
    import subprocess
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    text = _mdf(np.random.RandomState(77), 300)
    src = tmp_path / "in.mdf"
    src.write_text(text)
    mols = mo.stream_mdf(text, unroll=True)
    o1 = tmp_path / "pcr.mdf"
    r = subprocess.run([exe, "pcr", "-i", str(src), "-o", str(o1), "--cycles", "3", "-x", "T4", "--molecule-count", "500", "-s", "7"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    er, ef = mo.PRESETS["T4"]
    assert o1.read_text() == mo.write_mdf(mo.pcr_spec(mols, 3, ef, er, 500, 7))
    o2 = tmp_path / "trc.mdf"
    r = subprocess.run([exe, "truncate", "-i", str(o1), "-o", str(o2), "--lognormal", "5.5,0.7", "-s", "9"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    pm = mo.stream_mdf(o1.read_text(), unroll=True)
    assert o2.read_text() == mo.write_mdf([mo.trc_spec(md, g, 9, lognormal=(5.5, 0.7)) for g, md in enumerate(pm)])
    # argument checks (src/pcr.cpp:148-185, src/truncate.cpp:278-300)
    r = subprocess.run([exe, "pcr", "-i", str(src), "-o", str(o1), "--cycles", "3", "--molecule-count", "5"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error rate is required!" in r.stderr and "Efficiency is required!" in r.stderr
    r = subprocess.run([exe, "pcr", "-i", str(src), "-o", str(o1), "--cycles", "3", "--molecule-count", "5", "-x", "Pfu"], capture_output=True, text=True)
    assert r.returncode == 1 and "Preset Pfu not found" in r.stderr
    r = subprocess.run([exe, "truncate", "-i", str(src), "-o", str(o2)], capture_output=True, text=True)
    assert r.returncode == 1 and "One of kde-model, normal or lognormal is required!" in r.stderr
    r = subprocess.run([exe, "truncate", "-i", str(src), "-o", str(o2), "--normal", "5,1", "--lognormal", "5,1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Only one of kde-model, normal or lognormal is allowed!" in r.stderr


@pytest.mark.gpu
def test_chained_cli_equals_the_three_module_route(mo, tmp_path):
    """BASELINE config 5 as ONE command: `tksm sequence --pcr-... --truncate-...` keeps the molecule tables on the device between
    PCR, truncation and sequencing.  Its FASTQ is byte-equal to `tksm pcr` -> `tksm truncate` -> `tksm sequence` over MDF files with
    the same -s (src/pcr.cpp:91-260, src/truncate.cpp:236-451, py/sequence.py:323-376) -- with the module defaults, with small
    slices / batches and a repeated device (--devices 0,0), and with the KDE model the reference's writer produced; `tksm pcr` and
    `tksm truncate` themselves do not depend on their slice / batch sizes or device lists."""
    import subprocess
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    rs = np.random.RandomState(41)
    ref = _genome(rs)
    fa = tmp_path / "ref.fa"
    fa.write_text("".join(f">{k}\n{v}\n" for k, v in ref.items()))
    src = tmp_path / "in.mdf"
    src.write_text(_mdf(rs, 600))
    kde = os.path.join(GOLDEN, "kde_truncation_model.json")

    def run(*args):
        r = subprocess.run([exe, *[str(x) for x in args]], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (args, r.stderr[-600:])
        return r

    for tag, trc_mod, trc_chain in (("lognormal", ["--lognormal", "6.0,0.5"], ["--truncate-lognormal", "6.0,0.5"]),
                                    ("kde", ["--kde-model", kde], ["--truncate-kde-model", kde])):
        a, b, c3 = tmp_path / f"pcr_{tag}.mdf", tmp_path / f"trc_{tag}.mdf", tmp_path / f"three_{tag}.fastq"
        run("pcr", "-i", src, "-o", a, "--cycles", "9", "--molecule-count", "5000", "-x", "Taq-setting1", "-s", "5")
        run("truncate", "-i", a, "-o", b, *trc_mod, "-s", "5")
        run("sequence", "-i", b, "-r", fa, "-o", c3, "-s", "5")
        want = c3.read_bytes()
        assert want.count(b"\n") // 4 > 4000
        # the modules do not depend on slice / batch sizes or the device list
        a2, b2 = tmp_path / f"pcr2_{tag}.mdf", tmp_path / f"trc2_{tag}.mdf"
        run("pcr", "-i", src, "-o", a2, "--cycles", "9", "--molecule-count", "5000", "-x", "Taq-setting1", "-s", "5", "--slice-molecules", "700", "--devices", "0,0",
            "--verbosity", "DEBUG", "--log-file", tmp_path / "pcr.log")
        assert a2.read_bytes() == a.read_bytes() and "piece 3" in (tmp_path / "pcr.log").read_text()
        run("truncate", "-i", a, "-o", b2, *trc_mod, "-s", "5", "--batch-bytes", "60000", "--devices", "0,0", "--verbosity", "OFF")
        assert b2.read_bytes() == b.read_bytes()
        # one command, tables on the device
        for extra in ([], ["--pcr-slice-molecules", "900", "--devices", "0,0", "--in-flight", "2", "-t", "3"]):
            one = tmp_path / f"chained_{tag}_{len(extra)}.fastq"
            run("sequence", "-i", src, "-r", fa, "-o", one, "-s", "5", "--pcr-cycles", "9", "--pcr-molecule-count", "5000", "--pcr-preset", "Taq-setting1",
                *trc_chain, *extra)
            assert one.read_bytes() == want, (tag, extra)
    # truncation alone in front of the sequencer, streaming in small batches
    t1, s1, s2 = tmp_path / "t1.mdf", tmp_path / "s1.fastq", tmp_path / "s2.fastq"
    run("truncate", "-i", src, "-o", t1, "--normal", "450,120", "-s", "8")
    run("sequence", "-i", t1, "-r", fa, "-o", s1, "-s", "8", "--skip-qual-compute")
    run("sequence", "-i", src, "-r", fa, "-o", s2, "-s", "8", "--skip-qual-compute", "--truncate-normal", "450,120", "--batch-bytes", "20000")
    assert s1.read_bytes() == s2.read_bytes()
    # the chained stages' argument checks
    r = subprocess.run([exe, "sequence", "-i", str(src), "-r", str(fa), "-o", str(s2), "--pcr-cycles", "3"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "molecule-count is required!" in r.stderr and "Error rate is required!" in r.stderr
    r = subprocess.run([exe, "sequence", "-i", str(src), "-r", str(fa), "-o", str(s2), "--truncate-normal", "5,1", "--truncate-lognormal", "5,1"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "Only one of kde-model, normal or lognormal is allowed!" in r.stderr


@pytest.mark.gpu
def test_config5_at_20_million_molecules_does_not_depend_on_slices_or_device_groups(tmp_path):
    """BASELINE config 5 a tenth of its stated size, as the one command: 20 000 templates -> 20 M molecules (20 PCR cycles, Taq-setting1),
    lognormal truncation, Badread reads -- ~25 GB of FASTQ streamed into a pipe and hashed on the fly (xxh3-128), never held anywhere.
    Two runs: slices of 2 M copies on one device group, slices of 700 k copies on two groups (--devices 0,0, two contexts each, three
    parser threads): the same digest, the same 20 M reads; the host's resident set stays bounded (the reference's PCR holds every
    molecule in RAM: src/pcr.cpp:215).  The 200 M-molecule run of the same command: tools/config5_200M.py, profiles/r04_config5_200M.log."""
    import fcntl
    import re
    import resource
    import subprocess
    import threading
    import xxhash
    from tksm_amd import synthetic
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
    rs = np.random.RandomState(5)
    lens = [4_000_000] * 4
    with open(tmp_path / "ref.fa", "w") as f:
        for c, L in enumerate(lens):
            f.write(f">chr{c + 1}\n" + rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode() + "\n")
    m = synthetic.make_molecules(rs, lens, 20_000, 1000, 200)
    (tmp_path / "in.mdf").write_text(synthetic.mdf_text(m, [f"chr{c + 1}" for c in range(4)]))
    target = 20_000_000

    def run(tag, *extra):
        fifo = tmp_path / f"{tag}.fastq"
        os.mkfifo(fifo)
        got = {}

        def drain():
            h, n, lines = xxhash.xxh3_128(), 0, 0
            with open(fifo, "rb", buffering=0) as p:
                try:
                    fcntl.fcntl(p.fileno(), 1031, 1 << 20)            # F_SETPIPE_SZ
                except OSError:
                    pass
                while True:
                    b = p.read(1 << 24)
                    if not b:
                        break
                    h.update(b); n += len(b); lines += b.count(b"\n")
            got.update(digest=h.hexdigest(), bytes=n, lines=lines)
        th = threading.Thread(target=drain, daemon=True)
        th.start()
        r = subprocess.run([exe, "sequence", "-i", str(tmp_path / "in.mdf"), "-r", str(tmp_path / "ref.fa"), "-o", str(fifo), "-s", "7", "-t", "3",
                            "--pcr-cycles", "20", "--pcr-molecule-count", str(target), "--pcr-preset", "Taq-setting1", "--truncate-lognormal", "6.9,0.5", *extra],
                           capture_output=True, text=True, env=env, timeout=900)
        th.join(timeout=120)
        assert r.returncode == 0, r.stderr[-800:]
        reads = int(re.search(r"Sequencing: (\d+) reads", r.stderr).group(1))
        return got, reads
    a, reads_a = run("a", "--pcr-slice-molecules", "2000000", "--devices", "0")
    b, reads_b = run("b", "--pcr-slice-molecules", "700000", "--devices", "0,0", "--in-flight", "2")
    assert reads_a == reads_b and 0.97 * target < reads_a < 1.03 * target            # (the written copies are a draw around the target)
    assert a["lines"] == 4 * reads_a and a["bytes"] > 1000 * reads_a
    assert a == b
    assert resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss < 24 * 2**20          # KiB: under 24 GiB of host memory for ~25 GB of output
