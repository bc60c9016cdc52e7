#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the REAL reference Python
(/root/reference/py/sequence.py + tksm_badread.py), imported in the build container.

The reference needs `edlib`, which is neither vendored nor installed (env.yaml:8, unpinned).  An
in-memory `sys.modules["edlib"]` stand-in exposing align(query, target, task="path") -> {"cigar"} is
provided by oracle/pyoracle.edlib_align (exact NW, edlib's traceback preference restated from its
published algorithm).  Everything that never reaches edlib is therefore pinned bit-exactly by the
reference itself; edlib-dependent values are pinned only up to alignment tie-breaking
("parity unpinned" for that).

Run (container only; the reference does not exist on the GPU box):
    python tests/golden/make_golden.py [--reads-1000 N] ...
Outputs are small data files; no reference source text is copied.
"""
import argparse
import contextlib
import io
import json
import os
import random
import runpy
import sys
import types
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
MODELS = os.path.join(ROOT, "tksm_amd", "models")


def import_reference():
    import pyoracle
    shim = types.ModuleType("edlib")
    shim.align = pyoracle.edlib_align
    sys.modules["edlib"] = shim
    sys.path.insert(0, os.path.join(REF, "py"))
    import tksm_badread
    import sequence as ref_sequence
    return pyoracle, tksm_badread, ref_sequence


# ----------------------------------------------------------------------------- G3 splice corpus
def make_splice_corpus(rs):
    """FASTA with lower-case, N runs and IUPAC codes + MDF covering the edge cases of
    py/sequence.py:197-239,303-313."""
    alpha = np.array(list("ACGT"))
    contigs = {}
    for name, n in (("chr1", 5000), ("chr2", 3001), ("2", 777), ("scaf_3", 64)):
        s = rs.choice(alpha, n)
        contigs[name] = s
    c1 = contigs["chr1"]
    c1[100:180] = "N"
    c1[1000:1400] = np.char.lower(c1[1000:1400])
    for p, ch in ((2000, "R"), (2001, "y"), (2500, "K"), (2600, "n"), (2601, "M"), (0, "N"), (4999, "s")):
        c1[p] = ch
    contigs["chr2"][1500:1510] = "N"
    fasta = []
    for name, s in contigs.items():
        s = "".join(s)
        fasta.append(f">{name} some description\n")
        w = 60 if name != "2" else 10_000
        fasta += [s[i:i + w] + "\n" for i in range(0, len(s), w)]
    ref = {k: "".join(v) for k, v in contigs.items()}
    mdf = []
    names = list(contigs)
    mid = 0

    def mods_for(length, k):
        out = []
        for _ in range(k):
            out.append(f"{rs.randint(0, length)}{rs.choice(list('ACGTacgtN'))}")
        return ",".join(out)

    # hand-written edge cases
    edge = [
        ("e_first_last_mod", 1, [("chr1", 10, 20, "+", "0T,9G"), ("chr1", 30, 41, "-", "0A,10C")]),
        ("e_overlap_mods", 1, [("chr2", 5, 25, "+", "3A,3C,3G")]),
        ("e_clamp_end", 1, [("scaf_3", 60, 100, "+", ""), ("scaf_3", 60, 100, "-", "1T")]),
        ("e_clamp_all", 1, [("scaf_3", 64, 70, "+", ""), ("chr2", 0, 5, "+", "")]),
        ("e_empty_iv", 1, [("chr1", 50, 50, "+", ""), ("chr1", 60, 50, "-", ""), ("chr2", 7, 9, "+", "")]),
        ("e_Nrun", 1, [("chr1", 90, 190, "+", ""), ("chr1", 90, 190, "-", "")]),
        ("e_lower", 1, [("chr1", 990, 1410, "-", "5a,6c")]),
        ("e_iupac", 1, [("chr1", 1995, 2005, "+", ""), ("chr1", 1995, 2005, "-", ""), ("chr1", 2595, 2605, "-", "")]),
        ("e_literal_polyA", 1, [("chr2", 100, 160, "+", ""), ("AAAAAAAAAAAAAAAAAAAAAAAAA", 0, 25, "+", "")]),
        ("e_literal_tag", 1, [("ACGTNRYacgt", 0, 11, "+", ""), ("ACGTNRYacgt", 2, 9, "-", "0T"), ("chr1", 0, 3, "+", "")]),
        ("e_literal_barcode", 1, [("GATTACAGATTACAGA", 0, 16, "+", ""), ("TTTTTTTTTT", 0, 10, "+", ""), ("chr2", 2000, 2100, "-", "")]),
        ("e_depth3", 3, [("2", 0, 40, "+", "39N")]),
        ("e_depth0", 0, [("2", 0, 40, "+", "")]),
        ("e_numeric_contig", 1, [("2", 700, 777, "-", "")]),
        ("e_whole_contig", 1, [("scaf_3", 0, 64, "-", "")]),
        ("e_comment", 1, [("chr1", 1, 2, "+", "")]),
    ]
    for name, depth, ivs in edge:
        comment = "tid=ENST0001;CB=ACGT;" if name == "e_comment" else ""
        mdf.append(f"+{name}\t{depth}\t{comment}\n")
        for c, s, e, st, m in ivs:
            mdf.append(f"{c}\t{s}\t{e}\t{st}\t{m}\n")
    for _ in range(200):
        mid += 1
        depth = 1 if rs.rand() < 0.9 else int(rs.randint(2, 4))
        mdf.append(f"+mol_{mid}\t{depth}\tdepth={depth};\n")
        for _ in range(int(rs.randint(1, 7))):
            c = names[rs.randint(0, len(names))]
            n = len(ref[c])
            length = int(min(n, max(1, rs.normal(300, 200))))
            s = int(rs.randint(0, n - length + 1))
            e = s + length
            if rs.rand() < 0.1:
                e += int(rs.randint(0, 50))          # slice clamp past the contig end
            eff = min(e, n) - s
            k = 0 if rs.rand() < 0.5 else int(rs.randint(1, 5))
            mdf.append(f"{c}\t{s}\t{e}\t{'+' if rs.rand() < 0.5 else '-'}\t{mods_for(eff, k)}\n")
        if rs.rand() < 0.3:
            tail = "A" * int(rs.randint(1, 40))
            mdf.append(f"{tail}\t0\t{len(tail)}\t+\t\n")
    return "".join(fasta), "".join(mdf)


def run_reference_cli(argv):
    """Runs the reference's real main block (py/sequence.py:323-376) in-process."""
    old = sys.argv
    sys.argv = ["sequence"] + argv
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            runpy.run_path(os.path.join(REF, "py", "sequence.py"), run_name="__main__")
    finally:
        sys.argv = old


def strip_uuid(text):
    out = []
    for line in text.splitlines(keepends=True):
        if line[:1] in "@>" and len(line) > 37 and line[37] == " ":
            line = line[0] + "UUID" + line[37:]
        out.append(line)
    return "".join(out)


# ----------------------------------------------------------------------------- G6 stochastic stats
_W = {}
SHIPPED_MODELS = ("nanopore2020", "nanopore2018", "pacbio2016")


def _worker_init(model="nanopore2020"):
    os.nice(10)
    pyoracle, tb, _ = import_reference()
    _W["po"], _W["tb"] = pyoracle, tb
    sink = io.StringIO()
    _W["em"] = tb.ERROR_MODEL_PY.ErrorModel(os.path.join(MODELS, "badread", model + ".error.gz"), sink)
    _W["qm"] = tb.QSCOREMODEL_PY.QScoreModel(os.path.join(MODELS, "badread", model + ".qscore.gz"), sink)
    _W["tail"] = tb.TAIL_NOISE_MODEL_PY.KDE_noise_generator.load("no_noise")
    _W["ident"] = tb.IDENTITIES_PY.Identities(84.0, 5.5, 99.0, sink)


def _worker(job):
    """One read through the reference's own sequence_fragment (+ get_qscores), seeded here (the reference is unseeded)."""
    from stats_common import cigar_stats, qscore_hist
    L, seed, compute_q = job
    po, tb = _W["po"], _W["tb"]
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    rs = np.random.RandomState(seed % (2 ** 32))
    raw = "".join(rs.choice(list("ACGT"), L))
    em = _W["em"]
    calls = {"n": 0, "noop": 0, "aln": 0}
    orig_add = em.add_errors_to_kmer

    def counting_add(kmer):
        r = orig_add(kmer)
        calls["n"] += 1
        if kmer == "".join(r):
            calls["noop"] += 1
        return r
    em.add_errors_to_kmer = counting_add
    edl = tb.edlib
    orig_align = edl.align

    def counting_align(*a, **k):
        calls["aln"] += 1
        return orig_align(*a, **k)
    edl.align = counting_align
    try:
        target = _W["ident"].get_identity()
        seq, qual, ident, _ = tb.SIMULATE_PY.sequence_fragment(raw, target, em, _W["qm"], _W["tail"], compute_q)
    finally:
        em.add_errors_to_kmer = orig_add
        edl.align = orig_align
    if len(seq):
        _, cig = po.nw_cigar(seq, raw)
        cnt, ins_hist, pos = cigar_stats(cig, L)
        qh = qscore_hist(cig, qual) if compute_q else np.zeros((3, 94), np.int64)
    else:
        from stats_common import INS_BINS, POS_BINS
        cnt, ins_hist, pos, qh = {"=": 0, "X": 0, "I": 0, "D": L}, np.zeros(INS_BINS, np.int64), np.zeros((3, POS_BINS), np.int64), np.zeros((3, 94), np.int64)
    return dict(L=L, out_len=len(seq), identity=ident, target=target, draws=calls["n"], noop=calls["noop"],
                aligns=calls["aln"] - (1 if compute_q else 0), X=cnt["X"], I=cnt["I"], D=cnt["D"], M=cnt["="],
                qh=qh, ins_hist=ins_hist, pos=pos)


def make_stochastic(model, jobs, procs):
    with Pool(procs, initializer=_worker_init, initargs=(model,)) as p:
        return p.map(_worker, jobs, chunksize=16)


def stochastic_fixture(model, n_q, n_noq, procs, lengths=(300, 1000, 3000)):
    """>= 20 k reference reads with q-scores and n_noq without, per length, for one shipped model."""
    mi = SHIPPED_MODELS.index(model)
    jobs = []
    for L in lengths:
        jobs += [(L, 1000000 * (10 * mi + L // 300) + i, True) for i in range(n_q)]
        jobs += [(L, 500000000 + 1000000 * (10 * mi + L // 300) + i, False) for i in range(n_noq)]
    res = make_stochastic(model, jobs, procs)
    cols = ["L", "out_len", "identity", "target", "draws", "noop", "aligns", "X", "I", "D", "M"]
    arr = {}
    for c in cols:
        a = np.array([r[c] for r in res])
        arr[c] = a.astype(np.float32) if a.dtype.kind == "f" else a.astype(np.int32)
    arr["with_q"] = np.array([j[2] for j in jobs])
    for L in lengths:
        for wq in (True, False):
            sel = [r for r, j in zip(res, jobs) if j[0] == L and j[2] == wq]
            tag = f"{L}_{'q' if wq else 'noq'}"
            arr[f"ins_hist_{tag}"] = np.sum([r["ins_hist"] for r in sel], axis=0)
            arr[f"pos_{tag}"] = np.sum([r["pos"] for r in sel], axis=0)
            if wq:
                arr[f"qhist_{L}"] = np.sum([r["qh"] for r in sel], axis=0)
    np.savez_compressed(os.path.join(HERE, f"badread_reference_stats_{model}.npz"), **arr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads-q", type=int, default=20000, help="reads with q-scores per model and length")
    ap.add_argument("--reads-noq", type=int, default=10000, help="reads without q-scores per model and length")
    ap.add_argument("--models", default=",".join(SHIPPED_MODELS))
    ap.add_argument("--only-stochastic", action="store_true", help="regenerate the G6 statistics only")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--skip-stochastic", action="store_true")
    args = ap.parse_args()
    po, tb, ref_sequence = import_reference()
    if args.only_stochastic:
        for model in args.models.split(","):
            stochastic_fixture(model, args.reads_q, args.reads_noq, args.procs)
            print("G6", model, "done", flush=True)
        return

    # G1 / G2 / G5: documentation known answers, evaluated by the reference's own functions
    ka = {}
    ref_sequence.reference_seqs = {"1": "AGTCCCGTAA"}
    t = {"p": [ref_sequence.perfect, ref_sequence.fasta_formatter]}
    r = ref_sequence.mdf_to_seq(("m1", [("1", 0, 4, "+", "2C,3T"), ("1", 6, 9, "+", "1G")]), t)["p"]
    ka["readme_mods"] = r.split("\n")[1]                       # README.md:233-251 -> AGCTGGA
    ref_sequence.reference_seqs = {"1": "AGTC"}
    r = ref_sequence.mdf_to_seq(("m1", [("TT", 0, 2, "+", ""), ("1", 0, 4, "+", "")]), t)["p"]
    ka["readme_literal"] = r.split("\n")[1]                    # README.md:253-270 -> TTAGTC
    ka["revcomp"] = {s: ref_sequence.reverse_complement(s) for s in
                     ["", "AGTCATGC", "AGTCATCGATCGACGACTACG", "A", "g", "U", "t", "C", "N", "ACGTNRYKMacgtn"]}
    ka["align_kmers"] = {f"{a},{b}": tb.ERROR_MODEL_PY.align_kmers(a, b) for a, b in
                         [("ACGT", "ACGTT"), ("ACGT", "ACT"), ("AAAAAAA", "AAAAAA"), ("AAAAAAA", "AA"),
                          ("ACGTACG", "ACTTTACG"), ("ACGTACG", "AGTCACG"), ("GATTACA", "GATTTTACA")]}
    ka["beta_parameters_84_5.5_99"] = list(tb.IDENTITIES_PY.beta_parameters(84.0, 5.5, 99.0))
    ka["beta_parameters_90_4_98"] = list(tb.IDENTITIES_PY.beta_parameters(90.0, 4.0, 98.0))
    ka["pct_format"] = {repr(x): f"{x * 100.0:.2f}" for x in
                        [1.0, 0.0, 0.87704130643, 0.999949999, 0.99995, 0.5, 0.123456789, 0.855, 0.84125, 0.84135,
                         877 / 1000, 931 / 1063, 2 / 3, 0.99994999999999, 0.100005, 0.100015]}
    json.dump(ka, open(os.path.join(HERE, "known_answers.json"), "w"), indent=1, sort_keys=True)

    # G3: splice corpus through the reference's real CLI main block, FASTQ and FASTA
    rs = np.random.RandomState(7)
    fasta, mdf = make_splice_corpus(rs)
    d = os.path.join(HERE, "splice_corpus")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "ref.fa"), "w").write(fasta)
    open(os.path.join(d, "mols.mdf"), "w").write(mdf)
    for ext in ("fastq", "fasta"):
        tmp = os.path.join("/tmp", f"golden_perfect.{ext}")
        run_reference_cli(["-i", os.path.join(d, "mols.mdf"), "-r", os.path.join(d, "ref.fa"), "--perfect", tmp])
        open(os.path.join(d, f"expected_perfect.{ext}"), "w").write(strip_uuid(open(tmp).read()))

    # G4: model parsing through the reference's loaders
    sink = io.StringIO()
    em = tb.ERROR_MODEL_PY.ErrorModel(os.path.join(MODELS, "badread", "nanopore2020.error.gz"), sink)
    kmers = sorted(em.alternatives)
    pick = [kmers[i] for i in rs.choice(len(kmers), 64, replace=False)] + ["AAAAAAA", "TTTTTTT", "ACGTACG"]
    g4 = {"kmer_size": em.kmer_size, "n_kmers": len(kmers),
          "sample": {k: {"alts": em.alternatives[k], "probs": em.probabilities[k]} for k in pick}}
    qm = tb.QSCOREMODEL_PY.QScoreModel(os.path.join(MODELS, "badread", "nanopore2020.qscore.gz"), sink)
    keys = list(qm.scores)
    pickq = [keys[i] for i in rs.choice(len(keys), 50, replace=False)] + ["=", "X", "I"]
    g4["qscore"] = {"kmer_size": qm.kmer_size, "n_keys": len(keys),
                    "sample": {k: {"scores": qm.scores[k], "probs": qm.probabilities[k]} for k in pickq}}
    json.dump(g4, open(os.path.join(HERE, "model_parse_nanopore2020.json"), "w"), sort_keys=True)

    # G6: per-read statistics of the reference's stochastic path, every shipped model (seeded here; the reference is unseeded)
    if not args.skip_stochastic:
        for model in args.models.split(","):
            stochastic_fixture(model, args.reads_q, args.reads_noq, args.procs)
            print("G6", model, "done", flush=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
