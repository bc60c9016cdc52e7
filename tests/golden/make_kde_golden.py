#!/usr/bin/env python3
"""Model files WRITTEN BY THE REFERENCE'S OWN WRITERS, for the tests of the truncation and tail-noise paths.

The reference ships neither a KDE truncation model nor a tail-noise model; both are built by its Python from alignments:
  * py/truncate_kde.py main() -> ComputeKDELikelihoods -> printModelJson (:298-320): the JSON `tksm truncate --kde-model` reads
    (src/truncate.cpp:362-381) -- "KDE_mtx" with `grid.T.flatten()`, labels `x[1:] + y[1:]`, and the 100-bin "end_mtx";
  * py/tksm_badread.py KDE_noise_generator.from_data (:888-901) and .save (:935-942): the tail-noise model of
    `tksm sequence --badread-tail-model`.
This script imports that code in the build container (scikit-learn is installed here; the reference does not exist on the GPU
box), feeds it synthetic alignments -- a PAF-like file of 12 000 primary mappings for the truncation model, 6 000 (mapped,
unmapped) length pairs and a transition matrix for the tail model -- and commits what the reference wrote:
    kde_truncation_model.json   written by printModelJson
    tail_model_reference.json   written by KDE_noise_generator.save
    tail_model_reference_stats.npz   tail lengths / first bases / transitions of 40 000 noise_seq() calls of the REFERENCE class
                                     per fragment length on that model (what the oracle's tail path is compared with)
Run (container only): python tests/golden/make_kde_golden.py
No reference source text is copied; the outputs are data."""
import importlib.util
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, import_reference  # noqa: E402

FRAG_LENS = [260, 700, 1200, 1975, 2600]      # 2600 is past the last label: the len(ly) / ly[-1] factor applies
N = 40000


def synthetic_paf(path, rs, n=12000):
    """primary mappings of truncated reads on transcripts: columns as py/truncate_kde.py:164-185 reads them (strand, target
    length / start / end, the tp:A:P tag)"""
    with open(path, "w") as f:
        for i in range(n):
            tlen = int(np.clip(rs.lognormal(7.0, 0.45), 300, 2900))
            trunc = int(min(tlen - 100, rs.gamma(1.6, 140.0))) if rs.rand() < 0.8 else 0
            at_end = int(round(trunc * rs.beta(0.7, 0.5)))
            strand = "+-"[int(rs.rand() < 0.5)]
            if strand == "+":
                tstart, tend = trunc - at_end, tlen - at_end
            else:
                tstart, tend = at_end, tlen - (trunc - at_end)
            f.write(f"r{i}\t{tend - tstart}\t0\t{tend - tstart}\t{strand}\tt{i % 500}\t{tlen}\t{tstart}\t{tend}\t{tend - tstart}\t{tend - tstart}\t60\ttp:A:P\n")
            if i % 7 == 0:                       # a secondary mapping: ignored by the reference (no tp:A:P)
                f.write(f"r{i}\t100\t0\t100\t+\tt0\t{tlen}\t0\t100\t100\t100\t0\ttp:A:S\n")


def truncation_model(rs):
    spec = importlib.util.spec_from_file_location("ref_truncate_kde", os.path.join(REF, "py", "truncate_kde.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as d:
        paf = os.path.join(d, "reads.paf")
        synthetic_paf(paf, rs)
        out = os.path.join(HERE, "kde_truncation_model.json")
        argv = sys.argv
        sys.argv = ["truncate_kde.py", "-i", paf, "-o", out, "-b", "120", "--grid-start", "0", "--grid-end", "3000", "--grid-step", "100", "-t", "1"]
        try:
            mod.main()                           # the reference's own main(): KDE on the grid, printModelJson
        finally:
            sys.argv = argv
    return out


def tail_model(rs):
    _, tb, _ = import_reference()
    gen_cls = tb.TAIL_NOISE_MODEL_PY.KDE_noise_generator
    n = 6000
    mapped = np.clip(rs.lognormal(6.6, 0.5, n), 200, 1900).astype(int)
    unmapped = np.where(rs.rand(n) < 0.4, np.clip(rs.gamma(2.0, 60.0, n) + 0.05 * mapped, 1, 700), 0).astype(int)
    # from_data scores the square grid labels x labels -- grid[a][b] = density(mapped = labels[a], unmapped = labels[b]) -- and the
    # sampler then takes row a by the fragment length and draws a tail length from the same labels: one label axis for both, as
    # the reference's code has it
    labels = np.arange(0, 2000, 50)
    trans = [[0.25, 0.25, 0.25, 0.25],
             [[0.55, 0.15, 0.20, 0.10], [0.20, 0.45, 0.25, 0.10], [0.30, 0.10, 0.50, 0.10], [0.15, 0.30, 0.15, 0.40]]]
    gen = gen_cls.from_data(list(mapped), list(unmapped), labels, labels, trans, 60.0, threads=1)
    path = os.path.join(HERE, "tail_model_reference.json")
    with open(path, "w") as f:
        gen.save(f)                              # KDE_noise_generator.save
    gen = gen_cls.load(path)                     # ... and read back the way `tksm sequence` does
    random.seed(77)
    out = {"frag_lens": np.array(FRAG_LENS), "n": np.array(N)}
    code = {c: i for i, c in enumerate("ACGT")}
    for fl in FRAG_LENS:
        lens = np.zeros(N, np.int64); first = np.zeros(4, np.int64); tr = np.zeros((4, 4), np.int64)
        for t in range(N):
            s = gen.noise_seq(fl)
            lens[t] = len(s)
            if s:
                c = np.array([code[ch] for ch in s])
                first[c[0]] += 1
                np.add.at(tr, (c[:-1], c[1:]), 1)
        vals, cnt = np.unique(lens, return_counts=True)
        out[f"len_values_{fl}"] = vals; out[f"len_counts_{fl}"] = cnt
        out[f"first_{fl}"] = first; out[f"trans_{fl}"] = tr
        print("tail", fl, "empty", (lens == 0).mean(), "mean", lens.mean(), "max", lens.max(), flush=True)
    np.savez_compressed(os.path.join(HERE, "tail_model_reference_stats.npz"), **out)
    return path


def main():
    rs = np.random.RandomState(20261004)
    print("truncation model:", truncation_model(rs))
    print("tail model:", tail_model(rs))


if __name__ == "__main__":
    main()
