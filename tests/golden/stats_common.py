"""Per-read statistics of a simulated read against its error-free molecule, shared by the fixture generator
(tests/golden/make_golden.py: the reference's reads) and the CPU test (tests/test_oracle_golden.py: the oracle's
reads), so that both sides are measured with the same ruler.  Original code; nothing here comes from the reference."""
import re

import numpy as np

POS_BINS = 20            # relative position along the molecule, for the per-position edit-type rates
INS_BINS = 16            # insertion-run lengths 1..14, 15+
_CIG = re.compile(r"(\d+)([=XID])")


def cigar_stats(cigar, raw_len):
    """cigar = alignment of the read (query) against the error-free molecule (target): 'I' read-only, 'D' molecule-only.
    Returns (counts per op, insertion-run histogram [INS_BINS], per-position op counts [3 x POS_BINS] for X / I / D)."""
    cnt = {"=": 0, "X": 0, "I": 0, "D": 0}
    ins_hist = np.zeros(INS_BINS, np.int64)
    pos = np.zeros((3, POS_BINS), np.int64)
    t = 0                                              # position in the molecule
    scale = POS_BINS / max(1, raw_len)
    for m in _CIG.finditer(cigar):
        n, op = int(m.group(1)), m.group(2)
        cnt[op] += n
        if op == "I":
            ins_hist[min(n, INS_BINS - 1)] += 1
            pos[1, min(POS_BINS - 1, int(t * scale))] += n
        elif op == "=":
            t += n
        else:
            row = 0 if op == "X" else 2
            b0 = np.minimum(POS_BINS - 1, ((t + np.arange(n)) * scale).astype(np.int64))
            np.add.at(pos[row], b0, 1)
            t += n
    return cnt, ins_hist, pos


def qscore_hist(cigar, qual):
    """[3 x 94] histogram of the read's q-scores conditioned on the op of the read position (=, X, I)."""
    qh = np.zeros((3, 94), np.int64)
    p = 0
    q = np.frombuffer(qual if isinstance(qual, bytes) else qual.encode(), np.uint8).astype(np.int64) - 33
    for m in _CIG.finditer(cigar):
        n, op = int(m.group(1)), m.group(2)
        if op == "D":
            continue
        np.add.at(qh["=XI".index(op)], q[p:p + n], 1)
        p += n
    return qh
