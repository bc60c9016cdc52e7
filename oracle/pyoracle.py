"""oracle/pyoracle.py -- Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (tksm_amd) never does.

It restates the reference's host-side Python (file:line into /root/reference) in numpy and binds
the scalar C restatement (oracle/tksm_oracle.c) for the per-read loops:

  generate_fasta / get_reference_seqs     py/sequence.py:168-194
  mdf_generator                           py/sequence.py:197-221
  mdf_to_seq + perfect/badread + formats  py/sequence.py:242-320
  ErrorModel.load_from_file, align_kmers  py/tksm_badread.py:91-117, :146-197
  QScoreModel.load_from_file              py/tksm_badread.py:546-582
  Identities / beta_parameters            py/tksm_badread.py:703-757

The error / q-score tables it builds are the *specification* of the packed layouts the product
loader (tksm_amd/csrc/models.cpp) must reproduce bit for bit.
"""
import ctypes as C
import gzip
import json
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtksm_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "tksm_oracle.c")):
        build()
    return C.CDLL(_SO)


_lib = _load()
_u8p = C.POINTER(C.c_uint8)


class _ErrModel(C.Structure):
    _fields_ = [("type", C.c_int32), ("k", C.c_int32), ("max_alts", C.c_int32), ("pad", C.c_int32),
                ("cdf", C.c_void_p), ("alts", C.c_void_p), ("nalts", C.c_void_p)]


class _QsModel(C.Structure):
    _fields_ = [("n_slots", C.c_int32), ("kmer_size", C.c_int32), ("keys", C.c_void_p),
                ("row_off", C.c_void_p), ("row_cnt", C.c_void_p), ("cdf_pool", C.c_void_p),
                ("q_pool", C.c_void_p)]


class _IdentModel(C.Structure):
    _fields_ = [("constant", C.c_int32), ("pad", C.c_int32), ("value", C.c_double), ("qtab", C.c_void_p)]


class _TailModel(C.Structure):
    _fields_ = [("n_lx", C.c_int32), ("n_ly", C.c_int32), ("lx", C.c_void_p), ("ly", C.c_void_p), ("grid", C.c_void_p),
                ("trans", C.c_double * 16), ("ratio", C.c_double), ("bases", C.c_uint8 * 4), ("pad", C.c_uint8 * 4)]


class FragStats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_draws", "n_noop", "n_kmers_applied", "change_count", "n_aligns", "n_random_change",
                 "n_sub", "n_ins_slots", "n_del", "ins_bases", "frag_len", "new_len", "start_trim",
                 "end_trim", "band_fail", "pad0")] + [("errors", C.c_double), ("target_identity", C.c_double)]


_lib.oracle_nw_path.restype = C.c_int
_lib.oracle_nw_path.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
_lib.oracle_splice_interval.restype = C.c_int64
_lib.oracle_target_identity.restype = C.c_double
_lib.oracle_target_identity.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
_lib.oracle_sequence_fragment.restype = C.c_int
_lib.oracle_sequence_fragment.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p,
                                          C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(FragStats)]
_lib.oracle_set_qscore_alignment_variant.argtypes = [C.c_int]


def set_qscore_alignment_variant(v):
    """TEST-ONLY switch of the q-score alignment's path choice (tksm_oracle.c: 0 shipped, 1 opposite indel preference, 2 edlib as
    published incl. its Hirschberg branch, 3 shipped order without the band); process-wide."""
    _lib.oracle_set_qscore_alignment_variant(int(v))


_lib.oracle_format_record.restype = C.c_int64
_lib.oracle_format_record.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p,
                                      C.c_int64, C.c_int64, C.c_double, C.c_char_p, C.c_int]
_lib.oracle_pct_hundredths.restype = C.c_int64
_lib.oracle_pct_hundredths.argtypes = [C.c_double]
_lib.oracle_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
_lib.oracle_qs_hash.restype = C.c_uint64
_lib.oracle_qs_hash.argtypes = [C.c_uint64]
_lib.oracle_tail_length.restype = C.c_int
_lib.oracle_tail_length.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]
_lib.oracle_tail_noise.restype = C.c_int
_lib.oracle_tail_noise.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]


def philox(seed, read, stream, n):
    out = (C.c_uint32 * 4)()
    _lib.oracle_philox(seed, read, stream, n, out)
    return list(out)


# ----------------------------------------------------------------------------- edlib stand-in
def nw_cigar(query, target):
    """edlib.align(query, target, task='path')['cigar'] restated (extended cigar, =XID)."""
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    ops = C.create_string_buffer(len(q) + len(t) + 1)
    n = C.c_int(0)
    d = _lib.oracle_nw_path(q, len(q), t, len(t), ops, C.byref(n))
    if d < 0:
        raise MemoryError("oracle_nw_path")
    s = ops.raw[: n.value].decode()
    return d, "".join(f"{len(m.group(0))}{m.group(0)[0]}" for m in re.finditer(r"=+|X+|I+|D+", s))


def edlib_align(query, target, task="path", **_):
    """Drop-in for the one edlib entry point the reference uses."""
    d, cigar = nw_cigar(query, target)
    return {"editDistance": d, "cigar": cigar}


# ----------------------------------------------------------------------------- host-side restatement
def _open(path, mode="rt"):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, mode) if magic == b"\x1f\x8b" else open(path, mode)


def generate_fasta(path):                      # py/sequence.py:168-186
    name, seq = "", []
    f = gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")
    with f:
        for l in f:
            l = l.rstrip("\n")
            if l[0] == ">":
                if len(seq) == 0:
                    name = l[1:].split(" ")[0]
                    continue
                yield name, "".join(seq)
                seq = []
                name = l[1:].split(" ")[0]
            else:
                seq.append(l)
    yield name, "".join(seq)


def get_reference_seqs(paths):                 # py/sequence.py:189-194
    ref = {}
    for p in paths:
        ref.update({n: s for n, s in generate_fasta(p)})
    return ref


def mdf_generator(lines):                      # py/sequence.py:197-221
    read_id, intervals, depth = None, [], 0
    for line in lines:
        line = line.strip("\n").split("\t")
        if line[0][0] == "+":
            if read_id is not None:
                for _ in range(depth):
                    yield read_id, intervals
            read_id, depth, intervals = line[0][1:], int(line[1]), []
        else:
            chrom, start, end, strand, mods = line
            intervals.append((chrom, int(start), int(end), strand, mods))
    if read_id is not None:
        for _ in range(depth):
            yield read_id, intervals


def splice(reference_seqs, intervals):         # py/sequence.py:303-313 via the C restatement
    out = []
    for chrom, start, end, strand, mods in intervals:
        contig = reference_seqs.get(chrom, chrom).encode()
        mp, mc = [], []
        if mods != "":
            for mod in mods.split(","):
                mc.append(ord(mod[-1]))
                mp.append(int(mod[:-1]))
        n = max(0, min(end, len(contig)) - min(start, len(contig)))
        buf = C.create_string_buffer(n + 1)
        r = _lib.oracle_splice_interval(contig, C.c_int64(len(contig)), C.c_int64(start), C.c_int64(end),
                                        C.c_int(1 if strand == "+" else 0),
                                        (C.c_int64 * len(mp))(*mp), (C.c_uint8 * len(mc))(*mc), C.c_int(len(mp)), buf)
        if r < 0:
            raise IndexError("modification position out of range")
        out.append(buf.raw[:r])
    return b"".join(out)


# ----------------------------------------------------------------------------- models
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
ALT_NOOP = 1 << 63


def align_kmers(kmer, alt):                    # py/tksm_badread.py:146-197
    assert len(kmer) > 2 and len(alt) > 1
    result = [kmer[0]] + [None] * (len(kmer) - 2) + [kmer[-1]]
    assert kmer[0] == alt[0] and kmer[-1] == alt[-1]
    kmer, alt = kmer[1:-1], alt[1:-1]
    if len(alt) == 0:
        cigar = "{}D".format(len(kmer))
    else:
        cigar = nw_cigar(alt, kmer)[1]
    kmer_pos, alt_pos = 0, 0
    for c in re.findall(r"\d+[IDX=]", cigar):
        t, size = c[-1], int(c[:-1])
        if t in "=X":
            for _ in range(size):
                result[kmer_pos + 1] = alt[alt_pos]
                alt_pos += 1
                kmer_pos += 1
        elif t == "D":
            for _ in range(size):
                result[kmer_pos + 1] = ""
                kmer_pos += 1
        else:
            result[kmer_pos] += alt[alt_pos: alt_pos + size]
            alt_pos += size
    if len(result[0]) == 2:
        first, ins = result[0]
        result[0] = first
        result[1] = ins + result[1]
    return result


def pack_alt(slots, noop):
    v, b = 0, 0
    for j, s in enumerate(slots):
        assert len(s) <= 7
        v |= len(s) << (3 * j)
        for ch in s:
            assert b < 19
            v |= _CODE[ch] << (24 + 2 * b)
            b += 1
    return v | (ALT_NOOP if noop else 0)


def cdf_thresholds(probs, residual_to_one):
    """u32 cumulative thresholds: P(choice <= a) = thr[a] / 2^32.
    residual_to_one: error model (total = 1.0 when sum < 1, py/tksm_badread.py:135-140);
    else total = sum (random.choices semantics, py/tksm_badread.py:594)."""
    cum, out = 0.0, []
    s = 0.0
    for p in probs:
        s += p
    total = 1.0 if (residual_to_one and s < 1.0) else s
    for p in probs:
        cum += p
        x = cum / total * 4294967296.0
        out.append(0xFFFFFFFF if x >= 4294967295.0 else int(x))
    return out


class ErrorModel:
    def __init__(self, name_or_path):
        if name_or_path == "random":           # py/tksm_badread.py:80-83
            self.type, self.k, self.max_alts = 0, 1, 1
            self.cdf = np.zeros((4, 1), np.uint32)
            self.alts = np.zeros((4, 1), np.uint64)
            self.nalts = np.zeros(4, np.uint8)
        else:
            self._load(name_or_path)
        self._c = _ErrModel(self.type, self.k, self.max_alts, 0, self.cdf.ctypes.data, self.alts.ctypes.data,
                            self.nalts.ctypes.data)

    def _load(self, path):
        rows = []
        k = None
        with _open(path) as f:
            for line in f:
                kmer = line.split(",", 1)[0]
                if k is None:
                    k = len(kmer)
                assert k == len(kmer)
                alternatives = [x.split(",") for x in line.strip().split(";") if x]
                assert alternatives[0][0] == kmer
                slots = [pack_alt(align_kmers(kmer, a[0]), a[0] == kmer) for a in alternatives]
                probs = [float(a[1]) for a in alternatives]
                rows.append((kmer, slots, cdf_thresholds(probs, True)))
        A = max(len(r[1]) for r in rows)
        n = 4 ** k
        self.type, self.k, self.max_alts = 1, k, A
        self.cdf = np.zeros((n, A), np.uint32)
        self.alts = np.zeros((n, A), np.uint64)
        self.nalts = np.zeros(n, np.uint8)
        for kmer, slots, thr in rows:
            idx = 0
            for ch in kmer:
                idx = idx * 4 + _CODE[ch]
            self.nalts[idx] = len(slots)
            self.alts[idx, : len(slots)] = np.array(slots, np.uint64)
            self.cdf[idx, : len(thr)] = np.array(thr, np.uint32)
            self.cdf[idx, len(thr):] = thr[-1]


_OP = {"=": 0, "X": 1, "I": 2, "D": 3}


def encode_cigar_key(cigar):
    if len(cigar) > 29:
        return None
    v = 0
    for i, ch in enumerate(cigar):
        v |= _OP[ch] << (2 * i)
    return v | (len(cigar) << 58)


class QScoreModel:
    def __init__(self, name_or_path):
        scores, probs = {}, {}
        self.kmer_size = 1
        if name_or_path == "random":           # py/tksm_badread.py:487-497
            for c in "=XI":
                scores[c], probs[c] = list(range(1, 21)), [1 / 20] * 20
        elif name_or_path == "ideal":          # py/tksm_badread.py:499-544
            self.kmer_size = 9
            for key, (lo, hi) in {"X": (1, 3), "I": (1, 3), "=": (4, 7), "===": (8, 20), "=====": (21, 30),
                                  "=======": (31, 40), "=========": (41, 50)}.items():
                cnt = hi - lo + 1
                scores[key], probs[key] = list(range(lo, hi + 1)), [1 / cnt] * cnt
        else:
            with _open(name_or_path) as f:     # py/tksm_badread.py:546-582
                for line in f:
                    parts = line.strip().split(";")
                    if parts[0] == "overall":
                        continue
                    cigar = parts[0]
                    self.kmer_size = max(self.kmer_size, len(cigar.replace("D", "")))
                    sp = [x.split(":") for x in parts[2].split(",") if x]
                    scores[cigar] = [int(x[0]) for x in sp]
                    probs[cigar] = [float(x[1]) for x in sp]
        assert "=" in scores and "X" in scores and "I" in scores
        self.scores, self.probs = scores, probs
        n_slots = 1
        while n_slots < 2 * len(scores):
            n_slots *= 2
        self.keys = np.zeros(n_slots, np.uint64)
        self.row_off = np.zeros(n_slots, np.uint32)
        self.row_cnt = np.zeros(n_slots, np.uint32)
        cdf_pool, q_pool = [], []
        for cigar in scores:                   # file order
            key = encode_cigar_key(cigar)
            assert key is not None, "q-score key longer than 29 ops"
            s = _lib.oracle_qs_hash(key) & (n_slots - 1)
            while self.keys[s] != 0:
                assert self.keys[s] != key
                s = (s + 1) & (n_slots - 1)
            self.keys[s] = key
            self.row_off[s] = len(q_pool)
            self.row_cnt[s] = len(scores[cigar])
            cdf_pool += cdf_thresholds(probs[cigar], False)
            q_pool += scores[cigar]
        self.cdf_pool = np.array(cdf_pool, np.uint32)
        self.q_pool = np.array(q_pool, np.uint8)
        self._c = _QsModel(n_slots, self.kmer_size, self.keys.ctypes.data, self.row_off.ctypes.data,
                           self.row_cnt.ctypes.data, self.cdf_pool.ctypes.data, self.q_pool.ctypes.data)


def beta_parameters(beta_mean, beta_stdev, beta_max):   # py/tksm_badread.py:747-757
    u, s, m = beta_mean, beta_stdev, beta_max
    beta_a = (((1 - (u / m)) / ((s / m) ** 2)) - (m / u)) * ((u / m) ** 2)
    beta_b = beta_a * ((m / u) - 1)
    if beta_a < 0.0 or beta_b < 0.0:
        raise SystemExit("Error: invalid beta parameters for identity distribution - trying increasing "
                         "the maximum identity or reducing the standard deviation")
    return beta_a, beta_b


class Identities:                               # py/tksm_badread.py:703-745
    def __init__(self, mean, stdev, max_identity, qtab=None):
        self.mean, self.stdev, self.max_identity = mean / 100.0, stdev / 100.0, max_identity / 100.0
        self.beta_a = self.beta_b = None
        if self.mean == self.max_identity:
            self.constant = True
        elif self.stdev == 0.0:
            self.max_identity = self.mean
            self.constant = True
        else:
            self.constant = False
            self.beta_a, self.beta_b = beta_parameters(mean, stdev, max_identity)
        if self.constant:
            self.qtab = np.zeros(1)
            self._c = _IdentModel(1, 0, self.mean, self.qtab.ctypes.data)
        else:
            if qtab is None:
                from scipy.stats import beta
                qtab = beta.ppf(np.arange(65537) / 65536.0, self.beta_a, self.beta_b)
            self.qtab = np.ascontiguousarray(qtab, np.float64)
            assert self.qtab.shape == (65537,)
            self._c = _IdentModel(0, 0, self.max_identity, self.qtab.ctypes.data)

    def get_identity(self, seed, read):
        return _lib.oracle_target_identity(C.byref(self._c), seed, read)


# ----------------------------------------------------------------------------- per-read path
def sequence_fragment(raw, target_identity, error_model, qscore_model, compute_qscores, seed, read,
                      use_full=False):
    """SIMULATE_PY.sequence_fragment (py/tksm_badread.py:324-451) -> (seq, qual, identity, stats)."""
    raw = bytes(raw)
    cap = (len(raw) + 2 * error_model.k) * 6 + 32
    seq, qual = C.create_string_buffer(cap), C.create_string_buffer(cap)
    n, ident, st = C.c_int(0), C.c_double(0), FragStats()
    rc = _lib.oracle_sequence_fragment(raw, len(raw), target_identity, C.byref(error_model._c),
                                       C.byref(qscore_model._c) if qscore_model is not None else None,
                                       1 if compute_qscores else 0, seed, read, 1 if use_full else 0,
                                       seq, qual, C.byref(n), C.byref(ident), C.byref(st))
    assert rc == 0
    return seq.raw[: n.value], qual.raw[: n.value], ident.value, st


def format_record(fastq, seed, read, seq, qual, error_free_len, identity, molecule_id):
    mid = molecule_id.encode() if isinstance(molecule_id, str) else molecule_id
    buf = C.create_string_buffer(2 * len(seq) + len(mid) + 256)
    n = _lib.oracle_format_record(buf, 1 if fastq else 0, seed, read, seq, qual, len(seq), error_free_len,
                                  identity, mid, len(mid))
    return buf.raw[:n]


def perfect_record(fastq, seed, read, seq, molecule_id):      # py/sequence.py:261-270
    return format_record(fastq, seed, read, seq, b"K" * len(seq), len(seq), 1.0, molecule_id)


class TailModel:
    """KDE_noise_generator.load (py/tksm_badread.py:944-962): JSON[.gz] {lx, ly, grid, begin, trans, ratio, bases}."""

    def __init__(self, path_or_dict):
        dc = path_or_dict
        if not isinstance(dc, dict):
            with _open(path_or_dict) as f:
                dc = json.load(f)
        self.lx = np.ascontiguousarray(dc["lx"], dtype=np.float64)
        self.ly = np.ascontiguousarray(dc["ly"], dtype=np.float64)
        self.grid = np.ascontiguousarray(dc["grid"], dtype=np.float64)
        assert self.grid.shape == (len(self.ly), len(self.lx))
        self.trans = np.ascontiguousarray(dc["trans"], dtype=np.float64)
        assert self.trans.shape == (4, 4)
        self.ratio = float(dc["ratio"])
        self.bases = "".join(dc["bases"]).encode()
        assert len(self.bases) == 4
        self._c = _TailModel(len(self.lx), len(self.ly), self.lx.ctypes.data, self.ly.ctypes.data, self.grid.ctypes.data,
                             (C.c_double * 16)(*self.trans.ravel()), self.ratio, (C.c_uint8 * 4)(*self.bases),
                             (C.c_uint8 * 4)())

    def length(self, frag_len, seed, read):
        return _lib.oracle_tail_length(C.byref(self._c), frag_len, seed, read)

    def noise_seq(self, frag_len, seed, read):
        n = self.length(frag_len, seed, read)
        buf = C.create_string_buffer(n + 8)
        got = _lib.oracle_tail_noise(C.byref(self._c), frag_len, seed, read, buf, n)
        assert got == n
        return buf.raw[:n]


def badread_record(fastq, seed, read, raw, identities, error_model, qscore_model, compute_qual, molecule_id,
                   use_full=False, tail_model=None):         # py/sequence.py:242-258
    target = identities.get_identity(seed, read)
    frag = bytes(raw) + (tail_model.noise_seq(len(raw), seed, read) if tail_model is not None else b"")
    seq, qual, ident, st = sequence_fragment(frag, target, error_model, qscore_model, compute_qual, seed, read,
                                             use_full)
    return format_record(fastq, seed, read, seq, qual, len(raw), ident, molecule_id), st
