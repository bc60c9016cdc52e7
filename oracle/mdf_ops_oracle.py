"""CPU oracle of the molecule-description transforms upstream of Seq: PCR and truncation.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under tksm_amd/).  A restatement in plain Python of
    stream_mdf(unroll=True) / operator<< / comment() / dump_comment()   src/mdf.h:64-110, src/interval.h:809-830, :880-905
    PCR::do_pcr / perform                                                src/pcr.cpp:40-89
    molecule_descriptor::add_error                                       src/interval.h:866-874
    truncate() / einterval::truncate / flip_molecule                     src/truncate.cpp:23-65, src/interval.h:708-735, :908-920
    custom_distribution / custom_distribution2D                          src/truncate.cpp:77-203
    truncate_transformer / truncate_transformer_kde                      src/truncate.cpp:322-351
The reference draws from one sequential std::mt19937 (no test pins its stream, and the C++ cannot be built here: cxxopts / fmt /
nlohmann_json are absent), so two things live here:
  * `*_reference`: the reference's algorithm line by line with numpy's generator standing in for mt19937 -- full tree
    enumeration for PCR -- used to check DISTRIBUTIONS (parity unpinned beyond that: see DESIGN.md);
  * `*_spec`: the same algorithms with the build's counter-based RNG (Philox keyed by molecule / copy path / purpose), which
    the HIP kernels reproduce bit for bit.  truncate() itself is pinned by the reference's unit-test vectors
    (test/truncate_test.cpp:12-55, tests/test_mdf_ops.py).
"""
import bisect
import copy
import math

import pyoracle as po

ST_PCR_PICK, ST_PCR_EMIT, ST_PCR_CHILD, ST_PCR_MUT, ST_TRC_LEN, ST_TRC_SIDE = 16, 17, 18, 19, 24, 25
PCR_MAX_MUT = 32
PRESETS = {"Taq-setting1": (2 * math.pow(0.1, 4), 0.88), "Taq-setting2": (7.2 * math.pow(0.1, 5), 0.36),
           "Klenow": (1.3 * math.pow(0.1, 4), 0.80), "T7": (3.4 * math.pow(0.1, 5), 0.90), "T4": (3.0 * math.pow(0.1, 6), 0.56),
           "Vent": (4.5 * math.pow(0.1, 5), 0.70)}                      # src/pcr.cpp:136-140


# ------------------------------------------------------------------------------------------------ MDF model
def parse_comment(comment):                                            # molecule_descriptor::comment, src/interval.h:809-830
    meta = {}
    for f in [x for x in comment.split(";") if x]:
        if "=" not in f:
            meta.setdefault(f, []).append(".")
        else:
            kv = [x for x in f.split("=") if x]
            for v in [x for x in (kv[1] if len(kv) > 1 else "").split(",") if x]:
                meta.setdefault(kv[0], []).append(v)
    return meta


def dump_comment(meta):                                                # src/interval.h:880-890 (std::map: keys sorted)
    out = []
    for k in sorted(meta):
        out.append(k + ("=" + ",".join(meta[k]) if meta[k][0] != "." else "") + ";")
    return "".join(out)


def stream_mdf(text, unroll=True):                                     # src/mdf.h:64-110
    mols, cur = [], None
    for line in text.splitlines():
        f = line.split("\t")
        if line.startswith("+"):
            cur = dict(id=f[0][1:], depth=int(f[1]), meta=parse_comment(f[2] if len(f) > 2 else ""), segments=[])
            mols.append(cur)
        else:
            errs = []
            for m in (f[4] if len(f) > 4 else "").split(","):          # parse_and_add_errors, src/interval.h:737-747
                if m:
                    errs.append((int(m[:-1]), m[-1]))
            cur["segments"].append(dict(chr=f[0], start=int(f[1]), end=int(f[2]), plus=f[3] == "+", errors=errs))
    out = []
    for md in mols:
        if unroll and md["depth"] > 1:
            for i in range(md["depth"]):
                c = copy.deepcopy(md)
                c["depth"], c["id"] = 1, f"{md['id']}_{i}"
                out.append(c)
        else:
            out.append(md)
    return out


def seg_size(s):
    return max(0, s["end"] - s["start"])


def mol_size(md):
    return sum(seg_size(s) for s in md["segments"])


def write_mdf(mols):                                                   # operator<<, src/interval.h:898-905
    out = []
    for md in mols:
        out.append(f"+{md['id']}\t{md['depth']}\t{dump_comment(md['meta'])}\n")
        for s in md["segments"]:
            out.append(f"{s['chr']}\t{s['start']}\t{s['end']}\t{'+' if s['plus'] else '-'}\t" +
                       ",".join(f"{p}{b}" for p, b in s["errors"]) + "\n")
    return "".join(out)


def add_error(md, pos, base):                                          # src/interval.h:866-874
    it = 0
    while seg_size(md["segments"][it]) <= pos:
        pos -= seg_size(md["segments"][it])
        it += 1
    md["segments"][it]["errors"].append((pos, base))


# ------------------------------------------------------------------------------------------------ PCR
def pcr_reference(mols, cycles, efficiency, error_rate, target, rs):
    """PCR::perform / do_pcr (src/pcr.cpp:40-89) with numpy's generator `rs` in place of mt19937: walks the whole tree."""
    rate = (4 * error_rate) / 3
    bases = "ACTG"
    expected_after = math.pow(1 + efficiency, cycles) * sum(m["depth"] for m in mols)
    drop = target / expected_after
    out = []

    def do_pcr(md, step, positions):
        if rs.random_sample() > efficiency:
            return
        expected = rate * len(positions)
        count = int(expected)
        count += rs.random_sample() < (expected - count)
        mpos = sorted(rs.choice(len(positions), count, replace=False).tolist()) if count else []    # std::sample keeps the order
        mdc = copy.deepcopy(md)
        for p in mpos:
            add_error(mdc, positions[p], bases[rs.randint(0, 4)])
        mdc["id"] = md["id"] + "." + str(step)
        if rs.random_sample() < drop:
            out.append(mdc)
        for cycle in range(step + 1, cycles):
            do_pcr(mdc, cycle, positions)

    for m in mols:
        positions = list(range(mol_size(m)))
        for cycle in range(cycles):
            do_pcr(m, cycle, positions)
    return out


def _u01(x):
    return x * (1.0 / 4294967296.0)


def _philox_node(seed, u, mask, stream, n):
    return po.philox(seed, u | ((mask & 0xffffffff) << 32), stream | ((mask >> 32) << 8), n)


def pcr_tables(cycles, efficiency, drop):
    """q[t] = P(nothing is written in the subtree of an existing copy made in cycle t); A[t] = P(none of the copies made in
    cycles t.. from one template leads to a written copy)."""
    q, A = [1.0] * (cycles + 1), [1.0] * (cycles + 2)
    for t in range(cycles - 1, -1, -1):
        q[t] = (1.0 - drop) * A[t + 1]
        A[t] = A[t + 1] * (1.0 - efficiency * (1.0 - q[t]))
    return q, A


def pcr_mutations(seed, u, mask, rate, size):
    expected = rate * float(size)
    cnt = int(expected)
    cnt += 1 if _u01(_philox_node(seed, u, mask, ST_PCR_MUT, 0)[0]) < (expected - float(cnt)) else 0
    cnt = min(cnt, PCR_MAX_MUT, size)
    chosen, attempt = [], 1
    while len(chosen) < cnt:
        w = _philox_node(seed, u, mask, ST_PCR_MUT, attempt)
        attempt += 1
        p = (w[0] * size) >> 32
        if any(p == c[0] for c in chosen):
            continue
        chosen.append((p, "ACTG"[w[1] & 3]))
    return sorted(chosen, key=lambda c: c[0])


def pcr_spec(mols, cycles, efficiency, error_rate, target, seed, only=None):
    """The specification the HIP kernels implement: only the branches that lead to a written copy are walked (kernels:
    tksm_amd/csrc/mdf_kernels.hip, pcr_walk); written set, ancestry and substitutions have the reference's distribution.
    only = (lo, hi): the copies of the templates at positions lo <= i < hi of the processing order alone (the templates are
    independent given the drop ratio, so the tests compute a large output in slices, side by side).  Processing order: input order;
    with more than 2 x target templates (src/pcr.cpp:217-220: std::shuffle, then resize) the 2 x target templates with the smallest
    keys IN KEY ORDER -- a uniformly random ordered subset, as the reference's shuffle + cut gives."""
    n = len(mols)
    keep = list(range(n))
    if n > 2 * target:
        keys = sorted(((lambda w: (w[0] << 32) | w[1])(po.philox(seed, u, ST_PCR_PICK, 0)), u) for u in range(n))
        keep = [u for _, u in keys[: 2 * target]]
    rate = (4 * error_rate) / 3
    expected_after = math.pow(1 + efficiency, cycles) * float(len(keep))
    drop = min(1.0, target / expected_after) if expected_after > 0 else 0.0
    q, A = pcr_tables(cycles, efficiency, drop)
    out = []
    for pos, u in enumerate(keep):
        if only is not None and not only[0] <= pos < only[1]:
            continue
        md, size = mols[u], mol_size(mols[u])
        stack = [[0, 0, True]]                                          # mask, next cycle, satisfied
        while stack:
            R, t, sat = stack[-1]
            if t >= cycles:
                stack.pop()
                continue
            stack[-1][1] = t + 1
            pm = efficiency * (1.0 - q[t])
            p = pm if sat else pm / (1.0 - A[t])
            C = R | (1 << t)
            if not _u01(_philox_node(seed, u, C, ST_PCR_CHILD, 0)[0]) < p:
                continue
            stack[-1][2] = True
            emit = _u01(_philox_node(seed, u, C, ST_PCR_EMIT, 0)[0]) < drop / (1.0 - q[t])
            if emit:
                c = copy.deepcopy(md)
                steps = [s for s in range(cycles) if (C >> s) & 1]
                pre = 0
                for s in steps:                                        # substitutions of every copy event on the path, oldest first
                    pre |= 1 << s
                    for pos, base in pcr_mutations(seed, u, pre, rate, size):
                        add_error(c, pos, base)
                c["id"] = md["id"] + "".join("." + str(s) for s in steps)
                out.append(c)
            stack.append([C, t + 1, emit])
    return out


# ------------------------------------------------------------------------------------------------ truncation
def einterval_truncate(s, start, end):                                 # src/interval.h:708-735
    s["errors"] = sorted(s["errors"], key=lambda e: e[0])              # std::sort by position (stable here)
    s["start"] += start
    s["end"] = s["start"] + (end - start)
    s["errors"] = [(p - start, b) for p, b in s["errors"] if 0 <= p - start < end - start]


def truncate(md, post_truncation_length, min_val=100):                 # src/truncate.cpp:23-65
    post_truncation_length = int(post_truncation_length)               # the parameter is an int: doubles convert toward zero
    if post_truncation_length == mol_size(md):
        return
    if min_val > post_truncation_length:
        post_truncation_length = min_val
    i, kept = 0, 0
    segs = md["segments"]
    for g in segs:
        if kept + seg_size(g) >= post_truncation_length:
            break
        kept += seg_size(g)
        i += 1
    if i != len(segs):
        keep = post_truncation_length - kept
        if segs[i]["plus"]:
            ts, te = segs[i]["start"] + keep, segs[i]["end"]
            einterval_truncate(segs[i], 0, keep)
        else:
            ts, te = segs[i]["start"], segs[i]["end"] - keep
            einterval_truncate(segs[i], seg_size(segs[i]) - keep, seg_size(segs[i]))
        md["meta"].setdefault("truncated", []).append(f"{segs[i]['chr']}:{ts}-{te}")
        for j in range(i + 1, len(segs)):
            md["meta"].setdefault("truncated", []).append(f"{segs[j]['chr']}:{segs[j]['start']}-{segs[j]['end']}")
        del segs[i + 1:]


def flip_molecule(md):                                                 # src/interval.h:908-920
    f = dict(id=md["id"], depth=md["depth"], meta=md["meta"], segments=[])
    for s in reversed(md["segments"]):
        c = copy.deepcopy(s)
        c["plus"] = not c["plus"]
        f["segments"].append(c)
    return f


class CustomDistribution:                                              # src/truncate.cpp:77-146
    def __init__(self, pdf, bins, integral):
        self.pdf, self.bins, self.integral = list(pdf), list(bins), integral
        s = sum(self.pdf)
        self.cdf = [0.0]
        for d in self.pdf:
            self.cdf.append(d / s + self.cdf[-1])

    def bin_of(self, u):
        return max(0, min(bisect.bisect_left(self.cdf, u) - 1, len(self.bins) - 1))

    def draw(self, u, w):
        """operator()(g, u): w is the 32-bit word that drives the smoother of the chosen bin"""
        b = self.bin_of(u)
        lo, hi = (0 if b == 0 else self.bins[b - 1]), self.bins[b]
        if self.integral:
            return float(lo + ((w * (hi - lo + 1)) >> 32))             # uniform_int_distribution<long>(lo, hi)
        return lo + (hi - lo) * _u01(w)                                # uniform_real_distribution<double>(lo, hi)


class TruncationModel:                                                 # custom_distribution2D + end_mtx, src/truncate.cpp:148-203, :362-381
    def __init__(self, parts):
        kde = next(p for p in parts if p["name"] == "KDE_mtx")
        w, h = kde["shape"]
        self.x = [int(v) for v in kde["labels"][:w]]
        self.y = [int(v) for v in kde["labels"][w:w + h]]
        self.rows = [CustomDistribution(kde["data"][i * w: i * w + min(i + 1, w)], self.x, True) for i in range(h)]
        end = [p for p in parts if p["name"] == "end_mtx"]
        self.sider = CustomDistribution(end[0]["data"], end[0]["labels"][:len(end[0]["data"])], False) if end else None

    def row_of(self, size):
        lo = bisect.bisect_left(self.y, size)
        d = min(lo, len(self.y) - 1)
        if lo < len(self.y) and d > 0 and abs(self.y[d] - size) > abs(self.y[d - 1] - size):
            d -= 1
        return d


def fmt_double(v):
    """fmt's "{}" of a double: the shortest representation that round-trips, without a trailing ".0" """
    r = repr(float(v))
    return r[:-2] if r.endswith(".0") else r


def _to_int(v):
    return int(max(-2147483648.0, min(2147483647.0, v)))


def trc_spec(md, g, seed, normal=None, lognormal=None, model=None, always_end=False, models_length=False):
    """One molecule through truncate_transformer / truncate_transformer_kde (src/truncate.cpp:322-351) with the build's RNG
    streams; g = index of the molecule in the whole input."""
    md = copy.deepcopy(md)
    if model is None:
        w = po.philox(seed, g, ST_TRC_LEN, 0)
        u1, u2 = (w[0] + 1.0) * (1.0 / 4294967296.0), _u01(w[1])
        z = math.sqrt(-2.0 * math.log(u1)) * math.cos(6.283185307179586 * u2)
        mu, sigma = normal if normal is not None else lognormal
        v = mu + sigma * z
        if lognormal is not None:
            v = math.exp(v)
        truncate(md, _to_int(v))
        return md
    size = mol_size(md)
    w = po.philox(seed, g, ST_TRC_LEN, 0)
    d = model.row_of(size)
    u = _u01(w[0])
    val = model.rows[d].draw(u, w[1])
    if d + 1 < len(model.rows):
        val = (val + model.rows[d + 1].draw(u, w[2])) / 2.0
    tl = float(size) - val if models_length else val
    if always_end and model.sider is None:
        side = 1.0
    else:
        s = po.philox(seed, g, ST_TRC_SIDE, 0)
        side = model.sider.draw(_u01(s[0]), s[1])
    truncate(md, _to_int(float(mol_size(md)) - tl * side))
    rev = flip_molecule(md)
    truncate(rev, _to_int(float(mol_size(rev)) - tl * (1.0 - side)))
    md = flip_molecule(rev)
    md["meta"].setdefault("TR", []).append(f"{fmt_double(tl)},{side:.2f}")
    return md
