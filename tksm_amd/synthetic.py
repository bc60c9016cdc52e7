"""Seeded synthetic inputs for the benchmark configurations of BASELINE.json.

GRCh38 / GENCODE are not available offline, so the genome is i.i.d. uniform ACGT and the molecules
are transcript-like multi-interval records:
  lengths  ~ round(Normal(mean, sd)) clipped to >= 200           ("1 kb mean")
  bulk     S ~ U{1..4} intervals on random contigs, one strand per molecule, 10 % of intervals
           carry one substitution
  scrna    + 16-base barcode literal + 10-base UMI literal + polyA literal ~ Normal(15, 7.5)
           clipped to [0, 5000] (README.md:150-157 pattern; src/scb.cpp:73-82, src/polyA.cpp:133-148)
  pcr      bulk molecules as they come out of 20 PCR cycles (BASELINE config 5): every interval carries
           Poisson(20 x 2.67e-4 x length) substitutions (Taq-setting1 2e-4 x 4/3 per cycle, src/pcr.cpp:36, :138) at uniform
           positions, one in twenty forced onto the interval's first or last base, on either strand
Arrays are produced directly in the binary batch layout of include/tksmseq.h.
"""
import numpy as np


def make_molecules(rs, contig_lens, n, mean_len=1000, sd_len=200, kind="bulk", id_prefix="mol", lognormal_sigma=None,
                   polya_mean=15.0):
    """Returns dict(reads, intervals, mods, literals, literal_pool, ids, id_pool, raw_len) as numpy arrays.
    lognormal_sigma: transcript-like skewed lengths, median mean_len, clipped to [200, 16000] (instead of the normal)."""
    contig_lens = np.asarray(contig_lens, np.int64)
    if lognormal_sigma:
        length = np.clip(np.rint(mean_len * np.exp(rs.normal(0.0, lognormal_sigma, n))), 200, 16000).astype(np.int64)
    else:
        length = np.maximum(200, np.rint(rs.normal(mean_len, sd_len, n))).astype(np.int64)
    S = rs.randint(1, 5, n)
    minus = rs.randint(0, 2, n).astype(np.uint32)
    n_gen = int(S.sum())
    mol_of = np.repeat(np.arange(n), S)
    first = np.concatenate([[0], np.cumsum(S)[:-1]])
    j_in = np.arange(n_gen) - first[mol_of]
    base = length[mol_of] // S[mol_of]
    ilen = base + (j_in == S[mol_of] - 1) * (length[mol_of] - base * S[mol_of])
    contig = rs.randint(0, len(contig_lens), n_gen)
    start = (rs.random_sample(n_gen) * (contig_lens[contig] - ilen)).astype(np.int64)
    if kind == "pcr":
        n_mod_iv = rs.poisson(20 * 2.67e-4 * ilen)
        iv_of = np.repeat(np.arange(n_gen), n_mod_iv)
        mod_pos = (rs.random_sample(len(iv_of)) * ilen[iv_of]).astype(np.int64)
        edge = rs.random_sample(len(iv_of))
        mod_pos = np.where(edge < 0.025, 0, np.where(edge < 0.05, ilen[iv_of] - 1, mod_pos))
        mod_chr = np.frombuffer(b"ACTG", np.uint8)[rs.randint(0, 4, len(iv_of))]
        has_mod = n_mod_iv                                     # substitutions per interval
    else:
        has_mod = rs.random_sample(n_gen) < 0.10
        mod_pos = (rs.random_sample(n_gen) * ilen).astype(np.int64)[has_mod]
        mod_chr = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, int(has_mod.sum()))]

    lit_per_mol = 0
    literals, pool = np.zeros((0, 2), np.uint64), b""
    if kind == "scrna":
        lit_per_mol = 3
        # barcode whitelist of 4096 16-mers, UMIs unique per molecule, polyA of variable length
        wl = rs.choice(np.frombuffer(b"ACGT", np.uint8), (4096, 16))
        umi = rs.choice(np.frombuffer(b"ACGT", np.uint8), (n, 10))
        pa_len = np.clip(np.rint(rs.normal(polya_mean, polya_mean / 2.0, n)), 0, 5000).astype(np.int64)
        max_pa = int(pa_len.max()) if n else 0
        pool_arr = np.concatenate([wl.reshape(-1), umi.reshape(-1), np.full(max_pa, ord("A"), np.uint8)])
        pool = pool_arr.tobytes()
        lit = [(i * 16, 16) for i in range(4096)] + [(4096 * 16 + i * 10, 10) for i in range(n)] + [(4096 * 16 + n * 10, max_pa)]
        literals = np.array(lit, np.uint64)
        bc_of = rs.randint(0, 4096, n)
    tot_iv = n_gen + lit_per_mol * n
    intervals = np.zeros((tot_iv, 4), np.uint32)
    ivl_count = S + lit_per_mol
    ivl_begin = np.concatenate([[0], np.cumsum(ivl_count)[:-1]])
    gpos = ivl_begin[mol_of] + j_in
    mods_per_iv = np.zeros(tot_iv, np.int64)
    mods_per_iv[gpos] = has_mod
    mod_begin = np.concatenate([[0], np.cumsum(mods_per_iv)[:-1]])
    intervals[gpos, 0] = contig
    intervals[gpos, 1] = start
    intervals[gpos, 2] = start + ilen
    intervals[:, 3] = mod_begin
    intervals[gpos, 3] |= (minus[mol_of] << 31)
    if kind == "scrna":
        lp = ivl_begin + S
        intervals[lp, 0] = 0x80000000 | bc_of.astype(np.uint32)
        intervals[lp, 2] = 16
        intervals[lp + 1, 0] = 0x80000000 | (4096 + np.arange(n)).astype(np.uint32)
        intervals[lp + 1, 2] = 10
        intervals[lp + 2, 0] = 0x80000000 | np.uint32(4096 + n)
        intervals[lp + 2, 2] = pa_len
        length = length + 26 + pa_len
    mods = np.stack([mod_pos, mod_chr.astype(np.int64)], 1).astype(np.uint32) if len(mod_pos) else np.zeros((0, 2), np.uint32)
    reads = np.stack([ivl_begin, ivl_count], 1).astype(np.uint32)
    id_strs = [f"{id_prefix}_{i}".encode() for i in range(n)]
    id_len = np.array([len(s) for s in id_strs], np.uint32)
    id_off = np.concatenate([[0], np.cumsum(id_len)[:-1]]).astype(np.uint32)
    return dict(reads=reads, intervals=intervals, mods=mods, literals=literals, literal_pool=pool,
                ids=np.stack([id_off, id_len], 1).astype(np.uint32), id_pool=b"".join(id_strs), raw_len=length,
                n_intervals_per_read=ivl_count, n_mods=len(mods))


def take(m, sel):
    """The molecules `sel` (indices, in that order) of a make_molecules() set as a set of their own: intervals and substitutions
    regathered so that every interval's substitutions stay one contiguous run (the batch layout's rule); id and literal pools are
    shared with the source.  What a rank's round-robin shard of a common set is built with."""
    sel = np.asarray(sel, np.int64)
    reads = m["reads"][sel]
    cnt, beg = reads[:, 1].astype(np.int64), reads[:, 0].astype(np.int64)
    n_iv = int(cnt.sum())
    new_begin = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64) if len(sel) else np.zeros(0, np.int64)
    iv_idx = np.repeat(beg - new_begin, cnt) + np.arange(n_iv)
    iv = m["intervals"][iv_idx].copy()
    mb_all = (m["intervals"][:, 3] & 0x7fffffff).astype(np.int64)
    me_all = np.concatenate([mb_all[1:], [len(m["mods"])]])
    mcnt = (me_all - mb_all)[iv_idx]
    new_mb = np.concatenate([[0], np.cumsum(mcnt)[:-1]]).astype(np.int64) if n_iv else np.zeros(0, np.int64)
    mod_idx = np.repeat(mb_all[iv_idx] - new_mb, mcnt) + np.arange(int(mcnt.sum()))
    mods = m["mods"][mod_idx] if len(m["mods"]) else m["mods"]
    iv[:, 3] = (iv[:, 3] & np.uint32(0x80000000)) | new_mb.astype(np.uint32)
    out = dict(m)
    out.update(reads=np.stack([new_begin, cnt], 1).astype(np.uint32), intervals=iv, mods=mods, ids=m["ids"][sel], raw_len=m["raw_len"][sel],
               n_intervals_per_read=cnt, n_mods=len(mods))
    return out


def algorithmic_bytes(m, records_bytes):
    """B = B_mdf + B_ref + B_out per launch.
    B_mdf = 8 + 16 S + 8 M per read (+ literal bytes / 4), B_ref = ceil(L / 4), B_out = the emitted record bytes."""
    b_mdf = 8 * len(m["reads"]) + 16 * len(m["intervals"]) + 8 * len(m["mods"])
    b_ref = int(np.sum((m["raw_len"] + 3) // 4))
    return b_mdf + b_ref + int(records_bytes)


def mdf_text(m, contig_names, literal_strings=None):
    """The same molecules as MDF text (for the CPU oracle / text-path tests).  Small inputs only."""
    out = []
    ids, pool = m["ids"], m["id_pool"]
    lits = m["literals"]
    lp = m["literal_pool"]
    nm = len(m["mods"])
    for r, (b, c) in enumerate(m["reads"]):
        out.append(f"+{pool[ids[r, 0]:ids[r, 0] + ids[r, 1]].decode()}\t1\t\n")
        for i in range(b, b + c):
            cg, st, en, mi = (int(x) for x in m["intervals"][i])
            mb = mi & 0x7fffffff
            me = int(m["intervals"][i + 1][3]) & 0x7fffffff if i + 1 < len(m["intervals"]) else nm
            name = contig_names[cg] if not cg >> 31 else lp[int(lits[cg & 0x7fffffff, 0]):int(lits[cg & 0x7fffffff, 0]) + int(lits[cg & 0x7fffffff, 1])].decode()
            mods = ",".join(f"{int(p)}{chr(int(ch))}" for p, ch in m["mods"][mb:me])
            out.append(f"{name}\t{st}\t{en}\t{'-' if mi >> 31 else '+'}\t{mods}\n")
    return "".join(out)
