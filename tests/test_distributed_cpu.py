"""CPU tests (gloo, world sizes 2 and 3, unequal shards) of the N > 1 path: round-robin sharding of reads with a counter-based RNG
keyed by the global read index makes every rank's records a pure function of (seed, global index), so the interleave of the
per-rank streams equals the single-process output.  The record bytes come from the oracle here (no GPU); the exchange is the one
bench.py runs over RCCL -- the SAME helpers (tksm_amd/ordering.py): exact-size gather to rank 0 + interleave, and the
no-byte-moves variant (all_gather of record lengths -> every rank's final file offsets)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _records(po, rank, world, n_total, models):
    ident = po.Identities(84.0, 5.5, 99.0)
    out = []
    for g in range(rank, n_total, world):
        rs = np.random.RandomState(1000 + g)
        raw = bytes(rs.choice(list(b"ACGT"), 150 + (g % 7) * 10).tolist())
        rec, _ = po.badread_record(True, 21, g, raw, ident, models[0], models[1], True, f"mol{g}")
        out.append(rec)
    return out


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import pyoracle as po
    from tksm_amd import ordering
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    models = (po.ErrorModel("random"), po.QScoreModel("random"))
    recs = _records(po, rank, world, n_total, models)
    stream = torch.tensor(list(b"".join(recs)), dtype=torch.uint8)
    offs = torch.tensor(np.concatenate([[0], np.cumsum([len(r) for r in recs])]), dtype=torch.int64)
    sizes = ordering.exchange_sizes(stream.numel(), len(recs), world, "cpu")
    n_all = [int(sizes[p, 1]) for p in range(world)]
    assert n_all == [len(range(p, n_total, world)) for p in range(world)]
    # (a) gather of exactly the record bytes to rank 0, interleave there
    got = ordering.gather_exact(stream, offs, sizes, rank, world)
    merged = ordering.interleave_host(*got, n_all) if rank == 0 else None
    # (b) no record byte moves: every rank learns where its records go
    mine, total = ordering.global_offsets(offs[1:] - offs[:-1], n_all, rank, world)
    q.put((rank, merged, mine.tolist(), total, [len(r) for r in recs]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 23), (3, 23), (3, 2)])
def test_round_robin_shards_interleave_to_single_process_output(po, world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 131 + n_total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=180)
        res[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    models = (po.ErrorModel("random"), po.QScoreModel("random"))
    single_recs = _records(po, 0, 1, n_total, models)
    single = b"".join(single_recs)
    assert res[0][1] == single                                   # exact-size gather + interleave
    starts = np.concatenate([[0], np.cumsum([len(r) for r in single_recs])])
    for rank in range(world):                                    # offsets variant: record i of rank p sits at the offset of read i * P + p
        _, _, mine, total, lens = res[rank]
        assert total == len(single)
        assert mine == [int(starts[g]) for g in range(rank, n_total, world)]
        assert lens == [len(single_recs[g]) for g in range(rank, n_total, world)]
