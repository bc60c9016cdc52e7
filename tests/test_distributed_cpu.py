"""CPU tests (gloo, world_size 2) of the N>1 path: round-robin sharding of reads with a counter-based RNG keyed by
the global read index makes every rank's records a pure function of (seed, global index), so the interleave of the
per-rank streams equals the single-process output.  The record bytes come from the oracle here (no GPU); the
exchange step is the same gather + interleave bench.py runs over RCCL."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, ERR_MODEL, QS_MODEL


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
    import pyoracle as po
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    models = (po.ErrorModel("random"), po.QScoreModel("random"))
    recs = _records(po, rank, world, n_total, models)
    stream = torch.tensor(list(b"".join(recs)), dtype=torch.uint8)
    offs = torch.tensor(np.concatenate([[0], np.cumsum([len(r) for r in recs])]), dtype=torch.int64)
    sizes = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([stream.numel(), offs.numel()], dtype=torch.int64))
    mx = max(int(s[0]) for s in sizes)
    mo = max(int(s[1]) for s in sizes)
    pad = torch.zeros(mx, dtype=torch.uint8)
    pad[: stream.numel()] = stream
    opad = torch.zeros(mo, dtype=torch.int64)
    opad[: offs.numel()] = offs
    gs = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    go = [torch.zeros(mo, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, gs, dst=0)
    dist.gather(opad, go, dst=0)
    if rank == 0:
        merged = []
        for g in range(n_total):
            p, i = g % world, g // world
            merged.append(bytes(gs[p][int(go[p][i]):int(go[p][i + 1])].tolist()))
        q.put(b"".join(merged))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_shards_interleave_to_single_process_output(po):
    n_total, world = 23, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    models = (po.ErrorModel("random"), po.QScoreModel("random"))
    single = b"".join(_records(po, 0, 1, n_total, models))
    assert merged == single
