"""FASTQ ordering across ranks (N > 1): what bench.py runs at the end of every step over RCCL, and what tests/test_distributed_cpu.py
runs over gloo -- the same functions.

Molecules are sharded round-robin (global read g -> rank g mod P) and every record is a pure function of (seed, g), so the ordered
output is the interleave of the per-rank record streams.  Two ways to get there:

  gather   (north_star: "an RCCL gather over xGMI only for final FASTQ ordering") -- all ranks exchange their byte and record counts
           (one all_gather of two int64), then every rank sends EXACTLY its record bytes and its offsets to rank 0 (point-to-point
           send / recv: RCCL has no gatherv), which places record i of rank p at global position i * P + p on the device
           (tksmseq_interleave_records).  Rank 0 holds the sum of the streams once more, not P fixed-width buffers.
  offsets  no record byte moves: the ranks all_gather their per-read record LENGTHS (4 bytes per read), every rank scans the
           interleaved lengths and knows the final file offset of each of its records -- what the product CLI does across
           --devices (each worker pwrites its records at their place).
"""
import numpy as np
import torch
import torch.distributed as dist


def exchange_sizes(n_bytes, n_reads, world, device):
    """every rank's (record bytes, reads) -> int64 tensor [world, 2] on the host"""
    mine = torch.tensor([int(n_bytes), int(n_reads)], dtype=torch.int64, device=device)
    out = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(out, mine)
    return torch.stack(out).cpu()


def gather_exact(stream, offsets, sizes, rank, world, flat_bytes=None, flat_offs=None, dst=0):
    """`stream`: this rank's record bytes (uint8, exactly sizes[rank, 0] elements); `offsets`: int64 [sizes[rank, 1] + 1].
    On rank `dst` returns (flat_bytes, byte_starts, flat_offs, off_starts): rank p's records are flat_bytes[byte_starts[p] :
    byte_starts[p + 1]] and its offsets flat_offs[off_starts[p] : off_starts[p + 1]]; the buffers may be passed in (at least the
    summed sizes) so that a step does not allocate.  Other ranks return None."""
    nb = [int(sizes[p, 0]) for p in range(world)]
    no = [int(sizes[p, 1]) + 1 for p in range(world)]
    assert stream.numel() == nb[rank] and offsets.numel() == no[rank]
    if rank != dst:
        ops = []
        if nb[rank]:
            ops.append(dist.P2POp(dist.isend, stream, dst))
        ops.append(dist.P2POp(dist.isend, offsets, dst))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return None
    byte_starts = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
    off_starts = np.concatenate([[0], np.cumsum(no)]).astype(np.int64)
    if flat_bytes is None or flat_bytes.numel() < byte_starts[-1]:
        flat_bytes = torch.empty(int(byte_starts[-1]), dtype=torch.uint8, device=stream.device)
    if flat_offs is None or flat_offs.numel() < off_starts[-1]:
        flat_offs = torch.empty(int(off_starts[-1]), dtype=torch.int64, device=stream.device)
    ops = []
    for p in range(world):
        bs, os_ = flat_bytes[byte_starts[p]:byte_starts[p + 1]], flat_offs[off_starts[p]:off_starts[p + 1]]
        if p == dst:
            bs.copy_(stream)
            os_.copy_(offsets)
        else:
            if nb[p]:
                ops.append(dist.P2POp(dist.irecv, bs, p))
            ops.append(dist.P2POp(dist.irecv, os_, p))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return flat_bytes, byte_starts, flat_offs, off_starts


def global_offsets(lengths, n_reads_all, rank, world):
    """`lengths`: this rank's per-read record lengths (int64 [n]).  All ranks exchange their lengths (padded to the largest shard)
    and return the byte offset, in the ordered output, of each of this rank's records (int64 [n]) and the total size."""
    n_max = max(int(x) for x in n_reads_all)
    pad = torch.zeros(n_max, dtype=torch.int64, device=lengths.device)
    pad[: lengths.numel()] = lengths
    allp = [torch.zeros(n_max, dtype=torch.int64, device=lengths.device) for _ in range(world)]
    dist.all_gather(allp, pad)
    inter = torch.stack(allp, dim=1).reshape(-1)                 # position i * P + p (absent reads of the shorter shards: length 0)
    ends = torch.cumsum(inter, 0)
    mine = (ends - inter).reshape(n_max, world)[: lengths.numel(), rank]
    return mine, int(ends[-1])


def interleave_host(flat_bytes, byte_starts, flat_offs, off_starts, n_reads_all):
    """reference of the device interleave (tksmseq_interleave_records) on host arrays: record i of rank p -> position i * P + p"""
    world = len(n_reads_all)
    fb = flat_bytes.cpu().numpy() if isinstance(flat_bytes, torch.Tensor) else np.asarray(flat_bytes)
    fo = flat_offs.cpu().numpy() if isinstance(flat_offs, torch.Tensor) else np.asarray(flat_offs)
    out = []
    for g in range(int(sum(n_reads_all))):
        p, i = g % world, g // world
        o = fo[off_starts[p]:off_starts[p + 1]]
        out.append(fb[byte_starts[p] + o[i]: byte_starts[p] + o[i + 1]].tobytes())
    return b"".join(out)
