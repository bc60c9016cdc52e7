"""bench.py's N > 1 entry: `--gpus N` starts the ranks itself (child processes; the launcher never touches a GPU), refuses a
WORLD_SIZE that disagrees, and ends every N > 1 run (or the forced single-rank exchange) with the FASTQ order check of BASELINE
config 4 -- the ordered stream of the ranks' round-robin shards against one GPU's run of the whole set.  Molecules are independent
in the reference too (py/sequence.py:360-368, Pool.imap_unordered over molecules); ordering is what this build adds."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _bench(*args, env=None, timeout=900):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, env=e, timeout=timeout)


def test_gpus_flag_is_checked_before_anything_runs():
    """more ranks than devices: the launcher refuses (exit 2) without starting a rank; a WORLD_SIZE that is not --gpus: refused"""
    import torch
    have = torch.cuda.device_count()
    r = _bench("--gpus", str(max(2, have + 1)), "--no-cpu-baseline")
    assert r.returncode == 2 and f"this node shows {have} GPU(s)" in r.stderr and r.stdout.strip() == ""
    r = _bench("--gpus", "1", "--no-cpu-baseline", env={"WORLD_SIZE": "2"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_launcher_starts_one_rank_per_gpu():
    """the launcher's command line (torch.distributed.run, rendezvous on 127.0.0.1): without GPUs both ranks start, see their
    RANK / WORLD_SIZE, and stop at the no-CPU-fallback check -- the launcher passes the failure on"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU tests start real ranks")
    r = _bench("--gpus", "2", "--no-cpu-baseline", env={"TKSM_BENCH_SKIP_DEVICE_CHECK": "1"}, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    # (the launcher stops the other rank as soon as the first one has failed: both must have STARTED, at least one reaches the check)
    assert "[bench] rank 0 of 2 started" in r.stderr and "[bench] rank 1 of 2 started" in r.stderr, r.stderr[-1500:]
    assert r.stderr.count("bench.py needs a GPU: the Seq hot path has no CPU fallback") >= 1, r.stderr[-1500:]


@pytest.mark.parametrize("kind", ["bulk", "pcr", "scrna"])
def test_take_rebuilds_a_shard_of_a_molecule_set(kind):
    """synthetic.take(): the shard a rank runs in the order check is the same molecules, text for text"""
    from tksm_amd import synthetic
    rs = np.random.RandomState(1)
    m = synthetic.make_molecules(rs, [100000] * 3, 50, 300, 50, kind=kind)
    names = ["a", "b", "c"]
    full = synthetic.mdf_text(m, names)
    mols = ["+" + x for x in ("\n" + full).split("\n+")[1:]]
    mols = [x if x.endswith("\n") else x + "\n" for x in mols]
    for sel in (np.arange(1, 50, 3), np.arange(50), np.array([49, 0, 7]), np.zeros(0, np.int64)):
        assert synthetic.mdf_text(synthetic.take(m, sel), names) == "".join(mols[i] for i in sel)


SMALL = ["--batch", "49152", "--steps", "4", "--warmup", "1", "--genome-contigs", "4", "--contig-mb", "4", "--no-cpu-baseline", "--no-e2e",
         "--no-side-legs", "--order-check-reads", "20000"]


@pytest.mark.gpu
@pytest.mark.parametrize("ordering", ["gather", "offsets"])
def test_exchange_path_on_one_rank_orders_like_the_plain_run(ordering):
    """the N > 1 code of bench.py (RCCL process group, exchange thread and stream, two output buffers per context, device
    interleave / offsets scan) forced on ONE rank: every step goes through it and the order check compares its stream with the plain run"""
    r = _bench("--gpus", "1", "--ordering", ordering, *SMALL, env={"TKSM_BENCH_FORCE_EXCHANGE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["ordering"] == ordering
    oc = line["order_check"]
    assert oc["equal"] is True and oc["reads"] == 20000 and oc["bytes"] > 20000 * 1500


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,ordering", [(2, "gather"), (3, "gather"), (2, "offsets")])
def test_n_rank_logic_rehearsed_on_one_gpu(ranks, ordering):
    """`python bench.py --gpus N` with TKSM_BENCH_REHEARSE=gloo: N real ranks (launcher, torch.distributed.run, RANK / WORLD_SIZE), all on
    GPU 0, the collectives over gloo on host copies -- RCCL refuses two ranks on one device, so this is what a one-GPU box can run of an
    N-rank job: round-robin shards with stride N through the kernels, the exchange thread and its buffers, the device interleave of N
    streams, max-over-ranks timing, and the FASTQ order check against one rank's run of the whole set."""
    r = _bench("--gpus", str(ranks), "--ordering", ordering, *SMALL, env={"TKSM_BENCH_REHEARSE": "gloo"}, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == ranks and line["config"]["sharding"] == f"round-robin x{ranks}" and "rehearsal" in line["config"]
    oc = line["order_check"]
    assert oc["equal"] is True and oc["reads"] == 20000 * ranks and oc["bytes"] > 20000 * ranks * 1500


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """`python bench.py --gpus 2` with no launcher around it: two ranks over RCCL, n_gpus == 2, order check green"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    r = _bench("--gpus", "2", *SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["order_check"]["equal"] is True and line["order_check"]["reads"] == 40000
