#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Seq hot path (BASELINE.json: sequenced reads/s + Gbases/s, %HBM roofline).

A "step" = one pass of the hot path (splice -> Badread errors -> q-scores -> FASTQ records) over one batch of
synthetic molecules with inputs resident in HBM.  Steps are issued the way a streaming tool issues batches:
--pipeline (3) contexts per GPU, each on its own stream and host thread, take the steps in turn, so that the kernels of
consecutive batches fill each other's gaps (the instruction-bound error loop next to the memory-bound alignment, the
latency-bound last rounds of one batch underneath the bulk of the next).  Workload at N=1: BASELINE.json configs[1], "Bulk 10M molecules,
Badread error+qual model" -- synthetic 24 x 128 Mb genome (GRCh38 is not available offline), nanopore2020 error +
q-score models, identity 84,99,5.5, FASTQ with computed qualities; processed as 9 steps of --batch = 1,703,936 molecules
(15.3 M molecules, the default run; 133 GiB of HBM in use; sized so that rank 0 of an 8-GPU run also holds the gathered record
streams: 3 x 44 GB of contexts + 14 GB of second output buffers + 74 GB of gather and interleave buffers).

N>1 (torchrun, one rank per GPU): molecules are sharded round-robin (global read g -> rank g mod P, counter-based
RNG keyed by g), per-GPU batch fixed (weak scaling); every step ends with the RCCL gather of the per-rank record
streams to rank 0 and the device-side interleave into global read order (the FASTQ-order exchange step).

Prints ONE JSON line on rank 0 (contract in the round instructions), including
  `roofline`      dominant kernel = the larger of k_loop (error loop, one lane per read; with k_loopw, its wave-per-read form for the late rounds) and k_alnf (alignment windows
                  decoded and aligned bit-parallel, one lane per alignment; all its instantiations: 14-row pass, full-width redo and small rounds, q-score round) by EXCLUSIVE time: after the timed steps one
                  more step runs on one context alone (nothing else on the GPU) with HIP events around every launch on its stream;
                  achieved = algorithmic bytes of a step / that kernel's summed launch durations in that step.  The overlapped
                  sums measured during the timed steps (three contexts sharing the GPU) are reported next to it.
                  `traffic` is the PMC figure of the committed profile named in `traffic_source` (same command), or null;
  `cpu_baseline`  the CPU oracle (oracle/tksm_oracle.c, -O3 -march=native, kind "port") on a bounded sample of the same workload,
                  all host cores;
  `e2e_reads_per_s`  (N = 1) the `tksm sequence` binary on files (temporary directory): MDF text in, FASTQ file out, wall time of the whole process --
                  PCIe and host I/O inclusive, never `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline_worker(args):
    """Oracle leg: splice + badread record for a slice of the sample (runs in a forked worker)."""
    lo, hi = args
    po, S = _CB["po"], _CB
    t0 = time.time()
    nb = 0
    for i in range(lo, hi):
        raw = po.splice(S["ref"], S["mols"][i][1])
        rec, _ = po.badread_record(True, 42, i, raw, S["ident"], S["em"], S["qm"], True, S["mols"][i][0])
        nb += len(raw)
    return hi - lo, nb, time.time() - t0


_CB = {}


def host_cores():
    """Worker count for the CPU leg: the affinity mask, capped at the 16-core share a one-GPU box gives us."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("TKSM_BENCH_CORES", "16"))))


def cpu_baseline(n_reads, mean_len, seconds_budget=20.0):
    """Times the CPU oracle (oracle/tksm_oracle.c, kind "port") on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from multiprocessing import Pool
    from tksm_amd import synthetic
    cores = host_cores()
    rs = np.random.RandomState(11)
    lens = [1_000_000] * 24
    ref = {f"chr{i + 1}": rs.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes().decode() for i, n in enumerate(lens)}
    m = synthetic.make_molecules(rs, lens, n_reads, mean_len, mean_len * 0.2)
    text = synthetic.mdf_text(m, list(ref))
    mols = list(po.mdf_generator(text.splitlines(keepends=True)))
    models = os.path.join(ROOT, "tksm_amd", "models", "badread")
    _CB.update(po=po, ref=ref, mols=mols, em=po.ErrorModel(os.path.join(models, "nanopore2020.error.gz")),
               qm=po.QScoreModel(os.path.join(models, "nanopore2020.qscore.gz")), ident=po.Identities(84.0, 5.5, 99.0))
    chunks = [(i * n_reads // cores, (i + 1) * n_reads // cores) for i in range(cores)]
    t0 = time.time()
    with Pool(cores) as p:
        res = p.map(cpu_baseline_worker, chunks)
    wall = time.time() - t0
    n = sum(r[0] for r in res)
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    return {"value": n / wall, "unit": "reads/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "sample": f"{n} reads of the same synthetic bulk workload (mean {mean_len} b, 24 x 1 Mb host genome), "
                      f"CPU oracle oracle/tksm_oracle.c (gcc -O3 -march=native) on {cores} processes, {wall:.1f} s wall",
            "gbases_per_s": sum(r[1] for r in res) / wall / 1e9}


def e2e_leg(n_molecules, n_stream):
    """`tksm sequence` on files -- PCIe and host I/O inclusive, never `value`.  Two runs of the binary on a 4 x 8 Mb genome (MDF text made of
    blocks of 1 M distinct molecules):
      * n_molecules (8 M) into a FASTQ FILE in the temporary directory, wall time of the whole process (start, device, reference packing,
        models, MDF parse, PCIe, 16 GB written): `reads_per_s`.  Bounded by the box's single-file write rate (11 - 13.5 GB/s = ~5.5 M reads/s
        whatever the device does, profiles/r03_fs_write_probe.log);
      * n_stream (32 M) into /dev/null (a character device: the ordered-writer path, no file system), with the CLI's own clocks
        (TKSMSEQ_STATS_FILE): `to_dev_null.stream_reads_per_s` = reads / (first chunk read -> last record byte written), the per-stage
        seconds summed over each stage's threads and the rates that follow from them."""
    import shutil
    import subprocess
    import tempfile
    from tksm_amd import synthetic
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    base = tempfile.gettempdir()
    if shutil.disk_usage(base).free < 40e9 and os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 40e9:
        base = "/dev/shm"
    d = tempfile.mkdtemp(prefix="tksm_e2e_", dir=base)
    try:
        rs = np.random.RandomState(1)
        lens = [8_000_000] * 4
        with open(os.path.join(d, "ref.fa"), "w") as f:
            for c, L in enumerate(lens):
                s = rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode()
                f.write(f">chr{c + 1}\n")
                f.write("\n".join(s[i:i + 80] for i in range(0, L, 80)))
                f.write("\n")
        block = min(n_molecules, 1_000_000)
        m = synthetic.make_molecules(rs, lens, block, 1000, 200)
        text = synthetic.mdf_text(m, [f"chr{c + 1}" for c in range(4)])

        def mdf_file(name, n):
            reps = max(1, n // block)
            with open(os.path.join(d, name), "w") as f:
                for _ in range(reps):
                    f.write(text)
            return block * reps
        n = mdf_file("mols.mdf", n_molecules)
        env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
        cores = host_cores()
        threads = max(1, cores // 2)

        def one(mdf, out_path, stats=None):
            cmd = [exe, "sequence", "-i", os.path.join(d, mdf), "-r", os.path.join(d, "ref.fa"), "-o", out_path, "-t", str(threads), "--verbosity", "ERROR"]
            e = dict(env, TKSMSEQ_STATS_FILE=stats) if stats else env
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, env=e)
            dt = time.time() - t0
            return (None, r.stderr[-300:]) if r.returncode else (dt, None)
        dt, err = one("mols.mdf", os.path.join(d, "out.fastq"))
        if dt is None:
            return {"reads_per_s": None, "error": err}
        res = {"reads_per_s": n / dt, "molecules": n, "wall_s": dt, "mdf_bytes": os.path.getsize(os.path.join(d, "mols.mdf")),
               "fastq_bytes": os.path.getsize(os.path.join(d, "out.fastq")), "files_on": base,
               "bound": "one FASTQ file: the box's file system takes 11 - 13.5 GB/s into a single file (profiles/r03_fs_write_probe.log), ~5.5 M reads/s before any fixed cost",
               "command": "tksm sequence -i mols.mdf -r ref.fa -o out.fastq -t %d (Badread + q-scores, nanopore2020)" % threads}
        os.remove(os.path.join(d, "out.fastq"))
        os.remove(os.path.join(d, "mols.mdf"))
        # the streaming rate: more molecules, records into a character device, the CLI's own clocks.  A pause in front: a process that
        # starts right after another one released its ~100 GiB of device memory (and while the page cache still writes the previous
        # leg's 16 GB back) takes 1.5 - 2 x as long (tools/e2e_ab.sh)
        n2 = mdf_file("stream.mdf", n_stream)
        os.symlink("/dev/null", os.path.join(d, "null.fastq"))
        time.sleep(8)
        stats = os.path.join(d, "stats.json")
        dt0, err0 = one("stream.mdf", os.path.join(d, "null.fastq"), stats)
        leg = {"molecules": n2, "wall_s": dt0, "reads_per_s": n2 / dt0 if dt0 else None, "error": err0}
        if dt0 and os.path.exists(stats):
            st = json.load(open(stats))
            leg["stream_s"] = st["stream_s"]
            leg["setup_s"] = st["setup_s"]
            leg["stream_reads_per_s"] = st["reads"] / st["stream_s"]
            leg["stream_record_GBps"] = st["record_bytes"] / st["stream_s"] / 1e9
            leg["stage_seconds_summed_over_threads"] = {k: st[k] for k in ("read_count_s", "parse_s", "run_s", "device_copy_s", "d2h_wait_s", "write_s", "wait_for_writer_s")}
            safe = lambda x, y: (x / y) if y else None
            leg["rates"] = {"read_and_count_mdf_GBps": safe(st["mdf_bytes"] / 1e9, st["read_count_s"]),
                            "parse_and_upload_mdf_GBps_per_parser": safe(st["mdf_bytes"] / 1e9, st["parse_s"]), "parsers": st["parsers"], "parse_threads_each": st["parse_threads"],
                            "device_Mreads_per_s_per_context": safe(st["reads"] / 1e6, st["run_s"]), "contexts": st["workers"],
                            "d2h_GBps_per_writer_while_waiting": safe(st["d2h_bytes"] / 1e9, st["d2h_wait_s"]),
                            "write_GBps_per_writer": safe(st["record_bytes"] / 1e9, st["write_s"]), "writers": st["workers"]}
        res["to_dev_null"] = leg
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: N child ranks under torch.distributed.run.  Counting devices does not
    initialise the GPU in this process (torch.cuda.device_count reads the driver's list)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("TKSM_BENCH_SKIP_DEVICE_CHECK") and not os.environ.get("TKSM_BENCH_REHEARSE"):     # (the variables: CPU test of this launcher; rehearsal on one GPU)
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1703936, help="molecules per GPU per step (with 8 ranks the ordering buffers of rank 0 bring its HBM use to ~80 % at this size)")
    ap.add_argument("--mean-len", type=int, default=1000)
    ap.add_argument("--genome-contigs", type=int, default=24)
    ap.add_argument("--contig-mb", type=int, default=128)
    ap.add_argument("--kind", default="bulk", choices=["bulk", "scrna", "pcr"], help="bulk (config 2), scRNA-like literals (config 3), substitution-heavy molecules as PCR leaves them (config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU sample (0 = sized for ~20 s)")
    ap.add_argument("--skip-qual", action="store_true")
    ap.add_argument("--perfect", action="store_true", help="bench the integer splice path only (--perfect)")
    ap.add_argument("--lognormal-sigma", type=float, default=0.0, help="transcript-like skewed lengths: lognormal, median --mean-len")
    ap.add_argument("--pipeline", type=int, default=3, help="contexts in flight per GPU (1 = one batch at a time)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI leg")
    ap.add_argument("--ordering", default="gather", choices=["gather", "offsets"],
                    help="N > 1: gather = exact-size send / recv of every rank's records to rank 0 + device interleave (north_star); "
                         "offsets = all_gather of record lengths only, every rank learns its records' final file offsets (tksm_amd/ordering.py)")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the short legs on the other workloads (scRNA-like, PCR-like, lognormal lengths)")
    ap.add_argument("--e2e-molecules", type=int, default=8_000_000, help="end-to-end leg into a FASTQ file")
    ap.add_argument("--e2e-stream-molecules", type=int, default=32_000_000, help="end-to-end leg into /dev/null with the CLI's stage clocks")
    ap.add_argument("--order-check-reads", type=int, default=32768,
                    help="N > 1 (or the forced exchange): after the timed steps every rank runs its round-robin shard of ONE common set of "
                         "N x this many molecules through the same exchange, and rank 0 compares the ordered stream with its own "
                         "single-GPU run of the whole set (BASELINE config 4's FASTQ order check); 0 = skip")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as `python bench.py --gpus N`: this process becomes the launcher and never touches a GPU -- the N ranks are CHILD
        # processes (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1); rank 0's JSON line reaches our stdout as it is
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:      # (before anything slow: a launcher that stops all ranks when the first one fails still shows that every rank started)
        print(f"[bench] rank {os.environ.get('RANK', '0')} of {world} started (local rank {os.environ.get('LOCAL_RANK', '0')})", file=sys.stderr, flush=True)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start N ranks with --gpus N (or leave WORLD_SIZE unset and bench.py starts them)")
    # stdout carries the one JSON line and nothing else: libraries that print while they initialise (RCCL) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    cpu_base = None
    if world == 1 and not args.no_cpu_baseline and not args.perfect:
        # oracle leg first, before this process touches the GPU (it forks worker processes)
        cores = host_cores()
        n_cpu = args.cpu_sample or max(64, int(cores * 20.0 / 0.0023 * (1000.0 / args.mean_len)))
        cpu_base = cpu_baseline(n_cpu, args.mean_len)

    # one hardware queue per stream in flight: the runtime's default of 4 makes the main streams of several contexts share a queue
    # (their kernels then run one after the other); read when the runtime starts
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import torch.distributed as dist
    from tksm_amd import synthetic
    from tksm_amd.sequence import Sequencer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TKSM_BENCH_REHEARSE=gloo: the N > 1 code of this file -- sharding, exchange thread, buffers, order check -- with every rank on
    # GPU 0 and the collectives over gloo on host copies (RCCL refuses two ranks on one device): what a one-GPU box can rehearse of
    # an N-rank run.  Never a measurement: the line carries "rehearsal" and the ranks share one card.
    rehearse = os.environ.get("TKSM_BENCH_REHEARSE") == "gloo"
    if rehearse:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the Seq hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TKSM_BENCH_FORCE_EXCHANGE=1: run the N > 1 code path (RCCL gather + device interleave of every step) on one rank,
    # to exercise it where only one GPU is available
    exchange_on = world > 1 or bool(os.environ.get("TKSM_BENCH_FORCE_EXCHANGE"))
    if exchange_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    cdev = torch.device("cpu") if rehearse else dev          # where the small collectives' tensors live
    import threading

    # ---- inputs resident in HBM before the timed region: genome packed on the device, models, one batch per context
    n_ctx = max(1, args.pipeline)
    clen = args.contig_mb * 1_000_000
    models = os.path.join(ROOT, "tksm_amd", "models", "badread")
    target = "perfect" if args.perfect else "badread"
    compute_q = not args.skip_qual
    cap = int(args.batch * (2.3 * (args.mean_len * (1.25 if args.lognormal_sigma else 1.0) + 60) + 256))

    class Ctx:
        pass
    ctxs = []
    for i in range(n_ctx):
        c = Ctx()
        c.stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("BENCH_STREAM_PRIORITY", "0")))    # (diagnostic: queue priority of the contexts' main streams)
        ctxs.append(c)
    first = ctxs[0]
    first.seqr = Sequencer(local_rank, stream=first.stream.cuda_stream)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for ci in range(args.genome_contigs):
        codes = torch.randint(0, 4, (clen,), dtype=torch.uint8, device=dev, generator=gen)
        ascii_t = lut[codes.long()]
        torch.cuda.synchronize()
        first.seqr.add_contig(f"chr{ci + 1}", ascii_t)
        del codes, ascii_t
    torch.cuda.empty_cache()
    if not args.perfect:
        first.seqr.set_identity(84.0, 99.0, 5.5)
        first.seqr.load_error_model(os.path.join(models, "nanopore2020.error.gz"))
        first.seqr.load_qscore_model(os.path.join(models, "nanopore2020.qscore.gz"))
    for c in ctxs[1:]:
        c.seqr = first.seqr.clone(stream=c.stream.cuda_stream)      # shares the packed reference and the model tables
        if os.environ.get("BENCH_PRIVATE_TABLES"):                    # diagnostic: private copies of the model tables
            c.seqr.load_error_model(os.path.join(models, "nanopore2020.error.gz"))
            c.seqr.load_qscore_model(os.path.join(models, "nanopore2020.qscore.gz"))
    for i, c in enumerate(ctxs):
        rs = np.random.RandomState(2 + rank + 1000 * i)
        c.m = synthetic.make_molecules(rs, [clen] * args.genome_contigs, args.batch, args.mean_len, args.mean_len * 0.2, kind=args.kind,
                                       id_prefix=f"m{rank}", lognormal_sigma=args.lognormal_sigma or None)
        c.batch = c.seqr.batch_from_arrays(c.m["reads"], c.m["intervals"], c.m["mods"], c.m["literals"], c.m["literal_pool"],
                                           c.m["ids"], c.m["id_pool"])
        # with the ordering exchange (N > 1) a context has two output buffers: it computes its next step into one while the other
        # is being gathered
        c.n_out = 2 if exchange_on else 1
        c.out_ts = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(c.n_out)]
        c.off_ts = [torch.empty(args.batch + 1, dtype=torch.int64, device=dev) for _ in range(c.n_out)]
        c.turn = 0
        c.out_t = c.out_ts[0]
        c.seqr.set_output_buffer(c.out_t.data_ptr(), cap)
        c.seqr.set_timing(True)
        c.free = threading.Semaphore(c.n_out)    # output buffers of the context that may be overwritten
    m = ctxs[0].m

    from tksm_amd import ordering
    xstream = torch.cuda.Stream(device=dev) if exchange_on else None

    def run_step(c, t):
        return c.seqr.run(c.batch, target=target, fastq=True, compute_qual=compute_q, seed=42,
                          first_read_index=t * args.batch * world + rank, stride=world)

    # rank 0's interleave runs on the exchange stream itself (a context of its own on that stream): it follows the receives in
    # stream order, no host synchronisation in between
    xseq = first.seqr.clone(stream=xstream.cuda_stream) if exchange_on else None
    xbuf = {"bytes": None, "offs": None, "out": None, "keep": None, "n_out": 0}

    def exchange(out_t, off_t, n_bytes, n_reads=None):
        """FASTQ ordering (N > 1, tksm_amd/ordering.py) on the exchange thread and stream, while the compute contexts run their next
        steps.  gather: the ranks' byte counts (all_gather), then exactly the record bytes and offsets of every rank to rank 0
        (send / recv over RCCL), and the device interleave into global read order.  offsets: the ranks' record lengths only.
        Returns when the exchange stream has drained: with RCCL a finished `wait()` orders the STREAM behind the transfer, not the
        host, and the caller hands `out_t` / `off_t` back to a compute context (another stream) right after."""
        n_reads = args.batch if n_reads is None else n_reads
        with torch.cuda.stream(xstream):
            if args.ordering == "offsets":
                lens = off_t[1:n_reads + 1] - off_t[:n_reads]
                xbuf["keep"] = ordering.global_offsets(lens.cpu() if rehearse else lens, [n_reads] * world, rank, world)
                xstream.synchronize()
                return
            sizes = ordering.exchange_sizes(n_bytes, n_reads, world, cdev)
            if rank == 0:
                need_b, need_o = int(sizes[:, 0].sum()), int(sizes[:, 1].sum()) + world
                if xbuf["bytes"] is None or xbuf["bytes"].numel() < need_b:
                    xbuf["bytes"] = torch.empty(int(need_b * 1.02) + 4096, dtype=torch.uint8, device=dev)
                    xbuf["out"] = torch.empty(int(need_b * 1.02) + 4096, dtype=torch.uint8, device=dev)
                if xbuf["offs"] is None or xbuf["offs"].numel() < need_o:
                    xbuf["offs"] = torch.empty(need_o, dtype=torch.int64, device=dev)
            if rehearse:
                got = ordering.gather_exact(out_t[:n_bytes].cpu(), off_t[:n_reads + 1].cpu(), sizes, rank, world)
                if rank == 0:                                     # (host copies over gloo; the interleave runs on the device as always)
                    hb, bstart, ho, ostart = got
                    xbuf["bytes"][:hb.numel()].copy_(hb); xbuf["offs"][:ho.numel()].copy_(ho)
                    got = (xbuf["bytes"], bstart, xbuf["offs"], ostart)
            else:
                got = ordering.gather_exact(out_t[:n_bytes], off_t[:n_reads + 1], sizes, rank, world, xbuf["bytes"], xbuf["offs"])
            if rank == 0:
                fb, bstart, fo, ostart = got
                xbuf["n_out"] = xseq.interleave_records([fb.data_ptr() + int(bstart[p]) for p in range(world)], [fo.data_ptr() + 8 * int(ostart[p]) for p in range(world)],
                                                        [int(sizes[p, 1]) for p in range(world)], xbuf["out"].data_ptr(), xbuf["out"].numel())
            xstream.synchronize()

    def run_steps(first, count):
        """steps first .. first+count-1: context t mod n_ctx runs step t on its own thread; with N > 1 the main thread
        performs the ordering exchange of every step in step order while the other context keeps computing."""
        results = [None] * count
        handed = [None] * count
        done = [threading.Event() for _ in range(count)]
        errors = []

        def worker(i):
            try:
                torch.cuda.set_device(local_rank)
                for j in range(i, count, n_ctx):
                    c = ctxs[i]
                    c.free.acquire()
                    results[j] = run_step(c, first + j)
                    if not exchange_on:
                        c.free.release()
                    else:
                        # offsets next to the records, then the next step of this context goes to its other buffer
                        off_t = c.off_ts[c.turn]
                        results[j].copy_to_device(None, off_t.data_ptr())
                        c.seqr.synchronize()
                        handed[j] = (c.out_ts[c.turn], off_t, int(results[j].records_bytes))
                        c.turn = (c.turn + 1) % c.n_out
                        c.seqr.set_output_buffer(c.out_ts[c.turn].data_ptr(), cap)
                    done[j].set()
            except Exception as e:      # surface in the main thread
                errors.append(e)
                for d in done:
                    d.set()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(min(n_ctx, count))]
        for x in th:
            x.start()
        if exchange_on:
            for j in range(count):
                done[j].wait()
                if errors:
                    break
                c = ctxs[j % n_ctx]
                exchange(*handed[j])
                c.free.release()
        for x in th:
            x.join()
        if errors:
            raise errors[0]
        return results

    def run_order_check():
        """BASELINE config 4's FASTQ order check on the ranks of this run (untimed, after the timed steps): ONE common set of
        world x nv molecules (every rank draws the same set), rank p runs reads p, p + world, ... of it and the SAME exchange as the timed
        steps orders them on rank 0 -- which also runs the whole set alone (stride 1) and compares the two streams on the device,
        byte for byte (gather), or every rank's offsets with the single-GPU record offsets (offsets)."""
        nv = args.order_check_reads
        c = ctxs[0]
        rs = np.random.RandomState(4242)
        g = synthetic.make_molecules(rs, [clen] * args.genome_contigs, nv * world, args.mean_len, args.mean_len * 0.2, kind=args.kind,
                                     id_prefix="oc", lognormal_sigma=args.lognormal_sigma or None)
        mine = synthetic.take(g, np.arange(rank, nv * world, world))
        keep = c.batch
        sb = c.seqr.batch_from_arrays(mine["reads"], mine["intervals"], mine["mods"], mine["literals"], mine["literal_pool"], mine["ids"], mine["id_pool"])
        out_t, off_t = c.out_ts[0], c.off_ts[0]
        c.seqr.set_output_buffer(out_t.data_ptr(), cap)
        r = c.seqr.run(sb, target=target, fastq=True, compute_qual=compute_q, seed=42, first_read_index=rank, stride=world)
        r.copy_to_device(None, off_t.data_ptr())
        c.seqr.synchronize()
        exchange(out_t, off_t, int(r.records_bytes), nv)
        sb.free()
        res = {"reads": nv * world, "ordering": args.ordering, "equal": None}
        if args.ordering == "offsets":
            my_off, total = xbuf["keep"]
            my_off = my_off.clone()
        if rank == 0 or args.ordering == "offsets":
            wb = c.seqr.batch_from_arrays(g["reads"], g["intervals"], g["mods"], g["literals"], g["literal_pool"], g["ids"], g["id_pool"])
            rw = c.seqr.run(wb, target=target, fastq=True, compute_qual=compute_q, seed=42, first_read_index=0, stride=1)
            woff = torch.empty(nv * world + 1, dtype=torch.int64, device=dev)
            rw.copy_to_device(None, woff.data_ptr())
            c.seqr.synchronize()
            if args.ordering == "gather":
                n_out = int(xbuf["n_out"])
                res["bytes"] = n_out
                res["equal"] = bool(n_out == int(rw.records_bytes) and torch.equal(xbuf["out"][:n_out], out_t[:n_out]))
            else:
                ok = bool(total == int(rw.records_bytes) and torch.equal(my_off.to(dev), woff[rank:nv * world:world]))
                res["bytes"] = int(total)
                res["equal"] = ok
            wb.free()
        c.batch = keep
        if args.ordering == "offsets":
            flag = torch.tensor([1 if res["equal"] else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            res["equal"] = bool(flag.item())
        return res

    def fence():
        torch.cuda.synchronize()
        if exchange_on:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(0, args.warmup * n_ctx)        # every context warms up (allocations, code objects): untimed
    fence()
    t0 = time.perf_counter()
    results = run_steps(args.warmup * n_ctx, args.steps)
    if rank == 0:
        free_b, total_b = torch.cuda.mem_get_info()
        print(f"[bench] HBM in use after the timed steps: {(total_b - free_b) / 2**30:.1f} of {total_b / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    fence()
    elapsed = time.perf_counter() - t0
    sim_ms = [r.kernel_ms[1] for r in results]
    tot_ms = [r.kernel_ms[4] for r in results]
    rec_bytes = sum(r.records_bytes for r in results)
    bases_in = sum(r.bases_in for r in results)
    bases_out = sum(r.bases_out for r in results)
    overlapped = {"k_loop": float(np.mean([r.kernel_ms[5] for r in results])), "k_alnf": float(np.mean([r.kernel_ms[6] for r in results]))}
    # exclusive per-kernel time: one more step on one context with the GPU to itself (untimed for `value`)
    fence()
    rx = run_step(ctxs[0], args.warmup * n_ctx + args.steps)
    ctxs[0].seqr.synchronize()
    fence()
    exclusive = {"k_loop": float(rx.kernel_ms[5]), "k_alnf": float(rx.kernel_ms[6]),
                 "simulate_stage_total": float(rx.kernel_ms[1]), "all_kernels": float(rx.kernel_ms[4])}
    order_check = None
    if exchange_on and args.order_check_reads > 0 and not args.perfect:
        order_check = run_order_check()
    if exchange_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        agg = torch.tensor([bases_in, bases_out], dtype=torch.float64, device=cdev)
        dist.all_reduce(agg)
        bases_in_all, bases_out_all = float(agg[0].item()), float(agg[1].item())
    else:
        bases_in_all, bases_out_all = float(bases_in), float(bases_out)
    if rank != 0:
        if xseq is not None:
            xseq.close()
        for c in reversed(ctxs):
            c.seqr.close()
        dist.destroy_process_group()
        return
    reads = args.batch * world * args.steps
    value = reads / elapsed
    alg = synthetic.algorithmic_bytes(m, rec_bytes / args.steps)
    if args.perfect:
        # --perfect: one kernel writes the records straight from the packed reference (timed as the emit stage)
        dom, dom_ms = "k_perfect", float(rx.kernel_ms[3])
        overlapped = {"k_perfect": float(np.mean([r.kernel_ms[3] for r in results]))}
        exclusive = {"k_perfect": dom_ms, "all_kernels": float(rx.kernel_ms[4])}
    elif exclusive["k_loop"] + exclusive["k_alnf"] == 0.0:
        dom, dom_ms = "k_simulate", exclusive["simulate_stage_total"]
    else:
        dom = max(("k_loop", "k_alnf"), key=lambda k: exclusive[k])
        dom_ms = exclusive[dom]
    achieved = alg / (dom_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("batch") == args.batch and tj.get("kind", "bulk") == args.kind and not args.perfect:
                traffic = tj.get("hbm_bytes_per_step", {}).get(dom)
                traffic_source = f"profiles/{tj.get('profile', 'traffic_latest.json')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)"
        except Exception:
            traffic = None
    kind_name = {"bulk": "Bulk", "scrna": "scRNA-like (barcode, UMI, polyA literals)", "pcr": "PCR-amplified (substitution-heavy)"}[args.kind]
    out = {
        "metric": "sequenced reads/s", "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{kind_name} molecules, Badread error+qual model (nanopore2020, identity 84,99,5.5), "
                               f"{args.genome_contigs} x {args.contig_mb} Mb random genome, FASTQ"
                               if not args.perfect else f"{kind_name} molecules, --perfect splice path, FASTQ",
                   "kind": args.kind, "molecules_per_gpu_per_step": args.batch, "mean_len": args.mean_len,
                   "length_distribution": f"lognormal(median {args.mean_len}, sigma {args.lognormal_sigma})" if args.lognormal_sigma else f"normal({args.mean_len}, {args.mean_len * 0.2:.0f})",
                   "compute_qual": bool(compute_q and not args.perfect), "sharding": f"round-robin x{world}",
                   "ordering": (args.ordering if exchange_on else None), "contexts_in_flight_per_gpu": n_ctx,
                   **({"rehearsal": "every rank on GPU 0, collectives over gloo on host copies: not a measurement"} if rehearse else {})},
        "gbases_per_s": bases_in_all / elapsed / 1e9,
        "gbases_out_per_s": bases_out_all / elapsed / 1e9,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "kernel": dom,
                     "kernel_ms": dom_ms, "kernel_ms_is": "exclusive: summed launch durations of the kernel in one step run alone on the GPU",
                     "algorithmic_bytes_per_launch": alg,
                     "exclusive_ms_per_step": exclusive, "overlapped_ms_per_step": overlapped,
                     "overlapped_all_kernels_ms": float(np.mean(tot_ms)), "overlapped_simulate_stage_ms": float(np.mean(sim_ms))},
    }
    if world == 1 and not exchange_on and not args.no_side_legs and not args.perfect and args.kind == "bulk" and not args.lognormal_sigma:
        # short driver-timed legs on the other workloads (same contexts, models and genome; 1 warm-up round + 3 timed steps each;
        # never `value`): BASELINE config 3's scRNA-like molecules, config 5's substitution-heavy molecules, transcript-like lengths
        side = {}
        legs = (("scrna", dict(kind="scrna"), args.batch), ("pcr", dict(kind="pcr"), args.batch),
                ("lognormal_sigma_0.6", dict(kind="bulk", lognormal_sigma=0.6), min(args.batch, 1048576)))
        for name, kw, nb in legs:
            try:
                for i, c in enumerate(ctxs):
                    c.batch.free()
                    rs = np.random.RandomState(7000 + 10 * i + len(name))
                    mm = synthetic.make_molecules(rs, [clen] * args.genome_contigs, nb, args.mean_len, args.mean_len * 0.2, id_prefix="s", **kw)
                    c.batch = c.seqr.batch_from_arrays(mm["reads"], mm["intervals"], mm["mods"], mm["literals"], mm["literal_pool"], mm["ids"], mm["id_pool"])
                    c.seqr.set_output_buffer(0, 0)             # (records of other sizes: the context's own buffer)
                def leg_step(c, t):
                    return c.seqr.run(c.batch, target=target, fastq=True, compute_qual=compute_q, seed=42, first_read_index=t * nb, stride=1)
                def leg_steps(t0, count):
                    res = [None] * count
                    def w(i):
                        torch.cuda.set_device(local_rank)
                        for j in range(i, count, n_ctx):
                            res[j] = leg_step(ctxs[i], t0 + j)
                    th = [threading.Thread(target=w, args=(i,)) for i in range(min(n_ctx, count))]
                    [x.start() for x in th]; [x.join() for x in th]
                    return res
                leg_steps(0, n_ctx)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rr = leg_steps(n_ctx, 3)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                side[name] = {"reads_per_s": 3 * nb / dt, "gbases_per_s": sum(r.bases_in for r in rr) / dt / 1e9, "steps": 3, "molecules_per_step": nb,
                              "ms_per_step": dt / 3 * 1e3}
            except Exception as e:                   # a side leg never costs the bench line
                side[name] = {"reads_per_s": None, "error": repr(e)[:300]}
        out["side_legs"] = side
    if world == 1 and not args.no_e2e and not args.perfect and args.kind == "bulk":
        # the CLI is another process on the same GPU: release this one's contexts and buffers first
        if xseq is not None:
            xseq.close()
            xseq = None
        for c in reversed(ctxs):
            c.batch.free()
            c.seqr.close()
            c.out_t = c.out_ts = c.off_ts = None
        ctxs.clear()
        torch.cuda.empty_cache()
        try:
            e2e = e2e_leg(args.e2e_molecules, args.e2e_stream_molecules)
        except Exception as e:                   # the end-to-end leg is an extra: it never costs the bench line
            e2e = {"reads_per_s": None, "error": repr(e)[:300]}
        out["e2e_reads_per_s"] = e2e["reads_per_s"]
        out["e2e"] = e2e
    out["cpu_baseline"] = cpu_base
    if order_check is not None:
        out["order_check"] = order_check
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if xseq is not None:
        xseq.close()
    for c in reversed(ctxs):
        c.seqr.close()
    if exchange_on:
        dist.destroy_process_group()
    if order_check is not None and order_check["equal"] is False:
        sys.exit("bench.py: FASTQ order check FAILED -- the ordered stream of the ranks differs from the single-GPU run")


if __name__ == "__main__":
    main()
