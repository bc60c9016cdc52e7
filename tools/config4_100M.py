"""BASELINE config 4's FASTQ order check at its stated size, as far as ONE GPU can show it: `tksm sequence` on 100 M bulk molecules
(Badread error + q-scores), once on one device group (`--devices 0`: the unsharded order, = the reference's `-t 1` order,
py/sequence.py:360-368) and once sharded round-robin over two device groups that happen to be the same card (`--devices 0,0`: the
product's multi-GPU path -- a worker per group, reads g mod 2, every record written at its scanned place).  ~220 GB of FASTQ per run go
through a pipe into xxh3-128 and a line count, never onto a disk; the check is that both runs give the same digest and 4 lines per molecule.

    python tools/config4_100M.py [molecules=100000000]          (GPU box; copy the output to profiles/)
"""
import fcntl
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tksm_amd import synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    d = os.environ.get("E2E_DIR", "/tmp/c4_100M")
    os.makedirs(d, exist_ok=True)
    rs = np.random.RandomState(1)
    lens = [8_000_000] * 4
    with open(f"{d}/ref.fa", "w") as f:
        for c, L in enumerate(lens):
            s = rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode()
            f.write(f">chr{c + 1}\n" + "\n".join(s[i:i + 80] for i in range(0, L, 80)) + "\n")
    block = min(n, 1_000_000)
    n = (n // block) * block
    t0 = time.time()
    text = synthetic.mdf_text(synthetic.make_molecules(rs, lens, block, 1000, 200), [f"chr{c + 1}" for c in range(4)])
    with open(f"{d}/in.mdf", "w") as f:
        for _ in range(max(1, n // block)):
            f.write(text)
    print(f"{n} molecules ({n // block} blocks of {block} distinct ones), {os.path.getsize(f'{d}/in.mdf') / 1e9:.2f} GB of MDF text, written in {time.time() - t0:.0f} s", flush=True)
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    res = {}
    for tag, devices in (("one group", "0"), ("two groups, round-robin", "0,0")):
        fifo = f"{d}/out_{devices.replace(',', '_')}.fastq"
        if os.path.exists(fifo):
            os.unlink(fifo)
        os.mkfifo(fifo)
        stats = f"{d}/stats_{devices.replace(',', '_')}.json"
        env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"), TKSMSEQ_STATS_FILE=stats)
        got = {}

        def drain():
            h, nb, lines, last = xxhash.xxh3_128(), 0, 0, time.time()
            with open(fifo, "rb", buffering=0) as p:
                try:
                    fcntl.fcntl(p.fileno(), 1031, 1 << 20)            # F_SETPIPE_SZ
                except OSError:
                    pass
                while True:
                    b = p.read(1 << 24)
                    if not b:
                        break
                    h.update(b); nb += len(b); lines += b.count(b"\n")
                    if time.time() - last > 30:
                        last = time.time()
                        print(f"    ... {nb / 1e9:.0f} GB", flush=True)
            got.update(digest=h.hexdigest(), bytes=nb, lines=lines)
        th = threading.Thread(target=drain, daemon=True)
        th.start()
        t0 = time.time()
        r = subprocess.run([exe, "sequence", "-i", f"{d}/in.mdf", "-r", f"{d}/ref.fa", "-o", fifo, "-s", "42", "-t", "8", "--devices", devices, "--verbosity", "INFO"],
                           capture_output=True, text=True, env=env)
        wall = time.time() - t0
        th.join(timeout=300)
        st = json.load(open(stats)) if os.path.exists(stats) else {}
        print(f"--devices {devices} ({tag}): rc={r.returncode}, {wall:.1f} s wall, {st.get('reads')} reads in {st.get('batches')} batches, {got.get('bytes', 0) / 1e9:.1f} GB of FASTQ "
              f"through the pipe ({got.get('bytes', 0) / wall / 1e9:.2f} GB/s: the pipe and the hash are the limit here, not the device), {got.get('lines')} lines, "
              f"xxh3-128 {got.get('digest')}", flush=True)
        if r.returncode:
            sys.stderr.write(r.stderr[-2000:])
            sys.exit(1)
        res[devices] = (got, st.get("reads"))
        os.unlink(fifo)
    a, b = res["0"], res["0,0"]
    ok = a[0] == b[0] and a[1] == b[1] == n and a[0]["lines"] == 4 * n
    print(f"FASTQ order check at {n} molecules: {'EQUAL' if ok else 'DIFFERENT'} (same digest, same byte count, 4 lines per molecule)", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
