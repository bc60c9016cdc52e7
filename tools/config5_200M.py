"""BASELINE config 5 at its stated size on ONE GPU: `tksm sequence --pcr-cycles 20 --pcr-molecule-count 200000000 --pcr-preset Taq-setting1
--truncate-lognormal 6.9,0.5 -o <FASTQ into /dev/null>` -- 200 k templates amplified to 200 M molecules (src/pcr.cpp:66-89; the reference holds
every molecule in RAM, :215), truncated (src/truncate.cpp:322-351) and sequenced (Badread + q-scores), all on the device in slices of
--pcr-slice-molecules copies.  Prints wall time, molecules/s, the CLI's own streaming clock, the host's peak resident set (children of this
process) and the device's peak memory in use (sysfs, polled) -- the record of the bounded-memory claim.

    python tools/config5_200M.py [molecules=200000000] [templates=200000] [devices=0]      (GPU box; copy the output to profiles/)
"""
import glob
import json
import os
import resource
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tksm_amd import synthetic  # noqa: E402


def vram_files():
    return sorted(glob.glob("/sys/class/drm/card*/device/mem_info_vram_used"))


def main():
    a = sys.argv[1:]
    target = int(a[0]) if len(a) > 0 else 200_000_000
    n_templates = int(a[1]) if len(a) > 1 else 200_000
    devices = a[2] if len(a) > 2 else "0"
    d = os.environ.get("E2E_DIR", "/tmp/c5_200M")
    os.makedirs(d, exist_ok=True)
    rs = np.random.RandomState(1)
    lens = [8_000_000] * 4
    with open(f"{d}/ref.fa", "w") as f:
        for c, L in enumerate(lens):
            s = rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode()
            f.write(f">chr{c + 1}\n")
            f.write("\n".join(s[i:i + 80] for i in range(0, L, 80)))
            f.write("\n")
    m = synthetic.make_molecules(rs, lens, n_templates, 1000, 200)
    with open(f"{d}/in.mdf", "w") as f:
        f.write(synthetic.mdf_text(m, [f"chr{c + 1}" for c in range(4)]))
    exe = os.environ.get("E2E_EXE", os.path.join(ROOT, "tksm_amd", "tksm"))
    stats = f"{d}/stats.json"
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"), TKSMSEQ_STATS_FILE=stats)
    out = f"{d}/null.fastq"
    if not os.path.islink(out):
        os.symlink("/dev/null", out)
    cmd = [exe, "sequence", "-i", f"{d}/in.mdf", "-r", f"{d}/ref.fa", "-o", out, "-t", "8", "--devices", devices, "--pcr-cycles", "20",
           "--pcr-molecule-count", str(target), "--pcr-preset", "Taq-setting1", "--truncate-lognormal", "6.9,0.5", "--verbosity", "INFO"]
    files = vram_files()
    base = [int(open(p).read()) for p in files]
    peak = list(base)
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            for i, p in enumerate(files):
                try:
                    peak[i] = max(peak[i], int(open(p).read()))
                except OSError:
                    pass
            time.sleep(0.1)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    print("command:", " ".join(cmd[1:]).replace(d, "."), flush=True)
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    stop.set()
    th.join()
    rss_kb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    print(f"rc={r.returncode}, {wall:.1f} s wall ({n_templates} templates, {os.path.getsize(f'{d}/in.mdf') / 1e6:.0f} MB of MDF text in)")
    for line in r.stderr.splitlines():
        if "Sequencing:" in line or "Error" in line or (os.environ.get("TKSMSEQ_VERBOSE") and line.startswith("[sequence]")):
            print("   ", line)
    if os.path.exists(stats):
        st = json.load(open(stats))
        print(f"molecules sequenced: {st['reads']} in {st['batches']} slices = {st['reads'] / wall / 1e6:.2f} M molecules/s of wall time, "
              f"{st['reads'] / st['stream_s'] / 1e6:.2f} M/s while streaming ({st['stream_s']:.1f} s; set-up {st['setup_s']:.1f} s); "
              f"{st['record_bytes'] / 1e9:.1f} GB of FASTQ records = {st['record_bytes'] / st['stream_s'] / 1e9:.1f} GB/s into /dev/null")
        print("stage seconds, summed over their threads:", {k: st[k] for k in st if k.endswith("_s")})
    print(f"peak host resident set of the command: {rss_kb / 2**20:.2f} GiB")
    # (sysfs lists every card of the host; ours is the one whose use moved)
    for p, b, k in zip(files, base, peak):
        if k - b > (1 << 30):
            print(f"device memory in use, {p.split('/')[4]}: {b / 2**30:.1f} GiB before, peak {k / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
