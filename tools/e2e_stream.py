"""The end-to-end streaming leg of bench.py on its own, for A/B runs of the CLI's knobs: `tksm sequence` on N molecules (blocks of 1 M distinct
ones) into /dev/null with the CLI's stage clocks (TKSMSEQ_STATS_FILE) and, with V=2, its per-batch timeline on stderr.

    python tools/e2e_stream.py [molecules=32000000] [-- extra CLI arguments ...]      env: V=<TKSMSEQ_VERBOSE level>, E2E_DIR, OUT=<path instead of /dev/null>,
                                                                                          KIND=bulk|scrna|pcr (synthetic.make_molecules; scrna = BASELINE config 3's barcode / UMI / polyA literals)
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tksm_amd import synthetic  # noqa: E402


def main():
    a = sys.argv[1:]
    extra = []
    if "--" in a:
        extra = a[a.index("--") + 1:]
        a = a[:a.index("--")]
    n = int(a[0]) if a else 32_000_000
    d = os.environ.get("E2E_DIR", "/tmp/e2e_stream")
    os.makedirs(d, exist_ok=True)
    kind = os.environ.get("KIND", "bulk")
    mdf = f"{d}/stream_{kind}_{n}.mdf"
    if not os.path.exists(mdf) or not os.path.exists(f"{d}/ref.fa"):
        rs = np.random.RandomState(1)
        lens = [8_000_000] * 4
        with open(f"{d}/ref.fa", "w") as f:
            for c, L in enumerate(lens):
                s = rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode()
                f.write(f">chr{c + 1}\n" + "\n".join(s[i:i + 80] for i in range(0, L, 80)) + "\n")
        block = min(n, 1_000_000)
        text = synthetic.mdf_text(synthetic.make_molecules(rs, lens, block, 1000, 200, kind=kind), [f"chr{c + 1}" for c in range(4)])
        with open(mdf, "w") as f:
            for _ in range(max(1, n // block)):
                f.write(text)
    out = os.environ.get("OUT")
    if not out:
        out = f"{d}/null.fastq"
        if not os.path.islink(out):
            os.symlink("/dev/null", out)
    stats = f"{d}/stats.json"
    env = dict(os.environ, TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"), TKSMSEQ_STATS_FILE=stats)
    if os.environ.get("V"):
        env["TKSMSEQ_VERBOSE"] = os.environ["V"]
    cmd = [os.path.join(ROOT, "tksm_amd", "tksm"), "sequence", "-i", mdf, "-r", f"{d}/ref.fa", "-o", out, "-t", "8", "--verbosity", "ERROR"] + extra
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    st = json.load(open(stats)) if os.path.exists(stats) else {}
    print(f"kind={kind} molecules={st.get('reads')} mdf {os.path.getsize(mdf) / 1e9:.2f} GB, records {st.get('record_bytes', 0) / 1e9:.1f} GB; extra={extra} rc={r.returncode} wall {wall:.2f} s; stream {st.get('stream_s')} s = {st.get('reads', 0) / max(1e-9, st.get('stream_s', 1)) / 1e6:.2f} M reads/s; "
          f"batches {st.get('batches')}; summed stage seconds: read {st.get('read_count_s')}, parse {st.get('parse_s')}, run {st.get('run_s')}, device copy {st.get('device_copy_s')}, "
          f"d2h wait {st.get('d2h_wait_s')}, write {st.get('write_s')}, wait for writer {st.get('wait_for_writer_s')}", flush=True)
    if os.environ.get("V"):
        sys.stderr.write(r.stderr)
    elif r.returncode:
        sys.stderr.write(r.stderr[-2000:])


if __name__ == "__main__":
    main()
