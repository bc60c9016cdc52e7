"""end-to-end (PCIe + host I/O inclusive) rate of the `tksm sequence` CLI on synthetic files"""
import os, sys, time, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tksm_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
extras = [[]]                                          # e.g. --in-flight 1 --batch-bytes 16777216 [:: another set of flags ...]
for a in sys.argv[2:]:
    if a == "::": extras.append([])
    else: extras[-1].append(a)
modes = os.environ.get("E2E_MODES", "perfect,badread").split(",")
d = os.environ.get("E2E_DIR", "/tmp/e2e"); os.makedirs(d, exist_ok=True)
rs = np.random.RandomState(1)
lens = [8_000_000] * 4
with open(f"{d}/ref.fa", "w") as f:
    for c, L in enumerate(lens):
        s = rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode()
        f.write(f">chr{c+1}\n"); f.write("\n".join(s[i:i+80] for i in range(0, L, 80))); f.write("\n")
m = synthetic.make_molecules(rs, lens, n, 1000, 200)
t = time.time(); text = synthetic.mdf_text(m, [f"chr{c+1}" for c in range(4)]); open(f"{d}/mols.mdf", "w").write(text)
print(f"MDF text {len(text)/1e6:.0f} MB for {n} molecules (generated in {time.time()-t:.0f} s)", flush=True)
env = dict(os.environ, TKSMSEQ_VERBOSE=os.environ.get("E2E_VERBOSE", "1"), TKSM_MODELS=os.path.join(os.getcwd(), "tksm_amd", "models"))
ext = ".fastq.gz" if os.environ.get("E2E_GZ") else ".fastq"
for extra, args, name in [(e, a, nm) for e in extras for a, nm in ((["--perfect", f"{d}/p{ext}"], "perfect"), (["-o", f"{d}/b{ext}"], "badread+qual")) if nm.split("+")[0] in modes]:
    if os.path.exists(args[-1]): os.remove(args[-1])   # (overwriting a 16 GB file costs 1 - 3 s more: truncation, and ext4 flushes a file replaced via truncate when it is closed)
    t = time.time()
    r = subprocess.run([os.environ.get("E2E_EXE", os.path.join("tksm_amd", "tksm")), "sequence", "-i", f"{d}/mols.mdf", "-r", f"{d}/ref.fa"] + args + extra, capture_output=True, text=True, env=env)
    dt = time.time() - t
    out = args[-1]
    print(f"{name} {' '.join(extra)}: rc={r.returncode} {dt:.2f} s wall -> {n/dt:.0f} reads/s end to end, output {os.path.getsize(out)/1e6:.0f} MB", flush=True)
    if r.returncode: print(r.stderr[-500:])
    for line in r.stderr.splitlines():
        if line.startswith("["): print(line)
