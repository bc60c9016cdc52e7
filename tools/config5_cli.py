"""config 5 on files: `tksm pcr` -> `tksm truncate` -> `tksm sequence`, wall time of every process, then the same as ONE command
(`tksm sequence --pcr-... --truncate-...`: molecule tables stay on the device, no MDF text in between) and a byte comparison of the two FASTQ files (diagnostic)"""
import os, sys, time, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
from tksm_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
target = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * n
d = os.environ.get("E2E_DIR", "/tmp/c5"); os.makedirs(d, exist_ok=True)
rs = np.random.RandomState(1)
lens = [8_000_000] * 4
with open(f"{d}/ref.fa", "w") as f:
    for c, L in enumerate(lens):
        s = rs.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes().decode()
        f.write(f">chr{c+1}\n"); f.write("\n".join(s[i:i+80] for i in range(0, L, 80))); f.write("\n")
m = synthetic.make_molecules(rs, lens, n, 1000, 200)
open(f"{d}/in.mdf", "w").write(synthetic.mdf_text(m, [f"chr{c+1}" for c in range(4)]))
exe = os.environ.get("E2E_EXE", os.path.join("tksm_amd", "tksm"))
env = dict(os.environ, TKSMSEQ_VERBOSE="1", TKSM_MODELS=os.path.join(ROOT, "tksm_amd", "models"))
import hashlib
steps = [("pcr", ["pcr", "-i", f"{d}/in.mdf", "-o", f"{d}/pcr.mdf", "--molecule-count", str(target), "--cycles", "20", "-x", "Taq-setting1"]),
         ("truncate", ["truncate", "-i", f"{d}/pcr.mdf", "-o", f"{d}/trc.mdf", "--lognormal", "6.9,0.5"]),
         ("sequence", ["sequence", "-i", f"{d}/trc.mdf", "-r", f"{d}/ref.fa", "-o", f"{d}/out.fastq", "-t", "8"])]
for name, args in steps:
    for p in (args[args.index("-o") + 1],):
        if os.path.exists(p): os.remove(p)
    t = time.time(); r = subprocess.run([exe] + args, capture_output=True, text=True, env=env); dt = time.time() - t
    out = args[args.index("-o") + 1]
    size = os.path.getsize(out) if os.path.exists(out) else 0
    nm = sum(1 for line in open(out) if line[0] in "+@") if name != "sequence" and size else 0
    print(f"{name}: rc={r.returncode} {dt:.2f} s wall, output {size/1e6:.0f} MB" + (f", {nm} molecules -> {nm/dt/1e6:.2f} M molecules/s" if nm else ""), flush=True)
    if r.returncode: print(r.stderr[-400:])
    for line in r.stderr.splitlines():
        if line.startswith("[") and name != "sequence": print("   ", line)

def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()
three = md5(f"{d}/out.fastq")
chained = ["sequence", "-i", f"{d}/in.mdf", "-r", f"{d}/ref.fa", "-o", f"{d}/chained.fastq", "-t", "8", "--pcr-cycles", "20", "--pcr-molecule-count", str(target),
           "--pcr-preset", "Taq-setting1", "--truncate-lognormal", "6.9,0.5"]
if os.path.exists(f"{d}/chained.fastq"): os.remove(f"{d}/chained.fastq")
t = time.time(); r = subprocess.run([exe] + chained, capture_output=True, text=True, env=dict(env, TKSMSEQ_VERBOSE="0")); dt = time.time() - t
print(f"chained (one command): rc={r.returncode} {dt:.2f} s wall, output {os.path.getsize(f'{d}/chained.fastq')/1e6:.0f} MB, "
      f"FASTQ {'identical to' if md5(f'{d}/chained.fastq') == three else 'DIFFERENT from'} the three-module route", flush=True)
if r.returncode: print(r.stderr[-400:])
