"""Per-kernel HBM bytes per step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both report KiB)."""
import csv, glob, collections, json, re, sys

d, nsteps = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1310720


def sums(sub, counter):
    f = glob.glob(f"{d}/{sub}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        mt = re.search(r"tk::(k_\w+)", r["Kernel_Name"])
        if mt and r["Counter_Name"] == counter:
            k = mt.group(1)
            agg[k] += float(r["Counter_Value"]) * 1024.0 / nsteps
    return dict(agg)


fe, wr = sums("fetch", "FETCH_SIZE"), sums("write", "WRITE_SIZE")
print(json.dumps({
    "batch": batch, "kind": "bulk",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --steps 3 --warmup 1 "
              f"--no-cpu-baseline (3 contexts in flight: 3 warm-up + 3 timed steps); per-kernel sums over the {nsteps} steps divided by {nsteps}",
    "units": f"bytes per step ({batch:,} reads); the counters report KiB",
    "calibration": "FETCH_SIZE used uncorrected: the dominant loads are 8-byte-per-lane and 64-byte-block accesses, not the "
                   "16 B/lane streaming reads the gfx950 1/2 factor of MI355X_MICROARCH.md applies to",
    "hbm_bytes_per_step": {k: fe.get(k, 0) + wr.get(k, 0) for k in sorted(set(fe) | set(wr))},
    "fetch_bytes_per_step": fe, "write_bytes_per_step": wr}, indent=1))
