"""Per-kernel HBM bytes per step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both report KiB), corrected with
the factors calibrated on this repo's access patterns (tools/calib/fetch_calib.hip -> profiles/r02_fetch_calibration.json):
FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide coalesced reads are under-reported by 2 (MI355X_MICROARCH.md),
per-lane 64-byte runs (k_alnf's code lines and slot codes -- every lane streams rows of its own; k_job's slot codes and k_aln's
record and code lines in round 2) by 1 / 0.606 (whole 64-byte lines moved by 4
lanes x 16 bytes, k_aln up to r02_b: 1 / 0.542); single 16-byte gathers (k_loop's table reads) are counted as one 64-byte sector
each (factor 1); WRITE_SIZE is exact."""
import csv, glob, collections, json, re, sys

d, nsteps = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1703936
tag = sys.argv[4] if len(sys.argv) > 4 else ""
FETCH_FACTOR = {"k_alnf": 1 / 0.606, "k_aln": 1 / 0.606, "k_job": 1 / 0.606, "k_loop": 1.0, "k_qjobs": 1.0, "k_err": 2.0, "k_init": 2.0, "k_emit": 2.0, "k_perfect": 2.0,
                "k_pack": 2.0, "k_simulate": 1.0}


def sums(sub, counter):
    f = glob.glob(f"{d}/{sub}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        mt = re.search(r"tk::(k_\w+)", r["Kernel_Name"])
        if mt and r["Counter_Name"] == counter:
            agg[mt.group(1)] += float(r["Counter_Value"]) * 1024.0 / nsteps
    return dict(agg)


fe, wr = sums("fetch", "FETCH_SIZE"), sums("write", "WRITE_SIZE")
fc = {k: v * FETCH_FACTOR.get(k, 1.0) for k, v in fe.items()}
print(json.dumps({
    "batch": batch, "kind": "bulk", "profile": f"{tag}_hbm_traffic.json",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --steps 3 --warmup 1 "
              f"--no-cpu-baseline --no-e2e (3 contexts in flight: 3 warm-up + 3 timed steps + the exclusive step); per-kernel sums divided by {nsteps} steps",
    "units": f"bytes per step ({batch:,} reads); the counters report KiB",
    "calibration": "fetch_bytes_per_step = FETCH_SIZE x the per-pattern factor of profiles/r02_fetch_calibration.json (see this tool's docstring); "
                   "raw counter values in fetch_counter_bytes_per_step",
    "fetch_factor": {k: FETCH_FACTOR.get(k, 1.0) for k in sorted(fe)},
    "hbm_bytes_per_step": {k: fc.get(k, 0) + wr.get(k, 0) for k in sorted(set(fe) | set(wr))},
    "fetch_bytes_per_step": fc, "fetch_counter_bytes_per_step": fe, "write_bytes_per_step": wr}, indent=1))
