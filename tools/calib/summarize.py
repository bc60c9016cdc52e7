"""FETCH_SIZE / WRITE_SIZE (KiB) per calibration kernel against the known byte counts printed by fetch_calib."""
import csv, glob, json, re, sys
known = {}
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) >= 3:
        known[f[0]] = {f[i]: int(f[i + 1]) for i in range(1, len(f) - 1, 2)}
out = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(f"{sys.argv[2]}/{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(calib_\w+)", r["Kernel_Name"])
            if m and r["Counter_Name"] == counter:
                out.setdefault(m.group(1), {})[counter + "_bytes"] = out.get(m.group(1), {}).get(counter + "_bytes", 0) + float(r["Counter_Value"]) * 1024
for k, v in out.items():
    v["known"] = known.get(k)
    kb = known.get(k, {})
    ref = kb.get("read_bytes") or kb.get("write_bytes") or kb.get("lines64")
    c = v.get("FETCH_SIZE_bytes" if "read_bytes" in kb or "lines64" in kb else "WRITE_SIZE_bytes")
    if ref and c:
        v["counter_over_known"] = c / ref
print(json.dumps(out, indent=1))
