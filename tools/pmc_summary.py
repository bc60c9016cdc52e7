import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("tk::", "")
    if not k.startswith("k_"):
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    agg[k]["_n_" + r["Counter_Name"]] += 1
for k, v in agg.items():
    print(k, {a: "%.4g" % b for a, b in v.items()})
