import csv, glob, collections, sys
d, kname, nshow = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/*/*counter_collection.csv")[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if kname not in r["Kernel_Name"]:
        continue
    disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp[r["Dispatch_Id"]]["grid"] = r["Grid_Size"]
for i, (k, v) in enumerate(disp.items()):
    if i >= nshow:
        break
    print(k, {a: ("%.4g" % b if isinstance(b, float) else b) for a, b in v.items()})
