#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs under a directory: one column per run directory, one row per counter
(mean per dispatch of the spmv kernels), plus mean kernel duration."""
import csv, sys, glob, collections, os
cols = []
table = collections.OrderedDict()
for path in sys.argv[1:]:
    vals = {}
    durs = []
    kn = ""
    for f in sorted(glob.glob(path + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(list)
        seen = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "spmv::" not in r["Kernel_Name"] or "fixup" in r["Kernel_Name"]:
                    continue
                kn = r["Kernel_Name"].split("(")[0].replace("void spmv::", "")
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                seen[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for c, v in acc.items():
            vals[c] = sum(v) / len(v)
        durs += list(seen.values())
    vals["~duration_us"] = sum(durs) / max(len(durs), 1) / 1e3
    cols.append((os.path.basename(path.rstrip("/")) + " " + kn[:40], vals))
names = sorted({c for _, v in cols for c in v})
print("%-42s" % "counter" + "".join("%26s" % c[0][-25:] for c in cols))
for n in names:
    print("%-42s" % n + "".join("%26.6g" % c[1].get(n, float("nan")) for c in cols))
