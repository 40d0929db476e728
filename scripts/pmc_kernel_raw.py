#!/usr/bin/env python3
"""dev: every counter of scripts/pmc_sq.sh's passes for the kernels whose name contains <pattern> (averages per launch).
usage: pmc_kernel_raw.py <dir> <pattern>"""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
acc, cnt, dur = collections.defaultdict(float), collections.Counter(), []
for f in sorted(glob.glob(f"{d}/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for f in sorted(glob.glob(f"{d}/p*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{pat}: avg {sum(dur) / max(len(dur), 1):.1f} us over {len(dur)} launches")
for c in sorted(acc):
    print(f"  {c:28s} {acc[c] / cnt[c]:16.0f}")
