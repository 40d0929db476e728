#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE csv), per launch.
Units and correction follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB
(bytes = value·1024); on gfx950 FETCH_SIZE reports ½ of the bytes of a wide coalesced stream, so the
read side is reported raw AND doubled ("corrected"); WRITE_SIZE is uncalibrated (reported raw).
usage: pmc_summary.py <dir-with-fetch_/write_ csv> [out.json]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
res = collections.defaultdict(dict)
for tag, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(f"{d}/{tag}_counter_collection.csv")
    if not files:
        continue
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    for k in acc:
        res[k][counter + "_KiB_per_launch"] = acc[k] / cnt[k]
        res[k]["launches_" + tag] = cnt[k]
out = {}
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KiB_per_launch", 0)):
    f = v.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024
    w = v.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    out[k] = dict(fetch_bytes_raw=f, fetch_bytes_corrected=2 * f, write_bytes_raw=w,
                  hbm_bytes_per_launch=2 * f + w, launches=v.get("launches_fetch"))
    print(f"{k[:56]:56s} fetch_raw {f / 1e6:9.2f} MB  fetch_x2 {2 * f / 1e6:9.2f} MB  write {w / 1e6:9.2f} MB")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
