#!/usr/bin/env python3
"""Per-kernel SQ counter summary of scripts/pmc_sq.sh's passes.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles (MI355X_MICROARCH.md); VALU busy = SQ_ACTIVE_INST_VALU ÷ (launch duration × 2.4 GHz ÷ 4 ×
1024 SIMDs) — a lower bound, the profiled passes clock lower.  usage: pmc_sq_summary.py <dir> [out.json]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(list)
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
for f in sorted(glob.glob(f"{d}/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[name(r)][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name(r)][r["Counter_Name"]] += 1
for f in sorted(glob.glob(f"{d}/p*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k in acc:
    if not k.startswith("ggr::"):
        continue
    v = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    us = sum(dur[k]) / len(dur[k])
    simd_quads = us * 1e-6 * 2.4e9 / 4 * 1024
    out[k] = dict(avg_us_under_pmc=round(us, 2), insts_valu=v.get("SQ_INSTS_VALU"), insts_salu=v.get("SQ_INSTS_SALU"),
                  insts_lds=v.get("SQ_INSTS_LDS"), active_inst_valu_quadcycles=v.get("SQ_ACTIVE_INST_VALU"),
                  wave_quadcycles=v.get("SQ_WAVE_CYCLES"), wait_inst_any_quadcycles=v.get("SQ_WAIT_INST_ANY"),
                  lds_bank_conflict=v.get("SQ_LDS_BANK_CONFLICT"), waves=v.get("SQ_WAVES"),
                  valu_busy_frac_at_2p4GHz=round(v.get("SQ_ACTIVE_INST_VALU", 0) / simd_quads, 3))
    print(f"{k[:44]:44s} {us:8.1f} us  VALU insts {v.get('SQ_INSTS_VALU', 0) / 1e6:8.1f} M  "
          f"valu_busy {out[k]['valu_busy_frac_at_2p4GHz']:.2f}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
