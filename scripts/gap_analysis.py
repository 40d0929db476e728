"""dev: idle time between consecutive kernels of the bench loop, from a rocprofv3 --kernel-trace CSV
(usage: python scripts/gap_analysis.py <kernel_trace.csv>): per kernel name, the mean gap in FRONT of it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
gaps = collections.defaultdict(list)
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    g = (s1 - e0) / 1e3
    if g < 200:       # (skip the breaks between phases of the script)
        gaps[(n0[:38], n1[:38])].append(g)
tot = 0
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"{a:40s} -> {b:40s} n={len(v):4d} mean gap {sum(v)/len(v):7.2f} us")
