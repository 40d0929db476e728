#!/usr/bin/env python3
"""Per-PHASE VALU budget of blend_bwd's hot loops, from the gfx950 ISA ("publish the per-slot budget").

The kernel is compiled with line tables (`-gline-tables-only -S`); every VALU instruction of the innermost loops is
attributed — through its `.loc` — to the phase whose `// [budget: NAME]` marker is the last one above its source line
(blend_bwd.hip, blend_common.h), classified and priced as in scripts/valu_mix.py (plain 2 cycles per wave64 instruction,
DPP / compare / literal 4, exp / rcp / permlane-swap 8: tools/valu_peak_bench.hip).  One trip of the reduction loop
handles RB = 8 surviving (quadrant, entry) slots, one trip of the cull loop 64 staged entries.

usage: scripts/valu_budget.py [--dynamic SLOTS STAGED_PAIRS INSTS_VALU]   → JSON on stdout (no GPU needed)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from valu_mix import COST, ROOT, classify  # noqa: E402

SRC = os.path.join(ROOT, "ggrt_official_amd", "csrc")
RB = 8
SURV_GROUP = 4


def markers(path):
    """source line → phase (the last `[budget: NAME]` marker at or above it)"""
    out, cur = {}, "other"
    for i, l in enumerate(open(path), 1):
        m = re.search(r"\[budget:\s*([a-z\-]+)\]", l)
        if m:
            cur = m.group(1)
        out[i] = cur
    return out


def main():
    fwd = "--fwd" in sys.argv          # the forward blend instead: one trip of its survivor loop = SURV_GROUP (4) survivors
    if fwd:
        sys.argv.remove("--fwd")
    src_name = "blend_fwd.hip" if fwd else "blend_bwd.hip"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-munsafe-fp-atomics",
                        "-gline-tables-only", "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, src_name)],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    phase_of = {src_name: markers(os.path.join(SRC, src_name)),
                "blend_common.h": markers(os.path.join(SRC, "blend_common.h"))}
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN3ggr16blend_fwd_kernelILb1" if fwd else r"^_ZN3ggr16blend_bwd_kernelILb0", l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    depth, cur, in_label = [], 0, False
    for l in body:
        m = re.search(r"Depth=(\d+)", l)
        if l.startswith(".LBB"):
            cur, in_label = (int(m.group(1)) if m else 0), True
        elif in_label and l.strip().startswith(";") and m:
            cur = max(cur, int(m.group(1)))
        elif not l.strip().startswith(";") and not l.strip().startswith(".loc"):
            in_label = False
        depth.append(cur)
    dmax = max(depth)
    phase, table = "other", {}
    for l, d in zip(body, depth):
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            f = files.get(int(m.group(1)), "?")
            phase = phase_of.get(f, {}).get(int(m.group(2)), "other:" + f)
            continue
        c = classify(l)
        if c is None or d < max(dmax - 1, 1):
            continue
        t = table.setdefault(phase, {})
        t[c] = t.get(c, 0) + 1
    rb = SURV_GROUP if fwd else RB
    res = {"kernel": "blend_fwd" if fwd else "blend_bwd", "costs_cycles_per_wave64_inst": COST, "slots_per_trip": rb, "phases": {}}
    tot_i = tot_c = 0
    for ph, mix in sorted(table.items()):
        n = sum(mix.values())
        cyc = sum(COST[c] * k for c, k in mix.items())
        tot_i += n
        tot_c += cyc
        res["phases"][ph] = {"static_valu": mix, "insts": n, "issue_cycles": cyc}
    per_batch = [p for p in res["phases"] if p not in ("cull", "stage", "prologue", "epilogue")]
    bi = sum(res["phases"][p]["insts"] for p in per_batch)
    bc = sum(res["phases"][p]["issue_cycles"] for p in per_batch)
    res["per_reduction_trip"] = {"insts": bi, "issue_cycles": bc, "insts_per_slot": round(bi / rb, 1),
                                 "issue_cycles_per_slot": round(bc / rb, 1),
                                 "share_of_cycles": {p: round(res["phases"][p]["issue_cycles"] / bc, 3) for p in per_batch}}
    if "cull" in res["phases"]:
        res["per_cull_trip_64_entries"] = res["phases"]["cull"]
    res["hot_loops_total"] = {"insts": tot_i, "issue_cycles": tot_c}
    if len(sys.argv) >= 5 and sys.argv[1] == "--dynamic":
        slots, staged, insts = (float(x) for x in sys.argv[2:5])
        trips = slots / rb * (1.02 if fwd else 1.045)   # (the last trip of a wave's batch is partly filled)
        culls = staged / 64.0
        est = trips * bi + culls * res["phases"].get("cull", {}).get("insts", 0)
        res["dynamic_check"] = {"surviving_slots": slots, "staged_wave_entry_pairs": staged, "reduction_trips": round(trips),
                                "cull_trips": round(culls), "predicted_insts_valu": round(est), "measured_insts_valu": insts,
                                "predicted_over_measured": round(est / insts, 3)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
