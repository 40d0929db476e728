#!/usr/bin/env python3
"""Static VALU instruction mix of the blend kernels' hot loops, from the gfx950 ISA hipcc emits for them, weighted with
the per-class issue costs measured by tools/valu_peak_bench.hip (profiles/r03_valu_peak.txt, r03_valu_peak2.txt): plain
VALU 2 cycles per wave64 instruction, DPP-modified VALU 4, v_exp / v_rcp / v_permlane*_swap 8, packed fp32 4, vector
compares 4 (4.6 at the nominal clock; a select is a plain instruction), a 32-bit literal operand 2 more.

usage: scripts/valu_mix.py            (needs hipcc; cross-compiles, no GPU)  → JSON on stdout
The mix is taken over the innermost loops only (the survivor / reduction-batch bodies, where > 90 % of the dynamic
instructions are); the average cost per instruction it yields turns PMC's SQ_INSTS_VALU into SIMD cycles."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = {"plain": 2.0, "dpp": 4.0, "trans": 8.0, "permlane_swap": 8.0, "packed": 4.0, "cmp": 4.0, "plain_literal": 4.0}


def classify(line: str):
    m = re.match(r"\s+(v_[a-z0-9_]+)", line)
    if not m:
        return None
    op = m.group(1)
    if op.startswith("v_permlane") and "swap" in op:
        return "permlane_swap"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
        return "trans"
    if "_dpp" in op or " quad_perm" in line or " row_" in line:
        return "dpp"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith("v_cmp"):
        return "cmp"
    if re.search(r"[ ,]0x[0-9a-f]{5,8}\b", line):   # a 32-bit literal (inline constants print as decimals)
        return "plain_literal"
    return "plain"


def kernel_mix(src: str, symbol_re: str):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-munsafe-fp-atomics",
                        "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "ggrt_official_amd", "csrc", src)],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    start = next(i for i, l in enumerate(text) if re.match(symbol_re, l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))   # (an early return is an s_endpgm too)
    body = text[start:end]
    # loop depth of every line from LLVM's block comments ("Depth=N"); keep the deepest loops
    depth, cur, in_label = [], 0, False
    for l in body:
        m = re.search(r"Depth=(\d+)", l)
        if l.startswith(".LBB"):
            cur, in_label = (int(m.group(1)) if m else 0), True
        elif in_label and l.strip().startswith(";") and m:   # "Parent Loop … / This Inner Loop Header: Depth=N" lines
            cur = max(cur, int(m.group(1)))
        elif not l.strip().startswith(";"):
            in_label = False
        depth.append(cur)
    dmax = max(depth)
    mix_all, mix_hot = {}, {}
    for l, d in zip(body, depth):
        c = classify(l)
        if c is None:
            continue
        mix_all[c] = mix_all.get(c, 0) + 1
        if d >= max(dmax - 1, 1):
            mix_hot[c] = mix_hot.get(c, 0) + 1
    n = sum(mix_hot.values())
    avg = sum(COST[c] * k for c, k in mix_hot.items()) / max(n, 1)
    return {"static_valu_in_hot_loops": mix_hot, "static_valu_whole_kernel": mix_all,
            "avg_issue_cycles_per_valu_inst": round(avg, 3)}


if __name__ == "__main__":
    res = {"costs_cycles_per_wave64_inst": COST, "cost_source": "profiles/r03_valu_peak.txt, profiles/r03_valu_peak2.txt (tools/valu_peak_bench.hip)",
           "blend_fwd_kernel": kernel_mix("blend_fwd.hip", r"^_ZN3ggr16blend_fwd_kernelILb1"),
           "blend_bwd_kernel": kernel_mix("blend_bwd.hip", r"^_ZN3ggr16blend_bwd_kernelILb0")}
    json.dump(res, sys.stdout, indent=1)
    print()
