#!/usr/bin/env python3
"""Copies the summaries `scripts/profile_round.sh <tag> [config]` left under gpurun_out/prof_<tag>/ into profiles/ (what
the judge reads and what bench.py's PMC-derived fields come from), named <tag>_*.  With --replace <old_tag> the files of an
earlier tag of the same round are removed (a re-profile after a source change: the stamp in <tag>_meta.json is what
bench.py matches against the library's source hash).
usage: python scripts/install_profiles.py r05_v6 [--replace r05_v5] [--config-name c5p]"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    tag = args[0]
    old = args[args.index("--replace") + 1] if "--replace" in args else None
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    names = ["c3_kernel_stats.txt", "pmc_traffic.json", "pmc_sq.json", "meta.json", "views4_c5p_kernel_stats.txt"]
    for n in names:
        p = os.path.join(src, n)
        if os.path.exists(p):
            out = n if not (n == "c3_kernel_stats.txt" and "c5p" in tag) else "kernel_stats.txt"
            if "c5p" in tag and n == "views4_c5p_kernel_stats.txt":
                continue
            shutil.copy(p, os.path.join(dst, f"{tag}_{out}"))
            print("installed", f"profiles/{tag}_{out}")
    if old:
        for p in glob.glob(os.path.join(dst, f"{old}_*")):
            if any(p.endswith(n) for n in names):
                os.remove(p)
                print("removed", os.path.relpath(p, ROOT))


if __name__ == "__main__":
    main()
