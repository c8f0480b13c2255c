"""Condense the rocprofv3 --pmc passes of tools/gpu_pmc.sh into one CSV for profiles/:
kernel, grid size, counter, dispatches, mean value per dispatch.  For every pass directory the newest
run is used.  Usage: python tools/pmc_summary.py gpurun_out/pmc > profiles/<name>_pmc_summary.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def main(root: str) -> None:
    acc = defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, "*", ""))):
        runs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not runs:
            continue
        newest = max(runs, key=os.path.getmtime)
        for r in csv.DictReader(open(newest)):
            acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL)
    print("kernel,grid_threads,counter,dispatches,mean_value")
    for (k, g, c), v in sorted(acc.items()):
        if not k.startswith("k_"):
            continue
        print(f"\"{k}\",{g},{c},{len(v)},{sum(v) / len(v):.4f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
