"""Condense the rocprofv3 --pmc passes of tools/gpu_pmc.sh into one CSV for profiles/:
kernel, grid size, counter, dispatches, mean value per dispatch, mean duration (ns) of those dispatches IN THAT PASS
(counter passes run slower than an unprofiled launch; cycles of a pass must be divided by that pass's own duration).
For every pass directory the newest run is used.  Usage: python tools/pmc_summary.py gpurun_out/pmc > profiles/<name>_pmc_summary.csv"""
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
    dur = defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, "*", ""))):
        runs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not runs:
            continue
        newest = max(runs, key=os.path.getmtime)
        for r in csv.DictReader(open(newest)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), r["Counter_Name"])
            acc[key].append(float(r["Counter_Value"]))
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL)
    print("kernel,grid_threads,counter,dispatches,mean_value,mean_duration_ns")
    for (k, g, c), v in sorted(acc.items()):
        if not k.startswith("k_"):
            continue
        d = dur.get((k, g, c))
        print(f"\"{k}\",{g},{c},{len(v)},{sum(v) / len(v):.4f},{(sum(d) / len(d)) if d else 0:.0f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
