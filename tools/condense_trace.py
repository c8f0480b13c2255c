"""Condense a rocprofv3 *_kernel_trace.csv into dispatch,kernel,grid_threads,duration_ns (launch order) for
profiles/.  Usage: python tools/condense_trace.py <kernel_trace.csv> > profiles/<name>_kernel_trace.csv"""
import csv
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def main(path: str) -> None:
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print("dispatch,kernel,grid_threads,duration_ns")
    for i, r in enumerate(rows, 1):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
        print(f"{i},\"{k}\",{grid},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])}")


if __name__ == "__main__":
    main(sys.argv[1])
