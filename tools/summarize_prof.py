#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` (or a `--pmc` `*_counter_collection.csv`) into a
short markdown table with abbreviated kernel names, for committing under profiles/.

    python tools/summarize_prof.py gpurun_out/prof/**/X_kernel_stats.csv [--top 25] > profiles/r01_x.md
    python tools/summarize_prof.py --pmc gpurun_out/pmc/**/X_counter_collection.csv --kernel spmm_wide
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name, width=70):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"(Cijk_[A-Za-z]+_[A-Za-z]+_[A-Z]+).*?(MT\d+x\d+x\d+).*?(MI\d+x\d+x\d+)", name)
    if m:
        return "hipBLASLt " + "_".join(m.groups())
    name = re.sub(r"\(.*$", "", name)             # drop the argument list
    if len(name) > width:
        name = name[:width - 3] + "..."
    return name


def kernel_stats(path, top):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows[:top]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
              f"{float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
              f"{float(r['MaxNs'])/1e6:.4f} | {100*float(r['TotalDurationNs'])/total:.2f} |")
    rest = rows[top:]
    if rest:
        t = sum(float(r["TotalDurationNs"]) for r in rest)
        print(f"| ({len(rest)} more kernels) | {sum(int(r['Calls']) for r in rest)} | {t/1e6:.3f} "
              f"| | | | {100*t/total:.2f} |")
    print(f"\nTotal kernel time {total/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches.")


def pmc(path, kernel):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Name") or ""
        if kernel and kernel not in name:
            continue
        acc[short(name)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("| kernel | counter | dispatches | mean per dispatch | min | max |")
    print("|---|---|---:|---:|---:|---:|")
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            print(f"| `{k}` | {c} | {len(v)} | {sum(v)/len(v):.6g} | {min(v):.6g} | {max(v):.6g} |")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--kernel", default="")
    a = ap.parse_args()
    (pmc(a.csv, a.kernel) if a.pmc else kernel_stats(a.csv, a.top))
