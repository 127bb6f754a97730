"""Condense rocprofv3 CSV output (gpurun_out/...) into the small summaries kept under profiles/.

  python scripts/summarize_prof.py kernel_stats <dir> <out.txt> [regex]
  python scripts/summarize_prof.py pmc <dir> <out.txt> [regex]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[-1]                      # the newest run when several were merged into one directory


def kernel_stats(d, out, rx):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in ns)\n")
        f.write(f"# source: {d}; all kernels: {len(rows)}, total {total:.0f} ns\n")
        f.write("name,calls,total_ns,average_ns,min_ns,max_ns,stddev,percent_of_all_gpu_time\n")
        for r in rows:
            if re.search(rx, r["Name"]):
                name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                f.write(f'{name},{r["Calls"]},{r["TotalDurationNs"]},{r["AverageNs"]},{r["MinNs"]},'
                        f'{r["MaxNs"]},{r["StdDev"]},{r["Percentage"]}\n')
    print(open(out).read())


def pmc(d, out, rx):
    path = find(d, "counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        if re.search(rx, r["Kernel_Name"]):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc summary: per-dispatch mean of each counter\n")
        f.write(f"# source: {d}\n")
        f.write("kernel,counter,dispatches,mean,min,max\n")
        for k, cs in acc.items():
            for c, v in sorted(cs.items()):
                f.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g},{min(v):.6g},{max(v):.6g}\n")
    print(open(out).read())


if __name__ == "__main__":
    mode, d, out = sys.argv[1:4]
    rx = sys.argv[4] if len(sys.argv) > 4 else r"af_|af::"
    {"kernel_stats": kernel_stats, "pmc": pmc}[mode](d, out, rx)
