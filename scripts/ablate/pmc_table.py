"""python3 scripts/ablate/pmc_table.py <rocprofv3 -d dir> <kernel regex> <reps> NAME [NAME ...]
Splits the dispatches of the matching kernel, in dispatch order, into groups of <reps> (one group per library given
to run_features.py) and prints the mean of every counter per group."""
import csv, glob, os, re, sys
from collections import defaultdict
d, rx, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]); names = sys.argv[4:]
path = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(path)) if re.search(rx, r["Kernel_Name"])]
disp = sorted({int(r["Dispatch_Id"]) for r in rows})
grp = {did: i // reps for i, did in enumerate(disp)}
acc = defaultdict(lambda: defaultdict(list))
for r in rows: acc[grp[int(r["Dispatch_Id"])]][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for g in acc.values() for c in g})
print("variant," + ",".join(ctrs))
for g in sorted(acc):
    print((names[g] if g < len(names) else str(g)) + "," + ",".join(f"{sum(acc[g][c]) / len(acc[g][c]):.6g}" for c in ctrs))
