"""Sums rocprofv3 --pmc counter CSVs per kernel and counter: python summarize_pmc.py <dir with pass subdirs>."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("ptd::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add((f, r["Dispatch_Id"]))
for k in sorted(tot):
    print(f"{k}  dispatches={len(calls[k])//max(1,len(set(f for f,_ in calls[k])))}")
    for c in sorted(tot[k]):
        print(f"    {c:36s} {tot[k][c]:.6g}")
