"""Sum rocprofv3 --pmc counter CSVs per kernel:  python scripts/pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "score_kernel"
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:32s} {acc[k] / max(n[k], 1):16.1f}  per launch ({n[k]} launches)")
