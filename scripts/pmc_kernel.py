"""Per-launch averages of every counter in a rocprofv3 --pmc output directory for kernels matching a substring.
usage: python scripts/pmc_kernel.py <dir> <kernel substring>"""
import collections, glob, sqlite3, sys
acc, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if sys.argv[2] in k:
            acc[c] += float(v); n[c] += 1
for c in sorted(acc):
    print(f"{c:32s} {acc[c] / n[c]:18.1f}  per launch ({n[c]} launches)")
