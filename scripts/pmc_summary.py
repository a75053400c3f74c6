"""Average the counters of a rocprofv3 --pmc counter_collection.csv per kernel name substring.
usage: pmc_summary.py file.csv [substr ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2:] or ["march_kernel"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    for s in subs:
        if s in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%-3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
