"""Summarise rocprofv3 --pmc output (sqlite *_results.db, view counters_collection): per kernel, mean counter value."""
import collections, glob, sqlite3, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    for k, n, v, d in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        k = k.split("(")[0][:44]
        acc[k][n].append(v)
        acc[k]["_duration_ns"].append(d)
for k, d in sorted(acc.items()):
    print(k)
    for cn, v in sorted(d.items()):
        print(f"   {cn:32s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
