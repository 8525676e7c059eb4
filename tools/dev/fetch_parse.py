"""Developer script: per-kernel means of the counters in a rocprofv3 counter_collection CSV written under tools/dev/fetch_vs_lib.py
(three launches per shape, in shape order).  python tools/dev/fetch_parse.py <csv> <counter> [<counter> ...]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
want = set(sys.argv[2:])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] in want and ("gemm_nt256" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]):
        acc[r["Counter_Name"]][r["Kernel_Name"].split("(")[0][:28]].append(float(r["Counter_Value"]))
print("shapes (N x K at 102 000 rows): 1280x1280, 3840x1280, 5120x1280, 1280x5120")
for cn in sys.argv[2:]:
    for k, v in acc[cn].items():
        per = [sum(v[3 * i:3 * i + 3]) / 3 for i in range(len(v) // 3)]
        print(f"{cn:36s} {k:30s}", [f"{x:.4g}" for x in per])
