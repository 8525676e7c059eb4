import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and ("gemm_nt256" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]):
        acc[r["Kernel_Name"].split("(")[0][:40]].append(float(r["Counter_Value"]))
M = 102000
alg = [(M * k + n * k) * 2 / 1e6 for n, k in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120))]
for k, v in acc.items():
    # three launches per shape, in shape order
    per = [sum(v[3 * i:3 * i + 3]) / 3 for i in range(len(v) // 3)]
    print(k, sys.argv[2], "per launch (counter units):", [round(x) for x in per], " operand MB:", [round(x) for x in alg])
