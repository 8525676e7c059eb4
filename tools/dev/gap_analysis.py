"""Developer script (GPU box): where the GPU idles inside one steady-state step.  Input: a rocprofv3 --kernel-trace CSV of a
bench.py run; the step is delimited by consecutive `logmel_kernel` launches (one per step).
    python tools/dev/gap_analysis.py <kernel_trace.csv> [step_index_from_end=2]"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("logmel_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = marks[-k - 1], marks[-k]
step = rows[a:b]
wall = step[-1][1] - step[0][0]
wall_to_next = rows[b][0] - step[0][0]
busy = sum(e - s for s, e, _ in step)
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(step, step[1:] + [rows[b]]):
    gaps.append((max(0, s1 - e0), n0, n1))
idle = sum(g for g, _, _ in gaps)
print(f"launches {len(step)}  step (first start -> next step's first start) {wall_to_next / 1e6:.2f} ms  kernels busy {busy / 1e6:.2f} ms  idle {idle / 1e6:.2f} ms")
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-30us" if g < 30000 else "30-100us" if g < 100000 else ">100us"] += g
cnt = collections.Counter()
for g, _, _ in gaps:
    cnt["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-30us" if g < 30000 else "30-100us" if g < 100000 else ">100us"] += 1
for kk in ("<2us", "2-5us", "5-10us", "10-30us", "30-100us", ">100us"):
    print(f"  gaps {kk:9s} count {cnt[kk]:5d}  total {hist[kk] / 1e6:7.3f} ms")
by = collections.defaultdict(lambda: [0, 0])
for g, n0, n1 in gaps:
    by[(n0, n1)][0] += g; by[(n0, n1)][1] += 1
print("idle by (previous kernel -> next kernel):")
for (n0, n1), (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {g / 1e6:7.3f} ms  {c:5d} x {g / c / 1e3:7.1f} us   {n0} -> {n1}")
print("largest single gaps:")
for g, n0, n1 in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:8.1f} us  {n0} -> {n1}")
