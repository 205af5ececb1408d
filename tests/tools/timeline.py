"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-queue busy time, overlap between queues and idle gaps of the
last step.  usage: python tests/tools/timeline.py <kernel_trace.csv> [steps_in_trace]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:40]) for r in rows))
# last step = kernels after the (n-1)/n-th adam_kernel
adam = [i for i, e in enumerate(ev) if e[3].startswith("adam_kernel")]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # 0 = last step (bench.py runs it single-stream under HIP events)
lo = adam[-2 - back] + 1
hi = adam[-1 - back] + 1
ev = ev[lo:hi]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print(f"last step: {len(ev)} kernels, {(t1 - t0) / 1e6:.3f} ms")
byq = collections.defaultdict(float)
for s, e, q, n in ev:
    byq[q] += (e - s)
for q, v in byq.items():
    print(f"  queue {q}: sum of kernel durations {v / 1e6:8.3f} ms")
# sweep: time with 0 / 1 / >=2 kernels in flight
pts = []
for s, e, q, n in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
lvl, last, acc = 0, t0, collections.defaultdict(float)
for t, d in pts:
    acc[min(lvl, 2)] += t - last
    last = t; lvl += d
print("  time with 0 kernels in flight: %.3f ms, 1: %.3f ms, >=2: %.3f ms" % (acc[0] / 1e6, acc[1] / 1e6, acc[2] / 1e6))
# biggest idle gaps
gaps = []
end = ev[0][1]
for s, e, q, n in ev[1:]:
    if s > end:
        gaps.append((s - end, n))
    end = max(end, e)
gaps.sort(reverse=True)
print("  largest idle gaps (us, next kernel):", [(round(g / 1e3, 1), n) for g, n in gaps[:8]])
print("  idle total: %.3f ms in %d gaps" % (sum(g for g, _ in gaps) / 1e6, len(gaps)))
# per queue: busy time, and how its busy time splits by what the OTHER queues run meanwhile (GEMM / other kernel / nothing)
qs = sorted(byq)
iv = {q: sorted((s, e, n) for s, e, qq, n in ev if qq == q) for q in qs}
def classify(n):
    return "gemm" if "gemm" in n or "attn" in n else "other"
marks = []
for s, e, q, n in ev:
    marks.append((s, 0, q, classify(n))); marks.append((e, 1, q, classify(n)))
marks.sort(key=lambda m: (m[0], -m[1]))
cur = {}
last = t0
acc2 = collections.defaultdict(float)
for t, kind, q, c in marks:
    if t > last and cur:
        for q1, c1 in cur.items():
            others = [c2 for q2, c2 in cur.items() if q2 != q1]
            o = "alone" if not others else ("with gemm" if "gemm" in others else "with other")
            acc2[(q1, c1, o)] += t - last
    last = t
    if kind == 0:
        cur[q] = c
    else:
        cur.pop(q, None)
print("  queue, kernel class, what the other queues run meanwhile: ms")
for k in sorted(acc2):
    print(f"    queue {k[0]} {k[1]:6s} {k[2]:11s} {acc2[k] / 1e6:8.3f}")
