"""Per-kernel launches and time of ONE step (the last but one between two adam_kernel launches) of a rocprofv3 --kernel-trace CSV of
bench.py, GEMM launches also by grid size.   usage: python tests/tools/step_histogram.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel")]
st = ev[adam[-3] + 1:adam[-2] + 1]
c = collections.defaultdict(lambda: [0, 0])
g = collections.defaultdict(lambda: [0, 0])
for s, e, n, grid in st:
    c[n][0] += 1; c[n][1] += e - s
    if "gemm" in n:
        k = (n.replace("void gemm_", "").replace("_kernel", ""), grid)
        g[k][0] += 1; g[k][1] += e - s
tot = sum(v[1] for v in c.values())
print(f"{len(st)} launches, {(st[-1][1] - st[0][0]) / 1e6:.3f} ms wall, {tot / 1e6:.3f} ms of kernel time; GEMM {sum(v[1] for k, v in c.items() if 'gemm' in k) / 1e6:.3f} ms")
for n, (k, t) in sorted(c.items(), key=lambda x: -x[1][1])[:60]:
    print(f"{t / 1e3:8.1f} us {k:4d}  avg {t / k / 1e3:6.1f}  {n}")
print("GEMM launches by (kernel, workgroups):")
for (n, grid), (k, t) in sorted(g.items(), key=lambda x: -x[1][1])[:40]:
    print(f"{t / 1e3:8.1f} us {k:4d} avg {t / k / 1e3:6.1f} grid {grid:6d}  {n}")
