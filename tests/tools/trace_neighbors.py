"""Who are the neighbours (previous / next dispatch in start order) of the dispatches whose name contains PATTERN?
usage: python tests/tools/trace_neighbors.py <dir with *_kernel_trace.csv> [PATTERN=copyBuffer]     (measurement tool)"""
import collections, csv, glob, os, sys
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "copyBuffer")
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0][-60:]
prev, nxt, sizes = collections.Counter(), collections.Counter(), collections.Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        prev[short(rows[i - 1]["Kernel_Name"]) if i else "-"] += 1
        nxt[short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"] += 1
        sizes[(r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Queue_Id", "?"))] += 1
print("matches", sum(prev.values()), "of", len(rows))
print("previous:"); [print(f"  {n:5d} {k}") for k, n in prev.most_common(15)]
print("next:"); [print(f"  {n:5d} {k}") for k, n in nxt.most_common(15)]
print("grid / workgroup / queue:"); [print(f"  {n:5d} {k}") for k, n in sizes.most_common(10)]
m = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
if m:
    mr = list(csv.DictReader(open(m[0])))
    c = collections.Counter((r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))) for r in mr)
    print("memory copies:", len(mr)); [print(f"  {n:5d} {k}") for k, n in c.most_common(20)]
