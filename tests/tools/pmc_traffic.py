"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into the HBM traffic
per launch of every kernel and of the bench line's dominant kernel.
usage: python tests/tools/pmc_traffic.py <fetch_dir> <write_dir> <out_prefix> "<kernel name>" "<command>"
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and gfx950's FETCH_SIZE reports half of a wide
coalesced read (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from stil_tta_amd._lib import source_hash  # noqa: E402


def fold(d, counter):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


fetch_dir, write_dir, prefix, kernel, cmd = sys.argv[1:6]
bench_json = sys.argv[6] if len(sys.argv) > 6 else None      # a bench line of the SAME build: its per-launch algorithmic bytes are stored beside the measured ones
alg = None
if bench_json and os.path.exists(bench_json):
    try:
        rf = json.load(open(bench_json))["roofline"]
        alg = rf["algorithmic_bytes"] if rf["kernel"] == kernel else None
    except Exception:
        alg = None
fe, wr = fold(fetch_dir, "FETCH_SIZE"), fold(write_dir, "WRITE_SIZE")
rows, total = [], 0.0
for k in sorted(set(fe) | set(wr)):
    n = max(fe.get(k, [0])[0], wr.get(k, [0])[0])
    rb = 2.0 * fe.get(k, [0, 0.0])[1] * 1024
    wb = wr.get(k, [0, 0.0])[1] * 1024
    total += rb + wb
    rows.append((k, n, rb / max(1, n), wb / max(1, n), (rb + wb)))
rows.sort(key=lambda r: -r[4])
with open(prefix + "_pmc_traffic_by_kernel.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "read_bytes_per_launch", "write_bytes_per_launch", "total_bytes"])
    w.writerows(rows)
hit = [r for r in rows if r[0].replace("void ", "").startswith(kernel)]
assert hit, f"{kernel} not in the trace"
k, n, rb, wb, _ = hit[0]
json.dump({"kernel": kernel, "kernel_source_sha": source_hash(), "algorithmic_bytes_per_launch": alg, "launches": n, "read_bytes_per_launch": rb, "write_bytes_per_launch": wb, "bytes_per_launch": rb + wb,
           "method": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes of `{cmd}`; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                     "(gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section), averaged over the kernel's launches",
           "all_kernels_total_bytes": total}, open(prefix + "_pmc_traffic.json", "w"), indent=1)
print(open(prefix + "_pmc_traffic.json").read())
