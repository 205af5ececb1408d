"""Fold a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE, one run of the bench command)
into the MFMA utilisation of every GEMM kernel.
usage: python tests/tools/mfma_util.py <pmc_dir> <out_json> "<command>"
  mfma_busy_of_cu_busy  = SQ_VALU_MFMA_BUSY_CYCLES / 4 / SQ_BUSY_CU_CYCLES   (MFMA-pipe busy cycles are counted per SIMD,
                          4 SIMDs per CU; CU-busy cycles per CU; both summed over the chip)
  mfma_busy_of_gpu_time = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)  (includes ramp-up / tail)"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from stil_tta_amd._lib import source_hash  # noqa: E402

pmc_dir, out, cmd = sys.argv[1:4]
agg = {}
for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not (k.startswith("gemm_nt_kernel") or k.startswith("gemm_tn_kernel") or k.startswith("attn_")):
            continue
        a = agg.setdefault(k, {})
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
res = {}
for k, a in agg.items():
    if not all(c in a for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE")):
        continue
    n = a["SQ_VALU_MFMA_BUSY_CYCLES"][0]
    mf, cu, gui = (a[c][1] for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"))
    res[k] = dict(launches=n, mfma_busy_of_cu_busy=round(mf / 4.0 / cu, 4) if cu else None,
                  mfma_busy_of_gpu_time=round((mf / 1024.0) / (gui / 8.0), 4) if gui else None,
                  SQ_VALU_MFMA_BUSY_CYCLES=mf, SQ_BUSY_CU_CYCLES=cu, GRBM_GUI_ACTIVE=gui)
json.dump({"kernel_source_sha": source_hash(), "method": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- `{cmd}` (counters only, no trace domains); "
                     "sums over every launch of the kernel", "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]))},
          open(out, "w"), indent=1)
print(open(out).read()[:3000])
