# SQ counter passes over tests/tools/gemm_pmc.py (one rocprofv3 run per counter group); output: gpurun_out/pmcg/<group>/
R=$PWD
mkdir -p gpurun_out/pmcg
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcg/g$i -o run -- python3 $R/tests/tools/gemm_pmc.py > $R/gpurun_out/pmcg/g$i.log 2>&1 || echo "group $i failed"
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmcg/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_kernel" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0][-28:], int(r["Grid_Size"]))
        a = agg.setdefault(key, {})
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1; c[1] += float(r["Counter_Value"])
for key, a in agg.items():
    print(key)
    for k, (n, v) in a.items():
        print(f"    {k:28s} {v / n:16.1f}")
PY
