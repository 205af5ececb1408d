mkdir -p gpurun_out/r05o
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r05o/fp32_$i.json 2> gpurun_out/r05o/fp32_$i.err
timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --precision bf16x3 > gpurun_out/r05o/b3_$i.json 2> gpurun_out/r05o/b3_$i.err
done
python - <<'PY'
import json
for t in ("fp32_1","b3_1","fp32_2","b3_2"):
    try:
        d=json.load(open(f"gpurun_out/r05o/{t}.json")); r=d["roofline"]; print(t, d["value"], d["ms_per_step"], d["loss"], r["kernel"], r["achieved"], {k: (v["ms"], v["tflops"]) for k, v in list(r["all_gemm_nt"].items())[:6]})
    except Exception as e: print(t, "ERR", e); print(open(f"gpurun_out/r05o/{t}.err").read()[-1500:])
PY
STIL_PRECISION=bf16x3 timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r05o/pytest_b3.log 2>&1; echo rc=$? >> gpurun_out/r05o/pytest_b3.log; tail -25 gpurun_out/r05o/pytest_b3.log | cut -c1-220
