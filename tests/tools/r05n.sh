mkdir -p gpurun_out/r05n
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -x -q -m gpu -k "gemm or tile or baseline_shape or bench_shape_properties or golden" > gpurun_out/r05n/pytest.log 2>&1; echo rc=$? >> gpurun_out/r05n/pytest.log; tail -3 gpurun_out/r05n/pytest.log
TAG=r05n bash tests/tools/ab_policy.sh "STIL_GEMM_PANEL=0|STIL_GEMM_PANEL=4|STIL_GEMM_PANEL=0|STIL_GEMM_PANEL=8" 2>&1 | tail -12
R=$PWD; cd /tmp; export TMPDIR=/tmp; export STIL_WGRAD_STREAM=0
C="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum"
timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/r05n/bench_rd -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r05n/bench_rd.json 2> $R/gpurun_out/r05n/bench_rd.err
