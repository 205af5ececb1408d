set -e
rm -rf gpurun_out/r01c
mkdir -p gpurun_out/r01c
timeout -k 10 400 python bench.py > gpurun_out/r01c/bench_default.json 2> gpurun_out/r01c/bench_default.err
cat gpurun_out/r01c/bench_default.json | cut -c1-400
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/prof_cmd.json 2> $R/gpurun_out/r01c/prof_cmd.err
# single-stream passes: per-kernel durations / counters without a concurrent weight-gradient kernel on the side stream
export STIL_WGRAD_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/stats_single_stream -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/prof_cmd_single_stream.json 2> $R/gpurun_out/r01c/prof_cmd_single_stream.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01c/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/pmc_fetch.json 2> $R/gpurun_out/r01c/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01c/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/pmc_write.json 2> $R/gpurun_out/r01c/pmc_write.err
unset STIL_WGRAD_STREAM
cd $R
du -sh gpurun_out/r01c/*
set +e
python tests/tools/pmc_traffic.py gpurun_out/r01c/pmc_fetch gpurun_out/r01c/pmc_write gpurun_out/r01c/r01c "gemm_nt_kernel<1, 1, 16, true>" "STIL_WGRAD_STREAM=0 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
find gpurun_out/r01c -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/r01c gpurun_out/r01c/stats/* | head -40
