set -e
rm -rf gpurun_out/${TAG:-r01d}
mkdir -p gpurun_out/${TAG:-r01d}
timeout -k 10 400 python bench.py > gpurun_out/${TAG:-r01d}/bench_default.json 2> gpurun_out/${TAG:-r01d}/bench_default.err
cat gpurun_out/${TAG:-r01d}/bench_default.json | cut -c1-400
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG:-r01d}/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG:-r01d}/prof_cmd.json 2> $R/gpurun_out/${TAG:-r01d}/prof_cmd.err
# single-stream passes: per-kernel durations / counters without a concurrent weight-gradient kernel on the side stream
export STIL_WGRAD_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG:-r01d}/stats_single_stream -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG:-r01d}/prof_cmd_single_stream.json 2> $R/gpurun_out/${TAG:-r01d}/prof_cmd_single_stream.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG:-r01d}/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG:-r01d}/pmc_fetch.json 2> $R/gpurun_out/${TAG:-r01d}/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG:-r01d}/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG:-r01d}/pmc_write.json 2> $R/gpurun_out/${TAG:-r01d}/pmc_write.err
unset STIL_WGRAD_STREAM
cd $R
du -sh gpurun_out/${TAG:-r01d}/*
set +e
python tests/tools/pmc_traffic.py gpurun_out/${TAG:-r01d}/pmc_fetch gpurun_out/${TAG:-r01d}/pmc_write gpurun_out/${TAG:-r01d}/${TAG:-r01d} "gemm_nt_kernel<1, 1, 16, true>" "STIL_WGRAD_STREAM=0 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
find gpurun_out/${TAG:-r01d} -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/${TAG:-r01d} gpurun_out/${TAG:-r01d}/stats/* | head -40
# the other measurement shapes of SURVEY.md 8(d)
T=${TAG:-r01d}
timeout -k 10 200 python bench.py --variant saint --no-cpu-baseline > gpurun_out/$T/bench_saint.json 2>/dev/null
timeout -k 10 200 python bench.py --img 128 --ncat 4 --ncon 13 --no-cpu-baseline > gpurun_out/$T/bench_native128.json 2>/dev/null
timeout -k 10 200 python bench.py --variant cardiac --img 128 --batch 64 --no-cpu-baseline > gpurun_out/$T/bench_cardiac.json 2>/dev/null
timeout -k 10 200 python bench.py --batch 32 --no-cpu-baseline > gpurun_out/$T/bench_b32.json 2>/dev/null
for f in saint native128 cardiac b32; do cut -c1-200 gpurun_out/$T/bench_$f.json; done
