# One profiling round of the bench command on the GPU box: bench line, rocprofv3 kernel stats (default = two streams, and
# single stream), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA utilisation (SQ counters) of the GEMM
# kernels.  usage: TAG=r02a bash tests/tools/profile_round.sh   -> gpurun_out/$TAG/ (copy what is to be judged to profiles/)
set -e
T=${TAG:-r02}
rm -rf gpurun_out/$T
mkdir -p gpurun_out/$T
timeout -k 10 500 python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
cut -c1-400 gpurun_out/$T/bench_default.json
KERNEL=$(python -c "import json,sys; print(json.load(open('gpurun_out/$T/bench_default.json'))['roofline']['kernel'])")
echo "dominant kernel: $KERNEL"
R=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/prof_cmd.json 2> $R/gpurun_out/$T/prof_cmd.err
# single-stream passes: per-kernel durations / counters without a concurrent kernel on the side stream
export STIL_WGRAD_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats_single_stream -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/prof_cmd_single_stream.json 2> $R/gpurun_out/$T/prof_cmd_single_stream.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/pmc_fetch.json 2> $R/gpurun_out/$T/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/pmc_write.json 2> $R/gpurun_out/$T/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/pmc_mfma -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/pmc_mfma.json 2> $R/gpurun_out/$T/pmc_mfma.err
unset STIL_WGRAD_STREAM
cd $R
set +e
python tests/tools/pmc_traffic.py gpurun_out/$T/pmc_fetch gpurun_out/$T/pmc_write gpurun_out/$T/$T "$KERNEL" "STIL_WGRAD_STREAM=0 $CMD" gpurun_out/$T/bench_default.json | head -14
python tests/tools/mfma_util.py gpurun_out/$T/pmc_mfma gpurun_out/$T/${T}_mfma_util.json "STIL_WGRAD_STREAM=0 bench.py --steps 2 --warmup 1 --no-cpu-baseline" | head -40
find gpurun_out/$T -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/$T -name "*counter_collection.csv" -size +20M -delete
for d in stats stats_single_stream; do f=$(find gpurun_out/$T/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$T/${T}_bench_kernel_stats_$d.csv; done
ls -la gpurun_out/$T | head -40
if [ -n "$OTHER_SHAPES" ]; then   # the other measurement shapes of SURVEY.md 8(d)
  timeout -k 10 200 python bench.py --variant saint --no-cpu-baseline > gpurun_out/$T/bench_saint.json 2>/dev/null
  timeout -k 10 200 python bench.py --img 128 --ncat 4 --ncon 13 --no-cpu-baseline > gpurun_out/$T/bench_native128.json 2>/dev/null
  timeout -k 10 200 python bench.py --variant cardiac --img 128 --batch 64 --no-cpu-baseline > gpurun_out/$T/bench_cardiac.json 2>/dev/null
  timeout -k 10 200 python bench.py --batch 32 --no-cpu-baseline > gpurun_out/$T/bench_b32.json 2>/dev/null
  timeout -k 10 200 python bench.py --variant cardiac --img 128 --batch 16 --no-cpu-baseline > gpurun_out/$T/bench_cardiac16.json 2>/dev/null
  for f in saint native128 cardiac b32 cardiac16; do cut -c1-200 gpurun_out/$T/bench_$f.json; done
fi
