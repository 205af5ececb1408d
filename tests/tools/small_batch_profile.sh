# Kernel traces of the small-batch shapes (B = 32 at 224 px, cardiac 16 at 128 px) under hipGraph replay:
# per-kernel stats + timeline (busy / idle / overlap) of one step.   usage: TAG=r05s bash tests/tools/small_batch_profile.sh
set -e
T=${TAG:-r05s}
R=$PWD
rm -rf gpurun_out/$T; mkdir -p gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/$name -o run -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/$T/$name.json 2> $R/gpurun_out/$T/$name.err
  cut -c1-160 $R/gpurun_out/$T/$name.json
  local tr=$(find $R/gpurun_out/$T/$name -name "*kernel_trace.csv" | head -1)
  python3 $R/tests/tools/timeline.py $tr 6 > $R/gpurun_out/$T/${name}_timeline.txt 2>&1 || true
  python3 $R/tests/tools/step_histogram.py $tr > $R/gpurun_out/$T/${name}_step_histogram.txt 2>&1 || true
  head -6 $R/gpurun_out/$T/${name}_timeline.txt
  cp $(find $R/gpurun_out/$T/$name -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$T/${name}_kernel_stats.csv
  find $R/gpurun_out/$T/$name -name "*kernel_trace.csv" -size +20M -delete
}
run b32_graph --batch 32 --launch graph
run c16_graph --variant cardiac --img 128 --batch 16 --launch graph
