# rocprofv3 kernel stats (two-stream default and single-stream) + bench line of one bench variant.
# usage: TAG=r04_cardiac ARGS="--variant cardiac --img 128 --batch 64" bash tests/tools/profile_variant.sh   -> gpurun_out/$TAG/
set -e
T=${TAG:?}; R=$PWD
rm -rf gpurun_out/$T; mkdir -p gpurun_out/$T
timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cut -c1-160 gpurun_out/$T/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats -o run -- python3 $R/bench.py $ARGS --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$T/prof_cmd.json 2> $R/gpurun_out/$T/prof_cmd.err
export STIL_WGRAD_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats_single_stream -o run -- python3 $R/bench.py $ARGS --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$T/prof_cmd_single_stream.json 2> $R/gpurun_out/$T/prof_cmd_single_stream.err
unset STIL_WGRAD_STREAM
cd $R
find gpurun_out/$T -name "*kernel_trace.csv" -size +20M -delete
for d in stats stats_single_stream; do f=$(find gpurun_out/$T/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$T/${T}_kernel_stats_$d.csv; done
ls gpurun_out/$T
