mkdir -p gpurun_out/r05j
timeout -k 10 300 python -m pytest tests/test_gpu_step.py -x -q -m gpu -s -k "cardiac_bench_shape_backward" > gpurun_out/r05j/pytest.log 2>&1; echo rc=$? >> gpurun_out/r05j/pytest.log
grep -E "backward:|passed|failed|rc=" gpurun_out/r05j/pytest.log | cut -c1-300
for v in "cardiac16 --variant cardiac --img 128 --batch 16" "cardiac16_eager --variant cardiac --img 128 --batch 16 --launch eager" "cardiac64 --variant cardiac --img 128 --batch 64" "b32 --batch 32" "b32_graph --batch 32 --launch graph" "saint --variant saint" "native128 --img 128 --ncat 4 --ncon 13"; do
  set -- $v; tag=$1; shift
  timeout -k 10 200 python bench.py "$@" --no-cpu-baseline > gpurun_out/r05j/bench_$tag.json 2> gpurun_out/r05j/bench_$tag.err
  python -c "
import json; d=json.load(open('gpurun_out/r05j/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['launch'], d['config']['split_k'])"
done
