mkdir -p gpurun_out/r05r
run() { tag=$1; shift; env "$@" > /dev/null 2>&1; }
for cfg in "b32_eager|X=1|--batch 32 --launch eager" "b32_graph|X=1|--batch 32 --launch graph" "b32_graph_side|STIL_GRAPH_SIDE=1|--batch 32 --launch graph" \
           "c16_graph|X=1|--variant cardiac --img 128 --batch 16 --launch graph" "c16_graph_side|STIL_GRAPH_SIDE=1|--variant cardiac --img 128 --batch 16 --launch graph" \
           "c64_eager|X=1|--variant cardiac --img 128 --batch 64 --launch eager" "c64_graph|X=1|--variant cardiac --img 128 --batch 64 --launch graph" "c64_graph_side|STIL_GRAPH_SIDE=1|--variant cardiac --img 128 --batch 64 --launch graph" \
           "b64_eager|X=1|--batch 64 --launch eager" "b64_graph_side|STIL_GRAPH_SIDE=1|--batch 64 --launch graph"; do
  IFS='|' read -r tag envs args <<< "$cfg"
  env $envs timeout -k 10 200 python bench.py $args --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05r/$tag.json 2> gpurun_out/r05r/$tag.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/r05r/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['launch'])
except Exception as e: print('$tag', 'ERR', e)"
done
