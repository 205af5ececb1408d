set -e
mkdir -p gpurun_out/r05e
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r05e/$tag.json 2> gpurun_out/r05e/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/r05e/$tag.json')); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], r['achieved'], r.get('two_stream',{}).get('achieved'))"; }
run base0 X=1
run k2048_bk16 STIL_GEMM_POLICY=k2048:200
run k1024_bk16 STIL_GEMM_POLICY=k1024:200
run base1 X=1
run k2048_v22 STIL_GEMM_POLICY=k2048:22
run k1024_v22 STIL_GEMM_POLICY=k1024:22
run k2048_v22bk32 STIL_GEMM_POLICY=k2048:122
run base2 X=1
run k512_bk16 STIL_GEMM_POLICY=k512:200
