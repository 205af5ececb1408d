# in-step A/B of environment settings on the bench line: usage TAG=r05i bash tests/tools/ab_policy.sh "VAR=val VAR2=val2|VAR=val|..."
# (settings separated by '|'; every setting is bracketed by baseline runs on the same box)
T=${TAG:-ab}
mkdir -p gpurun_out/$T
run() { local tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/$T/$tag.json 2> gpurun_out/$T/$tag.err; python -c "
import json,sys; d=json.load(open('gpurun_out/$T/$tag.json')); r=d['roofline']; print('$tag', sys.argv[1:], d['value'], d['ms_per_step'], r['achieved'], r.get('two_stream',{}).get('achieved'))" "$@"; }
i=0
IFS='|' read -ra SETTINGS <<< "$1"
for st in "${SETTINGS[@]}"; do
  run base$i X=1
  run p$i $st
  i=$((i+1))
done
run base$i X=1
