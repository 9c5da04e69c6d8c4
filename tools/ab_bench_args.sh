#!/bin/bash
# same-box whole-step A/B for library builds with extra bench.py arguments: tools/ab_bench_args.sh "<bench args>" <tag> <tag> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
args=$1; shift
for rep in 1 2; do
  for tag in "$@"; do
    lib=$R/simwhisper_codec_amd/libswc_$tag.so
    echo "== $tag"
    SWC_LIB=$lib python bench.py --steps 10 --warmup 3 --cpu-baseline off --no-dist --no-inflight --other-configs off $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], r.get('kernel'), r.get('achieved'), {k:v['TFLOP/s'] for k,v in r.get('other',{}).items()})"
  done
done
