#!/bin/bash
# same-box per-kernel comparison of library builds under rocprofv3 --kernel-trace --stats (serial steps only):
# tools/ab_kernels.sh <lib tag> <lib tag> ...   ->  gpurun_out/abk/<tag>.txt (top kernels, per step)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/abk
mkdir -p $out
for tag in "$@"; do
  lib=$R/simwhisper_codec_amd/libswc_$tag.so
  ( cd /tmp && export TMPDIR=/tmp SWC_LIB=$lib && rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -o s -- python3 $R/bench.py --steps 13 --warmup 3 --cpu-baseline off --no-dist --no-timer --no-inflight --other-configs off > $out/$tag.json 2> $out/$tag.err )
  f=$(find $out/$tag -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/prof_summary.py $f 16 16 > $out/$tag.txt
  echo "== $tag"; cat $out/$tag.txt
done
