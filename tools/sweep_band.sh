R=$GRAFT_REPO_ROOT
for rep in 1 2; do for band in 0 1 2 3 4; do
  out=$R/gpurun_out/abk/band$band; mkdir -p $out
  ( cd /tmp && export TMPDIR=/tmp SWC_LIB=$R/simwhisper_codec_amd/libswc_tun.so SWC_GEMM_BAND=$band && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $R/bench.py --steps 13 --warmup 3 --cpu-baseline off --no-dist --no-timer --no-inflight --other-configs off > $out.json 2> $out.err )
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== band $band"; python3 $R/tools/prof_summary.py $f 16 9 | grep -E "total device|f16s_t, 8|f16s_t, 6|unsigned short, 8|unsigned short, 6"
  rm -rf $out
done; done
