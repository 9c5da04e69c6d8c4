#!/bin/bash
# same-box A/B on the hot-path shape list (tools/bench_gemm.py): new = libswc_hip.so, old = libswc_old.so
cd $GRAFT_REPO_ROOT
cp simwhisper_codec_amd/libswc_hip.so /tmp/new.so
for rep in 1 2; do
  echo "== new"; cp /tmp/new.so simwhisper_codec_amd/libswc_hip.so; python tools/bench_gemm.py ${1:-bf16} 2>&1 | grep TFLOP
  echo "== old"; cp simwhisper_codec_amd/libswc_old.so simwhisper_codec_amd/libswc_hip.so; python tools/bench_gemm.py ${1:-bf16} 2>&1 | grep TFLOP
done
cp /tmp/new.so simwhisper_codec_amd/libswc_hip.so
