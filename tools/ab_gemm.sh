#!/bin/bash
# same-box A/B of two libraries on a few GEMM shapes: new = libswc_hip.so, old = libswc_old.so
cd $GRAFT_REPO_ROOT
cp simwhisper_codec_amd/libswc_hip.so /tmp/new.so
run() {
  for sh in "32768 4096 4096 bf16" "32000 4096 512 bf16" "32000 512 4096 bf16" "16000 3072 768 f16s" "16000 768 3072 f16s"; do
    python tools/time_gemm.py $sh 2>&1 | grep TFLOP
  done
}
for rep in 1 2; do
  echo "== new"; cp /tmp/new.so simwhisper_codec_amd/libswc_hip.so; run
  echo "== new DBG=1"; SWC_GEMM_DBG=1 python tools/time_gemm.py 32768 4096 4096 bf16 2>&1 | grep TFLOP
  echo "== old"; cp simwhisper_codec_amd/libswc_old.so simwhisper_codec_amd/libswc_hip.so; run
  echo "== old DBG=1"; SWC_GEMM_DBG=1 python tools/time_gemm.py 32768 4096 4096 bf16 2>&1 | grep TFLOP
done
