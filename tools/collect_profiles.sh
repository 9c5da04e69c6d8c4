#!/bin/bash
# copy the summaries of one evidence set (gpurun_out/<tag>/, written by tools/profile_round.sh) into profiles/<tag>_*
# usage (build container, repo root): bash tools/collect_profiles.sh r03_a
set -e
tag=$1; src=gpurun_out/$tag; dst=profiles
cp $src/bench.json $dst/${tag}_bench.json.log
cp $src/bench_under_rocprof.json $dst/${tag}_bench_under_rocprof.json.log
cp $src/stats/s_kernel_stats.csv $dst/${tag}_kernel_stats.csv
cp $src/kernel_time_vs_step.txt $dst/${tag}_kernel_time_vs_step.txt
cp $src/pmc_traffic/traffic.json $dst/${tag}_pmc_traffic.json
cp $src/pmc_traffic/traffic.json $dst/pmc_traffic_latest.json
cp $src/pmc_mfma/mfma_util.json $dst/${tag}_pmc_mfma_util.json
cp $src/stage_times.txt $dst/${tag}_stage_times.txt
cp $src/stage_times_32x30.txt $dst/${tag}_stage_times_32x30.txt
cp $src/hbm_pointwise.txt $dst/${tag}_hbm_pointwise.txt
cp $src/other_configs.jsonl $dst/${tag}_other_configs.jsonl
ls -la $dst/${tag}_*
