#!/bin/bash
# Board power / clocks while the metric step runs back to back (rocm-smi polled beside a long bench run): is the step
# power-limited?  usage: bash tools/power_trace.sh [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
steps=${1:-3000}
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30
python3 bench.py --steps $steps --warmup 5 --cpu-baseline off --no-dist --no-inflight --other-configs off --no-timer > /tmp/pt_bench.json 2>/dev/null &
pid=$!
sleep 8    # imports, model build
for i in $(seq 1 40); do
  printf "t=%s " "$(date +%s.%N | cut -c7-14)"
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done
wait $pid
python3 -c "
import json; d=json.loads(open('/tmp/pt_bench.json').readline()); print('bench:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms/step')"
