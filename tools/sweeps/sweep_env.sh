#!/bin/bash
# sweep one environment variable on a synth config: tools/sweeps/sweep_env.sh <config> <VAR> <values...>
cfg=$1; var=$2; shift 2
for m in "$@"; do
  export $var=$m
  timeout 900 python tools/run_config.py $cfg --check 0 --repeat 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$var=$m', d['step_s'], 'chain', s['chain'], 'lpg', s.get('chain_lpg'))"
done
