#!/bin/bash
# sweep LRGE_HIP_LPG_MAX on a synth config: tools/sweeps/sweep_cfg.sh <config> <preset> <T...>
cfg=$1; preset=$2; shift 2
for m in "$@"; do
  if [ "$m" = auto ]; then unset LRGE_HIP_LPG_MAX; else export LRGE_HIP_LPG_MAX=$m; fi
  LRGE_HIP_VERBOSE=1 timeout 900 python tools/run_config.py $cfg --preset $preset --check 0 --repeat 1 2> gpurun_out/sweep_$m.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$m', d['step_s'], d['stage_ms']['chain'], d['stage_ms']['chain_lpg'])"
  grep "lrge_hip" gpurun_out/sweep_$m.err | head -3
done
