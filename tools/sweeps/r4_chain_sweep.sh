#!/bin/bash
# chain-stage variants on the GPU box (round 4, after the pruned scan): window width, residency, block size
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r4f; mkdir -p $out
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],3), 'chain', round(s['chain'],3), 'lpg', round(s.get('chain_lpg',0),3), 'split', d['work_per_step'].get('lpg_split'))"; }
for v in "$@"; do
  if [ $v = cur ]; then unset LRGE_HIP_LIB_AB; else export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_$v.so; fi
  (timeout 600 python -m pytest $root/tests/test_gpu_parity.py -x -q -k "chain or dead_pair" 2>&1 | tail -1 | sed "s/^/$v tests: /") >> $out/sweep.txt
  timeout 300 python $root/bench.py --config c4_dmel_twoset --steps 6 --warmup 2 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 $v" >> $out/sweep.txt
  LRGE_HIP_LPG_MAX=768 timeout 300 python $root/bench.py --config c4_dmel_twoset --steps 6 --warmup 2 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 $v T=768" >> $out/sweep.txt
  timeout 300 python $root/bench.py --steps 2 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5 $v" >> $out/sweep.txt
done
cat $out/sweep.txt
