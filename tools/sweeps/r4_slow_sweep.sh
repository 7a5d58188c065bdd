#!/bin/bash
# k_chain_lpg slow-path thresholds after the pruned scan (round 4): when does a lane give its group up to k_chain_hw_redo?
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r4m; mkdir -p $out
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'chain', round(s['chain'],2), 'lpg', round(s.get('chain_lpg',0),2))"; }
for cfg in "" "--config c4_dmel_twoset"; do
for e in "4 1024" "0 0" "16 4096" "64 16384" "1000000 100000000"; do
  set -- $e
  LRGE_HIP_LPG_SLOW_ENTRIES=$1 LRGE_HIP_LPG_SLOW_BUDGET=$2 timeout 300 python $root/bench.py $cfg --steps 2 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "${cfg:-c5} entries=$1 budget=$2" >> $out/slow.txt
done; done
cat $out/slow.txt
