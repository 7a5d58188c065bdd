#!/bin/bash
# round 5: where the split between k_chain_lpg and k_chain_hw should sit now that the scan is pruned (choose_chain_split still prices a
# k_chain_lpg step at round 3's 4.1 us): LRGE_HIP_LPG_MAX pinned against the chooser, C5 ava-pb / ava-ont / C4
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; print('$1', round(d['ms_per_step'],1), 'chain', round(s['chain'],1), 'lpg', round(s.get('chain_lpg',0),1), 'split', w.get('lpg_split'), 'lpg_anchors', int(w.get('lpg_anchors',0)), 'of', int(w.get('chain_anchors',0)))"; }
for t in auto 3072 6144 8192 auto; do
  if [ $t = auto ]; then unset LRGE_HIP_LPG_MAX; else export LRGE_HIP_LPG_MAX=$t; fi
  timeout 300 python bench.py --steps 2 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb LPG_MAX=$t"
done
for t in auto 1536 2560 3072 6144 8192; do
  if [ $t = auto ]; then unset LRGE_HIP_LPG_MAX; else export LRGE_HIP_LPG_MAX=$t; fi
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont LPG_MAX=$t"
done
for t in auto 512 1024 2048; do
  if [ $t = auto ]; then unset LRGE_HIP_LPG_MAX; else export LRGE_HIP_LPG_MAX=$t; fi
  timeout 300 python bench.py --config c4_dmel_twoset --steps 20 --warmup 2 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 LPG_MAX=$t"
done
