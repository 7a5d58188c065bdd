#!/bin/bash
# A/B of the chain kernels' loads of their own records (k_chain_common.h: ld_u64_l2) on full-size C5 ava-pb and on C4 (ava-ont), one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],2) for k in ('chain','chain_lpg','chain_hw','group','total') if k in s}, 'est', d.get('genome_size_estimate'))"; }
for i in 1 2; do
for v in ref cur; do
  if [ $v = cur ]; then unset LRGE_HIP_LIB_AB; else export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_$v.so; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb $v"
  timeout 300 python bench.py --config c4_dmel_twoset --steps 10 --warmup 2 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c4-ont $v"
done
done
