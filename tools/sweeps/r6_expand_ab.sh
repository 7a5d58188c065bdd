#!/bin/bash
# A/B of the expansion kernels (k_seed.h: expand_wave_chunk) on full-size C5 (both presets) and C4, one box, alternating builds
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],2) for k in ('expand','lookup','anchor_sort','chain','total') if k in s}, 'est', d.get('genome_size_estimate'))"; }
for i in 1 2; do
for v in ref cur; do
  if [ $v = cur ]; then unset LRGE_HIP_LIB_AB; else export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_$v.so; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb $v"
  [ $i = 1 ] && timeout 400 python bench.py --preset ont --steps 2 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont $v"
  timeout 300 python bench.py --config c4_dmel_twoset --steps 10 --warmup 2 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c4-ont $v"
done
done
