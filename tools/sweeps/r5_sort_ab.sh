#!/bin/bash
# round 5: the index sort with the entry's hash bits in significance order (e = 6, three keys-only passes at k = 19) against HEAD~'s
# (e = 2, four), alternating inside one box:  tools/ab.sh build <ref> first, here.
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','index_rs_scatter','total') if k in s}, 'scatter launches', w.get('rs_scatter_launches'), 'GB', round(w.get('rs_scatter_bytes',0)/1e9,1))"; }
for v in ref cur ref cur; do
  if [ $v = ref ]; then export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ref.so; else unset LRGE_HIP_LIB_AB; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb $v"
done
for v in ref cur; do
  if [ $v = ref ]; then export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ref.so; else unset LRGE_HIP_LIB_AB; fi
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont $v"
  timeout 300 python bench.py --config c4_dmel_twoset --steps 20 --warmup 2 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 $v"
done
