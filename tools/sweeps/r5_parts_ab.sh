#!/bin/bash
# round 5: index parts of the default size (3 at C5 ava-pb, 4 under ava-ont) against pinned smaller ones (LRGE_HIP_PART_BASES), host clock, one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','lookup','expand','anchor_sort','chain','total') if k in s}, 'parts', w.get('index_parts'), 'batches', w.get('batches'))"; }
for pb in auto 3800000000 5100000000 auto; do
  if [ $pb = auto ]; then unset LRGE_HIP_PART_BASES; else export LRGE_HIP_PART_BASES=$pb; fi
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont PART_BASES=$pb"
done
for pb in auto 3800000000 5100000000 7600000000; do
  if [ $pb = auto ]; then unset LRGE_HIP_PART_BASES; else export LRGE_HIP_PART_BASES=$pb; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb PART_BASES=$pb"
done
