#!/bin/bash
# round 5: the position-parallel sketch (k_sketch_tile.h, LRGE_HIP_SKETCH_TILE_FORM=1) against the lane-per-chunk form (default),
# alternating inside one box: C5 ava-pb, C5 ava-ont, C4.
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','sketch','total') if k in s})"; }
for v in ${*:-lane tile lane tile}; do
  if [ $v = tile ]; then export LRGE_HIP_SKETCH_TILE_FORM=1; else unset LRGE_HIP_SKETCH_TILE_FORM; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb $v"
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont $v"
  timeout 300 python bench.py --config c4_dmel_twoset --steps 20 --warmup 2 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 $v"
done
