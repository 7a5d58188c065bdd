#!/bin/bash
# round 6: the wave-dense sketch for PACKED index entries (C4-sized jobs) against per-chunk slots read by the sort's first pass (LRGE_HIP_NO_WAVE_SKETCH=1)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],2), {k: round(s[k],2) for k in ('index_sketch','index_index_sort','index_index_table','lookup','expand','anchor_sort','chain','total') if k in s}, 'mid_occ', d.get('mid_occ'), 'est', d.get('genome_size_estimate'))"; }
for v in slots wave slots wave; do
  if [ $v = slots ]; then export LRGE_HIP_NO_WAVE_SKETCH=1; else unset LRGE_HIP_NO_WAVE_SKETCH; fi
  timeout 300 python bench.py --config c4_dmel_twoset --steps 20 --warmup 3 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 resident $v"
done
for v in slots wave; do
  if [ $v = slots ]; then export LRGE_HIP_NO_WAVE_SKETCH=1; else unset LRGE_HIP_NO_WAVE_SKETCH; fi
  timeout 300 python bench.py --config c2_bact_twoset --steps 30 --warmup 3 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c2 resident $v"
done
unset LRGE_HIP_NO_WAVE_SKETCH
timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb resident (dword head flags)"
