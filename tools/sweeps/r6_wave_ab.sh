#!/bin/bash
# round 6: the wave-dense index sketch (k_sketch_wave: a wavefront fills its slot densely, the sort's first pass reads the slots, no compaction;
# default) against the slot-per-chunk form + k_sketch_compact (LRGE_HIP_NO_WAVE_SKETCH=1, rounds 2-5), one box, alternating
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_k_sketch','index_index_sort','index_index_table','index_rs_scatter','anchor_sort','total') if k in s}, 'scatter GB', round(w.get('rs_scatter_bytes',0)/1e9,1), 'mid_occ', d.get('mid_occ'), 'est', d.get('genome_size_estimate'))"; }
for v in slots wave slots wave; do
  if [ $v = slots ]; then export LRGE_HIP_NO_WAVE_SKETCH=1; else unset LRGE_HIP_NO_WAVE_SKETCH; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb resident $v"
done
for v in slots wave; do
  if [ $v = slots ]; then export LRGE_HIP_NO_WAVE_SKETCH=1; else unset LRGE_HIP_NO_WAVE_SKETCH; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb host $v"
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont host $v"
done
