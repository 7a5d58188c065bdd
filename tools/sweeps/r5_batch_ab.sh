#!/bin/bash
# round 5: anchor batches of up to 2^32 seed hits for filtered count-only runs (cur) against the 2^31 rule (LRGE_HIP_BATCH_HITS_2G=1), one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('lookup','expand','anchor_sort','group','chain','total') if k in s}, 'batches', w.get('batches'), 'split', w.get('lpg_split'), 'kept', int(w.get('anchors_kept',0)), 'mid_occ', d.get('mid_occ'), 'est', d.get('genome_size_estimate'))"; }
for v in old new old new; do
  if [ $v = old ]; then export LRGE_HIP_BATCH_HITS_2G=1; else unset LRGE_HIP_BATCH_HITS_2G; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb resident $v"
  timeout 300 python bench.py --steps 3 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb host $v"
done
for v in old new; do
  if [ $v = old ]; then export LRGE_HIP_BATCH_HITS_2G=1; else unset LRGE_HIP_BATCH_HITS_2G; fi
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont $v"
  timeout 300 python bench.py --config c4_dmel_twoset --steps 20 --warmup 2 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 $v"
done
