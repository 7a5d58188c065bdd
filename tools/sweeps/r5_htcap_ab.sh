#!/bin/bash
# round 5: home slots per key in the tables of a partitioned index: 1.25 (default for parts) against 1.5 / 2.0 (LRGE_HIP_HT_SLOTS_X100), host clock, one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_index_table','lookup','k_lookup','expand','total') if k in s}, 'parts', w.get('index_parts'), 'batches', w.get('batches'))"; }
for x in 125 150 200 125 200; do
  export LRGE_HIP_HT_SLOTS_X100=$x
  timeout 300 python bench.py --steps 3 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb slots x$x"
done
for x in 125 200; do
  export LRGE_HIP_HT_SLOTS_X100=$x
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont slots x$x"
done
