#!/bin/bash
# round 5: host-side pack threads against the CPUs the host grants (cgroup cpu.max = 16 on this pool), full-size C5 on the host clock
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','total')})"; }
for t in default 8 12 16 24 32 default; do
  if [ $t = default ]; then unset LRGE_HIP_HOST_PACK_THREADS; else export LRGE_HIP_HOST_PACK_THREADS=$t; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "pack_threads=$t"
done
