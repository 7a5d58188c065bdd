#!/bin/bash
# round 5: the host-side pack job submitted in front of the per-read arrays' preparation (cur) against HEAD~ (ref, tools/ab.sh build), host clock, one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','total') if k in s})"; }
timeout 400 python -m pytest tests/test_gpu_upload.py tests/test_gpu_configs.py -m gpu -x -q -k "not c5_full" 2>&1 | tail -2
for v in ref cur ref cur ref cur; do
  if [ $v = ref ]; then export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ref.so; else unset LRGE_HIP_LIB_AB; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb host $v"
done
for v in ref cur ref cur; do
  if [ $v = ref ]; then export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ref.so; else unset LRGE_HIP_LIB_AB; fi
  timeout 300 python bench.py --config c4_dmel_twoset --steps 20 --warmup 2 --clock host --no-resident --no-cpu-baseline 2>/dev/null | show "c4 host $v"
done
