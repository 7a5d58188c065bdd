#!/bin/bash
# round 5 (end): one-option-at-a-time sweep of the C5 ava-pb step on the host clock against the defaults, one box
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('%-42s' % '$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','lookup','expand','anchor_sort','chain','total') if k in s})"; }
run() { ( for kv in "$@"; do export "$kv"; done; timeout 300 python bench.py --steps 3 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "$*" ); }
run DEFAULT=1
run LRGE_HIP_HT_POWER=0
run LRGE_HIP_HT_POWER=2
run LRGE_HIP_HT_POWER=4
run LRGE_HIP_LOCAL_SORT_MAX=1
run LRGE_HIP_HOST_PACK_CHUNK_WORDS=4194304
run LRGE_HIP_HOST_PACK_CHUNK_WORDS=8388608
run LRGE_HIP_HOST_PACK_CHUNK_WORDS=33554432
run LRGE_HIP_LPG_PRIO=0
run LRGE_HIP_LPG_PRIO=1
run LRGE_HIP_HW_PRIO=2
run LRGE_HIP_COUNTS_AFTER_LOOKUP=1
run LRGE_HIP_LSORT_SERIAL=1
run DEFAULT=2
