#!/bin/bash
# A/B of the word-synchronous HPC sketch loop (k_sketch.h, sketch_chunk_hpc) against the build of tools/ab.sh's ref, one gpurun call
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_k_sketch','sketch','index_index_sort','total') if k in s}, 'mid_occ', d.get('mid_occ'), 'est', d.get('genome_size_estimate'))"; }
for i in 1 2; do
for v in ${VARIANTS:-ref ws cur}; do
  if [ $v = cur ]; then unset LRGE_HIP_LIB_AB; else export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_$v.so; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample ${PS:-0} 2>/dev/null | show "c5-pb resident $v"
done
done
