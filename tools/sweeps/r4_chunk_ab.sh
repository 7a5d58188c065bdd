root=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],2), {k: round(s[k],2) for k in ('index_sketch','sketch','index_index_sort','total')})"; }
cd $root
for v in cur ck256 ck192 cur ck256; do
  if [ $v = cur ]; then unset LRGE_HIP_LIB_AB; else export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_$v.so; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5 $v"
  timeout 300 python bench.py --config c4_dmel_twoset --steps 20 --warmup 2 --clock resident --no-cpu-baseline 2>/dev/null | show "c4 $v"
done
export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ck256.so
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sketch or minimizer or index or hpc" 2>&1 | tail -3
