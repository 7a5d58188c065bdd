root=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'index_sketch', round(s['index_sketch'],2), 'index_sort', round(s['index_index_sort'],1))"; }
lscpu | grep -E "^CPU\(s\)|NUMA node|Thread|Socket|Model name" | head -12
for t in "" 16 48 64 96; do
  if [ -z "$t" ]; then unset LRGE_HIP_HOST_PACK_THREADS; else export LRGE_HIP_HOST_PACK_THREADS=$t; fi
  timeout 300 python $root/bench.py --steps 3 --warmup 1 --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "threads=${t:-default}"
done
