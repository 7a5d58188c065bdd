root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_index_table','sketch','qfilter','lookup','expand','anchor_sort','group','chain','total')}, d['work_per_step'].get('batches'))"; }
for v in 125 150 125 150 175; do
  export LRGE_HIP_HT_SLOTS_X100=$v
  timeout 300 python bench.py --steps 4 --warmup 1 --clock resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "ht_slots=$v"
done
