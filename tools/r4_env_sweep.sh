root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_index_table','lookup','expand','total')})"; }
for v in 125 110 150 200 125; do
  export LRGE_HIP_HT_SLOTS_X100=$v
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "ht_slots=$v"
done
