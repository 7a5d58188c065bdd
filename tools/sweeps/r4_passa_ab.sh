root=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','index_rs_scatter','total')})"; }
for v in "" 1; do
  if [ -z "$v" ]; then unset LRGE_HIP_NO_SKETCH_PASS_A; else export LRGE_HIP_NO_SKETCH_PASS_A=1; fi
  timeout 300 python $root/bench.py --steps 4 --warmup 1 --clock resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "resident no_pass_a=${v:-0}"
  timeout 300 python $root/bench.py --steps 4 --warmup 1 --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "host no_pass_a=${v:-0}"
done
unset LRGE_HIP_NO_SKETCH_PASS_A

