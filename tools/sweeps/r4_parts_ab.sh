root=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','lookup','expand','anchor_sort','chain','total')}, d.get('work_per_step',{}).get('batches'))"; }
for pb in "" 15100000000; do
    if [ -z "$pb" ]; then unset LRGE_HIP_PART_BASES; else export LRGE_HIP_PART_BASES=$pb; fi
    timeout 300 python $root/bench.py --steps 3 --warmup 1 --no-resident --no-cpu-baseline --parity-sample 0 2>/tmp/err.txt | show "host part_bases=${pb:-auto}"
    tail -2 /tmp/err.txt | cut -c1-300
    timeout 300 python $root/bench.py --steps 3 --warmup 1 --clock resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "resident part_bases=${pb:-auto}"
done
