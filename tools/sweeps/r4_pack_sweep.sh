root=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','total')})"; }
cd $root; timeout 600 python -m pytest tests/test_gpu_upload.py -q -x 2>&1 | tail -2
for rep in 1 2 3; do
for ch in "" 2097152; do
    if [ -z "$ch" ]; then unset LRGE_HIP_HOST_PACK_CHUNK_WORDS; else export LRGE_HIP_HOST_PACK_CHUNK_WORDS=$ch; fi
    timeout 300 python $root/bench.py --steps 5 --warmup 1 --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "chunk=${ch:-auto}"
done
done
unset LRGE_HIP_HOST_PACK_CHUNK_WORDS
timeout 300 python $root/bench.py --config c4_dmel_twoset --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', d['ms_per_step'], d.get('resident'))"
