# the round's world emulations, re-run with the final code (gpurun_out/r4e_*.json)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
timeout 1500 python bench.py --emulate-world 8 --inverse --steps 2 --warmup 1 > gpurun_out/r4e_inv.json 2> gpurun_out/r4e_inv.err
for n in 2 4 8; do timeout 600 python bench.py --config c4_dmel_twoset --emulate-world $n > gpurun_out/r4e_c4_w$n.json 2> gpurun_out/r4e_c4_w$n.err; done
for f in gpurun_out/r4e_inv.json gpurun_out/r4e_c4_w2.json gpurun_out/r4e_c4_w4.json gpurun_out/r4e_c4_w8.json; do
  python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['one_gpu_ms_per_step'],1), round(d['max_rank_busy_ms_per_step'],1), round(d['projected_speedup_compute_only'],2), round(d['projected_speedup_with_link_model'],2), d['all_ranks_equal_one_gpu'], [round(r['busy_ms_per_step'],1) for r in d['ranks']])"
done
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r4_gpu_tests.log 2>&1; tail -3 gpurun_out/r4_gpu_tests.log
