#!/bin/bash
# round 5: the world-of-8 emulations at full-size C5 on BOTH clocks, the forward form three times on the resident clock (the slowest rank
# used to move from run to run), north_star's literal form (replicated index) beside them, and the host-only pack contention figure.
#   tools/sweeps/r5_emulations.sh [fwd3|fwdhost|inv|repl|pack|c4]...   (default: all)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
out=gpurun_out/r5emu; mkdir -p $out
what=${*:-fwd3 fwdhost inv repl pack}
show() { python -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1])
    print('$1', d.get('clock','')[:8], 'one', round(d['one_gpu_ms_per_step'],1), 'max', round(d['max_rank_busy_ms_per_step'],1), 'x', round(d['projected_speedup_compute_only'],2), 'link', d.get('projected_speedup_with_link_model') and round(d['projected_speedup_with_link_model'],2), d['all_ranks_equal_one_gpu'], [round(r['busy_ms_per_step'],1) for r in d['ranks']], [r['work_last_step']['batches'] for r in d['ranks']])
except Exception as e: print('$1', 'unreadable', e)
"; }
export LRGE_BENCH_EMULATE_TIMEOUT=600
for w in $what; do
  case $w in
    fwd3) for i in 1 2 3; do timeout 900 python bench.py --emulate-world 8 --clock resident --steps 2 --warmup 1 > $out/fwd_resident_$i.json 2> $out/fwd_resident_$i.err; show $out/fwd_resident_$i.json; done;;
    fwd1) timeout 900 python bench.py --emulate-world 8 --clock resident --steps 2 --warmup 1 > $out/fwd_resident_final.json 2> $out/fwd_resident_final.err; show $out/fwd_resident_final.json;;
    fwdhost) timeout 900 python bench.py --emulate-world 8 --clock host --steps 2 --warmup 1 > $out/fwd_host.json 2> $out/fwd_host.err; show $out/fwd_host.json;;
    inv) timeout 900 python bench.py --emulate-world 8 --inverse --clock resident --steps 2 --warmup 1 > $out/inv_resident.json 2> $out/inv_resident.err; show $out/inv_resident.json
         timeout 900 python bench.py --emulate-world 8 --inverse --clock host --steps 2 --warmup 1 > $out/inv_host.json 2> $out/inv_host.err; show $out/inv_host.json;;
    repl) LRGE_BENCH_FORWARD=replicated timeout 1500 python bench.py --emulate-world 8 --clock resident --steps 1 --warmup 1 > $out/fwd_replicated_resident.json 2> $out/fwd_replicated_resident.err; show $out/fwd_replicated_resident.json;;
    pack) timeout 600 python tools/micro/pack_contention.py > $out/pack_contention.json 2> $out/pack_contention.err; cat $out/pack_contention.json;;
    c4) for n in 2 4 8; do timeout 600 python bench.py --config c4_dmel_twoset --emulate-world $n > $out/c4_w$n.json 2> $out/c4_w$n.err; show $out/c4_w$n.json; done;;
  esac
done
