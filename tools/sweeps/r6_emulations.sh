#!/bin/bash
# round 6: the world-of-8 emulation of the forward strategy at full-size C5 on the resident clock, with the streamed set sketched once
# per world (lrge_hip_seqset_presketch_sharded; LRGE_BENCH_NO_QSHARD=1: every rank sketches all queries, round 5's form).
#   tools/sweeps/r6_emulations.sh [fwd|fwdold|fwdhost|inv]...   (default: fwd fwdold)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
out=gpurun_out/r6emu; mkdir -p $out
what=${*:-fwd fwdold}
show() { python -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1])
    print('$1', d.get('clock','')[:8], 'one', round(d['one_gpu_ms_per_step'],1), 'max', round(d['max_rank_busy_ms_per_step'],1), 'x', round(d['projected_speedup_compute_only'],2), 'link', d.get('projected_speedup_with_link_model') and round(d['projected_speedup_with_link_model'],2), d.get('link_model'), d['all_ranks_equal_one_gpu'], [round(r['busy_ms_per_step'],1) for r in d['ranks']], [r['work_last_step']['batches'] for r in d['ranks']])
except Exception as e: print('$1', 'unreadable', e)
"; }
export LRGE_BENCH_EMULATE_TIMEOUT=600
for w in $what; do
  case $w in
    fwd) timeout 900 python bench.py --emulate-world 8 --clock resident --steps 3 --warmup 1 > $out/fwd_resident.json 2> $out/fwd_resident.err; show $out/fwd_resident.json;;
    fwdold) LRGE_BENCH_NO_QSHARD=1 timeout 900 python bench.py --emulate-world 8 --clock resident --steps 3 --warmup 1 > $out/fwd_resident_noqshard.json 2> $out/fwd_resident_noqshard.err; show $out/fwd_resident_noqshard.json;;
    fwdhost) timeout 900 python bench.py --emulate-world 8 --clock host --steps 3 --warmup 1 > $out/fwd_host.json 2> $out/fwd_host.err; show $out/fwd_host.json;;
    inv) timeout 900 python bench.py --emulate-world 8 --inverse --clock resident --steps 3 --warmup 1 > $out/inv_resident.json 2> $out/inv_resident.err; show $out/inv_resident.json;;
  esac
done
