#!/bin/bash
# round 5, one box: (a) sketch / index tests of the build, (b) HEAD~ (liblrge_hip_ref.so, tools/ab.sh build) against the build on C5 ava-pb / ava-ont,
# (c) the ava-ont step on the resident clock with its stage times, (d) the C4 profile set
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in s if s[k] >= 5}, 'parts', w.get('index_parts'), 'batches', w.get('batches'))"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sketch or index" 2>&1 | tail -3
for v in ref cur ref cur; do
  if [ $v = ref ]; then export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ref.so; else unset LRGE_HIP_LIB_AB; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb $v"
done
for v in ref cur; do
  if [ $v = ref ]; then export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ref.so; else unset LRGE_HIP_LIB_AB; fi
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont host $v"
done
unset LRGE_HIP_LIB_AB
timeout 400 python bench.py --preset ont --steps 2 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>$out/r5_ont_resident.err | show "c5-ont resident"
tail -3 $out/r5_ont_resident.err
bash tools/profile_round.sh r5q --config c4_dmel_twoset --steps 20 --warmup 2
