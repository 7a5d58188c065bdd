#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box via gpurun):
#   tools/profile_round.sh <tag> [bench args]   -> gpurun_out/<tag>_{stats,sq,fetch,write}/ + gpurun_out/<tag>_bench.json
# Counter passes are separate runs with --kernel-trace only (TCC counters do not fit beside SQ ones; gpurun refuses
# --pmc together with other trace domains).
set -u
tag=${1:-prof}
shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --no-cpu-baseline --no-from-host $*"
timeout 900 python $root/bench.py "$@" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o s -- $B --steps 5 --warmup 1 > "$out/${tag}_stats.json" 2> "$out/${tag}_stats.err"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d "$out/${tag}_sq" -o q -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_sq.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/${tag}_fetch" -o f -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/${tag}_write" -o w -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_write.err"
ls -R "$out" | grep -c csv
tail -1 "$out/${tag}_bench.json" | cut -c1-600
