#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box via gpurun):
#   tools/profile_round.sh <tag> [bench args]   -> gpurun_out/<tag>_{stats,sq,fetch,write}/ + gpurun_out/<tag>_bench.json
# Counter passes are separate runs with --kernel-trace only (TCC counters do not fit beside SQ ones; gpurun refuses
# --pmc together with other trace domains).  The plain run is the full default bench (both clocks, CPU leg, parity sample);
# the profiled runs time the `value` clock only (--no-resident) and skip the CPU legs.
set -u
tag=${1:-prof}
shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --no-cpu-baseline --no-resident --parity-sample 0 $*"
if [ -z "${SKIP_PLAIN:-}" ]; then
  timeout 1200 python $root/bench.py "$@" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
fi
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o s -- $B --steps 3 --warmup 1 > "$out/${tag}_stats.json" 2> "$out/${tag}_stats.err"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d "$out/${tag}_sq" -o q -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_sq.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/${tag}_fetch" -o f -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_fetch.err"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/${tag}_write" -o w -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_write.err"
# the traces themselves are large: keep the per-kernel statistics and the counter tables
find "$out/${tag}_stats" -name "*kernel_trace*" -size +40M -delete 2>/dev/null
ls -R "$out" | grep -c csv
tail -1 "$out/${tag}_bench.json" 2>/dev/null | cut -c1-400
