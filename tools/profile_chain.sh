#!/bin/bash
# kernel trace + SQ counters for the chain kernels only (GPU box); tag = $1
set -u
tag=${1:-pc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o s -- $B --steps 3 --warmup 1 > "$out/${tag}_stats.json" 2> "$out/${tag}_stats.err"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --kernel-include-regex "k_chain" --kernel-trace --output-format csv -d "$out/${tag}_sq" -o q -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_sq.err"
python - <<P
import csv,collections
rows=list(csv.DictReader(open("$out/${tag}_stats/s_kernel_stats.csv")))
for r in rows[:8]: print(r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3)
agg=collections.defaultdict(lambda: collections.defaultdict(float))
try:
    for r in csv.DictReader(open("$out/${tag}_sq/q_counter_collection.csv")): agg[r['Kernel_Name'][:20]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items(): print(k, {a:'%.3g'%b for a,b in v.items()})
except Exception as e: print('pmc fail', e)
P
tail -3 "$out/${tag}_sq.err"
