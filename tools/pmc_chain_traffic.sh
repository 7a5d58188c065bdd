#!/bin/bash
# HBM fetch / write bytes of the chain kernels only (two separate counter passes), GPU box
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; tag=${1:-ct}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "k_chain" --kernel-trace --output-format csv -d "$out/${tag}_$c" -o x -- python $root/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> "$out/${tag}_$c.err"
  python - <<P
import csv
for r in csv.DictReader(open("$out/${tag}_$c/x_counter_collection.csv")): print(r['Kernel_Name'][:30], r['Counter_Name'], float(r['Counter_Value'])*1024/1e6, 'MB (raw KiB->MB; fetch needs x2)')
P
done
