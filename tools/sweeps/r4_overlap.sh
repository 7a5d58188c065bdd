root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $root/bench.py --no-cpu-baseline --clock resident --no-resident --parity-sample 0 --steps 1 --warmup 1 > /tmp/tg.out 2> /tmp/tg.err
f=$(find /tmp/tg -name "*kernel_trace.csv" | head -1)
python $root/tools/trace_overlap.py $f k_sketch_direct k_sketch_compact 0.9
