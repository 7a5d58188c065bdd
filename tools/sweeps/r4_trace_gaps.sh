root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export LRGE_HIP_NO_SEG_PACK=1; else unset LRGE_HIP_NO_SEG_PACK; fi
  echo "== noseg=$v"
  rm -rf /tmp/tg; LRGE_HIP_VERBOSE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $root/bench.py --no-cpu-baseline --no-resident --parity-sample 0 --steps 1 --warmup ${WARM:-2} > /tmp/tg.out 2> /tmp/tg.err
  grep "pool at" /tmp/tg.err | tail -2
  f=$(find /tmp/tg -name "*kernel_trace.csv" | head -1)
  python $root/tools/trace_gaps.py $f ${FRAC:-0.85} | head -${LINES:-14}
done
