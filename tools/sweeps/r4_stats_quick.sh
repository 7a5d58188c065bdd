root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/qs
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -o s -- python $root/bench.py --no-cpu-baseline --no-resident --parity-sample 0 --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/qs/**/s_kernel_stats.csv', recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<26: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), '%9.3f'%(float(r['AverageNs'])/1e6), '%9.1f'%(float(r['TotalDurationNs'])/3e6))
PY
