root=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$1', round(d['ms_per_step'],2), 'chain', round(s['chain'],2), 'lpg', round(s.get('chain_lpg',0),2))"; }
for e in "4 32" "0 0" "2 64" "8 16" "16 8"; do
  set -- $e
  for cfg in "--config c2_repeats --steps 5 --warmup 2" "--steps 2 --warmup 1 --parity-sample 0" "--config c4_dmel_twoset --steps 5 --warmup 2"; do
    LRGE_HIP_LPG_SLOW_RATE=$1 LRGE_HIP_LPG_SLOW_ENTRY_EVERY=$2 timeout 300 python $root/bench.py $cfg --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "rate=$1 entry_every=$2 [$cfg]"
  done
done
