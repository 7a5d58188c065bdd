#!/bin/bash
# round 6: next-pass digit bytes (every scatter that writes packed words also leaves the byte the next pass ranks by; that pass's histogram
# reads 1 byte per entry instead of 8: default) against histograms over the words (LRGE_HIP_NO_DIGIT_BYTES=1, rounds 3-5), one box, alternating
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; w=d['work_per_step']; p=d.get('parity_vs_oracle_sample'); print('$1', round(d['ms_per_step'],1), {k: round(s[k],1) for k in ('index_sketch','index_index_sort','index_index_table','index_rs_scatter','anchor_sort','rs_scatter','total') if k in s}, 'scatter GB', round(w.get('rs_scatter_bytes',0)/1e9,1), 'mid_occ', d.get('mid_occ'), 'est', d.get('genome_size_estimate'), 'parity', p and (p.get('counts_equal'), p.get('n_minimizers_equal'), p.get('n_keys_equal'), p.get('mid_occ_equal')))"; }
for v in words bytes words bytes; do
  if [ $v = words ]; then export LRGE_HIP_NO_DIGIT_BYTES=1; else unset LRGE_HIP_NO_DIGIT_BYTES; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --clock resident --no-from-host --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-pb resident $v"
done
unset LRGE_HIP_NO_DIGIT_BYTES
timeout 400 python bench.py --steps 3 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 256 2>/dev/null | show "c5-pb host bytes +parity"
for v in words bytes; do
  if [ $v = words ]; then export LRGE_HIP_NO_DIGIT_BYTES=1; else unset LRGE_HIP_NO_DIGIT_BYTES; fi
  timeout 300 python bench.py --preset ont --steps 2 --warmup 1 --clock host --no-resident --no-cpu-baseline --parity-sample 0 2>/dev/null | show "c5-ont host $v"
  timeout 300 python bench.py --config c4_dmel_twoset --steps 5 --warmup 2 --clock resident --no-from-host --no-cpu-baseline 2>/dev/null | show "c4 resident $v"
done
