#!/bin/bash
# sweep the hw/lpg split threshold on the bench workload (GPU box)
for m in ${@:-256 384 448 512 640}; do
  LRGE_HIP_LPG_MAX=$m timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lpg_max=$m', round(d['ms_per_step'],3), round(d['stage_ms_per_step']['chain'],3), d['genome_size_estimate'])"
done
