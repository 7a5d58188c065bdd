#!/bin/bash
run() { # label, env...
  label=$1; shift
  env "$@" LRGE_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$label', round(d['ms_per_step'],3), 'isort', round(s['index_index_sort'],2), 'sketch', round(s['sketch'],2), 'asort', round(s['anchor_sort'],2), 'chain', round(s['chain'],2), 'lpg', round(s['chain_lpg'],2))"
}
plain() { label=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$label', round(d['ms_per_step'],3), 'isort', round(s['index_index_sort'],2), 'sketch', round(s['sketch'],2), 'asort', round(s['anchor_sort'],2), 'chain', round(s['chain'],2), 'lpg', round(s['chain_lpg'],2))"
}
run torchrun_probe X=1
run torchrun_noprobe LRGE_HIP_NO_STREAM_PROBE=1
plain plain_probe X=1
plain plain_noprobe LRGE_HIP_NO_STREAM_PROBE=1
plain plain_probe X=1
plain plain_noprobe LRGE_HIP_NO_STREAM_PROBE=1
