#!/bin/bash
# A/B two builds of liblrge_hip.so on the bench workload inside ONE gpurun call (box-to-box noise is ~5 %).
#   here (no GPU):  tools/ab.sh build <git-ref>     -> lrge_amd/_lib/liblrge_hip_ref.so from that commit's sources
#   on the GPU box: tools/ab.sh run [rounds]        -> alternates ref / current, prints ms per step and the chain stage
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  ref=${2:-HEAD}; tmp=$(mktemp -d)
  (cd "$root" && git archive "$ref" lrge_amd/csrc include | tar -x -C "$tmp")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I"$tmp/include" -I"$tmp/lrge_amd/csrc" \
      -o "$root/lrge_amd/_lib/liblrge_hip_ref.so" "$tmp/lrge_amd/csrc/lrge_hip.hip" -lz -ldl
  rm -rf "$tmp"; echo "built ref from $ref"
else
  n=${2:-3}
  for i in $(seq 1 $n); do
    for v in ref cur; do
      if [ $v = ref ]; then export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_ref.so; else unset LRGE_HIP_LIB_AB; fi
      python "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$v', round(d['ms_per_step'],3), 'chain', round(s['chain'],3), 'idx', round(sum(v for k,v in s.items() if k.startswith('index_') and k!='index_rs_scatter'),3), 'ovl', round(s['total'],3))"
    done
  done
fi
