#!/bin/bash
# build variants of the library with extra -D flags (here, no GPU):  tools/sweeps/ab_variants.sh build name "-DLPG_B=8" ...
# then on the GPU box: tools/sweeps/ab_variants.sh run name1 name2 ...   (alternates with the current build)
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  name=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" -I"$root/include" -I"$root/lrge_amd/csrc" \
      -o "$root/lrge_amd/_lib/liblrge_hip_$name.so" "$root/lrge_amd/csrc/lrge_hip.hip" && echo "built $name"
else
  shift
  for round in 1 2; do
    for v in cur "$@"; do
      if [ $v = cur ]; then unset LRGE_HIP_LIB_AB; else export LRGE_HIP_LIB_AB=$root/lrge_amd/_lib/liblrge_hip_$v.so; fi
      python "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$v', round(d['ms_per_step'],3), 'chain', round(s['chain'],3), 'lpg', round(s.get('chain_lpg',0),3))"
    done
  done
fi
