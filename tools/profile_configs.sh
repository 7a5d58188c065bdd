#!/bin/bash
# rocprofv3 evidence beyond the headline workload (run on the GPU box via gpurun): kernel stats + FETCH_SIZE / WRITE_SIZE passes
# for the HiFi regime (C5/10, both presets, forward and inverse) and the all-vs-all configuration (C3).
#   tools/profile_configs.sh <tag>    -> gpurun_out/<tag>_<name>_{stats,fetch,write}/  + gpurun_out/<tag>_<name>.json
# Counter passes are separate runs with --kernel-trace only (gpurun refuses --pmc together with other trace domains).
set -u
tag=${1:-cfg}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {   # name, command...
    local name=$1; shift
    timeout 600 "$@" > "$out/${tag}_${name}.json" 2> "$out/${tag}_${name}.err"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_${name}_stats" -o s -- "$@" > /dev/null 2> "$out/${tag}_${name}_stats.err"
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/${tag}_${name}_fetch" -o f -- "$@" > /dev/null 2> "$out/${tag}_${name}_fetch.err"
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/${tag}_${name}_write" -o w -- "$@" > /dev/null 2> "$out/${tag}_${name}_write.err"
    tail -c 300 "$out/${tag}_${name}.json"; echo
}
B="python $root/bench.py --generator cb --config c5_human_tenth --no-cpu-baseline --no-from-host --steps 2 --warmup 1"
run c5t_pb_fwd $B --preset pb
run c5t_pb_inv $B --preset pb --inverse
run c5t_ont_fwd $B --preset ont
run c5t_ont_inv $B --preset ont --inverse
run c3_ava python $root/tools/run_config.py c3_yeast_ava --check 0 --repeat 2
ls "$out" | grep -c "${tag}_"
