#!/bin/bash
# round 6: the round's closing collection, one box where it fits: the bench line + rocprofv3 passes of the default workload (C5 ava-pb), the same
# line under ava-ont, C4, the driver's own command, the world-of-8 emulations (forward with / without the sharded query sketch, inverse).
#   tools/sweeps/r6_final_campaign.sh [prof|ont|c4|driver|emu]...   (default: all)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
what=${*:-prof ont c4 driver emu}
for w in $what; do
  case $w in
    prof) bash tools/profile_round.sh r6p > gpurun_out/r6p_profile.log 2>&1; tail -2 gpurun_out/r6p_profile.log;;
    ont) timeout 900 python bench.py --preset ont > gpurun_out/r6_bench_c5_ont.json 2> gpurun_out/r6_bench_c5_ont.err; python tools/show_bench.py gpurun_out/r6_bench_c5_ont.json | head -4
         cd /tmp; TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r6o_stats -o s -- python $root/bench.py --preset ont --no-cpu-baseline --no-resident --parity-sample 0 --steps 3 --warmup 1 > $root/gpurun_out/r6o_stats.json 2> $root/gpurun_out/r6o_stats.err; cd $root
         find gpurun_out/r6o_stats -name "*kernel_trace*" -size +40M -delete 2>/dev/null;;
    c4) SKIP_PLAIN= bash tools/profile_round.sh r6q --config c4_dmel_twoset --steps 20 --warmup 2 > gpurun_out/r6q_profile.log 2>&1; tail -1 gpurun_out/r6q_profile.log;;
    driver) timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver_form.json 2> gpurun_out/r6_bench_driver_form.err; python tools/show_bench.py gpurun_out/r6_bench_driver_form.json | head -3;;
    emu) bash tools/sweeps/r6_emulations.sh fwd fwdold inv > gpurun_out/r6emu_final.txt 2>&1; cat gpurun_out/r6emu_final.txt;;
  esac
done
