#!/bin/bash
# usage (here, after tools/round_profile.sh <tag> ran on the GPU box): tools/collect_profiles.sh <tag> <name>
#   gpurun_out/<tag>_* -> profiles/<name>_* (the committed evidence set) + profiles/blend_bwd_counters.json
tag=$1; name=$2; O=gpurun_out
for f in bench bench_2ranks_gloo_1gpu bench_config5 bench_dense bench_one_at_a_time bench_opaque bench_exchange_at_1 \
         train_step train_step_fateavatar train_step_fateavatar_tex256 train_step_fateavatar_tex256_batch4 train_step_fateavatar_batch3 train_step_fateavatar_batch4 train_step_fateavatar_batch4_lanes \
         train_step_fateavatar_binding_op train_step_fateavatar_batch4_binding_op \
         train_step_fateavatar_random_order train_step_fateavatar_batch4_random_order; do
  [ -f $O/${tag}_$f.json ] && grep '^{' $O/${tag}_$f.json | tail -1 > profiles/${name}_$f.json
done
cp $O/${tag}_coherent_order.txt profiles/${name}_coherent_order.txt
cp $O/${tag}_cpu_baseline.txt profiles/${name}_cpu_baseline.txt
[ -f $O/${tag}_bwd_stats.txt ] && cp $O/${tag}_bwd_stats.txt profiles/${name}_bwd_stats.txt
[ -f $O/${tag}_bwd_phases.txt ] && cp $O/${tag}_bwd_phases.txt profiles/${name}_bwd_phases.txt
[ -f $O/${tag}_pre_phases.txt ] && cp $O/${tag}_pre_phases.txt profiles/${name}_pre_phases.txt
cp $O/${tag}_eager/kernels.txt profiles/${name}_kernel_stats_bench_eager.txt
cp $O/${tag}_graph/kernels.txt profiles/${name}_kernel_stats_bench_graph.txt
cp $O/${tag}_graph3/kernels.txt profiles/${name}_kernel_stats_bench_graph_3_in_flight.txt
cp $O/${tag}_graph/timeline.txt profiles/${name}_timeline.txt
[ -f $O/${tag}_fa1/timeline.txt ] && cp $O/${tag}_fa1/timeline.txt profiles/${name}_timeline_fateavatar_step.txt
[ -f $O/${tag}_fa4/timeline.txt ] && cp $O/${tag}_fa4/timeline.txt profiles/${name}_timeline_fateavatar_step_batch4.txt
cp $O/${tag}_fetch/pmc.txt profiles/${name}_pmc_FETCH_SIZE.txt
cp $O/${tag}_write/pmc.txt profiles/${name}_pmc_WRITE_SIZE.txt
cp $O/${tag}_sq1/pmc.txt profiles/${name}_sq_counters_1.txt
cp $O/${tag}_sq2/pmc.txt profiles/${name}_sq_counters_2.txt
python tools/make_counters_json.py "$tag (rocprofv3, tools/round_profile.sh)" profiles/${name}_pmc_FETCH_SIZE.txt \
    profiles/${name}_pmc_WRITE_SIZE.txt profiles/${name}_sq_counters_1.txt
