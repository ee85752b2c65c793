#!/bin/bash
# usage (GPU box, repo root): tools/round_profile.sh <tag>   -> gpurun_out/<tag>_*  (copy what matters into profiles/)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
python $R/bench.py --P 500000 --res 1024 --steps 100 --no-dp-reference --cpu-seconds 6 > $O/${tag}_bench_config5.json 2>> $O/${tag}_bench.err
python $R/bench.py --opacity 0.9 --steps 100 --no-dp-reference --cpu-seconds 0 > $O/${tag}_bench_opaque.json 2>> $O/${tag}_bench.err
python $R/bench.py --scale 3e-3 --opacity 0.3 --steps 100 --no-dp-reference --cpu-seconds 0 > $O/${tag}_bench_dense.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py > $O/${tag}_train_step.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py --fateavatar > $O/${tag}_train_step_fateavatar.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py --fateavatar --views-per-step 4 > $O/${tag}_train_step_fateavatar_batch4.json 2>> $O/${tag}_bench.err
# the reference's own size: tex_size 256 -> 65 536 Gaussians (config/fateavatar.yaml:28)
python $R/tools/train_synthetic.py --fateavatar --P 65536 > $O/${tag}_train_step_fateavatar_tex256.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py --fateavatar --P 65536 --views-per-step 4 > $O/${tag}_train_step_fateavatar_tex256_batch4.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py --fateavatar --views-per-step 4 --lanes > $O/${tag}_train_step_fateavatar_batch4_lanes.json 2>> $O/${tag}_bench.err
# the same two steps with the binding as its own kernels (the A/B of fr_aux::binding), and the steps' timelines
python $R/tools/train_synthetic.py --fateavatar --binding-op > $O/${tag}_train_step_fateavatar_binding_op.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py --fateavatar --views-per-step 4 --binding-op > $O/${tag}_train_step_fateavatar_batch4_binding_op.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py --fateavatar --random-order > $O/${tag}_train_step_fateavatar_random_order.json 2>> $O/${tag}_bench.err
python $R/tools/train_synthetic.py --fateavatar --views-per-step 4 --random-order > $O/${tag}_train_step_fateavatar_batch4_random_order.json 2>> $O/${tag}_bench.err
$R/tools/profile.sh ${tag}_fa1 python $R/tools/train_synthetic.py --fateavatar --steps 300 > /dev/null 2>&1
$R/tools/profile.sh ${tag}_fa4 python $R/tools/train_synthetic.py --fateavatar --steps 300 --views-per-step 4 > /dev/null 2>&1
FR_DIST_BACKEND=gloo python $R/bench.py --gpus 2 --steps 30 --warmup 5 > $O/${tag}_bench_2ranks_gloo_1gpu.json 2>> $O/${tag}_bench.err
# the N > 1 step on a one-rank RCCL group: what the exchange machinery costs apart from the wire
python $R/bench.py --exchange-at-1 --cpu-seconds 0 2>> $O/${tag}_bench.err | grep '^{' > $O/${tag}_bench_exchange_at_1.json
# per-kernel durations: one frame at a time (the roofline figures are per ISOLATED launch), then the default command
$R/tools/profile.sh ${tag}_eager python $R/bench.py --steps 50 --warmup 10 --no-dp-reference --cpu-seconds 0 --no-graph --in-flight 1 > /dev/null 2>&1
$R/tools/profile.sh ${tag}_graph python $R/bench.py --steps 100 --warmup 10 --no-dp-reference --cpu-seconds 0 --in-flight 1 > /dev/null 2>&1
$R/tools/profile.sh ${tag}_graph3 python $R/bench.py --steps 100 --warmup 10 --no-dp-reference --cpu-seconds 0 > /dev/null 2>&1
python $R/bench.py --in-flight 1 --no-dp-reference --cpu-seconds 0 > $O/${tag}_bench_one_at_a_time.json 2>> $O/${tag}_bench.err
$R/tools/pmc.sh ${tag}_fetch "FETCH_SIZE" python $R/tools/probe.py --iters 10 > /dev/null 2>&1
$R/tools/pmc.sh ${tag}_write "WRITE_SIZE" python $R/tools/probe.py --iters 10 > /dev/null 2>&1
$R/tools/pmc.sh ${tag}_sq1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" python $R/tools/probe.py --iters 10 > /dev/null 2>&1
$R/tools/pmc.sh ${tag}_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" python $R/tools/probe.py --iters 10 > /dev/null 2>&1
python $R/tools/probe_coherent.py > $O/${tag}_coherent_order.txt 2>/dev/null
python $R/tools/cpu_baseline.py > $O/${tag}_cpu_baseline.txt 2>/dev/null
bash $R/tools/diag/bwd_stats.sh > $O/${tag}_bwd_stats.txt 2>/dev/null
FR_HIP_LIB=$R/.ab/libfr_trace.so python $R/tools/diag/bwd_phases.py > $O/${tag}_bwd_phases.txt 2>/dev/null
{ for a in "" "--P 500000 --res 1024"; do echo "== tools/diag/pre_phases.py $a"; FR_HIP_LIB=$R/.ab/libfr_pretrace.so python $R/tools/diag/pre_phases.py $a 2>/dev/null | tail -17; done; } > $O/${tag}_pre_phases.txt
ls $O | grep $tag
