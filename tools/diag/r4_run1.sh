cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for v in prev hip; do
  L=$R/.ab/libfr_$v.so; [ $v = hip ] && L=$R/fateavatar_amd/libfr_hip.so
  FR_HIP_LIB=$L tools/profile.sh ab_$v python $R/tools/probe.py > /dev/null 2>&1
  echo "== c2 $v"; grep -E "blend" gpurun_out/ab_$v/kernels.txt
  FR_HIP_LIB=$L tools/pmc.sh absq_$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" python $R/tools/probe.py --iters 10 > /dev/null 2>&1
  echo "== sq $v"; grep -A4 "bwd_sparse" gpurun_out/absq_$v/pmc.txt
  FR_HIP_LIB=$L tools/pmc.sh absq2_$v "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" python $R/tools/probe.py --iters 10 > /dev/null 2>&1
  echo "== sq2 $v"; grep -A8 "bwd_sparse" gpurun_out/absq2_$v/pmc.txt
  FR_HIP_LIB=$L tools/profile.sh ab5_$v python $R/tools/probe.py --P 500000 --res 1024 --iters 20 > /dev/null 2>&1
  echo "== c5 $v"; grep -E "blend" gpurun_out/ab5_$v/kernels.txt
  FR_HIP_LIB=$L tools/profile.sh abo_$v python $R/tools/probe.py --opacity 0.9 --iters 30 > /dev/null 2>&1
  echo "== opaque $v"; grep -E "blend" gpurun_out/abo_$v/kernels.txt
done
