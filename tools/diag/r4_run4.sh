cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export FR_HEAVY_PAIRS=569
for v in hip abl2 abl4 abl8 abl16 abl30; do
  L=$R/.ab/libfr_$v.so; [ $v = hip ] && L=$R/fateavatar_amd/libfr_hip.so
  FR_HIP_LIB=$L tools/profile.sh abl_$v python $R/tools/probe.py > /dev/null 2>&1
  echo "== c2 $v $(grep -E 'blend_bwd' gpurun_out/abl_$v/kernels.txt)"
  FR_HIP_LIB=$L tools/profile.sh abl5_$v python $R/tools/probe.py --P 500000 --res 1024 --iters 20 > /dev/null 2>&1
  echo "== c5 $v $(grep -E 'blend_bwd' gpurun_out/abl5_$v/kernels.txt)"
done
