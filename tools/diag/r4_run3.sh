cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for hp in 641 569 500 400 300; do
  FR_HEAVY_PAIRS=$hp tools/profile.sh abh_$hp python $R/tools/probe.py > /dev/null 2>&1
  echo "== c2 heavy_pairs=$hp"; grep -E "blend_bwd" gpurun_out/abh_$hp/kernels.txt
done
for hp in 641 569 450; do
  FR_HEAVY_PAIRS=$hp tools/profile.sh abh5_$hp python $R/tools/probe.py --P 500000 --res 1024 --iters 20 > /dev/null 2>&1
  echo "== c5 heavy_pairs=$hp"; grep -E "blend_bwd" gpurun_out/abh5_$hp/kernels.txt
done
