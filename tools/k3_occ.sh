# sort kernel: workgroup size variants (rebuilds on the box)
cd $GRAFT_REPO_ROOT
for v in "-DSLICER_K3_BLOCK=1024 -DSLICER_K3_WAVES=8" "-DSLICER_K3_BLOCK=1024 -DSLICER_K3_WAVES=8 -DSLICER_K3_STAGE=8192" "-DSLICER_K3_BLOCK=256 -DSLICER_K3_WAVES=2"; do
  make -C slicer_amd/csrc -B EXTRA="$v" > /dev/null 2>&1
  echo "$v"; bash tools/env_bench.sh "SLICER_SORT2=0" "SLICER_K3_PER_CU=3" "SLICER_K3_PER_CU=4"
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
