# project+bin kernel: workgroup size / waves per SIMD variants (rebuilds on the box)
cd $GRAFT_REPO_ROOT
for v in "-DSLICER_K1_BLOCK=640 -DSLICER_K1_WAVES_PER_SIMD=5" "-DSLICER_K1_BLOCK=512 -DSLICER_K1_WAVES_PER_SIMD=4" "-DSLICER_K1_BLOCK=768 -DSLICER_K1_WAVES_PER_SIMD=3" "-DSLICER_K1_BLOCK=1024 -DSLICER_K1_WAVES_PER_SIMD=4"; do
  make -C slicer_amd/csrc -B EXTRA="$v" > /dev/null 2>&1
  echo "$v"; bash tools/env_bench.sh "SLICER_SORT2=0"
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
