# project+bin kernel: A/B of a compile-time variant inside one gpurun call (rebuilds on the box)
# usage: bash tools/k1_ab.sh "<EXTRA flags of variant A>" "<EXTRA flags of variant B>" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  make -C slicer_amd/csrc -B EXTRA="$v" > /dev/null 2>&1
  echo "[$v]"; bash tools/env_bench.sh "SLICER_SORT2=0" "SLICER_SORT2=0"
done
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
