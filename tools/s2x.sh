# two-level sort: where does the project+bin kernel's extra time go?  Timing-only builds (SLICER_S2X: results are wrong)
cd $GRAFT_REPO_ROOT
for x in "$@"; do
  make -C slicer_amd/csrc -B EXTRA="-DSLICER_S2X=$x" > /dev/null 2>&1
  echo "S2X=$x"; bash tools/env_bench.sh "SLICER_SORT2=1"
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
