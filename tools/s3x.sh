# two-level sort: where does the sort kernel's time go?  Timing-only builds (SLICER_S3X: results are wrong)
cd $GRAFT_REPO_ROOT
for x in "$@"; do
  make -C slicer_amd/csrc -B EXTRA="-DSLICER_S3X=$x" > /dev/null 2>&1
  echo "S3X=$x"; bash tools/env_bench.sh "SLICER_SORT2=1"
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
