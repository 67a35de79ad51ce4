# FETCH_SIZE / WRITE_SIZE passes of the default bench: bash tools/pmc_traffic.sh TAG [ENV...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
for v in "$@"; do export $v; done
OUT=gpurun_out/pmct_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--cpu-baseline off --parity off --e2e off --steps 2 --warmup 1 --profile-steps 1"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc pass $i failed"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
