# PMC passes of the default bench for one build variant (run through gpurun): bash tools/pmc_variant.sh TAG "<EXTRA flags>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
OUT=gpurun_out/pmcv_$TAG
rm -rf $OUT; mkdir -p $OUT
make -C slicer_amd/csrc -B EXTRA="$2" > /dev/null 2>&1
ARGS="--cpu-baseline off --parity off --e2e off --steps 2 --warmup 1 --profile-steps 1"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc pass $i failed"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
make -C slicer_amd/csrc -B > /dev/null 2>&1
awk '/^k_bin_scatter/{f=1} /^k_build/{f=0} f{print}' $OUT/pmc_summary.txt
