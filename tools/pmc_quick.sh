# quick PMC passes of the default bench (SQ counters only): bash tools/pmc_quick.sh TAG [ENV...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
for v in "$@"; do export $v; done
OUT=gpurun_out/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--cpu-baseline off --parity off --e2e off --steps 2 --warmup 1 --profile-steps 1"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc pass $i failed"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
