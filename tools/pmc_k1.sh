cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_k1; rm -rf $OUT; mkdir -p $OUT
ARGS="--cpu-baseline off --steps 2 --warmup 1 --profile-steps 1"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc pass $i failed"
done
python tools/pmc_summary.py $OUT | grep -A30 "^k_project_bin" | head -32
