cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_mix; rm -rf $OUT; mkdir -p $OUT
ARGS="--cpu-baseline off --steps 2 --warmup 1 --profile-steps 1 $BENCH_EXTRA"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc pass $i failed"
done
python tools/pmc_summary.py $OUT | grep -A26 "^k_project_bin" | head -28
