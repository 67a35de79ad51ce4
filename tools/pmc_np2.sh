cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc4000; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python bench.py --cpu-baseline off --parity off --e2e off --steps 2 --warmup 1 --profile-steps 1 --npix 4000 > $OUT/p1.json 2> $OUT/p1.err
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p2 -- python bench.py --cpu-baseline off --parity off --e2e off --steps 2 --warmup 1 --profile-steps 1 --npix 4000 > $OUT/p2.json 2> $OUT/p2.err
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
awk '/^k_tile_deposit/{f=1} f{print} /SQ_WAVE_CYCLES/{if(f){exit}}' $OUT/pmc_summary.txt
