# PMC passes over K4 alone (tools/bench_k4.py at 943 x 1682, d = 15, S = 128).  Run on the GPU box: bash tools/prof_k4.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_k4
mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  # (a third set with FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum ... aborted inside rocprofv3 and hung its
  # finalisation on this image: not collected)
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/p$i -o pmc -- python3 $R/tools/bench_k4.py 128 15 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_by_kernel.py "$f" > $O/pmc_$i.txt 2>&1
  rm -rf $O/p$i
done
cat $O/pmc_*.txt
