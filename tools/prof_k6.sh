# per-kernel profile of the batch engine with every hypothesis active (tools/bench_batch.py): kernel-trace stats at two
# batch sizes, the list of SQ counters, and one PMC pass.  Run on the GPU box: bash tools/prof_k6.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_k6
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/counters.txt | sort -u | grep -i "MFMA\|WAIT\|LDS\|BUSY\|WAVE_CYCLES\|ACTIVE_INST\|INSTS_VALU\b\|INSTS_VMEM\|INST_LEVEL\|OCCUP" > $O/sq_counters.txt
for H in 64 512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$H -o kt -- python3 $R/tools/bench_batch.py $H 12 > $O/kt_$H.log 2>&1
  f=$(find $O/kt_$H -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_H$H.csv
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS -d $O/pmc_512 -o pmc -- python3 $R/tools/bench_batch.py 512 6 > $O/pmc_512.log 2>&1
f=$(find $O/pmc_512 -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_by_kernel.py "$f" > $O/pmc_512_by_kernel.txt 2>&1
rm -rf $O/kt_64 $O/kt_512 $O/pmc_512
ls -la $O
