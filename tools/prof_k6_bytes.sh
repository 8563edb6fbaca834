# HBM bytes per kernel of the batch engine with every hypothesis active (tools/bench_batch.py 512 6): FETCH_SIZE and
# WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, HBM section), summed per kernel.  bash tools/prof_k6_bytes.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_k6_bytes
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/p_$c -o pmc -- python3 $R/tools/bench_batch.py 512 6 > $O/$c.log 2>&1
  f=$(find $O/p_$c -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_by_kernel.py "$f" > $O/by_kernel_$c.txt 2>&1
  rm -rf $O/p_$c
done
cat $O/by_kernel_FETCH_SIZE.txt $O/by_kernel_WRITE_SIZE.txt
