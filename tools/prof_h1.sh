# kernel time against wall time of the batch engine with ONE hypothesis (the launch-chain floor of an iteration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_h1
mkdir -p $O
python3 $R/tools/bench_batch.py 1 100 > $O/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/bench_batch.py 1 100 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_H1.csv
rm -rf $O/kt
tail -1 $O/plain.log; tail -1 $O/kt.log
python3 $R/tools/sumstats.py $O/kernel_stats_H1.csv > $O/sum.txt; cat $O/sum.txt
