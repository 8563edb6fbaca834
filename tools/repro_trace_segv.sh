# Diagnosis of the round-3 segfault: `rocprofv3 --kernel-trace` around a lookahead advancing as THREE stream groups
# (3 group threads + up to 3 slow-lane solver threads launching at once).  Run on the GPU box:
#   bash tools/repro_trace_segv.sh [groups]
# (GPU_MAX_HW_QUEUES=8 bash tools/repro_trace_segv.sh 3: every stream a hardware queue of its own - completes.)
# Writes gpurun_out/segv/trace_g<groups>_q<queues>.{out,err}: on a fault, the native back trace of the faulting thread
# (tools/segv_bt.c) and the Python stacks of all threads (faulthandler).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/segv
mkdir -p $O
G=${1:-3}
Q=${GPU_MAX_HW_QUEUES:-default}
T=g${G}_q$Q
gcc -shared -fPIC -O1 -g -o /tmp/libsegvbt.so $R/tools/segv_bt.c || exit 1
export APMF_SEGV_BT=/tmp/libsegvbt.so APMF_LOOKAHEAD_STREAMS=$G
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$T -o kt -- python3 $R/tools/bench_lookahead.py 100 > $O/trace_$T.out 2> $O/trace_$T.err
echo "groups=$G hw_queues=$Q rc=$?" | tee -a $O/summary.txt
f=$(find $O/kt_$T -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_$T.csv
rm -rf $O/kt_$T
tail -c 6000 $O/trace_$T.err
