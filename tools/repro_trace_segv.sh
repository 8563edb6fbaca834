# Diagnosis of the round-3 segfault: `rocprofv3 --kernel-trace` around a lookahead advancing as THREE stream groups
# (3 group threads + up to 3 slow-lane solver threads launching at once).  Run on the GPU box:
#   bash tools/repro_trace_segv.sh [groups]
# Writes gpurun_out/segv/trace_g<groups>.{out,err}: on a fault, the native back trace of the faulting thread
# (tools/segv_bt.c) and the Python stacks of all threads (faulthandler).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/segv
mkdir -p $O
G=${1:-3}
gcc -shared -fPIC -O1 -g -o /tmp/libsegvbt.so $R/tools/segv_bt.c || exit 1
export APMF_SEGV_BT=/tmp/libsegvbt.so APMF_LOOKAHEAD_STREAMS=$G
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_g$G -o kt -- python3 $R/tools/bench_lookahead.py 100 > $O/trace_g$G.out 2> $O/trace_g$G.err
echo "groups=$G rc=$?" | tee -a $O/summary.txt
f=$(find $O/kt_g$G -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_g$G.csv
rm -rf $O/kt_g$G
tail -c 6000 $O/trace_g$G.err
