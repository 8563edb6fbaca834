# Round-4 profile set (run on the GPU box: bash tools/prof_r04.sh [part]).  Everything lands in gpurun_out/prof_r04/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
part=${1:-all}
# Every PROFILED pass runs the lookahead as ONE stream group (APMF_LOOKAHEAD_STREAMS=1, set explicitly on its command line:
# the package itself never looks at the profiler's environment).  With three groups - six host threads submitting to streams
# that share the default 4 hardware queues - ROCm 7.2's rocprofv3 faults in its own AQL packet interceptor
# (librocprofiler-sdk.so, reached from the HSA runtime's queue intercept inside hipLaunchKernel; fault address = one past
# a queue's ring buffer): tools/repro_trace_segv.sh reproduces it and prints the back trace, and the same run with
# GPU_MAX_HW_QUEUES=8 (every stream a hardware queue of its own) completes - DESIGN.md section 4, K6.  Unprofiled runs use
# the default three groups.  Counter totals (flops, bytes) do not depend on the grouping.
stats() { f=$(find $1 -name "*kernel_stats.csv" | head -1); cp "$f" $2; rm -rf $1; }
ctr() { f=$(find $1 -name "*counter_collection.csv" | head -1); cp "$f" $2; rm -rf $1; }
if [ $part = all ] || [ $part = bench ] || [ $part = traces ]; then
  (cd $R && python3 bench.py > $O/r04_bench_default.json 2> $O/bench_default.err)
  APMF_LOOKAHEAD_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/r04_bench_profiled.json 2> $O/bench_profiled.err
  stats $O/kt_bench $O/r04_bench_kernel_stats.csv
  echo bench done
fi
if [ $part = all ] || [ $part = pmck2 ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    APMF_LOOKAHEAD_STREAMS=1 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-class-level --no-lookahead > $O/pmc_bench_$c.json 2> $O/pmc_bench_$c.err
    ctr $O/pmc_$c $O/r04_pmc_$c.csv
  done
  python3 $R/tools/pmc_summary.py "k_mn_grid<float" $O/r04_pmc_FETCH_SIZE.csv $O/r04_pmc_WRITE_SIZE.csv 3248386048 > $O/r04_pmc_k2.json
  python3 $R/tools/pmc_summary.py "k_mn_grid<double" $O/r04_pmc_FETCH_SIZE.csv $O/r04_pmc_WRITE_SIZE.csv > $O/r04_pmc_k2_f64_both_legs.json
  echo pmck2 done
fi
if [ $part = traces ]; then
  APMF_LOOKAHEAD_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_la -o kt -- python3 $R/tools/bench_lookahead.py 100 > $O/lookahead_H500_profiled.json 2> $O/la_profiled.err
  stats $O/kt_la $O/r04_lookahead_H500_kernel_stats.csv
  echo traces done
fi
if [ $part = all ] || [ $part = lookahead ]; then
  (cd $R && python3 tools/bench_lookahead.py 100 > $O/lookahead_H500.json 2>/dev/null)
  APMF_LOOKAHEAD_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_la -o kt -- python3 $R/tools/bench_lookahead.py 100 > $O/lookahead_H500_profiled.json 2> $O/la_profiled.err
  stats $O/kt_la $O/r04_lookahead_H500_kernel_stats.csv
  APMF_LOOKAHEAD_STREAMS=1 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/pmc_la_sq -o pmc -- python3 $R/tools/bench_lookahead.py 100 > $O/la_sq.json 2> $O/la_sq.err
  ctr $O/pmc_la_sq $O/la_sq.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    APMF_LOOKAHEAD_STREAMS=1 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc_la_$c -o pmc -- python3 $R/tools/bench_lookahead.py 100 > $O/la_$c.json 2> $O/la_$c.err
    ctr $O/pmc_la_$c $O/la_$c.csv
  done
  tail -1 $O/lookahead_H500.json > $O/la_run.json
  python3 $R/tools/pmc_lookahead.py $O/la_run.json $O/la_sq.csv $O/la_FETCH_SIZE.csv $O/la_WRITE_SIZE.csv > $O/r04_pmc_lookahead.json
  rm -f $O/la_sq.csv $O/la_FETCH_SIZE.csv $O/la_WRITE_SIZE.csv
  echo lookahead done
fi
if [ $part = all ] || [ $part = tools ]; then
  (cd $R && APMF_C4_SKIP_SEQ=1 python3 tools/run_c4.py --refit-lookahead > $O/r04_configs3_end_to_end.json 2> $O/c4.err)
  (cd $R && python3 tools/c5_scale.py > $O/c5.txt 2>&1; python3 - <<PY
import json,re
t=open('$O/c5.txt').read()
open('$O/r04_c5_scale.json','w').write(t[:t.rindex('}')+1])
PY
)
  (cd $R && python3 tools/bench_batch.py 512 12 > $O/r04_batch_H512.json; python3 tools/bench_batch.py 64 12 > $O/r04_batch_H64.json; python3 tools/bench_batch.py 1 100 > $O/r04_batch_H1.json)
  echo tools done
fi
ls -la $O
