# FETCH_SIZE per kernel of the batch engine at H = 512 (all active): does the XCD-aware tile mapping cut L2 misses?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_fetch; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/f -o pmc -- python3 $R/tools/bench_batch.py 512 6 > $O/run.log 2>&1
f=$(find $O/f -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
tot=collections.defaultdict(float); cnt=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:30]
    tot[n]+=float(r['Counter_Value']); cnt[n]+=1
for n in sorted(tot, key=lambda k:-tot[k])[:10]:
    print('%-32s launches %5d  FETCH x2 per launch %8.1f MB' % (n, cnt[n], 2*tot[n]*1024/cnt[n]/1e6))
PY
rm -rf $O/f
