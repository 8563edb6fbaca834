set -e
for f in 32 100 200 400 1200; do
  echo "FUSE_JOBS=$f"; APMF_CHOL_FUSE_JOBS=$f APMF_BATCH_STATS=1 timeout -k 10 120 python tools/bench_lookahead.py 100 2>&1 | grep -v "^\s" | tail -2
done
