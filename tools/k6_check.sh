# quick K6 regression: the batch tests, then the H = 500 lookahead (value must stay 150778.65773656353) and the
# all-active throughput at two batch sizes
set -e
timeout -k 10 500 python -m pytest tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3
APMF_BATCH_STATS=1 timeout -k 10 120 python tools/bench_lookahead.py 100 2>&1 | grep -v "^\s" | tail -2
timeout -k 10 120 python tools/bench_batch.py 512 12 | tail -1
timeout -k 10 120 python tools/bench_batch.py 64 12 | tail -1
