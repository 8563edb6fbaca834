# all-active throughput of the batch engine at three batch sizes + the H = 500 lookahead (value must stay 150778.65773656353)
for h in 512 128 32; do timeout -k 10 120 python tools/bench_batch.py $h 12 | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('H', r['H'], 'us/hyp-step %.2f' % r['us_per_hypothesis_step'], 'kl %.9f' % r['kl_mean'])"; done
timeout -k 10 120 python tools/bench_lookahead.py 100 2>/dev/null | tail -1
