# the all-active throughput of the batch engine, repeated (run-to-run spread), H = 512 and 128
for r in 1 2 3 4; do for h in 512 128; do timeout -k 10 120 python tools/bench_batch.py $h 12 | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('H', r['H'], 'us/hyp-step %.2f' % r['us_per_hypothesis_step'])"; done; done
