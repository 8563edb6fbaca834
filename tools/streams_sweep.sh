# the H = 500 lookahead (tools/bench_lookahead.py 100) by number of stream groups; the value must stay 150778.65773656353
for s in 1 2 3 4 6; do echo "streams $s: $(APMF_LOOKAHEAD_STREAMS=$s timeout -k 10 200 python tools/bench_lookahead.py 100 2>/dev/null | tail -1)"; done
