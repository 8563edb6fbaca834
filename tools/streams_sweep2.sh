for q in 8; do for s in 2 3 4 6; do echo "hwq $q streams $s: $(GPU_MAX_HW_QUEUES=$q APMF_LOOKAHEAD_STREAMS=$s timeout -k 10 200 python tools/bench_lookahead.py 100 2>/dev/null | tail -1)"; done; done
