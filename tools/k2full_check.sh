timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "full_grid" 2>&1 | tail -3
timeout -k 10 300 python tools/bench_k2full.py > gpurun_out/k2full.json 2>gpurun_out/k2full.err; tail -3 gpurun_out/k2full.err
python -c "
import json
for r in json.load(open('gpurun_out/k2full.json')): print(r['n'],r['m'],r['d'],{k:(round(v['kernel_ms'],4),round(v['frac_of_8TBs'],3)) for k,v in r.items() if isinstance(v,dict)})
"
