# N > 1 rehearsals on a ONE-GPU box (APMF_BENCH_SHARE_GPU: the ranks share device 0, RCCL refuses that, so the exchange
# is staged through the socket transport): plumbing only, the numbers mean nothing.
set -e
O=gpurun_out/rehearse; mkdir -p $O
export APMF_BENCH_SHARE_GPU=1
run() { timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus $1 "${@:3}" ; }
echo "== c5 N=1 (reduced)"; APMF_BENCH_SHARE_GPU= timeout -k 10 200 python bench.py --workload c5 --c5-rows 20000 --c5-ratings 5000000 --steps 10 --warmup 2 > $O/c5_n1.json 2> $O/c5_n1.err; tail -2 $O/c5_n1.err; cut -c1-600 $O/c5_n1.json
echo "== c5 N=2 (reduced)"; run 2 29611 --workload c5 --c5-rows 20000 --c5-ratings 5000000 --steps 10 --warmup 2 > $O/c5_n2.json 2> $O/c5_n2.err; tail -3 $O/c5_n2.err; cut -c1-600 $O/c5_n2.json
echo "== c2 strong N=2"; run 2 29631 --scaling strong --steps 20 --warmup 2 --no-cpu-baseline > $O/c2s_n2.json 2> $O/c2s_n2.err; tail -3 $O/c2s_n2.err; cut -c1-900 $O/c2s_n2.json
echo "== c2 weak N=3"; run 3 29651 --steps 20 --warmup 2 --no-cpu-baseline > $O/c2w_n3.json 2> $O/c2w_n3.err; tail -3 $O/c2w_n3.err; cut -c1-500 $O/c2w_n3.json
# the agreed fallback: two ranks on ONE device make RCCL's communicator fail (duplicate GPU); every rank must learn it,
# drop to the host-staged exchange together and say so in the line
echo "== c2 weak N=2, RCCL attempted on a shared device (expected: FALLBACK in the exchange note)"; APMF_BENCH_SHARE_GPU=rccl run 2 29671 --steps 20 --warmup 2 --no-cpu-baseline > $O/c2w_n2_fallback.json 2> $O/c2w_n2_fallback.err; grep "apmf comm" $O/c2w_n2_fallback.err | cut -c1-300; python3 -c "import json; o=json.load(open('$O/c2w_n2_fallback.json')); print(o['n_gpus'], o['value'], o['config']['exchange'][:200])"
