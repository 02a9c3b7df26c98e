# fixed (turn-independent) cost of the configs[2] kernel: launches with 0, 1, 2, 4 turns allowed
for p in 1 0; do for it in 0 1 2 4 8; do
LDPC_CSR_PERSIST=$p python bench.py --cpu-seconds 0 --proof 0 --steps 4 --warmup 2 --code 1920.1280.3.303 --rate none --variant tanh --ebn0 1 --iters $it 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist=$p iters=$it', d['ms_per_step'], 'ms/step', d['roofline']['avg_launch_ms'], 'ms/launch', d['mean_iters'])"
done; done
