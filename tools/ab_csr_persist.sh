for db in 1 4; do for p in 1 0; do
LDPC_CSR_PERSIST=$p python bench.py --cpu-seconds 0 --steps 4 --warmup 2 --code 1920.1280.3.303 --rate none --variant tanh --ebn0 $db 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist=$p', d['config']['code_name'], '$db dB', d['value'], 'Mbit/s', d['ms_per_step'], 'ms', d['mean_iters'], d['roofline']['frac'], d['proof_of_work']['ok'])"
done; done
for p in 1 0; do
LDPC_CSR_PERSIST=$p python bench.py --cpu-seconds 0 --steps 4 --warmup 2 --code 1920.1280.3.303 --rate none --variant minsum --ebn0 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist=$p', d['config']['code_name'], '1 dB', d['value'], 'Mbit/s', d['ms_per_step'], 'ms', d['mean_iters'], d['roofline']['frac'], d['proof_of_work']['ok'])"
done
echo "# fixed stride instead of the work counter"
for db in 1 4; do
LDPC_CSR_DYNAMIC=0 python bench.py --cpu-seconds 0 --steps 4 --warmup 2 --code 1920.1280.3.303 --rate none --variant tanh --ebn0 $db 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('static stride', d['config']['code_name'], '$db dB', d['value'], 'Mbit/s', d['ms_per_step'], 'ms', d['mean_iters'], d['proof_of_work']['ok'])"
done
