# configs[3] on the HBM path: fp16 state, frame-per-workgroup kernel (r03) against the batch-major pair (LDPC_FLOOD_QC=0) and the f32 state
for env in "" "LDPC_FLOOD_QC=0"; do
env $env LDPC_HIP_PATH=flood python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --batch 16384 --dtype f16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('f16 state', '$env', d['value'], 'Mbit/s', d['ms_per_step'], 'ms', r['kernel'][:44], 'hbm frac', r['frac'], 'proof', d['proof_of_work']['ok'])"
done
LDPC_HIP_PATH=flood python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --batch 16384 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('f32 state', d['value'], 'Mbit/s', d['ms_per_step'], 'ms', r['kernel'][:44], 'hbm frac', r['frac'], 'proof', d['proof_of_work']['ok'])"
