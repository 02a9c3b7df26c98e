# A/B in one box, configs[2] kernel: the STAGED instance (the next frame's LLRs on their way into LDS by LDS-DMA while the current
# frame is decoded; default) against the r03 form (LDPC_CSR_STAGE=0); interleaved repetitions; then the turn-independent cost
# (0, 1, 2 turns allowed); then min-sum
run() { LDPC_CSR_STAGE=$1 python bench.py --cpu-seconds 0 --proof 0 --steps 6 --warmup 2 --code 1920.1280.3.303 --rate none --variant $2 --ebn0 $3 --iters $4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage=$1 $2 ebn0=$3 iters=$4', d['value'], 'Mbit/s', d['roofline']['avg_launch_ms'], 'ms/launch', d['roofline'].get('frac'))"; }
for rep in 1 2 3; do for st in 1 0; do for db in 1 4; do run $st tanh $db 50; done; done; done
for st in 1 0; do for it in 0 1 2; do run $st tanh 1 $it; done; done
for st in 1 0; do for db in 1 4; do run $st minsum $db 50; done; done
