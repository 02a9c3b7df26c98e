#!/bin/bash
# Runs bench.py over the BASELINE.json configurations that fit one GPU; one JSON line each -> gpurun_out/bench_matrix.jsonl
mkdir -p gpurun_out; out=gpurun_out/bench_matrix.jsonl; : > $out
run() { echo "# $*" >&2; python bench.py --cpu-seconds 0 --fp16-leg 0 "$@" 2>/dev/null | tail -1 >> $out; }
run --steps 6 --warmup 2                                              # headline: jpl.4096 min-sum f32 (fused)
LDPC_HIP_PATH=flood run --steps 3 --warmup 1 --batch 16384            # same on the generic flood path
run --steps 6 --warmup 2 --dtype f16pk                                # configs[3]: fp16 LLRs AND arithmetic, two frames per lane (LDPC_F16PK)
run --steps 6 --warmup 2 --dtype f16pk --ebn0 3                       # the same in the waterfall
run --steps 6 --warmup 2 --dtype f16pk --code jpl.1024.4.5            # and on configs[1]'s code
run --steps 6 --warmup 2 --dtype f16                                  # configs[3]: fp16 LLRs, fused (state on-chip in f32)
LDPC_HIP_PATH=flood run --steps 3 --warmup 1 --batch 16384 --dtype f16 # configs[3] on the flood path: fp16 lam/messages in HBM
run --steps 3 --warmup 1 --batch 16384 --variant tanh                 # tanh rule, fused
LDPC_HIP_PATH=flood run --steps 3 --warmup 1 --batch 16384 --variant tanh
run --steps 6 --warmup 2 --code jpl.1024.4.5                          # configs[1]
run --steps 3 --warmup 1 --batch 16384 --code jpl.1024.4.5 --variant tanh
run --steps 3 --warmup 1 --batch 32768 --code dvbs2like.64800.1.2 --rate none --schedule layered --dtype f16 --ebn0 2   # configs[4]: DVB-S2-shaped long code, layered, fp16 lam ON-CHIP + streamed records (r04)
run --steps 3 --warmup 1 --batch 32768 --code dvbs2like.64800.1.2 --rate none --schedule layered --dtype f16 --ebn0 3
run --steps 3 --warmup 1 --batch 32768 --code dvbs2like.64800.1.2 --rate none --schedule layered --ebn0 2     # the same with f32 lam and records in HBM
run --steps 3 --warmup 1 --batch 8192 --code dvbs2like.64800.1.2 --rate none --ebn0 2                            # the same code, flooding (frame-per-workgroup HBM kernel)
run --steps 4 --warmup 2 --schedule layered --ebn0 3                                                             # jpl.4096 layered ON-CHIP in the waterfall
run --steps 4 --warmup 2 --schedule layered --ebn0 2                                                             # ... and below it
run --steps 4 --warmup 2 --schedule layered --dtype f16pk --ebn0 3.4                                             # layered on-chip in packed fp16
run --steps 4 --warmup 2 --schedule layered --dtype f16pk --ebn0 3
run --steps 4 --warmup 2 --schedule layered --dtype f16pk --ebn0 2
LDPC_HIP_PATH=flood run --steps 3 --warmup 1 --batch 16384 --schedule layered --ebn0 3                           # jpl.4096 layered from HBM
run --steps 3 --warmup 1 --batch 65536 --ebn0 3                                                                  # jpl.4096 flooding (on-chip) at the same point
for db in 1 2 3 4; do run --steps 3 --warmup 1 --code 1920.1280.3.303 --rate none --variant tanh --ebn0 $db; done   # configs[2], generic on-chip kernel
LDPC_HIP_PATH=flood run --steps 3 --warmup 1 --code 1920.1280.3.303 --rate none --variant tanh --ebn0 1
run --steps 3 --warmup 1 --code 1920.1280.3.303 --rate none --variant minsum --ebn0 1
# the same code through its 5760 redundant checks (codes/1920.1280.A, E = 32 000): on-chip (150 KB of LDS per frame) and from HBM
for v in tanh minsum; do
  LDPC_HIP_PATH=fused run --steps 2 --warmup 1 --batch 16384 --code 1920.1280.A --rate none --variant $v --ebn0 1
  LDPC_HIP_PATH=flood run --steps 2 --warmup 1 --batch 16384 --code 1920.1280.A --rate none --variant $v --ebn0 1
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_matrix.jsonl'):
    d = json.loads(l)
    r = d['roofline']
    print(f"{d['config']['code_name']:52s} {d['config']['path']:5s} {d['dtype']} {d['metric'].split('Eb/N0=')[1]:6s} B={d['config']['batch_per_gpu']:6d} {d['value']:9.1f} Mbit/s  {d['ms_per_step']:8.2f} ms  {r['bound']} roofline {('%.3f' % r['frac']) if r['frac'] is not None else 'n/a'}  ber {d['ber']:.3e} fer {d['fer']:.3f} it {d['mean_iters']:.1f}  {r['kernel'][:46]}")
PY
