#!/bin/bash
# A/B of the on-chip layered kernels (jpl.4096, 65 536 frames): rows split between the wave groups (the build's default, r03) against
# block rows dealt alternately (-DLAY_ROW_SPLIT=0), and the packed-fp16 kernel at 4 waves per SIMD (-DLAYERED_PK16_WAVES_PER_EU=4).
# Build the variants first (ablation/libldpc_hip_lay_{rs0,w4}.so: see the hipcc lines in profiles/README.md), run through gpurun.
for so in "" ablation/libldpc_hip_lay_rs0.so ablation/libldpc_hip_lay_w4.so; do
  for dt in f32 f16pk; do
    for db in 2 3; do
      LDPC_SO=$so python3 bench.py --schedule layered --dtype $dt --ebn0 $db --cpu-seconds 0 --proof 0 --fp16-leg 0 --steps 6 --warmup 2 2>/dev/null |
        python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-36s %-6s %d dB  %9.1f Mbit/s  %7.3f ms  ber %.3e' % ('${so:-default (row split)}', '$dt', $db, d['value'], d['ms_per_step'], d.get('ber', float('nan'))))"
    done
  done
done
