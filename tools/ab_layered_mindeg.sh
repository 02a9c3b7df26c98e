for so in "" ablation/libldpc_hip_lay_md4.so; do
  for dt in f32 f16pk; do
    for db in 2 3; do
      LDPC_SO=$so python3 bench.py --schedule layered --dtype $dt --ebn0 $db --cpu-seconds 0 --proof 0 --fp16-leg 0 --steps 6 --warmup 2 2>/dev/null |
        python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-36s %-6s %d dB  %9.1f Mbit/s  %7.3f ms  ber %.3e' % ('${so:-default}', '$dt', $db, d['value'], d['ms_per_step'], d.get('ber', float('nan'))))"
    done
  done
done
