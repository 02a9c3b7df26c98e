for n in "" np4w4 np3w4 np4w5 np2w4; do
for db in 2 3; do
so=""; [ -n "$n" ] && so="ablation/libldpc_hip_$n.so"
LDPC_SO=$so python bench.py --cpu-seconds 0 --proof 0 --steps 6 --warmup 2 --dtype f16pk --ebn0 $db 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${n:-default}', '$db dB', d['value'], 'Mbit/s', r['avg_launch_ms'], 'ms/launch', 'threads', r['threads_per_workgroup'], 'ber %.3e' % d['ber'])"
done; done
