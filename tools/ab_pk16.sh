# packed-fp16 min-sum (LDPC_F16PK, two frames per lane) next to the f32 split kernel on the headline workload and in the waterfall
for db in 2 3; do for dt in f32 f16pk; do
python bench.py --cpu-seconds 0 --steps 6 --warmup 2 --dtype $dt --ebn0 $db 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$dt', d['config']['code_name'], '$db dB', d['value'], 'Mbit/s', d['ms_per_step'], 'ms  iters', round(d['mean_iters'],2), 'ber %.3e fer %.4f' % (d['ber'], d['fer']), 'valu frac', r.get('frac'), 'valu/wave-turn', r.get('valu_instr_per_wave_turn'), 'proof', d['proof_of_work']['ok'])"
done; done
python bench.py --cpu-seconds 0 --steps 6 --warmup 2 --dtype f16pk --code jpl.1024.4.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16pk', d['config']['code_name'], d['value'], 'Mbit/s', d['ms_per_step'], 'ms', d['proof_of_work']['ok'])"
