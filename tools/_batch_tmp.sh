python -m pytest tests/test_layered_gpu.py tests/test_flood_gpu.py -m gpu -q 2>&1 | tail -3
for args in "--code dvbs2like.64800.1.2 --rate none --batch 32768 --ebn0 2.0 --schedule layered" "--code dvbs2like.64800.1.2 --rate none --batch 8192 --ebn0 2.0" "--code dvbs2like.64800.1.2 --rate none --batch 32768 --ebn0 2.6 --schedule layered"; do
 echo "== $args"; python bench.py $args --steps 3 --warmup 1 --cpu-seconds 0 --proof 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['kernel'], d['mean_iters'], d['fer'])"
done
bash tools/profile.sh r02_final_dvbs2like_layered_f32_minsum --code dvbs2like.64800.1.2 --rate none --schedule layered --batch 32768 --ebn0 2 > gpurun_out/p5.log 2>&1; tail -2 gpurun_out/p5.log
python bench.py --code dvbs2like.64800.1.2 --rate none --batch 32768 --ebn0 2.0 --schedule layered --steps 3 --warmup 1 --cpu-seconds 8 > gpurun_out/r02_bench_dvbs2like_layered.json 2>/dev/null; tail -c 1500 gpurun_out/r02_bench_dvbs2like_layered.json
