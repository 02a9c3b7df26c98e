set -x
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; tail -c 600 gpurun_out/r02_bench_default.json
bash tools/profile.sh r02_final_jpl4096_f32_minsum > gpurun_out/p1.log 2>&1; tail -2 gpurun_out/p1.log
bash tools/profile.sh r02_final_jpl4096_f32_tanh --variant tanh --batch 16384 > gpurun_out/p2.log 2>&1; tail -2 gpurun_out/p2.log
bash tools/profile.sh r02_final_jpl1024_f32_minsum --code jpl.1024.4.5 > gpurun_out/p3.log 2>&1; tail -2 gpurun_out/p3.log
bash tools/profile.sh r02_final_floodqc_jpl4096_f32_minsum --path flood --batch 16384 > gpurun_out/p4.log 2>&1; tail -2 gpurun_out/p4.log
bash tools/profile.sh r02_final_dvbs2like_layered_f32_minsum --code dvbs2like.64800.1.2 --rate none --schedule layered --batch 8192 --ebn0 2 > gpurun_out/p5.log 2>&1; tail -2 gpurun_out/p5.log
bash tools/bench_matrix.sh > gpurun_out/r02_bench_matrix.txt 2>&1; tail -22 gpurun_out/r02_bench_matrix.txt
