for it in 0 1 2 5 10 25 50; do
python bench.py --cpu-seconds 0 --proof 0 --steps 4 --warmup 2 --dtype f16pk --iters $it 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16pk iters=$it', d['ms_per_step'], 'ms/step', d['roofline']['avg_launch_ms'], 'ms/launch')"
done
