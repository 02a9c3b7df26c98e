#!/bin/bash
# tools/profile.sh TAG [bench.py args...]  -- run on the GPU box (through gpurun).
# Collects, for one bench.py workload, the evidence bench.py's roofline object refers to:
#   gpurun_out/prof_TAG/TAG_kernel_stats.csv          rocprofv3 --kernel-trace --stats (per-kernel durations)
#   gpurun_out/prof_TAG/TAG_bench_under_rocprof.json  the bench line printed by that same run (HIP-event timing)
#   gpurun_out/prof_TAG/TAG_pmc.json                  per-dispatch means of the PMC counters, separate --pmc passes
# Copy what should be judged into profiles/ afterwards.
set -o pipefail
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp
args="--cpu-seconds 0 --proof 0 --fp16-leg 0 --live-traffic 0 --steps 4 --warmup 2 $*"   # (--fp16-leg 0: only the workload named, so that "the dominant kernel" is its kernel)
# one UNPROFILED run first: whatever the workload needs from the tool chain (a run-time specialised kernel's code object,
# its <kernel>.isa.json) is built and cached now, so no hipcc child is ever started under the profiler's preload
echo "[profile] warm the caches (unprofiled)" >&2
python3 bench.py --cpu-seconds 0 --proof 0 --fp16-leg 0 --steps 1 --warmup 0 $* > /dev/null 2> $out/warm.log || { tail -5 $out/warm.log; exit 1; }
echo "[profile] kernel trace" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $args > $out/${tag}_bench_under_rocprof.json 2> $out/kt.log || { tail -5 $out/kt.log; exit 1; }
cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats.csv
pass=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    pass=$((pass+1)); echo "[profile] pmc pass $pass: $ctrs" >&2
    rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc$pass -o pmc -- python3 bench.py $args > /dev/null 2> $out/pmc$pass.log || { tail -5 $out/pmc$pass.log; exit 1; }
done
python3 tools/summarize_pmc.py $tag $out "$args"
