#!/bin/bash
# Timing-only ablation builds of the headline kernel (jpl.4096 min-sum f32 instance of fused_split.hip only):
#   tools/build_ablation.sh name1 "-DSPLIT_TID_EXEC=1" name2 "-DSPLIT_CH_TID=8 ..." ...  -> ablation/libldpc_hip_<name>.so
# Run with  LDPC_SO=ablation/libldpc_hip_<name>.so python bench.py --cpu-seconds 0   (the directory travels with gpurun).
set -e
cd "$(dirname "$0")/.."
O=ecc_ldpc_amd/build
[ -f $O/api.cc.o ] || python ecc_ldpc_amd/build.py >/dev/null
mkdir -p ablation
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -ffp-contract=off -fno-fast-math \
      -DSPLIT_ABLATION_MINSUM128 $f -x hip -c ecc_ldpc_amd/csrc/fused_split.hip -o ablation/fused_split.$n.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ablation/libldpc_hip_$n.so $O/api.cc.o $O/host.cc.o $O/batcher.cc.o $O/jit.cc.o $O/flood.hip.o \
      $O/layered_qc.hip.o $O/fused.hip.o $O/fused_msg.hip.o ablation/fused_split.$n.o $O/fused_csr.hip.o $O/sim.hip.o && echo built $n ) &
done
wait
