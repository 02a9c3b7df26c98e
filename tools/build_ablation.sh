#!/bin/bash
# Timing-only ablation builds of the fused kernel: tools/build_ablation.sh 1 2 4 ... -> gpurun_out/abl/libldpc_hip_dbgN.so
# Run with  LDPC_SO=gpurun_out/abl/libldpc_hip_dbgN.so python bench.py ...   (results are wrong by construction)
set -e
cd "$(dirname "$0")/.."
python ecc_ldpc_amd/build.py >/dev/null
mkdir -p ablation
for d in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DLDPC_DBG=$d -x hip -c ecc_ldpc_amd/csrc/fused_msg.hip -o ablation/fused_msg.dbg$d.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ablation/libldpc_hip_dbg$d.so ecc_ldpc_amd/build/api.cc.o ecc_ldpc_amd/build/host.cc.o ecc_ldpc_amd/build/flood.hip.o ecc_ldpc_amd/build/fused.hip.o ecc_ldpc_amd/build/sim.hip.o ablation/fused_msg.dbg$d.o && echo built dbg$d ) &
done
wait
