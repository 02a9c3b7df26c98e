#!/bin/bash
# BER / FER / mean sweeps on the DVB-S2-shaped long code (BASELINE configs[4]): layered min-sum with fp16 lam ON-CHIP and streamed row
# records (csrc/layered_lds.hip) next to the f32-state layered kernel and flooding, on the same frames (device frame source, same seed).
#   -> profiles/r04_ber_dvbs2like_f16_vs_f32.txt
M=${1:-32768}
python -m ecc_ldpc_amd.cli 1.6 1.8 2.0 2.2 2.4 2.6 ldpc/hip-minsum-layered-f16/dvbs2like.64800.1.2/50 ldpc/hip-minsum-layered/dvbs2like.64800.1.2/50 -m$M -b16384
python -m ecc_ldpc_amd.cli 2.0 2.4 ldpc/hip-minsum/dvbs2like.64800.1.2/50 -m8192 -b8192
