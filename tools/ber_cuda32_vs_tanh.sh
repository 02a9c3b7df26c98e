#!/bin/bash
# BER / FER / mean turns of the reference's CUDA plug-in arithmetic (LDPC_TANH_CUDA32, `cuda-arraylet2`) next to the Double-faithful tanh
# decoder (`hip-tanh`) on the same frames: where the float clamp of atanh_ (common.h:82-88) changes what is decoded.
#   -> profiles/r04_ber_cuda32_vs_tanh.txt
M=${1:-65536}
python -m ecc_ldpc_amd.cli 2.5 3.0 3.5 4.0 4.5 ldpc/cuda-arraylet2/jpl.1024.4.5/50/4/5 ldpc/hip-tanh/jpl.1024.4.5/50/4/5 -m$M -b16384
python -m ecc_ldpc_amd.cli 1.0 2.0 3.0 ldpc/cuda-arraylet2/1920.1280.3.303/50 ldpc/hip-tanh/1920.1280.3.303/50 -m$M -b16384
