#!/bin/bash
# BER / FER / mean iterations of the layered schedule against flooding on the same frames (device frame source,
# same seed): jpl.4096.4.5 rate 4/5 min-sum.  Layered at 25 sweeps vs flooding at 50 turns, and both at 50.
# Output: one eccPrinter-style row per (code, Eb/N0)  -> profiles/r02_ber_layered_vs_flooding.txt
M=${1:-65536}
python -m ecc_ldpc_amd.cli 2.6 2.8 3.0 3.2 3.4 3.6 \
   ldpc/hip-minsum/jpl.4096.4.5/50/4/5 ldpc/hip-minsum-layered/jpl.4096.4.5/50/4/5 ldpc/hip-minsum-layered/jpl.4096.4.5/25/4/5 ldpc/hip-minsum/jpl.4096.4.5/25/4/5 \
   -m$M -b16384
python -m ecc_ldpc_amd.cli 1.0 1.4 1.8 2.2 ldpc/hip-minsum-layered/dvbs2like.64800.1.2/50 ldpc/hip-minsum/dvbs2like.64800.1.2/50 -m8192 -b4096
