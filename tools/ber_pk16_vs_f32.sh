#!/bin/bash
# BER / FER / mean iterations of the packed-fp16 decoders (LDPC_F16PK, flooding and layered) next to the f32 decoders on the SAME
# frames (device frame source, same seed; the fp16 decoders see the LLRs rounded to fp16): jpl.4096.4.5 rate 4/5 min-sum, 50 turns.
M=${1:-262144}
python -m ecc_ldpc_amd.cli 2.6 2.8 3.0 3.2 3.4 \
   ldpc/hip-minsum/jpl.4096.4.5/50/4/5 ldpc/hip-minsum-f16pk/jpl.4096.4.5/50/4/5 \
   ldpc/hip-minsum-layered/jpl.4096.4.5/50/4/5 ldpc/hip-minsum-layered-f16pk/jpl.4096.4.5/50/4/5 -m$M -b32768
