#!/usr/bin/env python3
"""Development aid: stand-alone translation unit of layered_reg_body.h for one .q code (compile time / register check).
usage: tools/gen_lreg_test.py codes/<name>/H.q out.hip"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import formats

sz, words = formats.read_qc(open(sys.argv[1]).read())
off = formats.qc_offsets(sz, words)
off = np.asarray(off)
R, C = off.shape
lbeg, bc, rot = [0], [], []
for br in range(R):
    for c in range(C):
        if off[br, c] >= 0:
            bc.append(c)
            rot.append(int(off[br, c]))
    lbeg.append(len(bc))
NH = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n0 = int(sys.argv[4]) if len(sys.argv) > 4 else C // NH
# group 0 (the row workers) keeps n0 columns spread evenly, the other groups share the rest
own = [0 if (NH == 1 or (i * n0) // C != ((i + 1) * n0) // C) else 1 + (i % (NH - 1)) for i in range(C)]
src = f'''#include <hip/hip_runtime.h>
#include "layered_reg_body.h"
namespace ldpc {{
struct JT {{
    static constexpr int SZ = {sz}, NBR = {R}, NBC = {C}, NEDGE = {len(bc)}, NH = {NH};
    static constexpr int lbeg[{R + 1}] = {{{", ".join(map(str, lbeg))}}};
    static constexpr int bc[{len(bc)}] = {{{", ".join(map(str, bc))}}};
    static constexpr int rot[{len(bc)}] = {{{", ".join(map(str, rot))}}};
    static constexpr int own[{C}] = {{{", ".join(map(str, own))}}};
}};
}}
extern "C" __global__ __launch_bounds__({NH * ((sz + 63) // 64 * 64)}, 1) void lreg_test(ldpc::LregArgs A) {{ ldpc::layered_reg_body<ldpc::JT>(A); }}
'''
open(sys.argv[2], "w").write(src)
print(f"sz {sz}, {R} x {C} blocks, {len(bc)} circulants, max row weight {max(np.diff(lbeg))}")
