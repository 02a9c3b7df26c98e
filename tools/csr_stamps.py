"""Where a frame's time goes in the configs[2] kernel (fused_csr_batched_kernel): a debug build with -DLDPC_CSR_STAMPS adds up the
cycles thread 0 of every workgroup spends in each phase.  Build + run (GPU box):
    python tools/csr_stamps.py --build        # -> ab/libldpc_stamps.so   (CPU, cross-compiles)
    LDPC_SO=$PWD/ab/libldpc_stamps.so python tools/csr_stamps.py
"""
import ctypes, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--build" in sys.argv:
    from ecc_ldpc_amd import build as b
    b.build(verbose=False)
    os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
    obj = "/tmp/fused_csr_stamps.o"
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DLDPC_CSR_STAMPS", "-x", "hip", "-c", os.path.join(b.CSRC, "fused_csr.hip"), "-o", obj])
    objs = [obj if s == "fused_csr.hip" else os.path.join(b.HERE, "build", s + ".o") for s in b.SOURCES]
    subprocess.check_call([b.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(ROOT, "ab", "libldpc_stamps.so")] + objs)
    sys.exit(0)

import numpy as np
import ecc_ldpc_amd as E
from tests.helpers import load

E.init(0)
L = ctypes.CDLL(E.SO_PATH)
c = load("1920.1280.3.303")
B = 65536
names = ["prologue (LLRs -> lam, barrier, ticket / DMA issue)", "turn loop", "epilogue (stores issued)", "frame end (waits, barrier, next frame id)"]
for db in (1.0, 4.0):
    _, llr = c.frames(4096, db, seed=11)
    llr = np.tile(llr.astype(np.float32), (B // 4096, 1))
    for stage in ("1", "0"):
        os.environ["LDPC_CSR_STAGE"] = stage
        for iters in (0, 1, 50):
            dec = E.Decoder(c.hip_code(E), "tanh", "f32", B, path="fused")
            dec.decode_batch(llr, iters)
            out = (ctypes.c_ulonglong * 8)()
            assert L.ldpc_debug_csr_stamps(out, 1) == 0
            _, its, _ = dec.decode_batch(llr, iters)
            assert L.ldpc_debug_csr_stamps(out, 1) == 0
            per = [out[i] / B for i in range(8)]
            print(f"ebn0={db} stage={stage} iters<={iters} mean turns {its.mean():.2f}: cycles per frame (thread 0 of the workgroup): "
                  + "; ".join(f"{n} {per[i]:.0f}" for i, n in enumerate(names)) + f"; other {per[7]:.0f}; sum {sum(per):.0f}", flush=True)
            dec.close()
