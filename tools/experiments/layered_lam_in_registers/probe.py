import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecc_ldpc_amd as E
from tests.helpers import SyntheticQC
import torch
E.init(0)
dev = torch.device("cuda", 0)
for R, w in ((6, 7), (12, 7), (24, 7), (45, 7), (90, 7)):
    rng = np.random.default_rng(5)
    C = 180
    mask = np.zeros((R, C), bool)
    for br in range(R):
        mask[br, rng.choice(C, w, replace=False)] = True
    c = SyntheticQC(f"probe-{R}x{C}", 360, np.where(mask, rng.integers(0, 360, mask.shape), -1).astype(np.int32))
    code = c.hip_code(E)
    F = 4096
    for reg in ("1", "0"):
        os.environ["LDPC_LAYERED_REG"] = reg
        dec = E.Decoder(code, "min", "f32", F, schedule="layered")
        llr = torch.randn((F, c.N), dtype=torch.float32, device=dev) * 0.5 + 0.2
        bits = torch.empty((F, c.N), dtype=torch.uint8, device=dev)
        it = torch.empty((F,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.time()
            dec.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), F, 20, it.data_ptr(), None, None)
            dec.synchronize()
            dt = time.time() - t0
        sw = float(it.float().mean())
        per_layer_clk = dt / (F / 256) / max(sw, 1) / R * 2.4e9
        print(f"R={R:3d} reg={reg} {dec.kernel_name[:36]:36s} {dt*1e3:8.2f} ms  mean sweeps {sw:5.1f}  -> {per_layer_clk:8.0f} clk per layer-sweep per CU-frame", flush=True)
        del dec
