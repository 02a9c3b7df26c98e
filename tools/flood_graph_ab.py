"""A/B of the flood path's hipGraph replay (LDPC_FLOOD_GRAPH=0 disables it): wall time per decode at small batches,
where 2*max_iters+2 kernel launches dominate.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ecc_ldpc_amd as E
from tests.helpers import load
E.init(0)
c = load("jpl.4096.4.5")
code = c.hip_code(E, prefer_qc=False)          # as a plain CSR graph: too big for LDS -> flood path
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
for B in (1, 64, 1024):
    dec = E.Decoder(code, "min", "f32", B, path="flood")
    _, llr = c.frames(min(B, 8), 2.0, seed=5)
    x = torch.tensor(np.resize(llr, (B, c.N)), dtype=torch.float32, device=dev)
    bits = torch.empty((B, c.N), dtype=torch.uint8, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    for _ in range(3):
        dec.decode_batch_dev(x.data_ptr(), bits.data_ptr(), B, 50, it.data_ptr(), None, st.cuda_stream)
    st.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        dec.decode_batch_dev(x.data_ptr(), bits.data_ptr(), B, 50, it.data_ptr(), None, st.cuda_stream)
    st.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"graph={os.environ.get('LDPC_FLOOD_GRAPH', '1')} batch {B:5d}: {dt * 1e6:8.1f} us per 50-turn decode, checksum {int(bits.sum())} iters {int(it.sum())}", flush=True)
os._exit(0)
