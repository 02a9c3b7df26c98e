import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ecc_ldpc_amd as hip
from tests.helpers import CODES, load
hip.init(0)
ecc = hip.ECC(CODES, "ldpc/hip-minsum-f16pk/jpl.4096.4.5/50/4/5", max_batch=4096)
c = load("jpl.4096.4.5")
dev = torch.device("cuda", 0)
B, N, k = 4095, c.N, 4096
llr = torch.empty((B, N), dtype=torch.float16, device=dev)
msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
its = torch.empty((B,), dtype=torch.int32, device=dev)
conv = torch.empty((B,), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ecc.sim.generate(7, 0, B, float(sys.argv[1]) if len(sys.argv) > 1 else 2.8, llr.data_ptr(), msg.data_ptr(), None, llr_f16=True)
ecc.decoder.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, 50, its.data_ptr(), conv.data_ptr(), None, llr_f16=True)
torch.cuda.synchronize()
b, cv, it = bits.cpu().numpy(), conv.cpu().numpy().astype(bool), its.cpu().numpy()
hard_in = (llr > 0).to(torch.uint8).cpu().numpy()
bad = [f for f in np.flatnonzero(~cv) if not np.array_equal(b[f], hard_in[f])]
print("failed frames", (~cv).sum(), "with wrong bits", len(bad), "iters hist", np.bincount(it)[:60].tolist(), "msg errors in converged", int((b[cv][:, :k] != msg.cpu().numpy()[cv]).sum()))
for f in bad[:10]:
    d = np.flatnonzero(b[f] != hard_in[f])
    print("frame", f, "iters", it[f], "partner", f ^ 1, "partner conv", cv[f ^ 1], "partner iters", it[f ^ 1], "ndiff", len(d), "block cols of diffs", sorted(set((d // 128).tolist()))[:50],
          "equals partner bits:", np.array_equal(b[f], b[f ^ 1]), "is codeword-ish (msg match):", (b[f][:k] == msg[f].cpu().numpy()).mean())
