"""PCIe-inclusive rate of the host-buffer entry point ldpc_decode_batch (numpy arrays in pageable host
memory -> bits back on the host), beside the device-resident rate bench.py reports."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecc_ldpc_amd as E
E.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ecc = E.ECC("codes", "ldpc/hip-minsum/jpl.4096.4.5/50/4/5", max_batch=B)
N, k, n_tx = ecc.code.N, ecc.message_length, ecc.codeword_length
rng = np.random.default_rng(1)
s2 = 1.0 / (2 * 0.8 * 10 ** 0.2)
llr = np.zeros((B, N), np.float32)
llr[:, :n_tx] = (2.0 * (-1.0 + rng.normal(0, np.sqrt(s2), (B, n_tx)).astype(np.float32)) / s2)
ecc.decoder.decode_batch(llr[:1024], 50)
pin_out = E.PinnedArray((B, N), np.uint8)
for dt_np, label in ((np.float32, "f32 LLRs"), (np.float16, "fp16 LLRs")):
    src = llr.astype(dt_np)
    pin_in = E.PinnedArray((B, N), dt_np)
    pin_in.array[:] = src
    for rep in range(6):
        pinned = rep >= 3
        t0 = time.perf_counter()
        if pinned:
            bits, its, conv = ecc.decoder.decode_batch(pin_in.array, ITERS, out_bits=pin_out.array)
        else:
            bits, its, conv = ecc.decoder.decode_batch(src, ITERS)
        dt = time.perf_counter() - t0
        print(f"ldpc_decode_batch host->host {label} ({'pinned  ' if pinned else 'pageable'}): {B} frames in {dt * 1e3:7.1f} ms = {B * k / dt / 1e6:8.1f} Mbit/s  "
              f"(H2D {src.nbytes / 1e6:.0f} MB, D2H {bits.nbytes / 1e6:.0f} MB; mean iters {its.mean():.1f})", flush=True)
    del pin_in
os._exit(0)
