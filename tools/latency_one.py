"""Per-frame latency of ldpc_decode_one (the entry point the Haskell per-frame closure binds)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecc_ldpc_amd as E
E.init(0)
for name in ("ldpc/hip-minsum/jpl.4096.4.5/50/4/5", "ldpc/hip-tanh/jpl.4096.4.5/50/4/5", "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", "ldpc/hip-tanh/1920.1280.3.303/50"):
    ecc = E.ECC("codes", name, max_batch=1)
    rng = np.random.default_rng(0)
    for db, label in ((2.0, "2 dB (runs all 50 turns)"), (4.0, "4 dB (converges early)")):
        s2 = 1.0 / (2 * (ecc.message_length / ecc.codeword_length) * 10 ** (db / 10))
        llr = (2.0 * (-1.0 + rng.normal(0, np.sqrt(s2), ecc.codeword_length)) / s2)
        ecc.decode(llr)
        t0 = time.perf_counter(); n = 50
        for _ in range(n): ecc.decode(llr)
        dt = (time.perf_counter() - t0) / n
        print(f"{ecc.name:44s} {ecc.decoder.path:5s} {label:28s} {dt * 1e6:8.1f} us per frame")
    ecc.close()
os._exit(0)
