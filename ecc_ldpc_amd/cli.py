"""ecc-ldpc-like command line over libldpc_hip.so.

Mirrors how the reference's executable is driven (main/Main.hs:38-40, NOTES.txt:2-3):
    ecc-ldpc <Eb/N0 values ...> <code names ...> [-m<frames>]
e.g.  python -m ecc_ldpc_amd.cli 2 3 4 ldpc/hip-minsum/jpl.1024.4.5/50/4/5 -m65536
Code names use the reference's grammar ldpc/<decoder>/<matrix>/<max-rounds>[/x/y] (Utils.hs:82-88,
100-108) with <decoder> in {hip-tanh, hip-minsum}[-f32|-f64|-f16].  One row per (code, Eb/N0), shaped
like eccPrinter's (NOTES.txt:3):   seconds  name  Eb/N0  frames  bit-errors  BER   [+ FER, mean iterations, Mbit/s]
The external tester's statistics (confidence intervals, stopping rule) are not reproduced: frames are
decoded in device batches until -m frames are done.  Frames come from the library's device frame source
(counter-based RNG; channel model stated in DESIGN.md section 3.3)."""
from __future__ import annotations

import os
import sys
import time


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ebn0s, names, frames, batch, seed = [], [], 65536, 16384, 0x5EEDC0DE
    for a in argv:
        if a.startswith("-m"):
            frames = int(a[2:])
        elif a.startswith("-b"):
            batch = int(a[2:])
        elif a.startswith("-s"):
            seed = int(a[2:], 0)
        else:
            try:
                ebn0s.append(float(a))
            except ValueError:
                names.append(a)
    if not ebn0s or not names:
        print(__doc__)
        return 2
    import torch
    import ecc_ldpc_amd as E
    E.init(0)
    dev = torch.device("cuda", 0)
    codes_dir = os.environ.get("LDPC_CODES_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "codes"))
    batch = min(batch, frames)
    for name in names:
        try:
            ecc = E.ECC(codes_dir, name, max_batch=batch)
        except E.LdpcError as e:
            if e.code == E._lib.ENOTFOUND:
                print(f"# {name}: no such code ({e})", file=sys.stderr)
                continue
            raise
        k, N = ecc.message_length, ecc.code.N
        llr = torch.empty((batch, N), dtype=torch.float32, device=dev)
        bits = torch.empty((batch, N), dtype=torch.uint8, device=dev)
        iters = torch.empty((batch,), dtype=torch.int32, device=dev)
        conv = torch.empty((batch,), dtype=torch.uint8, device=dev)
        tally = torch.zeros(4, dtype=torch.int64, device=dev)
        # one explicit (non-default) stream for generate -> decode -> tally: with a NULL handle the decoder
        # would use its own non-blocking stream and race the frame source on the default stream
        stream = torch.cuda.Stream(device=dev)
        sp = stream.cuda_stream
        for db in ebn0s:
            tally.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = 0
            while done < frames:
                b = min(batch, frames - done)
                ecc.sim.generate(seed, done, b, db, llr.data_ptr(), None, sp)
                ecc.decoder.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), b, ecc.max_iters, iters.data_ptr(), conv.data_ptr(), sp)
                ecc.sim.tally(b, bits.data_ptr(), iters.data_ptr(), tally.data_ptr(), sp)
                done += b
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            f, fe, be, it = tally.tolist()
            ber = be / max(f * k, 1)
            print(f"{dt:8.2f} {ecc.name}  {db:4.2f} {f:8d} {be:8d}  {ber:.2e}   FER {fe / max(f, 1):.2e}  iters {it / max(f, 1):5.1f}  {f * k / dt / 1e6:9.1f} Mbit/s [{ecc.decoder.path}]",
                  flush=True)
        ecc.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
