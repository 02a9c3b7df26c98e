#!/usr/bin/env python3
"""tools/encoder_rate.py [frames] -- device encoder throughput, quasi-cyclic rotate-and-xor (Fast/Encoder.hs:26-63, sim.hip
sim_parity_qc_kernel) against the dense packed GF(2) mat-vec of the expanded generator (Orig.hs:25-26), on the shipped AR4JA
codes: the encoder alone (ldpc_sim_encode_batch: messages + parity -> codeword bytes) and the whole frame source
(ldpc_sim_generate: + BPSK, AWGN, LLRs).  HIP events on the launch stream, median of 7."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ecc_ldpc_amd as E  # noqa: E402


def timed(fn, stream, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); fn(); b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    E.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    for name in ("ldpc/hip-minsum/jpl.1024.4.5/50/4/5", "ldpc/hip-minsum/jpl.4096.4.5/50/4/5"):
        rows = {}
        for enc in ("qc", "dense"):
            if enc == "dense":
                os.environ["LDPC_SIM_ENCODER"] = "dense"
            else:
                os.environ.pop("LDPC_SIM_ENCODER", None)
            ecc = E.ECC(os.path.join(ROOT, "codes"), name, max_batch=B)
            k, n_tx, N = ecc.message_length, ecc.codeword_length, ecc.unpunctured_length
            assert ecc.sim.encoder == enc
            cw = torch.empty((B, n_tx), dtype=torch.uint8, device=dev)
            llr = torch.empty((B, N), dtype=torch.float32, device=dev)
            t_enc = timed(lambda: ecc.sim.encode_batch(1, 0, B, cw.data_ptr(), None, st.cuda_stream), st)
            t_gen = timed(lambda: ecc.sim.generate(1, 0, B, 2.0, llr.data_ptr(), None, st.cuda_stream), st)
            rows[enc] = (t_enc, t_gen, cw.cpu().numpy().copy())
            print(f"{name:42s} {enc:5s} encoder: {B} frames  encode_batch {t_enc:7.3f} ms = {B / t_enc / 1e3:8.2f} Mframes/s = {B * k / t_enc / 1e6:8.1f} Gbit/s info"
                  f" | generate (encode + AWGN + LLR) {t_gen:7.3f} ms", flush=True)
            ecc.close()
            del cw, llr
        assert (rows["qc"][2] == rows["dense"][2]).all(), "encoders disagree"
        print(f"{'':42s} codewords identical; qc/dense time: encode {rows['qc'][0] / rows['dense'][0]:.2f}, generate {rows['qc'][1] / rows['dense'][1]:.2f}")


if __name__ == "__main__":
    main()
