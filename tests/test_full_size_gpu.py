"""GPU, BASELINE.json's full sizes (65 536 frames per launch): properties that do not need the CPU oracle.
  * round trip: encode -> AWGN well above the waterfall -> decode returns every message, all frames converged
  * codeword symmetry: BP on (codeword c, noise n) and on (all-zero codeword, same noise) runs the same number of
    turns and returns bits that differ by exactly c -- sign flips are exact in floating point, so this holds bit
    for bit (hard 0 = False breaks the symmetry only where a frame returns a channel LLR that is exactly zero)
  * the device tally equals a recount from the returned bits; two runs of the same launch are identical."""
import numpy as np
import pytest

from tests.helpers import CODES

pytestmark = pytest.mark.gpu

B = 65536
CASES = [("ldpc/hip-minsum/jpl.4096.4.5/50/4/5", 2.9), ("ldpc/hip-minsum/jpl.1024.4.5/50/4/5", 3.0),
         ("ldpc/hip-tanh/jpl.1024.4.5/50/4/5", 2.8), ("ldpc/hip-minsum-f16/jpl.4096.4.5/50/4/5", 2.9)]


def _run(ecc, torch, llr, stream):
    dev = llr.device
    bits = torch.empty((B, ecc.code.N), dtype=torch.uint8, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    cv = torch.empty(B, dtype=torch.uint8, device=dev)
    ecc.decoder.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, 50, it.data_ptr(), cv.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    return bits, it, cv


@pytest.mark.parametrize("name,mixed_db", CASES)
def test_round_trip_symmetry_tally_at_full_size(hip, name, mixed_db):
    import torch
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    ecc = hip.ECC(CODES, name, max_batch=B)
    k, N = ecc.message_length, ecc.code.N
    with torch.cuda.stream(st):
        llr = torch.empty((B, N), dtype=torch.float32, device=dev)
        msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
        # ---- round trip, 6 dB: every frame decodes to its message; the decoded word is the codeword c
        ecc.sim.generate(7, 0, B, 6.0, llr.data_ptr(), msg.data_ptr(), st.cuda_stream)
        cw, it, cv = _run(ecc, torch, llr, st)
        assert bool(cv.all()) and bool((cw[:, :k] == msg).all()) and int(it.max()) < 50
        tally = torch.zeros(4, dtype=torch.int64, device=dev)
        ecc.sim.tally(B, cw.data_ptr(), it.data_ptr(), tally.data_ptr(), st.cuda_stream)
        st.synchronize()
        assert tally.tolist() == [B, 0, 0, int(it.sum())]
        # ---- same messages (same seed and frame ids => same codewords), noise of a marginal SNR
        ecc.sim.generate(7, 0, B, mixed_db, llr.data_ptr(), msg.data_ptr(), st.cuda_stream)
        bits, it1, cv1 = _run(ecc, torch, llr, st)
        frac = float(cv1.float().mean())
        assert 0.02 < frac < 0.98, frac                      # early and late finishers and failures, all in one launch
        tally.zero_()
        ecc.sim.tally(B, bits.data_ptr(), it1.data_ptr(), tally.data_ptr(), st.cuda_stream)
        st.synchronize()
        wrong = (bits[:, :k] != msg).sum(dim=1)
        assert tally.tolist() == [B, int((wrong > 0).sum()), int(wrong.sum()), int(it1.sum())]
        # determinism
        bits_b, it_b, cv_b = _run(ecc, torch, llr, st)
        assert bool((bits == bits_b).all()) and bool((it1 == it_b).all()) and bool((cv1 == cv_b).all())
        # ---- symmetry: move the noise onto the all-zero codeword
        llr0 = llr * (1.0 - 2.0 * cw.to(torch.float32))
        bits0, it0, cv0 = _run(ecc, torch, llr0, st)
        assert bool((it0 == it1).all()) and bool((cv0 == cv1).all())
        # (a frame that returns its channel LLRs breaks the symmetry exactly where an LLR is exactly zero --
        #  hard 0 = False for both signs: every punctured position, and about one transmitted sample in 1e8)
        nz = llr != 0
        assert bool((((bits ^ cw) == bits0) | ~nz).all())
        conv = cv1.bool()
        assert bool(((bits[conv] ^ cw[conv]) == bits0[conv]).all())   # converged frames: all N positions
    ecc.close()
