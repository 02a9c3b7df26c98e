"""GPU: the one number the reference holds for this path, /root/reference/NOTES.txt:1-3 --

    ./dist/build/ecc-ldpc/ecc-ldpc ldpc/model/jpl.1K/200 0 -m256
       39.70 ldpc/model/jpl.1K/200  0.00      256    29891  1.14e-1 +1% -1% [95%].

i.e. at Eb/N0 = 0 dB a pass-through decoder ("basically bpsk") on jpl.1K WITHOUT a puncturing rate (code name has no
/x/y, so rate = k / cols H = 1024/1408, Utils.hs:46-48) made 29 891 bit errors in 256 messages x 1024 bits:
BER = 0.11402.  That pins the channel convention of the absent tester (ecc-manifold): BER of uncoded BPSK is
Q(sqrt(2 R Eb/N0)) = Q(sqrt(2 * 1024/1408)) = 0.11390 only if sigma^2 = 1 / (2 R Eb/N0) with R = k / n_tx --
which is what sim.hip and oracle/channel.py assume.  (With sigma^2 = 1/(2 Eb/N0), rate not
folded in, the BER would be Q(sqrt 2) = 0.0786.)

Here: the same code name shape (no rate) through the ECC record, the device frame source, a decode with 0 turns
(= hard decisions of the channel LLRs, the reference's `Nothing`/model behaviour) and the device tally, over 2^24
message bits."""
import math

import numpy as np
import pytest

from tests.helpers import CODES

pytestmark = pytest.mark.gpu

REF_ERRORS, REF_BITS = 29891, 256 * 1024    # NOTES.txt:3


def q(x):
    return 0.5 * math.erfc(x / math.sqrt(2.0))


def test_bpsk_ber_at_0db_matches_notes_txt(hip):
    import torch
    F = 16384                                              # 2^24 message bits
    ecc = hip.ECC(CODES, "ldpc/hip-minsum/jpl.1024.4.5/0", max_batch=F)      # max-rounds 0, no rate -> 8/11, all 1408 sent
    assert ecc.name.endswith("/0/8/11") and ecc.codeword_length == 1408 and ecc.message_length == 1024
    dev = torch.device("cuda", 0)
    N, k = ecc.unpunctured_length, ecc.message_length
    llr = torch.empty((F, N), dtype=torch.float32, device=dev)
    bits = torch.empty((F, N), dtype=torch.uint8, device=dev)
    iters = torch.empty((F,), dtype=torch.int32, device=dev)
    tally = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    torch.cuda.synchronize()
    ecc.sim.generate(0x5EEDC0DE, 0, F, 0.0, llr.data_ptr(), None, sp)
    ecc.decoder.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), F, 0, iters.data_ptr(), None, sp)
    ecc.sim.tally(F, bits.data_ptr(), iters.data_ptr(), tally.data_ptr(), sp)
    torch.cuda.synchronize()
    frames, _, bit_errors, sum_iters = tally.tolist()
    assert frames == F and sum_iters == 0
    # nothing punctured (a punctured position is LLR 0.0 in every frame; an fp32 sample may hit 0.0 once in ~10^7)
    assert llr.shape[1] == 1408 and int((llr == 0).sum(dim=0).max().item()) <= 2
    n = F * k
    ber = bit_errors / n
    theory = q(math.sqrt(2.0 * 1024 / 1408))
    ref = REF_ERRORS / REF_BITS
    s_ours = math.sqrt(theory * (1 - theory) / n)
    s_ref = math.sqrt(theory * (1 - theory) / REF_BITS)
    print(f"BER at 0 dB over {n} bits: {ber:.5f}; Q(sqrt(2R)) = {theory:.5f}; NOTES.txt:3 = {ref:.5f} "
          f"(sigma ours {s_ours:.1e}, reference sample {s_ref:.1e})")
    assert abs(ber - theory) < 3 * s_ours                      # our frame source against the closed form
    assert abs(ber - ref) < 3 * math.hypot(s_ours, s_ref)      # and against the reference's published count
    assert abs(ref - theory) < 3 * s_ref                       # (the reference's count itself fits the convention ...
    assert abs(ref - q(math.sqrt(2.0))) > 20 * s_ref           #  ... and rules out the rate-free sigma)
    ecc.close()
