"""GPU: the device ENCODER compared with the oracle encoder directly.

At a noiseless Eb/N0 no sample changes sign, so hard(llr[:, :n_tx]) of ldpc_sim_generate IS the codeword the device encoded;
ldpc_sim_encode_batch returns the codeword bytes without a channel.  Both against oracle.encode_qc (Fast/Encoder.hs:26-63)
/ oracle.encode_dense (Orig.hs:25-26) for thousands of random messages, for the quasi-cyclic rotate-and-xor encoder and the
dense form, synthetic generators of every word size the reference's fast encoder takes (32, 64, 128, 256), and the all-zero
rule for matrices shipped without a generator."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import CODES, load

pytestmark = pytest.mark.gpu


def _env(**kw):
    class _E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            for k, v in kw.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

        def __exit__(self, *a):
            for k, v in self.old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return _E()


def _codewords(hip, sim, B, N, k, n_tx, seed=99, first=12345, ebn0=40.0):
    """-> (messages [B][k], codewords from the noiseless LLRs [B][n_tx], codewords from encode_batch [B][n_tx])"""
    import torch
    dev = torch.device("cuda", 0)
    llr = torch.empty((B, N), dtype=torch.float32, device=dev)
    msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
    cw = torch.full((B, n_tx), 7, dtype=torch.uint8, device=dev)
    msg2 = torch.empty((B, k), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    sim.generate(seed, first, B, ebn0, llr.data_ptr(), msg.data_ptr(), None)
    sim.encode_batch(seed, first, B, cw.data_ptr(), msg2.data_ptr(), None)
    torch.cuda.synchronize()
    l = llr.cpu().numpy()
    assert (l[:, n_tx:] == 0).all() and (np.abs(l[:, :n_tx]) > 1.0).all()      # noiseless: every sample far from zero
    assert np.array_equal(msg.cpu().numpy(), msg2.cpu().numpy())
    return msg.cpu().numpy(), (l[:, :n_tx] > 0).astype(np.uint8), cw.cpu().numpy()


@pytest.mark.parametrize("name,code_name,B", [("jpl.1024.4.5", "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", 4096),
                                             ("jpl.4096.4.5", "ldpc/hip-minsum/jpl.4096.4.5/50/4/5", 4096),
                                             ("jpl.4096.4.5", "ldpc/hip-minsum/jpl.4096.4.5/50", 1000),      # unpunctured: all 1536 parity bits sent
                                             ("moon.7.13", "ldpc/hip-tanh/moon.7.13/20", 4096)])
@pytest.mark.parametrize("encoder", ["qc", "dense"])
def test_device_codewords_equal_the_oracle_encoder(hip, name, code_name, B, encoder):
    c = load(name)
    with _env(LDPC_SIM_ENCODER="dense" if encoder == "dense" else None):
        ecc = hip.ECC(CODES, code_name, max_batch=B)
    want_kind = "qc" if (encoder == "qc" and c.gq is not None) else "dense"
    assert ecc.sim.encoder == want_kind
    k, n_tx, N = ecc.message_length, ecc.codeword_length, ecc.unpunctured_length
    msg, from_llr, from_enc = _codewords(hip, ecc.sim, B, N, k, n_tx)
    assert 0.45 < msg.mean() < 0.55
    if c.gq is not None:
        par = np.stack([oracle.encode_qc(c.gq[0], c.gq[1], m) for m in msg])
    else:
        par = np.stack([oracle.encode_dense(c.G, m) for m in msg])
    want = np.concatenate([msg, par], axis=1)[:, :n_tx]           # Utils.hs:61: msg ++ take (c_length - k) parity
    assert np.array_equal(from_llr, want) and np.array_equal(from_enc, want)
    # the host-side encoder of the record (ldpc_ecc_encode) is the same function
    for i in range(8):
        assert np.array_equal(ecc.encode(msg[i]), want[i])
    ecc.close()


def test_no_generator_means_all_zero_codewords(hip):
    """codes/1920.1280.3.303 ships H only: frames are the all-zero codeword (valid for a linear code on a symmetric channel)."""
    ecc = hip.ECC(CODES, "ldpc/hip-tanh/1920.1280.3.303/50", max_batch=512)
    assert ecc.sim.encoder == "none"
    msg, from_llr, from_enc = _codewords(hip, ecc.sim, 512, 1920, 640, 1920)
    assert not msg.any() and not from_llr.any() and not from_enc.any()
    ecc.close()


@pytest.mark.parametrize("sz,brows,bcols", [(32, 5, 3), (32, 3, 17), (64, 4, 9), (128, 3, 5), (256, 2, 3), (256, 1, 1)])
def test_quasi_cyclic_encoder_on_random_generators(hip, sz, brows, bcols):
    """every word size of Fast/Encoder.hs:28-33, column counts that do and do not fill the last group of 16/W columns,
    batches that do not fill the last wave; against the oracle's QC encoder AND the dense device encoder of the expanded G"""
    rng = np.random.default_rng(sz * 1000 + brows * 10 + bcols)
    W = sz // 32
    words = rng.integers(0, 2 ** 32, size=(brows, bcols, W), dtype=np.uint64).astype(np.uint32)
    words[0, 0] = 0                                               # an empty circulant
    if brows > 1:
        words[1, 0] = 0; words[1, 0, 0] = 1                       # the identity
    gbits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8).reshape(brows, bcols, sz)   # bit b of the integer
    k, p = sz * brows, sz * bcols
    # a parity-check graph is needed only for N: any code with N = k + p columns
    code = hip.Code.from_csr(np.array([0, 2], np.int32), np.array([0, 1], np.int32), k + p)
    B = 333
    n_tx = k + p - 7                                              # a punctured tail that cuts into the last parity word
    qc = hip.Sim(code, k, n_tx, max_batch=B, G_qc=(sz, words))
    assert qc.encoder == "qc"
    msg, from_llr, from_enc = _codewords(hip, qc, B, k + p, k, n_tx)
    par = np.stack([oracle.encode_qc(sz, gbits, m) for m in msg])
    want = np.concatenate([msg, par], axis=1)[:, :n_tx]
    assert np.array_equal(from_llr, want) and np.array_equal(from_enc, want)
    assert np.array_equal(qc.encode_host(msg[0], p), par[0])
    # the expanded generator through the dense encoder: QuasiCyclic.hs:19-25, G[(r*sz + i), (c*sz + j)] = bit (j - i) mod sz
    i = np.arange(sz)
    G = np.zeros((k, p), np.uint8)
    for r in range(brows):
        for c in range(bcols):
            G[r * sz:(r + 1) * sz, c * sz:(c + 1) * sz] = gbits[r, c][(i[None, :] - i[:, None]) % sz]
    dn = hip.Sim(code, k, n_tx, G=G, max_batch=B)
    assert dn.encoder == "dense"
    msg2, from_llr2, from_enc2 = _codewords(hip, dn, B, k + p, k, n_tx)
    assert np.array_equal(msg2, msg) and np.array_equal(from_llr2, want) and np.array_equal(from_enc2, want)
    qc.close(); dn.close(); code.close()


def test_unsupported_word_size_says_so(hip):
    code = hip.Code.from_csr(np.array([0, 2], np.int32), np.array([0, 1], np.int32), 96 + 48)
    with pytest.raises(hip.LdpcError) as e:
        hip.Sim(code, 96, 144, max_batch=4, G_qc=(48, np.ones((2, 1, 2), np.uint32)))
    assert e.value.code == -5 and "unsupported size for fast encoder" in str(e.value)      # Fast/Encoder.hs:33
    code.close()
