"""GPU: LDPC_F16PK -- min-sum with ARITHMETIC in IEEE binary16, two frames per lane in packed instructions (csrc/fused_pk16_body.h;
BASELINE.json configs[3]).  The reference has no fp16 decoder, so the checker is the build's own bit-exact emulation
(oracle/emulate_f16.py decode_minsum_pk16): whole free-running trajectories -- every LLR of every turn, hard bits, iteration
counts, flags -- for both built-in instances, odd batches (a lane whose second frame does not exist), the two frames of a lane
stopping at different turns, saturation / underflow / zero LLRs, and fp16 and f32 input buffers."""
import numpy as np
import pytest

from oracle import emulate_f16 as em
from oracle import oracle
from tests.helpers import CODES, load

pytestmark = pytest.mark.gpu


def _frames(c, per_db, dbs, seed):
    return np.concatenate([c.frames(per_db, db, seed + i)[1] for i, db in enumerate(dbs)]).astype(np.float32)


@pytest.mark.parametrize("name,per_db,dbs", [("jpl.1024.4.5", 7, (2.0, 3.0, 4.0)), ("jpl.4096.4.5", 3, (2.5, 3.3, 4.0))])
def test_trajectory_is_bit_exact_with_the_emulation(hip, name, per_db, dbs):
    c = load(name)
    llr = _frames(c, per_db, dbs, 4100)                      # 21 / 9 frames: odd -> the last lane pair has one frame only
    rng = np.random.default_rng(5)
    llr = llr[rng.permutation(len(llr))]                     # partners from different Eb/N0: they stop at different turns
    llr[0, :8] = [7e4, -7e4, 1e-9, -1e-9, 0.0, 65504.0, 3.0e-8, -6.0e-8]   # saturation, underflow to zero, zero, subnormal
    dec = hip.Decoder(c.hip_code(hip), "min", "f16pk", len(llr))
    assert dec.path == "fused" and "fused_pk16_kernel" in dec.kernel_name
    bits, its, conv, trace = dec.decode_trace(llr, 50)
    eb, ei, ec, et = em.decode_minsum_pk16(c.graph, llr, 50)
    assert np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec) and np.array_equal(bits, eb)
    assert len(set(its.tolist())) > 3 and 0 < conv.sum() < len(llr)
    pairs = its[: len(its) // 2 * 2].reshape(-1, 2)
    assert (pairs[:, 0] != pairs[:, 1]).any()                # two frames of one lane finishing at different turns
    for n, lam in enumerate(et):
        live = n <= ei                                        # a frame's rows up to the turn it stopped at
        assert np.array_equal(trace[live, n, :], lam[live].astype(np.float64)), n
    assert "fused_pk16_kernel" in dec.kernel_name
    # the throughput entry points: f32 buffer, fp16 buffer (rounded with the library's saturating rule), device pointers
    for x in (llr, em.r16(llr).astype(np.float16)):
        b2, i2, c2 = dec.decode_batch(x, 50)
        assert np.array_equal(b2, bits) and np.array_equal(i2, its) and np.array_equal(c2, conv)
    # fewer turns than a frame needs: the channel's hard decisions come back (Min.hs:76), on the fp16-rounded LLRs
    b3, i3, c3 = dec.decode_batch(llr, 3)
    e3 = em.decode_minsum_pk16(c.graph, llr, 3)
    assert np.array_equal(b3, e3[0]) and np.array_equal(i3, e3[1]) and np.array_equal(c3.astype(bool), e3[2])
    late = ~e3[2]
    assert late.any() and np.array_equal(b3[late], (em.r16(llr[late]) > 0).astype(np.uint8))
    dec.close()


def test_zero_llrs_and_single_frames(hip):
    """hard 0 = False (all-zero LLRs: zero turns, all False, `converged`); one frame alone in its lane pair; batch of one"""
    c = load("jpl.1024.4.5")
    dec = hip.Decoder(c.hip_code(hip), "min", "f16pk", 8)
    z = np.zeros((3, c.N), np.float32)
    z[1] = -3.0                                              # noiseless all-zero codeword
    bits, its, conv = dec.decode_batch(z, 50)
    assert its.tolist() == [0, 0, 0] and conv.all() and not bits.any()
    _, llr = c.frames(5, 3.5, seed=11)
    want = em.decode_minsum_pk16(c.graph, llr.astype(np.float32), 50)
    for f in range(5):
        b, i, cv = dec.decode_batch(llr[f:f + 1].astype(np.float32), 50)
        assert np.array_equal(b[0], want[0][f]) and i[0] == want[1][f] and bool(cv[0]) == want[2][f]
    dec.close()


def test_agrees_with_the_f32_decoder_in_the_waterfall(hip):
    """not a parity claim (different arithmetic) -- a sanity bound: at 3.4 dB on jpl.1024 the fp16 decoder and the f32 decoder
    decode the same frames, and the few they disagree on differ by the flag, not by a wrong codeword"""
    c = load("jpl.1024.4.5")
    _, llr = c.frames(600, 3.4, seed=21)
    llr = llr.astype(np.float32)
    a = hip.Decoder(c.hip_code(hip), "min", "f16pk", len(llr)).decode_batch(llr, 50)
    b = hip.Decoder(c.hip_code(hip), "min", "f32", len(llr)).decode_batch(llr, 50)
    both = a[2].astype(bool) & b[2].astype(bool)
    assert both.mean() > 0.9 and np.array_equal(a[0][both], b[0][both])            # converged in both: the same codeword
    assert abs(int(a[2].sum()) - int(b[2].sum())) <= 0.02 * len(llr)
    assert abs(a[1][both].astype(float).mean() - b[1][both].astype(float).mean()) < 0.5


def test_what_is_not_provided_says_so(hip):
    c = load("jpl.1024.4.5")
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(c.hip_code(hip), "tanh", "f16pk", 4)
    assert e.value.code == -5
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(c.hip_code(hip), "min", "f16pk", 4, path="flood")
    assert e.value.code == -5
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(c.hip_code(hip, prefer_qc=False), "min", "f16pk", 4)            # a CSR graph: no built-in instance
    assert e.value.code == -5 and "built-in" in str(e.value)
    with pytest.raises(hip.LdpcError):
        hip.Decoder(load("1920.1280.3.303").hip_code(hip), "min", "f16pk", 4)
    dec = hip.Decoder(c.hip_code(hip), "min", "f16pk", 4)
    with pytest.raises(hip.LdpcError):
        dec.debug_step(np.zeros((1, c.N)), np.zeros((1, c.N)), np.zeros((1, c.E)))


def test_record_by_name_and_full_size_invariants(hip):
    """ldpc/hip-minsum-f16pk/jpl.4096.4.5/50/4/5 at the benchmark's batch size: frames from the device source, every frame
    reported converged is a codeword (syndrome by the oracle's H), every failed one carries the channel's hard decisions."""
    import torch
    ecc = hip.ECC(CODES, "ldpc/hip-minsum-f16pk/jpl.4096.4.5/50/4/5", max_batch=4096)
    assert ecc.decoder.path == "fused" and ecc.name == "ldpc/hip-minsum-f16pk/jpl.4096.4.5/50/4/5"
    c = load("jpl.4096.4.5")
    dev = torch.device("cuda", 0)
    B, N, k = 4095, c.N, 4096                                 # odd batch
    llr = torch.empty((B, N), dtype=torch.float16, device=dev)
    msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
    bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
    its = torch.empty((B,), dtype=torch.int32, device=dev)
    conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ecc.sim.generate(7, 0, B, 2.8, llr.data_ptr(), msg.data_ptr(), None, llr_f16=True)
    torch.cuda.synchronize()       # (a NULL stream means the default stream for the frame source but the context's own for decode)
    ecc.decoder.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, 50, its.data_ptr(), conv.data_ptr(), None, llr_f16=True)
    torch.cuda.synchronize()
    b, cv, it = bits.cpu().numpy(), conv.cpu().numpy().astype(bool), its.cpu().numpy()
    assert 0.5 < cv.mean() < 1.0 and len(set(it.tolist())) > 10
    H = torch.tensor(c.H, dtype=torch.float32, device=dev)
    syn = (bits.to(torch.float32) @ H.T) % 2
    assert not syn[torch.tensor(cv, device=dev)].any().item()                       # converged frames are codewords
    hard_in = (llr > 0).to(torch.uint8).cpu().numpy()
    assert np.array_equal(b[~cv], hard_in[~cv])                                     # Min.hs:76
    err = (b[cv][:, :k] != msg.cpu().numpy()[cv]).sum()
    assert err == 0                                                                 # and the transmitted ones, at this Eb/N0
    ecc.close()


@pytest.mark.parametrize("name,per_db,dbs", [("jpl.1024.4.5", 7, (2.2, 3.0, 4.0)), ("jpl.4096.4.5", 3, (2.6, 3.2, 4.0))])
def test_layered_trajectory_is_bit_exact_with_the_emulation(hip, name, per_db, dbs):
    """LDPC_F16PK + LDPC_SCHED_LAYERED: the on-chip layered kernel in packed fp16 (csrc/fused_layered_body.h, laypk) against
    oracle/emulate_f16.py decode_minsum_pk16_layered -- every LLR after every sweep, bits, sweep counts, flags; odd batch, partners
    that stop at different sweeps, saturation / zero LLRs"""
    c = load(name)
    llr = _frames(c, per_db, dbs, 4700)
    llr = llr[np.random.default_rng(6).permutation(len(llr))]
    llr[1, :8] = [7e4, -7e4, 1e-9, -1e-9, 0.0, 65504.0, 3.0e-8, -6.0e-8]
    dec = hip.Decoder(c.hip_code(hip), "min", "f16pk", len(llr), schedule="layered")
    assert dec.path == "fused" and dec.schedule == "layered"
    bits, its, conv, trace = dec.decode_trace(llr, 40)
    assert "fused_layered_pk16_kernel" in dec.kernel_name
    eb, ei, ec, et = em.decode_minsum_pk16_layered(c.graph, llr, 40)
    assert np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec) and np.array_equal(bits, eb)
    assert len(set(its.tolist())) > 3 and 0 < conv.sum() < len(llr)
    for n, lam in enumerate(et):
        live = n <= np.where(ec, ei, 40)
        assert np.array_equal(trace[live, n, :], lam[live].astype(np.float64)), n
    for x in (llr, em.r16(llr).astype(np.float16)):
        b2, i2, c2 = dec.decode_batch(x, 40)
        assert np.array_equal(b2, bits) and np.array_equal(i2, its) and np.array_equal(c2, conv)
    z = np.zeros((2, c.N), np.float32)
    assert dec.decode_batch(z, 40)[1].tolist() == [0, 0]                                   # all-zero LLRs: syndrome zero before the first sweep
    b0, i0, c0 = dec.decode_batch(llr[:5], 0)                                              # no sweeps allowed
    assert np.array_equal(b0, (em.r16(llr[:5]) > 0).astype(np.uint8)) and not c0.any()
    # needs about half the sweeps of the flooding fp16 kernel's turns, finds the same codewords
    fl = hip.Decoder(c.hip_code(hip), "min", "f16pk", len(llr)).decode_batch(llr, 40)
    both = conv.astype(bool) & fl[2].astype(bool)
    assert both.sum() >= 3 and np.array_equal(bits[both], fl[0][both]) and its[both].mean() < 0.75 * fl[1][both].mean()
    dec.close()
