"""GPU: the LDPC_F16 contexts ("fp16 storage in HBM, f32 arithmetic", BASELINE.json configs[3]) and the fp16-LLR
entry points, through the C ABI.  The reference has no fp16 path, so the checker is the build's own emulation
(oracle/emulate_f16.py): flood F16 min-sum is reproduced bit for bit, fused F16 is the f32 decoder fed
fp16-rounded LLRs."""
import numpy as np
import pytest

from oracle import emulate_f16 as em
from tests.helpers import load

pytestmark = pytest.mark.gpu

CASES = [("moon.7.13", 20, (1.0, 3.0, 5.0)), ("1920.1280.3.303", 50, (1.0, 2.0, 3.0)), ("jpl.1024.4.5", 50, (2.0, 3.0, 4.0))]


def _frames(c, per_db, dbs, seed):
    return np.concatenate([c.frames(per_db, db, seed + i)[1] for i, db in enumerate(dbs)]).astype(np.float32)


@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_flood_f16_minsum_trajectory_is_bit_exact(hip, name, iters, dbs):
    c = load(name)
    llr = _frames(c, 4, dbs, 3100)
    llr[0, :7] = [7e4, -7e4, 1e-9, -1e-9, 0.0, 65504.0, 3.0e-8]   # saturation, underflow to 0 (hard 0 = False), subnormal
    dec = hip.Decoder(c.hip_code(hip), "min", "f16", len(llr), path="flood")
    bits, its, conv, trace = dec.decode_trace(llr, iters)
    eb, ei, ec, et = em.decode_minsum_f16_flood(c.graph, llr, iters)
    assert np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec) and np.array_equal(bits, eb)
    assert len(set(its.tolist())) > 1
    for n, lam in enumerate(et):
        assert np.array_equal(trace[:, n, :], lam.astype(np.float64)), n
    # the throughput entry points give the same answer, from an f32 buffer and from an fp16 buffer
    # (rounded here with the library's saturating rule: numpy's own float16 cast turns 7e4 into inf)
    for x in (llr, em.r16(llr).astype(np.float16)):
        b2, i2, c2 = dec.decode_batch(x, iters)
        assert np.array_equal(b2, bits) and np.array_equal(i2, its) and np.array_equal(c2, conv)


@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_flood_f16_minsum_teacher_forced_step(hip, name, iters, dbs):
    c = load(name)
    rng = np.random.default_rng(77)
    F = 6
    orig = em.r16(rng.normal(0, 6, (F, c.N)))
    lam = em.r16(orig + rng.normal(0, 3, (F, c.N)))
    ne = em.r16(rng.normal(0, 2, (F, c.E)))
    dec = hip.Decoder(c.hip_code(hip), "min", "f16", F, path="flood")
    ne2, lam2, syn = dec.debug_step(orig, lam, ne)
    e_ne, e_lam, e_syn = em.step_minsum_f16_flood(c.graph, orig, lam, ne)
    assert np.array_equal(ne2, e_ne.astype(np.float64)) and np.array_equal(lam2, e_lam.astype(np.float64))
    assert np.array_equal(syn, e_syn)


@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_flood_f16_tanh_teacher_forced_step(hip, name, iters, dbs):
    """tanh rule: the kernel's f32 arithmetic (hardware exp/rcp/log) is not reproducible bit for bit, so each stored
    message must be an fp16 neighbour of the exactly computed value: within one fp16 ulp everywhere, identical
    almost always.  The f32 rule (ldpc_math.h, hyperbolic recurrence) is accurate to ~6e-7 RELATIVE for |ne'| >= 1
    but to ~8e-7 ABSOLUTE below that (A - S cancels when the tanh product is small), so the tiny messages of
    weight-18 rows may land a few fp16 steps away: hence the absolute term, well inside the 1e-5 LLR bar."""
    c = load(name)
    rng = np.random.default_rng(78)
    F = 6
    orig = em.r16(rng.normal(0, 6, (F, c.N)))
    lam = em.r16(orig + rng.normal(0, 3, (F, c.N)))
    ne = em.r16(rng.normal(0, 2, (F, c.E)))
    dec = hip.Decoder(c.hip_code(hip), "tanh", "f16", F, path="flood")
    ne2, lam2, _ = dec.debug_step(orig, lam, ne)
    g = c.graph
    exact = np.empty((F, c.E))
    for m in range(g.M):
        e0, e1 = g.row_ptr[m], g.row_ptr[m + 1]
        t = (lam[:, g.col_idx[e0:e1]] - ne[:, e0:e1]).astype(np.float32).astype(np.float64)
        th = np.tanh(-t / 2)
        for k in range(e1 - e0):
            p = np.prod(np.delete(th, k, axis=1), axis=1)
            with np.errstate(divide="ignore"):
                y = 0.5 * np.log((1 + p) / (1 - p))
            y = np.where(np.isinf(y), np.sign(p) * 18.714973875118524, y)   # Utils.hs:113-117
            exact[:, e0 + k] = -2 * y
    want = em.r16(exact.astype(np.float32)).astype(np.float64)
    ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10 + 1e-6   # + the f32 rule's own absolute error bound
    assert (np.abs(ne2 - want) <= ulp).all()
    big = np.abs(want) >= 2.0 ** -6
    assert (ne2 == want)[big].mean() > 0.97 and (ne2 == want).mean() > 0.85
    # lam' = r16(orig + sum of the kernel's OWN stored messages, descending rows, f32): exact given ne2
    acc = orig.copy()
    for m in reversed(range(g.M)):
        e0, e1 = g.row_ptr[m], g.row_ptr[m + 1]
        cols = g.col_idx[e0:e1]
        acc[:, cols] = (ne2[:, e0:e1].astype(np.float32) + acc[:, cols]).astype(np.float32)
    assert np.array_equal(lam2, em.r16(acc).astype(np.float64))


@pytest.mark.parametrize("variant", ["min", "tanh"])
@pytest.mark.parametrize("name,qc", [("jpl.1024.4.5", True), ("jpl.4096.4.5", True), ("jpl.1024.4.5", False), ("1920.1280.3.303", False)])
def test_fused_f16_is_f32_on_rounded_llrs(hip, monkeypatch, name, qc, variant):
    """A fused F16 context keeps nothing but the channel LLRs in HBM: it must equal the f32 kernel fed r16(llr),
    whichever entry point (f32 or fp16 buffer) delivered them -- for the split, two-wave and generic kernels."""
    c = load(name)
    llr = _frames(c, 6, (2.0, 3.0, 4.0), 3300)
    llr[1, :6] = [7e4, -7e4, 1e-9, -1e-9, 0.0, 3.0e-8]
    code = c.hip_code(hip, prefer_qc=qc)
    kernels = ("split", "msg") if qc else ("csr",)
    for k in kernels:
        monkeypatch.setenv("LDPC_FUSED_KERNEL", k)
        d16 = hip.Decoder(code, variant, "f16", len(llr), path="fused")
        d32 = hip.Decoder(code, variant, "f32", len(llr), path="fused")
        want = d32.decode_batch(em.r16(llr).astype(np.float64), 50, want_lam=True)
        got = d16.decode_batch(llr.astype(np.float64), 50, want_lam=True)
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), k
        got32 = d16.decode_batch(llr, 50)
        got16 = d16.decode_batch(em.r16(llr).astype(np.float16), 50)
        via32 = d32.decode_batch(em.r16(llr).astype(np.float16), 50)   # fp16 buffer into an f32 context
        for g in (got32, got16, via32):
            assert all(np.array_equal(x, y) for x, y in zip(g, want[:3])), k
    assert hip.Decoder(code, variant, "f16", 8).path == "fused"   # and AUTO picks the on-chip kernel


def test_frame_source_fp16_and_device_entry_point(hip):
    """ldpc_sim_generate_f16 = r16 of ldpc_sim_generate; ldpc_decode_batch_dev_f16 on it = the f32 entry point
    on the rounded values (F16 and F32 contexts, flood and fused)."""
    import torch
    ecc = hip.ECC(hip_codes_dir(), "ldpc/hip-minsum-f16/jpl.1024.4.5/50/4/5", max_batch=256)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sp = st.cuda_stream
    B, N = 256, ecc.code.N
    with torch.cuda.stream(st):
        x32 = torch.empty((B, N), dtype=torch.float32, device=dev)
        x16 = torch.empty((B, N), dtype=torch.float16, device=dev)
        ecc.sim.generate(0x5EED, 100, B, 3.0, x32.data_ptr(), None, sp)
        ecc.sim.generate(0x5EED, 100, B, 3.0, x16.data_ptr(), None, sp, llr_f16=True)
        st.synchronize()
        a32 = x32.cpu().numpy()
        assert np.array_equal(x16.cpu().numpy().astype(np.float32), em.r16(a32))
        outs = []
        for dtype, path in (("f16", "fused"), ("f32", "fused"), ("f16", "flood")):
            dec = hip.Decoder(ecc.code, "min", dtype, B, path=path)
            bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
            it = torch.empty(B, dtype=torch.int32, device=dev)
            dec.decode_batch_dev(x16.data_ptr(), bits.data_ptr(), B, 50, it.data_ptr(), None, sp, llr_f16=True)
            st.synchronize()
            ref = dec.decode_batch(em.r16(a32), 50)
            assert np.array_equal(bits.cpu().numpy(), ref[0]) and np.array_equal(it.cpu().numpy(), ref[1]), (dtype, path)
            outs.append(ref)
        assert all(np.array_equal(x, y) for x, y in zip(outs[0], outs[1]))   # fused F16 == fused F32 on fp16 LLRs
    ecc.close()


def hip_codes_dir():
    import os
    from tests.helpers import CODES
    return CODES
