"""GPU: the packed-fp16 and the on-chip layered kernels for ANY single-circulant quasi-cyclic H (jit.cc kinds JIT_PK16, JIT_LAYERED,
JIT_LAYERED_PK16: the device bodies of the built-in AR4JA instances, specialised at context creation like the f32 flooding kernel).
Bars, as for the built-in instances: packed fp16 = the bit-exact emulation's whole trajectory (oracle/emulate_f16.py); on-chip layered
f32 = the HBM layered kernel bit for bit, and the Double oracle's hard bits / flags."""
import numpy as np
import pytest

from oracle import emulate_f16 as em
from oracle import oracle
from tests.helpers import synthetic

pytestmark = pytest.mark.gpu

# circulant sizes 27 / 32 (two frames per wave; with two frames per lane: four per wave), 64, 96 (not a power of two), 128, 256, 360;
# one, two and more wave groups
SHAPES = ["small-2x4-sz32", "wifi-12x24-sz27", "ira-12x24-sz64", "wimax-12x24-sz96", "regular36-sz128", "irregular-20x30-sz64",
          "wide-4x40-sz256", "jpl4096-permuted", "dvbs2short-20x45-sz360"]


def _frames(c, F, seed):
    dbs = (2.0, 3.0, 4.5) if c.N > 1000 else (3.0, 5.0, 7.0)
    llr = np.concatenate([c.frames(F // 3 + 1, db, seed + i)[1] for i, db in enumerate(dbs)])[:F].astype(np.float32)
    llr = llr[np.random.default_rng(seed).permutation(F)]
    llr[0, :8] = [7e4, -7e4, 1e-9, -1e-9, 0.0, 65504.0, 3.0e-8, -6.0e-8]
    return llr


@pytest.mark.parametrize("name", SHAPES)
def test_packed_fp16_flooding(hip, name):
    c = synthetic(name)
    F = 7 if c.N > 4000 else 13                                                     # odd: the last lane has one frame only
    llr = _frames(c, F, 8100)
    dec = hip.Decoder(c.hip_code(hip), "min", "f16pk", F)
    assert dec.path == "fused" and dec.kernel_name.startswith("ldpc_jit_pk16_minsum_sz%d_" % c.sz), dec.kernel_name
    bits, its, conv, trace = dec.decode_trace(llr, 40)
    eb, ei, ec, et = em.decode_minsum_pk16(c.graph, llr, 40)
    assert np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec) and np.array_equal(bits, eb), name
    for n, lam in enumerate(et):
        live = n <= ei
        assert np.isfinite(lam[live]).all()                                          # the saturation rules: frame 0 diverges, never to inf / NaN
        assert np.array_equal(trace[live, n, :], lam[live].astype(np.float64)), (name, n)
    b2, i2, c2 = dec.decode_batch(em.r16(llr).astype(np.float16), 40)
    assert np.array_equal(b2, bits) and np.array_equal(i2, its) and np.array_equal(c2, conv)
    # no packed-result limit in the run-time instances: more turns than the built-in ones take
    b3, i3, c3 = dec.decode_batch(llr[:3], 600)
    e3 = em.decode_minsum_pk16(c.graph, llr[:3], 600)
    assert np.array_equal(b3, e3[0]) and np.array_equal(i3, e3[1]) and np.array_equal(c3.astype(bool), e3[2])
    print(f"{name}: {dec.kernel_name} {int(conv.sum())}/{F} converged, turns {sorted(set(its.tolist()))}")
    dec.close()


@pytest.mark.parametrize("name", SHAPES)
def test_packed_fp16_layered(hip, name):
    c = synthetic(name)
    F = 7 if c.N > 4000 else 13
    llr = _frames(c, F, 8200)
    dec = hip.Decoder(c.hip_code(hip), "min", "f16pk", F, schedule="layered")
    assert dec.path == "fused" and dec.kernel_name.startswith("ldpc_jit_layered_pk16_minsum_sz%d_" % c.sz), dec.kernel_name
    bits, its, conv, trace = dec.decode_trace(llr, 40)
    eb, ei, ec, et = em.decode_minsum_pk16_layered(c.graph, llr, 40)
    assert np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec) and np.array_equal(bits, eb), name
    for n, lam in enumerate(et):
        live = n <= np.where(ec, ei, 40)
        assert np.isfinite(lam[live]).all()
        assert np.array_equal(trace[live, n, :], lam[live].astype(np.float64)), (name, n)
    b2, i2, c2 = dec.decode_batch(llr, 40)
    assert np.array_equal(b2, bits) and np.array_equal(i2, its) and np.array_equal(c2, conv)
    print(f"{name}: {dec.kernel_name} {int(conv.sum())}/{F} converged, sweeps {sorted(set(its.tolist()))}")
    dec.close()


@pytest.mark.parametrize("name", SHAPES)
def test_layered_f32_on_chip(hip, name):
    c = synthetic(name)
    F = 21 if c.N > 4000 else 61
    llr = _frames(c, F, 8300)
    llr[0, :8] = c.frames(1, 3.0, 1)[1][0, :8]                                       # (no fp16 edge values here)
    code = c.hip_code(hip)
    on = hip.Decoder(code, "min", "f32", F, schedule="layered")                       # AUTO
    hbm = hip.Decoder(code, "min", "f32", F, schedule="layered", path="flood")
    assert on.path == "fused" and on.kernel_name.startswith("ldpc_jit_layered_minsum_sz%d_" % c.sz), on.kernel_name
    assert "layered_qc_kernel" in hbm.kernel_name
    a = on.decode_batch(llr.astype(np.float64), 40, want_lam=True)
    b = hbm.decode_batch(llr.astype(np.float64), 40, want_lam=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)), name                     # bits, sweeps, flags, final LLRs
    ta, tb = on.decode_trace(llr[:5], 40), hbm.decode_trace(llr[:5], 40)
    for f in range(5):
        n = ta[1][f]
        assert np.array_equal(ta[3][f, : n + 1], tb[3][f, : n + 1]) and ta[1][f] == tb[1][f]
    lp = np.arange(0, c.M + 1, c.sz)
    ref = [oracle.decode_layered(c.graph, lp, "min", 40, l.astype(np.float64)) for l in llr]
    ob = np.stack([o["bits"] for o in ref]); oi = np.array([o["iters"] for o in ref]); oc = np.array([o["converged"] for o in ref])
    same = a[2].astype(bool) == oc
    assert same.mean() >= 0.9 and np.array_equal(a[0][same], ob[same])
    assert (a[1] == oi)[same].mean() >= 0.85
    b600 = on.decode_batch(llr[:4], 600)                                             # beyond the built-in instances' 511 sweeps
    h600 = hbm.decode_batch(llr[:4], 600)
    assert all(np.array_equal(x, y) for x, y in zip(b600, h600))
    print(f"{name}: {on.kernel_name} {int(a[2].sum())}/{F} converged, sweeps identical with the Double oracle {100 * (a[1] == oi)[same].mean():.0f}%")
    on.close(); hbm.close()


def test_what_has_no_run_time_instance_says_why(hip):
    c = synthetic("regular36-sz128")
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(c.hip_code(hip), "tanh", "f16pk", 4)
    assert e.value.code == -5 and "min-sum" in str(e.value)
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(c.hip_code(hip), "tanh", "f32", 4, schedule="layered", path="fused")
    assert e.value.code == -5
    assert hip.Decoder(c.hip_code(hip), "tanh", "f32", 4, schedule="layered").path == "flood"


def test_layered_auto_falls_back_to_hbm_when_the_compiler_fails(hip, monkeypatch, capfd):
    """LDPC_SCHED_LAYERED + LDPC_PATH_AUTO on a code whose on-chip layered kernel is compiled at run time: when that compilation fails
    the context keeps its state in HBM (the r02 behaviour) instead of failing -- same decoder, other kernel; LDPC_PATH_FUSED still fails"""
    c = synthetic("ira-12x24-sz64")
    _, llr = c.frames(12, 2.5, seed=8800)
    llr = llr.astype(np.float32)
    ok = hip.Decoder(c.hip_code(hip), "min", "f32", 12, schedule="layered")
    assert ok.path == "fused"
    ref = ok.decode_batch(llr, 30)
    monkeypatch.setenv("LDPC_JIT_EXTRA_OPTS", "--no-such-option-for-the-test")      # (experimental builds are never cached: the compiler runs and fails)
    dec = hip.Decoder(c.hip_code(hip), "min", "f32", 12, schedule="layered")
    assert dec.path == "flood" and "layered_qc_kernel" in dec.kernel_name, (dec.path, dec.kernel_name)
    assert "on-chip layered kernel unavailable" in capfd.readouterr().err
    assert all(np.array_equal(x, y) for x, y in zip(dec.decode_batch(llr, 30), ref))
    with pytest.raises(hip.LdpcError):
        hip.Decoder(c.hip_code(hip), "min", "f32", 12, schedule="layered", path="fused")
    dec.close(); ok.close()
