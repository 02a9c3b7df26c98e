"""GPU: the row-layered schedule ON-CHIP (csrc/fused_layered_body.h; LDPC_SCHED_LAYERED + LDPC_PATH_FUSED, what LDPC_PATH_AUTO
gives the shipped AR4JA matrices for min-sum f32).  An extension: its specification is oracle_decode_layered.  Bars: identical
to the HBM layered kernel (layered_qc.hip) bit for bit -- bits, sweeps, flags, the LLRs a frame stops with, whole traces -- and
against the Double oracle hard bits / flags exactly, sweep counts within the f32-vs-f64 bar."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import CODES, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,F,dbs", [("jpl.1024.4.5", 61, (2.4, 3.0, 3.8)), ("jpl.4096.4.5", 33, (2.6, 3.0, 3.6))])
def test_on_chip_layered_equals_the_hbm_kernel_and_the_oracle(hip, name, F, dbs):
    c = load(name)
    llr = np.concatenate([c.frames(F // 3 + 1, db, 6100 + i)[1] for i, db in enumerate(dbs)])[:F].astype(np.float32)   # odd batch
    llr = llr[np.random.default_rng(3).permutation(F)]
    code = c.hip_code(hip)
    on = hip.Decoder(code, "min", "f32", F, schedule="layered")                       # AUTO
    hbm = hip.Decoder(code, "min", "f32", F, schedule="layered", path="flood")
    assert on.path == "fused" and on.schedule == "layered" and "layered_qc_kernel" in hbm.kernel_name
    a = on.decode_batch(llr.astype(np.float64), 40, want_lam=True)
    b = hbm.decode_batch(llr.astype(np.float64), 40, want_lam=True)
    assert "fused_layered_kernel" in on.kernel_name
    assert all(np.array_equal(x, y) for x, y in zip(a, b))                             # bits, sweeps, flags, final LLRs
    assert 0 < a[2].sum() < F and len(set(a[1].tolist())) > 4
    ta, tb = on.decode_trace(llr[:7], 40), hbm.decode_trace(llr[:7], 40)
    for f in range(7):
        n = ta[1][f]
        assert np.array_equal(ta[3][f, : n + 1], tb[3][f, : n + 1]) and ta[1][f] == tb[1][f]
    lp = np.arange(0, c.M + 1, c.sz)
    ref = [oracle.decode_layered(c.graph, lp, "min", 40, l.astype(np.float64)) for l in llr]
    ob = np.stack([o["bits"] for o in ref]); oi = np.array([o["iters"] for o in ref]); oc = np.array([o["converged"] for o in ref])
    # f32 against the DOUBLE specification -- measured, not assumed (tools/layered_f32_vs_f64.py, profiles/r04_layered_f32_vs_f64.txt:
    # 20 400 min-sum frame decodes across the waterfall of jpl.1024 and jpl.4096, on-chip and HBM kernels alike): the converged flag
    # differs in 1.4 % of the frames (3.6 % at the waterfall's edge, as often one way as the other), the sweep count in 2.7 % (9.8 % at the
    # edge, by up to 16 sweeps: the serial schedule amplifies a rounding difference, min-sum's arg-min ties most of all -- the tanh rule:
    # 2e-4), and the hard bits of EVERY frame whose flags agree are identical.  Bars for this sample of F frames at two Eb/N0: flags >= 90 %,
    # sweeps >= 85 %, bits exact.
    same = a[2].astype(bool) == oc
    assert same.mean() >= 0.90 and np.array_equal(a[0][same], ob[same]), same.mean()
    assert (a[1] == oi)[same].mean() >= 0.85, (a[1] == oi)[same].mean()
    print(f"{name} layered on-chip f32 vs Double: flags agree {same.mean():.3f}, sweeps equal {(a[1] == oi)[same].mean():.3f} over {F} frames")
    # the throughput entry points: f32 and fp16 buffers
    b32 = on.decode_batch(llr, 40)
    assert all(np.array_equal(x, y) for x, y in zip(b32, a[:3]))
    on.close(); hbm.close()


def test_edges_and_what_is_not_provided(hip):
    c = load("jpl.1024.4.5")
    code = c.hip_code(hip)
    dec = hip.Decoder(code, "min", "f32", 8, schedule="layered", path="fused")
    z = np.zeros((3, c.N), np.float32)
    z[1] = -2.5                                # noiseless all-zero codeword
    bits, its, conv = dec.decode_batch(z, 30)
    assert its.tolist() == [0, 0, 0] and conv.all() and not bits.any()                 # hard 0 = False; syndrome before the first sweep
    _, llr = c.frames(5, 2.0, seed=3)
    bits, its, conv = dec.decode_batch(llr.astype(np.float32), 0)                      # no sweeps allowed: the channel's decisions
    assert np.array_equal(bits, (llr > 0).astype(np.uint8)) and not conv.any() and its.tolist() == [0] * 5
    with pytest.raises(hip.LdpcError):
        dec.debug_step(np.zeros((1, c.N)), np.zeros((1, c.N)), np.zeros((1, c.E)))
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(code, "tanh", "f32", 8, schedule="layered", path="fused")
    assert e.value.code == -5 and "min-sum" in str(e.value)
    assert hip.Decoder(code, "tanh", "f32", 8, schedule="layered").path == "flood"     # AUTO falls back to the HBM kernel
    assert hip.Decoder(code, "min", "f64", 8, schedule="layered").path == "flood"
    assert hip.Decoder(c.hip_code(hip, prefer_qc=False), "min", "f32", 8, schedule="layered").path == "flood"


def test_record_by_name_runs_on_chip(hip):
    ecc = hip.ECC(CODES, "ldpc/hip-minsum-layered/jpl.4096.4.5/50/4/5", max_batch=64)
    assert ecc.decoder.path == "fused" and ecc.decoder.schedule == "layered"
    c = load("jpl.4096.4.5")
    _, llr = c.frames(6, 3.2, seed=44)
    lp = np.arange(0, c.M + 1, c.sz)
    for f in range(6):
        msg, ok = ecc.decode(llr[f, :5120])
        o = oracle.decode_layered(c.graph, lp, "min", 50, np.concatenate([llr[f, :5120], np.zeros(512)]))
        assert np.array_equal(msg, o["bits"][:4096]) and ok
    ecc.close()
