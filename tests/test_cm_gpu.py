"""GPU: the reference's `arraylet-cm` numerics (src/ECC/Code/LDPC/Fast/CachedMult.hs: StableDiv row products, SURVEY.md row
a10) as a double-precision parity mode of the flood path (LDPC_TANH_CM), against its restatement in the oracle.  The
tanh rule and its `cm` flavour are the same real function ~1e-11 apart; the device follows each oracle flavour to the
libm's last ulps, teacher-forced to 1e-13."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import CODES, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,db", [("moon.7.13", 3.0), ("jpl.1024.4.5", 3.2), ("1920.1280.3.303", 2.5)])
def test_cm_follows_the_cm_oracle(hip, name, db):
    c = load(name)
    _, llr = c.frames(6, db, seed=61)
    dec = hip.Decoder(c.hip_code(hip), "cm", "f64", 6)
    assert dec.path == "flood"
    bits, its, conv, trace = dec.decode_trace(llr, 30)
    worst_cm = worst_tanh = 0.0
    for f in range(6):
        o = oracle.decode(c.graph, "cm", 30, llr[f], trace=True)
        t = oracle.decode(c.graph, "tanh", 30, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"] and np.array_equal(bits[f], o["bits"])
        n = o["iters"]
        worst_cm = max(worst_cm, (np.abs(trace[f, : n + 1] - o["trace_lam"]) / np.maximum(1, np.abs(o["trace_lam"]))).max())
        m = min(n, t["iters"])
        worst_tanh = max(worst_tanh, (np.abs(trace[f, : m + 1] - t["trace_lam"][: m + 1]) / np.maximum(1, np.abs(t["trace_lam"][: m + 1]))).max())
    assert worst_cm <= 1e-10
    print(f"{name}: device cm vs oracle cm {worst_cm:.1e}, vs oracle tanh {worst_tanh:.1e}")
    # one teacher-forced turn from oracle-cm states
    states = []
    for f in range(3):
        o = oracle.decode(c.graph, "cm", 30, llr[f], trace=True)
        ne = np.zeros(c.E)
        for n in range(min(o["iters"], 4)):
            states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
            ne = o["trace_ne"][n]
    if states:
        ne2, lam2, _ = dec_step(hip, c, states)
        for i, s in enumerate(states):
            scale = 1e-13 * (1 + np.exp(np.minimum(np.abs(s[3]).max(), 36)) * 2.0 ** -30)
            assert (np.abs(ne2[i] - s[3]) / np.maximum(1, np.abs(s[3]))).max() <= scale
            assert (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max() <= scale


def dec_step(hip, c, states):
    dec = hip.Decoder(c.hip_code(hip), "cm", "f64", len(states))
    return dec.debug_step(np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), np.stack([s[2] for s in states]))


def test_cm_is_a_parity_mode_only(hip):
    c = load("jpl.1024.4.5")
    code = c.hip_code(hip)
    for kw in (dict(dtype="f32"), dict(dtype="f64", path="fused"), dict(dtype="f64", schedule="layered")):
        with pytest.raises(hip.LdpcError) as e:
            hip.Decoder(code, "cm", max_batch=4, **kw)
        assert e.value.code == -5
    ecc = hip.ECC(CODES, "ldpc/hip-tanh-cm-f64/jpl.1024.4.5/20/4/5", max_batch=2)
    _, llr = c.frames(1, 4.0, seed=9)
    out, ok = ecc.decode(llr[0][:1280])
    assert ok and np.array_equal(out, oracle.decode(c.graph, "cm", 20, llr[0])["bits"][:1024])
