"""The reference registers several decoders of the same function (main/Main.hs:34-36): `reference` / `min` (Reference/Orig.hs,
Min.hs), `sparse` / `sparsemin` (Reference/Sparse.hs, SparseMin.hs), `arraylet` / `arraylet-min` (Fast/Arraylet.hs,
ArrayletMin.hs).  They compute the same check rule and add a column up in three different orders, so their Double trajectories
differ in the last ulps.  CPU: the oracle's variants against line-by-line transliterations of each decoder (oracle/literal.py).
GPU: the f64 parity modes (ldpc_sum_order) reproduce EACH decoder's trajectory bit for bit (min-sum) / to 1e-11 (tanh), and the
reference's own names select them."""
import numpy as np
import pytest

from oracle import literal, oracle
from tests.helpers import CODES, load


@pytest.mark.parametrize("variant,base", [("sparse", "tanh"), ("sparsemin", "min")])
def test_sparse_oracle_equals_the_literal_transliteration(variant, base):
    c = load("moon.7.13")
    _, llr = c.frames(6, 2.5, seed=51)
    for f in range(6):
        tr = []
        b, it, cv = literal.ldpc_sparse(c.H, base, 20, llr[f], trace=tr)
        o = oracle.decode(c.graph, variant, 20, llr[f], trace=True)
        assert it == o["iters"] and cv == o["converged"] and np.array_equal(b, o["bits"]) and np.array_equal(np.array(tr), o["trace_lam"])


@pytest.mark.parametrize("variant,base", [("arraylet", "tanh"), ("arraylet-min", "min")])
def test_arraylet_oracle_equals_the_literal_transliteration(variant, base):
    c = load("jpl.1024.4.5")
    _, llr = c.frames(1, 3.0, seed=52)
    tr = []
    b, it, cv = literal.ldpc_arraylet(c.sz, c.offsets, base, 6, llr[0], trace=tr)
    o = oracle.decode(c.graph, variant, 6, llr[0], trace=True)
    assert it == o["iters"] and cv == o["converged"] and np.array_equal(b, o["bits"]) and np.array_equal(np.array(tr), o["trace_lam"])


def test_the_orders_really_differ_in_the_last_ulps():
    """otherwise the parity modes below would prove nothing"""
    c = load("jpl.1024.4.5")
    _, llr = c.frames(1, 3.0, seed=53)
    ref = oracle.decode(c.graph, "tanh", 8, llr[0], trace=True)["trace_lam"]
    arr = oracle.decode(c.graph, "arraylet", 8, llr[0], trace=True)["trace_lam"]
    spa = oracle.decode(c.graph, "sparse", 8, llr[0], trace=True)["trace_lam"]
    for x, y in ((ref, arr), (ref, spa), (arr, spa)):
        d = np.abs(x - y).max()
        assert 0 < d < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["moon.7.13", "jpl.1024.4.5", "1920.1280.3.303"])
@pytest.mark.parametrize("order,tanh_name,min_name", [("arraylet", "arraylet", "arraylet-min"), ("sparse", "sparse", "sparsemin")])
def test_f64_parity_modes_reproduce_each_decoder(hip, name, order, tanh_name, min_name):
    c = load(name)
    llr = np.concatenate([c.frames(3, db, 5400 + i)[1] for i, db in enumerate((2.0, 3.5))])
    iters = 20 if name == "moon.7.13" else 30
    code = c.hip_code(hip)
    dec = hip.Decoder(code, "min", "f64", len(llr), sum_order=order)
    assert dec.path == "flood"
    bits, its, conv, trace = dec.decode_trace(llr, iters)
    for f in range(len(llr)):
        o = oracle.decode(c.graph, min_name, iters, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"] and np.array_equal(bits[f], o["bits"])
        assert np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"]), (name, order, f)
    dec.close()
    dec = hip.Decoder(code, "tanh", "f64", len(llr), sum_order=order)
    bits, its, conv, trace = dec.decode_trace(llr, iters)
    for f in range(len(llr)):
        o = oracle.decode(c.graph, tanh_name, iters, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"] and np.array_equal(bits[f], o["bits"])
        ne_max = np.abs(o["trace_ne"]).max() if o["iters"] else 0.0
        tol = 1e-11 * (1 + np.exp(min(ne_max, 36)) * 2.0 ** -30)
        assert (np.abs(trace[f, : o["iters"] + 1] - o["trace_lam"]) / np.maximum(1, np.abs(o["trace_lam"]))).max() <= tol
    dec.close()
    with pytest.raises(hip.LdpcError):
        hip.Decoder(code, "min", "f64", 4, sum_order=order, path="fused")     # parity modes live on the flood path


@pytest.mark.gpu
def test_reference_names_select_their_decoders_order(hip):
    """`ldpc/arraylet-min-f64/...` and `ldpc/sparsemin-f64/...` are bit-exact with THEIR decoder; the f32 names run the on-chip kernels"""
    c = load("jpl.1024.4.5")
    _, llr = c.frames(4, 3.0, seed=5500)
    for nm, variant in (("arraylet-min-f64", "arraylet-min"), ("sparsemin-f64", "sparsemin"), ("min-f64", "min")):
        ecc = hip.ECC(CODES, f"ldpc/{nm}/jpl.1024.4.5/30/4/5", max_batch=8)
        assert ecc.decoder.path == ("flood" if nm != "min-f64" else ecc.decoder.path)
        bits, its, conv, trace = ecc.decoder.decode_trace(llr, 30)
        for f in range(4):
            o = oracle.decode(c.graph, variant, 30, llr[f], trace=True)
            assert its[f] == o["iters"] and np.array_equal(bits[f], o["bits"]) and np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"]), nm
        ecc.close()
    ecc = hip.ECC(CODES, "ldpc/arraylet-min/jpl.1024.4.5/30/4/5", max_batch=8)
    assert ecc.decoder.path == "fused"
    ecc.close()
