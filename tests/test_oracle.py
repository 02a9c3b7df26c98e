"""CPU: the C oracle against the independent literal transliteration and against invariants read
off the reference source (the reference has no golden vectors: SURVEY.md section 4)."""
import numpy as np
import pytest

from oracle import literal, oracle
from tests.helpers import load


@pytest.mark.parametrize("variant", ["tanh", "min"])
def test_three_restatements_agree_bitwise_moon(variant):
    c = load("moon.7.13")
    for t in range(24):
        _, llr = c.frames(1, 1.0 + (t % 5), seed=10 + t)
        tr = []
        b, it, cv = literal.ldpc(c.H, variant, 20, llr[0], trace=tr)
        o = oracle.decode(c.graph, variant, 20, llr[0], trace=True)
        d = oracle.decode_dense(c.H, variant, 20, llr[0], trace=True)
        assert it == o["iters"] == d["iters"] and cv == o["converged"] == d["converged"]
        assert np.array_equal(np.array(tr), o["trace_lam"]) and np.array_equal(o["trace_lam"], d["trace_lam"])
        assert np.array_equal(b, o["bits"]) and np.array_equal(b, d["bits"])


@pytest.mark.parametrize("variant", ["tanh", "min"])
def test_dense_and_sparse_agree_bitwise_jpl1024(variant):
    c = load("jpl.1024.4.5")
    _, llr = c.frames(2, 3.0, seed=5)
    for f in range(2):
        o = oracle.decode(c.graph, variant, 6, llr[f], trace=True)
        d = oracle.decode_dense(c.H, variant, 6, llr[f], trace=True)
        assert o["iters"] == d["iters"]
        assert np.array_equal(o["trace_lam"], d["trace_lam"]) and np.array_equal(o["bits"], d["bits"])


def test_literal_python_matches_c_on_jpl1024_first_turns():
    c = load("jpl.1024.4.5")
    _, llr = c.frames(1, 3.0, seed=9)
    for variant in ("tanh", "min"):
        tr = []
        literal.ldpc(c.H, variant, 2, llr[0], trace=tr)
        o = oracle.decode(c.graph, variant, 2, llr[0], trace=True)
        assert np.array_equal(np.array(tr), o["trace_lam"])


@pytest.mark.parametrize("name", ["moon.7.13", "jpl.1024.4.5"])
@pytest.mark.parametrize("variant", ["tanh", "min"])
def test_source_invariants(name, variant):
    c = load(name)
    cws, _ = c.frames(3, 50.0, seed=1)
    for cw in cws:
        # noiseless codeword: syndrome zero at turn 0 -> 0 iterations, output = codeword (Orig.hs:69)
        llr = (2.0 * cw - 1.0) * 8.0
        o = oracle.decode(c.graph, variant, 50, llr)
        assert o["iters"] == 0 and o["converged"] and np.array_equal(o["bits"], cw)
    # all-zero LLR: hard 0 = False everywhere, syndrome zero -> all False, 0 iterations
    o = oracle.decode(c.graph, variant, 50, np.zeros(c.N))
    assert o["iters"] == 0 and o["converged"] and not o["bits"].any()


def test_non_convergence_returns_channel_hard_decisions():
    c = load("jpl.1024.4.5")
    _, llr = c.frames(2, 1.0, seed=3)  # far below the waterfall
    for variant in ("tanh", "min"):
        o = oracle.decode(c.graph, variant, 5, llr[0])
        assert not o["converged"] and o["iters"] == 5
        assert np.array_equal(o["bits"], (llr[0] > 0).astype(np.uint8))  # Orig.hs:70
        assert np.array_equal(o["lam"], llr[0])


def test_max_iters_zero():
    c = load("moon.7.13")
    _, llr = c.frames(1, 0.0, seed=4)
    o = oracle.decode(c.graph, "min", 0, llr[0])
    assert o["iters"] == 0 and np.array_equal(o["bits"], (llr[0] > 0).astype(np.uint8))


def test_atanh_clamp_value():
    # Utils.hs:115: the clamp is atanh of the largest double below 1 under base-4.9's formula
    x = np.nextafter(1.0, 0.0)
    assert 0.5 * np.log((1.0 + x) / (1.0 - x)) == literal.ATANH_CLAMP
    assert literal.atanh_prime(1.0) == literal.ATANH_CLAMP and literal.atanh_prime(-1.0) == -literal.ATANH_CLAMP


def test_minsum_degree_one_row_is_an_error():
    H = np.array([[1, 0, 0], [1, 1, 1]], np.uint8)
    g = oracle.Graph.from_dense(H)
    with pytest.raises(RuntimeError):
        oracle.decode(g, "min", 3, np.array([1.0, -2.0, 3.0]))
    o = oracle.decode(g, "tanh", 1, np.array([1.0, -2.0, 3.0]), trace=True)
    assert o["trace_ne"][0][0] == -2 * literal.ATANH_CLAMP  # product [] = 1 -> atanh' 1 -> clamp


def test_batch_driver_matches_single(tmp_path):
    c = load("jpl.1024.4.5")
    _, llr = c.frames(6, 3.0, seed=12)
    bits, iters, conv = oracle.decode_batch(c.graph, "min", 50, llr, nthreads=3)
    for f in range(6):
        o = oracle.decode(c.graph, "min", 50, llr[f])
        assert np.array_equal(bits[f], o["bits"]) and iters[f] == o["iters"] and bool(conv[f]) == o["converged"]


def test_step_matches_trace():
    c = load("jpl.1024.4.5")
    _, llr = c.frames(1, 3.0, seed=21)
    for variant in ("tanh", "min"):
        o = oracle.decode(c.graph, variant, 8, llr[0], trace=True)
        ne = np.zeros(c.E)
        for n in range(o["iters"]):
            ne2, lam2, syn0 = oracle.step(c.graph, variant, llr[0], o["trace_lam"][n], ne)
            assert np.array_equal(ne2, o["trace_ne"][n]) and np.array_equal(lam2, o["trace_lam"][n + 1]) and not syn0
            ne = ne2
