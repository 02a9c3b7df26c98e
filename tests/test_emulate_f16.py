"""CPU: sanity of the fp16-storage emulation used as the checker of the LDPC_F16 contexts (oracle/emulate_f16.py)
against the double-precision oracle and the behavioural invariants of the reference loop (Min.hs:54-104)."""
import numpy as np

from oracle import emulate_f16 as em
from oracle import oracle
from tests.helpers import load


def test_r16_saturates_and_rounds_to_nearest_even():
    x = np.array([7e4, -7e4, 65504.0, 65520.0, 1e-9, 0.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 2.0 ** -25, 3.0e-8], np.float32)
    want = np.array([65504.0, -65504.0, 65504.0, 65504.0, 0.0, 0.0, 1.0, 1.0 + 2.0 ** -9, 0.0, 2.0 ** -24 * 1], np.float32)
    assert np.array_equal(em.r16(x), want)


def test_cn_minsum_matches_the_oracle_rule():
    c = load("moon.7.13")
    rng = np.random.default_rng(5)
    lam = em.r16(rng.normal(0, 4, (3, c.N)))
    ne = em.r16(rng.normal(0, 1, (3, c.E)))
    for f in range(3):
        o_ne, _, _ = oracle.step(c.graph, "min", np.zeros(c.N), lam[f].astype(np.float64), ne[f].astype(np.float64))
        e_ne, _, _ = em.step_minsum_f16_flood(c.graph, lam[f:f + 1] * 0, lam[f:f + 1], ne[f:f + 1])
        # same rule; the emulation subtracts in f32 and stores fp16
        t_ok = np.abs(e_ne[0] - o_ne) <= np.abs(o_ne) * 2.0 ** -10 + 1e-7
        assert t_ok.all()


def test_invariants_and_agreement_with_the_f64_oracle():
    c = load("jpl.1024.4.5")
    cws, llr = c.frames(6, 4.0, seed=31)
    llr = llr.astype(np.float32)
    bits, its, conv, trace = em.decode_minsum_f16_flood(c.graph, llr, 50)
    for f in range(len(llr)):                       # well above the waterfall: both decoders return the codeword
        o = oracle.decode(c.graph, "min", 50, llr[f].astype(np.float64))
        assert conv[f] and o["converged"] and np.array_equal(bits[f], o["bits"]) and np.array_equal(bits[f], cws[f])
        assert abs(int(its[f]) - o["iters"]) <= 2
    # noiseless codeword: 0 iterations, output = codeword; all-zero LLRs: 0 iterations, all False (hard 0 = False)
    clean = np.where(cws[:2] > 0, 8.0, -8.0).astype(np.float32)
    b, i, cv, _ = em.decode_minsum_f16_flood(c.graph, clean, 50)
    assert (i == 0).all() and cv.all() and np.array_equal(b, cws[:2])
    b, i, cv, _ = em.decode_minsum_f16_flood(c.graph, np.zeros((1, c.N), np.float32), 50)
    assert i[0] == 0 and cv[0] and not b.any()
    # out of turns -> hard(channel LLRs) (Min.hs:76)
    _, noisy = c.frames(2, 0.0, seed=32)
    b, i, cv, _ = em.decode_minsum_f16_flood(c.graph, noisy.astype(np.float32), 5)
    assert (i == 5).all() and not cv.any() and np.array_equal(b, (em.r16(noisy) > 0).astype(np.uint8))
