"""The reference's LIVE GPU decoder, `cuda-arraylet2` (GPU/CUDA/Arraylet2.hs + cudabits/arraylet2.cu, common.h), does not compute
Orig.hs's Double arithmetic: its state is float, a factor is the double tanh stored as a float, the leave-one-out product runs in a
double register, and atanh_ takes it as a FLOAT -- so the +-18.71 clamp fires when the product rounds to +-1 in float (messages
saturate at |ne'| = 37.43 from |x| ~ 17 on, where the Double decoder still resolves them), and a column is summed in float, ascending.
CPU: the oracle variant "cuda32" against a kernel-by-kernel, thread-by-thread transliteration (oracle/literal.py), bit for bit; the two
arithmetics really differ.  GPU: the parity mode LDPC_TANH_CUDA32 (f32, flood path) against the oracle variant -- one update from
the oracle's states within 1e-5 (device tanh / atanhf against the C library's: last ulps), hard bits / flags / turn counts of
free-running frames; the reference's own name selects it."""
import numpy as np
import pytest

from oracle import literal, oracle
from tests.helpers import CODES, load, synthetic


def test_oracle_variant_equals_the_literal_transliteration():
    c = synthetic("small-2x4-sz32")
    llr = np.concatenate([c.frames(4, db, 6100 + i)[1] for i, db in enumerate((1.0, 3.0, 6.0))])
    turns = set()
    for f in range(len(llr)):
        tr = []
        b, it, cv = literal.ldpc_cuda_arraylet2(c.sz, c.offsets, 15, llr[f], trace=tr)
        o = oracle.decode(c.graph, "cuda32", 15, llr[f], trace=True)
        assert it == o["iters"] and cv == o["converged"] and np.array_equal(b, o["bits"])
        assert np.array_equal(np.array(tr[: it + 1]), o["trace_lam"])
        turns.add(it)
    assert len(turns) > 2
    c = load("jpl.1024.4.5")                                   # the shipped matrix, first turns (the transliteration walks every absent block)
    _, llr = c.frames(1, 3.0, seed=6200)
    tr = []
    b, it, cv = literal.ldpc_cuda_arraylet2(c.sz, c.offsets, 2, llr[0], trace=tr)
    o = oracle.decode(c.graph, "cuda32", 2, llr[0], trace=True)
    assert it == o["iters"] and np.array_equal(np.array(tr[: it + 1]), o["trace_lam"])


def test_it_is_another_function_than_the_double_decoder_where_messages_saturate():
    """strong LLRs: the float product rounds to +-1 and the clamp (37.43) fires where Orig.hs still resolves the message; weak LLRs:
    the two agree to float precision"""
    c = load("jpl.1024.4.5")
    rng = np.random.default_rng(5)
    weak = rng.normal(0, 2.0, c.N)
    strong = np.where(rng.random(c.N) < 0.5, 1.0, -1.0) * rng.uniform(18.0, 30.0, c.N)
    for llr, same in ((weak, True), (strong, False)):
        a = oracle.step(c.graph, "tanh", llr, llr, np.zeros(c.E))[0]
        b = oracle.step(c.graph, "cuda32", llr, llr, np.zeros(c.E))[0]
        rel = np.abs(a - b) / np.maximum(1, np.abs(a))
        assert (rel.max() < 1e-5) == same, rel.max()
        if not same:
            assert np.isclose(np.abs(b).max(), 2 * 18.714973875118524, rtol=1e-6) and np.abs(a).max() < 2 * 18.7


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["moon.7.13", "jpl.1024.4.5", "1920.1280.3.303"])
def test_parity_mode_against_the_oracle_variant(hip, name):
    c = load(name)
    llr = np.concatenate([c.frames(8, db, 6300 + i)[1] for i, db in enumerate((2.0, 3.5, 6.0))]).astype(np.float32)
    turns = 20 if name == "moon.7.13" else 40
    dec = hip.Decoder(c.hip_code(hip), "cuda32", "f32", len(llr))
    assert dec.path == "flood" and "fused" not in dec.kernel_name
    bits, its, conv = dec.decode_batch(llr, turns)
    ob, oi, oc = oracle.decode_batch(c.graph, "cuda32", turns, llr.astype(np.float64), nthreads=8)
    same = conv.astype(bool) == oc.astype(bool)
    assert same.mean() >= 0.95 and np.array_equal(bits[same], ob[same]), (name, same.mean())
    assert (its[same] == oi[same]).mean() >= 0.9 and np.abs(its[same].astype(int) - oi[same]).max() <= 1
    # one update from the oracle's own states (also: strong LLRs, where the clamp fires)
    states = []
    for f in (0, 9, 17):
        o = oracle.decode(c.graph, "cuda32", 6, llr[f].astype(np.float64), trace=True)
        ne = np.zeros(c.E)
        for n in range(o["iters"]):
            states.append((llr[f].astype(np.float64), o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
            ne = o["trace_ne"][n]
    rng = np.random.default_rng(7)
    strong = (np.where(rng.random(c.N) < 0.5, 1.0, -1.0) * rng.uniform(15.0, 30.0, c.N)).astype(np.float32).astype(np.float64)
    ne2, lam2, _ = oracle.step(c.graph, "cuda32", strong, strong, np.zeros(c.E))
    states.append((strong, strong, np.zeros(c.E), ne2, lam2))
    states = states[: len(llr)]
    dn, dl, _ = dec.debug_step(np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), np.stack([s[2] for s in states]))
    worst = 0.0
    for i, s in enumerate(states):
        # a message next to the clamp is atanh of a float one ulp from 1: an ulp of the product moves it by ~0.35; everything else 1e-5
        near = np.abs(np.abs(s[3]) - 2 * 18.714973875118524) < 3.0
        e_ne = np.abs(dn[i] - s[3]) / np.maximum(1, np.abs(s[3]))
        assert e_ne[~near].max() <= 1e-5 and np.abs(dn[i] - s[3])[near].max(initial=0) <= 1.5, (name, i, e_ne[~near].max())
        worst = max(worst, e_ne[~near].max())
    clamped = int((np.abs(states[-1][3]) == np.float32(2 * 18.714973875118524)).sum())
    assert clamped > 0 and np.array_equal(np.abs(dn[len(states) - 1]) == np.float32(2 * 18.714973875118524), np.abs(states[-1][3]) == np.float32(2 * 18.714973875118524))
    print(f"{name} cuda32: worst teacher-forced message error {worst:.2e} away from the clamp over {len(states)} updates; {clamped} messages at the clamp, the same ones on the device")
    with pytest.raises(hip.LdpcError):
        hip.Decoder(c.hip_code(hip), "cuda32", "f64", 4)
    with pytest.raises(hip.LdpcError):
        hip.Decoder(c.hip_code(hip), "cuda32", "f32", 4, path="fused")
    dec.close()


@pytest.mark.gpu
def test_the_reference_name_selects_it(hip):
    ecc = hip.ECC(CODES, "ldpc/cuda-arraylet2/jpl.1024.4.5/50/4/5", max_batch=16)
    assert ecc.decoder.path == "flood" and ecc.name.startswith("ldpc/cuda-arraylet2/")
    c = load("jpl.1024.4.5")
    _, llr = c.frames(6, 3.5, seed=6400)
    for f in range(6):
        out, ok = ecc.decode(llr[f][:1280])
        o = oracle.decode(c.graph, "cuda32", 50, llr[f])
        assert ok == o["converged"] and (not ok or np.array_equal(out, o["bits"][:1024]))
    ecc.close()
