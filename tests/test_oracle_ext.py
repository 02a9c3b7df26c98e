"""CPU: the two additions to the oracle made in round 2, each against an independent literal transliteration.

* `cm` -- the reference's `arraylet-cm` numerics (src/ECC/Code/LDPC/Fast/CachedMult.hs:25-56,233-264, SURVEY.md row
  a10): the tanh rule with the row product cached as a StableDiv and leave-one-out by division.  Same real function
  as Reference.Orig, different roundings: hard bits and iteration counts agree with the tanh rule on every test frame,
  LLR trajectories to ~1e-10.
* layered -- BASELINE.json configs[4]; an extension with no reference counterpart whose specification is the header
  of oracle_decode_layered.  Checked: the restatements agree bit for bit, a converged frame is a codeword, the
  source-style invariants (noiseless input: 0 sweeps; failure: hard decisions of the channel), the column-disjointness
  check, and that it needs about half the sweeps flooding needs."""
import numpy as np
import pytest

from oracle import literal, oracle
from tests.helpers import load


def test_cm_restatements_agree_bitwise_and_match_tanh_rule():
    c = load("jpl.1024.4.5")
    _, llr = c.frames(6, 3.2, seed=41)
    tr = []
    b, it, cv = literal.ldpc_cm(c.sz, c.offsets, 3, llr[0], trace=tr)          # literal Matrixlet folds, first turns
    o = oracle.decode(c.graph, "cm", 3, llr[0], trace=True)
    assert it == o["iters"] and cv == o["converged"] and np.array_equal(b, o["bits"])
    assert np.array_equal(np.array(tr), o["trace_lam"])
    worst = 0.0
    for f in range(6):
        a = oracle.decode(c.graph, "cm", 50, llr[f], trace=True)
        t = oracle.decode(c.graph, "tanh", 50, llr[f], trace=True)
        assert a["iters"] == t["iters"] and a["converged"] == t["converged"] and np.array_equal(a["bits"], t["bits"])
        d = np.abs(a["trace_lam"] - t["trace_lam"]) / np.maximum(1, np.abs(t["trace_lam"]))
        worst = max(worst, d.max())
        assert not np.array_equal(a["trace_lam"], t["trace_lam"]) or a["iters"] == 0    # it IS a different rounding sequence
    assert worst < 1e-8
    print(f"arraylet-cm vs Reference.Orig on jpl.1024: worst relative LLR difference {worst:.2e}, bits identical")


def test_cm_on_a_small_qc_code_all_turns():
    """a 3 x 6 block toy QC code: the literal fold order and the CSR form agree on whole trajectories"""
    rng = np.random.default_rng(3)
    sz = 8
    off = np.array([[0, 3, -1, 5, 1, -1], [2, -1, 6, 0, -1, 4], [-1, 7, 1, -1, 3, 2]], np.int32)
    H = np.zeros((3 * sz, 6 * sz), np.uint8)
    for br in range(3):
        for bc in range(6):
            if off[br, bc] >= 0:
                for r in range(sz):
                    H[br * sz + r, bc * sz + (r + off[br, bc]) % sz] = 1
    g = oracle.Graph.from_dense(H)
    for t in range(10):
        llr = rng.normal(1.5, 2.0, 6 * sz)
        tr = []
        b, it, cv = literal.ldpc_cm(sz, off, 12, llr, trace=tr)
        o = oracle.decode(g, "cm", 12, llr, trace=True)
        assert it == o["iters"] and cv == o["converged"] and np.array_equal(b, o["bits"]) and np.array_equal(np.array(tr), o["trace_lam"])


@pytest.mark.parametrize("variant", ["min", "tanh"])
def test_layered_restatements_agree_bitwise(variant):
    c = load("moon.7.13")
    lp = np.arange(c.M + 1)                       # every row its own layer
    for t in range(16):
        _, llr = c.frames(1, 1.0 + (t % 5), seed=300 + t)
        tr = []
        b, it, cv = literal.ldpc_layered(c.H, lp, variant, 20, llr[0], trace=tr)
        o = oracle.decode_layered(c.graph, lp, variant, 20, llr[0], trace=True)
        assert it == o["iters"] and cv == o["converged"] and np.array_equal(b, o["bits"])
        assert np.array_equal(np.array(tr), o["trace_lam"])
    j = load("jpl.1024.4.5")
    lpj = np.arange(0, j.M + 1, j.sz)             # one block row per layer
    _, llr = j.frames(1, 3.0, seed=7)
    tr = []
    b, it, cv = literal.ldpc_layered(j.H, lpj, variant, 2, llr[0], trace=tr)
    o = oracle.decode_layered(j.graph, lpj, variant, 2, llr[0], trace=True)
    assert it == o["iters"] and np.array_equal(np.array(tr), o["trace_lam"])


def test_layered_invariants_and_speed_of_convergence():
    c = load("jpl.1024.4.5")
    lp = np.arange(0, c.M + 1, c.sz)
    cws, llr = c.frames(40, 3.0, seed=11)
    fl = [oracle.decode(c.graph, "min", 50, l) for l in llr]
    la = [oracle.decode_layered(c.graph, lp, "min", 50, l) for l in llr]
    for f, o in enumerate(la):
        if o["converged"]:
            assert not ((c.H.astype(np.int64) @ o["bits"]) % 2).any()          # a converged frame IS a codeword
        else:
            assert o["iters"] == 50 and np.array_equal(o["bits"], (llr[f] > 0).astype(np.uint8))
    both = [(a["iters"], b["iters"]) for a, b in zip(fl, la) if a["converged"] and b["converged"]]
    assert len(both) >= 15 and sum(b for _, b in both) < 0.75 * sum(a for a, _ in both)
    assert sum(o["converged"] for o in la) >= sum(o["converged"] for o in fl)
    # noiseless codeword: detected before the first sweep; all-zero LLRs: the all-zero word, 0 sweeps
    clean = (2.0 * cws[0] - 1.0) * 9.0
    o = oracle.decode_layered(c.graph, lp, "min", 50, clean)
    assert o["iters"] == 0 and o["converged"] and np.array_equal(o["bits"], cws[0])
    o = oracle.decode_layered(c.graph, lp, "tanh", 50, np.zeros(c.N))
    assert o["iters"] == 0 and o["converged"] and not o["bits"].any()
    # layers whose rows share a column are rejected (rows 0..63 of jpl.1024 span two block rows)
    with pytest.raises(RuntimeError):
        oracle.decode_layered(c.graph, np.arange(0, c.M + 1, 2 * c.sz), "min", 5, llr[0])


def test_channel_convention_against_the_counts_in_notes_txt():
    """/root/reference/NOTES.txt holds three runs of one configuration (jpl.1K without a puncturing rate, 0 dB, a decoder
    that passes the channel through, 256 x 1024 message bits): 29 891, 29 751 and 29 705 bit errors.  They pin the
    channel convention of the absent tester: sigma^2 = 1 / (2 R Eb/N0) with R = k / n_tx gives Q(sqrt(2 * 1024/1408)) =
    0.1139; the CPU-side channel of the oracle (oracle/channel.py) must produce that rate, and every one of the three
    counts must be a plausible draw of it (tests/test_notes_pin_gpu.py does the same for the device frame source)."""
    import math
    from oracle import channel
    k, n = 1024, 1408
    p = 0.5 * math.erfc(math.sqrt(2.0 * k / n) / math.sqrt(2.0))
    assert abs(p - 0.11390) < 2e-5
    bits = 256 * 1024
    sd = math.sqrt(p * (1 - p) / bits)
    for errors in (29891, 29751, 29705):                       # NOTES.txt:3, 8, 13
        assert abs(errors / bits - p) < 3 * sd, errors
    assert abs(0.5 * math.erfc(1.0) - 29891 / bits) > 50 * sd   # sigma^2 = 1/(2 Eb/N0), rate not folded in: Q(sqrt 2) = 0.0786
    F = 2048
    cw = np.zeros((F, n), np.uint8)
    llr = channel.frames(cw, 0.0, k, n, n, seed=20)
    wrong = int((llr[:, :k] > 0).sum())                        # all-zero codeword: a positive LLR is a bit error
    sd2 = math.sqrt(p * (1 - p) * F * k)
    assert abs(wrong - p * F * k) < 4 * sd2, (wrong, p * F * k)
