"""codes/1920.1280.A: the reference's fourth shipped matrix (= src/1920.1280.A), 5760 redundant checks of rank 1280 over 1920
bits (row weights 4/6, column weights 14/18, E = 32 000) -- the SAME code as 1920.1280.3.303 seen through six times as many
checks.  CPU: both loaders against the oracle parser, the rank that gives k = 640.  GPU: every path that takes it against
the oracle (f64 min-sum trajectories bit-exact, f32 bits/flags of EVERY frame, teacher-forced LLRs within a backward-stable bound), the on-chip instance that
holds its 150 KB of state in LDS, the record by name."""
import os

import numpy as np
import pytest

import ecc_ldpc_amd as E
from oracle import formats, oracle
from tests.helpers import CODES, lam_tolerance, load

NAME = "1920.1280.A"


def test_loader_shape_weights_rank():
    c = load(NAME)
    assert (c.M, c.N, c.E) == (5760, 1920, 32000)
    rw, cw = c.H.sum(1), c.H.sum(0)
    assert {int(w): int((rw == w).sum()) for w in np.unique(rw)} == {4: 1280, 6: 4480}
    assert {int(w): int((cw == w).sum()) for w in np.unique(cw)} == {14: 640, 18: 1280}
    assert formats.gf2_rank(c.H) == 1280 and formats.gf2_rank(load("1920.1280.3.303").H) == 1280
    m = E.Matrix.load_mackay(os.path.join(CODES, NAME))                                          # the product's C loader
    assert (m.rows, m.cols, m.sz) == (5760, 1920, 0) and np.array_equal(m.dense(), c.H) and m.rank() == 1280
    m3 = E.Matrix.load_mackay(os.path.join(CODES, "1920.1280.3.303"))
    assert m3.rank() == 1280
    rng = np.random.default_rng(3)
    for _ in range(3):                                                                           # rank against a matrix built to a known rank
        A = rng.integers(0, 2, (40, 7)).astype(np.int64) @ rng.integers(0, 2, (7, 90)).astype(np.int64) % 2
        assert formats.gf2_rank(A.astype(np.uint8)) <= 7


def test_same_code_as_the_full_rank_matrix():
    """every check of 1920.1280.3.303 lies in the row space of 1920.1280.A and vice versa (stacked rank stays 1280)"""
    a, b = load(NAME), load("1920.1280.3.303")
    assert formats.gf2_rank(np.concatenate([a.H, b.H])) == 1280


@pytest.mark.gpu
def test_record_by_name(hip):
    ecc = hip.ECC(CODES, "ldpc/hip-minsum/1920.1280.A/50", max_batch=64)
    assert (ecc.message_length, ecc.codeword_length, ecc.unpunctured_length) == (640, 1920, 1920)
    assert ecc.name == "ldpc/hip-minsum/1920.1280.A/50/1/3" and ecc.sim.encoder == "none"
    c = load(NAME)
    _, llr = c.frames(8, 2.5, seed=5)
    for f in range(8):
        msg, ok = ecc.decode(llr[f])
        o = oracle.decode(c.graph, "min", 50, llr[f])
        assert np.array_equal(msg, o["bits"][:640]) and ok
    ecc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["min", "tanh"])
def test_every_path_against_the_oracle(hip, variant):
    """f32 against the Double oracle, EVERY frame: hard bits, flags, and the turn a frame stops at.  Min-sum on this matrix multiplies
    the LLRs of an unconverged frame by ~6 per turn (3/4 x 17 other checks per column at most): they would leave the float range
    before turn 50.  The any-H kernels rescale such a frame by an exact power of two (min-sum is homogeneous: ldpc_math.h
    kRescales; r03 clipped at 2^100 and lost the frames the reference decodes after turn ~35), so the float trajectory goes on where
    the Double one does -- including for the frames decoded late, which this batch has."""
    c = load(NAME)
    llr = np.concatenate([c.frames(10, db, 3100)[1] for db in (1.5, 2.5, 3.0)])
    code = c.hip_code(hip)
    ob, oi, oc = oracle.decode_batch(c.graph, variant, 50, llr, nthreads=8)
    assert 0 < oc.sum() < len(llr) and len(set(oi.tolist())) > 3                  # a mixed batch
    late = oc.astype(bool) & (oi > 35)
    outs = {}
    for path in ("fused", "flood", "auto"):
        dec = hip.Decoder(code, variant, "f32", len(llr), path=path)
        outs[path] = dec.decode_batch(llr.astype(np.float32), 50, want_lam=True)
        if path == "fused":   # row degree class 6: lam + 6M message cells = 146 KB of LDS, one 1024-thread workgroup per CU
            assert dec.kernel_name in (f"ldpc::fused_csr_batched_kernel<float, {1 if variant == 'min' else 0}, 6, 6, 2, 18, 1024, 2, false>",
                                       f"ldpc::fused_csr_kernel<float, {1 if variant == 'min' else 0}, 6, 0, 0, 1024>"), dec.kernel_name
        if path == "auto":
            print(f"{NAME} {variant}: LDPC_PATH_AUTO -> {dec.path} ({dec.kernel_name}); frames the oracle decodes after turn 35: {int(late.sum())}, "
                  f"largest |LLR| handed back {np.abs(outs[path][3]).max():.3g}")
        bits, its, conv, lam = outs[path]
        assert np.array_equal(bits, ob) and np.array_equal(conv.astype(bool), oc.astype(bool)), path
        assert (its == oi).mean() >= 0.9 and np.abs(its.astype(int) - oi).max() <= 1, (path, its, oi)
        assert np.isfinite(lam).all()
        dec.close()
    assert all(np.array_equal(x, y) for x, y in zip(outs["fused"], outs["flood"]))   # same arithmetic, same order
    # f64: the state (300 KB) does not fit on-chip -> the HBM path; min-sum trajectory bit-exact
    with pytest.raises(hip.LdpcError):
        hip.Decoder(code, variant, "f64", 4, path="fused")
    d64 = hip.Decoder(code, variant, "f64", 6)
    assert d64.path == "flood"
    bits, its, conv, trace = d64.decode_trace(llr[::5], 50)
    for i, f in enumerate(range(0, len(llr), 5)):
        o = oracle.decode(c.graph, variant, 50, llr[f], trace=True)
        assert its[i] == o["iters"] and bool(conv[i]) == o["converged"] and np.array_equal(bits[i], o["bits"])
        if variant == "min":
            assert np.array_equal(trace[i, : o["iters"] + 1], o["trace_lam"])
        else:
            assert (np.abs(trace[i, : o["iters"] + 1] - o["trace_lam"]) / np.maximum(1, np.abs(o["trace_lam"]))).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["min", "tanh"])
@pytest.mark.parametrize("path", ["fused", "flood"])
def test_teacher_forced_llrs_backward_stable_bound(hip, variant, path):
    """one update from the oracle's states.  NOT the plain 1e-5 bar: on this matrix a column adds 14 or 18 messages of size ~1e4 that
    cancel to ~1, and float is exact to 2^-24 of the OPERANDS; the assertion is 1e-5 max(1, |lam|) + 2e-6 x (sum of the magnitudes added),
    and the error against max(1, |lam|) alone is measured and printed (r03: up to 1.12 for min-sum)"""
    c = load(NAME)
    llr = np.concatenate([c.frames(2, db, 3300 + i)[1] for i, db in enumerate((1.0, 2.5))])
    dec = hip.Decoder(c.hip_code(hip), variant, "f32", 32, path=path)
    states = []
    for f in range(len(llr)):
        o = oracle.decode(c.graph, variant, 12, llr[f], trace=True)
        ne = np.zeros(c.E)
        for n in range(o["iters"]):
            states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
            ne = o["trace_ne"][n]
    states = states[:32]
    ne2, lam2, syn0 = dec.debug_step(np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), np.stack([s[2] for s in states]))
    assert not syn0.any()
    worst = 0.0
    worst_sum = 0.0
    for i, s in enumerate(states):
        tol_lam, tol_ne = lam_tolerance(c.graph, s[3], s[4], rel=1e-5)
        # A column here adds 14 or 18 messages that largely cancel (sum of |terms| up to 360 x |lam| in these trajectories): the
        # float sum is exact to ~18 * 2^-24 of the SUM OF MAGNITUDES, not of the result.  The 1e-5 bar is therefore applied to
        # max(1, |lam|) PLUS 2e-6 x (|orig| + sum |ne'|) -- the backward-stable form of the same requirement.
        mags = np.abs(s[0]).copy()
        np.add.at(mags, c.graph.col_idx, np.abs(s[3]))
        # likewise a message is made of t = lam - ne (both ~1e4 after a few turns here, their difference possibly ~1): its float
        # error is 2^-24 of the operands, the largest of the row for a min-sum output
        t_mag = np.abs(s[1])[c.graph.col_idx] + np.abs(s[2])
        row_mag = np.repeat(np.maximum.reduceat(t_mag, c.graph.row_ptr[:-1]), np.diff(c.graph.row_ptr))
        assert (np.abs(ne2[i] - s[3]) <= tol_ne + 2e-6 * row_mag).all() and (np.abs(lam2[i] - s[4]) <= tol_lam + 2e-6 * mags).all()
        worst = max(worst, (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max())
        worst_sum = max(worst_sum, (np.abs(lam2[i] - s[4]) / np.maximum(1, mags)).max())
    print(f"{NAME} {path} {variant} f32: worst teacher-forced LLR error {worst:.3e} relative to max(1,|lam|), {worst_sum:.3e} relative to the "
          f"column's sum of magnitudes, over {len(states)} turns")
