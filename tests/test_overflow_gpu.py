"""GPU: f32 min-sum on a code whose LLRs leave the float range (heavy columns, light redundant checks -- the QC counterpart of
codes/1920.1280.A).  The reference's Double does not overflow within 100 turns; a float does (here the LLRs of a failing frame grow x3.7 per turn: past FLT_MAX around turn 68): inf - inf = NaN, hard NaN = False, an
all-zero "codeword" with a zero syndrome.  The any-H kernels rescale such frames (tests/test_redundant_checks_gpu.py); the QC kernels --
run-time specialised split kernel, flood_qc_kernel from HBM, on-chip and HBM layered -- examine a frame when its stop rule fires and turn
"converged" into "failed" when an LLR is not finite (ldpc_math.h kVetoesNonFinite).  Bar: no frame the Double
oracle fails may come back converged; frames it decodes before anything overflows come back identical."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import synthetic

NAME = "heavycol-24x8-sz64"
FLT_MAX = 3.4028234663852886e38
TURNS = 100


def _frames(c):
    return np.concatenate([c.frames(16, db, 9100 + i)[1] for i, db in enumerate((0.0, 2.0, 4.0))])


def test_the_double_trajectories_leave_the_float_range():
    """(CPU) the premise: on this code the oracle's LLRs of frames it cannot decode pass FLT_MAX before turn 100 -- and stay finite in Double"""
    c = synthetic(NAME)
    llr = _frames(c)
    over = 0
    for f in range(0, len(llr), 4):
        o = oracle.decode(c.graph, "min", TURNS, llr[f], trace=True)
        big = np.abs(o["trace_lam"]).max()
        assert np.isfinite(big)
        over += (not o["converged"]) and big > FLT_MAX
    assert over >= 2, over


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["split", "flood_qc", "layered_on_chip", "layered_hbm"])
def test_overflowing_frames_fail_instead_of_converging(hip, which):
    c = synthetic(NAME)
    llr = _frames(c)
    lp = c.layer_ptr
    layered = which.startswith("layered")
    TURNS = 40 if layered else 100          # (a layered sweep multiplies the LLRs by ~1700 here: a float is gone by sweep 12, a Double by 95)
    if layered:
        ref = [oracle.decode_layered(c.graph, lp, "min", TURNS, l) for l in llr]
        ob = np.stack([o["bits"] for o in ref]); oc = np.array([o["converged"] for o in ref]); oi = np.array([o["iters"] for o in ref])
    else:
        ob, oi, oc = oracle.decode_batch(c.graph, "min", TURNS, llr, nthreads=8)
        oc = oc.astype(bool)
    assert 0 < oc.sum() < len(llr)
    dec = hip.Decoder(c.hip_code(hip), "min", "f32", len(llr), schedule="layered" if layered else "flooding",
                      path={"split": "fused", "flood_qc": "flood", "layered_on_chip": "fused", "layered_hbm": "flood"}[which])
    want = {"split": "ldpc_jit_split_", "flood_qc": "flood_qc_kernel", "layered_on_chip": "ldpc_jit_layered_", "layered_hbm": "layered_qc_kernel"}[which]
    assert want in dec.kernel_name, dec.kernel_name
    bits, its, conv, lam = dec.decode_batch(llr.astype(np.float32), TURNS, want_lam=True)
    conv = conv.astype(bool)
    hard_in = (llr.astype(np.float32) > 0).astype(np.uint8)
    assert not conv[~oc].any(), (which, np.flatnonzero(conv & ~oc))                    # THE bar: nothing the oracle fails comes back converged
    assert np.array_equal(bits[~conv], hard_in[~conv]) and (its[~conv] == TURNS).all()    # a failure carries the channel's decisions (Orig.hs:70)
    assert np.isfinite(lam).all()
    early = oc & (oi <= (4 if layered else 20))                                                            # decoded long before anything can overflow: identical
    assert early.sum() >= 8 and np.array_equal(bits[early], ob[early]) and conv[early].all()
    lost = oc & ~conv                                                                  # decoded by the Double oracle, failed here: only late ones
    assert (oi[lost] > (8 if layered else 40)).all(), (which, oi[lost])
    print(f"{NAME} {which}: oracle decodes {int(oc.sum())}/{len(llr)}, kernel {int(conv.sum())}; oracle-decoded frames reported failed: {int(lost.sum())}"
          f" (their turns: {oi[lost].tolist()})")
    dec.close()
