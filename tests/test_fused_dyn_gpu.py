"""GPU: the TABLE-DRIVEN instances of the fused QC kernels (fused_msg.hip `DynTab`, every circulant size the
library compiles: 32, 64, 128) against the CPU oracle.  The shipped matrices run kernels with compile-time
rotation tables and any other QC matrix gets such a kernel from the run-time compiler (tests/test_jit.py); with
that switched off (LDPC_JIT=0, or no hiprtc on the machine) a matrix with the AR4JA rate-4/5 block structure falls
back to a two-wave kernel that reads its rotations from memory -- and f64 (parity mode) always does.  Covered here:
  * synthetic codes: the block structure of codes/jpl.1024.4.5 with RANDOM rotations, sz = 32 / 64 / 128,
    f32 (min-sum, tanh) and f64 (min-sum): hard bits, flags, per-turn LLRs (f32 <= 1e-5 teacher-forced, f64 bit-exact);
  * the shipped codes forced onto the table-driven kernel (LDPC_FUSED_TABLE=dyn): identical to their
    compile-time-table kernels bit for bit."""
import numpy as np
import pytest

from oracle import channel, oracle
from tests.helpers import load, lam_tolerance, iters_agree

pytestmark = pytest.mark.gpu


def synthetic_ar4ja(sz, seed):
    """offsets [12][44] with the non-empty pattern of jpl.1024.4.5 and rotations uniform in [0, sz)."""
    base = load("jpl.1024.4.5").offsets
    rng = np.random.default_rng(seed)
    off = np.where(base >= 0, rng.integers(0, sz, base.shape), -1).astype(np.int32)
    H = np.zeros((12 * sz, 44 * sz), np.uint8)
    r = np.arange(sz)
    for br in range(12):
        for bc in range(44):
            if off[br, bc] >= 0:
                H[br * sz + r, bc * sz + (r + off[br, bc]) % sz] = 1   # QuasiCyclic.hs:19-25
    return off, oracle.Graph.from_dense(H)


def frames(g, sz, F, dbs, seed):
    """all-zero codeword (valid for every linear code) through the rate-4/5 punctured channel"""
    k, n_tx = 32 * sz, 40 * sz
    per = (F + len(dbs) - 1) // len(dbs)
    out = [channel.frames(np.zeros((per, g.N), np.uint8), db, k, n_tx, g.N, seed + i) for i, db in enumerate(dbs)]
    return np.concatenate(out)[:F]


@pytest.mark.parametrize("sz", [32, 64, 128])
def test_synthetic_rotations_f32(hip, sz, monkeypatch):
    monkeypatch.setenv("LDPC_JIT", "0")     # the table-driven kernels are what runs when the run-time compiler is off / absent
    off, g = synthetic_ar4ja(sz, 1000 + sz)
    code = hip.Code.from_qc(sz, off)
    F = 48 if sz < 128 else 24
    llr = frames(g, sz, F, (2.5, 3.5, 4.5), 2000 + sz)
    for variant in ("min", "tanh"):
        dec = hip.Decoder(code, variant, "f32", F, path="fused")
        assert "fused_msg_kernel" in dec.kernel_name      # no compile-time table for these rotations
        bits, its, conv = dec.decode_batch(llr.astype(np.float32), 50)
        ob, oi, oc = oracle.decode_batch(g, variant, 50, llr, nthreads=8)
        assert np.array_equal(bits, ob) and np.array_equal(conv, oc), (sz, variant)
        assert iters_agree(its, oi)
        flood = hip.Decoder(code, variant, "f32", F, path="flood")
        fb, fi, fc = flood.decode_batch(llr.astype(np.float32), 50)
        assert np.array_equal(bits, fb) and np.array_equal(its, fi) and np.array_equal(conv, fc)   # same arithmetic, same order
        print(f"sz={sz} {variant}: {int(conv.sum())}/{F} converged, iteration counts {100 * (its == oi).mean():.0f}% identical to the oracle")


@pytest.mark.parametrize("sz", [32, 64, 128])
def test_synthetic_rotations_teacher_forced_and_f64(hip, sz, monkeypatch):
    monkeypatch.setenv("LDPC_JIT", "0")
    off, g = synthetic_ar4ja(sz, 3000 + sz)
    code = hip.Code.from_qc(sz, off)
    llr = frames(g, sz, 6, (3.0, 4.0), 4000 + sz)
    # f64 min-sum: the whole trajectory is the oracle's, bit for bit
    d64 = hip.Decoder(code, "min", "f64", len(llr), path="fused")
    bits, its, conv, trace = d64.decode_trace(llr, 30)
    for f in range(len(llr)):
        o = oracle.decode(g, "min", 30, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"]
        assert np.array_equal(bits[f], o["bits"])
        assert np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"])
    # f32: one teacher-forced turn from oracle states, both rules, <= 1e-5 (+ the oracle's own conditioning term for tanh)
    for variant in ("min", "tanh"):
        dec = hip.Decoder(code, variant, "f32", 64, path="fused")
        assert "fused_msg_kernel" in dec.kernel_name
        states = []
        for f in range(len(llr)):
            o = oracle.decode(g, variant, 30, llr[f], trace=True)
            ne = np.zeros(g.E)
            for n in range(min(o["iters"], 8)):
                states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
                ne = o["trace_ne"][n]
        states = states[:64]
        ne2, lam2, _ = dec.debug_step(np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), np.stack([s[2] for s in states]))
        worst = 0.0
        for i, s in enumerate(states):
            tol_lam, tol_ne = lam_tolerance(g, s[3], s[4])
            assert (np.abs(ne2[i] - s[3]) <= tol_ne).all(), (sz, variant, i)
            assert (np.abs(lam2[i] - s[4]) <= tol_lam).all(), (sz, variant, i)
            worst = max(worst, (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max())
        print(f"sz={sz} {variant} f32 table-driven: worst teacher-forced relative LLR error {worst:.2e} over {len(states)} turns")


@pytest.mark.parametrize("name", ["jpl.1024.4.5", "jpl.4096.4.5"])
def test_shipped_codes_on_the_table_driven_kernel(hip, name, monkeypatch):
    c = load(name)
    F = 32 if name == "jpl.1024.4.5" else 12
    _, llr = c.frames(F, 3.2, seed=91)
    code = c.hip_code(hip)
    for variant in ("min", "tanh"):
        monkeypatch.delenv("LDPC_FUSED_TABLE", raising=False)
        stat = hip.Decoder(code, variant, "f32", F, path="fused")
        monkeypatch.setenv("LDPC_FUSED_TABLE", "dyn")
        dyn = hip.Decoder(code, variant, "f32", F, path="fused")
        monkeypatch.delenv("LDPC_FUSED_TABLE", raising=False)
        assert "fused_msg_kernel" in dyn.kernel_name and dyn.kernel_name != stat.kernel_name
        b1, i1, c1 = stat.decode_batch(llr.astype(np.float32), 50)
        b2, i2, c2 = dyn.decode_batch(llr.astype(np.float32), 50)
        assert np.array_equal(b1, b2) and np.array_equal(i1, i2) and np.array_equal(c1, c2)
        ob, _, oc = oracle.decode_batch(c.graph, variant, 50, llr, nthreads=8)
        assert np.array_equal(b2, ob) and np.array_equal(c2, oc)
