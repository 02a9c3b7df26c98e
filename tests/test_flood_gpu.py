"""GPU: parity of the generic flood path (state in HBM, any H) with the CPU oracle, through the C ABI.

Bars (north_star): hard bits bit-exact; per-iteration LLRs within 1e-5 for fp32 (teacher-forced:
every turn starts from the oracle's state); the f64 parity mode must
reproduce the oracle's whole free-running trajectory -- bit-for-bit for min-sum (only +,-,min,*),
to 1e-12 for tanh (device libm vs glibc differ in the last ulp)."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import lam_tolerance, load, iters_agree

pytestmark = pytest.mark.gpu

CASES = [("moon.7.13", 20, (1.0, 3.0, 5.0)), ("jpl.1024.4.5", 50, (2.0, 3.0, 4.0)), ("1920.1280.3.303", 50, (1.0, 2.5, 4.0))]


def _frames(c, per_db, dbs, seed):
    ll = [c.frames(per_db, db, seed + i)[1] for i, db in enumerate(dbs)]
    return np.concatenate(ll)


@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_minsum_f64_trajectory_is_bit_exact(hip, name, iters, dbs):
    c = load(name)
    llr = _frames(c, 4, dbs, 100)
    dec = hip.Decoder(c.hip_code(hip), "min", "f64", len(llr), path="flood")
    bits, its, conv, trace = dec.decode_trace(llr, iters)
    for f in range(len(llr)):
        o = oracle.decode(c.graph, "min", iters, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"]
        assert np.array_equal(bits[f], o["bits"])
        assert np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"]), f"frame {f}"


@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_tanh_f64_trajectory(hip, name, iters, dbs):
    c = load(name)
    llr = _frames(c, 4, dbs, 200)
    dec = hip.Decoder(c.hip_code(hip), "tanh", "f64", len(llr), path="flood")
    bits, its, conv, trace = dec.decode_trace(llr, iters)
    for f in range(len(llr)):
        o = oracle.decode(c.graph, "tanh", iters, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"]
        assert np.array_equal(bits[f], o["bits"])
        # compare turn by turn while the oracle's own conditioning allows a tight bound
        ne_max = np.abs(o["trace_ne"]).max() if o["iters"] else 0.0
        tol = 1e-11 * (1 + np.exp(min(ne_max, 36)) * 2.0 ** -30)
        err = np.abs(trace[f, : o["iters"] + 1] - o["trace_lam"]) / np.maximum(1, np.abs(o["trace_lam"]))
        assert err.max() <= tol, (f, err.max(), tol)


@pytest.mark.parametrize("variant", ["min", "tanh"])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_teacher_forced_step_llr_within_1e5(hip, name, iters, dbs, variant, dtype):
    c = load(name)
    llr = _frames(c, 2, dbs, 300)
    F = len(llr)
    dec = hip.Decoder(c.hip_code(hip), variant, dtype, 64, path="flood")
    # gather the oracle's states at the top of every turn, for every frame
    states = []
    for f in range(F):
        o = oracle.decode(c.graph, variant, iters, llr[f], trace=True)
        ne = np.zeros(c.E)
        for n in range(o["iters"]):
            states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
            ne = o["trace_ne"][n]
    worst = 0.0
    for s0 in range(0, len(states), 64):
        chunk = states[s0:s0 + 64]
        orig = np.stack([s[0] for s in chunk]); lam = np.stack([s[1] for s in chunk]); ne = np.stack([s[2] for s in chunk])
        ne2, lam2, syn0 = dec.debug_step(orig, lam, ne)
        assert not syn0.any()  # the oracle only updates when the syndrome is non-zero
        for i, s in enumerate(chunk):
            tol_lam, tol_ne = lam_tolerance(c.graph, s[3], s[4])
            if variant == "min" and dtype == "f64":
                assert np.array_equal(ne2[i], s[3]) and np.array_equal(lam2[i], s[4])
            else:
                if variant == "min":  # no transcendental: no oracle-conditioning allowance
                    tol_lam = 1e-5 * np.maximum(1, np.abs(s[4])); tol_ne = 1e-5 * np.maximum(1, np.abs(s[3]))
                assert (np.abs(ne2[i] - s[3]) <= tol_ne).all(), (name, variant, dtype, np.abs(ne2[i] - s[3]).max())
                assert (np.abs(lam2[i] - s[4]) <= tol_lam).all(), (name, variant, dtype, np.abs(lam2[i] - s[4]).max())
                worst = max(worst, (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max())
    print(f"{name} {variant} {dtype}: worst teacher-forced relative LLR error {worst:.3e} over {len(states)} turns")


@pytest.mark.parametrize("variant", ["min", "tanh"])
@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_f32_free_running_hard_bits_match(hip, name, iters, dbs, variant):
    c = load(name)
    llr = _frames(c, 24, dbs, 400)
    dec = hip.Decoder(c.hip_code(hip), variant, "f32", len(llr), path="flood")
    bits, its, conv = dec.decode_batch(llr.astype(np.float32), iters)
    obits, oits, oconv = oracle.decode_batch(c.graph, variant, iters, llr, nthreads=8)
    assert np.array_equal(bits, obits)
    assert np.array_equal(conv, oconv)
    # iteration counts may move by one turn on knife-edge frames (helpers.iters_agree has the measured rate)
    same = (its == oits).mean()
    print(f"{name} {variant}: {same * 100:.1f}% identical iteration counts, {conv.mean() * 100:.0f}% converged")
    assert iters_agree(its, oits)


def test_edge_cases(hip):
    c = load("jpl.1024.4.5")
    code = c.hip_code(hip)
    dec = hip.Decoder(code, "min", "f32", 130, path="flood")
    cws, llr = c.frames(130, 4.0, seed=7)  # ragged: 130 = 2 slabs + 2
    bits, its, conv = dec.decode_batch(llr.astype(np.float32), 50)
    obits, oits, oconv = oracle.decode_batch(c.graph, "min", 50, llr, nthreads=8)
    assert np.array_equal(bits, obits) and np.array_equal(conv, oconv)
    # batch of one through the per-frame entry point
    b1, it1, cv1 = dec.decode_one(llr[0], 50)
    assert np.array_equal(b1, obits[0]) and cv1 == bool(oconv[0])
    # empty batch is a no-op
    e = dec.decode_batch(np.zeros((0, c.N), np.float32), 50)
    assert e[0].shape == (0, c.N)
    # noiseless codeword and all-zero LLR: zero iterations
    z = np.concatenate([(2.0 * cws[:1] - 1.0) * 8.0, np.zeros((1, c.N))]).astype(np.float32)
    bits, its, conv = dec.decode_batch(z, 50)
    assert its.tolist() == [0, 0] and conv.all() and np.array_equal(bits[0], cws[0]) and not bits[1].any()
    # max_iters = 0: channel hard decisions unless already a codeword
    bits, its, conv = dec.decode_batch(llr[:3].astype(np.float32), 0)
    assert np.array_equal(bits, (llr[:3] > 0).astype(np.uint8)) and not conv.any()
    # batch larger than max_batch is rejected, not truncated
    with pytest.raises(hip.LdpcError):
        dec.decode_batch(np.zeros((131, c.N), np.float32), 1)


def test_csr_and_qc_graphs_decode_identically(hip):
    c = load("jpl.1024.4.5")
    _, llr = c.frames(16, 3.0, seed=17)
    a = hip.Decoder(c.hip_code(hip, prefer_qc=True), "tanh", "f32", 16, path="flood").decode_batch(llr.astype(np.float32), 30)
    b = hip.Decoder(c.hip_code(hip, prefer_qc=False), "tanh", "f32", 16, path="flood").decode_batch(llr.astype(np.float32), 30)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_minsum_degree_one_rejected(hip):
    code = hip.Code.from_dense(np.array([[1, 0, 0], [1, 1, 1]], np.uint8))
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(code, "min", "f32", 1, path="flood")
    assert e.value.code == -6
    bits, it, cv = hip.Decoder(code, "tanh", "f64", 1, path="flood").decode_one(np.array([1.0, -2.0, 3.0]), 3)
    o = oracle.decode(oracle.Graph.from_dense(np.array([[1, 0, 0], [1, 1, 1]], np.uint8)), "tanh", 3, np.array([1.0, -2.0, 3.0]))
    assert np.array_equal(bits, o["bits"]) and it == o["iters"]


@pytest.mark.parametrize("name", ["jpl.1024.4.5", "jpl.4096.4.5"])
def test_frame_major_and_batch_major_flood_kernels_agree(hip, name, monkeypatch):
    """QC codes on the flood path run one workgroup per frame (layered_qc.hip flood_qc_kernel); LDPC_FLOOD_QC=0 keeps the
    batch-major kernel pair (any H).  Same arithmetic, same orders: identical f32 results, f64 trajectories bit for bit."""
    c = load(name)
    F = 70 if name == "jpl.1024.4.5" else 20
    _, llr = c.frames(F, 3.0, seed=321)
    code = c.hip_code(hip)
    for variant in ("min", "tanh"):
        qc = hip.Decoder(code, variant, "f32", F, path="flood")
        monkeypatch.setenv("LDPC_FLOOD_QC", "0")
        bm = hip.Decoder(code, variant, "f32", F, path="flood")
        bm64 = hip.Decoder(code, variant, "f64", 3, path="flood")
        monkeypatch.delenv("LDPC_FLOOD_QC")
        assert "flood_qc_kernel" in qc.kernel_name and "flood_cn" in bm.kernel_name
        a, b = qc.decode_batch(llr.astype(np.float32), 40), bm.decode_batch(llr.astype(np.float32), 40)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (name, variant)
        if variant == "min":
            ta = hip.Decoder(code, variant, "f64", 3, path="flood").decode_trace(llr[:3], 20)
            tb = bm64.decode_trace(llr[:3], 20)
            assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
            for f in range(3):
                o = oracle.decode(c.graph, "min", 20, llr[f], trace=True)
                assert np.array_equal(ta[3][f, : o["iters"] + 1], o["trace_lam"])
