"""GPU: the generic on-chip kernel (fused_csr.hip: any H that fits in LDS) against the oracle and the
flood path, through the C ABI.  Same bars as the other paths."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import lam_tolerance, load, iters_agree

pytestmark = pytest.mark.gpu

CASES = [("moon.7.13", 20, (1.0, 3.0, 5.0)), ("1920.1280.3.303", 50, (1.0, 2.0, 3.0)), ("jpl.1024.4.5", 50, (2.0, 3.0, 4.0))]


def _frames(c, per_db, dbs, seed):
    return np.concatenate([c.frames(per_db, db, seed + i)[1] for i, db in enumerate(dbs)])


def _code(hip, c):
    return c.hip_code(hip, prefer_qc=False)  # plain CSR graph: no QC plan -> generic kernel


@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_auto_path(hip, name, iters, dbs):
    c = load(name)
    for variant in ("min", "tanh"):
        assert hip.Decoder(_code(hip, c), variant, "f32", 8).path == "fused"
    assert hip.Decoder(_code(hip, c), "min", "f16", 8).path == "fused"   # fp16 LLRs, state on-chip in f32
    big = load("jpl.4096.4.5")  # as a CSR graph: 2N + 20M floats = 168 KB > 160 KB of LDS
    assert hip.Decoder(_code(hip, big), "min", "f32", 8).path == "flood"
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(_code(hip, big), "min", "f32", 8, path="fused")
    assert e.value.code == -5 and "LDS" in str(e.value)


@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_minsum_f64_trajectory_is_bit_exact(hip, name, iters, dbs):
    c = load(name)
    llr = _frames(c, 4, dbs, 1100)
    dec = hip.Decoder(_code(hip, c), "min", "f64", len(llr), path="fused")
    bits, its, conv, trace = dec.decode_trace(llr, iters)
    for f in range(len(llr)):
        o = oracle.decode(c.graph, "min", iters, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"] and np.array_equal(bits[f], o["bits"])
        assert np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"]), f


@pytest.mark.parametrize("variant", ["min", "tanh"])
@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_f32_equals_flood_and_oracle_bits(hip, name, iters, dbs, variant):
    c = load(name)
    llr = _frames(c, 24, dbs, 1200)
    code = _code(hip, c)
    a = hip.Decoder(code, variant, "f32", len(llr), path="fused").decode_batch(llr.astype(np.float32), iters)
    b = hip.Decoder(code, variant, "f32", len(llr), path="flood").decode_batch(llr.astype(np.float32), iters)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))  # same arithmetic, same order
    ob, oi, oc = oracle.decode_batch(c.graph, variant, iters, llr, nthreads=8)
    assert np.array_equal(a[0], ob) and np.array_equal(a[2], oc) and iters_agree(a[1], oi)


@pytest.mark.parametrize("variant,dtype", [("min", "f32"), ("tanh", "f32"), ("tanh", "f64")])
@pytest.mark.parametrize("name,iters,dbs", CASES[:2])
def test_teacher_forced_step(hip, name, iters, dbs, variant, dtype):
    c = load(name)
    llr = _frames(c, 2, dbs, 1300)
    dec = hip.Decoder(_code(hip, c), variant, dtype, 64, path="fused")
    states = []
    for f in range(len(llr)):
        o = oracle.decode(c.graph, variant, iters, llr[f], trace=True)
        ne = np.zeros(c.E)
        for n in range(o["iters"]):
            states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
            ne = o["trace_ne"][n]
    states = states[:128]
    worst = 0.0
    for s0 in range(0, len(states), 64):
        ch = states[s0:s0 + 64]
        ne2, lam2, syn0 = dec.debug_step(np.stack([s[0] for s in ch]), np.stack([s[1] for s in ch]), np.stack([s[2] for s in ch]))
        assert not syn0.any()
        for i, s in enumerate(ch):
            tol_lam, tol_ne = lam_tolerance(c.graph, s[3], s[4], rel=1e-5 if dtype == "f32" else 1e-12)
            assert (np.abs(ne2[i] - s[3]) <= tol_ne).all() and (np.abs(lam2[i] - s[4]) <= tol_lam).all()
            worst = max(worst, (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max())
    print(f"{name} fused-csr {variant} {dtype}: worst teacher-forced relative LLR error {worst:.3e} over {len(states)} turns")


def test_edge_cases(hip):
    c = load("1920.1280.3.303")
    dec = hip.Decoder(_code(hip, c), "tanh", "f32", 70, path="fused")
    _, llr = c.frames(70, 2.0, seed=1400)
    bits, its, conv = dec.decode_batch(llr.astype(np.float32), 50)
    ob, oi, oc = oracle.decode_batch(c.graph, "tanh", 50, llr, nthreads=8)
    assert np.array_equal(bits, ob) and np.array_equal(conv, oc)
    z = np.zeros((2, c.N), np.float32)
    z[1] = -3.0  # noiseless all-zero codeword
    bits, its, conv = dec.decode_batch(z, 50)
    assert its.tolist() == [0, 0] and conv.all() and not bits.any()
    bits, its, conv = dec.decode_batch(llr[:3].astype(np.float32), 0)
    assert np.array_equal(bits, (llr[:3] > 0).astype(np.uint8))
    b1, it1, cv1 = dec.decode_one(llr[0], 50)
    assert np.array_equal(b1, ob[0]) and it1 == oi[0]


@pytest.mark.parametrize("variant", ["min", "tanh"])
@pytest.mark.parametrize("name,iters,dbs", CASES)
def test_batched_kernel_equals_row_by_row_kernel(hip, monkeypatch, name, iters, dbs, variant):
    """fused_csr_batched_kernel (indices, messages and channel LLRs in registers; all LDS gathers of a phase issued
    together) and the row-by-row fused_csr_kernel (LDPC_CSR_BATCHED=0) run the same arithmetic in the same order:
    bits, iteration counts, flags and returned LLRs must be identical."""
    c = load(name)
    llr = _frames(c, 40, dbs, 1900).astype(np.float32)
    code = _code(hip, c)
    a = hip.Decoder(code, variant, "f32", len(llr), path="fused")
    # (moon.7.13 has columns of weight > 8: no batched instance, both runs use the row-by-row kernel)
    assert a.kernel_name == ("fused_csr_kernel" if name == "moon.7.13" else "fused_csr_batched_kernel")
    ra = a.decode_batch(llr.astype(np.float64), iters, want_lam=True)
    monkeypatch.setenv("LDPC_CSR_BATCHED", "0")
    b = hip.Decoder(code, variant, "f32", len(llr), path="fused")
    assert b.kernel_name == "fused_csr_kernel"
    rb = b.decode_batch(llr.astype(np.float64), iters, want_lam=True)
    assert len(set(ra[1].tolist())) > 2
    assert all(np.array_equal(x, y) for x, y in zip(ra, rb))


def test_placement_changes_nothing_but_speed(hip, monkeypatch):
    """The batched kernel stores rows and columns at conflict-aware LDS positions (LDPC_CSR_PLACE=0: file order);
    positions never enter the arithmetic, so every output is identical, traces and teacher-forced steps included."""
    c = load("1920.1280.3.303")
    llr = _frames(c, 12, (1.0, 2.0), 2100)
    code = _code(hip, c)
    rng = np.random.default_rng(9)
    orig = rng.normal(0, 4, (4, c.N)); lam = orig + rng.normal(0, 2, (4, c.N)); ne = rng.normal(0, 1, (4, c.E))
    outs = []
    for place in ("1", "0"):
        monkeypatch.setenv("LDPC_CSR_PLACE", place)
        d = hip.Decoder(code, "tanh", "f32", len(llr), path="fused")
        outs.append((d.decode_batch(llr, 50, want_lam=True), d.decode_trace(llr[:6], 50), d.debug_step(orig, lam, ne)))
    for x, y in zip(outs[0], outs[1]):
        assert all(np.array_equal(p, q) for p, q in zip(x, y))


def test_wide_workgroups_change_nothing_but_speed(hip, monkeypatch):
    """1920.1280.3.303 runs the batched kernel with 512 threads per frame (3 rows / 4 columns per thread);
    LDPC_CSR_WIDE=0 keeps 256 threads.  Same arithmetic, same order: identical outputs."""
    c = load("1920.1280.3.303")
    llr = _frames(c, 20, (1.0, 2.0), 2300).astype(np.float32)
    code = _code(hip, c)
    outs = []
    for wide in ("1", "0"):
        monkeypatch.setenv("LDPC_CSR_WIDE", wide)
        for variant in ("min", "tanh"):
            outs.append(hip.Decoder(code, variant, "f32", len(llr), path="fused").decode_batch(llr.astype(np.float64), 50, want_lam=True))
    for x, y in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert all(np.array_equal(p, q) for p, q in zip(x, y))


def test_tiny_circulant_qc_tanh_paths_agree(hip):
    """a QC-described code the split family does not take (circulant size 8) with rows of weight <= 4: its on-chip path is the generic
    kernel, whose weight-4 instance evaluates the tanh rule by pair products -- the flood path of the same code must use that form too
    (one predicate, api.cc): f32 tanh results identical bit for bit, LLRs included"""
    from tests.helpers import SyntheticQC
    rng = np.random.default_rng(31)
    mask = np.zeros((6, 12), bool)
    for br in range(6):
        mask[br, rng.choice(12, 4 if br % 2 else 3, replace=False)] = True
    for bc in range(12):
        if not mask[:, bc].any():
            mask[rng.integers(0, 6), bc] = True
    mask[mask.sum(1) > 4] = False                       # (keep every row at weight <= 4)
    for br in range(6):
        while mask[br].sum() < 2:
            mask[br, rng.integers(0, 12)] = True
    off = np.where(mask, rng.integers(0, 8, mask.shape), -1).astype(np.int32)
    c = SyntheticQC("tiny-6x12-sz8", 8, off, rate=(48, 96))
    assert c.graph.row_ptr[1:].max() - 0 > 0 and np.diff(c.graph.row_ptr).max() <= 4
    llr = np.concatenate([c.frames(40, db, 3100 + i)[1] for i, db in enumerate((1.0, 4.0))]).astype(np.float32)
    fu = hip.Decoder(c.hip_code(hip), "tanh", "f32", len(llr), path="fused")
    fl = hip.Decoder(c.hip_code(hip), "tanh", "f32", len(llr), path="flood")
    assert "fused_csr" in fu.kernel_name and "flood_qc_kernel" not in fl.kernel_name, (fu.kernel_name, fl.kernel_name)
    a, b = fu.decode_batch(llr, 30, want_lam=True), fl.decode_batch(llr, 30, want_lam=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(set(a[1].tolist())) > 2
    ob, oi, oc = oracle.decode_batch(c.graph, "tanh", 30, llr.astype(np.float64), nthreads=4)
    assert np.array_equal(a[0], ob) and np.array_equal(a[2].astype(bool), oc.astype(bool))


@pytest.mark.parametrize("variant", ["min", "tanh"])
def test_staged_instance_changes_nothing_but_speed(hip, monkeypatch, variant):
    """f32 LLRs of a plain decode run the STAGED instance of the batched kernel: a workgroup holds the frame it decodes and the one
    it decodes next, whose LLRs travel into LDS by LDS-DMA meanwhile; over the last ids of a batch nothing is held ahead.  Which
    frame a workgroup takes when never enters the arithmetic: for batches below, at, just above and far above the number of resident
    workgroups (768 on an MI355X for this code) every output equals that of the other instance (LDPC_CSR_STAGE=0), frame by frame,
    and the oracle's on the frames compared with it."""
    c = load("1920.1280.3.303")
    code = _code(hip, c)
    base = _frames(c, 20, (1.0, 2.5, 4.0), 4100).astype(np.float32)   # 60 frames
    ob, oi, oc = oracle.decode_batch(c.graph, variant, 30, base[:24].astype(np.float64), nthreads=8)
    for batch in (1, 7, 767, 768, 769, 1543, 3 * 768 + 5, 6000):
        llr = np.tile(base, ((batch + len(base) - 1) // len(base), 1))[:batch]
        outs = {}
        for stage in ("1", "0"):
            monkeypatch.setenv("LDPC_CSR_STAGE", stage)
            d = hip.Decoder(code, variant, "f32", batch, path="fused")
            outs[stage] = d.decode_batch(llr, 30)    # (float32 in: the host path that keeps the LLRs' format)
            assert d.kernel_name.endswith(", 512, 0, true>" if stage == "1" else ", 512, 0, false>"), d.kernel_name
            outs[stage + "again"] = d.decode_batch(llr, 30)      # the work counter starts over with every launch
            d.close()
        if batch == 1543:   # the other ways frames are handed out: a fixed stride instead of the work counter, one workgroup per frame
            monkeypatch.setenv("LDPC_CSR_STAGE", "1")
            for var in ("LDPC_CSR_DYNAMIC", "LDPC_CSR_PERSIST"):
                monkeypatch.setenv(var, "0")
                d = hip.Decoder(code, variant, "f32", batch, path="fused")
                outs[var] = d.decode_batch(llr, 30)
                assert d.kernel_name.endswith(", 512, 0, true>")
                d.close()
                monkeypatch.delenv(var)
        for k in outs:
            assert all(np.array_equal(x, y) for x, y in zip(outs["1"], outs[k])), (batch, k)
        n = min(batch, 24)
        bits, its, conv = outs["1"]
        assert np.array_equal(bits[:n], ob[:n]) and np.array_equal(conv[:n].astype(bool), oc[:n].astype(bool))
        if batch > 60:   # the tiled frames repeat: so do their results, wherever in the batch and on whichever workgroup they ran
            assert np.array_equal(bits[60:120], bits[:60][: len(bits[60:120])]) and np.array_equal(its[60:120], its[:60][: len(its[60:120])])
