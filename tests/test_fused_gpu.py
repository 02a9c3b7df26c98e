"""GPU: parity of the fused on-chip min-sum kernel (fused.hip) with the CPU oracle, through the C ABI.
Same bars as the flood path: f64 reproduces the oracle trajectory bit-for-bit; f32 agrees within
1e-5 per teacher-forced turn and gives identical hard bits free-running."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import load, iters_agree

pytestmark = pytest.mark.gpu

CODES = [("jpl.1024.4.5", (2.0, 3.0, 4.0)), ("jpl.4096.4.5", (2.0, 3.0, 3.6))]


def _frames(c, per_db, dbs, seed):
    return np.concatenate([c.frames(per_db, db, seed + i)[1] for i, db in enumerate(dbs)])


@pytest.mark.parametrize("name,dbs", CODES)
def test_auto_path_is_fused_for_ar4ja_minsum(hip, name, dbs):
    c = load(name)
    assert hip.Decoder(c.hip_code(hip), "min", "f32", 8).path == "fused"
    assert hip.Decoder(c.hip_code(hip), "tanh", "f32", 8).path == "fused"
    assert hip.Decoder(c.hip_code(hip), "tanh", "f64", 8).path == "flood"
    # a CSR graph has no QC table: generic on-chip kernel if a frame fits in LDS (jpl.1024), else flood (jpl.4096)
    assert hip.Decoder(c.hip_code(hip, prefer_qc=False), "min", "f32", 8).path == ("fused" if name == "jpl.1024.4.5" else "flood")
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(c.hip_code(hip), "tanh", "f64", 8, path="fused")
    assert e.value.code == -5


@pytest.mark.parametrize("name,dbs", CODES)
def test_f64_trajectory_is_bit_exact(hip, name, dbs):
    c = load(name)
    llr = _frames(c, 3, dbs, 500)
    dec = hip.Decoder(c.hip_code(hip), "min", "f64", len(llr), path="fused")
    bits, its, conv, trace = dec.decode_trace(llr, 50)
    for f in range(len(llr)):
        o = oracle.decode(c.graph, "min", 50, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"], f
        assert np.array_equal(bits[f], o["bits"])
        assert np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"]), f"frame {f}"
    b2, i2, c2, lam = dec.decode_batch(llr, 50, want_lam=True)
    assert np.array_equal(b2, bits) and np.array_equal(i2, its)
    for f in range(len(llr)):
        assert np.array_equal(lam[f], oracle.decode(c.graph, "min", 50, llr[f])["lam"])


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name,dbs", CODES)
def test_teacher_forced_step(hip, name, dbs, dtype):
    c = load(name)
    llr = _frames(c, 2, dbs, 600)
    dec = hip.Decoder(c.hip_code(hip), "min", dtype, 64, path="fused")
    states = []
    for f in range(len(llr)):
        o = oracle.decode(c.graph, "min", 50, llr[f], trace=True)
        ne = np.zeros(c.E)
        for n in range(o["iters"]):
            states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
            ne = o["trace_ne"][n]
    states = states[::3][:192]
    worst = 0.0
    for s0 in range(0, len(states), 64):
        ch = states[s0:s0 + 64]
        ne2, lam2, syn0 = dec.debug_step(np.stack([s[0] for s in ch]), np.stack([s[1] for s in ch]), np.stack([s[2] for s in ch]))
        assert not syn0.any()
        for i, s in enumerate(ch):
            if dtype == "f64":
                assert np.array_equal(ne2[i], s[3]) and np.array_equal(lam2[i], s[4])
            else:
                assert (np.abs(ne2[i] - s[3]) <= 1e-5 * np.maximum(1, np.abs(s[3]))).all()
                err = np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))
                assert err.max() <= 1e-5, err.max()
                worst = max(worst, err.max())
    print(f"{name} fused min {dtype}: worst teacher-forced relative LLR error {worst:.3e} over {len(states)} turns")


@pytest.mark.parametrize("name,dbs", CODES)
def test_f32_free_running_matches_oracle_and_flood(hip, name, dbs):
    c = load(name)
    per = 40 if name == "jpl.1024.4.5" else 12
    llr = _frames(c, per, dbs, 700)
    code = c.hip_code(hip)
    fused = hip.Decoder(code, "min", "f32", len(llr), path="fused")
    flood = hip.Decoder(code, "min", "f32", len(llr), path="flood")
    b1, i1, c1 = fused.decode_batch(llr.astype(np.float32), 50)
    b2, i2, c2 = flood.decode_batch(llr.astype(np.float32), 50)
    # same arithmetic, same summation order: the two HIP paths must agree exactly
    assert np.array_equal(b1, b2) and np.array_equal(i1, i2) and np.array_equal(c1, c2)
    ob, oi, oc = oracle.decode_batch(c.graph, "min", 50, llr, nthreads=8)
    assert np.array_equal(b1, ob) and np.array_equal(c1, oc)
    assert iters_agree(i1, oi)
    print(f"{name}: fused == flood exactly; vs oracle {100 * (i1 == oi).mean():.1f}% identical iteration counts, {c1.mean() * 100:.0f}% converged")


def test_fused_edge_cases(hip):
    c = load("jpl.1024.4.5")  # two frames per wave: ragged tail must be handled per half-wave
    dec = hip.Decoder(c.hip_code(hip), "min", "f32", 131, path="fused")
    cws, llr = c.frames(131, 4.0, seed=77)
    bits, its, conv = dec.decode_batch(llr.astype(np.float32), 50)
    ob, oi, oc = oracle.decode_batch(c.graph, "min", 50, llr, nthreads=8)
    assert np.array_equal(bits, ob) and np.array_equal(conv, oc)
    b1, it1, cv1 = dec.decode_one(llr[5], 50)
    assert np.array_equal(b1, ob[5]) and it1 == oi[5]
    z = np.concatenate([(2.0 * cws[:1] - 1.0) * 8.0, np.zeros((1, c.N)), llr[:1]]).astype(np.float32)
    bits, its, conv = dec.decode_batch(z, 50)
    assert its[:2].tolist() == [0, 0] and conv[:2].all() and np.array_equal(bits[0], cws[0]) and not bits[1].any()
    assert np.array_equal(bits[2], ob[0])  # a converged neighbour in the same wave does not disturb the other frame
    bits, its, conv = dec.decode_batch(llr[:3].astype(np.float32), 0)
    assert np.array_equal(bits, (llr[:3] > 0).astype(np.uint8)) and not conv.any()
    c4 = load("jpl.4096.4.5")
    dec4 = hip.Decoder(c4.hip_code(hip), "min", "f32", 5, path="fused")
    cws4, llr4 = c4.frames(5, 3.8, seed=78)
    bits, its, conv = dec4.decode_batch(llr4.astype(np.float32), 50)
    ob, oi, oc = oracle.decode_batch(c4.graph, "min", 50, llr4, nthreads=5)
    assert np.array_equal(bits, ob) and np.array_equal(conv, oc)


def test_full_size_round_trip_property(hip):
    """BASELINE size (jpl.4096, thousands of frames) is beyond the oracle's reach in seconds; use
    size-independent properties: at high SNR every frame must decode to its own codeword (encode ->
    noise -> decode round trip), every output is a codeword (syndrome zero), and the fused and flood
    paths agree bit-for-bit."""
    import os
    from tests.helpers import ROOT
    ecc = hip.ECC(os.path.join(ROOT, "codes"), "ldpc/hip-minsum/jpl.4096.4.5/50/4/5", max_batch=2048)
    assert ecc.decoder.path == "fused"
    import torch
    dev = torch.device("cuda", 0)
    B, N, k = 2048, ecc.code.N, ecc.message_length
    llr = torch.empty((B, N), dtype=torch.float32, device=dev)
    msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
    bits = torch.empty((B, N), dtype=torch.uint8, device=dev)
    it = torch.empty((B,), dtype=torch.int32, device=dev)
    cv = torch.empty((B,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ecc.sim.generate(1234, 0, B, 4.5, llr.data_ptr(), msg.data_ptr(), None)
    torch.cuda.synchronize()
    ecc.decoder.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, 50, it.data_ptr(), cv.data_ptr(), None)
    ecc.decoder.synchronize()
    assert bool(cv.all()) and bool((bits[:, :k] == msg).all())
    H = torch.tensor(load("jpl.4096.4.5").H, dtype=torch.float32, device=dev)
    syn = (bits.float() @ H.T) % 2
    assert float(syn.abs().sum()) == 0.0
    # host encoder agrees with the device encoder (codeword = msg ++ parity)
    cw = ecc.encode(msg[0].cpu().numpy())
    assert np.array_equal(cw, bits[0, : ecc.codeword_length].cpu().numpy())
    flood = hip.Decoder(ecc.code, "min", "f32", B, path="flood")
    bits2 = torch.empty_like(bits)
    # marginal SNR: mixed converged / failed frames must still agree between the two paths
    ecc.sim.generate(99, 0, B, 2.9, llr.data_ptr(), msg.data_ptr(), None)
    torch.cuda.synchronize()
    ecc.decoder.decode_batch_dev(llr.data_ptr(), bits.data_ptr(), B, 50, it.data_ptr(), cv.data_ptr(), None)
    ecc.decoder.synchronize()
    it2 = torch.empty_like(it); cv2 = torch.empty_like(cv)
    flood.decode_batch_dev(llr.data_ptr(), bits2.data_ptr(), B, 50, it2.data_ptr(), cv2.data_ptr(), None)
    flood.synchronize()
    assert bool((bits == bits2).all()) and bool((it == it2).all()) and bool((cv == cv2).all())
    frac = float(cv.float().mean())
    assert 0.02 < frac < 0.999, frac  # really a mixed batch


@pytest.mark.parametrize("name,dbs", CODES)
def test_tanh_f32_fused(hip, name, dbs):
    """Fused tanh rule (fp32 hyperbolic-recurrence form): teacher-forced LLRs within the 1e-5 bar (+ the oracle's own
    conditioning allowance next to the clamp), free-running bits identical to the oracle, and the
    fused and flood paths identical to each other (same arithmetic, same summation order)."""
    from tests.helpers import lam_tolerance
    c = load(name)
    per = 24 if name == "jpl.1024.4.5" else 8
    llr = _frames(c, per, dbs, 800)
    code = c.hip_code(hip)
    fused = hip.Decoder(code, "tanh", "f32", max(len(llr), 64), path="fused")
    flood = hip.Decoder(code, "tanh", "f32", len(llr), path="flood")
    b1, i1, c1 = fused.decode_batch(llr.astype(np.float32), 50)
    b2, i2, c2 = flood.decode_batch(llr.astype(np.float32), 50)
    assert np.array_equal(b1, b2) and np.array_equal(i1, i2) and np.array_equal(c1, c2)
    ob, oi, oc = oracle.decode_batch(c.graph, "tanh", 50, llr, nthreads=8)
    assert np.array_equal(b1, ob) and np.array_equal(c1, oc) and iters_agree(i1, oi)
    states = []
    for f in range(0, len(llr), max(1, len(llr) // 6)):
        o = oracle.decode(c.graph, "tanh", 50, llr[f], trace=True)
        ne = np.zeros(c.E)
        for n in range(o["iters"]):
            states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
            ne = o["trace_ne"][n]
    states = states[::2][:128]
    worst = 0.0
    for s0 in range(0, len(states), 64):
        ch = states[s0:s0 + 64]
        ne2, lam2, syn0 = fused.debug_step(np.stack([s[0] for s in ch]), np.stack([s[1] for s in ch]), np.stack([s[2] for s in ch]))
        assert not syn0.any()
        for i, s in enumerate(ch):
            tol_lam, tol_ne = lam_tolerance(c.graph, s[3], s[4])
            assert (np.abs(ne2[i] - s[3]) <= tol_ne).all() and (np.abs(lam2[i] - s[4]) <= tol_lam).all()
            worst = max(worst, (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max())
    print(f"{name} fused tanh f32: worst teacher-forced relative LLR error {worst:.3e} over {len(states)} turns; {c1.mean() * 100:.0f}% converged")


def test_concurrent_contexts_on_separate_streams(hip):
    """Several decoder replicas (the reference's maxThreadCount > 1, Utils.hs:53) working at once on their
    own streams must not disturb each other: fused kernels keep no per-context device state, the flood path
    keeps it per context."""
    import torch
    c = load("jpl.1024.4.5")
    dev = torch.device("cuda", 0)
    code = c.hip_code(hip)
    B = 512
    jobs = []
    for i, (variant, path) in enumerate([("min", "fused"), ("tanh", "fused"), ("min", "flood"), ("min", "fused")]):
        _, llr = c.frames(B, 2.5 + 0.5 * i, seed=2000 + i)
        x = torch.tensor(llr, dtype=torch.float32, device=dev)
        jobs.append(dict(dec=hip.Decoder(code, variant, "f32", B, path=path), x=x, llr=llr, variant=variant,
                         bits=torch.empty((B, c.N), dtype=torch.uint8, device=dev), it=torch.empty(B, dtype=torch.int32, device=dev),
                         cv=torch.empty(B, dtype=torch.uint8, device=dev), stream=torch.cuda.Stream(device=dev)))
    torch.cuda.synchronize()
    for rep in range(3):  # interleave launches of all contexts
        for j in jobs:
            j["dec"].decode_batch_dev(j["x"].data_ptr(), j["bits"].data_ptr(), B, 50, j["it"].data_ptr(), j["cv"].data_ptr(), j["stream"].cuda_stream)
    torch.cuda.synchronize()
    for j in jobs:
        ref = j["dec"].decode_batch(j["llr"].astype(np.float32), 50)  # same context, alone, host path
        assert np.array_equal(j["bits"].cpu().numpy(), ref[0]) and np.array_equal(j["it"].cpu().numpy(), ref[1])


@pytest.mark.parametrize("name,variant", [("jpl.1024.4.5", "min"), ("jpl.1024.4.5", "tanh"), ("jpl.4096.4.5", "min")])
def test_split_kernel_matches_two_wave_kernel(hip, monkeypatch, name, variant):
    """fused_split.hip (block rows split between wave pairs; for sz = 32 two frames share a workgroup and finish
    at different iterations) must give the same bits, iteration counts and flags as fused_msg.hip -- including an
    odd batch, whose last workgroup holds one real frame and one shadow lane set."""
    c = load(name)
    B = 2049 if name == "jpl.1024.4.5" else 257
    _, llr = c.frames(B, 3.0, seed=4242)      # mixed: frames converge after 3..50 turns, some never
    llr = llr.astype(np.float32)
    code = c.hip_code(hip)
    a = hip.Decoder(code, variant, "f32", B, path="fused").decode_batch(llr, 50)
    monkeypatch.setenv("LDPC_FUSED_KERNEL", "msg")
    b = hip.Decoder(code, variant, "f32", B, path="fused").decode_batch(llr, 50)
    assert len(set(a[1].tolist())) > 3        # the batch really mixes early and late finishers
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    monkeypatch.delenv("LDPC_FUSED_KERNEL")
    a2 = hip.Decoder(code, variant, "f32", B, path="fused").decode_batch(llr, 50)   # and is repeatable
    assert all(np.array_equal(x, y) for x, y in zip(a, a2))


def test_many_iterations_leave_the_split_kernel(hip):
    """The split kernel reports the turn a frame converged at in 9 bits; beyond 511 turns the two-wave kernel runs.
    A frame far below the waterfall must come back with iters == max_iters either way, identical to the flood path."""
    c = load("jpl.1024.4.5")
    _, llr = c.frames(16, 1.0, seed=77)
    llr = llr.astype(np.float32)
    code = c.hip_code(hip)
    for iters in (511, 600):
        a = hip.Decoder(code, "min", "f32", len(llr), path="fused").decode_batch(llr, iters)
        b = hip.Decoder(code, "min", "f32", len(llr), path="flood").decode_batch(llr, iters)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) and int(a[1].max()) == iters


@pytest.mark.parametrize("name,qc,path", [("jpl.1024.4.5", True, "fused"), ("1920.1280.3.303", False, "fused"), ("jpl.1024.4.5", True, "flood")])
def test_zero_copy_host_path_equals_staged_path(hip, name, qc, path):
    """With page-locked llr and bits buffers (ldpc_host_alloc) the decode kernel reads the LLRs and writes the bits
    over PCIe itself; pageable buffers go through the chunked copy pipeline.  Same answers either way, for f32 and
    fp16 LLR buffers, with the iteration counts returned into pageable or page-locked memory."""
    c = load(name)
    B = 600
    _, llr = c.frames(B, 3.0 if qc else 1.5, seed=606)
    dec = hip.Decoder(c.hip_code(hip, prefer_qc=qc), "min", "f32", B, path=path)
    for dt in (np.float32, np.float16):
        src = np.clip(llr, -6e4, 6e4).astype(dt)
        want = dec.decode_batch(src, 50)                       # pageable -> staged copies
        pin_in = hip.PinnedArray((B, c.N), dt); pin_in.array[:] = src
        pin_out = hip.PinnedArray((B, c.N), np.uint8); pin_out.array[:] = 7
        got = dec.decode_batch(pin_in.array, 50, out_bits=pin_out.array)   # page-locked -> zero-copy
        assert got[0] is pin_out.array and all(np.array_equal(x, y) for x, y in zip(got, want))
        assert 0 < int(want[2].sum()) < B                      # a mixed batch


def test_page_locked_input_on_the_hbm_paths(hip):
    """flood_qc_kernel re-reads the channel LLRs every turn, so a page-locked input must NOT be handed to it as
    a zero-copy device pointer (it is staged once instead); the layered kernel and the batch-major flood kernels read the
    input once and do take it zero-copy.  Same answers as from pageable memory in all three cases, batch > 16."""
    import os
    c = load("jpl.1024.4.5")
    B = 200
    _, llr = c.frames(B, 3.0, seed=77)
    src = llr.astype(np.float32)
    pin_in = hip.PinnedArray((B, c.N), np.float32); pin_in.array[:] = src
    pin_out = hip.PinnedArray((B, c.N), np.uint8)
    for kw, env in (({"path": "flood"}, {}), ({"path": "flood", "schedule": "layered"}, {}), ({"path": "flood"}, {"LDPC_FLOOD_QC": "0"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            dec = hip.Decoder(c.hip_code(hip), "min", "f32", B, **kw)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        want = dec.decode_batch(src, 50)
        pin_out.array[:] = 9
        got = dec.decode_batch(pin_in.array, 50, out_bits=pin_out.array)
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), (kw, env, dec.kernel_name)
        assert 0 < int(want[2].sum()) < B
        dec.close()
