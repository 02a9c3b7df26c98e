"""GPU: the row-layered schedule (BASELINE.json configs[4]; an extension -- the reference has flooding only) against its
specification, oracle_decode_layered.  Same bars as the flooding paths: f64 min-sum reproduces the oracle's trajectory
bit for bit, f64 tanh to the device libm's last ulps, f32 gives identical hard bits / flags and per-sweep LLRs within
1e-5 (teacher-forced)."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import CODES, lam_tolerance, load, synthetic

pytestmark = pytest.mark.gpu


def _layers(c):
    return c.layer_ptr if hasattr(c, "layer_ptr") else (np.arange(0, c.M + 1, c.sz) if getattr(c, "sz", 0) else np.arange(c.M + 1))


def _get(name):
    return load(name) if name in ("moon.7.13", "jpl.1024.4.5", "jpl.4096.4.5", "1920.1280.3.303") else synthetic(name)


@pytest.mark.parametrize("name,db", [("moon.7.13", 3.0), ("jpl.1024.4.5", 3.0), ("jpl.4096.4.5", 3.2), ("ira-12x24-sz64", 2.0)])
@pytest.mark.parametrize("variant", ["min", "tanh"])
def test_f64_trajectory(hip, name, db, variant):
    c = _get(name)
    lp = _layers(c)
    _, llr = c.frames(6, db, seed=900)
    dec = hip.Decoder(c.hip_code(hip), variant, "f64", len(llr), schedule="layered")
    assert dec.schedule == "layered" and dec.path == "flood" and np.array_equal(dec.code.layers(), lp)
    # QC codes: one workgroup per frame (layered_qc.hip); any other H: the batch-major kernel (flood.hip)
    assert ("layered_qc_kernel" in dec.kernel_name) == (name != "moon.7.13"), dec.kernel_name
    bits, its, conv, trace = dec.decode_trace(llr, 25)
    for f in range(len(llr)):
        o = oracle.decode_layered(c.graph, lp, variant, 25, llr[f], trace=True)
        assert its[f] == o["iters"] and bool(conv[f]) == o["converged"], (name, variant, f, its[f], o["iters"])
        assert np.array_equal(bits[f], o["bits"])
        if variant == "min":
            assert np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"])
        else:
            rel = np.abs(trace[f, : o["iters"] + 1] - o["trace_lam"]) / np.maximum(1, np.abs(o["trace_lam"]))
            assert rel.max() <= 1e-9, rel.max()


@pytest.mark.parametrize("name,dbs", [("jpl.1024.4.5", (2.5, 3.5)), ("jpl.4096.4.5", (2.8, 3.6)), ("regular36-sz128", (2.0, 3.0)), ("1920.1280.3.303", (1.5, 2.5))])
def test_f32_free_running_and_teacher_forced(hip, name, dbs):
    c = _get(name)
    lp = _layers(c)
    F = 24 if c.N > 4000 else 48
    llr = np.concatenate([c.frames(F // 2, db, 910 + i)[1] for i, db in enumerate(dbs)])
    for variant in ("min", "tanh"):
        dec = hip.Decoder(c.hip_code(hip), variant, "f32", F, schedule="layered", path="flood")   # (the HBM kernels: they have the teacher-forced step)
        bits, its, conv = dec.decode_batch(llr.astype(np.float32), 40)
        ref = [oracle.decode_layered(c.graph, lp, variant, 40, l) for l in llr]
        ob = np.stack([o["bits"] for o in ref]); oi = np.array([o["iters"] for o in ref]); oc = np.array([o["converged"] for o in ref])
        # f32 against a double specification: a frame at the edge of convergence may fall the other way (the serial
        # schedule amplifies rounding differences faster than flooding does).  Measured over 30 600 frame decodes
        # (profiles/r04_layered_f32_vs_f64.txt): min-sum flags differ in 1.4 % of the frames (3.6 % at the waterfall's edge), sweep counts
        # in 2.7 % (9.8 %); tanh 2e-4 and 4e-4; the bits of every frame whose flags agree are identical.  Bars for these small samples:
        same = conv.astype(bool) == oc
        assert same.mean() >= (0.90 if variant == "min" else 0.97) and np.array_equal(bits[same], ob[same]), (name, variant, same.mean())
        assert (its == oi)[same].mean() >= (0.85 if variant == "min" else 0.95), (name, variant, (its == oi)[same].mean())
        # one sweep from oracle states
        states = []
        for f in range(4):
            lam, msg = llr[f].copy(), np.zeros(c.E)
            for n in range(min(ref[f]["iters"], 6)):
                m2, l2, _, _ = oracle.layered_step(c.graph, variant, lam, msg)
                states.append((llr[f], lam, msg, m2, l2))
                lam, msg = l2, m2
        states = states[:F]
        if states:
            ne2, lam2, _ = dec.debug_step(np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), np.stack([s[2] for s in states]))
            worst = 0.0
            for i, s in enumerate(states):
                tol_lam, tol_ne = lam_tolerance(c.graph, s[3], s[4])
                assert (np.abs(ne2[i] - s[3]) <= tol_ne).all() and (np.abs(lam2[i] - s[4]) <= tol_lam).all(), (name, variant, i)
                worst = max(worst, (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max())
            print(f"{name} layered {variant} f32: {int(conv.sum())}/{F} converged, sweeps {100 * (its == oi).mean():.0f}% identical, "
                  f"worst teacher-forced relative LLR error {worst:.2e} over {len(states)} sweeps")


@pytest.mark.parametrize("name", ["jpl.1024.4.5", "ira-12x24-sz64"])
def test_both_layered_kernels_agree_on_qc_codes(hip, name, monkeypatch):
    """the frame-per-workgroup QC kernel and the batch-major any-H kernel implement one specification: identical f32
    results (bits, sweeps, flags), f64 trajectories identical bit for bit"""
    c = _get(name)
    _, llr = c.frames(70, 3.0, seed=123)
    code = c.hip_code(hip)
    for variant in ("min", "tanh"):
        qc = hip.Decoder(code, variant, "f32", 70, schedule="layered", path="flood")
        monkeypatch.setenv("LDPC_LAYERED_QC", "0")
        bm = hip.Decoder(code, variant, "f32", 70, schedule="layered", path="flood")
        bm64 = hip.Decoder(code, variant, "f64", 4, schedule="layered")
        monkeypatch.delenv("LDPC_LAYERED_QC")
        assert "layered_qc_kernel" in qc.kernel_name and bm.kernel_name == "layered_kernel"
        a = qc.decode_batch(llr.astype(np.float32), 40)
        b = bm.decode_batch(llr.astype(np.float32), 40)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (name, variant)
        qc64 = hip.Decoder(code, variant, "f64", 4, schedule="layered")
        ta = qc64.decode_trace(llr[:4], 20)
        tb = bm64.decode_trace(llr[:4], 20)
        assert all(np.array_equal(x, y) for x, y in zip(ta, tb))


def test_layered_needs_fewer_sweeps_and_same_answers(hip):
    c = load("jpl.4096.4.5")
    _, llr = c.frames(256, 3.4, seed=77)
    code = c.hip_code(hip)
    fl = hip.Decoder(code, "min", "f32", 256)
    la = hip.Decoder(code, "min", "f32", 256, schedule="layered")
    fb, fi, fc = fl.decode_batch(llr.astype(np.float32), 50)
    lb, li, lc = la.decode_batch(llr.astype(np.float32), 50)
    both = (fc == 1) & (lc == 1)
    assert both.sum() > 200 and lc.sum() >= fc.sum()
    assert np.array_equal(fb[both], lb[both])                 # both found the (same) codeword
    assert li[both].mean() < 0.7 * fi[both].mean()
    print(f"jpl.4096 3.4 dB: flooding {fi[both].mean():.1f} turns, layered {li[both].mean():.1f} sweeps; converged {int(fc.sum())} vs {int(lc.sum())} of 256")


def test_edge_cases_and_custom_layers(hip):
    c = load("jpl.1024.4.5")
    cws, llr = c.frames(70, 4.0, seed=5)                   # ragged batch: 70 frames in two 64-frame slabs
    code_csr = c.hip_code(hip, prefer_qc=False)
    assert len(code_csr.layers()) == c.M + 1                # CSR default: every row its own layer
    code_csr.set_layers(np.arange(0, c.M + 1, c.sz))        # the block rows, by hand
    with pytest.raises(hip.LdpcError):
        c.hip_code(hip, prefer_qc=False).set_layers(np.arange(0, c.M + 1, 2 * c.sz))   # two block rows share columns
    dec = hip.Decoder(code_csr, "min", "f32", 70, schedule="layered")
    z = llr.copy(); z[0] = (2.0 * cws[0] - 1.0) * 8.0; z[1] = 0.0
    bits, its, conv = dec.decode_batch(z.astype(np.float32), 30)
    lp = np.arange(0, c.M + 1, c.sz)
    for f in range(70):
        o = oracle.decode_layered(c.graph, lp, "min", 30, z[f])
        assert np.array_equal(bits[f], o["bits"]) and bool(conv[f]) == o["converged"]
    assert its[0] == 0 and its[1] == 0 and not bits[1].any()
    with pytest.raises(hip.LdpcError):                      # contexts exist now: the partition is frozen
        code_csr.set_layers(np.arange(c.M + 1))
    b0, i0, c0 = dec.decode_batch(llr[:3].astype(np.float32), 0)     # no sweeps allowed: channel decisions
    assert np.array_equal(b0, (llr[:3] > 0).astype(np.uint8)) and not c0.any()
    with pytest.raises(hip.LdpcError) as e:                 # fp16 STORAGE of a layered decoder's HBM state: min-sum row records of QC codes only
        hip.Decoder(c.hip_code(hip), "tanh", "f16", 8, schedule="layered", path="flood")
    assert e.value.code == -5
    with pytest.raises(hip.LdpcError):
        hip.Decoder(code_csr, "min", "f16", 8, schedule="layered", path="flood")
    from oracle import emulate_f16 as em                    # on-chip the only thing in HBM is the LLRs: the f32 decoder on fp16-rounded LLRs
    d16 = hip.Decoder(c.hip_code(hip), "min", "f16", 70, schedule="layered")
    d32 = hip.Decoder(c.hip_code(hip), "min", "f32", 70, schedule="layered")
    assert d16.path == "fused"
    assert all(np.array_equal(x, y) for x, y in zip(d16.decode_batch(llr.astype(np.float32), 30), d32.decode_batch(em.r16(llr), 30)))
    ecc = hip.ECC(CODES, "ldpc/hip-minsum-layered/jpl.1024.4.5/50/4/5", max_batch=4)
    assert ecc.decoder.schedule == "layered"
    out, ok = ecc.decode(llr[5][:1280])
    assert ok and np.array_equal(out, oracle.decode_layered(c.graph, lp, "min", 50, llr[5])["bits"][:1024])


def test_dvbs2_shaped_long_code(hip):
    """BASELINE.json configs[4]: n = 64 800, period 360 (a SYNTHETIC matrix with the DVB-S2 rate-1/2 shape: the ETSI
    tables are not available; tools/gen_dvbs2_like.py).  A frame's LLRs alone are 253 KB -- beyond every on-chip
    kernel -- so both schedules run from HBM.  Parity against the oracle on a few frames, f64 bit-exact."""
    from oracle import formats
    import os
    sz, rows = formats.read_qc(open(os.path.join(CODES, "dvbs2like.64800.1.2", "H.q")).read())
    off = formats.qc_offsets(sz, rows)
    assert sz == 360 and off.shape == (90, 180)
    ecc = hip.ECC(CODES, "ldpc/hip-minsum-layered/dvbs2like.64800.1.2/50", max_batch=64)
    assert (ecc.message_length, ecc.codeword_length, ecc.unpunctured_length) == (32400, 64800, 64800)
    assert ecc.decoder.schedule == "layered" and ecc.decoder.path == "flood"
    rp, ci = ecc.code.csr()
    g = oracle.Graph(rp, ci, ecc.code.N)
    lp = ecc.code.layers()
    assert len(lp) == 91 and lp[1] == 360
    from oracle import channel
    llr = channel.frames(np.zeros((6, g.N), np.uint8), 1.6, 32400, 64800, g.N, seed=3)
    bits, its, conv = ecc.decoder.decode_batch(llr.astype(np.float32), 50)
    ref = [oracle.decode_layered(g, lp, "min", 50, l) for l in llr]
    assert all(np.array_equal(bits[f], ref[f]["bits"]) and bool(conv[f]) == ref[f]["converged"] for f in range(6))
    d64 = hip.Decoder(ecc.code, "min", "f64", 2, schedule="layered")
    b2, i2, c2, lam = d64.decode_batch(llr[:2], 50, want_lam=True)
    for f in range(2):
        assert i2[f] == ref[f]["iters"] and np.array_equal(lam[f], ref[f]["lam"])
    # flooding on the same code for comparison (flood path): also the oracle's answer, in more turns
    fl = hip.Decoder(ecc.code, "min", "f32", 6)
    assert fl.path == "flood"
    fb, fi, fc = fl.decode_batch(llr.astype(np.float32), 50)
    ob, oi, oc = oracle.decode_batch(g, "min", 50, llr, nthreads=6)
    assert np.array_equal(fb, ob) and np.array_equal(fc, oc)
    print(f"dvbs2like 1.6 dB: layered {its.tolist()} sweeps, flooding {fi.tolist()} turns")


@pytest.mark.parametrize("name", ["jpl.4096.4.5", "ira-12x24-sz64", "regular36-sz128"])
def test_row_records_equal_per_edge_messages(hip, name, monkeypatch):
    """layered min-sum on QC codes keeps a check row's messages in HBM as a record {3/4 min1, 3/4 min2, signs | arg-min}
    (12 bytes per row) instead of one word per edge; the rebuilt messages must be the per-edge kernel's bit for bit:
    identical f32 results and f64 trajectories, and the oracle's trajectory in f64."""
    c = _get(name)
    lp = _layers(c)
    F = 20 if c.N > 4000 else 48
    _, llr = c.frames(F, 3.0 if name != "ira-12x24-sz64" else 2.0, seed=555)
    code = c.hip_code(hip)
    rec = hip.Decoder(code, "min", "f32", F, schedule="layered", path="flood")
    rec64 = hip.Decoder(code, "min", "f64", 4, schedule="layered")
    monkeypatch.setenv("LDPC_LAYERED_RECORDS", "0")
    edge = hip.Decoder(code, "min", "f32", F, schedule="layered", path="flood")
    edge64 = hip.Decoder(code, "min", "f64", 4, schedule="layered")
    monkeypatch.delenv("LDPC_LAYERED_RECORDS")
    assert rec.kernel_name.endswith(", true>") and edge.kernel_name.endswith(", false>")
    a, b = rec.decode_batch(llr.astype(np.float32), 40), edge.decode_batch(llr.astype(np.float32), 40)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(set(a[1].tolist())) > 2
    ta, tb = rec64.decode_trace(llr[:4], 25), edge64.decode_trace(llr[:4], 25)
    assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
    for f in range(4):
        o = oracle.decode_layered(c.graph, lp, "min", 25, llr[f], trace=True)
        assert ta[1][f] == o["iters"] and np.array_equal(ta[3][f, : o["iters"] + 1], o["trace_lam"])


@pytest.mark.parametrize("name,F,dbs", [("jpl.1024.4.5", 41, (2.5, 3.5)), ("jpl.4096.4.5", 13, (2.8, 3.4))])
def test_fp16_lam_storage_from_hbm(hip, name, F, dbs, monkeypatch):
    """LDPC_F16 + LDPC_SCHED_LAYERED + LDPC_PATH_FLOOD: lam stored in fp16, f32 arithmetic and row records -- against
    oracle/emulate_f16.py decode_minsum_f16_layered: bits, sweeps, flags and the LLRs a frame stops with, exactly.  Two kernels
    implement it: lam in LDS with the records streamed from HBM (r04, layered_lds.hip: whenever a frame's fp16 LLRs fit the LDS) and
    lam in HBM too (layered_qc_kernel<..., __half>, LDPC_LAYERED_LDS=0)."""
    from oracle import emulate_f16 as em
    c = load(name)
    llr = np.concatenate([c.frames(F // 2 + 1, db, 7300 + i)[1] for i, db in enumerate(dbs)])[:F].astype(np.float32)
    llr = llr[np.random.default_rng(4).permutation(F)]
    llr[0, :8] = [7e4, -7e4, 1e-9, -1e-9, 0.0, 65504.0, 3.0e-8, -6.0e-8]     # saturation, underflow to zero (hard 0 = False), subnormal
    dec = hip.Decoder(c.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
    assert dec.path == "flood" and "layered_lds_kernel" in dec.kernel_name, dec.kernel_name
    bits, its, conv, lam = dec.decode_batch(llr, 30, want_lam=True)
    eb, ei, ec, el = em.decode_minsum_f16_layered(c.graph, llr, 30)
    assert np.array_equal(bits, eb) and np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec)
    assert np.array_equal(lam, el.astype(np.float64))
    assert 0 < conv.sum() < F and len(set(its.tolist())) > 3
    monkeypatch.setenv("LDPC_LAYERED_LDS", "0")                               # lam in HBM as well (2-byte accesses): the same decoder
    hbm = hip.Decoder(c.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
    monkeypatch.delenv("LDPC_LAYERED_LDS")
    assert "layered_qc_kernel" in hbm.kernel_name and "__half" in hbm.kernel_name, hbm.kernel_name
    one = hbm.decode_batch(llr, 30, want_lam=True)
    assert all(np.array_equal(x, y) for x, y in zip(one, (bits, its, conv, lam)))
    monkeypatch.setenv("LDPC_LAYERED_LDS_PREFETCH", "0")                      # records loaded where they are used: the same decoder
    nopf = hip.Decoder(c.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
    monkeypatch.delenv("LDPC_LAYERED_LDS_PREFETCH")
    assert nopf.kernel_name.endswith(", 0, 1>") and ", 4, " in dec.kernel_name, (nopf.kernel_name, dec.kernel_name)
    assert all(np.array_equal(x, y) for x, y in zip(nopf.decode_batch(llr, 30, want_lam=True), (bits, its, conv, lam)))
    for nf in (1, 2, 3):                                                      # small batches
        b1 = dec.decode_batch(llr[:nf], 30)
        assert np.array_equal(b1[0], bits[:nf]) and np.array_equal(b1[1], its[:nf])
    tb, ti, tc, tr = dec.decode_trace(llr[:5].astype(np.float64), 30)         # per-sweep LLRs (the instance without hand-counted prefetch writes them)
    assert np.array_equal(tb, bits[:5]) and np.array_equal(ti, its[:5]) and np.array_equal(tr[:, 0], em.r16(llr[:5]).astype(np.float64))
    for f in range(5):
        if conv[f]:
            assert np.array_equal(tr[f, its[f]], lam[f])                      # the row of the sweep it stopped at = the LLRs it returns
    b16 = dec.decode_batch(em.r16(llr).astype(np.float16), 30)                # fp16 input buffer: the same decoder
    assert np.array_equal(b16[0], bits) and np.array_equal(b16[1], its)
    f32 = hip.Decoder(c.hip_code(hip), "min", "f32", F, schedule="layered", path="flood").decode_batch(llr, 30)
    both = conv.astype(bool) & f32[2].astype(bool)
    assert both.sum() >= 0.8 * f32[2].sum() and np.array_equal(bits[both], f32[0][both])   # same codewords as the f32-state kernel
    with pytest.raises(hip.LdpcError):
        dec.debug_step(np.zeros((1, c.N)), np.zeros((1, c.N)), np.zeros((1, c.E)))
    for d in (dec, hbm, nopf):
        d.close()


@pytest.mark.parametrize("name", ["wifi-12x24-sz27", "wimax-12x24-sz96", "dvbs2short-20x45-sz360", "irregular-20x30-sz64", "latin-24x16-sz64", "latin-18x9-sz40"])
def test_fp16_lam_storage_other_shapes(hip, name, monkeypatch):
    """the lam-in-LDS kernel on circulant sizes that are not powers of two (idle lanes shadow rows of their own wave) and on rows above
    weight 8 (its padded row instances), with more frames than persistent workgroups would need and a per-sweep trace, against the emulation"""
    from oracle import emulate_f16 as em
    c = synthetic(name)
    F = 9
    llr = np.concatenate([c.frames(5, db, 7500 + i)[1] for i, db in enumerate((2.5, 5.0) if c.N > 1000 else (4.0, 7.0))])[:F].astype(np.float32)
    dec = hip.Decoder(c.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
    bits, its, conv, lam = dec.decode_batch(llr, 25, want_lam=True)
    assert "layered_lds_kernel" in dec.kernel_name, dec.kernel_name
    eb, ei, ec, el = em.decode_minsum_f16_layered(c.graph, llr, 25)
    assert np.array_equal(bits, eb) and np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec), name
    assert np.array_equal(lam, el.astype(np.float64))
    monkeypatch.setenv("LDPC_LAYERED_LDS", "0")
    hbm = hip.Decoder(c.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
    monkeypatch.delenv("LDPC_LAYERED_LDS")
    assert all(np.array_equal(x, y) for x, y in zip(hbm.decode_batch(llr, 25, want_lam=True), (bits, its, conv, lam)))
    monkeypatch.setenv("LDPC_LAYERED_LDS_GROUPS", "0")                       # one block row at a time: the same results
    one = hip.Decoder(c.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
    monkeypatch.delenv("LDPC_LAYERED_LDS_GROUPS")
    assert all(np.array_equal(x, y) for x, y in zip(one.decode_batch(llr, 25, want_lam=True), (bits, its, conv, lam)))
    if name.startswith("latin"):                                              # groups were formed: more threads than one block row has
        assert dec.kernel_geometry[0] > one.kernel_geometry[0], (dec.kernel_geometry, one.kernel_geometry)
    for rw in ("1", "2"):                                                     # one / two block rows of a group per set of waves: the same results
        monkeypatch.setenv("LDPC_LAYERED_LDS_RW", rw)
        d = hip.Decoder(c.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
        out = d.decode_batch(llr, 25, want_lam=True)
        assert d.kernel_name.endswith(f", {rw}>") or d.kernel_name.endswith(", 0, 1>"), d.kernel_name
        assert all(np.array_equal(x, y) for x, y in zip(out, (bits, its, conv, lam))), (name, rw, d.kernel_name)
        d.close()
    monkeypatch.delenv("LDPC_LAYERED_LDS_RW")
    dec.close(); hbm.close(); one.close()


def test_fp16_lam_on_chip_long_code(hip):
    """BASELINE.json configs[4] through LDPC_PATH_AUTO: the DVB-S2-shaped n = 64 800 code with fp16 LLR storage takes the lam-in-LDS
    kernel (130 KB of LDS per frame, one workgroup per CU, more frames than workgroups: the work counter runs) -- bit for bit the
    emulation, and the same codewords as the f32-state kernel"""
    from oracle import emulate_f16 as em, channel
    ecc = hip.ECC(CODES, "ldpc/hip-minsum-layered-f16/dvbs2like.64800.1.2/50", max_batch=600)
    dec = ecc.decoder
    assert dec.schedule == "layered" and dec.path == "flood" and "layered_lds_kernel<8, 4, 2>" in dec.kernel_name, dec.kernel_name
    rp, ci = ecc.code.csr()
    g = oracle.Graph(rp, ci, ecc.code.N)
    llr = np.concatenate([channel.frames(np.zeros((3, g.N), np.uint8), db, 32400, 64800, g.N, seed=30 + i) for i, db in enumerate((1.0, 1.5, 2.0))]).astype(np.float32)
    llr[1, :6] = [7e4, -7e4, 1e-9, -1e-9, 0.0, 65504.0]
    eb, ei, ec, el = em.decode_minsum_f16_layered(g, llr, 20)
    # 600 frames (the nine, repeated in a shuffled order): more frames than resident workgroups
    order = np.random.default_rng(8).integers(0, len(llr), 600)
    bits, its, conv, lam = dec.decode_batch(llr[order], 20, want_lam=True)
    assert np.array_equal(bits, eb[order]) and np.array_equal(its, ei[order]) and np.array_equal(conv.astype(bool), ec[order])
    assert np.array_equal(lam, el[order].astype(np.float64))
    assert 0 < ec.sum() < len(llr)
    assert dec.kernel_geometry[0] == 768                                       # four block rows at a time: two sets of 6 waves, two rows each
    f32 = hip.Decoder(ecc.code, "min", "f32", 9, schedule="layered").decode_batch(llr, 20)
    both = ec & f32[2].astype(bool)
    assert both.any() and np.array_equal(eb[both], f32[0][both])
    print(f"dvbs2like f16 lam on-chip: sweeps {ei.tolist()} (f32 state {f32[1].tolist()})")


def test_fp16_lam_on_chip_edge_cases(hip):
    """the lam-in-LDS kernel at its edges: a frame length that is not a multiple of 8 (the byte-wise prologue / epilogue), no sweeps allowed,
    one sweep, a noiseless codeword (stops before the first sweep), an all-zero LLR vector (hard 0 = False: the zero word, 0 sweeps), one frame,
    more frames than workgroups would need on a small code -- against the emulation"""
    from oracle import emulate_f16 as em
    from tests.helpers import SyntheticQC
    rng = np.random.default_rng(41)
    mask = np.zeros((6, 11), bool)
    for br in range(6):
        mask[br, 5 + br] = True
        if br:
            mask[br, 5 + br - 1] = True
        mask[br, rng.choice(5, 3, replace=False)] = True
    off = np.where(mask, rng.integers(0, 27, mask.shape), -1).astype(np.int32)
    c = SyntheticQC("ira-6x11-sz27", 27, off, rate=(135, 297))
    assert c.N % 8 != 0
    _, llr = c.frames(300, 3.0, seed=77)
    llr = llr.astype(np.float32)
    llr[0] = 9.0 * (2.0 * 0 - 1.0)        # the all-zero codeword, noiseless (LLR < 0 everywhere: hard = 0)
    llr[1] = 0.0
    dec = hip.Decoder(c.hip_code(hip), "min", "f16", 300, schedule="layered", path="flood")
    assert "layered_lds_kernel" in dec.kernel_name
    for turns in (0, 1, 25):
        bits, its, conv, lam = dec.decode_batch(llr, turns, want_lam=True)
        eb, ei, ec, el = em.decode_minsum_f16_layered(c.graph, llr, turns)
        assert np.array_equal(bits, eb) and np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec) and np.array_equal(lam, el.astype(np.float64)), turns
        b2, i2, c2 = dec.decode_batch(llr, turns)                                   # (without the LLR output: the other store path)
        assert np.array_equal(b2, eb) and np.array_equal(i2, ei)
        assert its[0] == 0 and conv[0] and its[1] == 0 and conv[1] and not bits[1].any()
    one = dec.decode_batch(llr[7:8], 25)
    assert np.array_equal(one[0][0], eb[7]) and one[1][0] == ei[7]
    dec.close()


def test_layer_order_helper_feeds_the_long_code_kernel(hip):
    """ldpc_qc_layer_order on a matrix in accumulator order (every block row shares a parity column with the next: the lam-on-chip kernel
    must take them one at a time): the proposed order is a permutation of the rows (same code), gives the kernel runs of independent block
    rows to work on together (its two-rows-per-set-of-waves instance), both orders are bit for bit their own emulation, and frames both
    decode come out as the same codewords."""
    from oracle import emulate_f16 as em
    from tests.helpers import SyntheticQC
    c = synthetic("dvbs2short-20x45-sz360")
    perm, full = hip.Code.qc_layer_order(c.offsets, 2)      # (rows of weight up to 13: one set of six waves per workgroup, pairs)
    assert sorted(perm.tolist()) == list(range(20)) and full == 10
    c2 = SyntheticQC("dvbs2short-reordered", 360, c.offsets[perm])
    assert np.array_equal(c2.H, c.H.reshape(20, 360, -1)[perm].reshape(c.H.shape))          # the same checks, another order
    F = 12
    _, llr = c.frames(F, 3.0, seed=812)
    llr = llr.astype(np.float32)
    outs = []
    for code, want in ((c, ", 1>"), (c2, ", 2>")):
        d = hip.Decoder(code.hip_code(hip), "min", "f16", F, schedule="layered", path="flood")
        bits, its, conv = d.decode_batch(llr, 30)
        assert "layered_lds_kernel" in d.kernel_name and d.kernel_name.endswith(want), d.kernel_name
        eb, ei, ec, _ = em.decode_minsum_f16_layered(code.graph, llr, 30)
        assert np.array_equal(bits, eb) and np.array_equal(its, ei) and np.array_equal(conv.astype(bool), ec)
        outs.append((bits, conv.astype(bool)))
        d.close()
    both = outs[0][1] & outs[1][1]
    assert both.sum() >= F // 2 and np.array_equal(outs[0][0][both], outs[1][0][both])
