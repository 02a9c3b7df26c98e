"""GPU: random irregular parity-check matrices (row degrees 0..40, column degrees 0..many) through every
path that accepts them, against the oracle.  Exercises the degree-generic code: flood.hip's re-read
fallback for row weights outside {1..8,18}, fused_csr.hip's padded rows (classes 4/8/20/32) and the
rejection rules (min-sum with a degree-1 row, rows above 32 edges on the on-chip kernel)."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def random_h(rng, M, N, degrees):
    H = np.zeros((M, N), np.uint8)
    for m in range(M):
        d = int(rng.choice(degrees))
        H[m, rng.choice(N, size=min(d, N), replace=False)] = 1
    return H


def llrs(rng, F, N, scale):
    x = rng.normal(1.5, scale, size=(F, N))  # mostly "bit 1" with noise: mixed converged / failed frames
    return x.astype(np.float32).astype(np.float64)


CASES = [
    ("tiny mixed", 24, 40, [0, 2, 3, 4, 5, 6, 7, 8], 30),
    ("mid degrees 9..20", 40, 96, [9, 10, 12, 15, 19, 20], 30),
    ("high degrees 21..32", 30, 128, [21, 25, 32, 3], 25),
    ("above 32 (flood only)", 20, 160, [33, 40, 4], 20),
]


@pytest.mark.parametrize("label,M,N,degs,iters", CASES)
def test_minsum_f64_bit_exact_on_every_path(hip, label, M, N, degs, iters):
    import zlib
    rng = np.random.default_rng(zlib.crc32(label.encode()))
    H = random_h(rng, M, N, degs)
    g = oracle.Graph.from_dense(H)
    x = llrs(rng, 20, N, 2.0)
    code = hip.Code.from_dense(H)
    paths = ["flood"] + (["fused"] if H.sum(1).max() <= 32 else [])
    for path in paths:
        dec = hip.Decoder(code, "min", "f64", len(x), path=path)
        bits, its, conv, trace = dec.decode_trace(x, iters)
        for f in range(len(x)):
            o = oracle.decode(g, "min", iters, x[f], trace=True)
            assert its[f] == o["iters"] and bool(conv[f]) == o["converged"], (label, path, f)
            assert np.array_equal(bits[f], o["bits"])
            assert np.array_equal(trace[f, : o["iters"] + 1], o["trace_lam"]), (label, path, f)
    if H.sum(1).max() > 32:
        with pytest.raises(hip.LdpcError) as e:
            hip.Decoder(code, "min", "f64", 4, path="fused")
        assert e.value.code == -5


@pytest.mark.parametrize("label,M,N,degs,iters", CASES)
@pytest.mark.parametrize("variant", ["min", "tanh"])
def test_f32_bits_and_path_agreement(hip, label, M, N, degs, iters, variant):
    import zlib
    rng = np.random.default_rng(zlib.crc32(label.encode()) + 7)
    H = random_h(rng, M, N, degs)
    g = oracle.Graph.from_dense(H)
    x = llrs(rng, 48, N, 2.5)
    code = hip.Code.from_dense(H)
    a = hip.Decoder(code, variant, "f32", len(x), path="flood").decode_batch(x.astype(np.float32), iters)
    ob, oi, oc = oracle.decode_batch(g, variant, iters, x, nthreads=4)
    assert np.array_equal(a[0], ob) and np.array_equal(a[2], oc)
    if H.sum(1).max() <= 32:
        b = hip.Decoder(code, variant, "f32", len(x), path="fused").decode_batch(x.astype(np.float32), iters)
        assert all(np.array_equal(p, q) for p, q in zip(a, b))


def test_tanh_f64_with_degree_one_rows(hip):
    rng = np.random.default_rng(5)
    H = random_h(rng, 16, 30, [1, 2, 3, 5, 8])  # degree-1 rows: product [] = 1 -> clamp (Orig.hs:86-91, Utils.hs:113-117)
    assert (H.sum(1) == 1).any()
    g = oracle.Graph.from_dense(H)
    x = llrs(rng, 12, 30, 2.0)
    code = hip.Code.from_dense(H)
    for path in ("flood", "fused"):
        dec = hip.Decoder(code, "tanh", "f64", len(x), path=path)
        bits, its, conv, trace = dec.decode_trace(x, 15)
        for f in range(len(x)):
            o = oracle.decode(g, "tanh", 15, x[f], trace=True)
            assert its[f] == o["iters"] and np.array_equal(bits[f], o["bits"])
            assert np.allclose(trace[f, : o["iters"] + 1], o["trace_lam"], rtol=1e-9, atol=1e-9)
    with pytest.raises(hip.LdpcError) as e:
        hip.Decoder(code, "min", "f32", 4)
    assert e.value.code == -6


@pytest.mark.parametrize("label,M,N,degs,iters", CASES[1:3])
def test_flood_padded_rows_equal_the_fallback(hip, monkeypatch, label, M, N, degs, iters):
    """Rows of weight 9..32 run in a second CN kernel instance with the row padded into registers
    (LDPC_FLOOD_WIDE=0: the O(d^2) re-read fallback).  min-sum is exact arithmetic: identical outputs, f32 and f64;
    the tanh rule differs only in summation order: identical bits on these frames."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(("wide" + label).encode()))
    H = random_h(rng, M, N, degs)
    g = oracle.Graph.from_dense(H)
    x = llrs(rng, 48, N, 2.5)
    out = {}
    for wide in ("1", "0"):
        monkeypatch.setenv("LDPC_FLOOD_WIDE", wide)
        code = hip.Code.from_csr(g.row_ptr, g.col_idx, N)
        for variant, dtype in (("min", "f32"), ("min", "f64"), ("tanh", "f32")):
            out[(wide, variant, dtype)] = hip.Decoder(code, variant, dtype, len(x), path="flood").decode_batch(x, iters, want_lam=True)
    for variant, dtype in (("min", "f32"), ("min", "f64")):
        assert all(np.array_equal(a, b) for a, b in zip(out[("1", variant, dtype)], out[("0", variant, dtype)]))
    assert np.array_equal(out[("1", "tanh", "f32")][0], out[("0", "tanh", "f32")][0])


@pytest.mark.parametrize("M,N,seed", [(4000, 1500, 1), (6100, 2040, 2), (3000, 1100, 3)])
@pytest.mark.parametrize("variant", ["min", "tanh"])
def test_large_frames_of_row_class_6_on_chip(hip, M, N, seed, variant):
    """fused_csr.hip's 1024-thread batched instance (built for codes/1920.1280.A: 16-bit DWORD offsets, offset tables re-read
    every turn) on random graphs of its class -- row weights <= 6, column weights <= 18, more than 64 KB of state per frame:
    ragged last rows / columns per thread, rows of weight 2..6, empty and light columns; against the flood path bit for bit
    and the oracle's hard bits"""
    rng = np.random.default_rng(seed)
    H = np.zeros((M, N), np.uint8)
    col_deg = np.zeros(N, int)
    for m in range(M):
        d = int(rng.choice([2, 3, 4, 5, 6, 6, 6]))
        free = np.flatnonzero(col_deg < 18)
        cols = rng.choice(free, size=d, replace=False)
        H[m, cols] = 1
        col_deg[cols] += 1
    assert H.sum(1).max() == 6 and H.sum(0).max() <= 18 and (N + 6 * M + 3) * 4 > 65536
    g = oracle.Graph.from_dense(H)
    x = rng.normal(2.2, 2.0, size=(9, N)).astype(np.float32)
    code = hip.Code.from_dense(H)
    on = hip.Decoder(code, variant, "f32", len(x), path="fused")
    a = on.decode_batch(x, 25)
    assert on.kernel_name == f"ldpc::fused_csr_batched_kernel<float, {1 if variant == 'min' else 0}, 6, 6, 2, 18, 1024, 2, false>", on.kernel_name
    b = hip.Decoder(code, variant, "f32", len(x), path="flood").decode_batch(x, 25)
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    ob, oi, oc = oracle.decode_batch(g, variant, 25, x.astype(np.float64), nthreads=8)
    assert np.array_equal(a[0], ob) and np.array_equal(a[2], oc)
    assert 0 < oc.sum() or oi.max() > 1
