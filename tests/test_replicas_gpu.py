"""GPU: what a multi-threaded host sees.  The reference makes maxThreadCount decoder replicas in ONE process and its
harness calls the record's decode once per frame from several threads (src/ECC/Code/LDPC/Utils.hs:53,63-69):
  * ldpc_ecc_create_replicas: several replicas behind one record, picked by calling thread; explicit devices;
  * ldpc_ecc_set_coalescing / ldpc_batcher: concurrent per-frame calls share one launch -- same answers as one by one;
  * objects on an explicitly named device from a thread that never called ldpc_init;
  * the native multi-rank CLI: one thread per rank, frame ranges sharded, tallies summed (host) / all-reduced (RCCL)."""
import subprocess
import threading
import time

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import CODES, load

pytestmark = pytest.mark.gpu
NAME = "ldpc/hip-minsum/jpl.1024.4.5/50/4/5"


def _run_threads(n, fn):
    errs, th = [], []
    def wrap(i):
        try:
            fn(i)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    for i in range(n):
        t = threading.Thread(target=wrap, args=(i,))
        t.start(); th.append(t)
    for t in th:
        t.join()
    assert not errs, errs[0]


def test_replicas_are_picked_by_thread_and_agree(hip):
    c = load("jpl.1024.4.5")
    _, llr = c.frames(48, 3.4, seed=21)
    want = [oracle.decode(c.graph, "min", 50, l)["bits"][:1024] for l in llr]
    ecc = hip.ECC(CODES, NAME, max_batch=8, devices=[0, 0, 0])
    assert ecc.replicas == 3
    got = [None] * 48
    def work(i):
        for f in range(i, 48, 6):
            got[f], ok = ecc.decode(llr[f][:1280])
            assert ok
    _run_threads(6, work)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    out, ok = ecc.decode(llr[0][:1280], replica=2)
    assert ok and np.array_equal(out, want[0])
    dec2, sim2 = ecc.replica(2)
    assert dec2.path == "fused"
    ecc.close()


def test_coalesced_calls_equal_one_by_one_and_share_launches(hip):
    c = load("jpl.1024.4.5")
    _, llr = c.frames(256, 3.2, seed=22)
    ecc = hip.ECC(CODES, NAME, max_batch=64)
    one_by_one = [ecc.decode(l[:1280])[0] for l in llr]
    ecc.set_coalescing(32, 5000)
    got = [None] * 256
    def work(i):
        for f in range(i, 256, 32):
            got[f], ok = ecc.decode(llr[f][:1280])
            assert ok
    _run_threads(32, work)
    calls, launches = ecc.coalescing_stats()
    assert calls == 256 and launches <= 64, (calls, launches)        # 32 callers at a time: far fewer launches than frames
    assert all(np.array_equal(got[f], one_by_one[f]) for f in range(256))      # exactly what ldpc_decode_one gives
    ref, _, rconv = oracle.decode_batch(c.graph, "min", 50, llr, nthreads=8)
    agree = np.array([np.array_equal(got[f], ref[f][:1024]) for f in range(256)])
    assert agree.mean() > 0.97        # (f32 kernels vs the double oracle: a frame at the edge of convergence may differ)
    # a lone caller is served after the wait budget, not never
    t0 = time.time()
    out, ok = ecc.decode(llr[3][:1280])
    assert ok and np.array_equal(out, one_by_one[3]) and time.time() - t0 < 1.0
    print(f"coalescing: {calls} calls in {launches} launches")
    # harness-visible rate, frames decoded one per call from T threads (jpl.1024, 3.2 dB)
    for T, coalesce in ((1, 0), (8, 0), (8, 8), (64, 64)):
        ecc.set_coalescing(coalesce, 300)
        per = 256 // T if T <= 8 else 4
        def work2(i):
            for f in range(per):
                ecc.decode(llr[(i * per + f) % 256][:1280])
        t0 = time.time()
        _run_threads(T, work2)
        dt = time.time() - t0
        print(f"  {T:3d} threads, coalescing {coalesce:2d}: {T * per * 1024 / dt / 1e6:8.2f} Mbit/s harness-visible ({T * per / dt:7.0f} frames/s)")
    ecc.close()


def test_haskell_binding_call_sequence(hip):
    """haskell/ECC/Code/LDPC/GPU/HIP.hs closureFor, call for call: ldpc_code_create_qc / _csr -> ldpc_ctx_create_cfg
    {device, rule, f32, 64 frames, auto path, schedule} -> ldpc_batcher_create(ctx, 64, 200 us) -> ldpc_batcher_decode_one
    from hipThreads = 64 host threads.  Same answers as ldpc_decode_one on the same context, for the QC codes, the
    `Matrix Bool` (CSR) codes and the layered code; the bad-argument paths the binding could hit give error codes."""
    c = load("jpl.1024.4.5")
    _, llr = c.frames(128, 3.4, seed=31)
    flavours = [(c.hip_code(hip, prefer_qc=True), "min", "flooding"), (c.hip_code(hip, prefer_qc=True), "tanh", "flooding"),
                (c.hip_code(hip, prefer_qc=False), "min", "flooding"), (c.hip_code(hip, prefer_qc=True), "min", "layered")]
    for code, rule, sched in flavours:
        dec = hip.Decoder(code, rule, "f32", 64, device=0, schedule=sched)           # ldpc_ctx_create_cfg
        want = [dec.decode_one(l, 30) for l in llr]
        b = hip.Batcher(dec, 64, 200)
        got = [None] * 128
        def work(i):
            for f in (i, i + 64):
                got[f] = b.decode_one(llr[f], 30)
        _run_threads(64, work)
        calls, launches = b.stats()
        assert calls == 128 and launches < 128, (calls, launches)
        for f in range(128):
            assert np.array_equal(got[f][0], want[f][0]) and got[f][1:] == want[f][1:], (rule, sched, f)
        # callers with different max_iters never share a batch, and still get their own answers
        def work2(i):
            got[i] = b.decode_one(llr[i], 5 if i % 2 else 30)
        _run_threads(16, work2)
        for i in range(16):
            assert np.array_equal(got[i][0], dec.decode_one(llr[i], 5 if i % 2 else 30)[0])
        b.close()
        with pytest.raises(hip.LdpcError) as e:
            hip.Batcher(dec, 65, 200)                                                  # more frames than the context holds
        assert e.value.code == -1
        with pytest.raises(hip.LdpcError):
            hip.Batcher(dec, 0, 200)
        dec.close()
    # struct_size too small / unknown enum values are refused before anything is allocated
    import ctypes as C
    from ecc_ldpc_amd._lib import CtxConfig, lib
    code = c.hip_code(hip)
    for bad in (dict(struct_size=8), dict(variant=7), dict(dtype=9), dict(schedule=5), dict(max_batch=0), dict(device=99)):
        cfg = CtxConfig(C.sizeof(CtxConfig), 0, 1, 0, 64, 0, 0)
        for k, v in bad.items():
            setattr(cfg, k, v)
        assert not lib().ldpc_ctx_create_cfg(code._h, C.byref(cfg)), bad
        assert lib().ldpc_last_error_code() in (-1, -4, -5), (bad, lib().ldpc_last_error_code())


def test_explicit_device_from_a_thread_without_init(hip):
    c = load("jpl.1024.4.5")
    _, llr = c.frames(4, 4.0, seed=23)
    res = {}
    def work(_):
        assert hip.lib().ldpc_current_device() == 0          # never called ldpc_init here: the process default
        code = c.hip_code(hip)
        dec = hip.Decoder(code, "min", "f32", 4, device=0)   # ldpc_ctx_create_cfg with an explicit device
        res["bits"] = dec.decode_batch(llr.astype(np.float32), 50)[0]
        with pytest.raises(hip.LdpcError):
            hip.Decoder(code, "min", "f32", 4, device=99)
        dec.close(); code.close()
    _run_threads(1, work)
    assert np.array_equal(res["bits"], oracle.decode_batch(c.graph, "min", 50, llr, nthreads=4)[0])


def test_native_cli_ranks(hip):
    from ecc_ldpc_amd.build import CLI
    args = ["3.2", NAME, "-m20000", "-b4096", "-c" + CODES]
    def rows(extra):
        p = subprocess.run([CLI] + args + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        return [l.split() for l in p.stdout.splitlines() if len(l.split()) > 5 and l.split()[1].startswith("ldpc/")]   # (RCCL prints a banner)
    one = rows(["-d0"])
    two = rows(["-d0,0", "-thost"])            # two ranks (threads) sharing the GPU, tallies summed on the host
    three = rows(["-d0,0,0", "-thost"])        # 20000 frames do not divide by 3: ragged shards
    rccl = rows(["-d0", "-trccl"])             # the RCCL all-reduce with a single rank
    for other in (two, three, rccl):
        assert other[0][1:6] == one[0][1:6], (one, other)     # name, Eb/N0, frames, bit errors, BER: the same frames were decoded
    assert "ranks," in " ".join(two[0]) and "rccl" not in " ".join(one[0])
    p = subprocess.run([CLI] + args + ["-d0,0", "-trccl"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "distinct" in p.stderr
    # --json: the one-process twin of bench.py's N-rank line
    import json
    p = subprocess.run([CLI] + args + ["-d0,0", "-thost", "--json"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    js = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1 and js[0]["ranks"] == 2 and js[0]["tally"] == "host" and len(js[0]["per_rank_seconds"]) == 2
    assert js[0]["frames"] == 20000 and str(js[0]["bit_errors"]) == one[0][4] and js[0]["value"] > 0 and js[0]["unit"] == "Mbit/s"


def test_config_of_an_older_header_keeps_its_schedule(hip):
    """ldpc_ctx_create_cfg with the 32-byte structure of the r02 header (struct_size = 32: no sum_order member) and schedule = LAYERED,
    as haskell/ECC/Code/LDPC/GPU/HIP.hs of that round passed it (allocaBytes 32): the context must be layered, not silently flooding"""
    import ctypes as C

    class OldCfg(C.Structure):
        _fields_ = [("struct_size", C.c_size_t), ("device", C.c_int), ("variant", C.c_int), ("dtype", C.c_int), ("max_batch", C.c_int),
                    ("path", C.c_int), ("schedule", C.c_int)]
    assert C.sizeof(OldCfg) == 32
    L = C.CDLL(hip.SO_PATH)                         # (the same loaded library, with prototypes of this test's own)
    L.ldpc_ctx_create_cfg.restype = C.c_void_p
    L.ldpc_ctx_create_cfg.argtypes = [C.c_void_p, C.c_void_p]
    L.ldpc_ctx_schedule.argtypes = [C.c_void_p]
    L.ldpc_ctx_destroy.argtypes = [C.c_void_p]
    L.ldpc_ctx_destroy.restype = None
    L.ldpc_last_error.restype = C.c_char_p
    c = load("jpl.1024.4.5").hip_code(hip)
    cfg = OldCfg(32, -1, 1, 0, 4, 0, 1)            # min-sum, f32, 4 frames, PATH_AUTO, LDPC_SCHED_LAYERED
    ctx = L.ldpc_ctx_create_cfg(C.c_void_p(c._h), C.byref(cfg))
    assert ctx, L.ldpc_last_error()
    assert L.ldpc_ctx_schedule(C.c_void_p(ctx)) == 1
    L.ldpc_ctx_destroy(C.c_void_p(ctx))
    cfg28 = OldCfg(28, -1, 1, 0, 4, 0, 1)          # a structure that ends before `schedule`: flooding
    ctx = L.ldpc_ctx_create_cfg(C.c_void_p(c._h), C.byref(cfg28))
    assert ctx and L.ldpc_ctx_schedule(C.c_void_p(ctx)) == 0
    L.ldpc_ctx_destroy(C.c_void_p(ctx))


def test_packed_result_bits(hip):
    """ldpc_decode_batch_packed / ldpc_decode_batch_dev_packed: ceil(N/8) bytes per frame, bit i at byte i // 8, bit i % 8 -- after unpacking
    bit for bit what the byte-per-bit entry points return (on-chip and HBM kernels, N a multiple of 8 and not, more frames than one
    chunk of the host pipeline, fp16 input)"""
    import torch
    for name, variant, F in (("jpl.1024.4.5", "min", 9000), ("moon.7.13", "tanh", 77), ("1920.1280.3.303", "tanh", 300)):
        c = load(name)
        _, llr = c.frames(F if F < 400 else 64, 3.0, seed=990)
        llr = np.tile(llr, ((F + len(llr) - 1) // len(llr), 1))[:F].astype(np.float32)
        dec = hip.Decoder(c.hip_code(hip), variant, "f32", F)
        bits, its, conv = dec.decode_batch(llr, 30)
        packed, pi, pc = dec.decode_batch_packed(llr, 30)
        assert packed.shape == (F, (c.N + 7) // 8)
        assert np.array_equal(np.unpackbits(packed, axis=1, bitorder="little")[:, : c.N], bits) and np.array_equal(pi, its) and np.array_equal(pc, conv)
        p16, _, _ = dec.decode_batch_packed(llr.astype(np.float16), 30)
        b16, _, _ = dec.decode_batch(llr.astype(np.float16), 30)
        assert np.array_equal(np.unpackbits(p16, axis=1, bitorder="little")[:, : c.N], b16)
        # device pointers
        dev = torch.device("cuda", 0)
        n = min(F, 256)
        t_llr = torch.from_numpy(llr[:n]).to(dev)
        t_pk = torch.zeros((n, (c.N + 7) // 8), dtype=torch.uint8, device=dev)
        t_it = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        dec.decode_batch_dev_packed(t_llr.data_ptr(), t_pk.data_ptr(), n, 30, t_it.data_ptr(), None, None)
        dec.synchronize()
        assert np.array_equal(np.unpackbits(t_pk.cpu().numpy(), axis=1, bitorder="little")[:, : c.N], bits[:n]) and np.array_equal(t_it.cpu().numpy(), its[:n])
        dec.close()
