"""GPU: parameter fuzz of the fused QC kernels against the flood path (same arithmetic, same order => identical
bits, iteration counts and flags): odd batch sizes (sz = 32 packs two frames per workgroup, the last one may hold a
shadow frame), 0 / 1 / few / many turns, mixed SNRs, f32 and fp16-LLR contexts, both rules."""
import zlib

import numpy as np
import pytest

from tests.helpers import load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["jpl.1024.4.5", "jpl.4096.4.5"])
@pytest.mark.parametrize("variant", ["min", "tanh"])
def test_fused_equals_flood_over_random_parameters(hip, name, variant):
    c = load(name)
    rng = np.random.default_rng(zlib.crc32(f"{name}/{variant}".encode()))
    code = c.hip_code(hip)
    pool = np.concatenate([c.frames(40, db, 5000 + i)[1] for i, db in enumerate((1.5, 2.8, 3.2, 4.0))]).astype(np.float32)
    cap = 97
    fused = hip.Decoder(code, variant, "f32", cap, path="fused")
    flood = hip.Decoder(code, variant, "f32", cap, path="flood")
    seen_iters = set()
    for trial in range(14):
        B = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 64, 65, 96, 97]))
        iters = int(rng.choice([0, 1, 2, 7, 20, 50]))
        llr = pool[rng.choice(len(pool), B, replace=False)].copy()
        if trial % 5 == 0:
            llr[0] = 0.0                                     # all-zero LLRs: 0 turns, all False (hard 0 = False)
        a = fused.decode_batch(llr, iters)
        b = flood.decode_batch(llr, iters)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (B, iters)
        assert (a[1] <= iters).all() and ((a[1] == iters) | (a[2] == 1)).all()
        if trial % 5 == 0:
            assert a[1][0] == 0 and a[2][0] == 1 and not a[0][0].any()
        seen_iters |= set(a[1].tolist())
    assert len(seen_iters) > 4


def test_argument_errors_are_codes_not_crashes(hip):
    """what a host can get wrong at the decode entry points: more frames than the context holds, a negative turn count, null
    buffers -- LDPC_EINVAL each time, and the context still decodes afterwards"""
    import ctypes as C
    from ecc_ldpc_amd._lib import lib
    c = load("jpl.1024.4.5")
    _, llr = c.frames(9, 3.5, 77)
    for path, sched in (("fused", "flooding"), ("flood", "flooding"), ("auto", "layered")):
        dec = hip.Decoder(c.hip_code(hip), "min", "f32", 8, path=path, schedule=sched)
        with pytest.raises(hip.LdpcError) as e:
            dec.decode_batch(llr.astype(np.float32), 10)                 # 9 frames into a context for 8
        assert e.value.code == -1
        with pytest.raises(hip.LdpcError) as e:
            dec.decode_batch(llr[:4].astype(np.float32), -1)
        assert e.value.code == -1
        bits = np.zeros((4, c.N), np.uint8)
        f32 = np.ascontiguousarray(llr[:4], np.float32)
        assert lib().ldpc_decode_batch(dec._h, 10, 4, None, bits.ctypes.data_as(C.POINTER(C.c_uint8)), None, None) == -1
        assert lib().ldpc_decode_batch(dec._h, 10, 4, f32.ctypes.data_as(C.POINTER(C.c_float)), None, None, None) == -1
        assert lib().ldpc_decode_batch(None, 10, 4, f32.ctypes.data_as(C.POINTER(C.c_float)), bits.ctypes.data_as(C.POINTER(C.c_uint8)), None, None) == -1
        b0 = dec.decode_batch(np.zeros((0, c.N), np.float32), 10)        # an empty batch is not an error
        assert b0[0].shape == (0, c.N)
        good = dec.decode_batch(llr[:8].astype(np.float32), 30)
        assert good[2].all()
