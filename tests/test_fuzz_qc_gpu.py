"""GPU: a few seeds of tools/fuzz_qc.py inside the suite -- random single-circulant protographs (circulant sizes 48, 54, 160:
not powers of two), the on-chip kernel specialised for each at run time against the HBM flood path, and the two kernels
of the layered schedule against each other, f32, bit for bit.  The tool itself runs hundreds of codes
(profiles/r02_fuzz_qc.txt)."""
import importlib.util
import os

import numpy as np
import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_qc", os.path.join(ROOT, "tools", "fuzz_qc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [1003, 1004, 1008, 1011, 1019])
def test_random_protograph(hip, seed, monkeypatch):
    c = _tool().random_code(seed)
    assert c is not None and c.sz in (48, 54, 160)
    code = c.hip_code(hip)
    F = 96
    llr = np.concatenate([c.frames(F // 2, 2.5, seed)[1], c.frames(F // 2, 6.0, seed + 1)[1]]).astype(np.float32)
    for rule in ("min", "tanh"):
        fused = hip.Decoder(code, rule, "f32", F, path="fused")
        flood = hip.Decoder(code, rule, "f32", F, path="flood")
        assert fused.kernel_name.startswith("ldpc_jit_split") or "fused" in fused.kernel_name
        a, b = fused.decode_batch(llr, 30), flood.decode_batch(llr, 30)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (c.name, rule)
        assert 0 < a[2].sum() < F or a[1].max() > 1           # the SNR mix makes some frames work for it
        qc = hip.Decoder(code, rule, "f32", F, schedule="layered", path="flood")
        monkeypatch.setenv("LDPC_LAYERED_QC", "0")
        bm = hip.Decoder(code, rule, "f32", F, schedule="layered", path="flood")
        monkeypatch.delenv("LDPC_LAYERED_QC")
        a, b = qc.decode_batch(llr, 20), bm.decode_batch(llr, 20)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (c.name, rule, "layered")
        if rule == "min":                                     # ... and the on-chip layered kernel, specialised at run time
            on = hip.Decoder(code, rule, "f32", F, schedule="layered")
            assert on.path == "fused" and on.kernel_name.startswith("ldpc_jit_layered_"), on.kernel_name
            assert all(np.array_equal(x, y) for x, y in zip(on.decode_batch(llr, 20), a)), (c.name, "on-chip layered")
            pk = hip.Decoder(code, rule, "f16pk", F).decode_batch(llr, 30)
            from oracle import emulate_f16 as em
            e = em.decode_minsum_pk16(c.graph, llr[:9], 30)
            assert np.array_equal(pk[0][:9], e[0]) and np.array_equal(pk[1][:9], e[1])
