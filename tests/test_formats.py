"""CPU: structural pins for the matrix files."""
import os

import numpy as np

from oracle import formats, oracle
from tests.helpers import CODES, load


def test_jpl1024_alist_equals_expanded_q():
    c = load("jpl.1024.4.5")
    Ha = formats.read_alist_reference(open(os.path.join(CODES, "jpl.1024.4.5", "H.alist")).read())
    assert np.array_equal(Ha, c.H)


def test_generator_orthogonal_to_parity_check():
    for name in ("moon.7.13", "jpl.1024.4.5", "jpl.4096.4.5"):
        c = load(name)
        if c.gq is not None:
            G = formats.qc_expand(c.gq[0], [[int("".join(str(b) for b in blk[::-1]), 2) for blk in row] for row in c.gq[1]])
        else:
            G = c.G
        k = G.shape[0]
        IG = np.concatenate([np.eye(k, dtype=np.int64), G.astype(np.int64)], 1)
        assert IG.shape[1] == c.N  # Utils.hs:43 rows(G)+cols(G) == cols(H)
        assert not ((IG @ c.H.T.astype(np.int64)) % 2).any()


def test_shapes_and_weights():
    c = load("jpl.4096.4.5")
    assert (c.M, c.N, c.E, c.sz) == (1536, 5632, 19968, 128)
    assert sorted(np.unique(c.H.sum(1))) == [3, 18]
    assert (c.offsets >= 0).sum() == 156
    m = load("1920.1280.3.303")
    assert (m.M, m.N, m.E) == (1280, 1920, 5120) and set(np.unique(m.H.sum(1))) == {4}
    mo = load("moon.7.13")
    assert (mo.M, mo.N, mo.E) == (13, 20, 60)


def test_qc_encoder_matches_dense_encoder():
    c = load("jpl.1024.4.5")
    G = formats.qc_expand(c.gq[0], [[int("".join(str(b) for b in blk[::-1]), 2) for blk in row] for row in c.gq[1]])
    rng = np.random.default_rng(0)
    for _ in range(4):
        msg = rng.integers(0, 2, c.k).astype(np.uint8)
        assert np.array_equal(oracle.encode_qc(c.gq[0], c.gq[1], msg), oracle.encode_dense(G, msg))
        assert not ((c.H.astype(np.int64) @ c.encode(msg)) % 2).any()


def test_qc_rejects_multi_circulant_blocks():
    import pytest
    with pytest.raises(ValueError):
        formats.qc_offsets(8, [[3]])  # Fast/Arraylet.hs:72-73
