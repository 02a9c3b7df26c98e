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


def test_generator_matches_the_two_representations_the_reference_holds():
    """codes/Gmat.m.gz (dense [I | X], Matlab text) and codes/X.gz (X in the
    reference's alist layout) -- two descriptions of jpl.1024.4.5's generator that the reference's own loader never reads
    for this code (it reads G.q).  This repository's `.q` big-integer reader and circulant expansion (oracle AND the C
    loader of the product) must reproduce them bit for bit: a reference-held pin of the parse, of the rotation direction
    of the quasi-cyclic expansion (shared with H.q) and of both text readers.  And H (from H.q) annihilates the
    reference's dense generator."""
    import gzip
    here = CODES
    dense = np.array([[int(t) for t in l.split()] for l in gzip.open(os.path.join(here, "Gmat.m.gz"), "rt").read().splitlines() if l.strip()], np.uint8)
    assert dense.shape == (1024, 1408) and np.array_equal(dense[:, :1024], np.eye(1024, dtype=np.uint8))
    X = formats.read_alist_reference(gzip.open(os.path.join(here, "X.gz"), "rt").read())
    assert X.shape == (1024, 384) and np.array_equal(X, dense[:, 1024:])
    # oracle side: G.q -> expansion
    sz, rows = formats.read_qc(open(os.path.join(CODES, "jpl.1024.4.5", "G.q")).read())
    assert np.array_equal(formats.qc_expand(sz, rows), X)
    assert np.array_equal(formats.read_matlab_bits("\n".join(" ".join(map(str, r)) for r in dense.tolist())), dense)
    # product side: the C loader (ldpc_matrix_load picks G.q) and its dense view
    import ecc_ldpc_amd as E
    m = E.Matrix.load(CODES, "jpl.1024.4.5/G")
    assert (m.rows, m.cols, m.sz) == (1024, 384, 32) and np.array_equal(m.dense(), X)
    # H . [I | X]^T = 0 over GF(2) with the reference's own dense generator
    c = load("jpl.1024.4.5")
    assert not ((c.H.astype(np.int64) @ dense.T.astype(np.int64)) % 2).any()
