"""CPU: the C-ABI library loads, exports every symbol include/ldpc_hip.h declares, builds graphs on
the host, and FAILS LOUDLY (no fallback) when no GPU is bound."""
import ctypes
import os
import re

import numpy as np
import pytest

import ecc_ldpc_amd as E
from tests.helpers import ROOT, load


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ldpc_hip.h")).read()
    declared = set(re.findall(r"\b(ldpc_[a-z0-9_]+)\s*\(", hdr)) - {"ldpc_code", "ldpc_ctx"}
    assert declared == set(E.ABI_SYMBOLS), declared ^ set(E.ABI_SYMBOLS)
    L = ctypes.CDLL(E.SO_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert E.lib().ldpc_abi_version() >= 2


def test_qc_graph_expansion_matches_oracle_parser():
    for name in ("jpl.1024.4.5", "jpl.4096.4.5"):
        c = load(name)
        code = c.hip_code(E)
        assert (code.M, code.N, code.E) == (c.M, c.N, c.E)
        rp, ci = code.csr()
        assert np.array_equal(rp, c.graph.row_ptr) and np.array_equal(ci, c.graph.col_idx)


def test_csr_validation():
    with pytest.raises(E.LdpcError) as e:
        E.Code.from_csr([0, 2], [1, 1], 3)  # not strictly ascending
    assert e.value.code == -1
    with pytest.raises(E.LdpcError):
        E.Code.from_csr([0, 1], [5], 3)  # column out of range
    with pytest.raises(E.LdpcError):
        E.Code.from_qc(8, np.array([[9]]))  # rotation >= sz


@pytest.mark.skipif(E.lib().ldpc_device_count() > 0, reason="this check is for GPU-less hosts")
def test_no_gpu_means_error_not_fallback():
    with pytest.raises(E.LdpcError) as e:
        E.init(0)
    assert e.value.code == -4
    c = load("moon.7.13").hip_code(E)
    with pytest.raises(E.LdpcError) as e:
        E.Decoder(c, "min", "f32", 4)
    assert e.value.code == -4


def test_header_is_c99_and_a_c_program_links(tmp_path):
    import subprocess
    exe = tmp_path / "abi_c_smoke"
    src = os.path.join(ROOT, "tests", "abi_c_smoke.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), src,
                           "-o", str(exe), E.SO_PATH, "-Wl,-rpath," + os.path.dirname(E.SO_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip().endswith("ok")


def test_qc_layer_order_proposes_independent_runs():
    """ldpc_qc_layer_order (host code, no GPU): a permutation of the block rows in which runs of `run` consecutive rows share no block
    column -- on the shipped DVB-S2-shaped matrix (already written in such an order: 22 runs of four), on its rows shuffled, on an
    accumulator chain (every row shares a column with the next: pairs 0/2, 1/3 ... exist, fours too), and argument checks."""
    from oracle import formats
    from tests.helpers import CODES
    sz, rows = formats.read_qc(open(os.path.join(CODES, "dvbs2like.64800.1.2", "H.q")).read())
    off = np.asarray(formats.qc_offsets(sz, rows), dtype=np.int32)
    assert sz == 360 and off.shape == (90, 180)

    def runs_ok(o, perm, run, full):
        assert sorted(perm.tolist()) == list(range(o.shape[0]))
        hit = o[perm] >= 0
        for i in range(full):
            blk = hit[i * run:(i + 1) * run]
            assert blk.sum(0).max() <= 1, i

    for run in (2, 4):
        perm, full = E.Code.qc_layer_order(off, run)
        assert full >= (45 if run == 2 else 22)
        runs_ok(off, perm, run, full)
    shuffled = off[np.random.default_rng(1).permutation(90)]
    perm, full = E.Code.qc_layer_order(shuffled, 4)
    assert full >= 20
    runs_ok(shuffled, perm, 4, full)
    chain = -np.ones((12, 16), np.int32)                     # information part: a private column per row; accumulator: i and i - 1
    for i in range(12):
        chain[i, i % 4] = 1 + i; chain[i, 4 + i] = 0
        if i:
            chain[i, 4 + i - 1] = 0
    perm, full = E.Code.qc_layer_order(chain, 2)
    runs_ok(chain, perm, 2, full)
    assert full == 6
    with pytest.raises(E.LdpcError):
        E.Code.qc_layer_order(chain, 9)
