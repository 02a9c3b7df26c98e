"""CPU: the matrix loaders behind the C ABI (host.cc: .q big-integer parser, the reference-order and MacKay-order
alist readers, the Matlab reader) must turn damaged files into an error code + message -- never a crash, a hang or
an absurd allocation.  The reference's own parsers die with a Haskell `error` on such input (Loader.hs:58-81)."""
import os
import zlib

import numpy as np
import pytest

import ecc_ldpc_amd as E
from tests.helpers import CODES

FILES = [("jpl.1024.4.5/H.q", "q"), ("jpl.1024.4.5/H.alist", "alist"), ("moon.7.13/G.alist", "alist"),
         ("moon.7.13/H.alist", "alist"), ("1920.1280.3.303", "mackay")]


def _try(tmp_path, text, kind, tag):
    d = tmp_path / f"case{tag}"
    d.mkdir()
    if kind == "mackay":
        p = d / "m.alist"
        p.write_bytes(text)
        try:
            m = E.Matrix.load_mackay(str(p))
        except E.LdpcError as e:
            return str(e)
    else:
        (d / "x").mkdir()
        (d / "x" / f"H.{kind}").write_bytes(text)
        try:
            m = E.Matrix.load(str(d), "x/H")
        except E.LdpcError as e:
            return str(e)
    rows, cols = m.rows, m.cols
    assert 0 < rows <= 1 << 20 and 0 < cols <= 1 << 20
    m.close()
    return None


@pytest.mark.parametrize("rel,kind", FILES)
def test_damaged_files_are_rejected_or_parsed_never_crash(tmp_path, rel, kind):
    raw = open(os.path.join(CODES, rel), "rb").read()
    rng = np.random.default_rng(zlib.crc32(rel.encode()))
    outcomes = {"error": 0, "ok": 0}
    cases = [b"", b"\n", b"0 0\n", b"-1 5\n", b"99999999999 99999999999\n", raw[: len(raw) // 2], raw[:17], raw + b" 7 7 7",
             raw.replace(b" ", b"  ", 5), raw.replace(b"1", b"x", 3), b"\x00" * 64, b"4294967297 3\n1 2 3\n"]
    for _ in range(40):     # random byte damage, truncations, digit growth
        b = bytearray(raw)
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, len(b)))
            op = int(rng.integers(0, 4))
            if op == 0: b[pos] = int(rng.integers(0, 256))
            elif op == 1: del b[pos:pos + int(rng.integers(1, 50))]
            elif op == 2: b[pos:pos] = bytes(str(int(rng.integers(0, 10 ** 12))), "ascii") + b" "
            else: b = b[:pos]
            if not b: break
        cases.append(bytes(b))
    for i, text in enumerate(cases):
        msg = _try(tmp_path, text, kind, i)
        outcomes["error" if msg else "ok"] += 1
        assert msg is None or len(msg) > 0
    assert outcomes["error"] >= 8          # the plainly broken ones are errors, with a message


def test_qc_descriptor_limits():
    with pytest.raises(E.LdpcError):
        E.Code.from_qc(1 << 20, np.zeros((1 << 6, 1 << 6), np.int32))     # M, N overflow guard
    with pytest.raises(E.LdpcError):
        E.Code.from_qc(8, np.full((2, 2), 8, np.int32))                    # offset == sz


def test_mackay_weights_beyond_header_maxima_are_rejected(tmp_path):
    """A column (row) weight above the header's max must not walk past its slot: 4 columns x 2 rows, maxc = 2,
    but column 1 claims weight 9 (host.cc parse_alist_mackay)."""
    good = b"4 2\n2 4\n2 2 2 2\n4 4\n1 2\n1 2\n1 2\n1 2\n1 2 3 4\n1 2 3 4\n"
    assert _try(tmp_path, good, "mackay", "good") is None
    bad_col = good.replace(b"2 2 2 2\n", b"9 2 2 2\n", 1)
    msg = _try(tmp_path, bad_col, "mackay", "badcol")
    assert msg and "weight 9 of column 1" in msg
    bad_row = good.replace(b"4 4\n1 2", b"4 7\n1 2", 1)
    msg = _try(tmp_path, bad_row, "mackay", "badrow")
    assert msg and "weight 7 of row 2" in msg
    neg = good.replace(b"2 2 2 2\n", b"-1 2 2 2\n", 1)
    assert _try(tmp_path, neg, "mackay", "neg")
