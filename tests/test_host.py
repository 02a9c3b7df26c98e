"""The C++ host mirror: matrix ingest against the oracle-side parsers (CPU), the mkLDPC record and the
device frame source (GPU)."""
import os

import numpy as np
import pytest

import ecc_ldpc_amd as E
from oracle import formats, oracle
from tests.helpers import CODES, load, iters_agree


def test_loaders_match_the_restated_parsers():
    for name in ("jpl.1024.4.5", "jpl.4096.4.5"):
        for which in ("H", "G"):
            m = E.Matrix.load(CODES, f"{name}/{which}")
            sz, rows = formats.read_qc(open(os.path.join(CODES, name, f"{which}.q")).read())
            assert (m.sz, m.block_rows, m.block_cols) == (sz, len(rows), len(rows[0]))
            assert np.array_equal(m.dense(), formats.qc_expand(sz, rows))
            if which == "H":
                assert np.array_equal(m.qc_offsets(), formats.qc_offsets(sz, rows))
            else:
                with pytest.raises(E.LdpcError) as e:  # G blocks are dense circulants: Arraylet.hs:72-73 errors
                    m.qc_offsets()
                assert e.value.code == -5
    mo = E.Matrix.load(CODES, "moon.7.13/H")
    assert mo.sz == 0 and np.array_equal(mo.dense(), load("moon.7.13").H)
    mk = E.Matrix.load_mackay(os.path.join(CODES, "1920.1280.3.303"))
    assert np.array_equal(mk.dense(), load("1920.1280.3.303").H)


def test_loader_search_order_and_errors(tmp_path):
    d = tmp_path / "c" / "x"
    d.mkdir(parents=True)
    (d / "H.m").write_text("1 0 1\n0 1 1\n")
    m = E.Matrix.load(str(tmp_path / "c"), "x/H")
    assert np.array_equal(m.dense(), [[1, 0, 1], [0, 1, 1]])
    (d / "H.alist").write_text("2 3\n2 2\n2 2\n1 1 2\n1 3\n2 3\n1\n2\n1 2\n")
    m = E.Matrix.load(str(tmp_path / "c"), "x/H")  # .alist is tried before .m (Loader.hs:53-57)
    assert np.array_equal(m.dense(), [[1, 0, 1], [0, 1, 1]])
    (d / "H.q").write_text("4\n1 0\n2 8\n")       # .q before both
    m = E.Matrix.load(str(tmp_path / "c"), "x/H")
    assert (m.rows, m.cols, m.sz) == (8, 8, 4)
    assert np.array_equal(m.dense(), formats.qc_expand(4, [[1, 0], [2, 8]]))
    with pytest.raises(E.LdpcError) as e:
        E.Matrix.load(str(tmp_path / "c"), "nope/H")
    assert e.value.code == -8
    (d / "H.q").write_text("4\n1 x\n")
    with pytest.raises(E.LdpcError) as e:
        E.Matrix.load(str(tmp_path / "c"), "x/H")
    assert e.value.code == -7
    (d / "H.q").write_text("4\n16 0\n")  # bit above the cycle size
    with pytest.raises(E.LdpcError):
        E.Matrix.load(str(tmp_path / "c"), "x/H")


def test_code_name_grammar_rejections():
    for bad in ("ldpc/ldpc-zero/jpl.1024.4.5/50", "ldpc/model-3-4/jpl.1024.4.5/50", "bpsk", "ldpc/hip-minsum/jpl.1024.4.5/x", "ldpc/hip-minsum/jpl.1024.4.5/50/4"):
        with pytest.raises(E.LdpcError) as e:
            E.ECC(CODES, bad)
        assert e.value.code == -8, bad  # not ours: the factory's `_ -> return []`


def test_reference_decoder_names_are_recognised_without_a_gpu():
    """the alias table (host.cc kAliases) is part of the name grammar: on a box without a GPU a reference name gets past
    it and fails at the device (no CPU fallback), an unknown name fails at the grammar"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    for name in ("reference", "min", "sparse", "sparsemin", "arraylet", "arraylet-min", "arraylet-cm", "cuda-arraylet1", "cuda-arraylet2",
                 "two-arrays", "hip-minsum", "hip-tanh-layered", "hip-minsum-bool-f64"):
        with pytest.raises(E.LdpcError) as e:
            E.ECC(CODES, f"ldpc/{name}/jpl.1024.4.5/50/4/5")
        assert e.value.code == -4, (name, e.value.code, str(e.value))     # LDPC_ENODEVICE, not LDPC_ENOTFOUND
    with pytest.raises(E.LdpcError) as e:
        E.ECC(CODES, "ldpc/arraylet-max/jpl.1024.4.5/50/4/5")
    assert e.value.code == -8


@pytest.mark.gpu
def test_ecc_record_mirrors_mkLDPC(hip):
    ecc = hip.ECC(CODES, "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", max_batch=4)
    assert ecc.name == "ldpc/hip-minsum/jpl.1024.4.5/50/4/5"
    assert (ecc.message_length, ecc.codeword_length, ecc.unpunctured_length) == (1024, 1280, 1408)
    e2 = hip.ECC(CODES, "ldpc/hip-tanh-f64/jpl.1024.4.5/20", max_batch=4)  # no rate: k / cols(H) = 8/11
    assert e2.name == "ldpc/hip-tanh-f64/jpl.1024.4.5/20/8/11" and e2.codeword_length == 1408
    c = load("jpl.1024.4.5")
    rng = np.random.default_rng(3)
    msg = rng.integers(0, 2, 1024).astype(np.uint8)
    cw = ecc.encode(msg)
    assert np.array_equal(cw, c.encode(msg)[:1280])  # systematic, punctured tail dropped (Utils.hs:61)
    from oracle import channel
    llr = channel.frames(c.encode(msg)[None, :], 3.5, 1024, 1280, 1408, seed=5)[0]
    out, ok = ecc.decode(llr[:1280])
    o = oracle.decode(c.graph, "min", 50, llr)
    assert ok and np.array_equal(out, o["bits"][:1024])
    # "-bool": the same matrix taken as a plain Boolean matrix (the Haskell binding's `Matrix Bool` codes): CSR graph, same answers
    eb = hip.ECC(CODES, "ldpc/hip-minsum-bool/jpl.1024.4.5/50/4/5", max_batch=4)
    assert len(eb.code.layers()) == 384 + 1 and "csr" in eb.decoder.kernel_name
    outb, okb = eb.decode(llr[:1280])
    assert okb and np.array_equal(outb, out)
    # the reference's own decoder names are aliases (host.cc kAliases): same answers, and the ECC keeps the name it was given
    ea = hip.ECC(CODES, "ldpc/arraylet-min/jpl.1024.4.5/50/4/5", max_batch=4)
    assert ea.name == "ldpc/arraylet-min/jpl.1024.4.5/50/4/5"
    outa, oka = ea.decode(llr[:1280])
    assert oka and np.array_equal(outa, out) and ea.decoder.kernel_name == ecc.decoder.kernel_name
    em = hip.ECC(CODES, "ldpc/min/jpl.1024.4.5/50/4/5", max_batch=4)          # Reference.Min: Boolean H
    assert "csr" in em.decoder.kernel_name and np.array_equal(em.decode(llr[:1280])[0], out)
    ecm = hip.ECC(CODES, "ldpc/arraylet-cm/jpl.1024.4.5/20/4/5", max_batch=2)  # StableDiv numerics: f64 only
    ref_cm = hip.ECC(CODES, "ldpc/hip-tanh-cm-f64/jpl.1024.4.5/20/4/5", max_batch=2)
    assert np.array_equal(ecm.decode(llr[:1280])[0], ref_cm.decode(llr[:1280])[0])
    assert ecm.decoder.kernel_name == ref_cm.decoder.kernel_name
    mo = hip.ECC(CODES, "ldpc/hip-tanh/moon.7.13/20", max_batch=2)
    assert (mo.message_length, mo.codeword_length) == (7, 20)
    m = load("moon.7.13")
    msg = np.array([1, 0, 1, 1, 0, 0, 1], np.uint8)
    assert np.array_equal(mo.encode(msg), m.encode(msg))


@pytest.mark.gpu
def test_device_frame_source(hip):
    import torch
    ecc = hip.ECC(CODES, "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", max_batch=512)
    dev = torch.device("cuda", 0)
    B, N, k, n_tx = 512, 1408, 1024, 1280
    llr = torch.empty((B, N), dtype=torch.float32, device=dev)
    msg = torch.empty((B, k), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ecc.sim.generate(42, 1000, B, 3.0, llr.data_ptr(), msg.data_ptr(), None)
    torch.cuda.synchronize()
    l1, m1 = llr.cpu().numpy().copy(), msg.cpu().numpy().copy()
    # counter-based: the same (seed, frame id) gives the same frame wherever it is generated
    ecc.sim.generate(42, 1000 + 100, B - 100, 3.0, llr.data_ptr(), msg.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(llr.cpu().numpy()[: B - 100], l1[100:]) and np.array_equal(msg.cpu().numpy()[: B - 100], m1[100:])
    assert (l1[:, n_tx:] == 0).all()                      # punctured tail (Utils.hs:55)
    assert 0.45 < m1.mean() < 0.55
    c = load("jpl.1024.4.5")
    cws = np.stack([c.encode(m) for m in m1[:16]])
    s2 = 1.0 / (2 * 0.8 * 10 ** 0.3)
    y = l1[:16, :n_tx] * s2 / 2.0                          # LLR = 2y/sigma^2
    noise = y - (2.0 * cws[:, :n_tx] - 1.0)
    assert abs(noise.mean()) < 0.02 and abs(noise.var() / s2 - 1.0) < 0.05
    allnoise = (l1[:, :n_tx] * s2 / 2.0) - np.sign(l1[:, :n_tx] * 0 + 1) * 0  # (sanity only)
    assert np.isfinite(allnoise).all()
    # tally kernel agrees with a host count
    bits = torch.zeros((B, N), dtype=torch.uint8, device=dev)
    ecc.sim.generate(42, 1000, B, 3.0, llr.data_ptr(), msg.data_ptr(), None)
    bits[:, :k] = msg
    bits[3, 5] ^= 1
    bits[7, 0:4] ^= 1
    iters = torch.full((B,), 7, dtype=torch.int32, device=dev)
    tally = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ecc.sim.tally(B, bits.data_ptr(), iters.data_ptr(), tally.data_ptr(), None)
    torch.cuda.synchronize()
    assert tally.tolist() == [B, 2, 5, 7 * B]


@pytest.mark.gpu
def test_standalone_mackay_matrix_record(hip):
    """codes/1920.1280.3.303 is a single MacKay-order file without a generator: the record takes
    k = cols - rows, frames are all-zero codewords, decode works, encode says so."""
    ecc = hip.ECC(CODES, "ldpc/hip-tanh/1920.1280.3.303/50", max_batch=64)
    assert (ecc.message_length, ecc.codeword_length, ecc.unpunctured_length) == (640, 1920, 1920)
    assert ecc.name == "ldpc/hip-tanh/1920.1280.3.303/50/1/3" and ecc.decoder.path == "fused"
    c = load("1920.1280.3.303")
    _, llr = c.frames(8, 3.0, seed=31)
    for f in range(8):
        out, ok = ecc.decode(llr[f])
        o = oracle.decode(c.graph, "tanh", 50, llr[f])
        assert ok and np.array_equal(out, o["bits"][:640])
    with pytest.raises(hip.LdpcError) as e:
        ecc.encode(np.zeros(640, np.uint8))
    assert e.value.code == -5


@pytest.mark.gpu
def test_cli_rows(hip, capsys):
    """The ecc-ldpc-like CLI (reference usage: main/Main.hs:38-40, NOTES.txt:2-3)."""
    from ecc_ldpc_amd import cli
    rc = cli.main(["3", "4.5", "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", "ldpc/ldpc-zero/jpl.1024.4.5/50", "-m2048", "-b1024"])   # (ldpc-zero: not ours)
    out = capsys.readouterr()
    rows = [l.split() for l in out.out.strip().splitlines()]
    assert rc == 0 and len(rows) == 2 and "no such code" in out.err
    assert rows[0][1] == "ldpc/hip-minsum/jpl.1024.4.5/50/4/5" and int(rows[0][3]) == 2048
    assert float(rows[0][5]) > float(rows[1][5]) and float(rows[1][5]) == 0.0  # BER falls with Eb/N0; 4.5 dB is clean


@pytest.mark.gpu
def test_native_cli_matches_python_cli(hip, capsys):
    """ecc-ldpc-hip (C++ over the C ABI, csrc/cli_main.cc) and the Python command line drive the same library with
    the same frame source: identical frame / bit-error counts, row for row."""
    import subprocess
    from ecc_ldpc_amd import cli
    from ecc_ldpc_amd.build import CLI
    args = ["3", "3.5", "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", "ldpc/hip-tanh/1920.1280.3.303/50", "ldpc/nonsense/jpl.1024.4.5/50", "-m8192", "-b4096"]
    p = subprocess.run([CLI] + args + ["-c" + CODES], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "nonsense" in p.stderr                      # unknown decoder: skipped with a note, as the Python CLI does
    native = [l.split() for l in p.stdout.splitlines() if l.strip()]
    assert cli.main(args) == 0
    py = [l.split() for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(native) == len(py) == 4
    for a, b in zip(native, py):
        assert a[1:6] == b[1:6] and a[-1] == b[-1]     # name, Eb/N0, frames, bit errors, BER ... [path]


@pytest.mark.gpu
def test_contexts_release_their_memory(hip):
    """Create / use / destroy in a loop (every path, host entry points included so that the staging, latency-path and
    zero-copy buffers get allocated): the device must end up with the memory it started with."""
    import torch
    c = load("jpl.1024.4.5")
    mk = load("1920.1280.3.303")
    _, llr = c.frames(40, 3.0, seed=1)
    _, llr_mk = mk.frames(40, 2.0, seed=2)
    pin = hip.PinnedArray((40, c.N), np.float32); pin.array[:] = llr
    pout = hip.PinnedArray((40, c.N), np.uint8)

    def cycle():
        for path in ("fused", "flood"):
            d = hip.Decoder(c.hip_code(hip), "min", "f32", 64, path=path)
            d.decode_batch(llr.astype(np.float32), 10)
            d.decode_batch(llr[:3].astype(np.float32), 10)                 # latency path
            d.decode_batch(pin.array, 10, out_bits=pout.array)             # zero-copy path
            d.decode_batch(llr, 5, want_lam=True)
            d.close(); d.code.close()
        d = hip.Decoder(mk.hip_code(hip), "tanh", "f32", 64)
        d.decode_batch(llr_mk.astype(np.float32), 10)
        d.close(); d.code.close()
        e = hip.ECC(CODES, "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", max_batch=64)
        e.decode(llr[0][: e.codeword_length])
        e.close()

    cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(15):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)


@pytest.mark.gpu
def test_multi_circulant_q_file_decodes_through_the_csr_route(hip, tmp_path):
    """SURVEY.md section 8 row f3's named case: a `.q` whose blocks hold MORE than one circulant.  The reference's quasi-cyclic
    decoders reject it (Fast/Arraylet.hs:72-73 "non-powers of two initial value"); here the loader keeps the file, the
    rotation-table route says so, and the record decodes it as a plain graph (what `Matrix Bool` decoders of the reference get)
    -- against the oracle on the expansion of QuasiCyclic.hs:19-25."""
    sz, R, Cb = 40, 3, 7                                   # not a power of two either
    rng = np.random.default_rng(8)
    rows = [[0] * Cb for _ in range(R)]
    for br in range(R):
        for bc in range(Cb):
            if rng.random() < 0.75:
                ks = rng.choice(sz, size=int(rng.integers(1, 4)), replace=False)     # 1..3 circulants in one block
                rows[br][bc] = int(sum(1 << int(k) for k in ks))
    rows[0][0] = (1 << 3) | (1 << 17) | (1 << 39)
    d = tmp_path / "codes" / "multi"
    d.mkdir(parents=True)
    (d / "H.q").write_text(f"{sz}\n" + "\n".join(" ".join(str(v) for v in r) for r in rows) + "\n")
    m = hip.Matrix.load(str(tmp_path / "codes"), "multi/H")
    H = formats.qc_expand(sz, rows)
    assert (m.rows, m.cols, m.sz) == (R * sz, Cb * sz, sz) and np.array_equal(m.dense(), H)
    with pytest.raises(hip.LdpcError) as e:
        m.qc_offsets()
    assert e.value.code == -5 and "non-powers of two" in str(e.value)
    g = oracle.Graph.from_dense(H)
    from oracle import channel
    k, N = (Cb - R) * sz, Cb * sz
    llr = channel.frames(np.zeros((24, N), np.uint8), 3.0, k, N, N, 77)
    for nm, variant in (("hip-minsum", "min"), ("hip-tanh", "tanh")):
        ecc = hip.ECC(str(tmp_path / "codes"), f"ldpc/{nm}/multi/40", max_batch=24)   # H only: k = cols - rows, all-zero codewords
        assert (ecc.message_length, ecc.unpunctured_length) == (k, N) and ecc.code.E == g.E
        bits, its, conv = ecc.decoder.decode_batch(llr.astype(np.float32), 40)
        ob, oi, oc = oracle.decode_batch(g, variant, 40, llr, nthreads=4)
        assert np.array_equal(bits, ob) and np.array_equal(conv, oc) and iters_agree(its, oi)
        d64 = hip.Decoder(ecc.code, variant, "f64", 24)
        b64, i64, c64, tr = d64.decode_trace(llr, 40)
        for f in range(0, 24, 5):
            o = oracle.decode(g, variant, 40, llr[f], trace=True)
            assert i64[f] == o["iters"] and np.array_equal(b64[f], o["bits"])
            if variant == "min":
                assert np.array_equal(tr[f, : o["iters"] + 1], o["trace_lam"])
        d64.close(); ecc.close()
