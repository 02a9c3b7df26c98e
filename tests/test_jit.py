"""Run-time specialised QC kernels (jit.cc): generalising the four-wave split kernel from the two shipped matrices to
ANY single-circulant quasi-cyclic H (what the reference's QC decoders accept, Fast/Arraylet.hs:68-79).
CPU: the generated translation unit and its compilation with hiprtc (no GPU needed; also warms the on-disk cache the
GPU tests then hit).  GPU: the compiled kernels against the oracle."""
import os
import re

import numpy as np
import pytest

import ecc_ldpc_amd as E
from oracle import oracle
from tests.helpers import SYNTHETIC_NAMES, lam_tolerance, synthetic, iters_agree


@pytest.mark.parametrize("name", SYNTHETIC_NAMES)
def test_generated_plan_describes_the_code(name):
    c = synthetic(name)
    src = c.hip_code(E).jit_source("min")
    g = lambda pat: [int(x) for x in re.search(pat + r"\[[^\]]*\] = \{([^}]*)\}", src).group(1).split(",")]
    deg, ebeg, own, rot, bc = g("deg_"), g("ebeg_"), g("own_"), g("rot"), g("bc")
    mask = c.offsets >= 0
    assert deg == mask.sum(1).tolist() and ebeg == [0] + np.cumsum(mask.sum(1)).tolist()
    assert rot == c.offsets[mask].tolist() and bc == np.nonzero(mask)[1].tolist()           # block-row-major, ascending column
    np_ = int(re.search(r"NP = (\d+)", src).group(1))
    threads = int(re.search(r"__launch_bounds__\((\d+)\)", src).group(1))
    v = (max(c.sz, 64) + 63) // 64 * 64        # threads of one wave group: the circulant size rounded up to whole waves
    assert threads == np_ * v <= 1024 and set(own) == set(range(np_))
    per_group = [sum(d for d, o in zip(deg, own) if o == p) for p in range(np_)]
    assert max(per_group) <= 80                                                                # register budget
    if name == "irregular-20x30-sz64":
        assert np_ >= 2
    if name == "jpl4096-permuted":
        assert np_ == 2 and per_group == [78, 78]


def test_unsupported_shapes_say_why():
    with pytest.raises(E.LdpcError) as e:
        E.Code.from_qc(8, np.array([[1, 2], [3, 4]], np.int32)).jit_source("min")           # too small to fill a wave
    assert e.value.code == -5 and "below 16" in str(e.value)
    with pytest.raises(E.LdpcError) as e:
        E.Code.from_qc(64, np.array([[1, 2, -1], [3, 4, -1]], np.int32)).jit_source("min")   # empty block column
    assert "empty block column" in str(e.value)
    with pytest.raises(E.LdpcError):
        synthetic("regular36-sz128").hip_code(E).jit_source("min", "f64")                     # f32 only


@pytest.mark.parametrize("name", SYNTHETIC_NAMES)
def test_hiprtc_compiles_without_a_gpu_and_caches(name):
    code = synthetic(name).hip_code(E)
    for variant in ("min", "tanh"):
        kname, cached, sec = code.jit_prepare(variant)
        assert kname.startswith("ldpc_jit_split_") and f"sz{synthetic(name).sz}" in kname
        path = [f for f in os.listdir(E.lib().ldpc_jit_cache_dir().decode()) if f.startswith(kname)]
        assert len(path) >= 1 and all(f.endswith(".hsaco") for f in path)   # (one per version of the device headers)
        k2, cached2, _ = code.jit_prepare(variant)
        assert k2 == kname and cached2


def test_both_compiler_routes_build_the_same_kernel(tmp_path, monkeypatch):
    """hipcc --genco (a child process; default when the tool chain is installed) and hiprtc (in-process)"""
    code = synthetic("small-2x4-sz32").hip_code(E)
    names = {}
    for route in ("hiprtc", "hipcc"):
        monkeypatch.setenv("LDPC_JIT_COMPILER", route)
        monkeypatch.setenv("LDPC_JIT_NOCACHE", "1")
        names[route], cached, sec = code.jit_prepare("min")
        assert not cached and sec > 0.5
    assert names["hiprtc"] == names["hipcc"]


_CACHE_CHILD = """
import os, sys, threading
sys.path.insert(0, {root!r})
import ecc_ldpc_amd as E
from tests.helpers import synthetic
code = synthetic("small-2x4-sz32").hip_code(E)
mode = sys.argv[1]
if mode == "threads":     # replicas created from several threads of one process compile the same kernel at once
    res = []
    ts = [threading.Thread(target=lambda: res.append(code.jit_prepare("min"))) for _ in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print(sum(1 for r in res if not r[1]), len(res))
else:
    print(code.jit_prepare("min"))
print(sorted(os.listdir(os.environ["LDPC_JIT_CACHE"])))
"""


def _cache_child(tmp_path, mode, **env):
    import subprocess
    import sys
    from tests.helpers import ROOT
    e = dict(os.environ, LDPC_JIT_CACHE=str(tmp_path / "cache"), **env)
    os.makedirs(e["LDPC_JIT_CACHE"], exist_ok=True)
    p = subprocess.run([sys.executable, "-c", _CACHE_CHILD.format(root=ROOT), mode], env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    return p.stdout.strip().splitlines()


def test_cache_is_bypassed_by_experiments_and_keyed_by_compiler_route(tmp_path):
    """an LDPC_JIT_EXTRA_OPTS or LDPC_JIT_NOCACHE=1 build must not become the object later runs load, and an
    object from the in-process compiler must not stand in for the tool chain's."""
    assert _cache_child(tmp_path, "one", LDPC_JIT_NOCACHE="1")[-1] == "[]"
    assert _cache_child(tmp_path, "one", LDPC_JIT_EXTRA_OPTS="-DLDPC_EXPERIMENT=1")[-1] == "[]"
    files = eval(_cache_child(tmp_path, "one", LDPC_JIT_COMPILER="hiprtc")[-1])
    assert len(files) == 1 and files[0].endswith(".rtc.hsaco")
    out = _cache_child(tmp_path, "one")                      # default route = tool chain: compiles its own, does not load the .rtc one
    assert "False" in out[0]
    files = eval(out[-1])
    assert len(files) == 2 and sum(f.endswith(".rtc.hsaco") for f in files) == 1 and not any(".tmp" in f for f in files)
    assert "True" in _cache_child(tmp_path, "one")[0]          # and now it is cached


def test_threads_of_one_process_compile_a_kernel_once(tmp_path):
    out = _cache_child(tmp_path, "threads")
    compiled, total = map(int, out[0].split())
    assert (compiled, total) == (1, 4)
    files = eval(out[-1])
    assert len(files) == 1 and files[0].endswith(".hsaco") and ".tmp" not in files[0]


def test_tool_chain_child_runs_without_a_profilers_preload(tmp_path):
    """under rocprofv3 the environment carries LD_PRELOAD / HSA_TOOLS_LIB / ROCP_*; hipcc (which exec's clang)
    must not inherit them.  A fake hipcc records its environment and writes a dummy object."""
    fake = tmp_path / "hipcc"
    fake.write_text("#!/bin/sh\nenv > %s\nwhile [ $# -gt 1 ]; do [ \"$1\" = -o ] && out=$2; shift; done\nprintf 'x' > $out\n" % (tmp_path / "env.txt"))
    fake.chmod(0o755)
    _cache_child(tmp_path, "one", HIPCC=str(fake), LDPC_JIT_COMPILER="hipcc", LDPC_JIT_NOCACHE="1", ROCP_FAKE_TOOL="x.so",
                 HSA_TOOLS_REPORT_LOAD_FAILURE="1", ROCPROF_FAKE_SETTING="1", LD_PRELOAD="", LDPC_KEEP_ME="1")
    seen = (tmp_path / "env.txt").read_text()
    assert "LDPC_KEEP_ME=1" in seen
    for name in ("LD_PRELOAD=", "HSA_TOOLS_", "ROCP_FAKE_TOOL", "ROCPROF_FAKE_SETTING"):   # (names with the prefixes rocprofv3 sets, harmless here)
        assert name not in seen, name


@pytest.mark.gpu
def test_a_hiprtc_compiled_kernel_runs_correctly(hip, tmp_path, monkeypatch):
    """the in-process route end to end on the GPU (the other tests hit whatever the default route cached)"""
    monkeypatch.setenv("LDPC_JIT_COMPILER", "hiprtc")
    monkeypatch.setenv("LDPC_JIT_NOCACHE", "1")
    c = synthetic("small-2x4-sz32")
    _, llr = c.frames(32, 4.0, seed=5)
    dec = hip.Decoder(c.hip_code(hip), "min", "f32", 32)
    assert dec.kernel_name.startswith("ldpc_jit_split_")
    bits, its, conv = dec.decode_batch(llr.astype(np.float32), 30)
    ob, oi, oc = oracle.decode_batch(c.graph, "min", 30, llr, nthreads=4)
    assert np.array_equal(bits, ob) and np.array_equal(conv, oc) and np.array_equal(its, oi)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SYNTHETIC_NAMES)
def test_jit_kernels_match_the_oracle(hip, name):
    c = synthetic(name)
    code = c.hip_code(hip)
    F = 24 if c.N > 4000 else 64
    llr = np.concatenate([c.frames(F // 2, db, 700 + i)[1] for i, db in enumerate((2.0, 4.0) if c.N > 1000 else (3.0, 6.0))])
    if c.sz & (c.sz - 1):
        assert hip.Decoder(code, "min", "f32", 4).kernel_name.startswith("ldpc_jit_split_")     # sizes that are not powers of two too
    for variant in ("min", "tanh"):
        dec = hip.Decoder(code, variant, "f32", F)
        assert dec.path == "fused" and dec.kernel_name.startswith("ldpc_jit_split_"), dec.kernel_name
        bits, its, conv = dec.decode_batch(llr.astype(np.float32), 40)
        ob, oi, oc = oracle.decode_batch(c.graph, variant, 40, llr, nthreads=8)
        assert np.array_equal(bits, ob) and np.array_equal(conv, oc), (name, variant)
        assert iters_agree(its, oi)
        # same arithmetic in the same order as the flood path: identical, iteration counts included
        fb, fi, fc = hip.Decoder(code, variant, "f32", F, path="flood").decode_batch(llr.astype(np.float32), 40)
        assert np.array_equal(bits, fb) and np.array_equal(its, fi) and np.array_equal(conv, fc)
        # ragged batch + single frame + max_iters beyond the packed-result limit of the built-in instances
        b1, i1, c1 = dec.decode_batch(llr[:5].astype(np.float32), 600)
        o1 = oracle.decode_batch(c.graph, variant, 600, llr[:5], nthreads=5)
        assert np.array_equal(b1, o1[0]) and np.array_equal(c1, o1[2])
        print(f"{name} {variant}: {dec.kernel_name} {int(conv.sum())}/{F} converged, iteration counts {100 * (its == oi).mean():.0f}% identical")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["jpl4096-permuted", "ira-12x24-sz64", "small-2x4-sz32", "wimax-12x24-sz96", "dvbs2short-20x45-sz360"])
def test_jit_kernels_teacher_forced(hip, name):
    c = synthetic(name)
    code = c.hip_code(hip)
    _, llr = c.frames(4, 3.0, seed=801)
    for variant in ("min", "tanh"):
        dec = hip.Decoder(code, variant, "f32", 64)
        states = []
        for f in range(len(llr)):
            o = oracle.decode(c.graph, variant, 12, llr[f], trace=True)
            ne = np.zeros(c.E)
            for n in range(o["iters"]):
                states.append((llr[f], o["trace_lam"][n], ne, o["trace_ne"][n], o["trace_lam"][n + 1]))
                ne = o["trace_ne"][n]
        states = states[:64]
        ne2, lam2, _ = dec.debug_step(np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), np.stack([s[2] for s in states]))
        worst = 0.0
        for i, s in enumerate(states):
            tol_lam, tol_ne = lam_tolerance(c.graph, s[3], s[4])
            assert (np.abs(ne2[i] - s[3]) <= tol_ne).all() and (np.abs(lam2[i] - s[4]) <= tol_lam).all(), (name, variant, i)
            worst = max(worst, (np.abs(lam2[i] - s[4]) / np.maximum(1, np.abs(s[4]))).max())
        print(f"{name} {variant}: worst teacher-forced relative LLR error {worst:.2e} over {len(states)} turns")
