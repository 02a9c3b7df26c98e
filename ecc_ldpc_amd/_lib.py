"""ctypes binding of libldpc_hip.so (include/ldpc_hip.h).  The library is the product; there is
no Python or CPU fallback: if the shared object is missing or a call fails this module raises."""
from __future__ import annotations

import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = (os.environ.get("LDPC_SO") or None) or os.path.join(HERE, "libldpc_hip.so")  # LDPC_SO: ablation builds (tools/)

OK, EINVAL, ENOMEM, EHIP, ENODEVICE, EUNSUPPORTED, EDEGREE, EFORMAT, ENOTFOUND = 0, -1, -2, -3, -4, -5, -6, -7, -8
TANH, MINSUM, TANH_CM, TANH_CUDA32 = 0, 1, 2, 3
F32, F64, F16, F16PK = 0, 1, 2, 3
PATH_AUTO, PATH_FLOOD, PATH_FUSED = 0, 1, 2
SCHED_FLOODING, SCHED_LAYERED = 0, 1
SUM_REFERENCE, SUM_ARRAYLET, SUM_SPARSE = 0, 1, 2


class CtxConfig(C.Structure):   # ldpc_ctx_config
    _fields_ = [("struct_size", C.c_size_t), ("device", C.c_int), ("variant", C.c_int), ("dtype", C.c_int), ("max_batch", C.c_int),
                ("path", C.c_int), ("schedule", C.c_int), ("sum_order", C.c_int)]

# every symbol include/ldpc_hip.h declares (tests/test_abi.py checks the library exports them all)
ABI_SYMBOLS = [
    "ldpc_init", "ldpc_shutdown", "ldpc_current_device", "ldpc_ctx_create_on", "ldpc_sim_create_on", "ldpc_ctx_create_cfg", "ldpc_ctx_schedule", "ldpc_ctx_code", "ldpc_ctx_max_batch", "ldpc_ctx_device",
    "ldpc_batcher_create", "ldpc_batcher_destroy", "ldpc_batcher_decode_one", "ldpc_batcher_stats",
    "ldpc_ecc_create_replicas", "ldpc_ecc_replicas", "ldpc_ecc_ctx_at", "ldpc_ecc_sim_at", "ldpc_ecc_decode_on", "ldpc_ecc_set_coalescing", "ldpc_ecc_coalescing_stats", "ldpc_code_set_layers", "ldpc_code_layers", "ldpc_qc_layer_order", "ldpc_last_error", "ldpc_last_error_code", "ldpc_abi_version", "ldpc_device_count",
    "ldpc_code_create_qc", "ldpc_code_create_csr", "ldpc_code_destroy", "ldpc_code_dims", "ldpc_code_csr",
    "ldpc_ctx_create", "ldpc_ctx_create_ex", "ldpc_ctx_destroy", "ldpc_ctx_path", "ldpc_ctx_synchronize",
    "ldpc_decode_one", "ldpc_decode_batch", "ldpc_decode_batch_f64", "ldpc_decode_batch_dev",
    "ldpc_decode_batch_f16", "ldpc_decode_batch_dev_f16", "ldpc_sim_generate_f16", "ldpc_decode_batch_dev_packed", "ldpc_decode_batch_packed",
    "ldpc_debug_step", "ldpc_decode_trace",
    "ldpc_host_alloc", "ldpc_host_free",
    "ldpc_ctx_set_timing", "ldpc_ctx_kernel_time", "ldpc_ctx_kernel_name", "ldpc_ctx_kernel_geometry", "ldpc_jit_cache_dir", "ldpc_jit_source", "ldpc_jit_prepare", "ldpc_jit_source_for", "ldpc_jit_prepare_for",
    "ldpc_sim_create", "ldpc_sim_destroy", "ldpc_sim_generate", "ldpc_sim_tally", "ldpc_sim_encode_host",
    "ldpc_sim_create_qc_on", "ldpc_sim_encoder", "ldpc_sim_encode_batch", "ldpc_matrix_qc_words", "ldpc_matrix_rank",
    "ldpc_matrix_load", "ldpc_matrix_load_mackay", "ldpc_matrix_destroy", "ldpc_matrix_info", "ldpc_matrix_dense",
    "ldpc_matrix_qc_offsets", "ldpc_code_from_matrix",
    "ldpc_ecc_create", "ldpc_ecc_destroy", "ldpc_ecc_name", "ldpc_ecc_message_length", "ldpc_ecc_codeword_length",
    "ldpc_ecc_unpunctured_length", "ldpc_ecc_max_iters", "ldpc_ecc_encode", "ldpc_ecc_decode", "ldpc_ecc_ctx",
    "ldpc_ecc_sim", "ldpc_ecc_code",
]


class LdpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libldpc_hip error {code}: {msg}")
        self.code = code


_lib = None

# ---- deterministic teardown.  Every object that owns a native handle registers here; close_all() -- run by atexit,
# i.e. at the START of interpreter finalisation, while the HIP runtime and every module are still intact -- releases
# them in dependency order (records, replicas, frame sources, graphs, page-locked buffers).  After that, and whenever
# the interpreter is finalising, __del__ does nothing: no hipFree / hipStreamDestroy ever runs from a finaliser during
# shutdown, when the order against the HIP runtime's own exit handlers is not defined.
_live = weakref.WeakSet()
_CLOSE_ORDER = {"ECC": 0, "Batcher": 1, "Decoder": 2, "Sim": 3, "Code": 4, "Matrix": 5, "PinnedArray": 6}
_closed_all = False


def _register(obj):
    _live.add(obj)


def _finalizing():
    return _closed_all or sys is None or sys.is_finalizing()


def close_all():
    """Release every live native object now (idempotent).  Called automatically at interpreter exit."""
    for o in sorted(list(_live), key=lambda o: _CLOSE_ORDER.get(type(o).__name__, 9)):
        try:
            o.close()
        except Exception:
            pass


def _at_exit():
    global _closed_all
    close_all()
    _closed_all = True
    if _lib is not None:
        try:
            _lib.ldpc_shutdown()
        except Exception:
            pass


atexit.register(_at_exit)


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64.so; if
    libldpc_hip.so pulled in /opt/rocm's copy first, a later `import torch` would bring up a second
    runtime that sees no GPUs.  So when a torch wheel is installed, load ITS runtime first (by file,
    without importing torch); libldpc_hip.so then binds to the already-loaded libamdhip64.so.N."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH) and "LDPC_SO" not in os.environ:
        # a fresh checkout (built artefacts are not in git): compile the product, never substitute for it
        try:
            import fcntl
            from . import build as _build
            # several ranks of one job may get here together on a fresh checkout: one builds, the others wait
            with open(os.path.join(HERE, ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if not os.path.exists(SO_PATH):
                    print(f"[ecc_ldpc_amd] {SO_PATH} missing: building it with hipcc", file=sys.stderr, flush=True)
                    _build.build(verbose=False)
        except Exception as e:
            raise ImportError(f"{SO_PATH} is missing and could not be built ({e}); run `python ecc_ldpc_amd/build.py` "
                              "(the HIP library is the only decode path; there is no fallback)") from e
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: build it with `python ecc_ldpc_amd/build.py` "
                          "(the HIP library is the only decode path; there is no fallback)")
    _preload_hip_runtime()
    L = C.CDLL(SO_PATH)
    vp, i32p, u8p, f64p, f32p, ip = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.ldpc_last_error.restype = C.c_char_p
    L.ldpc_init.argtypes = [C.c_int]
    L.ldpc_code_create_qc.restype = vp
    L.ldpc_code_create_qc.argtypes = [C.c_int, C.c_int, C.c_int, i32p]
    L.ldpc_code_create_csr.restype = vp
    L.ldpc_code_create_csr.argtypes = [C.c_int, C.c_int, i32p, i32p]
    L.ldpc_code_destroy.restype = None
    L.ldpc_code_destroy.argtypes = [vp]
    L.ldpc_code_dims.argtypes = [vp, ip, ip, ip]
    L.ldpc_code_csr.argtypes = [vp, i32p, i32p]
    L.ldpc_ctx_create.restype = vp
    L.ldpc_ctx_create.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.ldpc_ctx_create_ex.restype = vp
    L.ldpc_ctx_create_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ldpc_ctx_create_on.restype = vp
    L.ldpc_ctx_create_on.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ldpc_ctx_create_cfg.restype = vp
    L.ldpc_ctx_create_cfg.argtypes = [vp, C.POINTER(CtxConfig)]
    L.ldpc_ctx_schedule.argtypes = [vp]
    L.ldpc_code_set_layers.argtypes = [vp, C.c_int, i32p]
    L.ldpc_qc_layer_order.argtypes = [C.c_int, C.c_int, i32p, C.c_int, i32p]
    L.ldpc_code_layers.argtypes = [vp, ip, i32p]
    L.ldpc_sim_create_on.restype = vp
    L.ldpc_sim_create_on.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
    L.ldpc_ctx_destroy.restype = None
    L.ldpc_ctx_destroy.argtypes = [vp]
    L.ldpc_ctx_path.argtypes = [vp]
    L.ldpc_ctx_synchronize.argtypes = [vp]
    L.ldpc_decode_one.argtypes = [vp, C.c_int, f64p, u8p, ip, ip]
    L.ldpc_decode_batch.argtypes = [vp, C.c_int, C.c_int, f32p, u8p, i32p, u8p]
    L.ldpc_decode_batch_f64.argtypes = [vp, C.c_int, C.c_int, f64p, u8p, i32p, u8p, f64p]
    L.ldpc_decode_batch_dev.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.ldpc_decode_batch_f16.argtypes = [vp, C.c_int, C.c_int, vp, u8p, i32p, u8p]
    L.ldpc_decode_batch_dev_f16.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.ldpc_decode_batch_dev_packed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    L.ldpc_decode_batch_packed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, u8p, i32p, u8p]
    L.ldpc_debug_step.argtypes = [vp, C.c_int, f64p, f64p, f64p, f64p, f64p, u8p]
    L.ldpc_decode_trace.argtypes = [vp, C.c_int, C.c_int, f64p, u8p, i32p, u8p, f64p]
    L.ldpc_host_alloc.restype = vp
    L.ldpc_host_alloc.argtypes = [C.c_size_t]
    L.ldpc_host_free.restype = None
    L.ldpc_host_free.argtypes = [vp]
    L.ldpc_ctx_set_timing.argtypes = [vp, C.c_int]
    L.ldpc_ctx_kernel_time.argtypes = [vp, ip, f64p]
    L.ldpc_ctx_kernel_name.restype = C.c_char_p
    L.ldpc_ctx_kernel_name.argtypes = [vp]
    L.ldpc_ctx_kernel_geometry.argtypes = [vp, ip, ip]
    L.ldpc_jit_cache_dir.restype = C.c_char_p
    L.ldpc_jit_source.restype = C.c_long
    L.ldpc_jit_source.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.ldpc_jit_prepare.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_size_t, ip, f64p]
    L.ldpc_jit_source_for.restype = C.c_long
    L.ldpc_jit_source_for.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.ldpc_jit_prepare_for.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, ip, f64p]
    L.ldpc_sim_create.restype = vp
    L.ldpc_sim_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
    L.ldpc_sim_destroy.restype = None
    L.ldpc_sim_destroy.argtypes = [vp]
    L.ldpc_sim_generate.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, C.c_double, vp, vp, vp]
    L.ldpc_sim_generate_f16.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, C.c_double, vp, vp, vp]
    L.ldpc_sim_tally.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.ldpc_sim_encode_host.argtypes = [vp, u8p, u8p]
    L.ldpc_sim_create_qc_on.restype = vp
    L.ldpc_sim_create_qc_on.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_int]
    L.ldpc_sim_encoder.argtypes = [vp]
    L.ldpc_sim_encode_batch.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, vp, vp, vp]
    L.ldpc_matrix_qc_words.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.ldpc_matrix_rank.argtypes = [vp]
    L.ldpc_matrix_load.restype = vp
    L.ldpc_matrix_load.argtypes = [C.c_char_p, C.c_char_p]
    L.ldpc_matrix_load_mackay.restype = vp
    L.ldpc_matrix_load_mackay.argtypes = [C.c_char_p]
    L.ldpc_matrix_destroy.restype = None
    L.ldpc_matrix_destroy.argtypes = [vp]
    L.ldpc_matrix_info.argtypes = [vp, ip, ip, ip, ip, ip]
    L.ldpc_matrix_dense.argtypes = [vp, u8p]
    L.ldpc_matrix_qc_offsets.argtypes = [vp, i32p]
    L.ldpc_code_from_matrix.restype = vp
    L.ldpc_code_from_matrix.argtypes = [vp]
    L.ldpc_ecc_create.restype = vp
    L.ldpc_ecc_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.ldpc_ecc_create_replicas.restype = vp
    L.ldpc_ecc_create_replicas.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, ip]
    L.ldpc_ecc_replicas.argtypes = [vp]
    L.ldpc_ecc_ctx_at.restype = vp
    L.ldpc_ecc_ctx_at.argtypes = [vp, C.c_int]
    L.ldpc_ecc_sim_at.restype = vp
    L.ldpc_ecc_sim_at.argtypes = [vp, C.c_int]
    L.ldpc_ecc_decode_on.argtypes = [vp, C.c_int, f64p, u8p, ip]
    L.ldpc_ecc_set_coalescing.argtypes = [vp, C.c_int, C.c_int]
    L.ldpc_ecc_coalescing_stats.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.ldpc_ctx_code.restype = vp
    L.ldpc_ctx_code.argtypes = [vp]
    L.ldpc_ctx_max_batch.argtypes = [vp]
    L.ldpc_ctx_device.argtypes = [vp]
    L.ldpc_batcher_create.restype = vp
    L.ldpc_batcher_create.argtypes = [vp, C.c_int, C.c_int]
    L.ldpc_batcher_destroy.restype = None
    L.ldpc_batcher_destroy.argtypes = [vp]
    L.ldpc_batcher_decode_one.argtypes = [vp, C.c_int, f64p, u8p, ip, ip]
    L.ldpc_batcher_stats.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.ldpc_ecc_destroy.restype = None
    L.ldpc_ecc_destroy.argtypes = [vp]
    L.ldpc_ecc_name.restype = C.c_char_p
    L.ldpc_ecc_name.argtypes = [vp]
    for f in ("message_length", "codeword_length", "unpunctured_length", "max_iters"):
        getattr(L, "ldpc_ecc_" + f).argtypes = [vp]
    L.ldpc_ecc_encode.argtypes = [vp, u8p, u8p]
    L.ldpc_ecc_decode.argtypes = [vp, f64p, u8p, ip]
    L.ldpc_ecc_ctx.restype = vp
    L.ldpc_ecc_ctx.argtypes = [vp]
    L.ldpc_ecc_sim.restype = vp
    L.ldpc_ecc_sim.argtypes = [vp]
    L.ldpc_ecc_code.restype = vp
    L.ldpc_ecc_code.argtypes = [vp]
    _lib = L
    return L


def last_error() -> str:
    return lib().ldpc_last_error().decode(errors="replace")


def check(rc):
    if rc != OK:
        raise LdpcError(rc, last_error())


def ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class PinnedArray:
    """numpy view of page-locked host memory from ldpc_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib().ldpc_host_alloc(self.nbytes)
        if not self._p:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        buf = (C.c_char * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)
        _register(self)

    def close(self):
        if self._p:
            self.array = None
            lib().ldpc_host_free(self._p)
            self._p = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def init(device: int = 0):
    check(lib().ldpc_init(int(device)))


_VARIANTS = {"tanh": TANH, "min": MINSUM, "minsum": MINSUM, "min-sum": MINSUM, "cm": TANH_CM, "tanh-cm": TANH_CM, TANH: TANH, MINSUM: MINSUM,
             TANH_CM: TANH_CM, "cuda32": TANH_CUDA32, "cuda-arraylet2": TANH_CUDA32, TANH_CUDA32: TANH_CUDA32}
_DTYPES = {"f32": F32, "f64": F64, "f16": F16, "f16pk": F16PK, F32: F32, F64: F64, F16: F16, F16PK: F16PK}
_PATHS = {"auto": PATH_AUTO, "flood": PATH_FLOOD, "fused": PATH_FUSED, 0: 0, 1: 1, 2: 2}
_SUM_ORDERS = {"reference": SUM_REFERENCE, "arraylet": SUM_ARRAYLET, "sparse": SUM_SPARSE, 0: 0, 1: 1, 2: 2}
_SCHEDULES = {"flooding": SCHED_FLOODING, "flood": SCHED_FLOODING, "layered": SCHED_LAYERED, 0: 0, 1: 1}


class Code:
    """Parity-check graph handle (ldpc_code)."""

    def __init__(self, handle, owned=True):
        if not handle:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        self._h = handle
        self._owned = owned
        _register(self)
        M, N, E = C.c_int(), C.c_int(), C.c_int()
        check(lib().ldpc_code_dims(self._h, C.byref(M), C.byref(N), C.byref(E)))
        self.M, self.N, self.E = M.value, N.value, E.value

    @classmethod
    def from_qc(cls, sz, offsets):
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        assert off.ndim == 2
        return cls(lib().ldpc_code_create_qc(int(sz), off.shape[0], off.shape[1], ptr(off, C.c_int32)))

    @staticmethod
    def qc_layer_order(offsets, run=4):
        """-> (perm, full_runs): an order of the block rows in which runs of up to `run` consecutive rows share no block column
        (ldpc_qc_layer_order; offsets[perm] is the matrix to hand to from_qc for the layered schedule)"""
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        perm = np.zeros(off.shape[0], np.int32)
        n = lib().ldpc_qc_layer_order(off.shape[0], off.shape[1], ptr(off, C.c_int32), int(run), ptr(perm, C.c_int32))
        if n < 0:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        return perm, n

    @classmethod
    def from_csr(cls, row_ptr, col_idx, N):
        rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
        ci = np.ascontiguousarray(col_idx, dtype=np.int32)
        return cls(lib().ldpc_code_create_csr(len(rp) - 1, int(N), ptr(rp, C.c_int32), ptr(ci, C.c_int32)))

    @classmethod
    def from_dense(cls, H):
        H = np.asarray(H)
        M, N = H.shape
        rp = np.zeros(M + 1, np.int32)
        rp[1:] = np.cumsum(H.astype(bool).sum(1))
        ci = np.nonzero(H)[1].astype(np.int32)
        return cls.from_csr(rp, ci, N)

    def csr(self):
        rp = np.zeros(self.M + 1, np.int32)
        ci = np.zeros(self.E, np.int32)
        check(lib().ldpc_code_csr(self._h, ptr(rp, C.c_int32), ptr(ci, C.c_int32)))
        return rp, ci

    def set_layers(self, layer_ptr):
        lp = np.ascontiguousarray(layer_ptr, dtype=np.int32)
        check(lib().ldpc_code_set_layers(self._h, len(lp) - 1, ptr(lp, C.c_int32)))

    def layers(self):
        n = C.c_int()
        check(lib().ldpc_code_layers(self._h, C.byref(n), None))
        lp = np.zeros(n.value + 1, np.int32)
        check(lib().ldpc_code_layers(self._h, C.byref(n), ptr(lp, C.c_int32)))
        return lp

    def jit_source(self, variant="min", dtype="f32", schedule="flooding") -> str:
        """the translation unit the run-time compiler would be given for this code (LdpcError -5 if there is none)"""
        sch = _SCHEDULES[schedule]
        n = lib().ldpc_jit_source_for(self._h, _VARIANTS[variant], _DTYPES[dtype], sch, None, 0)
        if n < 0:
            raise LdpcError(int(n), last_error())
        buf = C.create_string_buffer(n + 1)
        lib().ldpc_jit_source_for(self._h, _VARIANTS[variant], _DTYPES[dtype], sch, buf, n + 1)
        return buf.value.decode()

    def jit_prepare(self, variant="min", dtype="f32", schedule="flooding"):
        """compile this code's specialised kernel into the disk cache (no GPU needed) -> (kernel name, from_cache, seconds)"""
        name = C.create_string_buffer(128)
        fc, sec = C.c_int(), C.c_double()
        check(lib().ldpc_jit_prepare_for(self._h, _VARIANTS[variant], _DTYPES[dtype], _SCHEDULES[schedule], name, 128, C.byref(fc), C.byref(sec)))
        return name.value.decode(), bool(fc.value), sec.value

    @classmethod
    def from_matrix(cls, matrix: "Matrix"):
        return cls(lib().ldpc_code_from_matrix(matrix._h))

    def close(self):
        if self._h and self._owned:
            lib().ldpc_code_destroy(self._h)
        self._h = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """One decoder replica (ldpc_ctx): the object behind the reference's per-frame closure."""

    def __init__(self, code: Code, variant="min", dtype="f32", max_batch=64, path="auto", _handle=None, device=None, schedule="flooding",
                 sum_order="reference"):
        """sum_order: "reference" | "arraylet" | "sparse" -- the column-sum order of that family of the reference's decoders
        (ldpc_sum_order: parity modes, flood path)"""
        self.code = code
        self.max_batch = int(max_batch)
        self._owned = _handle is None
        if _handle is not None:
            self._h = _handle
        elif device is None and _SCHEDULES[schedule] == SCHED_FLOODING and _SUM_ORDERS[sum_order] == SUM_REFERENCE:
            self._h = lib().ldpc_ctx_create_ex(code._h, _VARIANTS[variant], _DTYPES[dtype], int(max_batch), _PATHS[path])
        else:   # explicit device (replicas of one code on several GPUs of this process) and/or the layered schedule
            cfg = CtxConfig(C.sizeof(CtxConfig), -1 if device is None else int(device), _VARIANTS[variant], _DTYPES[dtype], int(max_batch),
                            _PATHS[path], _SCHEDULES[schedule], _SUM_ORDERS[sum_order])
            self._h = lib().ldpc_ctx_create_cfg(code._h, C.byref(cfg))
        if not self._h:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        _register(self)
        self.path = {PATH_FLOOD: "flood", PATH_FUSED: "fused"}[lib().ldpc_ctx_path(self._h)]
        self.schedule = {SCHED_FLOODING: "flooding", SCHED_LAYERED: "layered"}[lib().ldpc_ctx_schedule(self._h)]

    def decode_one(self, llr, max_iters):
        llr = np.ascontiguousarray(llr, np.float64)
        assert llr.shape == (self.code.N,)
        bits = np.zeros(self.code.N, np.uint8)
        it, cv = C.c_int(), C.c_int()
        check(lib().ldpc_decode_one(self._h, int(max_iters), ptr(llr, C.c_double), ptr(bits, C.c_uint8), C.byref(it), C.byref(cv)))
        return bits, it.value, bool(cv.value)

    def decode_batch(self, llr, max_iters, want_lam=False, out_bits=None):
        """llr [F][N] float16, float32 or float64 (host).  -> bits [F][N], iters [F], converged [F] (, lam)"""
        llr = np.asarray(llr)
        F = llr.shape[0]
        assert llr.shape == (F, self.code.N)
        bits = out_bits if out_bits is not None else np.zeros((F, self.code.N), np.uint8)
        assert bits.shape == (F, self.code.N) and bits.dtype == np.uint8 and bits.flags.c_contiguous
        iters = np.zeros(F, np.int32)
        conv = np.zeros(F, np.uint8)
        if llr.dtype == np.float16 and not want_lam:
            llr = np.ascontiguousarray(llr)
            check(lib().ldpc_decode_batch_f16(self._h, int(max_iters), F, llr.ctypes.data_as(C.c_void_p), ptr(bits, C.c_uint8), ptr(iters, C.c_int32), ptr(conv, C.c_uint8)))
            return bits, iters, conv
        if llr.dtype == np.float32 and not want_lam:
            llr = np.ascontiguousarray(llr)
            check(lib().ldpc_decode_batch(self._h, int(max_iters), F, ptr(llr, C.c_float), ptr(bits, C.c_uint8), ptr(iters, C.c_int32), ptr(conv, C.c_uint8)))
            return bits, iters, conv
        llr = np.ascontiguousarray(llr, np.float64)
        lam = np.zeros((F, self.code.N), np.float64) if want_lam else None
        check(lib().ldpc_decode_batch_f64(self._h, int(max_iters), F, ptr(llr, C.c_double), ptr(bits, C.c_uint8), ptr(iters, C.c_int32), ptr(conv, C.c_uint8), ptr(lam, C.c_double)))
        return (bits, iters, conv, lam) if want_lam else (bits, iters, conv)

    def decode_batch_packed(self, llr, max_iters):
        """llr [F][N] float16 or float32 (host) -> packed bits [F][ceil(N/8)] (bit i of a frame: byte i // 8, bit i % 8), iters, converged"""
        llr = np.ascontiguousarray(llr)
        assert llr.dtype in (np.float16, np.float32) and llr.shape[1] == self.code.N
        F = llr.shape[0]
        packed = np.zeros((F, (self.code.N + 7) // 8), np.uint8)
        iters = np.zeros(F, np.int32)
        conv = np.zeros(F, np.uint8)
        check(lib().ldpc_decode_batch_packed(self._h, int(max_iters), F, llr.ctypes.data_as(C.c_void_p), 1 if llr.dtype == np.float16 else 0,
                                             ptr(packed, C.c_uint8), ptr(iters, C.c_int32), ptr(conv, C.c_uint8)))
        return packed, iters, conv

    def decode_batch_dev_packed(self, d_llr_ptr, d_packed_ptr, batch, max_iters, d_iters_ptr=None, d_conv_ptr=None, stream=None, llr_f16=False):
        """device pointers; d_packed [batch][ceil(N/8)] bytes"""
        check(lib().ldpc_decode_batch_dev_packed(self._h, int(max_iters), int(batch), d_llr_ptr, 1 if llr_f16 else 0, d_packed_ptr, d_iters_ptr, d_conv_ptr, stream))

    def decode_batch_dev(self, d_llr_ptr, d_bits_ptr, batch, max_iters, d_iters_ptr=None, d_conv_ptr=None, stream=None, llr_f16=False):
        """device pointers; d_llr [batch][N] float32, or float16 with llr_f16=True"""
        fn = lib().ldpc_decode_batch_dev_f16 if llr_f16 else lib().ldpc_decode_batch_dev
        check(fn(self._h, int(max_iters), int(batch), d_llr_ptr, d_bits_ptr, d_iters_ptr, d_conv_ptr, stream))

    def synchronize(self):
        check(lib().ldpc_ctx_synchronize(self._h))

    def decode_trace(self, llr, max_iters):
        llr = np.ascontiguousarray(llr, np.float64)
        F = llr.shape[0]
        bits = np.zeros((F, self.code.N), np.uint8)
        iters = np.zeros(F, np.int32)
        conv = np.zeros(F, np.uint8)
        trace = np.zeros((F, max_iters + 1, self.code.N), np.float64)
        check(lib().ldpc_decode_trace(self._h, int(max_iters), F, ptr(llr, C.c_double), ptr(bits, C.c_uint8), ptr(iters, C.c_int32), ptr(conv, C.c_uint8), ptr(trace, C.c_double)))
        return bits, iters, conv, trace

    def debug_step(self, orig, lam, ne):
        orig = np.ascontiguousarray(orig, np.float64)
        lam = np.ascontiguousarray(lam, np.float64)
        ne = np.ascontiguousarray(ne, np.float64)
        F = orig.shape[0]
        assert orig.shape == lam.shape == (F, self.code.N) and ne.shape == (F, self.code.E)
        ne2 = np.zeros_like(ne)
        lam2 = np.zeros_like(lam)
        syn = np.zeros(F, np.uint8)
        check(lib().ldpc_debug_step(self._h, F, ptr(orig, C.c_double), ptr(lam, C.c_double), ptr(ne, C.c_double), ptr(ne2, C.c_double), ptr(lam2, C.c_double), ptr(syn, C.c_uint8)))
        return ne2, lam2, syn.astype(bool)

    def set_timing(self, enabled=True):
        check(lib().ldpc_ctx_set_timing(self._h, int(bool(enabled))))

    def kernel_time(self):
        """-> (launches, total_ms) of the dominant kernel since the last call (HIP events)."""
        n, ms = C.c_int(), C.c_double()
        check(lib().ldpc_ctx_kernel_time(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    @property
    def kernel_name(self):
        return lib().ldpc_ctx_kernel_name(self._h).decode()

    @property
    def kernel_geometry(self):
        """-> (threads per workgroup, frames per workgroup) of the dominant kernel's last launch (0, 0: flood path)"""
        t, f = C.c_int(), C.c_int()
        check(lib().ldpc_ctx_kernel_geometry(self._h, C.byref(t), C.byref(f)))
        return t.value, f.value

    def close(self):
        if self._h and self._owned:
            lib().ldpc_ctx_destroy(self._h)
        self._h = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Batcher:
    """ldpc_batcher on one decoder replica: concurrent per-frame decode_one() calls (one per host thread, the reference
    harness's calling convention, Utils.hs:63-69) are collected into one launch.  This is the object the Haskell binding
    shares between the replicas of one GPU (haskell/ECC/Code/LDPC/GPU/HIP.hs closureFor)."""

    def __init__(self, decoder: Decoder, max_frames=64, max_wait_us=200):
        self.decoder = decoder
        self._h = lib().ldpc_batcher_create(decoder._h, int(max_frames), int(max_wait_us))
        if not self._h:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        _register(self)

    def decode_one(self, llr, max_iters):
        llr = np.ascontiguousarray(llr, np.float64)
        N = self.decoder.code.N
        assert llr.shape == (N,)
        bits = np.zeros(N, np.uint8)
        it, cv = C.c_int(), C.c_int()
        check(lib().ldpc_batcher_decode_one(self._h, int(max_iters), ptr(llr, C.c_double), ptr(bits, C.c_uint8), C.byref(it), C.byref(cv)))
        return bits, it.value, bool(cv.value)

    def stats(self):
        """-> (decode calls, launches) so far"""
        a, b = C.c_long(), C.c_long()
        check(lib().ldpc_batcher_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().ldpc_batcher_destroy(self._h)
        self._h = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Matrix:
    """A loaded matrix file (ldpc_matrix): the reference's LoaderMatrix (Data/BitMatrix/Loader.hs:49-52)."""

    def __init__(self, handle):
        if not handle:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        self._h = handle
        _register(self)
        v = [C.c_int() for _ in range(5)]
        check(lib().ldpc_matrix_info(self._h, *[C.byref(x) for x in v]))
        self.rows, self.cols, self.sz, self.block_rows, self.block_cols = [x.value for x in v]

    @classmethod
    def load(cls, codes_dir, name):
        return cls(lib().ldpc_matrix_load(str(codes_dir).encode(), name.encode()))

    @classmethod
    def load_mackay(cls, path):
        return cls(lib().ldpc_matrix_load_mackay(str(path).encode()))

    def dense(self):
        out = np.zeros((self.rows, self.cols), np.uint8)
        check(lib().ldpc_matrix_dense(self._h, ptr(out, C.c_uint8)))
        return out

    def rank(self):
        """rank over GF(2) of the expanded matrix"""
        r = lib().ldpc_matrix_rank(self._h)
        if r < 0:
            raise LdpcError(r, last_error())
        return r

    def qc_words(self):
        """first-row patterns of a QC source as little-endian 32-bit words [block_rows][block_cols][ceil(sz/32)]"""
        out = np.zeros((self.block_rows, self.block_cols, (self.sz + 31) // 32), np.uint32)
        check(lib().ldpc_matrix_qc_words(self._h, ptr(out, C.c_uint32)))
        return out

    def qc_offsets(self):
        out = np.zeros((self.block_rows, self.block_cols), np.int32)
        check(lib().ldpc_matrix_qc_offsets(self._h, ptr(out, C.c_int32)))
        return out

    def close(self):
        if self._h:
            lib().ldpc_matrix_destroy(self._h)
            self._h = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Sim:
    """Device-side frame source + error tally (ldpc_sim)."""

    def __init__(self, code: Code, k, n_tx, G=None, max_batch=64, _handle=None, device=None, G_qc=None):
        """G: dense generator [k][p] bytes; G_qc = (sz, words [block_rows][block_cols][sz/32] uint32): the quasi-cyclic form
        (Matrix.qc_words), encoded by rotate-and-xor like Fast/Encoder.hs; neither: all-zero codewords"""
        self.code, self.k, self.n_tx = code, int(k), int(n_tx)
        self._owned = _handle is None
        if _handle is None:
            dev = int(device) if device is not None else lib().ldpc_current_device()
            if dev < 0:
                raise LdpcError(ENODEVICE, "ldpc_init() has not succeeded")
            if G_qc is not None:
                sz, words = G_qc
                words = np.ascontiguousarray(words, np.uint32)
                _handle = lib().ldpc_sim_create_qc_on(code._h, dev, int(k), int(n_tx), int(sz), words.shape[0], words.shape[1], ptr(words, C.c_uint32), int(max_batch))
            elif G is not None:
                G = np.ascontiguousarray(G, np.uint8)
                assert G.shape[0] == k
                _handle = lib().ldpc_sim_create_on(code._h, dev, int(k), int(n_tx), G.shape[1], ptr(G, C.c_uint8), int(max_batch))
            else:
                _handle = lib().ldpc_sim_create_on(code._h, dev, int(k), int(n_tx), 0, None, int(max_batch))
        if not _handle:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        self._h = _handle
        _register(self)

    def generate(self, seed, first_frame, batch, ebn0_db, d_llr_ptr, d_msg_ptr=None, stream=None, llr_f16=False):
        fn = lib().ldpc_sim_generate_f16 if llr_f16 else lib().ldpc_sim_generate
        check(fn(self._h, int(seed), int(first_frame), int(batch), float(ebn0_db), d_llr_ptr, d_msg_ptr, stream))

    @property
    def encoder(self):
        return {0: "none", 1: "dense", 2: "qc"}[lib().ldpc_sim_encoder(self._h)]

    def encode_batch(self, seed, first_frame, batch, d_codewords_ptr, d_msg_ptr=None, stream=None):
        """the encoder alone: codewords [batch][n_tx] bytes on the device"""
        check(lib().ldpc_sim_encode_batch(self._h, int(seed), int(first_frame), int(batch), d_codewords_ptr, d_msg_ptr, stream))

    def tally(self, batch, d_bits_ptr, d_iters_ptr, d_tally_ptr, stream=None):
        check(lib().ldpc_sim_tally(self._h, int(batch), d_bits_ptr, d_iters_ptr, d_tally_ptr, stream))

    def encode_host(self, msg, p):
        msg = np.ascontiguousarray(msg, np.uint8)
        par = np.zeros(p, np.uint8)
        check(lib().ldpc_sim_encode_host(self._h, ptr(msg, C.c_uint8), ptr(par, C.c_uint8)))
        return par

    def close(self):
        if self._h and self._owned:
            lib().ldpc_sim_destroy(self._h)
        self._h = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class ECC:
    """The plug-in record mkLDPC builds (src/ECC/Code/LDPC/Utils.hs:59-75), by code name:
    ECC(codes_dir, "ldpc/hip-minsum/jpl.4096.4.5/50/4/5")."""

    def __init__(self, codes_dir, code_name, max_batch=64, replicas=1, devices=None):
        """replicas / devices: the reference's maxThreadCount decoder replicas in one process (Utils.hs:53), replica i on
        HIP device devices[i] (None: the calling thread's device)"""
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            replicas = len(devices)
        else:
            devs = None
        self._h = lib().ldpc_ecc_create_replicas(str(codes_dir).encode(), code_name.encode(), int(max_batch), int(replicas), devs)
        if not self._h:
            raise LdpcError(lib().ldpc_last_error_code(), last_error())
        _register(self)
        L = lib()
        self.name = L.ldpc_ecc_name(self._h).decode()
        self.message_length = L.ldpc_ecc_message_length(self._h)
        self.codeword_length = L.ldpc_ecc_codeword_length(self._h)
        self.unpunctured_length = L.ldpc_ecc_unpunctured_length(self._h)
        self.max_iters = L.ldpc_ecc_max_iters(self._h)
        self.code = Code(L.ldpc_ecc_code(self._h), owned=False)
        self.decoder = Decoder(self.code, max_batch=max_batch, _handle=L.ldpc_ecc_ctx(self._h))
        self.sim = Sim(self.code, self.message_length, self.codeword_length, _handle=L.ldpc_ecc_sim(self._h))

    def encode(self, msg):
        msg = np.ascontiguousarray(msg, np.uint8)
        assert msg.shape == (self.message_length,)
        cw = np.zeros(self.codeword_length, np.uint8)
        check(lib().ldpc_ecc_encode(self._h, ptr(msg, C.c_uint8), ptr(cw, C.c_uint8)))
        return cw

    def decode(self, llr, replica=None):
        """the record's decode (Utils.hs:62-72); replica None = picked by the calling thread, as the reference does"""
        llr = np.ascontiguousarray(llr, np.float64)
        assert llr.shape == (self.codeword_length,)
        out = np.zeros(self.message_length, np.uint8)
        ok = C.c_int()
        if replica is None:
            check(lib().ldpc_ecc_decode(self._h, ptr(llr, C.c_double), ptr(out, C.c_uint8), C.byref(ok)))
        else:
            check(lib().ldpc_ecc_decode_on(self._h, int(replica), ptr(llr, C.c_double), ptr(out, C.c_uint8), C.byref(ok)))
        return out, bool(ok.value)

    @property
    def replicas(self):
        return lib().ldpc_ecc_replicas(self._h)

    def replica(self, i):
        """-> (Decoder, Sim) views of replica i (owned by the record)"""
        L = lib()
        return (Decoder(self.code, max_batch=self.decoder.max_batch, _handle=L.ldpc_ecc_ctx_at(self._h, int(i))),
                Sim(self.code, self.message_length, self.codeword_length, _handle=L.ldpc_ecc_sim_at(self._h, int(i))))

    def coalescing_stats(self):
        """-> (decode calls, launches) through the batchers so far"""
        c, l = C.c_long(), C.c_long()
        check(lib().ldpc_ecc_coalescing_stats(self._h, C.byref(c), C.byref(l)))
        return c.value, l.value

    def set_coalescing(self, max_frames, max_wait_us=200):
        """concurrent decode() calls share launches (ldpc_batcher): up to max_frames per launch, a lone caller waits at
        most max_wait_us for company; 0 switches it off"""
        check(lib().ldpc_ecc_set_coalescing(self._h, int(max_frames), int(max_wait_us)))

    def close(self):
        if self._h:
            self.decoder._h = None
            self.sim._h = None
            self.code._h = None
            lib().ldpc_ecc_destroy(self._h)
            self._h = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass
