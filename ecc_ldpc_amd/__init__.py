"""ecc_ldpc_amd -- MI355X-native LDPC belief-propagation decode path behind the ku-fpg/ecc-ldpc
decoder record.  The compute lives in libldpc_hip.so (hand-written HIP for gfx950, C ABI in
include/ldpc_hip.h); this package is the thin host-side mirror used by tests and bench.py."""
from ._lib import (Code, Decoder, Batcher, Matrix, Sim, ECC, PinnedArray, LdpcError, init, lib, last_error, close_all, SO_PATH, ABI_SYMBOLS,  # noqa: F401
                   TANH, MINSUM, F32, F64, F16, PATH_AUTO, PATH_FLOOD, PATH_FUSED, SCHED_FLOODING, SCHED_LAYERED)
