"""oracle/emulate_f16.py -- TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product).

NumPy emulation of the LDPC_F16 decoders ("fp16 storage in HBM, fp32 arithmetic", include/ldpc_hip.h).
The reference has no fp16 path (its CUDA plug-ins compute in float32, cudabits/arraylet2.cu:43-83), so this
is the build's own extension with its own oracle -- SURVEY.md section 8f item 4 -- PARITY UNPINNED against
the reference by construction.  What it pins is that the HIP kernels do exactly what the header says:

  flood path : the loop of Reference/Min.hs:54-104 (syndrome -> return lam | out of turns -> return orig |
               check-node rule (-3/4) * foldr1 min', column sum foldr (+) orig in descending row order),
               with EVERY stored quantity -- channel LLRs, messages, lam -- saturated to +-65504 and rounded
               to fp16 (nearest even) when written, and all arithmetic on the float32 values read back.
               float32 add/sub/mul/min/compare are exactly rounded IEEE operations on both sides (the
               library is built with -ffp-contract=off), so the min-sum trajectory is reproduced BIT FOR BIT.
  fused paths: the f32 decoder fed r16(llr); only the channel LLRs ever live in HBM.  (Tested as
               fused-F16(llr) == fused-F32(r16(llr)); no separate emulation needed.)
  LDPC_F16PK : ARITHMETIC in binary16 (csrc/fused_pk16_body.h, two frames per lane in packed instructions): the same loop
               with state L = -lam and u = ne' / (3/4) in fp16 and the 3/4 applied inside fused multiply-adds --
               decode_minsum_pk16 below, bit for bit (numpy's float16 is IEEE binary16; an fp16 fma is computed exactly in
               float64 -- 0.75 * u + L needs at most 43 significant bits -- and rounded once).
"""
import numpy as np

F16_MAX = np.float32(65504.0)


def r16(x):
    """value of float32 x after being stored as fp16: saturate, round to nearest even, read back"""
    x = np.asarray(x, np.float32)
    return np.clip(x, -F16_MAX, F16_MAX).astype(np.float16).astype(np.float32)


def _rows(g):
    return [(m, g.col_idx[g.row_ptr[m]:g.row_ptr[m + 1]], int(g.row_ptr[m])) for m in range(g.M)]


def cn_minsum_f32(t):
    """t [F, D] float32 = lam_j - ne[m,j]  ->  ne'[m,:]  (ldpc_math.h cn_minsum; Min.hs:78-86)"""
    a = np.abs(t)
    pos = t > 0                                   # x_j = -t_j is negative <=> t_j > 0
    par = np.logical_xor.reduce(pos, axis=1, keepdims=True)
    i1 = np.argmin(a, axis=1)                     # first minimum, as the kernel's strict '<' scan keeps it
    m1 = np.take_along_axis(a, i1[:, None], axis=1)
    a2 = a.copy()
    np.put_along_axis(a2, i1[:, None], np.float32(np.inf), axis=1)
    m2 = a2.min(axis=1, keepdims=True)
    k = np.arange(t.shape[1])[None, :]
    mag = np.where(k == i1[:, None], m2, m1)
    neg = np.logical_xor(par, pos)
    acc = np.where(neg, -mag, mag).astype(np.float32)
    return (np.float32(-0.75) * acc).astype(np.float32)


def step_minsum_f16_flood(g, orig, lam, ne):
    """one update of the flood F16 decoder; all arrays float32 holding fp16-representable values.
    orig, lam [F, N]; ne [F, E] in CSR edge order  ->  ne', lam', syndrome_zero [F]"""
    F = lam.shape[0]
    hard = lam > 0
    ne2 = np.empty_like(ne)
    syn_ok = np.ones(F, bool)
    for m, cols, e0 in _rows(g):
        d = len(cols)
        syn_ok &= ~np.logical_xor.reduce(hard[:, cols], axis=1)
        t = (lam[:, cols] - ne[:, e0:e0 + d]).astype(np.float32)
        ne2[:, e0:e0 + d] = r16(cn_minsum_f32(t))
    acc = orig.copy()
    for m, cols, e0 in reversed(_rows(g)):        # foldr (+) orig: rows in descending order (Min.hs:101)
        acc[:, cols] = (ne2[:, e0:e0 + len(cols)] + acc[:, cols]).astype(np.float32)
    return ne2, r16(acc), syn_ok


def decode_minsum_f16_flood(g, llr, max_iters):
    """llr [F, N] float32 -> bits [F, N] u8, iters [F], converged [F], trace (list over turns of lam [F, N],
    holding for a finished frame its returned LLR vector, as ldpc_decode_trace does)"""
    llr = np.asarray(llr, np.float32)
    F = llr.shape[0]
    orig = r16(llr)
    lam = orig.copy()
    ne = np.zeros((F, g.E), np.float32)
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, bool)
    live = np.ones(F, bool)
    out = orig.copy()
    trace = []
    for n in range(max_iters + 1):
        trace.append(np.where(live[:, None], lam, np.float32(0)))
        ne2, lam2, ok = step_minsum_f16_flood(g, orig, lam, ne)
        fin = live & ok                           # Min.hs:75: syndrome zero -> lam
        out[fin] = lam[fin]; conv[fin] = True; iters[fin] = n
        live &= ~ok
        if n >= max_iters:                        # Min.hs:76: out of turns -> channel LLRs
            iters[live] = n
            break
        if not live.any():
            break
        lam = np.where(live[:, None], lam2, lam)
        ne = np.where(live[:, None], ne2, ne)
    return (out > 0).astype(np.uint8), iters, conv, trace


# ---------------------------------------------------------------------------------------------------------------------
# LDPC_F16PK: packed-fp16 min-sum (csrc/fused_pk16_body.h holds the specification this restates)
LLR16_MAX = np.float32(16384.0)    # channel LLRs saturate here ...
U16_MAX = np.float16(2048.0)       # ... and message magnitudes (before their 3/4) here: with column degree <= 30 nothing overflows


def neg_llr16(llr):
    """channel LLRs (float32) -> L0 = -(LLR saturated at +-16384 and rounded to fp16), a zero as +0"""
    v = np.clip(np.asarray(llr, np.float32), -LLR16_MAX, LLR16_MAX)
    h = (np.float32(0) - v).astype(np.float16)
    h[h == 0] = np.float16(0)
    return h


def fma16(a, k, c):
    """round16(a * k + c) with ONE rounding (v_pk_fma_f16); a, c float16 arrays, k a Python float exactly representable"""
    return (a.astype(np.float64) * k + c.astype(np.float64)).astype(np.float16)


def step_minsum_pk16(g, L0, L, u):
    """one update.  L0, L [F, N] float16 (negated LLRs), u [F, E] float16 in CSR edge order -> u', L', syndrome_zero [F]"""
    F = L.shape[0]
    hard = np.signbit(L)                               # hard(lam) = lam > 0 = sign of -lam (L is never -0)
    u2 = np.empty_like(u)
    syn_ok = np.ones(F, bool)
    rows = _rows(g)
    for m, cols, e0 in rows:
        d = len(cols)
        syn_ok &= ~np.logical_xor.reduce(hard[:, cols], axis=1)
        tn = fma16(u[:, e0:e0 + d], 0.75, L[:, cols])  # -(lam - ne)
        a = np.abs(tn)
        sg = np.signbit(tn)
        X = np.logical_xor.reduce(sg, axis=1, keepdims=True)
        i1 = np.argmin(a, axis=1)
        m1 = np.take_along_axis(a, i1[:, None], axis=1)
        a2 = a.copy()
        np.put_along_axis(a2, i1[:, None], np.float16(np.inf), axis=1)
        m2 = a2.min(axis=1, keepdims=True)
        k = np.arange(d)[None, :]
        mag = np.minimum(np.where(k == i1[:, None], m2, m1), U16_MAX)   # leave-one-out minimum (ties: m2 == m1), saturated
        neg = ~np.logical_xor(X, sg)                   # sign(u'_k) = 1 ^ X ^ sign(tN_k) = -prod_{j /= k} sgn(tN_j)
        u2[:, e0:e0 + d] = np.where(neg, -mag, mag).astype(np.float16)
    acc = L0.copy()
    for m, cols, e0 in reversed(rows):                 # Min.hs:101 foldr: rows in descending order
        acc[:, cols] = fma16(u2[:, e0:e0 + len(cols)], -0.75, acc[:, cols])
    return u2, acc, syn_ok


def decode_minsum_pk16(g, llr, max_iters):
    """llr [F, N] float32 -> bits [F, N] u8, iters [F], converged [F], trace (list over turns of lam = -L [F, N] float32,
    zero for a frame that has finished -- ldpc_decode_trace leaves those rows untouched)"""
    llr = np.asarray(llr, np.float32)
    F = llr.shape[0]
    L0 = neg_llr16(llr)
    L = L0.copy()
    u = np.zeros((F, g.E), np.float16)
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, bool)
    live = np.ones(F, bool)
    out = np.signbit(L0)
    trace = []
    with np.errstate(over="ignore", invalid="ignore"):
        for n in range(max_iters + 1):
            trace.append(np.where(live[:, None], -L.astype(np.float32), np.float32(0)))
            u2, L2, ok = step_minsum_pk16(g, L0, L, u)
            fin = live & ok                            # Min.hs:75: syndrome zero -> hard(lam)
            out[fin] = np.signbit(L[fin]); conv[fin] = True; iters[fin] = n
            live &= ~ok
            if n >= max_iters:                         # Min.hs:76: out of turns -> the channel's hard decisions
                iters[live] = n
                break
            if not live.any():
                break
            L = np.where(live[:, None], L2, L)
            u = np.where(live[:, None], u2, u)
    return out.astype(np.uint8), iters, conv, trace


# ---------------------------------------------------------------------------------------------------------------------
# LDPC_F16PK + LDPC_SCHED_LAYERED (csrc/fused_layered_body.h, namespace laypk): the row-layered schedule of
# oracle_decode_layered (ldpc_oracle.c) in the packed-fp16 arithmetic of decode_minsum_pk16 above
def sweep_minsum_pk16_layered(g, L, u):
    """one sweep over all rows in ascending order (= layer by layer: the rows of a layer share no column), in place on copies.
    L [F, N] float16 (negated LLRs), u [F, E] float16 -> L', u', moved [F] (some check was odd or some hard decision flipped)"""
    L, u = L.copy(), u.copy()
    F = L.shape[0]
    moved = np.zeros(F, bool)
    for m, cols, e0 in _rows(g):
        d = len(cols)
        l = L[:, cols]
        moved |= np.logical_xor.reduce(np.signbit(l), axis=1)           # odd: parity of the hard decisions this row saw
        tn = fma16(u[:, e0:e0 + d], 0.75, l)                            # -(lam - msg); first sweep: u = 0, tn = l
        a = np.abs(tn)
        sg = np.signbit(tn)
        X = np.logical_xor.reduce(sg, axis=1, keepdims=True)
        i1 = np.argmin(a, axis=1)
        m1 = np.take_along_axis(a, i1[:, None], axis=1)
        a2 = a.copy()
        np.put_along_axis(a2, i1[:, None], np.float16(np.inf), axis=1)
        m2 = a2.min(axis=1, keepdims=True)
        k = np.arange(d)[None, :]
        mag = np.minimum(np.where(k == i1[:, None], m2, m1), U16_MAX)
        un = np.where(~np.logical_xor(X, sg), -mag, mag).astype(np.float16)
        ln = fma16(un, -0.75, tn)                                       # -(t + msg'), one rounding
        moved |= (np.signbit(ln) != np.signbit(l)).any(axis=1)          # flip
        L[:, cols] = ln
        u[:, e0:e0 + d] = un
    return L, u, moved


def decode_minsum_pk16_layered(g, llr, max_iters):
    """llr [F, N] float32 -> bits, sweeps, converged, trace (lam = -L after sweep n at row n, row 0 = the channel LLRs in fp16;
    zero rows for a frame that has stopped)"""
    llr = np.asarray(llr, np.float32)
    F = llr.shape[0]
    L0 = neg_llr16(llr)
    L = L0.copy()
    u = np.zeros((F, g.E), np.float16)
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, bool)
    out = np.signbit(L0)
    # before the first sweep: the syndrome of the channel's hard decisions
    hard = np.signbit(L)
    ok = np.ones(F, bool)
    for m, cols, e0 in _rows(g):
        ok &= ~np.logical_xor.reduce(hard[:, cols], axis=1)
    conv |= ok
    live = ~ok
    trace = [-L.astype(np.float32)]
    with np.errstate(over="ignore", invalid="ignore"):
        for n in range(1, max_iters + 1):
            if not live.any():
                break
            L2, u2, moved = sweep_minsum_pk16_layered(g, L, u)
            L = np.where(live[:, None], L2, L)
            u = np.where(live[:, None], u2, u)
            trace.append(np.where(live[:, None], -L.astype(np.float32), np.float32(0)))
            fin = live & ~moved
            out[fin] = np.signbit(L[fin]); conv[fin] = True; iters[fin] = n
            live &= moved
    iters[live] = max_iters
    return out.astype(np.uint8), iters, conv, trace


# ---------------------------------------------------------------------------------------------------------------------
# LDPC_F16 + LDPC_SCHED_LAYERED from HBM (csrc/layered_qc.hip, layered_qc_kernel<float, 1, D, true, __half>): the row-layered
# schedule of oracle_decode_layered with lam STORED in fp16 (saturating round to nearest even at every write), f32 arithmetic,
# f32 messages (the kernel keeps them as row records {3/4 min1, 3/4 min2, signs | arg-min}: the same values)
def decode_minsum_f16_layered(g, llr, max_iters):
    """llr [F, N] float32 -> bits [F, N] u8, sweeps [F], converged [F], lam [F, N] float32 as each frame stopped (the fp16-rounded
    channel LLRs for a frame out of sweeps).  Rows in ascending order = layer by layer (the rows of a layer share no column)."""
    llr = np.asarray(llr, np.float32)
    F = llr.shape[0]
    orig = r16(llr)
    lam = orig.copy()
    msg = np.zeros((F, g.E), np.float32)
    rows = _rows(g)
    hard0 = lam > 0
    ok = np.ones(F, bool)
    for m, cols, e0 in rows:
        ok &= ~np.logical_xor.reduce(hard0[:, cols], axis=1)
    conv = ok.copy()
    live = ~ok
    iters = np.zeros(F, np.int32)
    out = lam.copy()
    for n in range(1, max_iters + 1):
        if not live.any():
            break
        moved = np.zeros(F, bool)
        L, M = lam.copy(), msg.copy()
        for m, cols, e0 in rows:
            d = len(cols)
            l = L[:, cols]
            moved |= np.logical_xor.reduce(l > 0, axis=1)                        # odd
            t = (l - M[:, e0:e0 + d]).astype(np.float32)
            nm = cn_minsum_f32(t)
            nw = r16((t + nm).astype(np.float32))                                # what the lam cell holds afterwards
            moved |= ((nw > 0) != (l > 0)).any(axis=1)                           # flip
            L[:, cols] = nw
            M[:, e0:e0 + d] = nm
        lam = np.where(live[:, None], L, lam)
        msg = np.where(live[:, None], M, msg)
        fin = live & ~moved
        out[fin] = lam[fin]; conv[fin] = True; iters[fin] = n
        live &= moved
    iters[live] = max_iters
    out[live] = orig[live]
    return (out > 0).astype(np.uint8), iters, conv, out
