#!/usr/bin/env python3
"""isa_histogram.py -- instruction histogram of the iteration loop of a gfx950 kernel, from the compiler's own
assembly (the `*-hip-amdgcn-amd-amdhsa-gfx950.s` files ecc_ldpc_amd/build.py keeps next to the objects:
`hipcc -save-temps=obj`).  Purpose: make the VALU-issue roofline of the on-chip decode kernels reproducible from
committed evidence -- how many VALU instructions of which issue-cost class one wave executes per BP iteration.

  python tools/isa_histogram.py ecc_ldpc_amd/build/fused_split-hip-amdgcn-amd-amdhsa-gfx950.s [-k REGEX] [-o out.json]

Method (static): the kernel's text is cut into basic blocks (labels, branch instructions), the control-flow graph is
built from the branch targets, back edges give the natural loops, and the ITERATION loop is the outermost loop with
the most instructions (a kernel with several such loops of similar size -- the split kernel runs one straight-line
program per wave group behind a wave-uniform branch -- reports each of them; a wave executes exactly one).  Inside
the loop two counts are given:
  every_turn : blocks that dominate the loop's latch -- executed on every turn of the loop
  whole_loop : every block of the loop (adds conditional work: convergence snapshot, trace stores, the
               syndrome-only last turn); an upper bound
A dynamic cross-check is the PMC pass of tools/profile.sh: SQ_INSTS_VALU / SQ_WAVES / (mean turns per frame).

Issue cost classes (clk per wave-instruction per SIMD, measured on MI355X with tools/microbench_valu.hip,
profiles/*_microbench*.txt): "2" full rate, "4" half rate, "8" transcendental (quarter rate, 8.2), "pk" packed f32 (4.7), "pk16" packed 16-bit (4.2),
"f64", "?" = not measured (priced at 4).  cost_weighted_clk = sum(count * clk): the VALU-pipe time one wave-turn needs at best.
"""
from __future__ import annotations

import argparse
import json
import re
import subprocess
import sys
from collections import Counter, defaultdict

CLK2 = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_mov_b32", "v_not_b32", "v_xnor_b32", "v_fmaak_f32", "v_fmamk_f32",
        "v_mul_legacy_f32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"}
CLK4 = {"v_min_f32", "v_max_f32", "v_med3_f32", "v_min3_f32", "v_max3_f32", "v_lshlrev_b32", "v_lshrrev_b32", "v_bfi_b32", "v_alignbit_b32",
        "v_and_or_b32", "v_or3_b32", "v_xor3_b32", "v_add3_u32", "v_lshl_or_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_xad_u32", "v_perm_b32",
        "v_bfe_i32", "v_bfe_u32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_cndmask_b32",
        "v_bitop3_b32", "v_mad_u32_u24", "v_mul_u32_u24", "v_cvt_f32_f16", "v_cvt_f16_f32", "v_cvt_f32_u32", "v_cvt_f32_i32", "v_cvt_u32_f32",
        "v_readfirstlane_b32", "v_readlane_b32", "v_writelane_b32", "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_exp_legacy_f32", "v_log_legacy_f32"}
PK2 = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32"}   # one instruction, two f32 results per lane
# packed 16-bit (two fp16 / u16 results per lane): 4.1-4.3 clk measured (tools/microbench_pk16.hip, profiles/r03_microbench_pk16.txt)
PK16 = {"v_pk_fma_f16", "v_pk_add_f16", "v_pk_mul_f16", "v_pk_min_f16", "v_pk_max_f16", "v_pk_min_u16", "v_pk_max_u16", "v_pk_add_u16", "v_pk_sub_u16",
        "v_pk_sub_i16", "v_pk_min_i16", "v_pk_max_i16", "v_pk_minimum3_f16", "v_pk_maximum3_f16"}


def valu_class(m):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", m)
    if base in CLK2:
        return "2", 2
    if base in CLK4 or base.startswith("v_cmp_") or base.startswith("v_cmpx_"):
        return "4", 4
    if base in TRANS:
        return "8", 8.2     # measured 8.1-8.4 (tools/microbench_valu2.hip, profiles/r02_microbench_valu2.txt)
    if base in PK16:
        return "pk16", 4.2
    if base in PK2:
        return "pk", 4.7    # measured 4.5-4.8 per instruction: two f32 results per lane at the rate of two plain ops
    if base.endswith("_f64") or "_f64_" in base:
        return "f64", 8
    return "?", 4


def unit_of(m):
    if m.startswith("v_"):
        return "valu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("scratch_"):
        return "spill"      # register spills / reloads (private memory, served by L1/L2)
    if m.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if m.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if m in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_endpgm") or m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_waitcnt")):
        return "ctrl"
    if m.startswith("s_"):
        return "salu"
    return "other"


INSTR = re.compile(r"^\s+([a-z][a-z0-9_]+)\b(.*)$")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")


def parse_functions(text):
    """-> {mangled name: [lines]} for every kernel (.amdhsa_kernel) in the file"""
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M))
    out, cur, name = {}, None, None
    for line in text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and m.group(1) in kernels:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur, name = None, None
                continue
            cur.append(line)
    return out


def blocks_of(lines):
    """basic blocks: list of dict(label, instrs=[(mnemonic, operands)], succ=[block indices])"""
    blocks, index = [], {}
    turn_marks = []
    cur, pending, pending_cold = None, ["entry"], False
    for line in lines:
        if "ldpc.turnloop" in line:
            turn_marks.append(len(blocks) - 1 if cur is not None else len(blocks))   # LDPC_TURN_LOOP(): the block this comment sits in
        if "ldpc.cold" in line and cur is not None:
            cur["cold"] = True            # LDPC_COLD_PATH() marker (ldpc_math.h)
        elif "ldpc.cold" in line:
            pending_cold = True
        line = line.split(";")[0].rstrip()
        lm = LABEL.match(line)
        if lm:
            pending.append(lm.group(1))   # a label starts a new block (consecutive labels alias the same block)
            cur = None
            continue
        im = INSTR.match(line)
        if not im or line.lstrip().startswith("."):
            continue
        mn, ops = im.group(1), im.group(2).strip()
        if cur is None:
            cur = {"label": pending[0] if pending else f"after_{len(blocks)}", "instrs": [], "cold": pending_cold}
            pending_cold = False
            for l in pending:
                index[l] = len(blocks)
            pending = []
            blocks.append(cur)
        cur["instrs"].append((mn, ops))
        if mn.startswith(("s_cbranch", "s_branch")) or mn in ("s_endpgm", "s_setpc_b64"):
            cur = None
    for i, b in enumerate(blocks):
        last = b["instrs"][-1]
        if last[0].startswith("s_cbranch"):
            succ = [index[last[1]]] + ([i + 1] if i + 1 < len(blocks) else [])
        elif last[0] == "s_branch":
            succ = [index[last[1]]]
        elif last[0] in ("s_endpgm", "s_setpc_b64"):
            succ = []
        else:
            succ = [i + 1] if i + 1 < len(blocks) else []
        b["succ"] = succ
    for i in turn_marks:
        if i < len(blocks):
            blocks[i]["turn_mark"] = True
    return blocks


def dominators(blocks):
    n = len(blocks)
    preds = defaultdict(list)
    for i, b in enumerate(blocks):
        for s in b["succ"]:
            preds[s].append(i)
    # iterative dataflow on reverse post-order
    order, seen = [], set()
    stack = [(0, iter(blocks[0]["succ"]))]
    seen.add(0)
    while stack:
        node, it = stack[-1]
        adv = False
        for s in it:
            if s not in seen:
                seen.add(s)
                stack.append((s, iter(blocks[s]["succ"])))
                adv = True
                break
        if not adv:
            order.append(node)
            stack.pop()
    rpo = order[::-1]
    dom = {i: None for i in rpo}
    dom[0] = {0}
    changed = True
    while changed:
        changed = False
        for i in rpo[1:]:
            ps = [dom[p] for p in preds[i] if dom.get(p) is not None]
            new = set.intersection(*ps) | {i} if ps else {i}
            if new != dom[i]:
                dom[i] = new
                changed = True
    return dom, preds


def natural_loops(blocks):
    dom, preds = dominators(blocks)
    loops = {}   # header -> set of blocks
    latches = defaultdict(list)
    for i, b in enumerate(blocks):
        if dom.get(i) is None:
            continue
        for s in b["succ"]:
            if s in dom[i]:   # back edge i -> s
                body = {s, i}
                work = [i]
                while work:
                    x = work.pop()
                    if x == s:
                        continue
                    for p in preds[x]:
                        if p not in body and dom.get(p) is not None:
                            body.add(p)
                            work.append(p)
                loops.setdefault(s, set()).update(body)
                latches[s].append(i)
    return loops, latches, dom


def histogram(blocks, ids):
    units, valu, mnem = Counter(), Counter(), Counter()
    clk = 0
    for i in ids:
        for mn, _ in blocks[i]["instrs"]:
            u = unit_of(mn)
            units[u] += 1
            if u == "valu":
                c, k = valu_class(mn)
                valu[c] += 1
                clk += k
                mnem[re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mn)] += 1
            elif u in ("lds", "vmem", "spill"):
                mnem[mn] += 1
    return {"units": dict(units), "valu_by_class": dict(valu), "valu_cost_weighted_clk": clk, "mnemonics": dict(mnem.most_common())}


def hot_path(blocks, body, header, latch_list, must=frozenset()):
    """The blocks of one ORDINARY turn: the heaviest (most VALU instructions) path from the loop header to a latch
    that avoids the blocks the source marks as cold side paths with LDPC_COLD_PATH() (an assembler comment:
    syndrome-only last turn, convergence snapshot, trace stores)."""
    cold = {i for i in body if blocks[i].get("cold")} - set(must)
    ok = body - cold
    w = {i: sum(1 for m, _ in blocks[i]["instrs"] if unit_of(m) == "valu") for i in ok}
    best, state = {}, {}

    def go(i):   # heaviest path weight from i to a latch (DAG once back edges to the header are dropped)
        if i in best:
            return best[i]
        if state.get(i) == 1:
            return None   # inner cycle: ignore that edge
        state[i] = 1
        cand = [(-1, None)] if i not in latch_list else [(0, None)]
        for s2 in blocks[i]["succ"]:
            if s2 == header or s2 not in ok:
                continue
            r = go(s2)
            if r is not None and r[0] >= 0:
                cand.append((r[0], s2))
        state[i] = 2
        top = max(cand, key=lambda c: c[0])
        best[i] = (top[0] + w[i] if top[0] >= 0 else -1, top[1])
        return best[i]

    sys.setrecursionlimit(100000)
    if header not in ok or go(header)[0] < 0:
        return sorted(ok)
    path, i = [], header
    while i is not None:
        path.append(i)
        i = best[i][1]
    return path


def analyse(name, lines):
    blocks = blocks_of(lines)
    loops, latches, dom = natural_loops(blocks)
    size = lambda ids: sum(len(blocks[i]["instrs"]) for i in ids)
    outer = [h for h in loops if not any(h2 != h and h in loops[h2] for h2 in loops)]
    marked = [i for i, b in enumerate(blocks) if b.get("turn_mark")]
    if marked:
        # the source names its iteration loop (LDPC_TURN_LOOP(): it is nested in a persistent workgroup's loop over frames):
        # the smallest natural loop around each marked block
        picked = []
        for i in marked:
            around = [h for h in loops if i in loops[h]]
            if around:
                picked.append(min(around, key=lambda h: size(loops[h])))
        if picked:
            outer = sorted(set(picked))
    if not outer:
        return None
    big = max(size(loops[h]) for h in outer)
    res = []
    for h in sorted(outer):
        if size(loops[h]) < 0.5 * big:
            continue
        body = loops[h]
        every = set(body)
        for l in latches[h]:
            every &= dom[l]
        every &= body
        res.append({"header": blocks[h]["label"], "blocks": len(body), "instructions": size(body),
                    "hot_turn": histogram(blocks, hot_path(blocks, body, h, latches[h], every)),
                    # (loops that hang off a side path the source marks cold -- trace stores -- do not make a turn data-dependent)
                    "inner_loops": any(h2 != h and h2 in body and loops[h2] < body and not any(blocks[d].get("cold") for d in dom[h2] if d in body)
                                       for h2 in loops),
                    "every_turn": histogram(blocks, sorted(every)), "whole_loop": histogram(blocks, sorted(body))})
    return {"kernel": name, "loops": res}


def demangle(names):
    try:
        p = subprocess.run(["c++filt"] + names, capture_output=True, text=True, timeout=60,
                           env={k: v for k, v in __import__("os").environ.items() if not k.startswith(("LD_PRELOAD", "HSA_TOOLS_", "ROCP_", "ROCPROF"))})
        d = p.stdout.strip().splitlines()
        if len(d) == len(names):
            return dict(zip(names, d))
    except Exception:
        pass
    return {n: n for n in names}


def run(paths, kernel_regex=None):
    out = []
    for path in paths:
        fns = parse_functions(open(path).read())
        dm = demangle(list(fns))
        for name, lines in fns.items():
            if kernel_regex and not re.search(kernel_regex, dm[name]):
                continue
            a = analyse(dm[name], lines)
            if a:
                a["source"] = path
                out.append(a)
    return out


def run_on_source(source_text, include_dir, kernel_regex=None, defines=("LDPC_JIT",)):
    """Compile a translation unit to gfx950 assembly with the tool chain and analyse it: the route for the run-time
    specialised kernels (jit.cc), whose source exists only as a generated string (Code.jit_source in the binding)."""
    import os
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        src, asm = os.path.join(d, "k.hip"), os.path.join(d, "k.s")
        open(src, "w").write(source_text)
        cmd = [hipcc, "-S", "--cuda-device-only", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-w",
               "-I" + include_dir] + ["-D" + x for x in defines] + ["-x", "hip", src, "-o", asm]
        # under a profiler (rocprofv3) the environment preloads a library that initialises the GPU in every child; hipcc
        # exec's clang, and an exec after GPU initialisation takes the box down: the tool chain gets a scrubbed environment
        drop = ("LD_PRELOAD", "HSA_TOOLS_", "ROCP_", "ROCPROF", "ROCTRACER_", "ROCTX_", "HIP_TOOLS_LIB")
        env = {k: v for k, v in os.environ.items() if not k.startswith(drop)}
        subprocess.run(cmd, check=True, capture_output=True, timeout=600, env=env)
        res = run([asm], kernel_regex)
        for r in res:
            r["source"] = "generated translation unit (jit.cc)"
        return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="+")
    ap.add_argument("-k", "--kernel", default=None, help="regex on the demangled kernel name")
    ap.add_argument("-o", "--out", default=None)
    args = ap.parse_args()
    res = run(args.asm, args.kernel)
    for a in res:
        print(a["kernel"])
        for lp in a["loops"]:
            e, w = lp["every_turn"], lp["whole_loop"]
            print(f"  loop at {lp['header']}: {lp['instructions']} instructions in {lp['blocks']} blocks")
            for tag, h in (("hot turn", lp["hot_turn"]), ("every turn", e), ("whole loop", w)):
                v = h["units"].get("valu", 0)
                print(f"    {tag:10s}: VALU {v} (by class {h['valu_by_class']}; cost-weighted {h['valu_cost_weighted_clk']} clk = "
                      f"{h['valu_cost_weighted_clk'] / max(v, 1):.2f} clk/instr)  LDS {h['units'].get('lds', 0)}  spill {h['units'].get('spill', 0)}  SALU {h['units'].get('salu', 0)}  "
                      f"VMEM {h['units'].get('vmem', 0)}  ctrl {h['units'].get('ctrl', 0)}")
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
