// jit.h -- run-time specialisation of the fused quasi-cyclic kernels (jit.cc).
//
// The fused kernels want the graph as COMPILE-TIME constants (rotations as literals, block-column bases in the DS
// offset field, register slots named statically).  For the shipped matrices those instances are built ahead of time
// (fused_split.hip, fused_pk16.hip, fused_layered.hip); for any other single-circulant quasi-cyclic H -- what the reference's
// QC decoders take, src/ECC/Code/LDPC/Fast/Arraylet.hs:68-79 -- the same device source (fused_split_body.h, fused_pk16_body.h,
// fused_layered_body.h) is compiled at context creation (hipcc --genco in a child process, or hiprtc) with a plan and a rotation
// table generated from the code's description, and the code object is cached on disk (key = hash of the generated source, the
// options and the compiler route).
#pragma once
#include <string>
#include <vector>

#include "internal.h"

namespace ldpc {

struct JitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    int threads = 0, frames_per_wg = 0, np = 0, waves_per_eu = 0, device = -1;
    std::string name;      // kernel symbol, as rocprofv3 lists it
    std::string cache_key; // hex
    bool from_cache = false;
    double compile_seconds = 0;
};

// which device body the generated kernel wraps (r03: the packed-fp16 and layered kernels are specialised the same way)
enum JitKind {
    JIT_SPLIT = 0,          // fused_split_body.h   flooding, f32 (min-sum or tanh)
    JIT_PK16 = 1,           // fused_pk16_body.h    flooding, packed fp16, two frames per lane (min-sum)
    JIT_LAYERED = 2,        // fused_layered_body.h layered, f32 (min-sum)
    JIT_LAYERED_PK16 = 3,   // fused_layered_body.h layered, packed fp16 (min-sum)
};
// Why the run-time specialised kernel cannot be built for this code/variant/dtype (nullptr = it can).  dtype: LDPC_F32, or
// LDPC_F16PK for the packed-fp16 kinds.
const char *jit_split_why_not(const ldpc_code &c, int variant, int dtype, int kind = JIT_SPLIT);
// Compile (or load from the disk cache) and load on the CURRENT device.  nullptr + set_error on failure.
JitKernel *jit_split_create(const ldpc_code &c, int variant, int dtype, int kind = JIT_SPLIT);
void jit_destroy(JitKernel *k);
// the generated translation unit (tests and tools/ look at it; also what the cache key hashes)
std::string jit_split_source(const ldpc_code &c, int variant, int dtype, JitKernel *geom_out, int kind = JIT_SPLIT);
// compile only (no device needed): returns the code object; used to pre-warm the cache at build time
int jit_compile_cached(const std::string &source, const std::string &kernel_name, std::vector<char> &code_object, bool *from_cache, double *seconds);
const char *jit_cache_dir();

}  // namespace ldpc
