// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the Tacotron training step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/taco_hip.h"   // prototypes are checked against the definitions

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
// 4 reduction terms of two rows that share the weights (w0..w3) as four v_pk_fma_f32: packed fp32 issues two FMAs per
// lane per instruction, which halves the VALU issue time of the register-resident recurrent matvecs
__device__ __forceinline__ void pk_dot4x2(const float4& v0, const float4& v1, float w0, float w1, float w2, float w3, f2& a0, f2& a1) {
    const f2 wa = {w0, w1}, wb = {w2, w3};
    a0 = __builtin_elementwise_fma((f2){v0.x, v0.y}, wa, a0); a0 = __builtin_elementwise_fma((f2){v0.z, v0.w}, wb, a0);
    a1 = __builtin_elementwise_fma((f2){v1.x, v1.y}, wa, a1); a1 = __builtin_elementwise_fma((f2){v1.z, v1.w}, wb, a1);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TACO_OK 0
#define TACO_EINVAL (-22)

// launch-error check used by every C-ABI entry point (never throws, never syncs)
#define TACO_RETURN_LAST() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? TACO_OK : (int)e_; } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (kernel, device) pair: remembered per DEVICE in a bit mask next to
// the kernel (a cache of an idempotent call, safe from any thread), never as "done once per process".
typedef std::atomic<unsigned long long> DevMask;
static inline int ensure_dyn_lds(const void* kernel, int bytes, DevMask& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TACO_EINVAL;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return TACO_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return TACO_EINVAL;
    done.fetch_or(bit, std::memory_order_release);
    return TACO_OK;
}

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_TANH = 3 };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh through exp: accurate to ~1e-7 relative, saturates cleanly
__device__ __forceinline__ float tanhf_(float x) {
    float ax = fabsf(x);
    float e = expf(-2.0f * ax);
    float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SIGMOID: return sigmoidf_(x);
        case ACT_TANH: return tanhf_(x);
        default: return x;
    }
}

// ---- latency-critical variants for the persistent recurrence kernels ----------------------------------------------
// v_exp_f32 / v_rcp_f32 based (1 ulp each): abs error ~1e-7, ~6 instructions instead of ~40
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
// Sum over groups of PARTS adjacent lanes (PARTS = 2,4,8,16,32; groups aligned) with DPP cross-lane adds: one VALU
// instruction per level instead of a ds_bpermute round trip through the LDS crossbar (~100+ cycles each).
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int PARTS>
__device__ __forceinline__ float group_sum(float v) {
    if (PARTS >= 2) v = dpp_add<0xB1>(v);
    if (PARTS >= 4) v = dpp_add<0x4E>(v);
    if (PARTS >= 8) v = dpp_add<0x141>(v);
    if (PARTS >= 16) v = dpp_add<0x140>(v);
    if (PARTS >= 32) v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // lane ^ 16
    return v;
}

// full-wave (64 lane) reductions without LDS round trips: DPP inside each 16-lane row, then the four row results are
// combined through scalar registers (v_readlane)
__device__ __forceinline__ float wave_sum_fast(float v) {
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48));
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float wave_max_fast(float v) {
    v = dpp_max<0xB1>(v); v = dpp_max<0x4E>(v); v = dpp_max<0x141>(v); v = dpp_max<0x140>(v);
    const int i = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}

// Workgroup barrier for the step loops of the persistent recurrence kernels: orders LDS traffic only.  __syncthreads() also
// drains the vector-memory counter (s_waitcnt vmcnt(0)), i.e. every barrier would wait for the step's global stores of saved
// activations (write acknowledgements, ~300-500 cycles) and for the one-step-ahead prefetch loads (an Infinity-Cache / HBM
// round trip, 550-900 cycles) although nothing on the dependent chain needs them: threads of a workgroup never read global
// data written by other threads of the same launch, and a prefetched value is waited for by the compiler at its first use.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 64-lane wavefront reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
