// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the Tacotron training step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/taco_hip.h"   // prototypes are checked against the definitions

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TACO_OK 0
#define TACO_EINVAL (-22)

// launch-error check used by every C-ABI entry point (never throws, never syncs)
#define TACO_RETURN_LAST() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? TACO_OK : (int)e_; } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

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
