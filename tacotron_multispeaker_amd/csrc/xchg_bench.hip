// Diagnostic micro-benchmark of the cluster all-gather used by the persistent recurrence kernels: CW workgroups
// per cluster publish LEN granules each and gather everyone else's, ITERS times (one __syncthreads per round).
// Reported by scripts/dev_xchg.py as microseconds per exchange round; not on the product path (csrc/dev_tools.h).
#include "common.hpp"
#include "dev_tools.h"
#include "xcd_granule.hpp"

struct XB { u64* x; int* err; int cw, len, iters, same_xcd, sleep, threads, mode; float* sink; };

__global__ __launch_bounds__(512) void xchg_bench_k(XB p) {
    const int tid = threadIdx.x;
    const int nclus = gridDim.x / p.cw;
    int w, cl;
    if (p.same_xcd && (nclus & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; w = q % p.cw; cl = (q / p.cw) * 8 + xcd; }
    else { w = blockIdx.x % p.cw; cl = blockIdx.x / p.cw; }
    u64* X0 = p.x + (long)cl * (2 * p.cw * p.len + 16) + 16;      // 16 placement slots, then two alternating regions
    __shared__ float lds[4096];
    __shared__ int local_s;
    // mode 1: the L2-local granule form is legal only if the whole cluster sits on one XCD -- verified, never assumed
    bool local = false;
    if (p.mode == 1) {
        if (tid == 0) local_s = cluster_on_one_xcd(p.x + (long)cl * (2 * p.cw * p.len + 16), w, p.cw, p.err) ? 1 : 0;
        __syncthreads();
        local = local_s != 0;
        if (!local && tid == 0) atomicExch(p.err, 2);
    }
    float acc = 0.f;
    for (int it = 0; it < p.iters; ++it) {
        const unsigned epoch = it + 1;
        u64* X = X0 + (it & 1) * p.cw * p.len;
        if (tid < p.len) {
            const float v = (float)(it + tid + w);
            if (local) put_granule_xcd(X + w * p.len + tid, epoch, v);
            else put_granule(X + w * p.len + tid, epoch, v);
        }
        const int tot = (p.cw - 1) * p.len;
        for (int g = tid; g < tot; g += blockDim.x) {
            const int peer = g / p.len, jj = g - peer * p.len;
            const int pw = peer + (peer >= w ? 1 : 0);
            const u64* q = X + pw * p.len + jj;
            u64 x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while ((unsigned)(x >> 32) != epoch) {
                if (++spins > (1 << 22)) { atomicExch(p.err, 1); break; }
                if (p.sleep) __builtin_amdgcn_s_sleep(1);
                x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const float got = __uint_as_float((unsigned)x);
            if (got != (float)(it + jj + pw)) atomicExch(p.err, 3);       // every word is checked
            lds[g & 4095] = got;
        }
        __syncthreads();
        acc += lds[tid & 4095];
        __syncthreads();
    }
    if (acc == 123.456f) p.sink[0] = acc;
}

extern "C" int taco_dev_xchg_bench(void* xchg, int* err, float* sink, int nclus, int cw, int len, int iters, int same_xcd,
                                   int sleep, int threads, int mode, hipStream_t st) {
    if (!xchg || !err || !sink || nclus * cw > 256 || len > 512 || threads > 512) return TACO_EINVAL;
    if (hipMemsetAsync(xchg, 0, (size_t)nclus * (2 * cw * len + 16) * sizeof(u64), st) != hipSuccess) return TACO_EINVAL;
    XB p{(u64*)xchg, err, cw, len, iters, same_xcd, sleep, threads, mode, sink};
    hipLaunchKernelGGL(xchg_bench_k, dim3(nclus * cw), dim3(threads), 0, st, p);
    TACO_RETURN_LAST();
}

// ---- synthetic background load (dev_tools.h: taco_dev_load) ------------------------------------------------------------
__global__ __launch_bounds__(256) void dev_load_k(const float4* __restrict__ p, long n4, int reps, int mode, float* sink) {
    extern __shared__ float dl_smem[];
    float acc = 0.f;
    if (mode == 0) {
        for (int r = 0; r < reps; ++r)
            for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
                const float4 v = p[i];
                acc += v.x + v.y + v.z + v.w;
            }
    } else if (mode == 3) {                      // fp32 MFMA at ~50 % duty: bursts of 8, then the SIMD is left to other waves
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        const float a = (float)threadIdx.x * 1e-9f, b = 1.0f;
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
            __builtin_amdgcn_s_sleep(8);         // 8 x 64 cycles
        }
        acc = c[0] + c[5];
    } else if (mode == 4) {                      // fp32 atomic adds into a small region (split-K partial sums of a weight gradient)
        float* q = reinterpret_cast<float*>(const_cast<float4*>(p));
        const long n = n4 * 4;
        for (int r = 0; r < reps; ++r)
            for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) atomicAdd(q + i, 1e-30f);
    } else if (mode == 2) {                      // packed fp32 VALU FMA chains, 8 independent accumulators, no memory traffic
        f2 c0 = {0.f, 1.f}, c1 = {1.f, 2.f}, c2 = {2.f, 3.f}, c3 = {3.f, 4.f}, c4 = {4.f, 5.f}, c5 = {5.f, 6.f}, c6 = {6.f, 7.f}, c7 = {7.f, 8.f};
        const f2 a = {1.0f + (float)threadIdx.x * 1e-9f, 0.999f}, b = {1e-6f, 1e-7f};
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                c0 = __builtin_elementwise_fma(c0, a, b); c1 = __builtin_elementwise_fma(c1, a, b);
                c2 = __builtin_elementwise_fma(c2, a, b); c3 = __builtin_elementwise_fma(c3, a, b);
                c4 = __builtin_elementwise_fma(c4, a, b); c5 = __builtin_elementwise_fma(c5, a, b);
                c6 = __builtin_elementwise_fma(c6, a, b); c7 = __builtin_elementwise_fma(c7, a, b);
            }
        }
        acc = c0.x + c1.y + c2.x + c3.y + c4.x + c5.y + c6.x + c7.y;
    } else {
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        const float a = (float)threadIdx.x * 1e-9f, b = 1.0f;
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int k = 0; k < 16; ++k) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
        }
        acc = c[0] + c[5];
    }
    if (acc == 123.456f) { sink[0] = acc; dl_smem[0] = acc; }
}

extern "C" int taco_dev_load(const float* p, long bytes, int reps, int mode, int wgs, int lds_bytes, float* sink, hipStream_t st) {
    if (!sink || wgs < 1 || reps < 0 || lds_bytes < 0 || lds_bytes > 160 * 1024 || (mode == 0 && !p)) return TACO_EINVAL;
    static DevMask attr{0};
    if (lds_bytes > 48 * 1024 && ensure_dyn_lds((const void*)dev_load_k, 160 * 1024, attr) != TACO_OK) return TACO_EINVAL;
    hipLaunchKernelGGL(dev_load_k, dim3(wgs), dim3(256), (size_t)lds_bytes, st, (const float4*)p, bytes / 16, reps, mode, sink);
    TACO_RETURN_LAST();
}
