// Diagnostic micro-benchmark of the cluster all-gather used by the persistent recurrence kernels: CW workgroups
// per cluster publish LEN granules each and gather everyone else's, ITERS times (one __syncthreads per round).
// Reported by scripts/dev_xchg.py as microseconds per exchange round; not on the product path.
#include "common.hpp"
typedef unsigned long long u64;

struct XB { u64* x; int* err; int cw, len, iters, same_xcd, sleep, threads; float* sink; };

__global__ __launch_bounds__(512) void xchg_bench_k(XB p) {
    const int tid = threadIdx.x;
    const int nclus = gridDim.x / p.cw;
    int w, cl;
    if (p.same_xcd && (nclus & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; w = q % p.cw; cl = (q / p.cw) * 8 + xcd; }
    else { w = blockIdx.x % p.cw; cl = blockIdx.x / p.cw; }
    u64* X0 = p.x + (long)cl * 2 * p.cw * p.len;      // two alternating regions (a round may only overwrite data
                                                      // that every peer has provably consumed)
    __shared__ float lds[4096];
    float acc = 0.f;
    for (int it = 0; it < p.iters; ++it) {
        const unsigned epoch = it + 1;
        u64* X = X0 + (it & 1) * p.cw * p.len;
        if (tid < p.len) {
            const float v = (float)(it + tid + w);
            __hip_atomic_store(X + w * p.len + tid, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            lds[g & 4095] = __uint_as_float((unsigned)x);
        }
        __syncthreads();
        acc += lds[tid & 4095];
        __syncthreads();
    }
    if (acc == 123.456f) p.sink[0] = acc;
}

extern "C" int taco_xchg_bench(void* xchg, int* err, float* sink, int nclus, int cw, int len, int iters, int same_xcd, int sleep,
                               int threads, hipStream_t st) {
    if (!xchg || !err || !sink || nclus * cw > 256 || len > 512 || threads > 512) return TACO_EINVAL;
    if (hipMemsetAsync(xchg, 0, (size_t)nclus * 2 * cw * len * sizeof(u64), st) != hipSuccess) return TACO_EINVAL;
    XB p{(u64*)xchg, err, cw, len, iters, same_xcd, sleep, threads, sink};
    hipLaunchKernelGGL(xchg_bench_k, dim3(nclus * cw), dim3(threads), 0, st, p);
    TACO_RETURN_LAST();
}
