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
