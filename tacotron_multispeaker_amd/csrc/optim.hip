// Optimizer of the Tacotron training step for gfx950: global-norm clip + TF-style Adam + Noam learning rate
// over ONE flat fp32 parameter buffer (multi-tensor by construction), and batch-norm moving statistics.
// Reference: models/tacotron.py:174-202 (tf.train.AdamOptimizer, tf.clip_by_global_norm(.,1.0),
// _learning_rate_decay, UPDATE_OPS); TF semantics SURVEY Appendix A.4, A.10, A.11.
// HBM-bound: 28 B read + 12 B written per parameter.  global_step lives on the device so that a whole
// training step can be replayed from a HIP graph without any host-side scalar.
#include "common.hpp"

__global__ void sumsq_k(const float* __restrict__ x, long n4, long n, double* __restrict__ acc) {
    float s = 0.0f;
    double d = 0.0;
    int cnt = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (++cnt == 64) { d += (double)s; s = 0.0f; cnt = 0; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[n4 * 4 + threadIdx.x]; s += v * v; }
    d += (double)s;
    d = wave_sum_d(d);
    __shared__ double red[4];               // one same-address atomic per workgroup (they serialise)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; long n4;
    const double* gnorm2; const int* step;
    float init_lr, beta1, beta2, eps, clip, gscale; int decay;
    float* info;     // optional [3]: global_norm, learning_rate, scale (written by thread 0)
    const int* err;  // optional: device error word of the persistent cluster kernels; non-zero = this step's gradients are invalid
};

__global__ void adam_k(AdamArgs a) {
    if (a.err && *a.err != 0) return;            // a hand-off timed out somewhere in this step: leave the weights untouched
    const double t = (double)(*a.step + 1);
    double lr = a.init_lr;
    if (a.decay) lr = (double)a.init_lr * sqrt(4000.0) * fmin(t * pow(4000.0, -1.5), 1.0 / sqrt(t));   // tacotron.py:198-202
    const float lr_t = (float)(lr * sqrt(1.0 - pow((double)a.beta2, t)) / (1.0 - pow((double)a.beta1, t)));
    // g holds gscale^-1 times the gradient (data parallel: the SUM over replicas, gscale = 1/world): norm and update use gscale*g
    const float norm = (float)sqrt(*a.gnorm2) * a.gscale;
    const float scale = a.clip / fmaxf(norm, a.clip) * a.gscale;      // tf.clip_by_global_norm
    if (a.info && blockIdx.x == 0 && threadIdx.x == 0) { a.info[0] = norm; a.info[1] = (float)lr; a.info[2] = a.clip / fmaxf(norm, a.clip); }
    const float b1 = a.beta1, b2 = a.beta2, eps = a.eps;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.n4; i += (long)gridDim.x * blockDim.x) {
        float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
        m.x = b1 * m.x + (1.f - b1) * g.x; m.y = b1 * m.y + (1.f - b1) * g.y; m.z = b1 * m.z + (1.f - b1) * g.z; m.w = b1 * m.w + (1.f - b1) * g.w;
        v.x = b2 * v.x + (1.f - b2) * g.x * g.x; v.y = b2 * v.y + (1.f - b2) * g.y * g.y;
        v.z = b2 * v.z + (1.f - b2) * g.z * g.z; v.w = b2 * v.w + (1.f - b2) * g.w * g.w;
        p.x -= lr_t * m.x / (sqrtf(v.x) + eps); p.y -= lr_t * m.y / (sqrtf(v.y) + eps);
        p.z -= lr_t * m.z / (sqrtf(v.z) + eps); p.w -= lr_t * m.w / (sqrtf(v.w) + eps);
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        reinterpret_cast<float4*>(a.p)[i] = p;
    }
}

// UPDATE_OPS of the batch norms + (optionally) global_step += 1 in the same launch; both skipped when *err != 0
__global__ void bn_ema_k(float* __restrict__ mov, const float* __restrict__ batch, int n, float momentum, int* step, const int* err) {
    if (err && *err != 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mov[i] -= (mov[i] - batch[i]) * (1.0f - momentum);
    if (step && i == 0) *step += 1;
}

__global__ void step_inc_k(int* step, const int* err) { if (threadIdx.x == 0 && blockIdx.x == 0 && !(err && *err != 0)) *step += 1; }

__global__ void scale_k(float* __restrict__ x, long n4, float s) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(x)[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        reinterpret_cast<float4*>(x)[i] = v;
    }
}

// Everything the host reads back after a step (train.py:142-152: global_step, loss, loss_regularity; :23-36: the summary scalars)
// packed into 16 doubles, so that ONE small device-to-host copy behind one event replaces four blocking reads:
//  0 mel L1 sum  1 linear L1 sum (all columns)  2 linear L1 sum (priority columns)  3 loss_regularity  4 global norm (before the
//  clip)  5 learning rate  6 clip factor  7 global_step (after the step)  8 error word  9 clusters on the agent-scope fallback
//  10 clusters that ran the placement check  11-15 reserved (0)
__global__ void step_status_k(const double* __restrict__ loss_sums, const double* __restrict__ reg, const float* __restrict__ info,
                              const int* __restrict__ err, const int* __restrict__ step, double* __restrict__ out) {
    const int t = threadIdx.x;
    if (t >= 16) return;
    double v = 0.0;
    if (t < 3 && loss_sums) {           // replica r: {all columns, priority columns}; mel replicas first, linear replicas at +16
        const int base = t == 0 ? 0 : 16, col = t == 2 ? 1 : 0;
        for (int r = 0; r < TACO_L1_REPL; ++r) v += loss_sums[base + 2 * r + col];
    } else if (t == 3) v = reg ? reg[0] : 0.0;
    else if (t >= 4 && t <= 6) v = info ? (double)info[t - 4] : 0.0;
    else if (t == 7) v = step ? (double)*step : 0.0;
    else if (t >= 8 && t <= 10) v = err ? (double)err[t - 8] : 0.0;
    out[t] = v;
}

extern "C" int taco_step_status(const double* loss_sums, const double* reg_sum, const float* info3, const int* err4,
                                const int* global_step, double* out16, hipStream_t stream) {
    if (!out16) return TACO_EINVAL;
    hipLaunchKernelGGL(step_status_k, dim3(1), dim3(64), 0, stream, loss_sums, reg_sum, info3, err4, global_step, out16);
    TACO_RETURN_LAST();
}

extern "C" int taco_sumsq(const float* x, long n, double* acc, hipStream_t stream) {
    if (!x || !acc || n < 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return TACO_EINVAL;
    const long n4 = n / 4;
    long g = (n4 + 255) / 256; if (g > 512) g = 512; if (g < 1) g = 1;
    hipLaunchKernelGGL(sumsq_k, dim3((int)g), dim3(256), 0, stream, x, n4, n, acc);
    TACO_RETURN_LAST();
}

extern "C" int taco_adam_step(float* params, const float* grads, float* m, float* v, long n, const double* gnorm2,
                              const int* global_step, float init_lr, int decay, float beta1, float beta2, float eps, float clip,
                              float grad_scale, float* info3, const int* err, hipStream_t stream) {
    if (!params || !grads || !m || !v || !gnorm2 || !global_step || (n & 3) || !(grad_scale > 0.0f)) return TACO_EINVAL;
    AdamArgs a{params, grads, m, v, n / 4, gnorm2, global_step, init_lr, beta1, beta2, eps, clip, grad_scale, decay, info3, err};
    long g = (a.n4 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(adam_k, dim3((int)g), dim3(256), 0, stream, a);
    TACO_RETURN_LAST();
}

extern "C" int taco_bn_ema(float* moving, const float* batch, int n, float momentum, int* global_step, const int* err,
                           hipStream_t stream) {
    if (!moving || !batch || n < 1) return TACO_EINVAL;
    hipLaunchKernelGGL(bn_ema_k, dim3(cdiv(n, 256)), dim3(256), 0, stream, moving, batch, n, momentum, global_step, err);
    TACO_RETURN_LAST();
}

extern "C" int taco_step_inc(int* global_step, const int* err, hipStream_t stream) {
    if (!global_step) return TACO_EINVAL;
    hipLaunchKernelGGL(step_inc_k, dim3(1), dim3(64), 0, stream, global_step, err);
    TACO_RETURN_LAST();
}

// one memset node (graph capturable) for the buffers that must start a step at zero: gradients + reduction scratch
extern "C" int taco_zero(void* p, size_t bytes, hipStream_t stream) {
    if (!p || (bytes & 15) || (reinterpret_cast<uintptr_t>(p) & 15)) return TACO_EINVAL;
    if (bytes && hipMemsetAsync(p, 0, bytes, stream) != hipSuccess) return TACO_EINVAL;
    return TACO_OK;
}

extern "C" int taco_scale(float* x, long n, float s, hipStream_t stream) {
    if (n & 3) return TACO_EINVAL;
    long g = (n / 4 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(scale_k, dim3((int)g), dim3(256), 0, stream, x, n / 4, s);
    TACO_RETURN_LAST();
}

// One wave that does nothing for `us` microseconds (constant 100 MHz wall clock): used by the host ONCE per engine to find out which of
// its streams share a hardware queue -- two such kernels on streams that share one run back to back, on different queues side by side.
__global__ void spin_us_k(long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int taco_spin_us(int us, hipStream_t stream) {
    if (us < 0 || us > 100000) return TACO_EINVAL;
    hipLaunchKernelGGL(spin_us_k, dim3(1), dim3(64), 0, stream, (long)us * 100);
    TACO_RETURN_LAST();
}

// One wave that returns once *counter >= target: puts work of its stream behind something only a RUNNING kernel can signal (the
// residency counter of taco_attn_rnn_bwd_chunk).  Bounded by the wall clock (~0.5 s at 100 MHz), then the error word is set; the caller
// enqueues it only AFTER it has issued the launch that signals the counter, so the wait is device time, never host time.
__global__ void wait_count_k(const int* __restrict__ counter, int target, int* err) {
    const long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > 50000000L) { if (err) atomicExch(err, 4); break; }      // (4: told apart from a hand-off timeout in logs; any non-zero word invalidates the step)
        __builtin_amdgcn_s_sleep(16);
    }
}

extern "C" int taco_wait_count(const int* counter, int target, int* err, hipStream_t stream) {
    if (!counter || target < 0) return TACO_EINVAL;
    hipLaunchKernelGGL(wait_count_k, dim3(1), dim3(64), 0, stream, counter, target, err);
    TACO_RETURN_LAST();
}
