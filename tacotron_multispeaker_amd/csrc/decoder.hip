// Attention decoder of the Tacotron training step for gfx950 (teacher forcing).
//
// Reference: models/tacotron.py:66-97 (AttentionWrapper(DecoderPrenetWrapper(GRUCell(256)),
// BahdanauAttention(256, enc)), ConcatOutputAndAttentionWrapper, MultiRNNCell[OutputProjectionWrapper,
// ResidualWrapper(GRUCell(256)) x2], OutputProjectionWrapper, dynamic_decode(BasicDecoder, TacoTrainingHelper)),
// models/rnn_wrappers.py:22-24,50-52, models/helpers.py:41-82; TF semantics: SURVEY Appendix A.5, A.7, A.8.
//
// Restructuring (MI355X-first, results identical): under teacher forcing the attention recurrence
// (prenet -> attention GRU -> Bahdanau attention -> context) does NOT depend on the two residual decoder
// GRUs, so the decoder is split into phase A (attention recurrence, S dependent steps), three hoisted
// MFMA GEMMs over all S steps (concat projection, GRU input projections, output projection) and two plain
// GRU(256) recurrences.  Per step only skinny [N<=32 x K] x [K x n] products remain; they run on
// v_mfma_f32_16x16x4_f32 with the K dimension split over the 4 waves of a workgroup (latency, not
// throughput, is what matters), weights streamed from L2, fused gate epilogues.  The attention score tile
// keys[n] (Ti x 256) is read coalesced (16 B/lane), tanh'ed and reduced over the 256 axis with wavefront
// reductions; the softmax over ALL Ti positions (no memory mask, A.7) is a workgroup reduction.
//
// All per-step tensors are laid out [N, S, dim] (row stride S*dim) so the hoisted GEMMs see plain matrices.
#include "attn_cluster.hpp"

// ---------------------------------------------------------------------------------------------------
// skinny GEMM: out[M<=32 rows/tile, 16 cols/workgroup] = sum over up to 2 (A,B) segments, K split over 8 waves
// ---------------------------------------------------------------------------------------------------
enum {
    M_PRENET1 = 0, M_BIAS_RELU, M_LINEAR, M_GRU_GATES, M_GRU_CAND, M_GRU_BWD1, M_GRU_BWD2, M_RELU_MASK
};

#define SK_WAVES 8

struct Skinny {
    const float* A0; int lda0; int K0; const float* B0; int ldb0;
    const float* A1; int lda1; int K1; const float* B1; int ldb1;
    int M, N, bt, mode, amode, Hd;
    const float* p[6]; int ld[6];
    float* o[4]; int ldo[4];
};

// One wave's share of a segment: chunks of 16 k (4 per lane-quarter); loads of up to 4 chunks are issued
// back-to-back before the first MFMA so a wave pays ~one L2 round trip per 64 k instead of one per 16 k.
template <bool BT>
__device__ __forceinline__ void skinny_acc(f32x4 (&acc)[2], const Skinny& p, const float* __restrict__ A, int lda, int K,
                                           const float* __restrict__ B, int ldb, int m0, int n0, int wave, int lane,
                                           bool amode1) {
    const int i = lane & 15, kq = lane >> 4;
    const int col = n0 + i;
    const bool col_ok = col < p.N;
    const int chunks = (K + 15) / 16;
    const int per = (chunks + SK_WAVES - 1) / SK_WAVES;
    const int c0 = wave * per, c1 = min(chunks, c0 + per);
    const int row0 = m0 + i, row1 = m0 + 16 + i;
    const bool r0ok = row0 < p.M, r1ok = row1 < p.M;
    for (int cb = c0; cb < c1; cb += 4) {
        float4 a0[4], a1[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = (cb + u) * 16 + 4 * kq;
            const bool k_ok = (cb + u) < c1 && k < K;
            a0[u] = make_float4(0.f, 0.f, 0.f, 0.f); a1[u] = a0[u]; b[u] = a0[u];
            if (r0ok && k_ok) a0[u] = *reinterpret_cast<const float4*>(A + (long)row0 * lda + k);
            if (r1ok && k_ok) a1[u] = *reinterpret_cast<const float4*>(A + (long)row1 * lda + k);
            if (BT) {
                if (col_ok && k_ok) b[u] = *reinterpret_cast<const float4*>(B + (long)col * ldb + k);
            } else if (col_ok && k_ok) {
                b[u].x = B[(long)(k + 0) * ldb + col]; b[u].y = B[(long)(k + 1) * ldb + col];
                b[u].z = B[(long)(k + 2) * ldb + col]; b[u].w = B[(long)(k + 3) * ldb + col];
            }
            if (amode1) {   // a = dhT * (1-u) * (1-c^2): candidate pre-activation gradient recomputed on load
                if (r0ok && k_ok) {
                    const float4 uu = *reinterpret_cast<const float4*>(p.p[2] + (long)row0 * p.ld[2] + k);
                    const float4 cc = *reinterpret_cast<const float4*>(p.p[3] + (long)row0 * p.ld[3] + k);
                    a0[u].x *= (1.f - uu.x) * (1.f - cc.x * cc.x); a0[u].y *= (1.f - uu.y) * (1.f - cc.y * cc.y);
                    a0[u].z *= (1.f - uu.z) * (1.f - cc.z * cc.z); a0[u].w *= (1.f - uu.w) * (1.f - cc.w * cc.w);
                }
                if (r1ok && k_ok) {
                    const float4 uu = *reinterpret_cast<const float4*>(p.p[2] + (long)row1 * p.ld[2] + k);
                    const float4 cc = *reinterpret_cast<const float4*>(p.p[3] + (long)row1 * p.ld[3] + k);
                    a1[u].x *= (1.f - uu.x) * (1.f - cc.x * cc.x); a1[u].y *= (1.f - uu.y) * (1.f - cc.y * cc.y);
                    a1[u].z *= (1.f - uu.z) * (1.f - cc.z * cc.z); a1[u].w *= (1.f - uu.w) * (1.f - cc.w * cc.w);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].x, b[u].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].x, b[u].x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].y, b[u].y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].y, b[u].y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].z, b[u].z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].z, b[u].z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].w, b[u].w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u].w, b[u].w, acc[1], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(SK_WAVES * 64) void skinny_k(Skinny p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 32;
    f32x4 acc[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][r] = 0.0f;
    if (p.bt) {
        if (p.A0) skinny_acc<true>(acc, p, p.A0, p.lda0, p.K0, p.B0, p.ldb0, m0, n0, wave, lane, p.amode == 1);
        if (p.A1) skinny_acc<true>(acc, p, p.A1, p.lda1, p.K1, p.B1, p.ldb1, m0, n0, wave, lane, false);
    } else {
        if (p.A0) skinny_acc<false>(acc, p, p.A0, p.lda0, p.K0, p.B0, p.ldb0, m0, n0, wave, lane, false);
        if (p.A1) skinny_acc<false>(acc, p, p.A1, p.lda1, p.K1, p.B1, p.ldb1, m0, n0, wave, lane, false);
    }
    __shared__ float red[SK_WAVES][2][256];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rt][lane * 4 + r] = acc[rt][r];
    __syncthreads();
    {
        const int rt = tid >> 8, e = tid & 255;       // 512 threads <-> 2 x 256 outputs
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) v += red[w][rt][e];
        const int l = e >> 2, r = e & 3;
        const int row = m0 + rt * 16 + (l >> 4) * 4 + r;
        const int col = n0 + (l & 15);
        if (row >= p.M || col >= p.N) return;
        const long R = row;
        switch (p.mode) {
            case M_PRENET1:
                p.o[0][R * p.ldo[0] + col] = fmaxf(v + p.p[0][R * p.ld[0] + col], 0.0f);
                break;
            case M_BIAS_RELU:
                p.o[0][R * p.ldo[0] + col] = fmaxf(v + p.p[0][col], 0.0f);
                break;
            case M_LINEAR: {
                float x = v;
                if (p.p[0]) x += p.p[0][col];
                if (p.p[1]) x += p.p[1][R * p.ld[1] + col];
                if (p.p[2]) x += p.p[2][R * p.ld[2] + col];
                p.o[0][R * p.ldo[0] + col] = x;
            } break;
            case M_GRU_GATES: {
                float pre = v;
                if (p.p[0]) pre += p.p[0][R * p.ld[0] + col];
                if (p.p[1]) pre += p.p[1][col];
                const float g = sigmoidf_(pre);
                if (col < p.Hd) {
                    p.o[0][R * p.ldo[0] + col] = g;
                    p.o[2][R * p.ldo[2] + col] = g * p.p[2][R * p.ld[2] + col];
                } else {
                    p.o[1][R * p.ldo[1] + col - p.Hd] = g;
                }
            } break;
            case M_GRU_CAND: {
                float pre = v;
                if (p.p[0]) pre += p.p[0][R * p.ld[0] + col];
                if (p.p[1]) pre += p.p[1][col];
                const float c = tanhf_(pre);
                const float u = p.p[3][R * p.ld[3] + col];
                const float hp = p.p[2][R * p.ld[2] + col];
                const float hn = u * hp + (1.0f - u) * c;
                p.o[0][R * p.ldo[0] + col] = c;
                p.o[1][R * p.ldo[1] + col] = hn;
                if (p.o[2]) p.o[2][R * p.ldo[2] + col] = p.p[4][R * p.ld[4] + col] + hn;
            } break;
            case M_GRU_BWD1: {
                // v = drh; p0 = dhT, p1 = r, p2 = u, p3 = c, p4 = hprev; o0 = dxp (r|u|c grads), o1 = dh partial
                const float dhT = p.p[0][R * p.ld[0] + col];
                const float r = p.p[1][R * p.ld[1] + col], u = p.p[2][R * p.ld[2] + col];
                const float c = p.p[3][R * p.ld[3] + col], hp = p.p[4][R * p.ld[4] + col];
                const float du = dhT * (hp - c);
                const float dcp = dhT * (1.0f - u) * (1.0f - c * c);
                const float dr = v * hp;
                float* dx = p.o[0] + R * p.ldo[0];
                dx[col] = dr * r * (1.0f - r);
                dx[p.Hd + col] = du * u * (1.0f - u);
                dx[2 * p.Hd + col] = dcp;
                p.o[1][R * p.ldo[1] + col] = dhT * u + v * r;
            } break;
            case M_GRU_BWD2: {
                float x = v + p.p[0][R * p.ld[0] + col];
                if (p.p[1]) x += p.p[1][R * p.ld[1] + col];
                p.o[0][R * p.ldo[0] + col] = x;
            } break;
            case M_RELU_MASK:
                p.o[0][R * p.ldo[0] + col] = p.p[0][R * p.ld[0] + col] > 0.0f ? v : 0.0f;
                break;
        }
    }
}

static inline void launch_skinny(const Skinny& p, hipStream_t st) {
    hipLaunchKernelGGL(skinny_k, dim3(cdiv(p.N, 16), cdiv(p.M, 32)), dim3(SK_WAVES * 64), 0, st, p);
}

// ---------------------------------------------------------------------------------------------------
// Bahdanau attention, one decoder step
// ---------------------------------------------------------------------------------------------------
// scores: e[n,t] = sum_d v[d] * tanh(keys[n,t,d] + q[n,d]); one wave per t, 4 d per lane (D = 256)
__global__ __launch_bounds__(256) void attn_scores_k(const float* __restrict__ keys, const float* __restrict__ q, int ldq,
                                                    const float* __restrict__ v, float* __restrict__ e, int lde, int Ti) {
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 qv = *reinterpret_cast<const float4*>(q + (long)n * ldq + lane * 4);
    const float4 vv = *reinterpret_cast<const float4*>(v + lane * 4);
    const int t0 = blockIdx.y * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + wave * 4 + k;
        if (t >= Ti) break;
        const float4 kv = *reinterpret_cast<const float4*>(keys + ((long)n * Ti + t) * 256 + lane * 4);
        float s = vv.x * fast_tanh(kv.x + qv.x) + vv.y * fast_tanh(kv.y + qv.y) + vv.z * fast_tanh(kv.z + qv.z) +
                  vv.w * fast_tanh(kv.w + qv.w);
        s = wave_sum(s);
        if (lane == 0) e[(long)n * lde + t] = s;
    }
}

// softmax over all Ti + context: a = softmax(e[n,:]) (written in place by blockIdx.y == 0),
// ctx[n, d] = sum_t a[t] * mem[n,t,d] for the 64 columns d of this block
__global__ __launch_bounds__(256) void attn_softmax_ctx_k(float* __restrict__ e, int lde, const float* __restrict__ mem,
                                                         float* __restrict__ ctx, int ldc, int Ti) {
    extern __shared__ float sm[];           // a[Ti] then 4 partial rows of 64
    float* a = sm;
    float* part = sm + ((Ti + 3) & ~3);
    __shared__ float redw[4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* er = e + (long)n * lde;
    float mx = -INFINITY;
    for (int t = tid; t < Ti; t += 256) { const float x = er[t]; a[t] = x; mx = fmaxf(mx, x); }
    mx = wave_max(mx);
    if (lane == 0) redw[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redw[0], redw[1]), fmaxf(redw[2], redw[3]));
    __syncthreads();
    float s = 0.0f;
    for (int t = tid; t < Ti; t += 256) { const float x = expf(a[t] - mx); a[t] = x; s += x; }
    s = wave_sum(s);
    if (lane == 0) redw[wave] = s;
    __syncthreads();
    const float inv = 1.0f / (redw[0] + redw[1] + redw[2] + redw[3]);
    for (int t = tid; t < Ti; t += 256) {
        const float x = a[t] * inv;
        a[t] = x;
        if (blockIdx.y == 0) er[t] = x;
    }
    __syncthreads();
    const int d = blockIdx.y * 64 + lane;
    float c = 0.0f;
    for (int t = wave; t < Ti; t += 4) c = fmaf(a[t], mem[((long)n * Ti + t) * 256 + d], c);
    part[wave * 64 + lane] = c;
    __syncthreads();
    if (wave == 0) ctx[(long)n * ldc + d] = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
}

// backward A1: da[n,t] = dctx[n,:] . mem[n,t,:];  dmem[n,t,:] += a[t] * dctx[n,:]   (one wave per t)
__global__ __launch_bounds__(256) void attn_bwd_da_k(const float* __restrict__ mem, const float* __restrict__ a, int lda_,
                                                    const float* __restrict__ dctx, int lddc, float* __restrict__ da,
                                                    float* __restrict__ dmem, int Ti, const float* __restrict__ da_ext, int ldext) {
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 g = *reinterpret_cast<const float4*>(dctx + (long)n * lddc + lane * 4);
    const int t0 = blockIdx.y * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + wave * 4 + k;
        if (t >= Ti) break;
        const long off = ((long)n * Ti + t) * 256 + lane * 4;
        const float4 m = *reinterpret_cast<const float4*>(mem + off);
        float s = wave_sum(g.x * m.x + g.y * m.y + g.z * m.z + g.w * m.w);
        if (lane == 0) da[(long)n * Ti + t] = s + (da_ext ? da_ext[(long)n * ldext + t] : 0.0f);
        const float at = a[(long)n * lda_ + t];
        float4 dm = *reinterpret_cast<float4*>(dmem + off);
        dm.x = fmaf(at, g.x, dm.x); dm.y = fmaf(at, g.y, dm.y); dm.z = fmaf(at, g.z, dm.z); dm.w = fmaf(at, g.w, dm.w);
        *reinterpret_cast<float4*>(dmem + off) = dm;
    }
}

// backward A2: de = a*(da - sum a*da); th = tanh(keys+q); dpre = de*v*(1-th^2); dkeys += dpre;
// dq[n,:] += sum_t dpre (atomics into the zeroed per-step slot); dvpart[n, chunk, :] += sum_t de*th
__global__ __launch_bounds__(256) void attn_bwd_score_k(const float* __restrict__ keys, const float* __restrict__ q, int ldq,
                                                       const float* __restrict__ v, const float* __restrict__ a, int lda_,
                                                       const float* __restrict__ da, float* __restrict__ dkeys,
                                                       float* __restrict__ dq, int lddq, float* __restrict__ dvpart, int Ti) {
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float redw[4];
    __shared__ float4 pq[4][64], pv[4][64];
    float dot = 0.0f;
    for (int t = tid; t < Ti; t += 256) dot = fmaf(a[(long)n * lda_ + t], da[(long)n * Ti + t], dot);
    dot = wave_sum(dot);
    if (lane == 0) redw[wave] = dot;
    __syncthreads();
    dot = redw[0] + redw[1] + redw[2] + redw[3];
    const float4 qv = *reinterpret_cast<const float4*>(q + (long)n * ldq + lane * 4);
    const float4 vv = *reinterpret_cast<const float4*>(v + lane * 4);
    float4 sq = make_float4(0.f, 0.f, 0.f, 0.f), sv = sq;
    const int t0 = blockIdx.y * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + wave * 4 + k;
        if (t >= Ti) break;
        const float at = a[(long)n * lda_ + t];
        const float de = at * (da[(long)n * Ti + t] - dot);
        const long off = ((long)n * Ti + t) * 256 + lane * 4;
        const float4 kv = *reinterpret_cast<const float4*>(keys + off);
        float4 th;
        th.x = fast_tanh(kv.x + qv.x); th.y = fast_tanh(kv.y + qv.y); th.z = fast_tanh(kv.z + qv.z); th.w = fast_tanh(kv.w + qv.w);
        float4 dp;
        dp.x = de * vv.x * (1.f - th.x * th.x); dp.y = de * vv.y * (1.f - th.y * th.y);
        dp.z = de * vv.z * (1.f - th.z * th.z); dp.w = de * vv.w * (1.f - th.w * th.w);
        float4 dk = *reinterpret_cast<float4*>(dkeys + off);
        dk.x += dp.x; dk.y += dp.y; dk.z += dp.z; dk.w += dp.w;
        *reinterpret_cast<float4*>(dkeys + off) = dk;
        sq.x += dp.x; sq.y += dp.y; sq.z += dp.z; sq.w += dp.w;
        sv.x = fmaf(de, th.x, sv.x); sv.y = fmaf(de, th.y, sv.y); sv.z = fmaf(de, th.z, sv.z); sv.w = fmaf(de, th.w, sv.w);
    }
    pq[wave][lane] = sq; pv[wave][lane] = sv;
    __syncthreads();
    if (wave == 0) {
        float4 x = pq[0][lane], y = pv[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 x2 = pq[w][lane], y2 = pv[w][lane];
            x.x += x2.x; x.y += x2.y; x.z += x2.z; x.w += x2.w;
            y.x += y2.x; y.y += y2.y; y.z += y2.z; y.w += y2.w;
        }
        float* dqr = dq + (long)n * lddq + lane * 4;
        atomicAdd(dqr + 0, x.x); atomicAdd(dqr + 1, x.y); atomicAdd(dqr + 2, x.z); atomicAdd(dqr + 3, x.w);
        float4* dvp = reinterpret_cast<float4*>(dvpart + ((long)n * gridDim.y + blockIdx.y) * 256 + lane * 4);
        float4 o = *dvp;
        o.x += y.x; o.y += y.y; o.z += y.z; o.w += y.w;
        *dvp = o;
    }
}

// teacher-forcing frames (helpers.py:49,76,80-82): F[n,s,:] = 0 for s == 0 else mel[n, r*s-1, :]
__global__ void gather_frames_k(const float* __restrict__ mel, float* __restrict__ F, int N, int S, int r, int nm) {
    const long total = (long)N * S * nm;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nm);
        const long ns = idx / nm;
        const int s = (int)(ns % S);
        const long n = ns / S;
        F[idx] = s == 0 ? 0.0f : mel[((n * S * r) + (long)r * s - 1) * nm + c];
    }
}

extern "C" int taco_gather_frames(const float* mel, float* frames, int N, int S, int r, int num_mels, hipStream_t stream) {
    const long total = (long)N * S * num_mels;
    hipLaunchKernelGGL(gather_frames_k, dim3((int)min((total + 255) / 256, 2048L)), dim3(256), 0, stream, mel, frames, N, S, r, num_mels);
    TACO_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// host-side step loops
// ---------------------------------------------------------------------------------------------------
static int attn_rnn_fwd_steps(const void* const* ptrs, const int* dims, hipStream_t st) {
    if (!ptrs || !dims) return TACO_EINVAL;
    const int N = dims[0], S = dims[1], Ti = dims[2];
    if (N <= 0 || S <= 0 || Ti <= 0) return TACO_EINVAL;
    auto F = [&](int i) { return (const float*)ptrs[i]; };
    auto G = [&](int i) { return (float*)const_cast<void*>(ptrs[i]); };
    const float* zeros = F(TACO_AP_ZEROS);
    const size_t smem = (((Ti + 3) & ~3) + 256) * sizeof(float);
    for (int s = 0; s < S; ++s) {
        const bool first = s == 0;
        float* hc_s = G(TACO_AP_HC) + (long)s * 512;                      // [N] rows, ld S*512: h at +0, ctx at +256
        const float* hprev = first ? zeros : hc_s - 512;
        const float* ctxprev = first ? zeros : hc_s - 512 + 256;
        const int ldprev = first ? 256 : S * 512;
        Skinny k{};
        // prenet dense_1: relu(F1[s] + ctx_{s-1} . W1[80:336])
        k = Skinny{}; k.A0 = ctxprev; k.lda0 = ldprev; k.K0 = 256; k.B0 = F(TACO_AP_W1C); k.ldb0 = 256;
        k.M = N; k.N = 256; k.mode = M_PRENET1; k.p[0] = F(TACO_AP_F1) + (long)s * 256; k.ld[0] = S * 256;
        k.o[0] = G(TACO_AP_P1) + (long)s * 256; k.ldo[0] = S * 256;
        launch_skinny(k, st);
        // prenet dense_2
        k = Skinny{}; k.A0 = F(TACO_AP_P1) + (long)s * 256; k.lda0 = S * 256; k.K0 = 256; k.B0 = F(TACO_AP_W2); k.ldb0 = 128;
        k.M = N; k.N = 128; k.mode = M_BIAS_RELU; k.p[0] = F(TACO_AP_B2);
        k.o[0] = G(TACO_AP_P2) + (long)s * 128; k.ldo[0] = S * 128;
        launch_skinny(k, st);
        // attention GRU gates
        k = Skinny{}; k.A0 = F(TACO_AP_P2) + (long)s * 128; k.lda0 = S * 128; k.K0 = 128; k.B0 = F(TACO_AP_WX); k.ldb0 = 768;
        k.A1 = hprev; k.lda1 = ldprev; k.K1 = 256; k.B1 = F(TACO_AP_WHG); k.ldb1 = 512;
        k.M = N; k.N = 512; k.mode = M_GRU_GATES; k.Hd = 256; k.p[1] = F(TACO_AP_BG); k.p[2] = hprev; k.ld[2] = ldprev;
        k.o[0] = G(TACO_AP_R) + (long)s * 256; k.ldo[0] = S * 256; k.o[1] = G(TACO_AP_U) + (long)s * 256; k.ldo[1] = S * 256;
        k.o[2] = G(TACO_AP_RH) + (long)s * 256; k.ldo[2] = S * 256;
        launch_skinny(k, st);
        // candidate + state
        k = Skinny{}; k.A0 = F(TACO_AP_P2) + (long)s * 128; k.lda0 = S * 128; k.K0 = 128; k.B0 = F(TACO_AP_WX) + 512; k.ldb0 = 768;
        k.A1 = F(TACO_AP_RH) + (long)s * 256; k.lda1 = S * 256; k.K1 = 256; k.B1 = F(TACO_AP_WHC); k.ldb1 = 256;
        k.M = N; k.N = 256; k.mode = M_GRU_CAND; k.Hd = 256; k.p[1] = F(TACO_AP_BG) + 512;
        k.p[2] = hprev; k.ld[2] = ldprev; k.p[3] = F(TACO_AP_U) + (long)s * 256; k.ld[3] = S * 256;
        k.o[0] = G(TACO_AP_C) + (long)s * 256; k.ldo[0] = S * 256; k.o[1] = hc_s; k.ldo[1] = S * 512;
        launch_skinny(k, st);
        // query
        k = Skinny{}; k.A0 = hc_s; k.lda0 = S * 512; k.K0 = 256; k.B0 = F(TACO_AP_WQ); k.ldb0 = 256;
        k.M = N; k.N = 256; k.mode = M_LINEAR; k.o[0] = G(TACO_AP_Q) + (long)s * 256; k.ldo[0] = S * 256;
        launch_skinny(k, st);
        // scores, softmax, context
        float* al = G(TACO_AP_ALIGN) + (long)s * Ti;
        hipLaunchKernelGGL(attn_scores_k, dim3(N, cdiv(Ti, 16)), dim3(256), 0, st, F(TACO_AP_KEYS), F(TACO_AP_Q) + (long)s * 256, S * 256,
                           F(TACO_AP_V), al, S * Ti, Ti);
        hipLaunchKernelGGL(attn_softmax_ctx_k, dim3(N, 4), dim3(256), smem, st, al, S * Ti, F(TACO_AP_MEM), hc_s + 256, S * 512, Ti);
    }
    TACO_RETURN_LAST();
}

// BPTT of the attention recurrence.  dHC [N,S,512] holds the external gradients wrt (h_s, ctx_s) from the
// concat projection.  Produces dXP (attention GRU pre-activations) [N,S,768], dP2/dP1 (prenet pre-activations),
// dQ [N,S,256] (pre-zeroed), dKEYS / dMEM [N,Ti,256] (pre-zeroed accumulators), dVPART [N,ceil(Ti/16),256] (pre-zeroed).
static int attn_rnn_bwd_steps(const void* const* ptrs, const int* dims, hipStream_t st) {
    if (!ptrs || !dims) return TACO_EINVAL;
    const int N = dims[0], S = dims[1], Ti = dims[2];
    if (N <= 0 || S <= 0 || Ti <= 0) return TACO_EINVAL;
    auto F = [&](int i) { return (const float*)ptrs[i]; };
    auto G = [&](int i) { return (float*)const_cast<void*>(ptrs[i]); };
    const float* zeros = F(TACO_AP_ZEROS);
    float* dhcarry = G(TACO_AP_DHCARRY);      // [N,256] gradient wrt h_s arriving from step s+1 (zero at s = S-1)
    float* dctxcarry = G(TACO_AP_DCTXCARRY);  // [N,256] gradient wrt ctx_s arriving from step s+1's prenet
    if (hipMemsetAsync(dhcarry, 0, (size_t)N * 256 * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(dctxcarry, 0, (size_t)N * 256 * sizeof(float), st) != hipSuccess) return TACO_EINVAL;
    const dim3 gT(N, cdiv(Ti, 16));
    for (int s = S - 1; s >= 0; --s) {
        const bool first = s == 0;
        const float* hc_s = F(TACO_AP_HC) + (long)s * 512;
        const float* hprev = first ? zeros : hc_s - 512;
        const int ldprev = first ? 256 : S * 512;
        const float* al = F(TACO_AP_ALIGN) + (long)s * Ti;
        Skinny k{};
        // dctx = dctx_ext[s] + carry
        k = Skinny{}; k.M = N; k.N = 256; k.mode = M_LINEAR;
        k.p[1] = F(TACO_AP_DHC) + (long)s * 512 + 256; k.ld[1] = S * 512; k.p[2] = dctxcarry; k.ld[2] = 256;
        k.o[0] = G(TACO_AP_DCTX); k.ldo[0] = 256;
        launch_skinny(k, st);
        // attention backward
        hipLaunchKernelGGL(attn_bwd_da_k, gT, dim3(256), 0, st, F(TACO_AP_MEM), al, S * Ti, F(TACO_AP_DCTX), 256, G(TACO_AP_DA), G(TACO_AP_DMEM), Ti,
                           ptrs[TACO_AP_DAEXT] ? F(TACO_AP_DAEXT) + (long)s * Ti : nullptr, S * Ti);
        hipLaunchKernelGGL(attn_bwd_score_k, gT, dim3(256), 0, st, F(TACO_AP_KEYS), F(TACO_AP_Q) + (long)s * 256, S * 256, F(TACO_AP_V), al, S * Ti,
                           F(TACO_AP_DA), G(TACO_AP_DKEYS), G(TACO_AP_DQ) + (long)s * 256, S * 256, G(TACO_AP_DVPART), Ti);
        // dhT = dh_ext[s] + carry + dq . Wq^T
        k = Skinny{}; k.A0 = F(TACO_AP_DQ) + (long)s * 256; k.lda0 = S * 256; k.K0 = 256; k.B0 = F(TACO_AP_WQ); k.ldb0 = 256; k.bt = 1;
        k.M = N; k.N = 256; k.mode = M_LINEAR; k.p[1] = F(TACO_AP_DHC) + (long)s * 512; k.ld[1] = S * 512; k.p[2] = dhcarry; k.ld[2] = 256;
        k.o[0] = G(TACO_AP_DHT); k.ldo[0] = 256;
        launch_skinny(k, st);
        // GRU backward 1: drh = dcp . Whc^T, gate gradients
        k = Skinny{}; k.A0 = F(TACO_AP_DHT); k.lda0 = 256; k.K0 = 256; k.B0 = F(TACO_AP_WHC); k.ldb0 = 256; k.bt = 1; k.amode = 1;
        k.M = N; k.N = 256; k.mode = M_GRU_BWD1; k.Hd = 256;
        k.p[0] = F(TACO_AP_DHT); k.ld[0] = 256; k.p[1] = F(TACO_AP_R) + (long)s * 256; k.ld[1] = S * 256;
        k.p[2] = F(TACO_AP_U) + (long)s * 256; k.ld[2] = S * 256; k.p[3] = F(TACO_AP_C) + (long)s * 256; k.ld[3] = S * 256;
        k.p[4] = hprev; k.ld[4] = ldprev;
        k.o[0] = G(TACO_AP_DXP) + (long)s * 768; k.ldo[0] = S * 768; k.o[1] = G(TACO_AP_DHPART); k.ldo[1] = 256;
        launch_skinny(k, st);
        // GRU backward 2: dh_{s-1} carry = dhpart + dg . Whg^T
        k = Skinny{}; k.A0 = F(TACO_AP_DXP) + (long)s * 768; k.lda0 = S * 768; k.K0 = 512; k.B0 = F(TACO_AP_WHG); k.ldb0 = 512; k.bt = 1;
        k.M = N; k.N = 256; k.mode = M_GRU_BWD2; k.p[0] = F(TACO_AP_DHPART); k.ld[0] = 256; k.o[0] = dhcarry; k.ldo[0] = 256;
        launch_skinny(k, st);
        // dp2pre = (dxp . Wx^T) * (p2 > 0)
        k = Skinny{}; k.A0 = F(TACO_AP_DXP) + (long)s * 768; k.lda0 = S * 768; k.K0 = 768; k.B0 = F(TACO_AP_WX); k.ldb0 = 768; k.bt = 1;
        k.M = N; k.N = 128; k.mode = M_RELU_MASK; k.p[0] = F(TACO_AP_P2) + (long)s * 128; k.ld[0] = S * 128;
        k.o[0] = G(TACO_AP_DP2) + (long)s * 128; k.ldo[0] = S * 128;
        launch_skinny(k, st);
        // dp1pre = (dp2pre . W2^T) * (p1 > 0)
        k = Skinny{}; k.A0 = F(TACO_AP_DP2) + (long)s * 128; k.lda0 = S * 128; k.K0 = 128; k.B0 = F(TACO_AP_W2); k.ldb0 = 128; k.bt = 1;
        k.M = N; k.N = 256; k.mode = M_RELU_MASK; k.p[0] = F(TACO_AP_P1) + (long)s * 256; k.ld[0] = S * 256;
        k.o[0] = G(TACO_AP_DP1) + (long)s * 256; k.ldo[0] = S * 256;
        launch_skinny(k, st);
        // dctx carry (gradient wrt ctx_{s-1}) = dp1pre . W1c^T
        if (!first) {
            k = Skinny{}; k.A0 = F(TACO_AP_DP1) + (long)s * 256; k.lda0 = S * 256; k.K0 = 256; k.B0 = F(TACO_AP_W1C); k.ldb0 = 256; k.bt = 1;
            k.M = N; k.N = 256; k.mode = M_LINEAR; k.o[0] = dctxcarry; k.ldo[0] = 256;
            launch_skinny(k, st);
        }
    }
    TACO_RETURN_LAST();
}

extern "C" int taco_attn_cluster_supported(int N, int Ti);

// Attention recurrence, forward: one persistent cluster launch (attn_cluster.hip) when the shape fits
// (N <= 64, T_in slices fit LDS) and scratch was provided, else one launch per dependent stage.
extern "C" int taco_attn_rnn_fwd(const void* const* ptrs, const int* dims, hipStream_t st) {
    if (!ptrs || !dims) return TACO_EINVAL;
    const int N = dims[0], S = dims[1], Ti = dims[2];
    if (N <= 0 || S <= 0 || Ti <= 0) return TACO_EINVAL;
    if (ptrs[TACO_AP_XCHG] && ptrs[TACO_AP_ERR] && taco_attn_cluster_supported(N, Ti)) {
        auto F = [&](int i) { return (const float*)ptrs[i]; };
        auto G = [&](int i) { return (float*)const_cast<void*>(ptrs[i]); };
        AttnClu p{};
        p.w1c = F(TACO_AP_W1C); p.f1 = F(TACO_AP_F1); p.w2 = F(TACO_AP_W2); p.b2 = F(TACO_AP_B2); p.wx = F(TACO_AP_WX);
        p.whg = F(TACO_AP_WHG); p.whc = F(TACO_AP_WHC); p.bg = F(TACO_AP_BG); p.wq = F(TACO_AP_WQ); p.v = F(TACO_AP_V);
        p.keys = F(TACO_AP_KEYS); p.mem = F(TACO_AP_MEM);
        p.p1 = G(TACO_AP_P1); p.p2 = G(TACO_AP_P2); p.r = G(TACO_AP_R); p.u = G(TACO_AP_U); p.c = G(TACO_AP_C);
        p.rh = G(TACO_AP_RH); p.hc = G(TACO_AP_HC); p.q = G(TACO_AP_Q); p.align = G(TACO_AP_ALIGN);
        p.xchg = (u64*)const_cast<void*>(ptrs[TACO_AP_XCHG]); p.err = (int*)const_cast<void*>(ptrs[TACO_AP_ERR]);
        p.N = N; p.S = S; p.Ti = Ti; p.s0 = dims[3]; p.s1 = dims[4];
        if (p.s0 < 0 || p.s1 > S || p.s0 >= p.s1) return TACO_EINVAL;
        return attn_cluster_fwd_launch(p, st);
    }
    if (dims[3] != 0 || dims[4] != S) return TACO_EINVAL;      // the per-stage fallback runs whole sequences only
    return attn_rnn_fwd_steps(ptrs, dims, st);
}

// Attention recurrence, backward.  Cluster path: also needs TACO_AP_DE [N,S,Ti], TACO_AP_DCTXS [N,S,256] and a dVPART of
// [N*Ti,256]; dKEYS / dMEM / dVPART are then WRITTEN (not accumulated) by the hoisted reduction kernel.
extern "C" int taco_attn_rnn_bwd(const void* const* ptrs, const int* dims, hipStream_t st) {
    return taco_attn_rnn_bwd_chunk(ptrs, dims, 0, nullptr, st);
}

// the reduction over the steps on its own (after a TACO_ATTN_NO_REDUCE launch, once the streams of all chunks are joined)
extern "C" int taco_attn_bwd_reduce(const void* const* ptrs, const int* dims, hipStream_t st) {
    if (!ptrs || !dims) return TACO_EINVAL;
    const int N = dims[0], S = dims[1], Ti = dims[2];
    if (N <= 0 || S <= 0 || Ti <= 0 || !ptrs[TACO_AP_DE] || !ptrs[TACO_AP_DCTXS] || !taco_attn_cluster_supported(N, Ti)) return TACO_EINVAL;
    AttnCluB p{};
    p.keys = (const float*)ptrs[TACO_AP_KEYS]; p.q = (const float*)ptrs[TACO_AP_Q]; p.align = (const float*)ptrs[TACO_AP_ALIGN];
    p.v = (const float*)ptrs[TACO_AP_V];
    p.de = (float*)const_cast<void*>(ptrs[TACO_AP_DE]); p.dctx = (float*)const_cast<void*>(ptrs[TACO_AP_DCTXS]);
    p.N = N; p.S = S; p.Ti = Ti;
    return attn_cluster_bwd_reduce(p, (float*)const_cast<void*>(ptrs[TACO_AP_DKEYS]), (float*)const_cast<void*>(ptrs[TACO_AP_DMEM]),
                                   (float*)const_cast<void*>(ptrs[TACO_AP_DVPART]), st);
}

extern "C" int taco_attn_rnn_bwd_chunk(const void* const* ptrs, const int* dims, int carry_flags, void* carry_xchg, hipStream_t st) {
    if (!ptrs || !dims || (carry_flags & ~7)) return TACO_EINVAL;
    const int N = dims[0], S = dims[1], Ti = dims[2];
    if (N <= 0 || S <= 0 || Ti <= 0) return TACO_EINVAL;
    if (ptrs[TACO_AP_XCHG] && ptrs[TACO_AP_ERR] && ptrs[TACO_AP_DE] && ptrs[TACO_AP_DCTXS] && taco_attn_cluster_supported(N, Ti)) {
        auto F = [&](int i) { return (const float*)ptrs[i]; };
        auto G = [&](int i) { return (float*)const_cast<void*>(ptrs[i]); };
        AttnCluB p{};
        p.w1c = F(TACO_AP_W1C); p.w2 = F(TACO_AP_W2); p.wx = F(TACO_AP_WX); p.whg = F(TACO_AP_WHG); p.whc = F(TACO_AP_WHC);
        p.wq = F(TACO_AP_WQ); p.v = F(TACO_AP_V); p.keys = F(TACO_AP_KEYS); p.mem = F(TACO_AP_MEM);
        p.p1 = F(TACO_AP_P1); p.p2 = F(TACO_AP_P2); p.r = F(TACO_AP_R); p.u = F(TACO_AP_U); p.c = F(TACO_AP_C);
        p.hc = F(TACO_AP_HC); p.q = F(TACO_AP_Q); p.align = F(TACO_AP_ALIGN); p.dhc = F(TACO_AP_DHC);
        p.dxp = G(TACO_AP_DXP); p.dp2 = G(TACO_AP_DP2); p.dp1 = G(TACO_AP_DP1); p.dq = G(TACO_AP_DQ);
        p.de = G(TACO_AP_DE); p.dctx = G(TACO_AP_DCTXS);
        p.da_ext = ptrs[TACO_AP_DAEXT] ? F(TACO_AP_DAEXT) : nullptr;
        p.xchg = (u64*)const_cast<void*>(ptrs[TACO_AP_XCHG]); p.err = (int*)const_cast<void*>(ptrs[TACO_AP_ERR]);
        p.dhcarry = G(TACO_AP_DHCARRY); p.dctxcarry = G(TACO_AP_DCTXCARRY);
        p.N = N; p.S = S; p.Ti = Ti; p.s0 = dims[3]; p.s1 = dims[4]; p.carry_flags = carry_flags; p.carry_xchg = (u64*)carry_xchg;
        if (p.s0 < 0 || p.s1 > S || p.s0 >= p.s1) return TACO_EINVAL;
        if ((carry_flags & TACO_ATTN_CARRY_WAIT) && (p.s1 == S || !carry_xchg || carry_xchg == (void*)p.xchg)) return TACO_EINVAL;   // no carries for
                                                                                           // the first launch of a pass; own exchange buffer required
        if ((carry_flags & TACO_ATTN_CARRY_POST) && p.s0 == 0) return TACO_EINVAL;      // the last one hands none over
        return attn_cluster_bwd_launch(p, G(TACO_AP_DKEYS), G(TACO_AP_DMEM), G(TACO_AP_DVPART), st);
    }
    if (carry_flags) return TACO_EINVAL;                 // the per-step kernels run a pass in one call, reduction included
    if (dims[3] != 0 || dims[4] != S) return TACO_EINVAL;
    return attn_rnn_bwd_steps(ptrs, dims, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Free-running inference decoder (reference models/helpers.py:7-38 TacoTestHelper + tacotron.py:86-94): the last of the r
// predicted frames is fed back, so nothing can be hoisted; one launch per dependent stage (13 per step).
// Stop condition (helpers.py:32-38 + dynamic_decode): a row is finished once a whole r-frame output of it is EXACTLY zero
// (sticky), decoding ends after the first step at which every row is finished, or at max_iters.  infer_stop_k keeps the
// per-row flags and the step count on the device (TACO_IP_STOP); the host enqueues the steps [s0, s1) of a launch
// unconditionally and reads the count between launches (Engine.infer), so nothing here synchronises.
// ptrs indexed by enum TacoInferPtr, dims = {N, S, Ti, r, num_mels, s0, s1}.
// ---------------------------------------------------------------------------------------------------------------------
// stop[0] = number of decoder steps to keep (initialised to S by the caller), stop[1 + n] = row n finished
__global__ void infer_stop_k(const float* __restrict__ out, long ld, int no, int N, int s, int S, int* __restrict__ stop) {
    __shared__ int all_done;
    if (threadIdx.x == 0) all_done = 1;
    __syncthreads();
    for (int n = threadIdx.x >> 6; n < N; n += blockDim.x >> 6) {          // one wave per row
        bool zero = true;
        for (int j = threadIdx.x & 63; j < no; j += 64) zero = zero && (out[n * ld + j] == 0.0f);
        const bool fin = (__all(zero) != 0) || stop[1 + n] != 0;
        if ((threadIdx.x & 63) == 0) { stop[1 + n] = fin ? 1 : 0; if (!fin) all_done = 0; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && all_done && stop[0] == S) stop[0] = s + 1;
}

extern "C" int taco_decoder_infer(const void* const* ptrs, const int* dims, hipStream_t st) {
    if (!ptrs || !dims) return TACO_EINVAL;
    const int N = dims[0], S = dims[1], Ti = dims[2], r = dims[3], nm = dims[4], s_lo = dims[5], s_hi = dims[6];
    if (N <= 0 || S <= 0 || Ti <= 0 || r <= 0 || nm <= 0 || (nm & 3) || s_lo < 0 || s_hi > S || s_lo >= s_hi) return TACO_EINVAL;
    for (int i = 0; i < TACO_IP_COUNT; ++i) if (!ptrs[i]) return TACO_EINVAL;
    auto F = [&](int i) { return (const float*)ptrs[i]; };
    auto G = [&](int i) { return (float*)const_cast<void*>(ptrs[i]); };
    const float* zeros = F(TACO_IP_ZEROS);
    const int no = nm * r;
    const size_t smem = (((Ti + 3) & ~3) + 256) * sizeof(float);
    auto gru = [&](const float* x, int ldx, int wx, int b, int whg, int whc, const float* hprev, int ldprev, float* R, float* U,
                   float* C, float* RH, float* Hn, int ldh, float* D) {
        Skinny k{};
        k.A0 = x; k.lda0 = ldx; k.K0 = 256; k.B0 = F(wx); k.ldb0 = 768; k.A1 = hprev; k.lda1 = ldprev; k.K1 = 256; k.B1 = F(whg); k.ldb1 = 512;
        k.M = N; k.N = 512; k.mode = M_GRU_GATES; k.Hd = 256; k.p[1] = F(b); k.p[2] = hprev; k.ld[2] = ldprev;
        k.o[0] = R; k.ldo[0] = 256; k.o[1] = U; k.ldo[1] = 256; k.o[2] = RH; k.ldo[2] = 256;
        launch_skinny(k, st);
        k = Skinny{};
        k.A0 = x; k.lda0 = ldx; k.K0 = 256; k.B0 = F(wx) + 512; k.ldb0 = 768; k.A1 = RH; k.lda1 = 256; k.K1 = 256; k.B1 = F(whc); k.ldb1 = 256;
        k.M = N; k.N = 256; k.mode = M_GRU_CAND; k.Hd = 256; k.p[1] = F(b) + 512; k.p[2] = hprev; k.ld[2] = ldprev; k.p[3] = U; k.ld[3] = 256;
        k.p[4] = x; k.ld[4] = ldx;
        k.o[0] = C; k.ldo[0] = 256; k.o[1] = Hn; k.ldo[1] = ldh; k.o[2] = D; k.ldo[2] = 256;
        launch_skinny(k, st);
    };
    float* H1 = G(TACO_IP_H1);            // [2][N,256] ping-pong states
    float* H2 = G(TACO_IP_H2);
    float* T = G(TACO_IP_TMP);            // [10][N,256] step scratch: p1, p2(128), r, u, c, rh, q, y, d1, d2
    float *p1 = T, *p2 = T + (long)N * 256, *R = T + 2L * N * 256, *U = T + 3L * N * 256, *C = T + 4L * N * 256,
          *RH = T + 5L * N * 256, *Q = T + 6L * N * 256, *Y = T + 7L * N * 256, *D1 = T + 8L * N * 256, *D2 = T + 9L * N * 256;
    for (int s = s_lo; s < s_hi; ++s) {
        const bool first = s == 0;
        float* hc_s = G(TACO_IP_HC) + (long)s * 512;                 // [N] rows, ld S*512
        const float* hprev = first ? zeros : hc_s - 512;
        const float* ctxprev = first ? zeros : hc_s - 512 + 256;
        const int ldprev = first ? 256 : S * 512;
        const float* frame = first ? zeros : F(TACO_IP_OUT) + (long)(s - 1) * no + (long)(r - 1) * nm;
        const int ldf = first ? 256 : S * no;
        Skinny k{};
        k.A0 = frame; k.lda0 = ldf; k.K0 = nm; k.B0 = F(TACO_IP_W1); k.ldb0 = 256;
        k.A1 = ctxprev; k.lda1 = ldprev; k.K1 = 256; k.B1 = F(TACO_IP_W1) + (long)nm * 256; k.ldb1 = 256;
        k.M = N; k.N = 256; k.mode = M_BIAS_RELU; k.p[0] = F(TACO_IP_B1); k.o[0] = p1; k.ldo[0] = 256;
        launch_skinny(k, st);
        k = Skinny{}; k.A0 = p1; k.lda0 = 256; k.K0 = 256; k.B0 = F(TACO_IP_W2); k.ldb0 = 128;
        k.M = N; k.N = 128; k.mode = M_BIAS_RELU; k.p[0] = F(TACO_IP_B2); k.o[0] = p2; k.ldo[0] = 128;
        launch_skinny(k, st);
        k = Skinny{}; k.A0 = p2; k.lda0 = 128; k.K0 = 128; k.B0 = F(TACO_IP_WX); k.ldb0 = 768;
        k.A1 = hprev; k.lda1 = ldprev; k.K1 = 256; k.B1 = F(TACO_IP_WHG); k.ldb1 = 512;
        k.M = N; k.N = 512; k.mode = M_GRU_GATES; k.Hd = 256; k.p[1] = F(TACO_IP_BG); k.p[2] = hprev; k.ld[2] = ldprev;
        k.o[0] = R; k.ldo[0] = 256; k.o[1] = U; k.ldo[1] = 256; k.o[2] = RH; k.ldo[2] = 256;
        launch_skinny(k, st);
        k = Skinny{}; k.A0 = p2; k.lda0 = 128; k.K0 = 128; k.B0 = F(TACO_IP_WX) + 512; k.ldb0 = 768;
        k.A1 = RH; k.lda1 = 256; k.K1 = 256; k.B1 = F(TACO_IP_WHC); k.ldb1 = 256;
        k.M = N; k.N = 256; k.mode = M_GRU_CAND; k.Hd = 256; k.p[1] = F(TACO_IP_BG) + 512;
        k.p[2] = hprev; k.ld[2] = ldprev; k.p[3] = U; k.ld[3] = 256;
        k.o[0] = C; k.ldo[0] = 256; k.o[1] = hc_s; k.ldo[1] = S * 512;
        launch_skinny(k, st);
        k = Skinny{}; k.A0 = hc_s; k.lda0 = S * 512; k.K0 = 256; k.B0 = F(TACO_IP_WQ); k.ldb0 = 256;
        k.M = N; k.N = 256; k.mode = M_LINEAR; k.o[0] = Q; k.ldo[0] = 256;
        launch_skinny(k, st);
        float* al = G(TACO_IP_ALIGN) + (long)s * Ti;
        hipLaunchKernelGGL(attn_scores_k, dim3(N, cdiv(Ti, 16)), dim3(256), 0, st, F(TACO_IP_KEYS), Q, 256, F(TACO_IP_V), al, S * Ti, Ti);
        hipLaunchKernelGGL(attn_softmax_ctx_k, dim3(N, 4), dim3(256), smem, st, al, S * Ti, F(TACO_IP_MEM), hc_s + 256, S * 512, Ti);
        k = Skinny{}; k.A0 = hc_s; k.lda0 = S * 512; k.K0 = 512; k.B0 = F(TACO_IP_WP); k.ldb0 = 256;
        k.M = N; k.N = 256; k.mode = M_LINEAR; k.p[0] = F(TACO_IP_BP); k.o[0] = Y; k.ldo[0] = 256;
        launch_skinny(k, st);
        float* h1n = H1 + (long)(s & 1) * N * 256;
        const float* h1p = first ? zeros : H1 + (long)((s - 1) & 1) * N * 256;
        gru(Y, 256, TACO_IP_G1WX, TACO_IP_G1B, TACO_IP_G1WHG, TACO_IP_G1WHC, h1p, 256, R, U, C, RH, h1n, 256, D1);
        float* h2n = H2 + (long)(s & 1) * N * 256;
        const float* h2p = first ? zeros : H2 + (long)((s - 1) & 1) * N * 256;
        gru(D1, 256, TACO_IP_G2WX, TACO_IP_G2B, TACO_IP_G2WHG, TACO_IP_G2WHC, h2p, 256, R, U, C, RH, h2n, 256, D2);
        k = Skinny{}; k.A0 = D2; k.lda0 = 256; k.K0 = 256; k.B0 = F(TACO_IP_WO); k.ldb0 = no;
        k.M = N; k.N = no; k.mode = M_LINEAR; k.p[0] = F(TACO_IP_BO); k.o[0] = G(TACO_IP_OUT) + (long)s * no; k.ldo[0] = S * no;
        launch_skinny(k, st);
        hipLaunchKernelGGL(infer_stop_k, dim3(1), dim3(256), 0, st, F(TACO_IP_OUT) + (long)s * no, (long)S * no, no, N, s, S,
                           (int*)const_cast<void*>(ptrs[TACO_IP_STOP]));
    }
    TACO_RETURN_LAST();
}
