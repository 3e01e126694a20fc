// HBM-bound streaming kernels of the Tacotron training step (gfx950): embedding gather/scatter,
// batch-norm statistics / apply (+ fused maxpool, residual) forward and backward, highway gating,
// L1 losses with in-place sign gradients, column sums (bias gradients).
// All tensors are channel-last row-major fp32; channel counts are multiples of 4 so every thread moves
// 16 B per access (global_load_dwordx4), rows are covered by consecutive lanes -> fully coalesced.
#include "common.hpp"

// =====================================================================================================
// Embedding  (reference models/tacotron.py:42-55: embedding_lookup + speaker lookup/tile/concat)
// =====================================================================================================
__global__ void embed_gather_k(const int* __restrict__ ids, const int* __restrict__ identities,
                               const float* __restrict__ table, const float* __restrict__ spk, float* __restrict__ out,
                               int N, int Ti, int Et, int Es, int vocab, int id_num) {
    const int E = Et + Es, e4 = E / 4;
    const long total = (long)N * Ti * e4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % e4) * 4;
        const long row = idx / e4;
        float4 v;
        if (c < Et) {
            int id = ids[row];
            id = min(max(id, 0), vocab - 1);
            v = *reinterpret_cast<const float4*>(table + (long)id * Et + c);
        } else {
            int sid = identities[row / Ti];
            sid = min(max(sid, 0), id_num - 1);
            v = *reinterpret_cast<const float4*>(spk + (long)sid * Es + (c - Et));
        }
        *reinterpret_cast<float4*>(out + row * E + c) = v;
    }
}

// dTable[ids[row]] += dE[row, :Et]; also accumulates sum of squares of the un-deduplicated rows
// (tf.global_norm over IndexedSlices.values) into sumsq[0].
__global__ __launch_bounds__(256) void embed_scatter_text_k(const int* __restrict__ ids, const float* __restrict__ dE, float* __restrict__ dTable,
                                                           double* __restrict__ sumsq, int rows, int Et, int E, int vocab) {
    // one float per lane: a wave's atomic instruction covers 256 contiguous bytes of one table row (Et % 64 == 0) or of two;
    // the sum of squares goes through one atomic per WORKGROUP (same-address atomics serialise in L2)
    const long total = (long)rows * Et;
    float ss = 0.0f;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / Et;
        const int c = (int)(idx - row * Et);
        const float g = dE[row * E + c];
        const int id = min(max(ids[row], 0), vocab - 1);
        atomicAdd(dTable + (long)id * Et + c, g);
        ss = fmaf(g, g, ss);
    }
    const double w = wave_sum_d((double)ss);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0 && sumsq) atomicAdd(sumsq, red[0] + red[1] + red[2] + red[3]);
}

// dSpk[identities[n]] += sum_t dE[n,t,Et:]; one block per n; sumsq[0] += |sum_t ...|^2
__global__ void embed_scatter_spk_k(const int* __restrict__ identities, const float* __restrict__ dE,
                                    float* __restrict__ dSpk, double* __restrict__ sumsq, int Ti, int Et, int Es, int id_num) {
    const int n = blockIdx.x, E = Et + Es;
    __shared__ float part[256];
    const int c = threadIdx.x % Es, lane_t = threadIdx.x / Es, nt = blockDim.x / Es;
    float s = 0.0f;
    if (lane_t < nt)
        for (int t = lane_t; t < Ti; t += nt) s += dE[((long)n * Ti + t) * E + Et + c];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < Es) {
        float tot = 0.0f;
        for (int k = 0; k < nt; ++k) tot += part[k * Es + threadIdx.x];
        int sid = min(max(identities[n], 0), id_num - 1);
        atomicAdd(dSpk + (long)sid * Es + threadIdx.x, tot);
        if (sumsq) atomicAdd(sumsq, (double)tot * (double)tot);
    }
}

// =====================================================================================================
// Column reductions over [M, C] (C % 4 == 0): thread = one float4 column group, 4 row lanes per block
// =====================================================================================================
// MODE 0: sum(x)                     -> out0           (bias gradients)          [float atomics]
// MODE 1: sum(x), sum(x^2)           -> dstat[0:C], dstat[C:2C]   (BN statistics) [double atomics]
// MODE 2: sum(db), sum(db*xhat)      -> dstat             (BN backward; db from dY or through maxpool)
struct ColRed {
    const float* x; int ldx;
    const float* dy; int lddy;          // MODE 2: gradient wrt BN output (or wrt pooled output when pool)
    const float* mean; const float* rstd; const float* scale; const float* shift;
    float* out0; double* dstat;
    int M, C, T, pool, rows_per_block;
    int seq_len, seq_t0, seq_rows;      // MODE 1 over the frames [seq_t0, seq_t0 + seq_rows) of every length-seq_len sequence (blockIdx.z = sequence); 0 = all M rows
};

__device__ __forceinline__ float4 f4fma(float4 a, float4 s, float4 b) {
    return make_float4(fmaf(a.x, s.x, b.x), fmaf(a.y, s.y, b.y), fmaf(a.z, s.z, b.z), fmaf(a.w, s.w, b.w));
}

// gradient wrt the BN output at row m for 4 channels; with pool: routed through max(b[t], b[t+1])
// (first max wins ties, like TF CPU MaxPoolGrad / torch max_pool1d)
__device__ __forceinline__ float4 bn_out_grad(const ColRed& p, long m, int c, float4 xv, float4 sc, float4 sh) {
    const float4 g = *reinterpret_cast<const float4*>(p.dy + m * p.lddy + c);
    if (!p.pool) return g;
    const int t = (int)(m % p.T);
    const float4 b = f4fma(xv, sc, sh);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t == p.T - 1) r = g;
    else {
        const float4 bn = f4fma(*reinterpret_cast<const float4*>(p.x + (m + 1) * p.ldx + c), sc, sh);
        r.x = b.x >= bn.x ? g.x : 0.f; r.y = b.y >= bn.y ? g.y : 0.f; r.z = b.z >= bn.z ? g.z : 0.f; r.w = b.w >= bn.w ? g.w : 0.f;
    }
    if (t > 0) {
        const float4 bp = f4fma(*reinterpret_cast<const float4*>(p.x + (m - 1) * p.ldx + c), sc, sh);
        const float4 gp = *reinterpret_cast<const float4*>(p.dy + (m - 1) * p.lddy + c);
        r.x += b.x > bp.x ? gp.x : 0.f; r.y += b.y > bp.y ? gp.y : 0.f; r.z += b.z > bp.z ? gp.z : 0.f; r.w += b.w > bp.w ? gp.w : 0.f;
    }
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_k(ColRed p) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + tx) * 4;
    const bool active = c < p.C;
    const long base = p.seq_rows ? (long)blockIdx.z * p.seq_len + p.seq_t0 : 0;
    const long r0 = base + (long)blockIdx.y * p.rows_per_block;
    const long r1 = min(base + (long)(p.seq_rows ? p.seq_rows : p.M), r0 + p.rows_per_block);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    float4 sc = s0, sh = s0, mu = s0, rs = s0;
    if (active && MODE == 2) {
        mu = *reinterpret_cast<const float4*>(p.mean + c);
        rs = *reinterpret_cast<const float4*>(p.rstd + c);
        sc = *reinterpret_cast<const float4*>(p.scale + c);
        sh = *reinterpret_cast<const float4*>(p.shift + c);
    }
    if (active) {
        auto body = [&](long m) {
            const float4 xv = *reinterpret_cast<const float4*>(p.x + m * p.ldx + c);
            if (MODE == 0) { s0.x += xv.x; s0.y += xv.y; s0.z += xv.z; s0.w += xv.w; }
            else if (MODE == 1) {
                s0.x += xv.x; s0.y += xv.y; s0.z += xv.z; s0.w += xv.w;
                s1.x += xv.x * xv.x; s1.y += xv.y * xv.y; s1.z += xv.z * xv.z; s1.w += xv.w * xv.w;
            } else {
                const float4 g = bn_out_grad(p, m, c, xv, sc, sh);
                s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
                s1.x += g.x * (xv.x - mu.x) * rs.x; s1.y += g.y * (xv.y - mu.y) * rs.y;
                s1.z += g.z * (xv.z - mu.z) * rs.z; s1.w += g.w * (xv.w - mu.w) * rs.w;
            }
        };
        long m = r0 + ty;
        for (; m + 12 < r1; m += 16) { body(m); body(m + 4); body(m + 8); body(m + 12); }   // 4 independent loads in flight
        for (; m < r1; m += 4) body(m);
    }
    __shared__ float4 red[2][4][64];
    red[0][ty][tx] = s0; red[1][ty][tx] = s1;
    __syncthreads();
    if (ty == 0 && active) {
        float4 a = red[0][0][tx], b = red[1][0][tx];
        for (int k = 1; k < 4; ++k) {
            const float4 u = red[0][k][tx], v = red[1][k][tx];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (MODE == 0) {
            atomicAdd(p.out0 + c + 0, a.x); atomicAdd(p.out0 + c + 1, a.y);
            atomicAdd(p.out0 + c + 2, a.z); atomicAdd(p.out0 + c + 3, a.w);
        } else {
            // TACO_BN_REPL replicas of the per-column sums: same-address double atomics from hundreds of workgroups
            // serialise in L2 (34 us for a 21 MB tensor); the consumers add the replicas up
            double* ds = p.dstat + (long)((blockIdx.y + blockIdx.z) % TACO_BN_REPL) * 3 * p.C;
            atomicAdd(ds + c + 0, (double)a.x); atomicAdd(ds + c + 1, (double)a.y);
            atomicAdd(ds + c + 2, (double)a.z); atomicAdd(ds + c + 3, (double)a.w);
            atomicAdd(ds + p.C + c + 0, (double)b.x); atomicAdd(ds + p.C + c + 1, (double)b.y);
            atomicAdd(ds + p.C + c + 2, (double)b.z); atomicAdd(ds + p.C + c + 3, (double)b.w);
        }
    }
}

static void col_reduce_grid(int M, int C, dim3& g, int& rpb) {
    const int gx = cdiv(C, 256);
    int gy = cdiv(768, gx);              // ~768 workgroups: enough to fill 256 CUs, few enough to keep the final
    rpb = cdiv(M, gy);                   // per-column atomics (one set per workgroup) off the contention cliff
    if (rpb < 64) rpb = 64;
    rpb = (rpb + 3) & ~3;
    gy = cdiv(M, rpb);
    g = dim3(gx, gy);
}

// =====================================================================================================
// BatchNorm (tf.layers.batch_normalization, training mode; modules.py:101)
// =====================================================================================================
// stats double[TACO_BN_REPL][3C] (replicated sum, sumsq over M rows) -> mean, var(biased), rstd, scale=gamma*rstd, shift=beta-mean*scale
__global__ void bn_finalize_k(const double* __restrict__ dstat, const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ mean, float* __restrict__ var, float* __restrict__ rstd,
                              float* __restrict__ scale, float* __restrict__ shift, int M, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < TACO_BN_REPL; ++r) { s0 += dstat[(long)r * 3 * C + c]; s1 += dstat[(long)r * 3 * C + C + c]; }
    const double mu = s0 / M;
    double v = s1 / M - mu * mu;
    if (v < 0.0) v = 0.0;
    const float r = (float)(1.0 / sqrt(v + (double)eps));
    mean[c] = (float)mu; var[c] = (float)v; rstd[c] = r;
    const float s = gamma[c] * r;
    scale[c] = s; shift[c] = beta[c] - (float)mu * s;
}

// inference-mode parameters from moving statistics
__global__ void bn_infer_params_k(const float* __restrict__ mm, const float* __restrict__ mv, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(mv[c] + eps);
    scale[c] = s; shift[c] = beta[c] - mm[c] * s;
}

// y = [maxpool2](x*scale + shift) [+ res]
__global__ void bn_apply_k(const float* __restrict__ x, int ldx, const float* __restrict__ scale, const float* __restrict__ shift,
                           const float* __restrict__ res, int ldr, float* __restrict__ y, int ldy, int M, int C, int T, int pool) {
    const int c4n = C / 4;
    const long total = (long)M * c4n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long m = idx / c4n;
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        float4 b = f4fma(*reinterpret_cast<const float4*>(x + m * ldx + c), sc, sh);
        if (pool && (int)(m % T) != T - 1) {
            const float4 bn = f4fma(*reinterpret_cast<const float4*>(x + (m + 1) * ldx + c), sc, sh);
            b.x = fmaxf(b.x, bn.x); b.y = fmaxf(b.y, bn.y); b.z = fmaxf(b.z, bn.z); b.w = fmaxf(b.w, bn.w);
        }
        if (res) {
            const float4 r = *reinterpret_cast<const float4*>(res + m * ldr + c);
            b.x += r.x; b.y += r.y; b.z += r.z; b.w += r.w;
        }
        *reinterpret_cast<float4*>(y + m * ldy + c) = b;
    }
}

// dx = gamma*rstd*(db - sum(db)/M - xhat*sum(db*xhat)/M), then the conv activation's mask (relu: x > 0).
// Also emits dgamma = sum(db*xhat), dbeta = sum(db) (block 0 only, ADDED into the gradient buffers).
struct BnBwd {
    ColRed r;  // x, dy, mean, rstd, scale, shift, dstat (sums from col_reduce MODE 2), M, C, T, pool
    const float* gamma; float* dgamma; float* dbeta; float* dbias; float* dx; int lddx; int relu;
};
__global__ __launch_bounds__(256) void bn_bwd_apply_k(BnBwd p) {
    // thread = one float4 column group (64 per workgroup = 256 columns), 4 row lanes; rows [r0, r1) of this workgroup
    const ColRed& q = p.r;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + tx) * 4;
    const bool active = c < q.C;
    const long r0 = (long)blockIdx.y * q.rows_per_block;
    const long r1 = min((long)q.M, r0 + q.rows_per_block);
    const float invM = 1.0f / (float)q.M;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const float4 sc = *reinterpret_cast<const float4*>(q.scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(q.shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(q.mean + c);
        const float4 rs = *reinterpret_cast<const float4*>(q.rstd + c);
        float sdb[4], sdx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double a = 0.0, b = 0.0;
            for (int r = 0; r < TACO_BN_REPL; ++r) { a += q.dstat[(long)r * 3 * q.C + c + k]; b += q.dstat[(long)r * 3 * q.C + q.C + c + k]; }
            sdb[k] = (float)a; sdx[k] = (float)b;
        }
        const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, rss[4] = {rs.x, rs.y, rs.z, rs.w}, scs[4] = {sc.x, sc.y, sc.z, sc.w};
        for (long m = r0 + ty; m < r1; m += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(q.x + m * q.ldx + c);
            const float4 g = bn_out_grad(q, m, c, xv, sc, sh);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {g.x, g.y, g.z, g.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xs[k] - mus[k]) * rss[k];
                float d = scs[k] * (gs[k] - sdb[k] * invM - xh * sdx[k] * invM);
                if (p.relu && !(xs[k] > 0.0f)) d = 0.0f;
                o[k] = d;
                bs[k] += d;
            }
            *reinterpret_cast<float4*>(p.dx + m * p.lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __shared__ float red[4][4][64];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][ty][tx] = bs[k];
    __syncthreads();
    if (ty == 0 && active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = red[k][0][tx] + red[k][1][tx] + red[k][2][tx] + red[k][3][tx];
            if (p.dbias)                                             // conv bias gradient = sum_m dx[m, c] (replicated, see above)
                atomicAdd(q.dstat + (long)(blockIdx.y % TACO_BN_REPL) * 3 * q.C + 2 * q.C + c + k, (double)t);
        }
    }
}

// dgamma += sum dY*xhat, dbeta += sum dY, dbias += sum dx (replica sums of the two kernels above)
__global__ void bn_bwd_finish_k(const double* __restrict__ dstat, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                float* __restrict__ dbias, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < TACO_BN_REPL; ++r) {
        s0 += dstat[(long)r * 3 * C + c]; s1 += dstat[(long)r * 3 * C + C + c]; s2 += dstat[(long)r * 3 * C + 2 * C + c];
    }
    dgamma[c] += (float)s1;
    dbeta[c] += (float)s0;
    if (dbias) dbias[c] += (float)s2;
}

// =====================================================================================================
// Highway gating (modules.py:77-90).  Z = x.[W_H|W_T] + [b_H|b_T] is [M,256]; in place: Z <- [H, Tg]
// =====================================================================================================
__global__ void highway_gate_k(float* __restrict__ Z, const float* __restrict__ x, float* __restrict__ y, int M) {
    const long total = (long)M * 32;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx & 31) * 4;
        const long m = idx >> 5;
        float4 h = *reinterpret_cast<float4*>(Z + m * 256 + c);
        float4 t = *reinterpret_cast<float4*>(Z + m * 256 + 128 + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + m * 128 + c);
        h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
        t.x = sigmoidf_(t.x); t.y = sigmoidf_(t.y); t.z = sigmoidf_(t.z); t.w = sigmoidf_(t.w);
        float4 o;
        o.x = h.x * t.x + xv.x * (1.f - t.x); o.y = h.y * t.y + xv.y * (1.f - t.y);
        o.z = h.z * t.z + xv.z * (1.f - t.z); o.w = h.w * t.w + xv.w * (1.f - t.w);
        *reinterpret_cast<float4*>(Z + m * 256 + c) = h;
        *reinterpret_cast<float4*>(Z + m * 256 + 128 + c) = t;
        *reinterpret_cast<float4*>(y + m * 128 + c) = o;
    }
}

// dZ = [dy*Tg*(H>0), dy*(H-x)*Tg*(1-Tg)];  dx_direct = dy*(1-Tg)
__global__ void highway_gate_bwd_k(const float* __restrict__ HT, const float* __restrict__ x, const float* __restrict__ dy,
                                   float* __restrict__ dZ, float* __restrict__ dx, int M) {
    const long total = (long)M * 32;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx & 31) * 4;
        const long m = idx >> 5;
        const float4 h = *reinterpret_cast<const float4*>(HT + m * 256 + c);
        const float4 t = *reinterpret_cast<const float4*>(HT + m * 256 + 128 + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + m * 128 + c);
        const float4 g = *reinterpret_cast<const float4*>(dy + m * 128 + c);
        float4 dh, dt, dd;
        dh.x = h.x > 0.f ? g.x * t.x : 0.f; dh.y = h.y > 0.f ? g.y * t.y : 0.f;
        dh.z = h.z > 0.f ? g.z * t.z : 0.f; dh.w = h.w > 0.f ? g.w * t.w : 0.f;
        dt.x = g.x * (h.x - xv.x) * t.x * (1.f - t.x); dt.y = g.y * (h.y - xv.y) * t.y * (1.f - t.y);
        dt.z = g.z * (h.z - xv.z) * t.z * (1.f - t.z); dt.w = g.w * (h.w - xv.w) * t.w * (1.f - t.w);
        dd.x = g.x * (1.f - t.x); dd.y = g.y * (1.f - t.y); dd.z = g.z * (1.f - t.z); dd.w = g.w * (1.f - t.w);
        *reinterpret_cast<float4*>(dZ + m * 256 + c) = dh;
        *reinterpret_cast<float4*>(dZ + m * 256 + 128 + c) = dt;
        *reinterpret_cast<float4*>(dx + m * 128 + c) = dd;
    }
}

// relu backward on a dense layer output: dpre = y > 0 ? dy : 0  (in place allowed)
__global__ void relu_bwd_k(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dpre, long n4) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        const float4 yv = reinterpret_cast<const float4*>(y)[idx];
        float4 g = reinterpret_cast<const float4*>(dy)[idx];
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        reinterpret_cast<float4*>(dpre)[idx] = g;
    }
}

// y (+)= a + b  (b optional)
__global__ void add_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n4, int accumulate) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(a)[idx];
        if (b) { const float4 w = reinterpret_cast<const float4*>(b)[idx]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        if (accumulate) { const float4 w = reinterpret_cast<const float4*>(y)[idx]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        reinterpret_cast<float4*>(y)[idx] = v;
    }
}

// =====================================================================================================
// L1 losses (models/tacotron.py:127-137) + sign gradients.  out/tgt [rows, C] with leading dims;
// sums[2*rep + 0] += sum|d| over all columns, sums[2*rep + 1] += sum|d| over columns < npri  (rep = workgroup % TACO_L1_REPL);
// grad[row, c] = sign(out - tgt) * (w_all + (c < npri ? w_pri : 0)); padded columns [C, ldg) get 0.
// =====================================================================================================
// rb_len > 0: logical row r is frame rb_off + r % rb_len of sequence r / rb_len (physical row (r / rb_len) * rb_stride + ...)
__global__ __launch_bounds__(256) void l1_loss_k(const float* __restrict__ out, int ldo, const float* __restrict__ tgt, int ldt,
                                                float* __restrict__ grad, int ldg, double* __restrict__ sums, long rows, int C,
                                                int npri, float w_all, float w_pri, int rb_len, int rb_stride, int rb_off) {
    // flat (row, column) index over the padded gradient width: consecutive lanes touch consecutive floats (rows of 1025 floats
    // are not 16-byte aligned, so 4-byte lanes are the coalesced access) and all lanes stay busy for narrow rows (80 mels);
    // four independent elements per lane and iteration keep enough loads in flight
    float s_all = 0.f, s_pri = 0.f;
    const unsigned total = (unsigned)(rows * ldg), stride = gridDim.x * 256u;
    auto one = [&](unsigned e) {
        const unsigned rl = e / (unsigned)ldg, c = e - rl * (unsigned)ldg;
        const unsigned r = rb_len ? (rl / (unsigned)rb_len) * (unsigned)rb_stride + (unsigned)rb_off + rl % (unsigned)rb_len : rl;
        float gv = 0.f;
        if ((int)c < C) {
            const float d = out[(long)r * ldo + c] - tgt[(long)r * ldt + c];
            const float a = fabsf(d);
            s_all += a;
            float w = w_all;
            if ((int)c < npri) { s_pri += a; w += w_pri; }
            gv = d > 0.f ? w : (d < 0.f ? -w : 0.f);
        }
        if (grad) grad[(long)r * ldg + c] = gv;
    };
    unsigned e = blockIdx.x * 256u + threadIdx.x;
    for (; e + 3u * stride < total && e + 3u * stride >= e; e += 4u * stride) { one(e); one(e + stride); one(e + 2u * stride); one(e + 3u * stride); }
    for (; e < total; e += stride) one(e);
    // one atomic pair per WORKGROUP (same-address atomics serialise: keep them to ~1k per launch)
    const double a = wave_sum_d((double)s_all), b = wave_sum_d((double)s_pri);
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {                       // TACO_L1_REPL replica pairs: ~1k same-address atomics cost ~20 us in L2
        double* sr = sums + 2 * (blockIdx.x % TACO_L1_REPL);
        atomicAdd(sr, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(sr + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// Wide rows (the linear spectrogram: 1025 columns in rows of 1028): ONE WAVE PER ROW.  The flat form above spends an integer division per
// element and keeps 4 elements per lane in flight -- 8 MB across the chip, less than the memory system needs to stream (85 us for 252 MB).
// Here a lane owns float4 columns lane, lane + 64, ... of its row: the gradient moves as 16-byte stores (its rows are 16-byte aligned),
// prediction and target -- rows of C = 1025 floats, not aligned -- as four dword loads each that the memory pipeline merges per cache
// line; all loads of a row are issued before the first use.  Needs ldg % 4 == 0, a 16-byte aligned grad, ldg <= 4 * 64 * L1_NV.
#define L1_NV 5
__global__ __launch_bounds__(256) void l1_loss_wide_k(const float* __restrict__ out, int ldo, const float* __restrict__ tgt, int ldt,
                                                     float* __restrict__ grad, int ldg, double* __restrict__ sums, int rows, int C,
                                                     int npri, float w_all, float w_pri, int rb_len, int rb_stride, int rb_off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s_all = 0.f, s_pri = 0.f;
    const int nv = (ldg / 4 + 63) / 64;
    for (int rl = blockIdx.x * 4 + wave; rl < rows; rl += gridDim.x * 4) {
        const long r = rb_len ? (long)(rl / rb_len) * rb_stride + rb_off + rl % rb_len : (long)rl;
        const float* o = out + r * ldo;
        const float* t = tgt + r * ldt;
        float ov[L1_NV][4], tv[L1_NV][4];
#pragma unroll
        for (int v = 0; v < L1_NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (v < nv && c < ldg) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ov[v][k] = c + k < C ? o[c + k] : 0.f;
                    tv[v][k] = c + k < C ? t[c + k] : 0.f;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < L1_NV; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (v < nv && c < ldg) {
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    g[k] = 0.f;
                    if (c + k < C) {
                        const float d = ov[v][k] - tv[v][k];
                        const float a = fabsf(d);
                        s_all += a;
                        float w = w_all;
                        if (c + k < npri) { s_pri += a; w += w_pri; }
                        g[k] = d > 0.f ? w : (d < 0.f ? -w : 0.f);
                    }
                }
                *reinterpret_cast<float4*>(grad + r * ldg + c) = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
    }
    const double a = wave_sum_d((double)s_all), b = wave_sum_d((double)s_pri);
    __shared__ double red[2][4];
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* sr = sums + 2 * (blockIdx.x % TACO_L1_REPL);
        atomicAdd(sr, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(sr + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

static void l1_launch(const float* out, int ldo, const float* tgt, int ldt, float* grad, int ldg, double* sums2, long rows, int C, int npri,
                      float w_all, float w_pri, int rb_len, int rb_stride, int rb_off, hipStream_t stream) {
    static const int no_wide = getenv("TACO_L1_WIDE") && atoi(getenv("TACO_L1_WIDE")) == 0;
    const bool wide = !no_wide && grad && C >= 256 && ldg <= 4 * 64 * L1_NV && !(ldg & 3) && !(reinterpret_cast<uintptr_t>(grad) & 15) &&
                      rows < (1L << 31);
    if (wide) {
        const long blocks = (rows + 3) / 4;
        hipLaunchKernelGGL(l1_loss_wide_k, dim3((int)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, stream, out, ldo, tgt, ldt, grad, ldg, sums2,
                           (int)rows, C, npri, w_all, w_pri, rb_len, rb_stride, rb_off);
        return;
    }
    const long blocks = (rows * ldg + 1023) / 1024;
    hipLaunchKernelGGL(l1_loss_k, dim3((int)(blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks)), dim3(256), 0, stream, out, ldo, tgt, ldt, grad, ldg, sums2,
                       rows, C, npri, w_all, w_pri, rb_len, rb_stride, rb_off);
}

// =====================================================================================================
// C-ABI wrappers
// =====================================================================================================
static inline int grid_for(long work_items, int block = 256) {
    long g = (work_items + block - 1) / block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int taco_embed_gather_fwd(const int* ids, const int* identities, const float* table, const float* spk_table,
                                     float* out, int N, int Ti, int Et, int Es, int vocab, int id_num, hipStream_t stream) {
    if (!ids || !table || !out || (Et & 3) || (Es & 3) || (Es > 0 && (!identities || !spk_table))) return TACO_EINVAL;
    const long total = (long)N * Ti * ((Et + Es) / 4);
    hipLaunchKernelGGL(embed_gather_k, dim3(grid_for(total)), dim3(256), 0, stream, ids, identities, table, spk_table, out,
                       N, Ti, Et, Es, vocab, id_num);
    TACO_RETURN_LAST();
}

extern "C" int taco_embed_scatter_bwd(const int* ids, const int* identities, const float* dE, float* dTable, float* dSpk,
                                      double* sparse_sumsq, int N, int Ti, int Et, int Es, int vocab, int id_num,
                                      hipStream_t stream) {
    if (!ids || !dE || !dTable || (Et & 3) || (Es & 3)) return TACO_EINVAL;
    const int blocks = grid_for((long)N * Ti * Et / 4);          // 4 floats per lane and pass; <= 1024 workgroups
    hipLaunchKernelGGL(embed_scatter_text_k, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, stream, ids, dE, dTable,
                       sparse_sumsq, N * Ti, Et, Et + Es, vocab);
    if (Es > 0) {
        if (!identities || !dSpk || Es > 256 || 256 % Es) return TACO_EINVAL;
        hipLaunchKernelGGL(embed_scatter_spk_k, dim3(N), dim3(256), 0, stream, identities, dE, dSpk, sparse_sumsq, Ti, Et, Es, id_num);
    }
    TACO_RETURN_LAST();
}

extern "C" int taco_col_sum(const float* x, int ldx, float* out, int M, int C, hipStream_t stream) {
    if (!x || !out || (C & 3) || (ldx & 3)) return TACO_EINVAL;
    ColRed p{}; p.x = x; p.ldx = ldx; p.out0 = out; p.M = M; p.C = C; p.T = M;
    dim3 g; col_reduce_grid(M, C, g, p.rows_per_block);
    hipLaunchKernelGGL(col_reduce_k<0>, g, dim3(256), 0, stream, p);
    TACO_RETURN_LAST();
}

// grouped bias gradients: one grid over the (column group, row block) workgroups of up to TACO_CS_MAX tensors
#define TACO_CS_MAX 40
struct ColSumGroup {
    int count;
    int first[TACO_CS_MAX + 1];
    unsigned short gx[TACO_CS_MAX];
    const float* x[TACO_CS_MAX]; float* out[TACO_CS_MAX];
    int ldx[TACO_CS_MAX], M[TACO_CS_MAX], C[TACO_CS_MAX], rpb[TACO_CS_MAX];
};
static_assert(sizeof(ColSumGroup) <= 4096, "kernel argument block");
__global__ __launch_bounds__(256) void col_sum_group_k(ColSumGroup g) {
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < g.count && b >= g.first[i + 1]) ++i;
    const int r = b - g.first[i], gx = g.gx[i];
    const int bx = r % gx, by = r / gx;
    const float* x = g.x[i];
    const int ldx = g.ldx[i], M = g.M[i], C = g.C[i], rpb = g.rpb[i];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = (bx * 64 + tx) * 4;
    const bool active = c < C;
    const long r0 = (long)by * rpb, r1 = min((long)M, r0 + rpb);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        auto body = [&](long m) {
            const float4 xv = *reinterpret_cast<const float4*>(x + m * ldx + c);
            s0.x += xv.x; s0.y += xv.y; s0.z += xv.z; s0.w += xv.w;
        };
        long m = r0 + ty;
        for (; m + 12 < r1; m += 16) { body(m); body(m + 4); body(m + 8); body(m + 12); }
        for (; m < r1; m += 4) body(m);
    }
    __shared__ float4 red[4][64];
    red[ty][tx] = s0;
    __syncthreads();
    if (ty == 0 && active) {
        float4 a = red[0][tx];
        for (int k = 1; k < 4; ++k) { const float4 u = red[k][tx]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
        float* o = g.out[i];
        atomicAdd(o + c + 0, a.x); atomicAdd(o + c + 1, a.y); atomicAdd(o + c + 2, a.z); atomicAdd(o + c + 3, a.w);
    }
}

extern "C" int taco_col_sum_group(const TacoColSum* items, int count, hipStream_t stream) {
    if (!items || count < 0) return TACO_EINVAL;
    for (int i0 = 0; i0 < count; i0 += TACO_CS_MAX) {
        ColSumGroup g;
        const int n = count - i0 < TACO_CS_MAX ? count - i0 : TACO_CS_MAX;
        g.count = n;
        int first = 0;
        for (int a = 0; a < n; ++a) {
            const TacoColSum& it = items[i0 + a];
            if (!it.x || !it.out || (it.C & 3) || (it.ldx & 3) || it.M <= 0) return TACO_EINVAL;
            dim3 d; int rpb;
            // a group shares the chip: ~192 workgroups per tensor instead of the ~768 of a lone launch
            const int gx = cdiv(it.C, 256);
            int gy = cdiv(192, gx); rpb = cdiv(it.M, gy); if (rpb < 64) rpb = 64; rpb = (rpb + 3) & ~3; gy = cdiv(it.M, rpb);
            g.first[a] = first; g.gx[a] = (unsigned short)gx; g.x[a] = it.x; g.out[a] = it.out; g.ldx[a] = it.ldx; g.M[a] = it.M;
            g.C[a] = it.C; g.rpb[a] = rpb;
            first += gx * gy;
        }
        for (int a = n; a <= TACO_CS_MAX; ++a) g.first[a] = first;
        for (int a = n; a < TACO_CS_MAX; ++a) { g.gx[a] = 1; g.x[a] = nullptr; g.out[a] = nullptr; g.ldx[a] = g.M[a] = g.C[a] = g.rpb[a] = 0; }
        hipLaunchKernelGGL(col_sum_group_k, dim3(first), dim3(256), 0, stream, g);
    }
    TACO_RETURN_LAST();
}

extern "C" int taco_bn_stats_fwd(const float* x, int ldx, const float* gamma, const float* beta, double* dstat_zeroed,
                                 float* mean, float* var, float* rstd, float* scale, float* shift, int M, int C, float eps,
                                 hipStream_t stream) {
    if (!x || !dstat_zeroed || (C & 3) || (ldx & 3)) return TACO_EINVAL;
    ColRed p{}; p.x = x; p.ldx = ldx; p.dstat = dstat_zeroed; p.M = M; p.C = C; p.T = M;
    dim3 g; col_reduce_grid(M, C, g, p.rows_per_block);
    hipLaunchKernelGGL(col_reduce_k<1>, g, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(bn_finalize_k, dim3(cdiv(C, 256)), dim3(256), 0, stream, dstat_zeroed, gamma, beta, mean, var, rstd, scale, shift, M, C, eps);
    TACO_RETURN_LAST();
}

// the same in two parts, for a tensor that is produced in pieces: sums over the frames [t0, t1) of every length-T sequence of x [N*T, C]
// (any number of calls, in any order, into the same zeroed dstat), then the finalisation over all M = N*T rows
extern "C" int taco_bn_stats_rows(const float* x, int ldx, double* dstat, int N, int T, int t0, int t1, int C, hipStream_t stream) {
    if (!x || !dstat || (C & 3) || (ldx & 3) || N <= 0 || t0 < 0 || t1 > T || t0 >= t1) return TACO_EINVAL;
    ColRed p{}; p.x = x; p.ldx = ldx; p.dstat = dstat; p.M = N * T; p.C = C; p.T = T;
    p.seq_len = T; p.seq_t0 = t0; p.seq_rows = t1 - t0;
    int rpb = cdiv(t1 - t0, cdiv(768, cdiv(C, 256) * N));
    if (rpb < 64) rpb = 64;
    rpb = (rpb + 3) & ~3;
    p.rows_per_block = rpb;
    hipLaunchKernelGGL(col_reduce_k<1>, dim3(cdiv(C, 256), cdiv(t1 - t0, rpb), N), dim3(256), 0, stream, p);
    TACO_RETURN_LAST();
}
extern "C" int taco_bn_finalize(const double* dstat, const float* gamma, const float* beta, float* mean, float* var, float* rstd,
                                float* scale, float* shift, int M, int C, float eps, hipStream_t stream) {
    if (!dstat || !gamma || !beta || !mean || !var || !rstd || !scale || !shift || M <= 0 || C <= 0) return TACO_EINVAL;
    hipLaunchKernelGGL(bn_finalize_k, dim3(cdiv(C, 256)), dim3(256), 0, stream, dstat, gamma, beta, mean, var, rstd, scale, shift, M, C, eps);
    TACO_RETURN_LAST();
}

extern "C" int taco_bn_infer_params(const float* moving_mean, const float* moving_var, const float* gamma, const float* beta,
                                    float* scale, float* shift, int C, float eps, hipStream_t stream) {
    hipLaunchKernelGGL(bn_infer_params_k, dim3(cdiv(C, 256)), dim3(256), 0, stream, moving_mean, moving_var, gamma, beta, scale, shift, C, eps);
    TACO_RETURN_LAST();
}

extern "C" int taco_bn_apply_fwd(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldr,
                                 float* y, int ldy, int M, int C, int T, int pool, hipStream_t stream) {
    if (!x || !y || (C & 3) || (ldx & 3) || (ldy & 3) || (res && (ldr & 3)) || M % T) return TACO_EINVAL;
    hipLaunchKernelGGL(bn_apply_k, dim3(grid_for((long)M * C / 4)), dim3(256), 0, stream, x, ldx, scale, shift, res, ldr, y, ldy, M, C, T, pool);
    TACO_RETURN_LAST();
}

extern "C" int taco_bn_bwd(const float* x, int ldx, const float* dy, int lddy, const float* mean, const float* rstd,
                           const float* scale, const float* shift, const float* gamma, double* dstat_zeroed, float* dgamma,
                           float* dbeta, float* dbias, float* dx, int lddx, int M, int C, int T, int pool, int relu,
                           hipStream_t stream) {
    if (!x || !dy || !dx || !dstat_zeroed || (C & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3) || M % T) return TACO_EINVAL;
    BnBwd b{};
    ColRed& p = b.r;
    p.x = x; p.ldx = ldx; p.dy = dy; p.lddy = lddy; p.mean = mean; p.rstd = rstd; p.scale = scale; p.shift = shift;
    p.dstat = dstat_zeroed; p.M = M; p.C = C; p.T = T; p.pool = pool;
    dim3 g; col_reduce_grid(M, C, g, p.rows_per_block);
    hipLaunchKernelGGL(col_reduce_k<2>, g, dim3(256), 0, stream, p);
    b.gamma = gamma; b.dgamma = dgamma; b.dbeta = dbeta; b.dbias = dbias; b.dx = dx; b.lddx = lddx; b.relu = relu;
    hipLaunchKernelGGL(bn_bwd_apply_k, g, dim3(256), 0, stream, b);
    hipLaunchKernelGGL(bn_bwd_finish_k, dim3(cdiv(C, 256)), dim3(256), 0, stream, dstat_zeroed, dgamma, dbeta, dbias, C);
    TACO_RETURN_LAST();
}

extern "C" int taco_highway_gate_fwd(float* Z, const float* x, float* y, int M, hipStream_t stream) {
    hipLaunchKernelGGL(highway_gate_k, dim3(grid_for((long)M * 32)), dim3(256), 0, stream, Z, x, y, M);
    TACO_RETURN_LAST();
}

extern "C" int taco_highway_gate_bwd(const float* HT, const float* x, const float* dy, float* dZ, float* dx, int M, hipStream_t stream) {
    hipLaunchKernelGGL(highway_gate_bwd_k, dim3(grid_for((long)M * 32)), dim3(256), 0, stream, HT, x, dy, dZ, dx, M);
    TACO_RETURN_LAST();
}

extern "C" int taco_relu_bwd(const float* y, const float* dy, float* dpre, long n, hipStream_t stream) {
    if (n & 3) return TACO_EINVAL;
    hipLaunchKernelGGL(relu_bwd_k, dim3(grid_for(n / 4)), dim3(256), 0, stream, y, dy, dpre, n / 4);
    TACO_RETURN_LAST();
}

extern "C" int taco_add(const float* a, const float* b, float* y, long n, int accumulate, hipStream_t stream) {
    if (n & 3) return TACO_EINVAL;
    hipLaunchKernelGGL(add_k, dim3(grid_for(n / 4)), dim3(256), 0, stream, a, b, y, n / 4, accumulate);
    TACO_RETURN_LAST();
}

// =====================================================================================================
// Alignment regularisers (models/tacotron.py:140-171).  One workgroup per batch row, thread = encoder position ti (loops
// for Ti > blockDim); each thread walks its column of S decoder steps (reads are coalesced across ti).
//   pr = softmax over s of a[n, :, ti];  g = dL/dpr;  dL/da[s] = pr[s] * (g[s] - sum_s' pr[s'] g[s'])  (+ variance term on a)
// =====================================================================================================
struct AlignReg { const float* a; float* da; double* loss; int N, S, Ti; float w_over, w_one, w_var, w_ent; };

__global__ __launch_bounds__(256) void align_reg_k(AlignReg p) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* a = p.a + (long)n * p.S * p.Ti;
    float* da = p.da + (long)n * p.S * p.Ti;
    const int S = p.S, Ti = p.Ti;
    __shared__ double red[256];
    __shared__ float rowmean;
    // ---- variance_between_row needs mean_ti(sum_s a) first
    double part = 0.0;
    for (int ti = tid; ti < Ti; ti += 256) {
        float r = 0.f;
        for (int s = 0; s < S; ++s) r += a[(long)s * Ti + ti];
        part += r;
    }
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) rowmean = (float)(red[0] / Ti);
    __syncthreads();
    const float m = rowmean;
    const bool use_pr = p.w_over != 0.f || p.w_one != 0.f || p.w_ent != 0.f;
    const float ent_scale = p.w_ent / ((float)p.N * (float)Ti * (float)S);
    const int over_end = S - 41 == -1 ? S : 40 + (S - 41);           // exclusive end of the tf.slice at :159
    double loss = 0.0;
    for (int ti = tid; ti < Ti; ti += 256) {
        float mx = -3.0e38f, rsum = 0.f;
        for (int s = 0; s < S; ++s) { const float v = a[(long)s * Ti + ti]; mx = fmaxf(mx, v); rsum += v; }
        float den = 0.f;
        if (use_pr) for (int s = 0; s < S; ++s) den += expf(a[(long)s * Ti + ti] - mx);
        const float inv = use_pr ? 1.0f / den : 0.f;
        const float dvar = p.w_var * 2.0f * (rsum - m);
        loss += (double)p.w_var * (double)(m - rsum) * (double)(m - rsum);
        // pass 1: g and sum pr*g
        float dot = 0.f, prev = 0.f;
        if (use_pr) {
            for (int s = 0; s < S; ++s) {
                const float pr = expf(a[(long)s * Ti + ti] - mx) * inv;
                const float nxt = s + 1 < S ? expf(a[(long)(s + 1) * Ti + ti] - mx) * inv : 0.f;
                float g = 0.f;
                if (p.w_ent != 0.f) { const float lg = logf(pr); g -= ent_scale * (lg + 1.0f); loss -= (double)ent_scale * pr * lg; }
                if (p.w_one != 0.f) {
                    if (s + 1 < S) { const float d = pr - nxt; g += p.w_one * ((d > 0.f) - (d < 0.f)); loss += (double)p.w_one * fabsf(d); }
                    if (s > 0) { const float d = prev - pr; g -= p.w_one * ((d > 0.f) - (d < 0.f)); }
                }
                if (p.w_over != 0.f && ti == 0 && s >= 40 && s < over_end) { g += p.w_over; loss += (double)p.w_over * pr; }
                dot = fmaf(pr, g, dot);
                prev = pr;
            }
        }
        // pass 2: gradient wrt a
        prev = 0.f;
        for (int s = 0; s < S; ++s) {
            float out = dvar;
            if (use_pr) {
                const float pr = expf(a[(long)s * Ti + ti] - mx) * inv;
                const float nxt = s + 1 < S ? expf(a[(long)(s + 1) * Ti + ti] - mx) * inv : 0.f;
                float g = 0.f;
                if (p.w_ent != 0.f) g -= ent_scale * (logf(pr) + 1.0f);
                if (p.w_one != 0.f) {
                    if (s + 1 < S) { const float d = pr - nxt; g += p.w_one * ((d > 0.f) - (d < 0.f)); }
                    if (s > 0) { const float d = prev - pr; g -= p.w_one * ((d > 0.f) - (d < 0.f)); }
                }
                if (p.w_over != 0.f && ti == 0 && s >= 40 && s < over_end) g += p.w_over;
                out += pr * (g - dot);
                prev = pr;
            }
            da[(long)s * Ti + ti] = out;
        }
    }
    red[tid] = loss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) atomicAdd(p.loss, red[0]);
}

extern "C" int taco_align_regularity(const float* align, float* dalign, double* loss_sum, int N, int S, int Ti, float overwrought,
                                     float oneorder_dynamic, float variance_between_row, float alignment_entropy,
                                     hipStream_t stream) {
    if (!align || !dalign || !loss_sum || N <= 0 || S <= 0 || Ti <= 0) return TACO_EINVAL;
    if (overwrought != 0.f && S < 40) return TACO_EINVAL;      // tf.slice(pr, [0,0,40], [N,1,S-41]) needs S >= 40
    AlignReg p{align, dalign, loss_sum, N, S, Ti, overwrought, oneorder_dynamic, variance_between_row, alignment_entropy};
    hipLaunchKernelGGL(align_reg_k, dim3(N), dim3(256), 0, stream, p);
    TACO_RETURN_LAST();
}

extern "C" int taco_l1_loss(const float* out, int ldo, const float* tgt, int ldt, float* grad, int ldg, double* sums2,
                            long rows, int C, int npri, float w_all, float w_pri, hipStream_t stream) {
    if (!out || !tgt || !sums2 || ldg < C || (ldg & 3) || rows * ldg >= (1L << 31)) return TACO_EINVAL;
    l1_launch(out, ldo, tgt, ldt, grad, ldg, sums2, rows, C, npri, w_all, w_pri, 0, 0, 0, stream);
    TACO_RETURN_LAST();
}

extern "C" int taco_l1_loss_rows(const float* out, int ldo, const float* tgt, int ldt, float* grad, int ldg, double* sums2, int N,
                                 int T, int f0, int f1, int C, int npri, float w_all, float w_pri, hipStream_t stream) {
    if (!out || !tgt || !sums2 || ldg < C || (ldg & 3) || N <= 0 || T <= 0 || f0 < 0 || f1 > T || f0 >= f1) return TACO_EINVAL;
    if ((long)N * T * ldg >= (1L << 31)) return TACO_EINVAL;
    const long rows = (long)N * (f1 - f0);
    l1_launch(out, ldo, tgt, ldt, grad, ldg, sums2, rows, C, npri, w_all, w_pri, f1 - f0, T, f0, stream);
    TACO_RETURN_LAST();
}
