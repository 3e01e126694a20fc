// Persistent CLUSTER kernels for the two residual decoder GRU(256) recurrences (reference
// models/tacotron.py:76-80: MultiRNNCell[..., ResidualWrapper(GRUCell(256)), ResidualWrapper(GRUCell(256))]).
//
// Why: a launch per dependent stage costs ~6 us on MI355X (kernel boundary + cold L2 on 8 XCDs), and a GRU
// step has two dependent mat-vec stages -> 13 us/step.  Here the whole S-step recurrence is ONE launch.
// W_h (768 KB fp32) does not fit one CU's register file, so a CLUSTER of 4 workgroups shares two batch rows:
// workgroup w owns hidden indices J_w = [64w, 64w+64): the r/u/c columns J_w of the recurrent kernels stay in
// its registers (192 VGPRs/lane) for the whole kernel.  Per step the cluster all-gathers r*h (before the
// candidate product) and h' (before the next gate product): 2 x 256 floats per row, moved as 8-byte
// {epoch, value} granules written with ONE agent-scope store each and polled with agent-scope loads
// (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; no fences, placement independent).
// Different clusters never communicate.  Grid = 4 * ceil(N/2) workgroups, all co-resident (<= 256 CUs);
// every spin is bounded and reports through an error word instead of hanging the GPU.
//
// GRU semantics: tf.contrib.rnn.GRUCell, input projection hoisted (xp = x.Wx + b).
#include "common.hpp"

typedef unsigned long long u64;
#define HD 256
#define SPIN_LIMIT (1 << 22)

__device__ __forceinline__ void put_granule(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// poll NG granules together: all loads are issued back-to-back, so a thread pays ONE round trip per poll pass
template <int NG>
__device__ __forceinline__ void get_granules(const u64* const (&ptr)[NG], unsigned epoch, float (&out)[NG], int* err) {
    u64 x[NG];
    int spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < NG; ++i) x[i] = __hip_atomic_load(ptr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < NG; ++i) all = all && ((unsigned)(x[i] >> 32) == epoch);
        if (all) break;
        if (++spins > SPIN_LIMIT) { atomicExch(err, 1); break; }
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) out[i] = __uint_as_float((unsigned)x[i]);
}

struct Gru256 {
    const float* xp;      // [N,S,768] hoisted input projection (+bias): r | u | c
    const float* whg;     // [256,512]
    const float* whc;     // [256,256]
    const float* res;     // [N,S,256] residual input (or null)
    float *r, *u, *c, *rh, *h, *d;     // [N,S,256] saved gates / r*h_prev / state / residual output
    u64* xchg;            // [2][clusters][2 rows][256] granules (zeroed before launch): rh, h'
    int* err;
    int N, S;
    // backward
    const float* dout;    // [N,S,256]
    float* dxp;           // [N,S,768]
};

// LDS row layout: every 64-float quarter is followed by a 16-byte pad, so the 2 (or 4) distinct addresses a
// wave reads per ds_read_b128 (one per K half / quarter) fall on different banks
#define HPAD 4
#define HROW (HD + 4 * HPAD)
__device__ __forceinline__ int hidx(int k) { return k + (k >> 6) * HPAD; }

__global__ __launch_bounds__(256, 1) void gru256_cluster_fwd_k(Gru256 p) {
    const int tid = threadIdx.x;
    // cluster members share blockIdx % 8 (same XCD under round-robin placement: speed only, never correctness)
    const int nclus = gridDim.x / 4;
    int w, cl;
    if ((nclus & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; w = q & 3; cl = (q >> 2) * 8 + xcd; }
    else { w = blockIdx.x & 3; cl = blockIdx.x >> 2; }
    const int row0 = cl * 2;
    __shared__ __attribute__((aligned(16))) float h_lds[2][HROW];
    __shared__ __attribute__((aligned(16))) float rh_lds[2][HROW];
    __shared__ float u_lds[2][64];

    // ---- register-resident weights
    // gates: thread (gcol = tid>>1 in [0,128), kh = tid&1): k in [128kh, 128kh+128), column r_{64w+gcol} or u_{...}
    const int gcol = tid >> 1, kh = tid & 1;
    const int gc = gcol < 64 ? 64 * w + gcol : 256 + 64 * w + (gcol - 64);
    float wg[128];
#pragma unroll
    for (int k = 0; k < 128; ++k) wg[k] = p.whg[(long)(kh * 128 + k) * 512 + gc];
    // candidate: thread (ccol = tid>>2 in [0,64), kq = tid&3): k in [64kq, 64kq+64), column c_{64w+ccol}
    const int ccol = tid >> 2, kq = tid & 3;
    float wc[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wc[k] = p.whc[(long)(kq * 64 + k) * 256 + 64 * w + ccol];

    for (int i = tid; i < 2 * HROW; i += 256) { (&h_lds[0][0])[i] = 0.0f; (&rh_lds[0][0])[i] = 0.0f; }
    __syncthreads();

    const bool ok0 = row0 < p.N, ok1 = row0 + 1 < p.N;
    const long rb0 = (long)min(row0, p.N - 1) * p.S, rb1 = (long)min(row0 + 1, p.N - 1) * p.S;
    u64* xr = p.xchg + ((long)cl * 2) * HD;                               // rh granules  [2][256]
    u64* xh = p.xchg + ((long)nclus * 2 + (long)cl * 2) * HD;             // h' granules
    const int j_own = 64 * w + (gcol & 63);                               // hidden index of this thread's gate column
    const int jc = 64 * w + ccol;                                         // hidden index of the candidate column

    for (int s = 0; s < p.S; ++s) {
        const unsigned epoch = (unsigned)s + 1;
        // ---- gates (both rows), K half per lane, lane pairs combined with one cross-lane add
        float a0 = 0.0f, a1 = 0.0f;
        // issue the step's input-projection loads first: their latency hides under the FMA loops
        const float xg0 = p.xp[(rb0 + s) * 768 + gc], xg1 = p.xp[(rb1 + s) * 768 + gc];
        const float xc0 = p.xp[(rb0 + s) * 768 + 512 + jc], xc1 = p.xp[(rb1 + s) * 768 + 512 + jc];
        const float* h0 = &h_lds[0][kh * (128 + 2 * HPAD)];
        const float* h1 = &h_lds[1][kh * (128 + 2 * HPAD)];
#pragma unroll
        for (int k4 = 0; k4 < 32; ++k4) {
            const float4 v0 = *reinterpret_cast<const float4*>(h0 + hidx(k4 * 4));
            const float4 v1 = *reinterpret_cast<const float4*>(h1 + hidx(k4 * 4));
            a0 = fmaf(v0.x, wg[k4 * 4], a0); a1 = fmaf(v1.x, wg[k4 * 4], a1);
            a0 = fmaf(v0.y, wg[k4 * 4 + 1], a0); a1 = fmaf(v1.y, wg[k4 * 4 + 1], a1);
            a0 = fmaf(v0.z, wg[k4 * 4 + 2], a0); a1 = fmaf(v1.z, wg[k4 * 4 + 2], a1);
            a0 = fmaf(v0.w, wg[k4 * 4 + 3], a0); a1 = fmaf(v1.w, wg[k4 * 4 + 3], a1);
        }
        a0 = group_sum<2>(a0);
        a1 = group_sum<2>(a1);
        if (kh == 0) {
            const float g0 = fast_sigmoid(a0 + xg0);
            const float g1 = fast_sigmoid(a1 + xg1);
            if (gcol < 64) {
                const float q0 = g0 * h_lds[0][hidx(j_own)], q1 = g1 * h_lds[1][hidx(j_own)];
                rh_lds[0][hidx(j_own)] = q0; rh_lds[1][hidx(j_own)] = q1;
                put_granule(xr + j_own, epoch, q0);
                put_granule(xr + HD + j_own, epoch, q1);
                if (ok0) { p.r[(rb0 + s) * HD + j_own] = g0; p.rh[(rb0 + s) * HD + j_own] = q0; }
                if (ok1) { p.r[(rb1 + s) * HD + j_own] = g1; p.rh[(rb1 + s) * HD + j_own] = q1; }
            } else {
                u_lds[0][gcol - 64] = g0; u_lds[1][gcol - 64] = g1;
                if (ok0) p.u[(rb0 + s) * HD + j_own] = g0;
                if (ok1) p.u[(rb1 + s) * HD + j_own] = g1;
            }
        }
        // ---- gather the peers' r*h (384 granules, threads 0..191 take two each)
        if (tid < 192) {
            const int peer = tid >> 6, jj = tid & 63;
            const int pw = peer + (peer >= w ? 1 : 0);
            const int j = 64 * pw + jj;
            const u64* const ptr[2] = {xr + j, xr + HD + j};
            float v[2];
            get_granules<2>(ptr, epoch, v, p.err);
            rh_lds[0][hidx(j)] = v[0]; rh_lds[1][hidx(j)] = v[1];
        }
        __syncthreads();
        // ---- candidate + state for hidden index jc (both rows), K quarter per lane
        float c0 = 0.0f, c1 = 0.0f;
        const float* q0p = &rh_lds[0][hidx(kq * 64)];
        const float* q1p = &rh_lds[1][hidx(kq * 64)];
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            const float4 v0 = *reinterpret_cast<const float4*>(q0p + k4 * 4);
            const float4 v1 = *reinterpret_cast<const float4*>(q1p + k4 * 4);
            c0 = fmaf(v0.x, wc[k4 * 4], c0); c1 = fmaf(v1.x, wc[k4 * 4], c1);
            c0 = fmaf(v0.y, wc[k4 * 4 + 1], c0); c1 = fmaf(v1.y, wc[k4 * 4 + 1], c1);
            c0 = fmaf(v0.z, wc[k4 * 4 + 2], c0); c1 = fmaf(v1.z, wc[k4 * 4 + 2], c1);
            c0 = fmaf(v0.w, wc[k4 * 4 + 3], c0); c1 = fmaf(v1.w, wc[k4 * 4 + 3], c1);
        }
        c0 = group_sum<4>(c0); c1 = group_sum<4>(c1);
        if (kq == 0) {
            const float cc0 = fast_tanh(c0 + xc0);
            const float cc1 = fast_tanh(c1 + xc1);
            const float u0 = u_lds[0][ccol], u1 = u_lds[1][ccol];
            const float hn0 = u0 * h_lds[0][hidx(jc)] + (1.0f - u0) * cc0;
            const float hn1 = u1 * h_lds[1][hidx(jc)] + (1.0f - u1) * cc1;
            h_lds[0][hidx(jc)] = hn0; h_lds[1][hidx(jc)] = hn1;
            put_granule(xh + jc, epoch, hn0);
            put_granule(xh + HD + jc, epoch, hn1);
            if (ok0) {
                p.c[(rb0 + s) * HD + jc] = cc0; p.h[(rb0 + s) * HD + jc] = hn0;
                if (p.d) p.d[(rb0 + s) * HD + jc] = p.res[(rb0 + s) * HD + jc] + hn0;
            }
            if (ok1) {
                p.c[(rb1 + s) * HD + jc] = cc1; p.h[(rb1 + s) * HD + jc] = hn1;
                if (p.d) p.d[(rb1 + s) * HD + jc] = p.res[(rb1 + s) * HD + jc] + hn1;
            }
        }
        // ---- gather the peers' h'
        if (tid < 192) {
            const int peer = tid >> 6, jj = tid & 63;
            const int pw = peer + (peer >= w ? 1 : 0);
            const int j = 64 * pw + jj;
            const u64* const ptr[2] = {xh + j, xh + HD + j};
            float v[2];
            get_granules<2>(ptr, epoch, v, p.err);
            h_lds[0][hidx(j)] = v[0]; h_lds[1][hidx(j)] = v[1];
        }
        __syncthreads();
    }
}

// BPTT twin.  Workgroup w owns hidden indices k in J_w: it keeps ROWS J_w of Whc ([64,256]) and Whg ([64,512]) in
// registers; per step the cluster all-gathers dcp (candidate pre-activation gradient, 256/row) and
// dg (gate pre-activation gradients, 512/row).
__global__ __launch_bounds__(256, 1) void gru256_cluster_bwd_k(Gru256 p) {
    const int tid = threadIdx.x;
    const int nclus = gridDim.x / 4;
    int w, cl;
    if ((nclus & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; w = q & 3; cl = (q >> 2) * 8 + xcd; }
    else { w = blockIdx.x & 3; cl = blockIdx.x >> 2; }
    const int row0 = cl * 2;
    __shared__ __attribute__((aligned(16))) float dcp_lds[2][HD + 3 * HPAD];        // 4 quarters, padded
    __shared__ __attribute__((aligned(16))) float dg_lds[2][2 * HD + 3 * HPAD];     // 4 quarters of 128, padded
    __shared__ float own_lds[2][64][4];                                              // r, hprev, dhdirect, du*u(1-u)

    // thread (kk = tid>>2 in [0,64), jq = tid&3)
    const int kk = tid >> 2, jq = tid & 3;
    const int k_own = 64 * w + kk;
    float wcT[64];     // Whc[k_own][64jq .. 64jq+64)
    float wgT[128];    // Whg[k_own][128jq .. 128jq+128)
#pragma unroll
    for (int j = 0; j < 64; ++j) wcT[j] = p.whc[(long)k_own * 256 + jq * 64 + j];
#pragma unroll
    for (int j = 0; j < 128; ++j) wgT[j] = p.whg[(long)k_own * 512 + jq * 128 + j];

    const bool ok[2] = {row0 < p.N, row0 + 1 < p.N};
    const long rb[2] = {(long)min(row0, p.N - 1) * p.S, (long)min(row0 + 1, p.N - 1) * p.S};
    u64* xc = p.xchg + ((long)cl * 2) * HD;                                  // dcp granules [2][256]
    u64* xg = p.xchg + ((long)nclus * 2) * HD + ((long)cl * 2) * 2 * HD;     // dg granules  [2][512]

    // owner mapping for the elementwise part: thread t < 128: (row = t>>6, jj = t&63) hidden index 64w + jj
    const int er = tid >> 6, ej = tid & 63;
    const int e_j = 64 * w + ej;
    float dhT = 0.0f;
    if (tid < 128) dhT = p.dout[(rb[er] + p.S - 1) * HD + e_j];

    for (int s = p.S - 1; s >= 0; --s) {
        const unsigned epoch = (unsigned)(p.S - s);
        // ---- elementwise at the owner: dcp, du, direct dh
        if (tid < 128) {
            const long o = (rb[er] + s) * HD + e_j;
            const float r = p.r[o], u = p.u[o], c = p.c[o];
            const float hp = s > 0 ? p.h[o - HD] : 0.0f;
            const float du = dhT * (hp - c);
            const float dcp = dhT * (1.0f - u) * (1.0f - c * c);
            dcp_lds[er][e_j + (e_j >> 6) * HPAD] = dcp;
            put_granule(xc + er * HD + e_j, epoch, dcp);
            own_lds[er][ej][0] = r; own_lds[er][ej][1] = hp; own_lds[er][ej][2] = dhT * u; own_lds[er][ej][3] = du * u * (1.0f - u);
            if (ok[er]) p.dxp[(rb[er] + s) * 768 + 512 + e_j] = dcp;
        }
        if (tid < 192) {
            const int peer = tid >> 6, jj = tid & 63;
            const int pw = peer + (peer >= w ? 1 : 0);
            const int j = 64 * pw + jj;
            const u64* const ptr[2] = {xc + j, xc + HD + j};
            float v[2];
            get_granules<2>(ptr, epoch, v, p.err);
            dcp_lds[0][j + (j >> 6) * HPAD] = v[0]; dcp_lds[1][j + (j >> 6) * HPAD] = v[1];
        }
        __syncthreads();
        // ---- drh[k_own] = sum_j dcp[j] * Whc[k_own][j]  (quarter per lane, both rows)
        float d0 = 0.0f, d1 = 0.0f;
        {
            const float* a0 = &dcp_lds[0][jq * (64 + HPAD)];
            const float* a1 = &dcp_lds[1][jq * (64 + HPAD)];
#pragma unroll
            for (int j4 = 0; j4 < 16; ++j4) {
                const float4 v0 = *reinterpret_cast<const float4*>(a0 + j4 * 4);
                const float4 v1 = *reinterpret_cast<const float4*>(a1 + j4 * 4);
                d0 = fmaf(v0.x, wcT[j4 * 4], d0); d1 = fmaf(v1.x, wcT[j4 * 4], d1);
                d0 = fmaf(v0.y, wcT[j4 * 4 + 1], d0); d1 = fmaf(v1.y, wcT[j4 * 4 + 1], d1);
                d0 = fmaf(v0.z, wcT[j4 * 4 + 2], d0); d1 = fmaf(v1.z, wcT[j4 * 4 + 2], d1);
                d0 = fmaf(v0.w, wcT[j4 * 4 + 3], d0); d1 = fmaf(v1.w, wcT[j4 * 4 + 3], d1);
            }
        }
        d0 = group_sum<4>(d0); d1 = group_sum<4>(d1);
        float dhp0 = 0.0f, dhp1 = 0.0f;         // partial dh_{s-1}[k_own] (valid in lanes jq == 0)
        if (jq == 0) {
            const float drh[2] = {d0, d1};
            float dhp[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float r = own_lds[b][kk][0], hp = own_lds[b][kk][1];
                const float dgr = drh[b] * hp * r * (1.0f - r);
                const float dgu = own_lds[b][kk][3];
                dhp[b] = own_lds[b][kk][2] + drh[b] * r;
                const int ir = k_own, iu = HD + k_own;
                dg_lds[b][ir + (ir >> 7) * HPAD] = dgr;
                dg_lds[b][iu + (iu >> 7) * HPAD] = dgu;
                put_granule(xg + b * 2 * HD + ir, epoch, dgr);
                put_granule(xg + b * 2 * HD + iu, epoch, dgu);
                if (ok[b]) { p.dxp[(rb[b] + s) * 768 + k_own] = dgr; p.dxp[(rb[b] + s) * 768 + HD + k_own] = dgu; }
            }
            dhp0 = dhp[0]; dhp1 = dhp[1];
        }
        // ---- gather the peers' dg: 3 peers x (64 r + 64 u) x 2 rows = 768 granules, 3 per thread
        {
            const u64* ptr[3];
            int li[3], lb[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int g = tid + it * 256;                 // [0,768): (row, peer, half, jj)
                const int b = g / 384, rem = g - b * 384;
                const int peer = rem >> 7, hj = rem & 127;
                const int pw = peer + (peer >= w ? 1 : 0);
                const int idx = (hj >> 6) * HD + 64 * pw + (hj & 63);
                ptr[it] = xg + b * 2 * HD + idx; li[it] = idx + (idx >> 7) * HPAD; lb[it] = b;
            }
            const u64* const cptr[3] = {ptr[0], ptr[1], ptr[2]};
            float v[3];
            get_granules<3>(cptr, epoch, v, p.err);
#pragma unroll
            for (int it = 0; it < 3; ++it) dg_lds[lb[it]][li[it]] = v[it];
        }
        __syncthreads();
        // ---- dh_{s-1}[k_own] += sum_j dg[j] * Whg[k_own][j]   (j over 512, quarter of 128 per lane)
        float e0 = 0.0f, e1 = 0.0f;
        {
            const float* a0 = &dg_lds[0][jq * (128 + HPAD)];
            const float* a1 = &dg_lds[1][jq * (128 + HPAD)];
#pragma unroll
            for (int j4 = 0; j4 < 32; ++j4) {
                const float4 v0 = *reinterpret_cast<const float4*>(a0 + j4 * 4);
                const float4 v1 = *reinterpret_cast<const float4*>(a1 + j4 * 4);
                e0 = fmaf(v0.x, wgT[j4 * 4], e0); e1 = fmaf(v1.x, wgT[j4 * 4], e1);
                e0 = fmaf(v0.y, wgT[j4 * 4 + 1], e0); e1 = fmaf(v1.y, wgT[j4 * 4 + 1], e1);
                e0 = fmaf(v0.z, wgT[j4 * 4 + 2], e0); e1 = fmaf(v1.z, wgT[j4 * 4 + 2], e1);
                e0 = fmaf(v0.w, wgT[j4 * 4 + 3], e0); e1 = fmaf(v1.w, wgT[j4 * 4 + 3], e1);
            }
        }
        e0 = group_sum<4>(e0); e1 = group_sum<4>(e1);
        // hand dh_{s-1} to the elementwise owner threads through LDS (own_lds slot 2 is free again)
        if (jq == 0) { own_lds[0][kk][2] = dhp0 + e0; own_lds[1][kk][2] = dhp1 + e1; }
        __syncthreads();
        if (tid < 128 && s > 0) dhT = own_lds[er][ej][2] + p.dout[(rb[er] + s - 1) * HD + e_j];
    }
}

static int gru256_grid(int N) { return 4 * ((N + 1) / 2); }

extern "C" int taco_gru256_seq_fwd(const float* xp, const float* whg, const float* whc, const float* res, float* r, float* u,
                                   float* c, float* rh, float* h, float* d, void* xchg, int* err, int N, int S,
                                   hipStream_t st) {
    if (!xp || !whg || !whc || !r || !u || !c || !rh || !h || !xchg || !err || N <= 0 || S <= 0) return TACO_EINVAL;
    if (d && !res) return TACO_EINVAL;
    if (gru256_grid(N) > 256) return TACO_EINVAL;           // all workgroups must be co-resident
    const int nclus = (N + 1) / 2;
    if (hipMemsetAsync(xchg, 0, (size_t)nclus * 2 * 2 * HD * sizeof(u64), st) != hipSuccess) return TACO_EINVAL;
    Gru256 p{};
    p.xp = xp; p.whg = whg; p.whc = whc; p.res = res; p.r = r; p.u = u; p.c = c; p.rh = rh; p.h = h; p.d = d;
    p.xchg = (u64*)xchg; p.err = err; p.N = N; p.S = S;
    hipLaunchKernelGGL(gru256_cluster_fwd_k, dim3(gru256_grid(N)), dim3(256), 0, st, p);
    TACO_RETURN_LAST();
}

extern "C" int taco_gru256_seq_bwd(const float* dout, const float* whg, const float* whc, const float* r, const float* u,
                                   const float* c, const float* h, float* dxp, void* xchg, int* err, int N, int S,
                                   hipStream_t st) {
    if (!dout || !whg || !whc || !r || !u || !c || !h || !dxp || !xchg || !err || N <= 0 || S <= 0) return TACO_EINVAL;
    if (gru256_grid(N) > 256) return TACO_EINVAL;
    const int nclus = (N + 1) / 2;
    if (hipMemsetAsync(xchg, 0, (size_t)nclus * 2 * 3 * HD * sizeof(u64), st) != hipSuccess) return TACO_EINVAL;
    Gru256 p{};
    p.dout = dout; p.whg = whg; p.whc = whc; p.r = const_cast<float*>(r); p.u = const_cast<float*>(u);
    p.c = const_cast<float*>(c); p.h = const_cast<float*>(h); p.dxp = dxp;
    p.xchg = (u64*)xchg; p.err = err; p.N = N; p.S = S;
    hipLaunchKernelGGL(gru256_cluster_bwd_k, dim3(gru256_grid(N)), dim3(256), 0, st, p);
    TACO_RETURN_LAST();
}
