// Persistent CLUSTER kernels for the two residual decoder GRU(256) recurrences (reference
// models/tacotron.py:76-80: MultiRNNCell[..., ResidualWrapper(GRUCell(256)), ResidualWrapper(GRUCell(256))]).
//
// Why: a launch per dependent stage costs ~6 us on MI355X (kernel boundary + cold L2 on 8 XCDs), and a GRU
// step has two dependent mat-vec stages -> 13 us/step.  Here the whole S-step recurrence is ONE launch.
// W_h (768 KB fp32) does not fit one CU's register file, so a CLUSTER of 4 workgroups (512 threads each) shares two
// batch rows: workgroup w owns hidden indices J_w = [64w, 64w+64): the r/u/c columns J_w of the recurrent kernels
// stay in its registers (96 VGPRs/lane, K split over 4 / 8 adjacent lanes, DPP cross-lane sums) for the whole
// kernel.  Per step the cluster all-gathers r*h (before the candidate product) and h' (before the next gate
// product): 2 x 256 floats per row, moved as 8-byte {epoch, value} granules written with ONE agent-scope store each
// and polled with agent-scope loads (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; no
// fences, placement independent; members share an XCD under round-robin placement for speed only).
// Different clusters never communicate.  Grid = 4 * ceil(N/2) workgroups, all co-resident (<= 256 CUs);
// every spin is bounded and reports through an error word instead of hanging the GPU.
//
// GRU semantics: tf.contrib.rnn.GRUCell, input projection hoisted (xp = x.Wx + b).
#include "common.hpp"
#include "xcd_granule.hpp"

#define HD 256
#define GT2 512
#define SPIN_LIMIT (1 << 22)
#define QIDX(k) ((k) + ((k) >> 5) * 4)       // LDS vectors: 4 pad floats after every 32
#define QLEN(n) ((n) + ((n) >> 5) * 4)

// agent-scope granule, or the L2-local form once the cluster has verified that its 4 members share an XCD (xcd_granule.hpp)
__device__ __forceinline__ void put_gr(u64* p, unsigned epoch, float v, bool local) {
    if (local) put_granule_xcd(p, epoch, v); else put_granule(p, epoch, v);
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

static int gru_xcd_local_allowed() { const char* e = getenv("TACO_XCD_LOCAL"); return (e && e[0] == '0') ? 0 : 1; }

struct Gru256 {
    const float* xp;      // [N,S,768] hoisted input projection (+bias): r | u | c
    const float* whg;     // [256,512]
    const float* whc;     // [256,256]
    const float* res;     // [N,S,256] residual input (or null)
    float *r, *u, *c, *rh, *h, *d;     // [N,S,256] saved gates / r*h_prev / state / residual output
    u64* xchg;            // granule regions (zeroed before launch)
    int* err;
    int N, S;
    int s0, s1;           // step range [s0, s1) of this launch (chunked pipelining); state enters / leaves through h / carry
    int xcd_local;        // 1: clusters whose members share an XCD use the L2-local granule form
    // backward
    const float* dout;    // [N,S,256]
    float* dxp;           // [N,S,768]
    float* carry;         // [N,256] recurrent part of dh carried between chunk launches (dout is added by the consumer)
};

// gather one value per row from each of the 3 peers' 64-slices of a [2][256] granule region into QIDX LDS vectors
__device__ __forceinline__ void gather256(const u64* region, float* l0, float* l1, int off, int w, unsigned epoch, int tid, int* err) {
    asm volatile("" : "+v"(tid));          // opaque: no per-thread granule pointer kept live across the step loop
    if (tid < 384) {
        const int row = tid / 192, rem = tid - row * 192;
        const int peer = rem >> 6, jj = rem & 63;
        const int pw = peer + (peer >= w ? 1 : 0);
        const int j = 64 * pw + jj;
        const u64* const ptr[1] = {region + row * HD + j};
        float v[1];
        get_granules<1>(ptr, epoch, v, err);
        (row ? l1 : l0)[QIDX(off + j)] = v[0];
    }
}

__global__ __launch_bounds__(GT2, 2) void gru256_cluster_fwd_k(Gru256 p) {
    int tid = threadIdx.x;
    // cluster members share blockIdx % 8 (same XCD under round-robin placement: speed only, never correctness)
    const int nclus = gridDim.x / 4;
    int w, cl;
    if ((nclus & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; w = q & 3; cl = (q >> 2) * 8 + xcd; }
    else { w = blockIdx.x & 3; cl = blockIdx.x >> 2; }
    const int row0 = cl * 2;
    __shared__ __attribute__((aligned(16))) float h_l[2][QLEN(HD)];
    __shared__ __attribute__((aligned(16))) float rh_l[2][QLEN(HD)];
    __shared__ float u_l[2][64];

    // gates: thread (gcol = tid>>2 in [0,128), kq = tid&3): k in [64kq, +64), column r_{64w+gcol} or u_{64w+gcol-64}
    const int gcol = tid >> 2, kq = tid & 3;
    const int gc = gcol < 64 ? 64 * w + gcol : 256 + 64 * w + (gcol - 64);
    float wg[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wg[k] = p.whg[(long)(kq * 64 + k) * 512 + gc];
    // candidate: thread (ccol = tid>>3 in [0,64), k8 = tid&7): k in [32k8, +32), column c_{64w+ccol}
    const int ccol = tid >> 3, k8 = tid & 7;
    float wc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) wc[k] = p.whc[(long)(k8 * 32 + k) * 256 + 64 * w + ccol];

    const bool ok0 = row0 < p.N, ok1 = row0 + 1 < p.N;
    const unsigned rb0 = (unsigned)(min(row0, p.N - 1) * p.S), rb1 = (unsigned)(min(row0 + 1, p.N - 1) * p.S);
    for (int i = tid; i < 2 * QLEN(HD); i += GT2) { (&h_l[0][0])[i] = 0.0f; (&rh_l[0][0])[i] = 0.0f; }
    __syncthreads();
    if (p.s0 > 0 && tid < HD) {                           // state of the previous chunk
        h_l[0][QIDX(tid)] = p.h[(rb0 + p.s0 - 1) * 256u + tid];
        h_l[1][QIDX(tid)] = p.h[(rb1 + p.s0 - 1) * 256u + tid];
    }
    __syncthreads();
    u64* xr = p.xchg + ((long)cl * 2) * HD;                               // rh granules  [2][256]
    u64* xh = p.xchg + ((long)nclus * 2 + (long)cl * 2) * HD;             // h' granules
    __shared__ int local_s;
    const bool local = cluster_shares_xcd(p.xchg + (long)nclus * 4 * HD + (long)cl * 4, w, 4, p.err, &local_s, tid, p.xcd_local, (unsigned)p.s0);
    const int j_own = 64 * w + (gcol & 63);                               // hidden index of this thread's gate column
    const int jc = 64 * w + ccol;                                         // hidden index of the candidate column

    // input projections are fetched one step ahead
    float nxg0 = 0.f, nxg1 = 0.f, nxc0 = 0.f, nxc1 = 0.f;
    if (kq == 0) { nxg0 = p.xp[(rb0 + p.s0) * 768u + gc]; nxg1 = p.xp[(rb1 + p.s0) * 768u + gc]; }
    if (k8 == 0) { nxc0 = p.xp[(rb0 + p.s0) * 768u + 512u + jc]; nxc1 = p.xp[(rb1 + p.s0) * 768u + 512u + jc]; }

    for (int s = p.s0; s < p.s1; ++s) {
        const unsigned epoch = (unsigned)s + 1;          // step of the PASS: the chunk launches of a pass share one zero-filled buffer
        unsigned o0 = rb0 + s, o1 = rb1 + s;
        asm volatile("" : "+v"(o0), "+v"(o1));           // opaque: store addresses are formed at the point of use
        const float xg0 = nxg0, xg1 = nxg1, xc0 = nxc0, xc1 = nxc1;
        if (s + 1 < p.s1) {
            if (kq == 0) { nxg0 = p.xp[(o0 + 1u) * 768u + gc]; nxg1 = p.xp[(o1 + 1u) * 768u + gc]; }
            if (k8 == 0) { nxc0 = p.xp[(o0 + 1u) * 768u + 512u + jc]; nxc1 = p.xp[(o1 + 1u) * 768u + 512u + jc]; }
        }
        // ---- gates (both rows), K quarter per lane
        float a0, a1;
        {
            f2 a0p = {0.0f, 0.0f}, a1p = {0.0f, 0.0f};
            const float* h0 = &h_l[0][QIDX(kq * 64)];
            const float* h1 = &h_l[1][QIDX(kq * 64)];
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
                if (k4 && (k4 & 3) == 0) asm volatile("" ::: "memory");
                const float4 v0 = *reinterpret_cast<const float4*>(h0 + QIDX(k4 * 4));
                const float4 v1 = *reinterpret_cast<const float4*>(h1 + QIDX(k4 * 4));
                pk_dot4x2(v0, v1, wg[k4 * 4], wg[k4 * 4 + 1], wg[k4 * 4 + 2], wg[k4 * 4 + 3], a0p, a1p);
            }
            a0 = a0p.x + a0p.y; a1 = a1p.x + a1p.y;
        }
        a0 = group_sum<4>(a0);
        a1 = group_sum<4>(a1);
        if (kq == 0) {
            const float g0 = fast_sigmoid(a0 + xg0);
            const float g1 = fast_sigmoid(a1 + xg1);
            if (gcol < 64) {
                const float q0 = g0 * h_l[0][QIDX(j_own)], q1 = g1 * h_l[1][QIDX(j_own)];
                rh_l[0][QIDX(j_own)] = q0; rh_l[1][QIDX(j_own)] = q1;
                put_gr(xr + j_own, epoch, q0, local);
                put_gr(xr + HD + j_own, epoch, q1, local);
                if (ok0) { p.r[o0 * 256u + j_own] = g0; p.rh[o0 * 256u + j_own] = q0; }
                if (ok1) { p.r[o1 * 256u + j_own] = g1; p.rh[o1 * 256u + j_own] = q1; }
            } else {
                u_l[0][gcol - 64] = g0; u_l[1][gcol - 64] = g1;
                if (ok0) p.u[o0 * 256u + j_own] = g0;
                if (ok1) p.u[o1 * 256u + j_own] = g1;
            }
        }
        gather256(xr, rh_l[0], rh_l[1], 0, w, epoch, tid, p.err);
        lds_barrier();
        // ---- candidate + state for hidden index jc (both rows), K eighth per lane
        float c0, c1;
        {
            f2 c0p = {0.0f, 0.0f}, c1p = {0.0f, 0.0f};
            const float* q0p = &rh_l[0][QIDX(k8 * 32)];
            const float* q1p = &rh_l[1][QIDX(k8 * 32)];
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const float4 v0 = *reinterpret_cast<const float4*>(q0p + k4 * 4);
                const float4 v1 = *reinterpret_cast<const float4*>(q1p + k4 * 4);
                pk_dot4x2(v0, v1, wc[k4 * 4], wc[k4 * 4 + 1], wc[k4 * 4 + 2], wc[k4 * 4 + 3], c0p, c1p);
            }
            c0 = c0p.x + c0p.y; c1 = c1p.x + c1p.y;
        }
        c0 = group_sum<8>(c0); c1 = group_sum<8>(c1);
        if (k8 == 0) {
            const float cc0 = fast_tanh(c0 + xc0);
            const float cc1 = fast_tanh(c1 + xc1);
            const float u0 = u_l[0][ccol], u1 = u_l[1][ccol];
            const float hn0 = u0 * h_l[0][QIDX(jc)] + (1.0f - u0) * cc0;
            const float hn1 = u1 * h_l[1][QIDX(jc)] + (1.0f - u1) * cc1;
            h_l[0][QIDX(jc)] = hn0; h_l[1][QIDX(jc)] = hn1;
            put_gr(xh + jc, epoch, hn0, local);
            put_gr(xh + HD + jc, epoch, hn1, local);
            if (ok0) {
                p.c[o0 * 256u + jc] = cc0; p.h[o0 * 256u + jc] = hn0;
                if (p.d) p.d[o0 * 256u + jc] = p.res[o0 * 256u + jc] + hn0;
            }
            if (ok1) {
                p.c[o1 * 256u + jc] = cc1; p.h[o1 * 256u + jc] = hn1;
                if (p.d) p.d[o1 * 256u + jc] = p.res[o1 * 256u + jc] + hn1;
            }
        }
        gather256(xh, h_l[0], h_l[1], 0, w, epoch, tid, p.err);
        lds_barrier();
    }
}

// BPTT twin.  Workgroup w owns hidden indices k in J_w: it keeps ROWS J_w of Whc ([64,256]) and Whg ([64,512]) in
// registers (8 lanes per row); per step the cluster all-gathers dcp (candidate pre-activation gradient, 256/row) and
// dg (gate pre-activation gradients, 512/row).  Hidden index k = 64w + kk is owned by lane 8*kk (j8 == 0).
__global__ __launch_bounds__(GT2, 2) void gru256_cluster_bwd_k(Gru256 p) {
    int tid = threadIdx.x;
    const int nclus = gridDim.x / 4;
    int w, cl;
    if ((nclus & 7) == 0) { const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3; w = q & 3; cl = (q >> 2) * 8 + xcd; }
    else { w = blockIdx.x & 3; cl = blockIdx.x >> 2; }
    const int row0 = cl * 2;
    __shared__ __attribute__((aligned(16))) float dx_l[2][QLEN(3 * HD)];     // dg_r (256) | dg_u (256) | dcp (256)

    const int kk = tid >> 3, j8 = tid & 7;
    const int k_own = 64 * w + kk;
    float wcT[32];     // Whc[k_own][32 j8 .. +32)
    float wgT[64];     // Whg[k_own][64 j8 .. +64)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(p.whc + (long)k_own * 256 + j8 * 32 + q * 4);
        wcT[q * 4] = a.x; wcT[q * 4 + 1] = a.y; wcT[q * 4 + 2] = a.z; wcT[q * 4 + 3] = a.w;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(p.whg + (long)k_own * 512 + j8 * 64 + q * 4);
        wgT[q * 4] = a.x; wgT[q * 4 + 1] = a.y; wgT[q * 4 + 2] = a.z; wgT[q * 4 + 3] = a.w;
    }
    const bool ok[2] = {row0 < p.N, row0 + 1 < p.N};
    const unsigned rb[2] = {(unsigned)(min(row0, p.N - 1) * p.S), (unsigned)(min(row0 + 1, p.N - 1) * p.S)};
    u64* xc = p.xchg + ((long)cl * 2) * HD;                                  // dcp granules [2][256]
    u64* xgr = p.xchg + ((long)nclus * 2) * HD + ((long)cl * 2) * HD;        // dg_r granules [2][256]
    u64* xgu = p.xchg + ((long)nclus * 4) * HD + ((long)cl * 2) * HD;        // dg_u granules [2][256]
    __shared__ int local_s;
    const bool local = cluster_shares_xcd(p.xchg + (long)nclus * 6 * HD + (long)cl * 4, w, 4, p.err, &local_s, tid, p.xcd_local, (unsigned)(p.S - p.s1));
    const bool owner = j8 == 0;

    float dhT[2] = {0.f, 0.f}, dhn[2] = {0.f, 0.f};
    // saved activations of the step are fetched one step ahead
    float pr[2] = {0, 0}, pu[2] = {0, 0}, pc[2] = {0, 0}, ph[2] = {0, 0}, pdo[2] = {0, 0};
    if (owner) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned o = rb[b] + p.s1 - 1;
            // gradient wrt h_{s1-1}: external part (read now: a pipelined producer finishes it just before this launch)
            // plus the recurrent part handed over by the launch of the later chunk
            dhT[b] = p.dout[o * 256u + k_own] + (p.s1 == p.S ? 0.f : p.carry[(unsigned)min(row0 + b, p.N - 1) * 256u + k_own]);
            pr[b] = p.r[o * 256u + k_own]; pu[b] = p.u[o * 256u + k_own]; pc[b] = p.c[o * 256u + k_own];
            ph[b] = p.s1 > 1 ? p.h[(o - 1u) * 256u + k_own] : 0.f;
            pdo[b] = p.s1 > 1 ? p.dout[(o - 1u) * 256u + k_own] : 0.f;
        }
    }

    for (int s = p.s1 - 1; s >= p.s0; --s) {
        const unsigned epoch = (unsigned)(p.S - s);
        unsigned o[2] = {rb[0] + s, rb[1] + s};
        asm volatile("" : "+v"(o[0]), "+v"(o[1]));
        float r_[2], u_[2], c_[2], hp_[2], don[2], du[2] = {0.f, 0.f}, dhd[2] = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b) { r_[b] = pr[b]; u_[b] = pu[b]; c_[b] = pc[b]; hp_[b] = ph[b]; don[b] = pdo[b]; }
        // ---- elementwise at the owner: dcp, du, direct dh
        if (owner) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                du[b] = dhT[b] * (hp_[b] - c_[b]);
                dhd[b] = dhT[b] * u_[b];
                const float dcp = dhT[b] * (1.0f - u_[b]) * (1.0f - c_[b] * c_[b]);
                dx_l[b][QIDX(512 + k_own)] = dcp;
                put_gr(xc + b * HD + k_own, epoch, dcp, local);
                if (ok[b]) p.dxp[o[b] * 768u + 512u + k_own] = dcp;
            }
            if (s > p.s0) {                                // prefetch step s-1
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const unsigned on = o[b] - 1u;
                    pr[b] = p.r[on * 256u + k_own]; pu[b] = p.u[on * 256u + k_own]; pc[b] = p.c[on * 256u + k_own];
                    ph[b] = s > 1 ? p.h[(on - 1u) * 256u + k_own] : 0.f;
                    pdo[b] = s > 1 ? p.dout[(on - 1u) * 256u + k_own] : 0.f;
                }
            }
        }
        gather256(xc, dx_l[0], dx_l[1], 512, w, epoch, tid, p.err);
        lds_barrier();
        // ---- drh[k_own] = sum_j dcp[j] * Whc[k_own][j]  (eighth per lane, both rows)
        float d0, d1;
        {
            f2 d0p = {0.0f, 0.0f}, d1p = {0.0f, 0.0f};
            const float* a0 = &dx_l[0][QIDX(512 + j8 * 32)];
            const float* a1 = &dx_l[1][QIDX(512 + j8 * 32)];
#pragma unroll
            for (int j4 = 0; j4 < 8; ++j4) {
                const float4 v0 = *reinterpret_cast<const float4*>(a0 + j4 * 4);
                const float4 v1 = *reinterpret_cast<const float4*>(a1 + j4 * 4);
                pk_dot4x2(v0, v1, wcT[j4 * 4], wcT[j4 * 4 + 1], wcT[j4 * 4 + 2], wcT[j4 * 4 + 3], d0p, d1p);
            }
            d0 = d0p.x + d0p.y; d1 = d1p.x + d1p.y;
        }
        d0 = group_sum<8>(d0); d1 = group_sum<8>(d1);
        float dhp[2] = {0.f, 0.f};
        if (owner) {
            const float drh[2] = {d0, d1};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float dgr = drh[b] * hp_[b] * r_[b] * (1.0f - r_[b]);
                const float dgu = du[b] * u_[b] * (1.0f - u_[b]);
                dhp[b] = dhd[b] + drh[b] * r_[b];
                dx_l[b][QIDX(k_own)] = dgr;
                dx_l[b][QIDX(HD + k_own)] = dgu;
                put_gr(xgr + b * HD + k_own, epoch, dgr, local);
                put_gr(xgu + b * HD + k_own, epoch, dgu, local);
                if (ok[b]) { p.dxp[o[b] * 768u + k_own] = dgr; p.dxp[o[b] * 768u + 256u + k_own] = dgu; }
            }
        }
        {   // gather the peers' dg_r and dg_u with one poll round trip (2 granules per thread)
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            if (t2 < 384) {
                const int row = t2 / 192, rem = t2 - row * 192;
                const int peer = rem >> 6, jj = rem & 63;
                const int pw = peer + (peer >= w ? 1 : 0);
                const int j = 64 * pw + jj;
                const u64* const ptr[2] = {xgr + row * HD + j, xgu + row * HD + j};
                float v[2];
                get_granules<2>(ptr, epoch, v, p.err);
                dx_l[row][QIDX(j)] = v[0];
                dx_l[row][QIDX(HD + j)] = v[1];
            }
        }
        lds_barrier();
        // ---- dh_{s-1}[k_own] = dhp + sum_j dg[j] * Whg[k_own][j]   (j over 512, eighth of 64 per lane)
        float e0, e1;
        {
            f2 e0p = {0.0f, 0.0f}, e1p = {0.0f, 0.0f};
            const float* a0 = &dx_l[0][QIDX(j8 * 64)];
            const float* a1 = &dx_l[1][QIDX(j8 * 64)];
#pragma unroll
            for (int j4 = 0; j4 < 16; ++j4) {
                if (j4 && (j4 & 3) == 0) asm volatile("" ::: "memory");
                const float4 v0 = *reinterpret_cast<const float4*>(a0 + QIDX(j4 * 4));
                const float4 v1 = *reinterpret_cast<const float4*>(a1 + QIDX(j4 * 4));
                pk_dot4x2(v0, v1, wgT[j4 * 4], wgT[j4 * 4 + 1], wgT[j4 * 4 + 2], wgT[j4 * 4 + 3], e0p, e1p);
            }
            e0 = e0p.x + e0p.y; e1 = e1p.x + e1p.y;
        }
        e0 = group_sum<8>(e0); e1 = group_sum<8>(e1);
        if (owner) { dhn[0] = dhp[0] + e0; dhn[1] = dhp[1] + e1; dhT[0] = dhn[0] + don[0]; dhT[1] = dhn[1] + don[1]; }
        // no trailing barrier: the dcp slots rewritten at the top of the next iteration were last read before this
        // iteration's second barrier, and the gate-gradient slots are rewritten only after the next barrier
    }
    if (p.s0 > 0 && owner) {                               // recurrent part of the gradient wrt h_{s0-1} for the next launch
        if (ok[0]) p.carry[(unsigned)row0 * 256u + k_own] = dhn[0];
        if (ok[1]) p.carry[(unsigned)(row0 + 1) * 256u + k_own] = dhn[1];
    }
}

static int gru256_grid(int N) { return 4 * ((N + 1) / 2); }
// Isolation pad (isolate_lds_bytes, chosen by the caller): unused dynamic LDS bytes per workgroup.  The kernels need 6-7 KB of LDS, so
// up to three GEMM workgroups (48 KB each: fp32 MFMA streams that hold the SIMDs for 64 cycles per instruction) can share a CU with a GRU
// workgroup and stretch its steps 1.3-3x whenever projection / conv / weight-gradient GEMMs are in flight.  With 150 KB of pad a GRU
// workgroup owns its CU, like the attention workgroups do by their register footprint.  Pays while every persistent workgroup of the
// decoder pipeline still gets a CU of its own (attention 4 N + two GRUs 2 N each <= 256, i.e. N <= 32: C2 7.72 -> 7.65 ms, C4 5.91 -> 5.86,
// C5 12.28 -> 12.22) and nothing else needs those CUs: the host (engine.py) passes 0 for larger batches and under data parallelism, where
// the RCCL kernels must find room beside the GRU workgroups.
static size_t gru256_lds_pad(const void* k, int bytes, DevMask& done) {
    if (bytes <= 0) return 0;
    if (ensure_dyn_lds(k, 155000, done) != TACO_OK) return 0;
    return (size_t)(bytes > 155000 ? 155000 : bytes);
}

extern "C" int taco_gru256_seq_fwd(const float* xp, const float* whg, const float* whc, const float* res, float* r, float* u,
                                   float* c, float* rh, float* h, float* d, void* xchg, int* err, int N, int S,
                                   int s0, int s1, int isolate_lds_bytes, hipStream_t st) {
    if (!xp || !whg || !whc || !r || !u || !c || !rh || !h || !xchg || !err || N <= 0 || S <= 0) return TACO_EINVAL;
    if (s0 < 0 || s1 > S || s0 >= s1) return TACO_EINVAL;
    if (d && !res) return TACO_EINVAL;
    if (gru256_grid(N) > 256) return TACO_EINVAL;           // all workgroups must be co-resident
    const int nclus = (N + 1) / 2;
    // granule epochs count the steps of the whole pass: only the pass's first chunk launch needs a zero-filled buffer
    if (s0 == 0 && hipMemsetAsync(xchg, 0, (size_t)nclus * (2 * 2 * HD + 4) * sizeof(u64), st) != hipSuccess) return TACO_EINVAL;     // + placement granules
    Gru256 p{};
    p.xp = xp; p.whg = whg; p.whc = whc; p.res = res; p.r = r; p.u = u; p.c = c; p.rh = rh; p.h = h; p.d = d;
    p.xchg = (u64*)xchg; p.err = err; p.N = N; p.S = S; p.s0 = s0; p.s1 = s1; p.xcd_local = gru_xcd_local_allowed();
    static DevMask attr{0};
    hipLaunchKernelGGL(gru256_cluster_fwd_k, dim3(gru256_grid(N)), dim3(GT2), gru256_lds_pad((const void*)gru256_cluster_fwd_k, isolate_lds_bytes, attr), st, p);
    TACO_RETURN_LAST();
}

extern "C" int taco_gru256_seq_bwd(const float* dout, const float* whg, const float* whc, const float* r, const float* u,
                                   const float* c, const float* h, float* dxp, float* carry, void* xchg, int* err, int N,
                                   int S, int s0, int s1, int isolate_lds_bytes, hipStream_t st) {
    if (!dout || !whg || !whc || !r || !u || !c || !h || !dxp || !carry || !xchg || !err || N <= 0 || S <= 0) return TACO_EINVAL;
    if (s0 < 0 || s1 > S || s0 >= s1) return TACO_EINVAL;
    if (gru256_grid(N) > 256) return TACO_EINVAL;
    const int nclus = (N + 1) / 2;
    if (s1 == S && hipMemsetAsync(xchg, 0, (size_t)nclus * (2 * 3 * HD + 4) * sizeof(u64), st) != hipSuccess) return TACO_EINVAL;     // + placement granules
    Gru256 p{};
    p.dout = dout; p.whg = whg; p.whc = whc; p.r = const_cast<float*>(r); p.u = const_cast<float*>(u);
    p.c = const_cast<float*>(c); p.h = const_cast<float*>(h); p.dxp = dxp;
    p.xchg = (u64*)xchg; p.err = err; p.N = N; p.S = S; p.s0 = s0; p.s1 = s1; p.carry = carry; p.xcd_local = gru_xcd_local_allowed();
    static DevMask attr{0};
    hipLaunchKernelGGL(gru256_cluster_bwd_k, dim3(gru256_grid(N)), dim3(GT2), gru256_lds_pad((const void*)gru256_cluster_bwd_k, isolate_lds_bytes, attr), st, p);
    TACO_RETURN_LAST();
}
