// Persistent GRU(128) sequence kernels for gfx950: the encoder / post-net bidirectional GRUs of the CBHG
// (reference models/modules.py:68-74: tf.nn.bidirectional_dynamic_rnn(GRUCell(128), GRUCell(128), ...)).
//
// Design (MI355X-first): the recurrence is latency-bound (T = 128..800 dependent steps), so instead of a
// launch per step ONE workgroup owns ONE batch row of one direction for the whole sequence and never talks to
// another workgroup (2N workgroups: 64 CUs at N = 32).  The recurrent weights W_h ([128,256] gates + [128,128]
// candidate = 192 KB fp32) live in the workgroup's REGISTER FILE for the entire kernel: 512 threads, each column's
// K axis split over 2 (gates) / 4 (candidate) adjacent lanes = 96 weight VGPRs per lane, two waves per SIMD so the
// fp32 VALU issues every cycle; partial dot products are combined with cross-lane adds (no LDS round trip).  The
// hidden state is broadcast through a padded LDS vector (the 2 / 4 distinct addresses of a ds_read_b128 fall on
// different banks).  The input half of the GRU matmul (x.W_x + b) is hoisted out of the loop as one MFMA GEMM over
// all time steps (gemm.hip).
//
// GRU semantics (tf.contrib.rnn.GRUCell):  [r,u] = sigmoid([x,h].Wg + bg);
// c = tanh([x, r*h].Wc + bc);  h' = u*h + (1-u)*c.   Sequence lengths (Appendix A.6): for t >= len the
// output row is zero and the state is carried; the backward direction runs t = len-1 .. 0 from zero state.
#include "common.hpp"

#define H 128
// 4-term slice of a dot product as two v_pk_fma_f32 (packed fp32 issues two FMAs per lane per instruction)
__device__ __forceinline__ void pk_dot4(const float4& v, const f2* w2, f2& acc) {
    acc = __builtin_elementwise_fma((f2){v.x, v.y}, w2[0], acc);
    acc = __builtin_elementwise_fma((f2){v.z, v.w}, w2[1], acc);
}
#define GT 512
#define LIDX(k) ((k) + ((k) >> 5) * 4)      // 4 pad floats after every 32
#define LLEN(n) ((n) + ((n) >> 5) * 4)

struct GruSeq {
    const float* xp;      // [N,T,ldxp] hoisted input projections (+bias): per direction [r(128) u(128) c(128)]
    int ldxp;             // row stride (floats); direction d uses columns [d*384, d*384+384)
    const float* wg[2];   // per direction: recurrent gate weights   [128,256] (rows = h index)
    const float* wc[2];   // per direction: recurrent candidate weights [128,128]
    const int* lengths;   // [N] or nullptr (full length)
    float* out;           // [N,T,ldo]; direction d writes columns [d*128, d*128+128)
    int ldo;
    float* ruc;           // [ndir,N,T,384] saved r,u,c (forward) / read (backward)
    int N, T;
    // backward only
    const float* dout;    // [N,T,lddo] gradient wrt out (direction d at column offset d*128)
    int lddo;
    float* dxp;           // [N,T,ldxp]   gradient wrt xp (same layout as xp)
    float* hp;            // [ndir,N,T,128] h_{prev} per step  (for dW_h gates  = hp^T . dxp[:, 0:256])
    float* rh;            // [ndir,N,T,128] r*h_{prev}         (for dW_h cand   = rh^T . dxp[:, 256:384])
    // step range of this launch (processing order: step s touches frame s of the forward direction and frame T-1-s of the backward
    // direction in the forward pass, the mirror image in BPTT): the recurrence may be cut into chunk launches issued in order, so
    // that work on the frames a chunk completes can start while the next chunk runs; the state travels through `state`
    int s0, s1;
    float* state;         // [ndir,N,128]: h after step s1-1 (forward) / dh after step s1-1 (BPTT); read when s0 > 0, written when s1 < T
};

// NTH threads per workgroup: 512 (2 waves per SIMD: K of a gate column split over 2 lanes, of a candidate column over 4; default)
// or 256 (one wave per SIMD: a gate column's whole K in one lane; 192 weight VGPRs per lane).  Measured at C2: the 256-thread form
// takes 1.0 us per step against 0.76 us -- the step is bound by the dependent chain of a phase (LDS read -> FMA chain -> reduce ->
// gate -> LDS write -> barrier), which gets LONGER with fewer, fatter lanes, not by instruction issue.  Two accumulation chains per
// lane (below) shorten it a little (-26 us per training step).
template <int NTH>
__global__ __launch_bounds__(NTH, NTH / 256) void gru128_seq_fwd_k(GruSeq p) {
    constexpr int KSG = NTH / 256, KSC = NTH / 128;       // lanes per gate / candidate column
    constexpr int KG = H / KSG, KC = H / KSC;             // K terms per lane
    const int tid = threadIdx.x;
    const int row = blockIdx.x, dir = blockIdx.y;
    __shared__ __attribute__((aligned(16))) float h_l[LLEN(H)];
    __shared__ __attribute__((aligned(16))) float rh_l[LLEN(H)];
    __shared__ float u_l[H];

    const int gcol = tid / KSG, kh = tid % KSG;     // gates: 256 columns x KSG K-parts
    const int ccol = tid / KSC, kq = tid % KSC;     // candidate: 128 columns x KSC K-parts
    f2 wg[KG / 2], wc[KC / 2];
    {
        const float* Wg = p.wg[dir];
        const float* Wc = p.wc[dir];
#pragma unroll
        for (int k = 0; k < KG / 2; ++k) wg[k] = (f2){Wg[(kh * KG + 2 * k) * 256 + gcol], Wg[(kh * KG + 2 * k + 1) * 256 + gcol]};
#pragma unroll
        for (int k = 0; k < KC / 2; ++k) wc[k] = (f2){Wc[(kq * KC + 2 * k) * H + ccol], Wc[(kq * KC + 2 * k + 1) * H + ccol]};
    }
    const int len = p.lengths ? min(max(p.lengths[row], 0), p.T) : p.T;
    float* hst = p.state ? p.state + ((long)dir * p.N + row) * H : nullptr;
    for (int i = tid; i < LLEN(H); i += NTH) { h_l[i] = 0.0f; rh_l[i] = 0.0f; }
    __syncthreads();
    if (p.s0 > 0 && tid < H) h_l[LIDX(tid)] = hst[tid];          // state left by the previous chunk launch
    __syncthreads();

    const int xoff = dir * 3 * H;
    const float* xrow = p.xp + (long)row * p.T * p.ldxp + xoff;
    float* ruc = p.ruc + ((long)dir * p.N + row) * p.T * 3 * H;
    float* orow = p.out + (long)row * p.T * p.ldo + dir * H;

    // The step loop is BRANCH-FREE: every lane of a column's K-group computes the (identical) reduced sum, the activation and the
    // stores -- a wave instruction costs the same with one active lane or 64, and duplicate stores of equal values are harmless.
    // What it buys: with the stores inside `if (owner)` blocks the compiler could not count the memory operations behind the
    // one-step-ahead prefetch loads and closed every step with `s_waitcnt vmcnt(0)`, i.e. waited for the write acknowledgements of
    // the step's saved activations (round 3: 0.79 -> 0.6x us per step).
    auto tstep = [&](int s) { return dir == 0 ? s : p.T - 1 - s; };
    // gate lanes: r columns write r*h into rh_l, u columns write u into u_l; one address computation, no branch
    const bool is_r = gcol < H;
    float* gdst = is_r ? (rh_l + LIDX(gcol)) : (u_l + (gcol - H));
    const float* hsrc = h_l + LIDX(gcol & (H - 1));
    float xg, xc;
    {
        const int t = tstep(p.s0);
        xg = xrow[(long)t * p.ldxp + gcol];
        xc = xrow[(long)t * p.ldxp + 2 * H + ccol];
    }
    for (int s = p.s0; s < p.s1; ++s) {
        const int t = tstep(s);
        const float ag = xg, ac = xc;
        {                                               // prefetch the next step's input projection (the last step re-reads its own)
            const int tn = tstep(min(s + 1, p.s1 - 1));
            xg = xrow[(long)tn * p.ldxp + gcol];
            xc = xrow[(long)tn * p.ldxp + 2 * H + ccol];
        }
        // ---- gates
        float a;
        {
            f2 a2 = {0.0f, 0.0f}, a3 = {0.0f, 0.0f};          // two accumulation chains
#pragma unroll
            for (int k4 = 0; k4 < KG / 4; k4 += 2) {
                pk_dot4(*reinterpret_cast<const float4*>(h_l + LIDX(kh * KG + k4 * 4)), wg + 2 * k4, a2);
                pk_dot4(*reinterpret_cast<const float4*>(h_l + LIDX(kh * KG + k4 * 4 + 4)), wg + 2 * k4 + 2, a3);
            }
            a = (a2.x + a2.y) + (a3.x + a3.y);
        }
        a = group_sum<KSG>(a);                          // every lane of the group holds the sum
        {
            const float g = fast_sigmoid(a + ag);
            ruc[(long)t * 3 * H + gcol] = g;
            *gdst = is_r ? g * *hsrc : g;
        }
        lds_barrier();
        // ---- candidate + state update
        float b;
        {
            f2 b2 = {0.0f, 0.0f}, b3 = {0.0f, 0.0f};
#pragma unroll
            for (int k4 = 0; k4 < KC / 4; k4 += 2) {
                pk_dot4(*reinterpret_cast<const float4*>(rh_l + LIDX(kq * KC + k4 * 4)), wc + 2 * k4, b2);
                pk_dot4(*reinterpret_cast<const float4*>(rh_l + LIDX(kq * KC + k4 * 4 + 4)), wc + 2 * k4 + 2, b3);
            }
            b = (b2.x + b2.y) + (b3.x + b3.y);
        }
        b = group_sum<KSC>(b);
        {
            const float c = fast_tanh(b + ac);
            const float hprev = h_l[LIDX(ccol)];
            const float u = u_l[ccol];
            const float hn = u * hprev + (1.0f - u) * c;
            const bool valid = t < len;
            ruc[(long)t * 3 * H + 2 * H + ccol] = c;
            orow[(long)t * p.ldo + ccol] = valid ? hn : 0.0f;
            // the four lanes of a column sit in one wave: its read of hprev above is an earlier LDS instruction than this write
            h_l[LIDX(ccol)] = valid ? hn : hprev;
        }
        lds_barrier();
    }
    if (p.s1 < p.T && tid < H) hst[tid] = h_l[LIDX(tid)];        // for the next chunk launch
}

// BPTT twin.  Processing order is the reverse of the forward order of that direction.  Hidden index k is owned by
// lane 4k (the jq == 0 lane of its 4-lane group); transposed weight rows Wc[k][:], Wg[k][:] are split over the group.
template <int NTH>
__global__ __launch_bounds__(NTH, NTH / 256) void gru128_seq_bwd_k(GruSeq p) {
    constexpr int JS = NTH / 128;                   // lanes per hidden index (4 or 2)
    constexpr int JC = H / JS, JG = 2 * H / JS;     // reduction terms per lane: candidate / gate product
    const int tid = threadIdx.x;
    const int row = blockIdx.x, dir = blockIdx.y;
    const int k = tid / JS, jq = tid % JS;
    __shared__ __attribute__((aligned(16))) float dcp_l[LLEN(H)];
    __shared__ __attribute__((aligned(16))) float dg_l[LLEN(2 * H)];

    f2 wcT[JC / 2], wgT[JG / 2];
    {
        const float* Wg = p.wg[dir];
        const float* Wc = p.wc[dir];
#pragma unroll
        for (int q = 0; q < JC / 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(Wc + k * H + jq * JC + q * 4);
            wcT[q * 2] = (f2){a.x, a.y}; wcT[q * 2 + 1] = (f2){a.z, a.w};
        }
#pragma unroll
        for (int q = 0; q < JG / 4; ++q) {
            const float4 g = *reinterpret_cast<const float4*>(Wg + k * 256 + jq * JG + q * 4);
            wgT[q * 2] = (f2){g.x, g.y}; wgT[q * 2 + 1] = (f2){g.z, g.w};
        }
    }
    const int len = p.lengths ? min(max(p.lengths[row], 0), p.T) : p.T;
    const int xoff = dir * 3 * H;
    const long dn = (long)dir * p.N + row;
    const float* ruc = p.ruc + dn * p.T * 3 * H;
    const float* orow = p.out + (long)row * p.T * p.ldo + dir * H;
    const float* dorow = p.dout + (long)row * p.T * p.lddo + dir * H;
    float* dxrow = p.dxp + (long)row * p.T * p.ldxp + xoff;
    float* hprow = p.hp + dn * p.T * H;
    float* rhrow = p.rh + dn * p.T * H;
    float* dst = p.state ? p.state + dn * H : nullptr;
    float dh = p.s0 > 0 ? dst[k] : 0.0f;                          // dh left by the previous chunk launch
    // saved activations / incoming gradient of a step do not depend on the recurrence: fetched ONE STEP AHEAD into
    // registers so that their HBM/L2 latency is off the dependent chain.  Like the forward kernel the loop is BRANCH-FREE (all JS
    // lanes of a hidden index compute and store the same values; out-of-range steps load a clamped address and select zero), so the
    // compiler can count the stores behind the prefetch loads instead of draining them every step (s_waitcnt vmcnt(0)).
    float pf_r = 0.f, pf_u = 0.f, pf_c = 0.f, pf_h = 0.f, pf_d = 0.f;
    auto prefetch = [&](int s) {
        const int t = dir == 0 ? p.T - 1 - s : s;
        const int tl = min(t, max(len - 1, 0));
        const float* q = ruc + (long)tl * 3 * H;
        const float r_ = q[k], u_ = q[H + k], c_ = q[2 * H + k], d_ = dorow[(long)tl * p.lddo + k];
        const int tp = dir == 0 ? t - 1 : t + 1;           // forward-order predecessor
        const float h_ = orow[(long)min(max(tp, 0), p.T - 1) * p.ldo + k];
        const bool v = t < len;
        pf_r = v ? r_ : 0.f; pf_u = v ? u_ : 0.f; pf_c = v ? c_ : 0.f; pf_d = v ? d_ : 0.f;
        pf_h = (v && tp >= 0 && tp < len) ? h_ : 0.f;
    };
    prefetch(p.s0);

    for (int s = p.s0; s < p.s1; ++s) {
        const int t = dir == 0 ? p.T - 1 - s : s;          // reverse of the forward order
        const bool valid = t < len;
        const float r = pf_r, u = pf_u, c = pf_c, hprev = pf_h;
        const float dhT = valid ? dh + pf_d : 0.f;
        prefetch(min(s + 1, p.s1 - 1));                    // (the last step re-reads its own)
        const float du = dhT * (hprev - c);
        float dh_new = valid ? dhT * u : dh;
        const float dcp = dhT * (1.0f - u) * (1.0f - c * c);
        dcp_l[LIDX(k)] = dcp;
        lds_barrier();
        // ---- drh[k] = sum_j dcp[j] * Wc[k][j]
        float drh;
        {
            f2 d2 = {0.0f, 0.0f}, d3 = {0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < JC / 4; q += 2) {
                pk_dot4(*reinterpret_cast<const float4*>(dcp_l + LIDX(jq * JC + q * 4)), wcT + 2 * q, d2);
                pk_dot4(*reinterpret_cast<const float4*>(dcp_l + LIDX(jq * JC + q * 4 + 4)), wcT + 2 * q + 2, d3);
            }
            drh = (d2.x + d2.y) + (d3.x + d3.y);
        }
        drh = group_sum<JS>(drh);
        {
            dh_new += drh * r;
            const float dgr = drh * hprev * r * (1.0f - r);
            const float dgu = du * u * (1.0f - u);
            dg_l[LIDX(k)] = dgr;
            dg_l[LIDX(H + k)] = dgu;
            float* dx = dxrow + (long)t * p.ldxp;
            dx[k] = dgr; dx[H + k] = dgu; dx[2 * H + k] = dcp;
            hprow[(long)t * H + k] = hprev;
            rhrow[(long)t * H + k] = r * hprev;
        }
        lds_barrier();
        // ---- dh_{prev}[k] += sum_j dg[j] * Wg[k][j]   (j over 256, quarter of 64 per lane)
        float e;
        {
            f2 e2 = {0.0f, 0.0f}, e3 = {0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < JG / 4; q += 2) {
                pk_dot4(*reinterpret_cast<const float4*>(dg_l + LIDX(jq * JG + q * 4)), wgT + 2 * q, e2);
                pk_dot4(*reinterpret_cast<const float4*>(dg_l + LIDX(jq * JG + q * 4 + 4)), wgT + 2 * q + 2, e3);
            }
            e = (e2.x + e2.y) + (e3.x + e3.y);
        }
        e = group_sum<JS>(e);
        dh = dh_new + e;
    }
    if (p.s1 < p.T) dst[k] = dh;
}

// TACO_GRU128_THREADS = 256 | 512 (default 512; measured: 256 threads = one wave per SIMD is 27 % SLOWER per step, see gru128_seq_fwd_k)
static int gru128_threads() { const char* e = getenv("TACO_GRU128_THREADS"); return (e && atoi(e) == 256) ? 256 : 512; }

// isolation pad: see csrc/gru256.hip (unused dynamic LDS: with ~150 KB a workgroup has its CU to itself)
static size_t gru128_pad(const void* k, int bytes, DevMask& done) {
    if (bytes <= 0) return 0;
    if (ensure_dyn_lds(k, 158000, done) != TACO_OK) return 0;
    return (size_t)(bytes > 150000 ? 150000 : bytes);
}

extern "C" int taco_gru128_seq_fwd(const float* xp, int ldxp, const float* wg_fw, const float* wc_fw, const float* wg_bw,
                                   const float* wc_bw, const int* lengths, float* out, int ldo, float* ruc, int N, int T,
                                   int ndir, int s0, int s1, float* state, int isolate_lds_bytes, hipStream_t stream) {
    if (!xp || !wg_fw || !wc_fw || !out || !ruc || N <= 0 || T <= 0 || ndir < 1 || ndir > 2) return TACO_EINVAL;
    if (ndir == 2 && (!wg_bw || !wc_bw)) return TACO_EINVAL;
    if (s0 < 0 || s1 > T || s0 >= s1 || ((s0 > 0 || s1 < T) && !state)) return TACO_EINVAL;
    GruSeq p{};
    p.xp = xp; p.ldxp = ldxp; p.wg[0] = wg_fw; p.wc[0] = wc_fw; p.wg[1] = wg_bw; p.wc[1] = wc_bw;
    p.lengths = lengths; p.out = out; p.ldo = ldo; p.ruc = ruc; p.N = N; p.T = T; p.s0 = s0; p.s1 = s1; p.state = state;
    static DevMask a256{0}, a512{0};
    if (gru128_threads() == 256)
        hipLaunchKernelGGL(gru128_seq_fwd_k<256>, dim3(N, ndir), dim3(256), gru128_pad((const void*)gru128_seq_fwd_k<256>, isolate_lds_bytes, a256), stream, p);
    else
        hipLaunchKernelGGL(gru128_seq_fwd_k<512>, dim3(N, ndir), dim3(512), gru128_pad((const void*)gru128_seq_fwd_k<512>, isolate_lds_bytes, a512), stream, p);
    TACO_RETURN_LAST();
}

extern "C" int taco_gru128_seq_bwd(const float* dout, int lddo, const float* wg_fw, const float* wc_fw, const float* wg_bw,
                                   const float* wc_bw, const int* lengths, const float* out, int ldo, const float* ruc,
                                   float* dxp, int ldxp, float* hp, float* rh, int N, int T, int ndir, int s0, int s1, float* state,
                                   int isolate_lds_bytes, hipStream_t stream) {
    if (!dout || !wg_fw || !wc_fw || !out || !ruc || !dxp || !hp || !rh || N <= 0 || T <= 0 || ndir < 1 || ndir > 2) return TACO_EINVAL;
    if (ndir == 2 && (!wg_bw || !wc_bw)) return TACO_EINVAL;
    if (ldxp & 3) return TACO_EINVAL;
    if (s0 < 0 || s1 > T || s0 >= s1 || ((s0 > 0 || s1 < T) && !state)) return TACO_EINVAL;
    GruSeq p{};
    p.ldxp = ldxp; p.wg[0] = wg_fw; p.wc[0] = wc_fw; p.wg[1] = wg_bw; p.wc[1] = wc_bw;
    p.lengths = lengths; p.out = const_cast<float*>(out); p.ldo = ldo; p.ruc = const_cast<float*>(ruc); p.N = N; p.T = T;
    p.dout = dout; p.lddo = lddo; p.dxp = dxp; p.hp = hp; p.rh = rh; p.s0 = s0; p.s1 = s1; p.state = state;
    static DevMask a256{0}, a512{0};
    if (gru128_threads() == 256)
        hipLaunchKernelGGL(gru128_seq_bwd_k<256>, dim3(N, ndir), dim3(256), gru128_pad((const void*)gru128_seq_bwd_k<256>, isolate_lds_bytes, a256), stream, p);
    else
        hipLaunchKernelGGL(gru128_seq_bwd_k<512>, dim3(N, ndir), dim3(512), gru128_pad((const void*)gru128_seq_bwd_k<512>, isolate_lds_bytes, a512), stream, p);
    TACO_RETURN_LAST();
}
