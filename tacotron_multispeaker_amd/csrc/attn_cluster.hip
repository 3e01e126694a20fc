// Persistent CLUSTER kernel for the attention recurrence of the Tacotron decoder (teacher forcing), forward.
//
// Reference: AttentionWrapper(DecoderPrenetWrapper(GRUCell(256)), BahdanauAttention(256, encoder_outputs))
// (models/tacotron.py:66-70, models/rnn_wrappers.py:22-24), semantics SURVEY Appendix A.5/A.7/A.8.
//
// One launch runs all S decoder steps.  A cluster of 8 workgroups (512 threads each) owns two batch rows; the
// FEATURE axis is split 8 ways: workgroup w keeps, for the whole kernel,
//   * in registers (112 VGPRs/lane): its column slices of the prenet (W1[80:336], W2), attention-GRU (Wx, Whg, Whc)
//     and query (Wq) kernels -- 32 (16 for W2, 64 for the gates) output columns each, K split over lanes;
//   * in LDS: the 32-column slices keys[row][:, D_w] and memory[row][:, D_w] of its two rows (Ti x 32 x 2 x 2 floats),
//     i.e. the attention score tile never leaves the CU after the prologue.
// Per step the cluster exchanges six vectors (ctx, p1, p2, r*h, h', partial scores) as 8-byte {epoch,value}
// granules (one agent-scope store each, polled with agent-scope loads: Guideline 16 form R2).  Scores are reduced over
// the 256-axis with cross-lane adds inside a workgroup and over workgroups by every member in the SAME order, so all
// members compute bit-identical softmax weights (softmax over ALL Ti positions, no memory mask: Appendix A.7).
// Clusters never talk to each other; grid = 8 * ceil(N/2) <= 256 co-resident workgroups; spins are bounded.
#include "attn_cluster.hpp"
#include "xcd_granule.hpp"

#define CW 8                 // workgroups per cluster
#define TACO_CARRY_TAG 0x43590000u      // 'CY..': epochs of the carry granules = this + the step index of the chunk boundary
// Diagnostic build (-DTACO_STAMP): thread 0 of workgroup 0 accumulates s_memtime deltas per phase and writes them behind
// the exchange region (never read by the kernel); the shipped build contains no stamp.
#ifdef TACO_STAMP
#define STAMP_DECL unsigned long long st_acc[16] = {0}; unsigned long long st_last = __builtin_readcyclecounter();
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_last; st_last = t_; } } while (0)
#define STAMP_OUT(ptr) do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) (ptr)[i_] = st_acc[i_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_OUT(ptr)
#endif
#define AT 512               // threads per workgroup
#define SPIN_LIMIT (1 << 22)
#define PIDX(k) ((k) + ((k) >> 4) * 4)          // LDS vector layout: 4 pad floats after every 16 (bank spreading)
#define PLEN(n) ((n) + ((n) >> 4) * 4)

// granule store: agent scope (any placement) or, when the cluster has verified that all its members sit on one XCD
// (xcd_granule.hpp: cluster_on_one_xcd), the L2-local form that a same-XCD reader hits in the shared L2 (0.48 vs 0.64 us per
// exchange round for 8 workgroups, scripts/dev_xchg.py)
__device__ __forceinline__ void put_g(u64* p, unsigned epoch, float v, bool local) {
    if (local) put_granule_xcd(p, epoch, v); else put_granule(p, epoch, v);
}
// indexed put with an opaque index: the 64-bit lane address is formed at the store instead of being hoisted out of the
// step loop as a live (and then spilled) VGPR pair
__device__ __forceinline__ void put_gi(u64* base, unsigned idx, unsigned epoch, float v, bool local) {
    asm volatile("" : "+v"(idx));
    put_g(base + idx, epoch, v, local);
}
// placement check of one cluster, broadcast to the workgroup (flag_l: an LDS int)
__device__ __forceinline__ bool cluster_local(u64* slots, int w, int* err, int* flag_l, int tid, int allow, unsigned salt) {
    return cluster_shares_xcd(slots, w, CW, err, flag_l, tid, allow, salt);
}
// TACO_XCD_LOCAL=0 keeps the agent-scope form everywhere (A/B timing, tests of the placement-independent path)
static int xcd_local_allowed() { const char* e = getenv("TACO_XCD_LOCAL"); return (e && e[0] == '0') ? 0 : 1; }
template <int NG>
__device__ __forceinline__ void get_g(const u64* const (&ptr)[NG], unsigned epoch, float (&out)[NG], int* err) {
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

// dot of this lane's K-part with the register weights, both rows; vec rows are PIDX-laid-out LDS vectors
template <int KPER>
__device__ __forceinline__ void dot2(const float* __restrict__ v0, const float* __restrict__ v1, int kbase,
                                     const float (&w)[KPER], float& a0, float& a1) {
    f2 p0 = {a0, 0.0f}, p1 = {a1, 0.0f};
#pragma unroll
    for (int k4 = 0; k4 < KPER / 4; ++k4) {
        // keep at most 4 LDS vectors in flight: hoisting every ds_read of a 32-deep part ahead of the FMAs costs
        // 64 VGPRs and pushes the register-resident weights into scratch
        if (k4 && (k4 & 1) == 0) asm volatile("" ::: "memory");
        const int k = kbase + k4 * 4;
        const float4 x0 = *reinterpret_cast<const float4*>(v0 + PIDX(k));
        const float4 x1 = *reinterpret_cast<const float4*>(v1 + PIDX(k));
        pk_dot4x2(x0, x1, w[k4 * 4], w[k4 * 4 + 1], w[k4 * 4 + 2], w[k4 * 4 + 3], p0, p1);
    }
    a0 = p0.x + p0.y; a1 = p1.x + p1.y;
}
// same, weights parked in LDS as per-thread float4 slots wl[(k4 * AT + tid) * 4 ..] (conflict-free b128 reads)
template <int KPER>
__device__ __forceinline__ void dot2_lds(const float* __restrict__ v0, const float* __restrict__ v1, int kbase,
                                         const float* __restrict__ wl, int tid, float& a0, float& a1) {
    f2 p0 = {a0, 0.0f}, p1 = {a1, 0.0f};
#pragma unroll
    for (int k4 = 0; k4 < KPER / 4; ++k4) {
        if (k4 && (k4 & 1) == 0) asm volatile("" ::: "memory");
        const int k = kbase + k4 * 4;
        const float4 x0 = *reinterpret_cast<const float4*>(v0 + PIDX(k));
        const float4 x1 = *reinterpret_cast<const float4*>(v1 + PIDX(k));
        const float4 wv = *reinterpret_cast<const float4*>(wl + (k4 * AT + tid) * 4);
        pk_dot4x2(x0, x1, wv.x, wv.y, wv.z, wv.w, p0, p1);
    }
    a0 = p0.x + p0.y; a1 = p1.x + p1.y;
}
template <int PARTS>
__device__ __forceinline__ float lane_reduce(float v) { return group_sum<PARTS>(v); }

// gather the 7 peers' slices (LEN values per row per workgroup) of a published vector into a PIDX LDS vector [2][..]
template <int LEN>
__device__ __forceinline__ void gather_vec(const u64* region, float* lds0, float* lds1, int w, unsigned epoch, int tid, int* err) {
    constexpr int TOT = 2 * (CW - 1) * LEN;
    asm volatile("" : "+v"(tid));          // opaque: recompute the granule address per call, keep no pointer live
    if (tid < TOT) {
        const int row = tid / ((CW - 1) * LEN), rem = tid - row * (CW - 1) * LEN;
        const int peer = rem / LEN, jj = rem - peer * LEN;
        const int pw = peer + (peer >= w ? 1 : 0);
        const int j = pw * LEN + jj;
        const u64* const ptr[1] = {region + row * (CW * LEN) + j};
        float val[1];
        get_g<1>(ptr, epoch, val, err);
        (row ? lds1 : lds0)[PIDX(j)] = val[0];
    }
}

// WLDS: the recurrent gate weights (Whg slice, 32 floats per thread) live in LDS instead of registers.  Their product with
// h is computed off the dependent chain (while the partial scores are exchanged), so the extra LDS reads are hidden, and the
// 32 freed VGPRs keep the kernel free of scratch spills (with all 112 weight floats in registers the loop-invariant granule
// addresses are spilled and reloaded inside the poll loops).  The register variant remains for long inputs whose key / memory
// tiles need the LDS (T_in > ~160).
// NT: register slots per lane of the per-wave softmax = ceil(T_in / 64), compiled for 2 (T_in <= 128: the C2 / C4 shapes) and 5
// (T_in <= 320 covers everything the cluster path holds): with 8 slots unrolled, 6 of 8 iterations ran predicated-off at T_in = 128.
template <bool WLDS, int NT>
__global__ __launch_bounds__(AT, 2) void attn_cluster_fwd_k(AttnClu p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int nclus = gridDim.x / CW;
    int w, cl;
    if ((nclus & 7) == 0) { const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3; w = qq & 7; cl = (qq >> 3) * 8 + xcd; }
    else { w = blockIdx.x & 7; cl = blockIdx.x >> 3; }
    const int Ti = p.Ti, S = p.S;
    const int row0 = cl * 2;
    // an odd batch's last cluster runs row N-1 twice: both copies compute and store identical bits, so no store is guarded
    const long rw[2] = {(long)min(row0, p.N - 1), (long)min(row0 + 1, p.N - 1)};

    // ---- LDS carve-up
    float* ctx_l = smem;                          // [2][PLEN(256)]
    float* p1_l = ctx_l + 2 * PLEN(256);
    float* p2_l = p1_l + 2 * PLEN(256);           // [2][PLEN(128)]
    float* h_l = p2_l + 2 * PLEN(128);
    float* rh_l = h_l + 2 * PLEN(256);
    float* u_l = rh_l + 2 * PLEN(256);            // [2][32]
    float* q_l = u_l + 64;                        // [2][32]
    float* an_l = q_l + 64;                       // [2*Ti] alignments (every wave of a row writes the same values)
    float* a_l = an_l + ((2 * Ti + 3) & ~3);      // [2*Ti] scores
    float* cp_l = a_l + ((2 * Ti + 3) & ~3);      // [16] scratch (placement verdict)
    float* red_l = cp_l + 16;                     // [32] attention_v slice
    float* K_l = red_l + 32;                      // [2][Ti][32]
    float* M_l = K_l + 2 * Ti * 32;               // [2][Ti][32]
    for (int i = tid; i < 2 * PLEN(256) * 4 + 2 * PLEN(128); i += AT) smem[i] = 0.0f;    // ctx,p1,p2,h,rh = 0
    for (int i = tid; i < 2 * Ti * 8; i += AT) {          // float4 granularity: (row, t, c4)
        const int c4 = i & 7, t = (i >> 3) % Ti, row = (i >> 3) / Ti;
        const long g = ((rw[row] * Ti) + t) * 256 + 32 * w + c4 * 4;
        *reinterpret_cast<float4*>(K_l + (row * Ti + t) * 32 + c4 * 4) = *reinterpret_cast<const float4*>(p.keys + g);
        // memory tile: the 16-byte chunks of row t are rotated by t & 7, i.e. column d lives at (d + 4 * (t & 7)) & 31: the
        // context product reads (t = tp + 8 i, d) from lanes (d & 3 | tp) of a 32-lane half -> 32 distinct banks
        *reinterpret_cast<float4*>(M_l + (row * Ti + t) * 32 + ((c4 + t) & 7) * 4) = *reinterpret_cast<const float4*>(p.mem + g);
    }

    // ---- register-resident weight slices
    const int cA = tid >> 4, pA = tid & 15;     // 32 cols x 16 parts (prenet1, cand, query)
    const int cB = tid >> 5, pB = tid & 31;     // 16 cols x 32 parts (prenet2)
    const int cC = tid >> 3, pC = tid & 7;      // 64 cols x  8 parts (gates)
    const int gc = cC < 32 ? 32 * w + cC : 256 + 32 * w + (cC - 32);     // gate column: r_J | u_J
    float* W_l = M_l + 2 * Ti * 32;             // WLDS: [8][AT][4] Whg slice as per-thread float4 slots
    float w1[16], w2[8], wgx[16], wgh[WLDS ? 1 : 32], wcx[8], wch[16], wqr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w1[k] = p.w1c[(long)(pA * 16 + k) * 256 + 32 * w + cA];
#pragma unroll
    for (int k = 0; k < 8; ++k) w2[k] = p.w2[(long)(pB * 8 + k) * 128 + 16 * w + cB];
#pragma unroll
    for (int k = 0; k < 16; ++k) wgx[k] = p.wx[(long)(pC * 16 + k) * 768 + gc];
    if (WLDS) {
#pragma unroll
        for (int k = 0; k < 32; ++k) W_l[((k >> 2) * AT + tid) * 4 + (k & 3)] = p.whg[(long)(pC * 32 + k) * 512 + gc];
        wgh[0] = 0.f;
    } else {
#pragma unroll
        for (int k = 0; k < (WLDS ? 1 : 32); ++k) wgh[k] = p.whg[(long)(pC * 32 + k) * 512 + gc];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) wcx[k] = p.wx[(long)(pA * 8 + k) * 768 + 512 + 32 * w + cA];
#pragma unroll
    for (int k = 0; k < 16; ++k) wch[k] = p.whc[(long)(pA * 16 + k) * 256 + 32 * w + cA];
#pragma unroll
    for (int k = 0; k < 16; ++k) wqr[k] = p.wq[(long)(pA * 16 + k) * 256 + 32 * w + cA];
    const float b2v = p.b2[16 * w + cB], bgv = p.bg[gc], bcv = p.bg[512 + 32 * w + cA];
    // attention_v slice of this workgroup's 32 score dims lives in LDS (red_l[0:32])
    if (tid < 32) red_l[tid] = p.v[32 * w + tid];

    // exchange regions of this cluster: CTX, P1 [2][256]; P2 [2][128]; RH, H [2][256]; E [8][2*Ti]
    const long per_clu = 2 * 256 * 4 + 2 * 128 + (long)CW * 2 * Ti;
    u64* X = p.xchg + (long)cl * per_clu;
    u64 *xCTX = X, *xP1 = X + 512, *xP2 = X + 1024, *xRH = X + 1280, *xH = X + 1792, *xE = X + 2304;
    const bool local = cluster_local(p.xchg + (long)nclus * per_clu + (long)cl * CW, w, p.err, reinterpret_cast<int*>(cp_l), tid, p.xcd_local, (unsigned)p.s0);
    float fnx0 = 0.f, fnx1 = 0.f;
    if (pA == 0) {
        fnx0 = p.f1[(unsigned)(rw[0] * S + p.s0) * 256u + 32 * w + cA];
        fnx1 = p.f1[(unsigned)(rw[1] * S + p.s0) * 256u + 32 * w + cA];
    }
    __syncthreads();
    if (p.s0 > 0 && tid < 256) {                          // state of the previous chunk: h_{s0-1}, ctx_{s0-1}
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned o = (unsigned)(rw[b] * S + p.s0 - 1) * 512u;
            h_l[b * PLEN(256) + PIDX(tid)] = p.hc[o + tid];
            ctx_l[b * PLEN(256) + PIDX(tid)] = p.hc[o + 256u + tid];
        }
    }
    __syncthreads();
    STAMP_DECL
    float gh0 = 0.f, gh1 = 0.f;                 // h part of the gate pre-activations of the coming step (per-lane partials)
    if constexpr (WLDS) dot2_lds<32>(h_l, h_l + PLEN(256), pC * 32, W_l, tid, gh0, gh1);
    else dot2<32>(h_l, h_l + PLEN(256), pC * 32, wgh, gh0, gh1);

    for (int s = p.s0; s < p.s1; ++s) {
        const unsigned epoch = (unsigned)s + 1;          // step of the PASS: the chunk launches of a pass share one zero-filled buffer
        STAMP(0);
        unsigned so0 = (unsigned)(rw[0] * S + s), so1 = (unsigned)(rw[1] * S + s);   // row offsets into [N,S,*] tensors
        asm volatile("" : "+v"(so0), "+v"(so1));      // opaque: addresses are formed at the point of use (see above)
        // ================= A: prenet dense_1 (context part; frame part + bias hoisted into f1) =================
        {
            const int j = 32 * w + cA;
            // frame part of dense_1 for this step was fetched one step ahead (HBM latency off the critical path)
            const float f0 = fnx0, f1v = fnx1;
            if (pA == 0 && s + 1 < p.s1) { fnx0 = p.f1[(so0 + 1u) * 256u + j]; fnx1 = p.f1[(so1 + 1u) * 256u + j]; }
            float a0 = 0.f, a1 = 0.f;
            dot2<16>(ctx_l, ctx_l + PLEN(256), pA * 16, w1, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) {
                a0 = fmaxf(a0 + f0, 0.f); a1 = fmaxf(a1 + f1v, 0.f);
                p1_l[PIDX(j)] = a0; p1_l[PLEN(256) + PIDX(j)] = a1;
                put_g(xP1 + j, epoch, a0, local); put_g(xP1 + 256 + j, epoch, a1, local);
                p.p1[so0 * 256 + j] = a0;
                p.p1[so1 * 256 + j] = a1;
            }
            STAMP(1);
            gather_vec<32>(xP1, p1_l, p1_l + PLEN(256), w, epoch, tid, p.err);
        }
        lds_barrier();
        STAMP(2);
        // ================= B: prenet dense_2 =================
        {
            const int j = 16 * w + cB;
            float a0 = 0.f, a1 = 0.f;
            dot2<8>(p1_l, p1_l + PLEN(256), pB * 8, w2, a0, a1);
            a0 = lane_reduce<32>(a0); a1 = lane_reduce<32>(a1);
            if (pB == 0) {
                a0 = fmaxf(a0 + b2v, 0.f); a1 = fmaxf(a1 + b2v, 0.f);
                p2_l[PIDX(j)] = a0; p2_l[PLEN(128) + PIDX(j)] = a1;
                put_g(xP2 + j, epoch, a0, local); put_g(xP2 + 128 + j, epoch, a1, local);
                p.p2[so0 * 128 + j] = a0;
                p.p2[so1 * 128 + j] = a1;
            }
            STAMP(3);
            gather_vec<16>(xP2, p2_l, p2_l + PLEN(128), w, epoch, tid, p.err);
        }
        lds_barrier();
        STAMP(4);
        // ================= C: GRU gates (the h part of the pre-activation was computed off the chain, see F) =================
        float cx0 = 0.f, cx1 = 0.f;                 // candidate: x part, computed while the r*h exchange is in flight
        {
            const int j = 32 * w + (cC & 31);
            float a0 = gh0, a1 = gh1;
            dot2<16>(p2_l, p2_l + PLEN(128), pC * 16, wgx, a0, a1);
            a0 = lane_reduce<8>(a0); a1 = lane_reduce<8>(a1);
            if (pC == 0) {
                const float g0 = fast_sigmoid(a0 + bgv), g1 = fast_sigmoid(a1 + bgv);
                if (cC < 32) {
                    const float q0 = g0 * h_l[PIDX(j)], q1 = g1 * h_l[PLEN(256) + PIDX(j)];
                    rh_l[PIDX(j)] = q0; rh_l[PLEN(256) + PIDX(j)] = q1;
                    put_g(xRH + j, epoch, q0, local); put_g(xRH + 256 + j, epoch, q1, local);
                    p.r[so0 * 256 + j] = g0; p.rh[so0 * 256 + j] = q0;
                    p.r[so1 * 256 + j] = g1; p.rh[so1 * 256 + j] = q1;
                } else {
                    u_l[cC - 32] = g0; u_l[32 + cC - 32] = g1;
                    p.u[so0 * 256 + j] = g0;
                    p.u[so1 * 256 + j] = g1;
                }
            }
            STAMP(5);
            dot2<8>(p2_l, p2_l + PLEN(128), pA * 8, wcx, cx0, cx1);
            gather_vec<32>(xRH, rh_l, rh_l + PLEN(256), w, epoch, tid, p.err);
        }
        lds_barrier();
        STAMP(6);
        // ================= D: candidate + new state =================
        {
            const int j = 32 * w + cA;
            float a0 = cx0, a1 = cx1;
            dot2<16>(rh_l, rh_l + PLEN(256), pA * 16, wch, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) {
                const float c0 = fast_tanh(a0 + bcv), c1 = fast_tanh(a1 + bcv);
                const float u0 = u_l[cA], u1 = u_l[32 + cA];
                const float hn0 = u0 * h_l[PIDX(j)] + (1.f - u0) * c0;
                const float hn1 = u1 * h_l[PLEN(256) + PIDX(j)] + (1.f - u1) * c1;
                h_l[PIDX(j)] = hn0; h_l[PLEN(256) + PIDX(j)] = hn1;
                put_g(xH + j, epoch, hn0, local); put_g(xH + 256 + j, epoch, hn1, local);
                p.c[so0 * 256 + j] = c0; p.hc[so0 * 512 + j] = hn0;
                p.c[so1 * 256 + j] = c1; p.hc[so1 * 512 + j] = hn1;
            }
            STAMP(7);
            gather_vec<32>(xH, h_l, h_l + PLEN(256), w, epoch, tid, p.err);
        }
        lds_barrier();
        STAMP(8);
        // ================= E: query slice =================
        {
            float a0 = 0.f, a1 = 0.f;
            dot2<16>(h_l, h_l + PLEN(256), pA * 16, wqr, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) {
                q_l[cA] = a0; q_l[32 + cA] = a1;
                p.q[so0 * 256 + 32 * w + cA] = a0;
                p.q[so1 * 256 + 32 * w + cA] = a1;
            }
        }
        lds_barrier();
        STAMP(9);
        // ================= F: partial scores over this workgroup's 32 dims, all t =================
        // pass 1: own partials, published (and parked in a_l); then work that is off the dependent chain; pass 2: gather + reduce
        int tf = threadIdx.x;
        asm volatile("" : "+v"(tf));               // opaque per step (see above)
        for (int i = tf >> 1; i < 2 * Ti; i += AT / 2) {
            const int half = tf & 1;
            const int row = i >= Ti;
            const float* kp = K_l + i * 32 + half * 16;
            const float* qp = q_l + row * 32 + half * 16;
            const float* vp = red_l + half * 16;
            float e = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const float4 kv = *reinterpret_cast<const float4*>(kp + d4 * 4);
                const float4 qv = *reinterpret_cast<const float4*>(qp + d4 * 4);
                const float4 vv = *reinterpret_cast<const float4*>(vp + d4 * 4);
                e = fmaf(vv.x, fast_tanh(kv.x + qv.x), e); e = fmaf(vv.y, fast_tanh(kv.y + qv.y), e);
                e = fmaf(vv.z, fast_tanh(kv.z + qv.z), e); e = fmaf(vv.w, fast_tanh(kv.w + qv.w), e);
            }
            e = dpp_add<0xB1>(e);                                  // both lanes of the pair now hold this workgroup's partial
            if (half == 0) { put_g(xE + (long)w * 2 * Ti + i, epoch, e, local); a_l[i] = e; }
        }
        // off the dependent chain, while the partial scores travel: the h part of the NEXT step's gate pre-activations
        // (h_l holds the complete h_s since the barrier after D and is not written again before the next step's D)
        gh0 = 0.f; gh1 = 0.f;
        if constexpr (WLDS) dot2_lds<32>(h_l, h_l + PLEN(256), (tf & 7) * 32, W_l, tf, gh0, gh1);
        else dot2<32>(h_l, h_l + PLEN(256), pC * 32, wgh, gh0, gh1);
        asm volatile("" : "+v"(tf));
        for (int i = tf >> 1; i < 2 * Ti; i += AT / 2) {
            const int half = tf & 1;
            const float e = a_l[i];                                // written by this lane pair above (same wave: in order)
            // gather + reduce the 8 partials of (row,t) in workgroup order (bit-identical in all members); the own
            // partial comes from LDS, its slot is replaced by a dummy peer slot in the poll
            float val[4];
            const u64* ptr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pw = half * 4 + k;
                ptr[k] = xE + (long)(pw == w ? ((w + 1) & 7) : pw) * 2 * Ti + i;
            }
            const u64* const cptr[4] = {ptr[0], ptr[1], ptr[2], ptr[3]};
            get_g<4>(cptr, epoch, val, p.err);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += (half * 4 + k == w) ? e : val[k];
            const float tot2 = dpp_add<0xB1>(sum);
            if (half == 0) a_l[i] = tot2;
        }
        lds_barrier();
        STAMP(10);
        STAMP(11);
        // ================= G+H: softmax and context slice, one pass per wave, no barrier in between =================
        // Wave v works on row v >> 2 and context dims 8 (v & 3) .. +8.  Every wave of a row recomputes that row's softmax from
        // the (bit-identical) total scores -- 2..8 exponentials per lane -- and writes the alignments to an_l; the four waves of
        // a row write the same bits, and each reads back only after its own writes (same-wave LDS ordering), so no workgroup
        // barrier separates softmax and context product.  ctx[row][d] = sum_t a[t] * mem[t][d]: lane (d & 7 | tp) sums
        // t = tp, tp + 8, ... and the 8 tp lanes are combined with DPP adds.
        {
            int tq = threadIdx.x;
            asm volatile("" : "+v"(tq));           // opaque: nothing lane-derived is hoisted out of the step loop (it would spill)
            const int wv = tq >> 6, lane = tq & 63;
            const int row = wv >> 2;
            const float* er = a_l + row * Ti;
            float* ar = an_l + row * Ti;
            float ev[NT];                                  // Ti <= 64 NT values per row live in registers
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < NT; ++i) { const int t = lane + 64 * i; ev[i] = t < Ti ? er[t] : -INFINITY; mx = fmaxf(mx, ev[i]); }
            mx = wave_max_fast(mx);
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) { ev[i] = __builtin_amdgcn_exp2f((ev[i] - mx) * 1.4426950408889634f); sm += ev[i]; }
            sm = wave_sum_fast(sm);
            const float inv = __builtin_amdgcn_rcpf(sm);
            const unsigned arow = (row ? so1 : so0) * (unsigned)Ti;
            const bool wr = (wv & 3) == 0 && (lane & 7) == w;                // member w stores t = w, w + 8, ... to HBM
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = lane + 64 * i;
                if (t < Ti) { const float av = ev[i] * inv; ar[t] = av; if (wr) p.align[arow + t] = av; }
            }
            const int tp = lane & 7, d = 8 * (wv & 3) + (lane >> 3);
            const float* mr = M_l + row * Ti * 32;
            float acc0 = 0.f, acc1 = 0.f;
            int t = tp;
            for (; t + 8 < Ti; t += 16) {
                acc0 = fmaf(ar[t], mr[t * 32 + ((d + 4 * tp) & 31)], acc0);
                acc1 = fmaf(ar[t + 8], mr[(t + 8) * 32 + ((d + 4 * tp) & 31)], acc1);
            }
            if (t < Ti) acc0 = fmaf(ar[t], mr[t * 32 + ((d + 4 * tp) & 31)], acc0);
            const float cx = group_sum<8>(acc0 + acc1);
            if (tp == 0) {
                const int j = 32 * w + d;
                ctx_l[row * PLEN(256) + PIDX(j)] = cx;
                put_g(xCTX + row * 256 + j, epoch, cx, local);
                p.hc[(row ? so1 : so0) * 512u + 256u + j] = cx;
            }
        }
        STAMP(13);
        gather_vec<32>(xCTX, ctx_l, ctx_l + PLEN(256), w, epoch, tid, p.err);
        lds_barrier();
        STAMP(14);
    }
    STAMP_OUT(p.xchg + (long)nclus * (per_clu + CW));
}

// per cluster: the exchange regions + CW placement granules; + 16 diagnostic stamp slots at the very end
extern "C" int taco_attn_cluster_xchg_slots(int N, int Ti) {
    return ((N + 1) / 2) * (2 * 256 * 4 + 2 * 128 + CW * 2 * Ti + CW) + 16;
}

static size_t attn_cluster_smem(int Ti, bool wlds = false) {
    size_t f = 2 * PLEN(256) * 4 + 2 * PLEN(128) + 64 + 64 + 2 * ((2 * Ti + 3) & ~3) + 16 + 32 + (size_t)4 * Ti * 32 +
               (wlds ? 8 * AT * 4 : 0);
    return f * sizeof(float);
}

// returns 1 when the cluster path can run this shape
static size_t attn_cluster_bwd_smem(int Ti, bool wlds);
extern "C" int taco_attn_cluster_supported(int N, int Ti) {
    return (CW * ((N + 1) / 2) <= 256 && attn_cluster_smem(Ti) <= 160 * 1024 && attn_cluster_bwd_smem(Ti, false) <= 160 * 1024 &&
            Ti >= 1 && Ti <= 320) ? 1 : 0;       // per-wave softmax: <= 5 register slots per lane
}

// which forward kernel a (N, Ti) launch runs: 0 = per-step kernels, 1 = attn_cluster_fwd_k<true> (Whg slice in LDS), 2 = <false>
extern "C" int taco_attn_cluster_fwd_variant(int N, int Ti) {
    if (!taco_attn_cluster_supported(N, Ti)) return 0;
    const char* e = getenv("TACO_ATTN_FWD_NO_WLDS");
    const bool no_wlds = e && e[0] && e[0] != '0';
    return (!no_wlds && attn_cluster_smem(Ti, true) <= 160 * 1024) ? 1 : 2;
}

int attn_cluster_fwd_launch(const AttnClu& p, hipStream_t st) {
    static DevMask attr_set[4] = {{0}, {0}, {0}, {0}};
    const void* ks[4] = {(const void*)attn_cluster_fwd_k<false, 2>, (const void*)attn_cluster_fwd_k<true, 2>,
                         (const void*)attn_cluster_fwd_k<false, 5>, (const void*)attn_cluster_fwd_k<true, 5>};
    for (int i = 0; i < 4; ++i)
        if (ensure_dyn_lds(ks[i], 160 * 1024, attr_set[i]) != TACO_OK) return TACO_EINVAL;
    // granule epochs count the steps of the whole pass, so only the pass's first chunk launch needs a zero-filled buffer
    if (p.s0 == 0 && hipMemsetAsync(p.xchg, 0, (size_t)(taco_attn_cluster_xchg_slots(p.N, p.Ti) - 16) * sizeof(u64), st) != hipSuccess)
        return TACO_EINVAL;
    AttnClu q = p;
    q.xcd_local = xcd_local_allowed();
    const dim3 grid(CW * ((p.N + 1) / 2));
    const bool wlds = taco_attn_cluster_fwd_variant(p.N, p.Ti) == 1, small = p.Ti <= 128;
    const size_t smem = attn_cluster_smem(p.Ti, wlds);
    if (wlds && small) hipLaunchKernelGGL((attn_cluster_fwd_k<true, 2>), grid, dim3(AT), smem, st, q);
    else if (wlds) hipLaunchKernelGGL((attn_cluster_fwd_k<true, 5>), grid, dim3(AT), smem, st, q);
    else if (small) hipLaunchKernelGGL((attn_cluster_fwd_k<false, 2>), grid, dim3(AT), smem, st, q);
    else hipLaunchKernelGGL((attn_cluster_fwd_k<false, 5>), grid, dim3(AT), smem, st, q);
    TACO_RETURN_LAST();
}

// =====================================================================================================================
// Backward (BPTT) of the attention recurrence as a persistent cluster kernel.
//
// Workgroup w owns hidden indices / context dims J_w = D_w = [32w, 32w+32), prenet columns [32w,+32) (dense_1) and
// [16w,+16) (dense_2); it keeps the matching ROW slices of Wq, Whc, Whg, Wx, W2, W1c in registers ("transposed"
// products: out[k in slice] = sum_j g[j] * W[k][j]) and memory[row][:, D_w], keys[row][:, D_w] in LDS.
// Per step: da partials (all-reduce), dq, dcp, dg, dp2pre, dp1pre (all-gathers) = 6 exchanges.
// The gradients wrt keys / memory / attention_v are NOT accumulated in the loop: the kernel saves de_s[t] and the total
// dctx_s, and attn_hoisted_bwd_k reduces over s afterwards, fully parallel (dM = sum_s a_s (x) dctx_s; dK needs the tanh
// tile recomputed per (s,t,d), which is why it cannot be a GEMM).
// =====================================================================================================================
// dot of this lane's K-part, both rows, weights = a contiguous ROW slice held in registers
template <int LEN>
__device__ __forceinline__ void gather2(const u64* regA, const u64* regB, float* lds0, float* lds1, int offA, int offB,
                                        int w, unsigned epoch, int tid, int* err) {
    // two published vectors of LEN values per row per workgroup gathered with ONE poll round trip (2 granules/thread)
    constexpr int TOT = 2 * (CW - 1) * LEN;
    asm volatile("" : "+v"(tid));          // opaque: recompute the granule address per call, keep no pointer live
    if (tid < TOT) {
        const int row = tid / ((CW - 1) * LEN), rem = tid - row * (CW - 1) * LEN;
        const int peer = rem / LEN, jj = rem - peer * LEN;
        const int pw = peer + (peer >= w ? 1 : 0);
        const int j = pw * LEN + jj;
        const u64* const ptr[2] = {regA + row * (CW * LEN) + j, regB + row * (CW * LEN) + j};
        float val[2];
        get_g<2>(ptr, epoch, val, err);
        float* l = row ? lds1 : lds0;
        l[PIDX(offA + j)] = val[0];
        l[PIDX(offB + j)] = val[1];
    }
}
template <int LEN>
__device__ __forceinline__ void gather_off(const u64* region, float* lds0, float* lds1, int off, int w, unsigned epoch,
                                           int tid, int* err) {
    constexpr int TOT = 2 * (CW - 1) * LEN;
    asm volatile("" : "+v"(tid));          // opaque: recompute the granule address per call, keep no pointer live
    if (tid < TOT) {
        const int row = tid / ((CW - 1) * LEN), rem = tid - row * (CW - 1) * LEN;
        const int peer = rem / LEN, jj = rem - peer * LEN;
        const int pw = peer + (peer >= w ? 1 : 0);
        const int j = pw * LEN + jj;
        const u64* const ptr[1] = {region + row * (CW * LEN) + j};
        float val[1];
        get_g<1>(ptr, epoch, val, err);
        (row ? lds1 : lds0)[PIDX(off + j)] = val[0];
    }
}

// WLDS: the prenet-gradient weight slices (Wx^T 24 + W2^T 8 floats per thread) live in LDS instead of registers.  With all
// 112 weight floats in VGPRs the compiler spills ~58 loop-invariant dwords to scratch and reloads ~50 of them EVERY step
// (each an exposed ~300-cycle scratch load on the recurrence chain: 16.2 vs 12.2 us/step); 64 KiB of LDS removes that.
// The register variant remains for long inputs whose key/memory tiles need the LDS (Ti > 148).
template <bool WLDS, int NT>
__global__ __launch_bounds__(AT, 2) void attn_cluster_bwd_k(AttnCluB p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int nclus = gridDim.x / CW;
    int w, cl;
    if ((nclus & 7) == 0) { const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3; w = qq & 7; cl = (qq >> 3) * 8 + xcd; }
    else { w = blockIdx.x & 7; cl = blockIdx.x >> 3; }
    const int Ti = p.Ti, S = p.S;
    const int row0 = cl * 2;
    // an odd batch's last cluster runs row N-1 twice: both copies compute and store identical bits, so no store is guarded
    const long rw[2] = {(long)min(row0, p.N - 1), (long)min(row0 + 1, p.N - 1)};

    float* dq_l = smem;                            // [2][PLEN(256)]
    float* dxp_l = dq_l + 2 * PLEN(256);           // [2][PLEN(768)]: dg_r(256) | dg_u(256) | dcp(256)
    float* dp2_l = dxp_l + 2 * PLEN(768);          // [2][PLEN(128)]
    float* dp1_l = dp2_l + 2 * PLEN(128);          // [2][PLEN(256)]
    float* dctx_l = dp1_l + 2 * PLEN(256);         // [2][32]
    float* q_l = dctx_l + 64;                      // [2][32]
    float* ep_l = q_l + 64;                        // [2*Ti] own da partials
    float* a_l = ep_l + ((2 * Ti + 3) & ~3);       // [2*Ti] alignments of this step
    float* de_l = a_l + ((2 * Ti + 3) & ~3);       // [2*Ti] da -> de
    float* cp_l = de_l + ((2 * Ti + 3) & ~3);      // [8][64]
    float* st_l = cp_l + 512;                      // WLDS: [480] saved activations / external gradients of the NEXT step (see PFW below)
    float* K_l = st_l + (WLDS ? 512 : 0);
    float* M_l = K_l + 2 * Ti * 32;
    for (int i = tid; i < 2 * Ti * 8; i += AT) {
        const int c4 = i & 7, t = (i >> 3) % Ti, row = (i >> 3) / Ti;
        const long g = ((rw[row] * Ti) + t) * 256 + 32 * w + c4 * 4;
        // key tile: 16-byte chunks of row t rotated by t & 7 (column d at (d + 4 * (t & 7)) & 31): the dq product reads
        // (t = tp + 8 i, d) from lanes (d & 3 | tp) of a 32-lane half -> 32 distinct banks (as the memory tile of the forward)
        *reinterpret_cast<float4*>(K_l + (row * Ti + t) * 32 + ((c4 + t) & 7) * 4) = *reinterpret_cast<const float4*>(p.keys + g);
        *reinterpret_cast<float4*>(M_l + (row * Ti + t) * 32 + c4 * 4) = *reinterpret_cast<const float4*>(p.mem + g);
    }

    // ---- register-resident ROW slices (K split over lanes)
    const int cA = tid >> 4, pA = tid & 15;     // 32 outputs x 16 parts
    const int cB = tid >> 5, pB = tid & 31;     // 16 outputs x 32 parts
    const int jA = 32 * w + cA;                 // hidden / ctx / prenet-1 index of the A mapping
    const int jB = 16 * w + cB;                 // prenet-2 index
    float* W_l = M_l + 2 * Ti * 32;                // WLDS: [6][AT][4] Wx^T slice, then [2][AT][4] W2^T slice
    float wq[16], wc[16], wg[32], wx[WLDS ? 1 : 24], w2[WLDS ? 1 : 8], w1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) wq[k] = p.wq[(long)jA * 256 + pA * 16 + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) wc[k] = p.whc[(long)jA * 256 + pA * 16 + k];
#pragma unroll
    for (int k = 0; k < 32; ++k) wg[k] = p.whg[(long)jA * 512 + pA * 32 + k];
    if (WLDS) {
#pragma unroll
        for (int k = 0; k < 24; ++k) W_l[((k >> 2) * AT + tid) * 4 + (k & 3)] = p.wx[(long)jB * 768 + pB * 24 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) W_l[((6 + (k >> 2)) * AT + tid) * 4 + (k & 3)] = p.w2[(long)jA * 128 + pA * 8 + k];
        wx[0] = 0.f; w2[0] = 0.f;
    } else {
#pragma unroll
        for (int k = 0; k < (WLDS ? 1 : 24); ++k) wx[k] = p.wx[(long)jB * 768 + pB * 24 + k];
#pragma unroll
        for (int k = 0; k < (WLDS ? 1 : 8); ++k) w2[k] = p.w2[(long)jA * 128 + pA * 8 + k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) w1[k] = p.w1c[(long)jA * 256 + pA * 16 + k];
    const float vd = p.v[32 * w + 8 * ((tid >> 6) & 3) + ((tid & 63) >> 3)];      // dq mapping: d = 8 (wave & 3) + (lane >> 3)

    // exchange regions: DA [8][2Ti]; DQ, DCP [2][256]; DGR, DGU [2][256]; DP2 [2][128]; DP1 [2][256]
    const long per_clu = (long)CW * 2 * Ti + 2 * 256 * 5 + 2 * 128;
    u64* X = p.xchg + (long)cl * per_clu;
    u64 *xDQ = X, *xDCP = X + 512, *xDGR = X + 1024, *xDGU = X + 1536, *xDP2 = X + 2048, *xDP1 = X + 2304, *xDA = X + 2816;
    const bool local = cluster_local(p.xchg + (long)nclus * per_clu + (long)cl * CW, w, p.err, reinterpret_cast<int*>(cp_l), tid, p.xcd_local,
                                     (unsigned)(p.S - p.s1));

    // carry hand-over region behind the exchange regions and the placement granules: per cluster [dh | dctx][2 rows][256] granules,
    // then ONE residency counter for the whole launch (taco_attn_rnn_bwd_chunk)
    // A launch that WAITS runs beside the posting launch and must not share its step-exchange slots: with the L2-local granule
    // form both kernels' clusters keep dirty lines of those slots in (possibly different) XCD L2s, and the posting kernel's
    // write-back at its end can put an old epoch over a newer one in memory (seen as hand-off timeouts in 2 of 12 steps).  It
    // therefore gets an exchange buffer of its own (p.xchg) and only the carry region / counter of the posting launch's buffer.
    u64* cbase = p.carry_xchg ? p.carry_xchg : p.xchg;
    u64* cX = cbase + (long)nclus * (per_clu + CW) + (long)cl * 1024;
    int* resident = reinterpret_cast<int*>(cbase + (long)nclus * (per_clu + CW + 1024));
    if ((p.carry_flags & TACO_ATTN_CARRY_WAIT) && tid == 0) atomicAdd(resident, 1);       // this workgroup is on its CU
    float dhc0 = 0.f, dhc1 = 0.f;      // dh carry   (owner lanes: pA == 0, index jA)
    float dcc0 = 0.f, dcc1 = 0.f;      // dctx carry (owner lanes: pA == 0, index jA)
    if (p.s1 < S && pA == 0) {         // chunked launch: state of the later chunk
        if (p.carry_flags & TACO_ATTN_CARRY_WAIT) {
            // this launch was started BESIDE the launch of the later chunk (so that it holds its CUs before that one lets go of
            // its own): the carries arrive as granules tagged with the step they belong to, published by that launch's last step
            const u64* const ptr[4] = {cX + jA, cX + 256 + jA, cX + 512 + jA, cX + 768 + jA};
            float v[4];
            get_g<4>(ptr, TACO_CARRY_TAG + (unsigned)p.s1, v, p.err);
            dhc0 = v[0]; dhc1 = v[1]; dcc0 = v[2]; dcc1 = v[3];
        } else {
            dhc0 = p.dhcarry[(unsigned)rw[0] * 256u + jA]; dhc1 = p.dhcarry[(unsigned)rw[1] * 256u + jA];
            dcc0 = p.dctxcarry[(unsigned)rw[0] * 256u + jA]; dcc1 = p.dctxcarry[(unsigned)rw[1] * 256u + jA];
        }
    }
    // ---- PFW (WLDS variant): the prefetch WAVE.  The saved activations / external gradients of step s-1 come from HBM, and every
    // granule poll of a wave (`s_waitcnt vmcnt(0)` to read the polled value) also waits for that wave's outstanding prefetch loads:
    // with each lane fetching its own values one step ahead, the HBM latency sat on the dependent chain at the next poll (measured:
    // 8.92 us per step, 8.06 with the loads removed).  Wave 7 (threads 448..511) takes part in NO vector gather (those use threads
    // < 448), so after its last poll of the step (the da partials in X1) it fetches everything the workgroup needs for step s-1 --
    // 7 x 64 + 32 + 64 + 2 Ti floats, coalesced, ~13 load instructions -- keeps it in registers while the step runs, and parks it in
    // LDS in front of the step's last barrier; all lanes read their values from there at the top of the next step.
    constexpr int NA = 2 * NT;                       // wave-loads of the 2 Ti alignments (NT = 2: Ti <= 128)
    float wv_v[WLDS ? 7 : 1], wv_p2 = 0.f, wv_q = 0.f, wv_a[WLDS ? NA : 1];
    // (pure loads, no select or branch on a loaded value: anything that consumes one makes the compiler wait for it -- and, loads
    // returning in order, for every load before it -- right here on the chain; clamps on the ADDRESS side, selects at park time)
    // Addresses are RUNNING 32-bit indices (one subtraction per step and tensor): formed per step as o * stride + lane offset the
    // compiler used 64-bit multiply-adds whose addend register pairs overlapped the destinations of the loads just issued, and
    // waited for them (s_waitcnt vmcnt(6) between the loads).
    int wv_first = 0;                                  // the parked step is step 0 (no predecessor state: hp = 0)
    unsigned ix_rj = 0, ix_hc = 0, ix_p2 = 0, ix_a[WLDS ? NA : 1];
    if constexpr (WLDS) {
        const int l = tid & 63, b = l >> 5, c = l & 31, l2 = l & 31;
        const unsigned o = (unsigned)(rw[b] * S + p.s1 - 1);
        ix_rj = o * 256u + 32 * w + c;
        ix_hc = o * 512u + 32 * w + c;
        ix_p2 = (unsigned)(rw[l2 >> 4] * S + p.s1 - 1) * 128u + 16 * w + (l2 & 15);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = min(l + 64 * i, 2 * Ti - 1), row = e >= Ti;
            ix_a[i] = (unsigned)(rw[row] * S + p.s1 - 1) * (unsigned)Ti + (e - row * Ti);
        }
    }
    auto stage_fetch = [&](int sidx) {                 // executed by ONE wave (lane = tid & 63); fetches step sidx, then points at sidx-1
        wv_first = sidx == 0;
        wv_v[0] = p.r[ix_rj]; wv_v[1] = p.u[ix_rj]; wv_v[2] = p.c[ix_rj];
        wv_v[3] = p.hc[ix_hc - (sidx > 0 ? 512u : 0u)];
        wv_v[4] = p.dhc[ix_hc]; wv_v[5] = p.dhc[ix_hc + 256u];
        wv_v[6] = p.p1[ix_rj];
        wv_p2 = p.p2[ix_p2];
        wv_q = p.q[ix_rj];
#pragma unroll
        for (int i = 0; i < (WLDS ? NA : 1); ++i) wv_a[i] = p.align[ix_a[i]];
    };
    auto stage_next = [&]() {                          // every wave keeps the indices in step (uniform code, seven subtractions)
        ix_rj -= 256u; ix_hc -= 512u; ix_p2 -= 128u;
#pragma unroll
        for (int i = 0; i < (WLDS ? NA : 1); ++i) ix_a[i] -= (unsigned)Ti;
    };
    auto stage_park = [&]() {                          // same wave: registers -> LDS (st_l, q_l, a_l)
        const int l = tid & 63;
#pragma unroll
        for (int v = 0; v < (WLDS ? 7 : 1); ++v) st_l[v * 64 + l] = (v == 3 && wv_first) ? 0.f : wv_v[v];
        if (l < 32) st_l[448 + l] = wv_p2;
        q_l[l] = wv_q;
#pragma unroll
        for (int i = 0; i < (WLDS ? NA : 1); ++i) { const int e = l + 64 * i; if (e < 2 * Ti) a_l[e] = wv_a[i]; }
    };
    // one-step-ahead prefetch registers (register variant), primed for s = s1-1
    float pf_a = 0.f, pf_q = 0.f, pf_r[2] = {0, 0}, pf_u[2] = {0, 0}, pf_c[2] = {0, 0}, pf_hp[2] = {0, 0}, pf_dhe[2] = {0, 0},
          pf_dce[2] = {0, 0}, pf_p1[2] = {0, 0}, pf_p2[2] = {0, 0};
    if constexpr (WLDS) {
        if (tid >= AT - 64) {
            stage_fetch(p.s1 - 1);
            stage_park();
        }
        stage_next();
    } else {
        const unsigned sl[2] = {(unsigned)(rw[0] * S + p.s1 - 1), (unsigned)(rw[1] * S + p.s1 - 1)};
        if (2 * Ti <= AT && tid < 2 * Ti) { const int row = tid >= Ti; pf_a = p.align[sl[row] * (unsigned)Ti + (tid - row * Ti)]; }
        if (tid < 64) pf_q = p.q[sl[tid >> 5] * 256u + 32 * w + (tid & 31)];
        if (pA == 0) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                pf_r[b] = p.r[sl[b] * 256u + jA]; pf_u[b] = p.u[sl[b] * 256u + jA]; pf_c[b] = p.c[sl[b] * 256u + jA];
                pf_hp[b] = p.s1 > 1 ? p.hc[(sl[b] - 1u) * 512u + jA] : 0.f;
                pf_dhe[b] = p.dhc[sl[b] * 512u + jA]; pf_dce[b] = p.dhc[sl[b] * 512u + 256u + jA];
                pf_p1[b] = p.p1[sl[b] * 256u + jA];
            }
        }
        if (pB == 0) { pf_p2[0] = p.p2[sl[0] * 128u + jB]; pf_p2[1] = p.p2[sl[1] * 128u + jB]; }
    }
    __syncthreads();
    STAMP_DECL

    for (int s = p.s1 - 1; s >= p.s0; --s) {
        STAMP(0);
        // Re-derive the lane indices from an opaque copy of the thread id every step: otherwise every LDS / granule address
        // built from them is hoisted out of the loop, and with 256 VGPRs full of weights the hoisted copies live in scratch
        // (reloaded on the recurrence chain each step).
        int tq = threadIdx.x;
        asm volatile("" : "+v"(tq));
        const int tid = tq;
        const int cA = tid >> 4, pA = tid & 15, cB = tid >> 5, pB = tid & 31;
        const int jA = 32 * w + cA, jB = 16 * w + cB;
        const unsigned epoch = (unsigned)(S - s);        // step of the PASS (descending s): see the forward kernel
        unsigned so[2] = {(unsigned)(rw[0] * S + s), (unsigned)(rw[1] * S + s)};
        asm volatile("" : "+v"(so[0]), "+v"(so[1]));   // opaque per-step offsets: no precomputed 64-bit addresses kept live
        // ---- this step's saved activations / external gradients were fetched ONE STEP AHEAD (registers), so their HBM
        //      latency is off the critical path; now issue the loads for step s-1
        float r_[2], u_[2], c_[2], hp_[2], dhe[2], dce[2], p1v[2], p2v[2];
        if constexpr (WLDS) {
            // a_l / q_l / st_l were parked by the prefetch wave in front of the previous step's last barrier (or by the prologue)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float* sp = st_l + b * 32 + cA;
                r_[b] = sp[0]; u_[b] = sp[64]; c_[b] = sp[128]; hp_[b] = sp[192]; dhe[b] = sp[256]; dce[b] = sp[320]; p1v[b] = sp[384];
                p2v[b] = st_l[448 + b * 16 + cB];
            }
        } else {
            if (2 * Ti <= AT) { if (tid < 2 * Ti) a_l[tid] = pf_a; }
            else for (int i = tid; i < 2 * Ti; i += AT) { const int row = i >= Ti; a_l[i] = p.align[so[row] * (unsigned)Ti + (i - row * Ti)]; }
            if (tid < 64) q_l[tid] = pf_q;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                r_[b] = pf_r[b]; u_[b] = pf_u[b]; c_[b] = pf_c[b]; hp_[b] = pf_hp[b]; dhe[b] = pf_dhe[b]; dce[b] = pf_dce[b];
                p1v[b] = pf_p1[b]; p2v[b] = pf_p2[b];
            }
        }
        // ================= X1: total dctx (own slice), da partials =================
        if (pA == 0) {
            const float d0 = dce[0] + dcc0, d1 = dce[1] + dcc1;
            dctx_l[cA] = d0; dctx_l[32 + cA] = d1;
            p.dctx[so[0] * 256 + jA] = d0;
            p.dctx[so[1] * 256 + jA] = d1;
        }
        lds_barrier();
        STAMP(1);
        for (int i = tid >> 1; i < 2 * Ti; i += AT / 2) {
            const int half = tid & 1, row = i >= Ti;
            const float* mp = M_l + i * 32 + half * 16;
            const float* gp = dctx_l + row * 32 + half * 16;
            float e = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const float4 mv = *reinterpret_cast<const float4*>(mp + d4 * 4);
                const float4 gv = *reinterpret_cast<const float4*>(gp + d4 * 4);
                e = fmaf(mv.x, gv.x, e); e = fmaf(mv.y, gv.y, e); e = fmaf(mv.z, gv.z, e); e = fmaf(mv.w, gv.w, e);
            }
            e = dpp_add<0xB1>(e);
            if (half == 0) put_gi(xDA, (unsigned)(w * 2 * Ti + i), epoch, e, local);
            float val[4];
            const u64* ptr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pw = half * 4 + k;
                ptr[k] = xDA + (long)(pw == w ? ((w + 1) & 7) : pw) * 2 * Ti + i;
            }
            const u64* const cptr[4] = {ptr[0], ptr[1], ptr[2], ptr[3]};
            get_g<4>(cptr, epoch, val, p.err);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += (half * 4 + k == w) ? e : val[k];
            const float tot2 = dpp_add<0xB1>(sum);
            if (half == 0) de_l[i] = tot2;                        // da[row][t]
        }
        if constexpr (WLDS) {
            // the prefetch wave's last poll of this step is behind it: fetch step s-1 now, park it in front of the last barrier
            if (tid >= AT - 64 && s > p.s0) stage_fetch(s - 1);
            stage_next();
        }
        STAMP(2);
        lds_barrier();
        STAMP(3);
        // ================= X2+X3: softmax backward and dq slice, one pass per wave, no barrier in between =================
        // Wave v works on row v >> 2 and score dims 8 (v & 3) .. +8.  Every wave of a row recomputes de = a * (da - sum a*da)
        // for the whole row (the four waves write the same bits to ep_l and read back only their own writes), then
        // dq[d] = v_d * sum_t de[t] * (1 - tanh^2(K[t,d] + q[d])): lane (d & 7 | tp) sums t = tp, tp + 8, ..., DPP adds combine
        // the 8 tp lanes.
        {
            const int wv = tid >> 6, lane = tid & 63;
            const int row = wv >> 2;
            const float* dr_ = de_l + row * Ti;
            const float* ar = a_l + row * Ti;
            float* er = ep_l + row * Ti;
            float dv[NT], av[NT];                          // Ti <= 64 NT values per row live in registers
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = lane + 64 * i;
                dv[i] = 0.f; av[i] = 0.f;
                if (t < Ti) {
                    dv[i] = dr_[t]; av[i] = ar[t];
                    if (p.da_ext) dv[i] += p.da_ext[so[row] * (unsigned)Ti + t];      // regulariser gradient wrt a_s (off by default)
                    dot = fmaf(av[i], dv[i], dot);
                }
            }
            dot = wave_sum_fast(dot);
            const bool wr = (wv & 3) == 0 && (lane & 7) == w;                // member w stores t = w, w + 8, ... to HBM
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = lane + 64 * i;
                if (t < Ti) { const float de = av[i] * (dv[i] - dot); er[t] = de; if (wr) p.de[so[row] * (unsigned)Ti + t] = de; }
            }
            const int tp = lane & 7, dl = 8 * (wv & 3) + (lane >> 3);
            const float* kr = K_l + row * Ti * 32;
            const float qd = q_l[row * 32 + dl];
            float acc0 = 0.f, acc1 = 0.f;
            int t = tp;
            for (; t + 8 < Ti; t += 16) {
                const float th0 = fast_tanh(kr[t * 32 + ((dl + 4 * tp) & 31)] + qd), th1 = fast_tanh(kr[(t + 8) * 32 + ((dl + 4 * tp) & 31)] + qd);
                acc0 = fmaf(er[t], 1.f - th0 * th0, acc0); acc1 = fmaf(er[t + 8], 1.f - th1 * th1, acc1);
            }
            if (t < Ti) { const float th = fast_tanh(kr[t * 32 + ((dl + 4 * tp) & 31)] + qd); acc0 = fmaf(er[t], 1.f - th * th, acc0); }
            const float x = group_sum<8>(acc0 + acc1) * vd;
            if (tp == 0) {
                const int j = 32 * w + dl;
                dq_l[row * PLEN(256) + PIDX(j)] = x;
                put_gi(xDQ, (unsigned)(row * 256 + j), epoch, x, local);
                p.dq[so[row] * 256 + j] = x;
            }
        }
        STAMP(4);
        gather_vec<32>(xDQ, dq_l, dq_l + PLEN(256), w, epoch, tid, p.err);
        lds_barrier();
        STAMP(5);
        // ================= X4: dhT = dh_ext + carry + dq . Wq^T ; candidate pre-activation gradient =================
        float dhT[2] = {0, 0}, du[2] = {0, 0}, dhd[2] = {0, 0};
        {
            float a0 = 0.f, a1 = 0.f;
            dot2<16>(dq_l, dq_l + PLEN(256), pA * 16, wq, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) {
                dhT[0] = a0 + dhe[0] + dhc0; dhT[1] = a1 + dhe[1] + dhc1;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    du[b] = dhT[b] * (hp_[b] - c_[b]);
                    dhd[b] = dhT[b] * u_[b];
                    const float dcp = dhT[b] * (1.f - u_[b]) * (1.f - c_[b] * c_[b]);
                    dxp_l[b * PLEN(768) + PIDX(512 + jA)] = dcp;
                    put_gi(xDCP, (unsigned)(b * 256 + jA), epoch, dcp, local);
                    p.dxp[so[b] * 768 + 512 + jA] = dcp;
                }
            }
            STAMP(6);
            gather_off<32>(xDCP, dxp_l, dxp_l + PLEN(768), 512, w, epoch, tid, p.err);
        }
        lds_barrier();
        STAMP(7);
        // ================= X5: drh = dcp . Whc^T ; gate pre-activation gradients =================
        float dhp[2] = {0, 0};
        {
            float a0 = 0.f, a1 = 0.f;
            dot2<16>(dxp_l + PIDX(512), dxp_l + PLEN(768) + PIDX(512), pA * 16, wc, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) {
                const float drh[2] = {a0, a1};
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float dgr = drh[b] * hp_[b] * r_[b] * (1.f - r_[b]);
                    const float dgu = du[b] * u_[b] * (1.f - u_[b]);
                    dhp[b] = dhd[b] + drh[b] * r_[b];
                    dxp_l[b * PLEN(768) + PIDX(jA)] = dgr;
                    dxp_l[b * PLEN(768) + PIDX(256 + jA)] = dgu;
                    put_gi(xDGR, (unsigned)(b * 256 + jA), epoch, dgr, local);
                    put_gi(xDGU, (unsigned)(b * 256 + jA), epoch, dgu, local);
                    p.dxp[so[b] * 768 + jA] = dgr; p.dxp[so[b] * 768 + 256 + jA] = dgu;
                }
            }
            STAMP(8);
            gather2<32>(xDGR, xDGU, dxp_l, dxp_l + PLEN(768), 0, 256, w, epoch, tid, p.err);
        }
        lds_barrier();
        STAMP(9);
        // ---- (register variant) issue the loads for step s-1 (consumed at the top of the next iteration)
        if (!WLDS && s > p.s0) {
            const unsigned sn[2] = {so[0] - 1u, so[1] - 1u};
            if (2 * Ti <= AT && tid < 2 * Ti) { const int row = tid >= Ti; pf_a = p.align[sn[row] * (unsigned)Ti + (tid - row * Ti)]; }
            if (tid < 64) pf_q = p.q[sn[tid >> 5] * 256u + 32 * w + (tid & 31)];
            if (pA == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    pf_r[b] = p.r[sn[b] * 256u + jA]; pf_u[b] = p.u[sn[b] * 256u + jA]; pf_c[b] = p.c[sn[b] * 256u + jA];
                    pf_hp[b] = s > 1 ? p.hc[(sn[b] - 1u) * 512u + jA] : 0.f;
                    pf_dhe[b] = p.dhc[sn[b] * 512u + jA]; pf_dce[b] = p.dhc[sn[b] * 512u + 256u + jA];
                                    }
            }
        }
        // ================= X6: dp2pre = (dxp . Wx^T) * (p2 > 0) ;  dh carry = dhp + dg . Whg^T (off the chain) =================
        {
            float b0 = 0.f, b1 = 0.f;
            if constexpr (WLDS) dot2_lds<24>(dxp_l, dxp_l + PLEN(768), pB * 24, W_l, tid, b0, b1);
            else dot2<24>(dxp_l, dxp_l + PLEN(768), pB * 24, wx, b0, b1);
            b0 = lane_reduce<32>(b0); b1 = lane_reduce<32>(b1);
            if (pB == 0) {
                b0 = p2v[0] > 0.f ? b0 : 0.f; b1 = p2v[1] > 0.f ? b1 : 0.f;
                dp2_l[PIDX(jB)] = b0; dp2_l[PLEN(128) + PIDX(jB)] = b1;
                put_gi(xDP2, (unsigned)(jB), epoch, b0, local); put_gi(xDP2, (unsigned)(128 + jB), epoch, b1, local);
                p.dp2[so[0] * 128 + jB] = b0;
                p.dp2[so[1] * 128 + jB] = b1;
            }
            // the recurrent carry is first needed in X4 of the next step: computed while the dp2 granules travel
            float a0 = 0.f, a1 = 0.f;
            dot2<32>(dxp_l, dxp_l + PLEN(768), pA * 32, wg, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) { dhc0 = dhp[0] + a0; dhc1 = dhp[1] + a1; }
            STAMP(10);
            gather_vec<16>(xDP2, dp2_l, dp2_l + PLEN(128), w, epoch, tid, p.err);
        }
        lds_barrier();
        STAMP(11);
        // ================= X7: dp1pre = (dp2pre . W2^T) * (p1 > 0) =================
        {
            float a0 = 0.f, a1 = 0.f;
            if constexpr (WLDS) dot2_lds<8>(dp2_l, dp2_l + PLEN(128), pA * 8, W_l + 6 * AT * 4, tid, a0, a1);
            else dot2<8>(dp2_l, dp2_l + PLEN(128), pA * 8, w2, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) {
                a0 = p1v[0] > 0.f ? a0 : 0.f; a1 = p1v[1] > 0.f ? a1 : 0.f;
                dp1_l[PIDX(jA)] = a0; dp1_l[PLEN(256) + PIDX(jA)] = a1;
                put_gi(xDP1, (unsigned)(jA), epoch, a0, local); put_gi(xDP1, (unsigned)(256 + jA), epoch, a1, local);
                p.dp1[so[0] * 256 + jA] = a0;
                p.dp1[so[1] * 256 + jA] = a1;
            }
            STAMP(12);
            gather_vec<32>(xDP1, dp1_l, dp1_l + PLEN(256), w, epoch, tid, p.err);
        }
        if constexpr (WLDS) {
            // a_l, q_l and the stage were last read in X2+X3 / at the top of THIS step; the barrier below publishes the next step's
            if (tid >= AT - 64 && s > p.s0) stage_park();
        }
        lds_barrier();
        STAMP(13);
        if (!WLDS && s > p.s0) {
            if (pA == 0) { pf_p1[0] = p.p1[(so[0] - 1u) * 256u + jA]; pf_p1[1] = p.p1[(so[1] - 1u) * 256u + jA]; }
            if (pB == 0) { pf_p2[0] = p.p2[(so[0] - 1u) * 128u + jB]; pf_p2[1] = p.p2[(so[1] - 1u) * 128u + jB]; }
        }
        // ================= X8: dctx carry (gradient wrt ctx_{s-1}) = dp1pre . W1c^T =================
        {
            float a0 = 0.f, a1 = 0.f;
            dot2<16>(dp1_l, dp1_l + PLEN(256), pA * 16, w1, a0, a1);
            a0 = lane_reduce<16>(a0); a1 = lane_reduce<16>(a1);
            if (pA == 0) { dcc0 = a0; dcc1 = a1; }
        }
        STAMP(14);
        // (the next iteration's first LDS writes -- a_l, q_l, dctx_l -- are not read by X8: no barrier needed here)
    }
    STAMP_OUT(p.xchg + (long)nclus * (per_clu + CW + 1024) + 1);
    if (p.s0 > 0 && pA == 0) {         // hand the carries to the launch of the previous chunk
        p.dhcarry[(unsigned)rw[0] * 256u + jA] = dhc0; p.dctxcarry[(unsigned)rw[0] * 256u + jA] = dcc0;
        p.dhcarry[(unsigned)rw[1] * 256u + jA] = dhc1; p.dctxcarry[(unsigned)rw[1] * 256u + jA] = dcc1;
        if (p.carry_flags & TACO_ATTN_CARRY_POST) {         // ... which may already be resident and polling (agent scope: any placement)
            const unsigned tag = TACO_CARRY_TAG + (unsigned)p.s0;
            put_granule(cX + jA, tag, dhc0); put_granule(cX + 256 + jA, tag, dhc1);
            put_granule(cX + 512 + jA, tag, dcc0); put_granule(cX + 768 + jA, tag, dcc1);
        }
    }
}

// Hoisted reductions over the S steps (fully parallel, thread = feature d):
//   dMEM[n,t,d]  = sum_s a[n,s,t] * dctx[n,s,d]                                   (gradient through the attention values)
//   dKEYS[n,t,d] = v[d] * sum_s de[n,s,t] * (1 - tanh^2(keys[n,t,d] + q[n,s,d]))  (tanh tile recomputed, never stored)
//   dvpart[n*Ti+t, d] = sum_s de[n,s,t] * tanh(...)                               (attention_v gradient partials)
// One workgroup per (n, tile of HT positions t): q[n,s,:] and dctx[n,s,:] are loaded once per step and used for all HT positions (one
// workgroup per (n,t) re-read them T_in times from L2: 1 GB per launch at C2, 90 us alone and 290 us beside the weight-gradient flood,
// on the critical path between the attention BPTT and the encoder backward).
#define HT 8
__global__ __launch_bounds__(256) void attn_hoisted_bwd_k(const float* __restrict__ keys, const float* __restrict__ q,
                                                         const float* __restrict__ align, const float* __restrict__ de,
                                                         const float* __restrict__ dctx, const float* __restrict__ v,
                                                         float* __restrict__ dkeys, float* __restrict__ dmem,
                                                         float* __restrict__ dvpart, int S, int Ti) {
    const int n = blockIdx.y, t0 = blockIdx.x * HT, d = threadIdx.x;
    float kd[HT], am[HT], ak[HT], av[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j) {
        const int t = min(t0 + j, Ti - 1);                 // a ragged last tile recomputes position Ti-1 and does not store it
        kd[j] = keys[((long)n * Ti + t) * 256 + d];
        am[j] = 0.f; ak[j] = 0.f; av[j] = 0.f;
    }
    const float* ap = align + (long)n * S * Ti;
    const float* ep = de + (long)n * S * Ti;
    const float* qp = q + (long)n * S * 256 + d;
    const float* cp = dctx + (long)n * S * 256 + d;
#pragma unroll 2
    for (int s = 0; s < S; ++s) {
        const float qv = qp[(long)s * 256], cv = cp[(long)s * 256];
        const float* as = ap + (long)s * Ti;
        const float* es = ep + (long)s * Ti;
#pragma unroll
        for (int j = 0; j < HT; ++j) {
            const int t = min(t0 + j, Ti - 1);
            const float a = as[t], e = es[t];              // wave-uniform addresses: scalar loads
            const float th = fast_tanh(kd[j] + qv);
            am[j] = fmaf(a, cv, am[j]);
            ak[j] = fmaf(e, 1.f - th * th, ak[j]);
            av[j] = fmaf(e, th, av[j]);
        }
    }
    const float vd = v[d];
#pragma unroll
    for (int j = 0; j < HT; ++j) {
        const int t = t0 + j;
        if (t < Ti) {
            const long nt = (long)n * Ti + t;
            dmem[nt * 256 + d] = am[j];
            dkeys[nt * 256 + d] = ak[j] * vd;
            dvpart[nt * 256 + d] = av[j];
        }
    }
}

static size_t attn_cluster_bwd_smem(int Ti, bool wlds = false) {
    size_t f = 2 * PLEN(256) + 2 * PLEN(768) + 2 * PLEN(128) + 2 * PLEN(256) + 64 + 64 + 3 * ((2 * Ti + 3) & ~3) + 512 +
               (size_t)4 * Ti * 32 + (wlds ? 8 * AT * 4 + 512 : 0);
    return f * sizeof(float);
}

// which BPTT kernel a (N, Ti) launch runs: 0 = shape not held by the cluster path (per-step kernels), 1 = attn_cluster_bwd_k<true>
// (prenet-gradient weight slices parked in LDS; fits while Ti <= 148), 2 = attn_cluster_bwd_k<false> (all weights in registers; the
// key / memory tiles of long inputs need the LDS).  TACO_ATTN_NO_WLDS=1 forces variant 2 (tests).
extern "C" int taco_attn_cluster_bwd_variant(int N, int Ti) {
    if (!taco_attn_cluster_supported(N, Ti)) return 0;
    const char* e = getenv("TACO_ATTN_NO_WLDS");
    const bool no_wlds = e && e[0] && e[0] != '0';
    return (!no_wlds && attn_cluster_bwd_smem(Ti, true) <= 160 * 1024) ? 1 : 2;
}

extern "C" int taco_attn_cluster_bwd_xchg_slots(int N, int Ti) {
    // + CW placement granules and 1024 carry granules per cluster; + 1 residency counter; + 16 diagnostic stamp slots at the end
    return ((N + 1) / 2) * (CW * 2 * Ti + 2 * 256 * 5 + 2 * 128 + CW + 1024) + 1 + 16;
}

extern "C" int taco_attn_bwd_resident_slot(int N, int Ti) {
    return ((N + 1) / 2) * (CW * 2 * Ti + 2 * 256 * 5 + 2 * 128 + CW + 1024);
}
extern "C" int taco_attn_bwd_workgroups(int N) { return CW * ((N + 1) / 2); }

int attn_cluster_bwd_launch(const AttnCluB& p, float* dkeys, float* dmem, float* dvpart, hipStream_t st) {
    static DevMask attr_set[4] = {{0}, {0}, {0}, {0}};
    const void* ks[4] = {(const void*)attn_cluster_bwd_k<false, 2>, (const void*)attn_cluster_bwd_k<true, 2>,
                         (const void*)attn_cluster_bwd_k<false, 5>, (const void*)attn_cluster_bwd_k<true, 5>};
    for (int i = 0; i < 4; ++i)
        if (ensure_dyn_lds(ks[i], 160 * 1024, attr_set[i]) != TACO_OK) return TACO_EINVAL;
    if (attn_cluster_bwd_smem(p.Ti, false) > 160 * 1024) return TACO_EINVAL;
    if (p.s1 == p.S && hipMemsetAsync(p.xchg, 0, (size_t)(taco_attn_cluster_bwd_xchg_slots(p.N, p.Ti) - 16) * sizeof(u64), st) != hipSuccess)
        return TACO_EINVAL;
    AttnCluB q = p;
    q.xcd_local = xcd_local_allowed();
    const dim3 grid(CW * ((p.N + 1) / 2));
    const bool wlds = taco_attn_cluster_bwd_variant(p.N, p.Ti) == 1, small = p.Ti <= 128;
    const size_t smem = attn_cluster_bwd_smem(p.Ti, wlds);
    if (wlds && small) hipLaunchKernelGGL((attn_cluster_bwd_k<true, 2>), grid, dim3(AT), smem, st, q);
    else if (wlds) hipLaunchKernelGGL((attn_cluster_bwd_k<true, 5>), grid, dim3(AT), smem, st, q);
    else if (small) hipLaunchKernelGGL((attn_cluster_bwd_k<false, 2>), grid, dim3(AT), smem, st, q);
    else hipLaunchKernelGGL((attn_cluster_bwd_k<false, 5>), grid, dim3(AT), smem, st, q);
    if (p.s0 == 0 && !(p.carry_flags & TACO_ATTN_NO_REDUCE))       // all chunks done: reduce over the S steps
        return attn_cluster_bwd_reduce(p, dkeys, dmem, dvpart, st);
    TACO_RETURN_LAST();
}

int attn_cluster_bwd_reduce(const AttnCluB& p, float* dkeys, float* dmem, float* dvpart, hipStream_t st) {
    hipLaunchKernelGGL(attn_hoisted_bwd_k, dim3((p.Ti + HT - 1) / HT, p.N), dim3(256), 0, st, p.keys, p.q, p.align, p.de, p.dctx, p.v,
                       dkeys, dmem, dvpart, p.S, p.Ti);
    TACO_RETURN_LAST();
}
