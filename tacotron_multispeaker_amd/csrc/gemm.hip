// fp32 MFMA implicit-GEMM kernels for gfx950: dense layers and 'same'-padded conv1d (incl. the fused
// CBHG conv bank) forward, input-gradient and weight-gradient.
//
// Reference ops replaced: tf.layers.dense / tf.layers.conv1d at models/modules.py:10,79-89,95-100 and
// models/tacotron.py:101, plus their TF-generated gradients (optimizer.compute_gradients, tacotron.py:187).
//
// Layout: activations are channel-last [N*T, C] row-major exactly like the reference tensors [N,T,C];
// conv kernels are TF layout [k, Cin, Cout].  A conv tap j is a GEMM over rows shifted by j-pad_left
// inside each length-T sequence (zeros outside) -> no im2col buffer is ever materialised.
//
// Math: v_mfma_f32_32x32x2_f32 (exact fp32 fma chains, 256 FLOP/clk/CU = 157 TFLOP/s chip peak).
// 4 waves per workgroup in a 2x2 arrangement.  Two generations live here:
//   v1 (conv_gemm_nn/nt/tn): tiles staged global -> VGPR -> LDS with register double buffering; generic 64-bit addressing;
//      kept as the fallback for operands beyond 2 GiB.
//   v2 (conv_gemm_nn2/nt2/tn2): tiles staged global -> LDS directly (buffer_load_dwordx4 ... lds), loop-invariant lane
//      offsets + scalar K walk, statically unrolled ring of LDS stages: no VALU work in the K loop, which on gfx950 is what
//      limits an fp32 MFMA kernel (profiles/r01_pmc_gemm_v1_v2.md).  The dispatchers below pick v2 whenever it applies.
#include "common.hpp"
#include <stdlib.h>

constexpr int KPAD = 4;      // v1 kernels: k-contiguous LDS tiles are padded by 4 floats per row
struct ConvGemm {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;        // NN/NT: M rows, N output cols, K reduction per tap.  TN: M reduction rows, K x N output
    int lda, ldb, ldc;
    int T;              // sequence length (rows m = n*T + t) for the tap-shift mask
    int kw_lo, kw_hi;   // conv widths iterated (plain: kw_lo == kw_hi == k; dense: 1; bank dX: 1..Kbank)
    int bank;           // conv-bank addressing (weights packed [sum k][Cin][cpb], outputs concatenated)
    int cpb;            // channels per bank conv (128)
    int act;            // NN epilogue activation
    int accumulate;     // NN/NT: C += result
    int splitk;         // TN: reduction split
    int shift0;         // TN: extra row shift of X (dW of recurrent weights: X = H shifted by one step)
    float* dbias;       // TN (dense problems, v2 64x64x32 kernel): optional bias gradient dbias[n] += sum_m dY[m, n], summed from the dY tiles in LDS
    double* bn_stat;    // NN: optional batch-norm sums of the OUTPUT (after bias + activation): TACO_BN_REPL replicas of [sum | sum of squares | -] x N
    int rb_len, rb_stride, rb_off;   // NN/NT row blocking: logical row m -> physical row (m / rb_len) * rb_stride + rb_off + m % rb_len
                                     // (a chunk of steps [s0, s0+rb_len) of [N,S,*] tensors seen as one [N*rb_len, *] matrix); 0 = identity
};

__device__ __forceinline__ long rowmap(const ConvGemm& p, int m) {
    return p.rb_len ? (long)(m / p.rb_len) * p.rb_stride + p.rb_off + (m % p.rb_len) : (long)m;
}
// position of logical row m inside its length-T sequence (tap-shift mask).  Row-blocked conv (frames [rb_off, rb_off + rb_len) of
// every sequence, taco_conv_rows_fwd): the position in the FULL sequence; row-blocked dense layers have no taps (mask always true).
__device__ __forceinline__ int seqpos(const ConvGemm& p, int m) {
    return (p.rb_len && p.kw_hi > 1) ? p.rb_off + m % p.rb_len : m % p.T;
}

// ---- MFMA inner product on one staged K-tile -------------------------------------------------------
// A tile: k-contiguous [BM][BK+4] (AKC) or k-strided [BK][BM];  B tile likewise with BN.
// k permutation: within an 8-wide chunk, MFMA q of lane-half h consumes k = 8*kk + 4*h + q for BOTH
// operands, so a k-contiguous operand is fetched with one ds_read_b128 per 4 MFMAs.
template <int BM, int BN, int BK, bool AKC, bool BKC>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Bs,
                                         f32x16 (&acc)[BM / 64][BN / 64], int wm, int wn, int lane) {
    constexpr int MI = BM / 64, NI = BN / 64, LDK = BK + KPAD;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
        float a[MI][4], b[NI][4];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * (BM / 2) + mi * 32 + i;
            if (AKC) {
                const float4 v = *reinterpret_cast<const float4*>(&As[row * LDK + kk * 8 + 4 * h]);
                a[mi][0] = v.x; a[mi][1] = v.y; a[mi][2] = v.z; a[mi][3] = v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) a[mi][q] = As[(kk * 8 + 4 * h + q) * BM + row];
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = wn * (BN / 2) + ni * 32 + i;
            if (BKC) {
                const float4 v = *reinterpret_cast<const float4*>(&Bs[col * LDK + kk * 8 + 4 * h]);
                b[ni][0] = v.x; b[ni][1] = v.y; b[ni][2] = v.z; b[ni][3] = v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) b[ni][q] = Bs[(kk * 8 + 4 * h + q) * BN + col];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][q], b[ni][q], acc[mi][ni], 0, 0, 0);
    }
}

// ---- XCD-aware tile order (speed only) -------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Give every XCD a contiguous run of "virtual"
// tile ids ordered column-tile-fastest, so the column tiles that re-read the same A rows (and the taps of one row tile)
// hit in ONE L2 instead of fetching the rows once per XCD (rocprofv3 FETCH_SIZE of the post-net proj_1 conv: 424 MB for
// 87 MB of operands before this remap).  Bijective for any grid size.
__device__ __forceinline__ void xcd_tile(int& bx, int& by) {
    const int nbx = gridDim.x, nby = gridDim.y;
    const int total = nbx * nby, lin = blockIdx.y * nbx + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = lin & 7, slot = lin >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bx = v / nby; by = v - bx * nby;
}

// ---- tile loaders (global -> registers, registers -> LDS) --------------------------------------------
// k-contiguous tile [ROWS][BK]: thread owns float4 #(tid + v*256); BK/4 float4 per row
template <int ROWS, int BK>
struct KC {
    static constexpr int NV = ROWS * BK / 4 / 256;
    static constexpr int TPR = BK / 4;
    static constexpr int LDK = BK + KPAD;
    __device__ static __forceinline__ int row(int tid, int v) { return (tid + v * 256) / TPR; }
    __device__ static __forceinline__ int c4(int tid, int v) { return (tid + v * 256) % TPR; }
    __device__ static __forceinline__ void store(float* s, int tid, const float4 (&r)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            *reinterpret_cast<float4*>(&s[row(tid, v) * LDK + c4(tid, v) * 4]) = r[v];
    }
};
// k-strided tile [BK][COLS]
template <int COLS, int BK>
struct KS {
    static constexpr int NV = COLS * BK / 4 / 256;
    static constexpr int TPR = COLS / 4;
    __device__ static __forceinline__ int row(int tid, int v) { return (tid + v * 256) / TPR; }
    __device__ static __forceinline__ int c4(int tid, int v) { return (tid + v * 256) % TPR; }
    __device__ static __forceinline__ void store(float* s, int tid, const float4 (&r)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            *reinterpret_cast<float4*>(&s[row(tid, v) * COLS + c4(tid, v) * 4]) = r[v];
    }
};

__device__ __forceinline__ float4 ld4(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- epilogue: 32x32 accumulator -> C (row-major), lane = column, registers = rows --------------------
template <int BM, int BN>
__device__ __forceinline__ void epilogue_store(const ConvGemm& p, f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0,
                                               int wm, int wn, int lane) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < BN / 64; ++ni) {
        const int col = n0 + wn * (BN / 2) + ni * 32 + i;
        const bool cok = col < p.N;
        if (!cok && !p.bn_stat) continue;
        const float bv = (cok && p.bias) ? p.bias[col] : 0.0f;
        float s0 = 0.f, s1 = 0.f;           // batch-norm sums of this lane's 16 (32) rows of the column
#pragma unroll
        for (int mi = 0; mi < BM / 64; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (cok && row < p.M) {
                    float* c = p.C + rowmap(p, row) * p.ldc + col;
                    float v = apply_act(acc[mi][ni][r] + bv, p.act);
                    if (p.accumulate) v += *c;
                    *c = v;
                    s0 += v; s1 = fmaf(v, v, s1);
                }
            }
        }
        if (p.bn_stat) {
            // tf.layers.batch_normalization statistics (modules.py:101) of the tensor this GEMM writes, straight from the accumulators:
            // the two lane halves hold the other rows of the same column; one pair of double atomics per column and wave
            s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
            if (h == 0 && cok) {
                double* ds = p.bn_stat + (long)((m0 / BM) % TACO_BN_REPL) * 3 * p.N;
                atomicAdd(ds + col, (double)s0);
                atomicAdd(ds + p.N + col, (double)s1);
            }
        }
    }
}

// =====================================================================================================
// NN: C[m,n] = act(bias[n] + sum_j sum_c A[m + j - pl, c] * W[j][c][n])        (forward dense/conv/bank)
// =====================================================================================================
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void conv_gemm_nn(ConvGemm p) {
    using TA = KC<BM, BK>;
    using TB = KS<BN, BK>;
    __shared__ __attribute__((aligned(16))) float smem[2][BM * (BK + KPAD) + BK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.bank) by = gridDim.y - 1 - blockIdx.y;                         // bank: widest convs first
    else xcd_tile(bx, by);
    const int m0 = bx * BM;
    const int n0 = by * BN;

    int kw = p.kw_lo, ldb = p.ldb, nloc0 = n0, nlim = p.N;
    const float* Bb = p.B;
    if (p.bank) {
        kw = 1 + n0 / p.cpb;
        Bb = p.B + (long)p.K * p.cpb * ((kw - 1) * kw / 2);
        ldb = p.cpb;
        nloc0 = n0 - (kw - 1) * p.cpb;
        nlim = p.cpb;
    }
    const int pl = (kw - 1) / 2;
    const int ksteps = (p.K + BK - 1) / BK;
    const int nsteps = kw * ksteps;

    int tpos[TA::NV];
    long arow[TA::NV];
#pragma unroll
    for (int v = 0; v < TA::NV; ++v) { tpos[v] = seqpos(p, m0 + TA::row(tid, v)); arow[v] = rowmap(p, m0 + TA::row(tid, v)); }

    f32x16 acc[BM / 64][BN / 64];
#pragma unroll
    for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
        for (int ni = 0; ni < BN / 64; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    float4 ra[TA::NV], rb[TB::NV];
    auto gload = [&](int s) {
        const int j = s / ksteps, kc = s - j * ksteps;
        const int shift = j - pl;
#pragma unroll
        for (int v = 0; v < TA::NV; ++v) {
            const int grow = m0 + TA::row(tid, v);
            const int k = kc * BK + TA::c4(tid, v) * 4;
            const bool ok = grow < p.M && k < p.K && (unsigned)(tpos[v] + shift) < (unsigned)p.T;
            ra[v] = ld4(p.A + (arow[v] + shift) * p.lda + k, ok);
        }
        const float* Wj = Bb + (long)j * p.K * ldb;
#pragma unroll
        for (int v = 0; v < TB::NV; ++v) {
            const int k = kc * BK + TB::row(tid, v);
            const int n = nloc0 + TB::c4(tid, v) * 4;
            rb[v] = ld4(Wj + (long)k * ldb + n, k < p.K && n < nlim);
        }
    };

    gload(0);
    TA::store(smem[0], tid, ra);
    TB::store(smem[0] + BM * (BK + KPAD), tid, rb);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        mma_tile<BM, BN, BK, true, false>(smem[cur], smem[cur] + BM * (BK + KPAD), acc, wm, wn, lane);
        if (s + 1 < nsteps) {
            TA::store(smem[cur ^ 1], tid, ra);
            TB::store(smem[cur ^ 1] + BM * (BK + KPAD), tid, rb);
        }
        __syncthreads();
    }
    epilogue_store<BM, BN>(p, acc, m0, n0, wm, wn, lane);
}

// =====================================================================================================
// NN v2: K-loop with (almost) no VALU work.  rocprofv3 PMC on v1 showed SQ_VALU_MFMA_COEXEC_CYCLES == 0 and ~3.7 non-MFMA
// VALU instructions per MFMA (address arithmetic, zero fills): on gfx950 every such instruction takes issue time away
// from the fp32 MFMA stream of the SIMD, which capped v1 at ~65 % MFMA utilisation.  Here
//   * tiles go global -> LDS with buffer_load_dwordx4 ... lds (no VGPR staging, no ds_write, no zero-fill moves);
//   * per-lane byte offsets are computed once per conv tap; the K walk only bumps the scalar soffset;
//   * masked elements (conv padding rows, ragged edges) use an out-of-range voffset -> the buffer unit returns 0;
//   * LDS stages are unrolled statically so every ds_read address is a loop-invariant VGPR + immediate.
// A tile [BM][32] k-contiguous, unpadded, 16-byte chunks XOR-swizzled by (row >> 1) & 7 (applied on the global side: each
// lane picks its source chunk, the LDS side of the instruction is lane-contiguous).  B tile [32][BN] k-strided.
// Requires operand byte sizes < 2 GiB (32-bit buffer offsets); the launcher falls back to v1 otherwise.
// =====================================================================================================
typedef __attribute__((address_space(3))) void* lptr_t;
#define TACO_OOB 0x80000000u

// (the resource type is only named inside __device__ helpers: a __global__ body that declares it loses its host stub)
__device__ __forceinline__ void buf_load_lds16(const void* base, float* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000),
                                             (lptr_t)lds, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int BK>
struct MmaSw {
    static constexpr int MI = BM / 64, NI = BN / 64, KK = BK / 8;
    int aaddr[MI][KK];     // float index of this lane's A chunk for kk = 0..KK-1
    int baddr[NI];         // float index of this lane's B column at k = 4h
    __device__ __forceinline__ void init(int wm, int wn, int lane) {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * (BM / 2) + mi * 32 + i;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) aaddr[mi][kk] = row * BK + (((kk * 2 + h) ^ ((row / (64 / BK)) & (BK / 4 - 1))) << 2);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) baddr[ni] = 4 * h * BN + wn * (BN / 2) + ni * 32 + i;
    }
    __device__ __forceinline__ void run(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[MI][NI]) const {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            float a[MI][4], b[NI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const float4 v = *reinterpret_cast<const float4*>(&As[aaddr[mi][kk]]);
                a[mi][0] = v.x; a[mi][1] = v.y; a[mi][2] = v.z; a[mi][3] = v.w;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) b[ni][q] = Bs[baddr[ni] + (kk * 8 + q) * BN];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][q], b[ni][q], acc[mi][ni], 0, 0, 0);
        }
    }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int BM, int BN, int BK, int STAGES, bool KTAIL>
__global__ __launch_bounds__(256) void conv_gemm_nn2(ConvGemm p) {
    constexpr int ASZ = BM * BK, BSZ = BK * BN, SSZ = ASZ + BSZ;
    constexpr int CPR = BK / 4;             // 16-byte chunks per A row
    constexpr int RPI = 64 / CPR;           // A rows per wave-instruction (1 KiB)
    constexpr int NVA = BM / (RPI * 4);
    constexpr int RB = 256 / BN;            // B k rows per wave-instruction
    constexpr int NVB = BK / (RB * 4);
    constexpr int LPR = BN / 4;             // lanes per B k-row
    constexpr int LPS = NVA + NVB;          // loads per step per wave
    static_assert(NVA >= 1 && NVB >= 1 && (STAGES == 2 || STAGES == 3), "tile config");
    __shared__ __attribute__((aligned(16))) float smem[STAGES][SSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.bank) by = gridDim.y - 1 - blockIdx.y;                         // bank: widest convs first
    else xcd_tile(bx, by);
    const int m0 = bx * BM, n0 = by * BN;

    int kw = p.kw_lo, ldb = p.ldb, nloc0 = n0, nlim = p.N;
    const float* Bb = p.B;
    if (p.bank) {
        kw = 1 + n0 / p.cpb;
        Bb = p.B + (long)p.K * p.cpb * ((kw - 1) * kw / 2);
        ldb = p.cpb;
        nloc0 = n0 - (kw - 1) * p.cpb;
        nlim = p.cpb;
    }
    const int pl = (kw - 1) / 2;
    const int ksteps = (p.K + BK - 1) / BK;
    // blockIdx.z owns a contiguous share of the flattened (tap, k chunk) steps (split-K for few-tile / long-reduction shapes:
    // partial sums are atomically added into zeroed C, bias + activation are applied by bias_act_k afterwards)
    const int total = kw * ksteps;
    const int per = (total + p.splitk - 1) / p.splitk;
    const int s_begin = blockIdx.z * per;
    const int nsteps = min(total, s_begin + per) - s_begin;
    if (nsteps <= 0) return;

    // per-lane constants
    int tpos[NVA], acol[NVA];
    unsigned arow[NVA];
    bool arok[NVA];
#pragma unroll
    for (int v = 0; v < NVA; ++v) {
        const int r = (v * 4 + wave) * RPI + lane / CPR;
        tpos[v] = seqpos(p, m0 + r);
        arow[v] = (unsigned)rowmap(p, m0 + r);
        arok[v] = m0 + r < p.M;
        acol[v] = ((lane % CPR) ^ ((r / (64 / BK)) & (CPR - 1))) * 4;    // k offset (floats) of the chunk this lane fetches
    }
    unsigned voa[NVA], voa_last[NVA], vob[NVB], vob_last[NVB];
    const int klast = (ksteps - 1) * BK;
#pragma unroll
    for (int v = 0; v < NVB; ++v) {
        const int kl = (v * 4 + wave) * RB + lane / LPR;
        const int n = nloc0 + (lane % LPR) * 4;
        vob[v] = n < nlim ? (unsigned)(kl * ldb + n) * 4u : TACO_OOB;
        vob_last[v] = klast + kl < p.K ? vob[v] : TACO_OOB;
    }
    auto tap_offsets = [&](int j) {
        const int shift = j - pl;
#pragma unroll
        for (int v = 0; v < NVA; ++v) {
            const bool ok = arok[v] && (unsigned)(tpos[v] + shift) < (unsigned)p.T;
            voa[v] = ok ? ((arow[v] + shift) * (unsigned)p.lda + acol[v]) * 4u : TACO_OOB;
            voa_last[v] = klast + acol[v] < p.K ? voa[v] : TACO_OOB;
        }
    };
    // next tile to issue: tap i_j, k chunk i_kc; scalar byte offsets soa (A) / sob (B) follow them
    const int bstep = BK * ldb * 4, btap = p.K * ldb * 4;
    int i_j = s_begin / ksteps, i_kc = s_begin - i_j * ksteps, issued = 0;
    int soa = i_kc * BK * 4, sob = __builtin_amdgcn_readfirstlane(i_j * btap + i_kc * bstep);
    bool need_tap = true;
    auto issue = [&](float* stage) {
        if (need_tap) { tap_offsets(i_j); need_tap = false; }
        const bool last = KTAIL && i_kc == ksteps - 1;
#pragma unroll
        for (int v = 0; v < NVA; ++v)
            buf_load_lds16(p.A, stage + (v * 4 + wave) * RPI * BK, last ? voa_last[v] : voa[v], soa);
#pragma unroll
        for (int v = 0; v < NVB; ++v)
            buf_load_lds16(Bb, stage + ASZ + (v * 4 + wave) * RB * BN, last ? vob_last[v] : vob[v], sob);
        ++i_kc; ++issued; soa += BK * 4; sob += bstep;
        if (i_kc == ksteps) { i_kc = 0; ++i_j; soa = 0; sob = __builtin_amdgcn_readfirstlane(i_j * btap); need_tap = true; }
    };

    f32x16 acc[BM / 64][BN / 64];
#pragma unroll
    for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
        for (int ni = 0; ni < BN / 64; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    MmaSw<BM, BN, BK> mm;
    mm.init(wm, wn, lane);

    // Ring of STAGES static LDS stages, tiles s+1 .. s+STAGES-1 in flight while tile s is multiplied.
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nsteps) issue(smem[t]);
    if (STAGES == 3 && nsteps >= 2) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
    __syncthreads();
    const int ngroups = nsteps / STAGES;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int u = 0; u < STAGES; ++u) {
            const bool more = issued < nsteps;
            if (more) issue(smem[(u + STAGES - 1) % STAGES]);
            mm.run(smem[u], smem[u] + ASZ, acc);
            if (STAGES == 3 && more) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            __syncthreads();
        }
    }
    const int rem = nsteps - ngroups * STAGES;
    if (rem >= 1) {
        mm.run(smem[0], smem[0] + ASZ, acc);
        if (STAGES == 3 && rem == 2) {
            wait_vmcnt<0>();
            __syncthreads();
            mm.run(smem[1], smem[1] + ASZ, acc);
        }
    }
    if (p.splitk == 1) { epilogue_store<BM, BN>(p, acc, m0, n0, wm, wn, lane); return; }
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < BN / 64; ++ni) {
        const int col = n0 + wn * (BN / 2) + ni * 32 + i;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.M) atomicAdd(p.C + rowmap(p, row) * p.ldc + col, acc[mi][ni][r]);
            }
    }
}

// C[m, n] = act(C[m, n] + bias[n]) in place (second pass of a split-K forward GEMM)
__global__ __launch_bounds__(256) void bias_act_k(ConvGemm p) {
    const long total = (long)p.M * p.N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int m = (int)(e / p.N), n = (int)(e - (long)m * p.N);
        float* c = p.C + rowmap(p, m) * p.ldc + n;
        *c = apply_act(*c + (p.bias ? p.bias[n] : 0.0f), p.act);
    }
}


// =====================================================================================================
// NT: C[m,c] (+)= sum_kw sum_j sum_n dY[m - (j - pl), aoff(kw) + n] * W_kw[j][c][n]      (input gradient)
//     here p.N = Cin (output cols), p.K = Cout per conv (reduction per tap)
// =====================================================================================================
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void conv_gemm_nt(ConvGemm p) {
    using TA = KC<BM, BK>;
    using TB = KC<BN, BK>;
    __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * (BK + KPAD)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * BM, n0 = by * BN;
    const int ksteps = (p.K + BK - 1) / BK;

    int tpos[TA::NV];
    long arow[TA::NV];
#pragma unroll
    for (int v = 0; v < TA::NV; ++v) { tpos[v] = seqpos(p, m0 + TA::row(tid, v)); arow[v] = rowmap(p, m0 + TA::row(tid, v)); }

    f32x16 acc[BM / 64][BN / 64];
#pragma unroll
    for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
        for (int ni = 0; ni < BN / 64; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // flattened (kw, j, kc) iteration state for the NEXT tile to load; blockIdx.z owns a contiguous share of the steps
    int l_kw = p.kw_lo, l_j = 0, l_kc = 0;
    int total = 0;
    for (int kw = p.kw_lo; kw <= p.kw_hi; ++kw) total += kw * ksteps;
    const int per = (total + p.splitk - 1) / p.splitk;
    const int s_begin = blockIdx.z * per, s_end = min(total, s_begin + per);
    if (s_begin >= s_end) return;
    {   // advance the iteration state to s_begin
        int skip = s_begin;
        while (skip >= l_kw * ksteps) { skip -= l_kw * ksteps; ++l_kw; }
        l_j = skip / ksteps; l_kc = skip - l_j * ksteps;
    }
    const int nsteps = s_end - s_begin;

    float4 ra[TA::NV], rb[TB::NV];
    auto gload = [&]() {
        const int kw = l_kw, j = l_j, kc = l_kc;
        const int shift = -(j - (kw - 1) / 2);
        const int aoff = p.bank ? (kw - 1) * p.cpb : 0;
        const int ldb = p.bank ? p.cpb : p.ldb;
        const float* Wj = (p.bank ? p.B + (long)p.N * p.cpb * ((kw - 1) * kw / 2) : p.B) + (long)j * p.N * ldb;
#pragma unroll
        for (int v = 0; v < TA::NV; ++v) {
            const int grow = m0 + TA::row(tid, v);
            const int k = kc * BK + TA::c4(tid, v) * 4;
            const bool ok = grow < p.M && k < p.K && (unsigned)(tpos[v] + shift) < (unsigned)p.T;
            ra[v] = ld4(p.A + (arow[v] + shift) * p.lda + aoff + k, ok);
        }
#pragma unroll
        for (int v = 0; v < TB::NV; ++v) {
            const int c = n0 + TB::row(tid, v);
            const int k = kc * BK + TB::c4(tid, v) * 4;
            rb[v] = ld4(Wj + (long)c * ldb + k, c < p.N && k < p.K);
        }
        if (++l_kc == ksteps) { l_kc = 0; if (++l_j == l_kw) { l_j = 0; ++l_kw; } }
    };

    gload();
    TA::store(smem[0], tid, ra);
    TB::store(smem[0] + BM * (BK + KPAD), tid, rb);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsteps) gload();
        mma_tile<BM, BN, BK, true, true>(smem[cur], smem[cur] + BM * (BK + KPAD), acc, wm, wn, lane);
        if (s + 1 < nsteps) {
            TA::store(smem[cur ^ 1], tid, ra);
            TB::store(smem[cur ^ 1] + BM * (BK + KPAD), tid, rb);
        }
        __syncthreads();
    }
    if (p.splitk == 1) { epilogue_store<BM, BN>(p, acc, m0, n0, wm, wn, lane); return; }
    const int i = lane & 31, h = lane >> 5;         // split-K: partial sums are atomically added into (zeroed) C
#pragma unroll
    for (int ni = 0; ni < BN / 64; ++ni) {
        const int col = n0 + wn * (BN / 2) + ni * 32 + i;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.M) atomicAdd(p.C + rowmap(p, row) * p.ldc + col, acc[mi][ni][r]);
            }
    }
}

// =====================================================================================================
// NT v2 (input gradient), same construction as NN v2: both operands k-contiguous (dY rows, W[j][c][:] rows), both staged
// unpadded + XOR-swizzled via buffer_load ... lds, VALU-free K loop, ring of STAGES static LDS stages.  blockIdx.z owns a
// contiguous share of the flattened (kw, tap, k chunk) steps (split-K, atomically accumulated) like v1.
// =====================================================================================================
template <int BM, int BN, int BK>
struct MmaSw2 {
    static constexpr int MI = BM / 64, NI = BN / 64, KK = BK / 8;
    int aaddr[MI][KK], baddr[NI][KK];
    __device__ __forceinline__ void init(int wm, int wn, int lane) {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * (BM / 2) + mi * 32 + i;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) aaddr[mi][kk] = row * BK + (((kk * 2 + h) ^ ((row / (64 / BK)) & (BK / 4 - 1))) << 2);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = wn * (BN / 2) + ni * 32 + i;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) baddr[ni][kk] = col * BK + (((kk * 2 + h) ^ ((col / (64 / BK)) & (BK / 4 - 1))) << 2);
        }
    }
    __device__ __forceinline__ void run(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[MI][NI]) const {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            float a[MI][4], b[NI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const float4 v = *reinterpret_cast<const float4*>(&As[aaddr[mi][kk]]);
                a[mi][0] = v.x; a[mi][1] = v.y; a[mi][2] = v.z; a[mi][3] = v.w;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const float4 v = *reinterpret_cast<const float4*>(&Bs[baddr[ni][kk]]);
                b[ni][0] = v.x; b[ni][1] = v.y; b[ni][2] = v.z; b[ni][3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][q], b[ni][q], acc[mi][ni], 0, 0, 0);
        }
    }
};

template <int BM, int BN, int BK, int STAGES, bool KTAIL>
__global__ __launch_bounds__(256) void conv_gemm_nt2(ConvGemm p) {
    constexpr int ASZ = BM * BK, BSZ = BN * BK, SSZ = ASZ + BSZ;
    constexpr int CPR = BK / 4;             // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;           // rows per wave-instruction (1 KiB)
    constexpr int NVA = BM / (RPI * 4), NVB = BN / (RPI * 4), LPS = NVA + NVB;
    static_assert(NVA >= 1 && NVB >= 1 && (STAGES == 2 || STAGES == 3), "tile config");
    __shared__ __attribute__((aligned(16))) float smem[STAGES][SSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * BM, n0 = by * BN;
    const int ksteps = (p.K + BK - 1) / BK;
    const int ldb = p.bank ? p.cpb : p.ldb;

    int total = 0;
    for (int kw = p.kw_lo; kw <= p.kw_hi; ++kw) total += kw * ksteps;
    const int per = (total + p.splitk - 1) / p.splitk;
    const int s_begin = blockIdx.z * per, s_end = min(total, s_begin + per);
    if (s_begin >= s_end) return;
    const int nsteps = s_end - s_begin;
    int l_kw = p.kw_lo, l_j = 0, l_kc = 0;                               // next tile to issue
    {
        int skip = s_begin;
        while (skip >= l_kw * ksteps) { skip -= l_kw * ksteps; ++l_kw; }
        l_j = skip / ksteps; l_kc = skip - l_j * ksteps;
    }

    int tpos[NVA], acol[NVA];
    unsigned arow[NVA];
    bool arok[NVA];
#pragma unroll
    for (int v = 0; v < NVA; ++v) {
        const int r = (v * 4 + wave) * RPI + lane / CPR;
        tpos[v] = seqpos(p, m0 + r);
        arow[v] = (unsigned)rowmap(p, m0 + r);
        arok[v] = m0 + r < p.M;
        acol[v] = ((lane % CPR) ^ ((r / (64 / BK)) & (CPR - 1))) * 4;
    }
    unsigned voa[NVA], voa_last[NVA], vob[NVB], vob_last[NVB];
    const int klast = (ksteps - 1) * BK;
#pragma unroll
    for (int v = 0; v < NVB; ++v) {
        const int r = (v * 4 + wave) * RPI + lane / CPR;
        const int bcol = ((lane % CPR) ^ ((r / (64 / BK)) & (CPR - 1))) * 4;
        vob[v] = n0 + r < p.N ? (unsigned)((n0 + r) * ldb + bcol) * 4u : TACO_OOB;
        vob_last[v] = klast + bcol < p.K ? vob[v] : TACO_OOB;
    }
    int sob_tap = 0, aoff = 0;
    bool need_tap = true;
    auto tap_offsets = [&]() {
        const int shift = -(l_j - (l_kw - 1) / 2);
        aoff = p.bank ? (l_kw - 1) * p.cpb : 0;
        sob_tap = __builtin_amdgcn_readfirstlane(((p.bank ? p.N * p.cpb * ((l_kw - 1) * l_kw / 2) : 0) + l_j * p.N * ldb) * 4);
#pragma unroll
        for (int v = 0; v < NVA; ++v) {
            const bool ok = arok[v] && (unsigned)(tpos[v] + shift) < (unsigned)p.T;
            voa[v] = ok ? ((arow[v] + shift) * (unsigned)p.lda + acol[v]) * 4u : TACO_OOB;
            voa_last[v] = klast + acol[v] < p.K ? voa[v] : TACO_OOB;
        }
    };
    int issued = 0;
    auto issue = [&](float* stage) {
        if (need_tap) { tap_offsets(); need_tap = false; }
        const bool last = KTAIL && l_kc == ksteps - 1;
        const int soa = (aoff + l_kc * BK) * 4, sob = sob_tap + l_kc * BK * 4;
#pragma unroll
        for (int v = 0; v < NVA; ++v)
            buf_load_lds16(p.A, stage + (v * 4 + wave) * RPI * BK, last ? voa_last[v] : voa[v], soa);
#pragma unroll
        for (int v = 0; v < NVB; ++v)
            buf_load_lds16(p.B, stage + ASZ + (v * 4 + wave) * RPI * BK, last ? vob_last[v] : vob[v], sob);
        ++issued;
        if (++l_kc == ksteps) { l_kc = 0; need_tap = true; if (++l_j == l_kw) { l_j = 0; ++l_kw; } }
    };

    f32x16 acc[BM / 64][BN / 64];
#pragma unroll
    for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
        for (int ni = 0; ni < BN / 64; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    MmaSw2<BM, BN, BK> mm;
    mm.init(wm, wn, lane);

#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nsteps) issue(smem[t]);
    if (STAGES == 3 && nsteps >= 2) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
    __syncthreads();
    const int ngroups = nsteps / STAGES;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int u = 0; u < STAGES; ++u) {
            const bool more = issued < nsteps;
            if (more) issue(smem[(u + STAGES - 1) % STAGES]);
            mm.run(smem[u], smem[u] + ASZ, acc);
            if (STAGES == 3 && more) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            __syncthreads();
        }
    }
    const int rem = nsteps - ngroups * STAGES;
    if (rem >= 1) {
        mm.run(smem[0], smem[0] + ASZ, acc);
        if (STAGES == 3 && rem == 2) {
            wait_vmcnt<0>();
            __syncthreads();
            mm.run(smem[1], smem[1] + ASZ, acc);
        }
    }
    if (p.splitk == 1) { epilogue_store<BM, BN>(p, acc, m0, n0, wm, wn, lane); return; }
    const int i = lane & 31, h = lane >> 5;         // split-K: partial sums are atomically added into (zeroed) C
#pragma unroll
    for (int ni = 0; ni < BN / 64; ++ni) {
        const int col = n0 + wn * (BN / 2) + ni * 32 + i;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.M) atomicAdd(p.C + rowmap(p, row) * p.ldc + col, acc[mi][ni][r]);
            }
    }
}

// =====================================================================================================
// NT v3 (input gradient of the large conv / dense layers): the fp32 products on the BF16 matrix pipe, three MFMAs per
// product ("bf16x3": a = a_hi + a_lo with a_hi = bf16(a), a_lo = bf16(a - a_hi), both round-to-nearest-even;
// a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulation).  The dropped terms are <= 2^-17 of |a*b| each (a_lo*b_lo and the
// two second-order residuals), against 2^-24 for an fp32 multiply: a dot product over K terms is off by ~1e-5 / sqrt(K) of its
// terms' magnitude, two orders of magnitude inside the 1e-3 the outputs are held to ("hard parts", names this
// splitting as the optimisation to prove against the oracle).  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of
// v_mfma_f32_32x32x2_f32, so three of them per 32x32x16 block cost 96 cycles against 512.
// Tiles go global -> registers (the same masked buffer loads as v2) -> split -> LDS as two bf16 planes per operand,
// [rows][32 k] with 64-byte rows whose four 16-byte chunks are XOR-swizzled by (row >> 2) & 3 (conflict-free ds_read_b128 of the
// 8-k fragments of 32 rows); two LDS stages, the next tile's global loads are in flight under the MFMAs of the current one.
// 128x128x32 tiles, 4 waves as 2x2, 2x2 blocks of 32x32 per wave: every fragment is converted once per workgroup and used by two waves.
// =====================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t buf_load16(const void* base, unsigned voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000),
                                                 voff, soff, 0);
}
// four fp32 -> 4 bf16 high parts (hi.x = elements 0,1; hi.y = 2,3) and 4 bf16 low parts
__device__ __forceinline__ void split_bf16x2(const u32x4_t v, uint2& hi, uint2& lo) {
    const f2 x01 = {__uint_as_float(v.x), __uint_as_float(v.y)}, x23 = {__uint_as_float(v.z), __uint_as_float(v.w)};
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(x01, bf16x2_t));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(x23, bf16x2_t));
    const f2 r01 = {x01.x - __uint_as_float(h01 << 16), x01.y - __uint_as_float(h01 & 0xFFFF0000u)};
    const f2 r23 = {x23.x - __uint_as_float(h23 << 16), x23.y - __uint_as_float(h23 & 0xFFFF0000u)};
    hi = make_uint2(h01, h23);
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2_t)));
}

template <bool KTAIL>
__global__ __launch_bounds__(256) void conv_gemm_nt3(ConvGemm p) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int PLANE = BM * BK * 2;                     // bytes of one bf16 plane of one operand (8 KiB)
    constexpr int STAGE = 4 * PLANE;                       // A hi, A lo, B hi, B lo
    constexpr int NV = BM / 32;                            // 16-byte global loads per thread and operand (rows tid / 8 + 32 v)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = bx * BM, n0 = by * BN;
    const int ksteps = (p.K + BK - 1) / BK;
    const int ldb = p.bank ? p.cpb : p.ldb;

    int total = 0;
    for (int kw = p.kw_lo; kw <= p.kw_hi; ++kw) total += kw * ksteps;
    const int per = (total + p.splitk - 1) / p.splitk;
    const int s_begin = blockIdx.z * per, s_end = min(total, s_begin + per);
    if (s_begin >= s_end) return;
    const int nsteps = s_end - s_begin;
    int l_kw = p.kw_lo, l_j = 0, l_kc = 0;                               // next tile to load
    {
        int skip = s_begin;
        while (skip >= l_kw * ksteps) { skip -= l_kw * ksteps; ++l_kw; }
        l_j = skip / ksteps; l_kc = skip - l_j * ksteps;
    }
    // this thread's rows (tid / 8 + 32 v) and 4-float chunk (tid % 8) of both tiles, and where they go in a plane
    const int rl = tid >> 3, ck = tid & 7;
    int tpos[NV];
    unsigned arow[NV];
    bool arok[NV];
    unsigned vob[NV], vob_last[NV];
    int wr_off[NV];
    const int klast = (ksteps - 1) * BK;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int r = rl + 32 * v;
        tpos[v] = seqpos(p, m0 + r);
        arow[v] = (unsigned)rowmap(p, m0 + r);
        arok[v] = m0 + r < p.M;
        vob[v] = n0 + r < p.N ? (unsigned)((n0 + r) * ldb + ck * 4) * 4u : TACO_OOB;
        vob_last[v] = klast + ck * 4 < p.K ? vob[v] : TACO_OOB;
        wr_off[v] = r * 64 + (((ck >> 1) ^ ((r >> 2) & 3)) << 4) + ((ck & 1) << 3);
    }
    unsigned voa[NV], voa_last[NV];
    int sob_tap = 0, aoff = 0;
    bool need_tap = true;
    auto tap_offsets = [&]() {
        const int shift = -(l_j - (l_kw - 1) / 2);
        aoff = p.bank ? (l_kw - 1) * p.cpb : 0;
        sob_tap = __builtin_amdgcn_readfirstlane(((p.bank ? p.N * p.cpb * ((l_kw - 1) * l_kw / 2) : 0) + l_j * p.N * ldb) * 4);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool ok = arok[v] && (unsigned)(tpos[v] + shift) < (unsigned)p.T;
            voa[v] = ok ? ((arow[v] + shift) * (unsigned)p.lda + ck * 4) * 4u : TACO_OOB;
            voa_last[v] = klast + ck * 4 < p.K ? voa[v] : TACO_OOB;
        }
    };
    u32x4_t ra[NV], rb[NV];
    int issued = 0;
    auto load = [&]() {
        if (need_tap) { tap_offsets(); need_tap = false; }
        const bool last = KTAIL && l_kc == ksteps - 1;
        const int soa = (aoff + l_kc * BK) * 4, sob = sob_tap + l_kc * BK * 4;
#pragma unroll
        for (int v = 0; v < NV; ++v) ra[v] = buf_load16(p.A, last ? voa_last[v] : voa[v], soa);
#pragma unroll
        for (int v = 0; v < NV; ++v) rb[v] = buf_load16(p.B, last ? vob_last[v] : vob[v], sob);
        ++issued;
        if (++l_kc == ksteps) { l_kc = 0; need_tap = true; if (++l_j == l_kw) { l_j = 0; ++l_kw; } }
    };
    auto stage_write = [&](unsigned char* st) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            uint2 hi, lo;
            split_bf16x2(ra[v], hi, lo);
            *reinterpret_cast<uint2*>(st + wr_off[v]) = hi;
            *reinterpret_cast<uint2*>(st + PLANE + wr_off[v]) = lo;
            split_bf16x2(rb[v], hi, lo);
            *reinterpret_cast<uint2*>(st + 2 * PLANE + wr_off[v]) = hi;
            *reinterpret_cast<uint2*>(st + 3 * PLANE + wr_off[v]) = lo;
        }
    };
    // fragment addresses: lane (r = lane & 31, h = lane >> 5) reads the 8 k = 16 kk + 8 h .. of row (block row + r): chunk 2 kk + h
    const int fr = lane & 31, fh = lane >> 5;
    int fa[2][2], fb[2][2];            // [block][kk] byte offsets inside a plane
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ra_ = wm * 64 + i * 32 + fr, rb_ = wn * 64 + i * 32 + fr;
            fa[i][kk] = ra_ * 64 + (((2 * kk + fh) ^ ((ra_ >> 2) & 3)) << 4);
            fb[i][kk] = rb_ * 64 + (((2 * kk + fh) ^ ((rb_ >> 2) & 3)) << 4);
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    auto mma = [&](const unsigned char* st) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8_t*>(st + fa[i][kk]);
                al[i] = *reinterpret_cast<const bf16x8_t*>(st + PLANE + fa[i][kk]);
                bh[i] = *reinterpret_cast<const bf16x8_t*>(st + 2 * PLANE + fb[i][kk]);
                bl[i] = *reinterpret_cast<const bf16x8_t*>(st + 3 * PLANE + fb[i][kk]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);   // small terms first
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        }
    };

    load();
    for (int t = 0; t < nsteps; ++t) {
        unsigned char* st = smem + (t & 1) * STAGE;
        stage_write(st);                        // (waits for the loads of step t)
        __syncthreads();
        if (issued < nsteps) load();            // step t + 1 in flight under the MFMAs of step t
        mma(st);
    }
    if (p.splitk == 1) { epilogue_store<BM, BN>(p, acc, m0, n0, wm, wn, lane); return; }
    const int i = lane & 31, h = lane >> 5;         // split-K: partial sums are atomically added into (zeroed) C
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + ni * 32 + i;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.M) atomicAdd(p.C + rowmap(p, row) * p.ldc + col, acc[mi][ni][r]);
            }
    }
}

// =====================================================================================================
// NN v3 (forward of the large conv / dense layers), bf16x3 products like conv_gemm_nt3.  The weights W[j][k][n] are k-strided: a
// fragment needs 8 consecutive k of one column n, so the B tile is fetched column-wise -- a thread owns column tid % 128 and the
// k quads tid / 128 + 2 v, sixteen dword loads whose lanes cover 256 contiguous bytes of a weight row -- and its four k values are
// packed in k order: no transposition in registers or LDS.  Both LDS images are [row or column][32 k] bf16 with rows padded to 80
// bytes: the fragment reads (32 consecutive rows, one 16-byte chunk) and the 8-byte tile writes of consecutive rows are conflict-free
// without a swizzle.  One conv width per 128-column tile (bank: cpb = 128).
// =====================================================================================================
template <bool KTAIL>
__global__ __launch_bounds__(256) void conv_gemm_nn3(ConvGemm p) {
    constexpr int BM = 128, BN = 128, BK = 32, RS = 80;    // RS: bytes per LDS row (64 + 16 pad)
    constexpr int PLANE = BM * RS;                         // one bf16 plane of one operand (10 KiB)
    constexpr int STAGE = 4 * PLANE;                       // A hi, A lo, B hi, B lo
    constexpr int NV = BM / 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.bank) by = gridDim.y - 1 - blockIdx.y;                         // bank: widest convs first
    else xcd_tile(bx, by);
    const int m0 = bx * BM, n0 = by * BN;
    int kw = p.kw_lo, ldb = p.ldb, nloc0 = n0, nlim = p.N;
    const float* Bb = p.B;
    if (p.bank) {
        kw = 1 + n0 / p.cpb;
        Bb = p.B + (long)p.K * p.cpb * ((kw - 1) * kw / 2);
        ldb = p.cpb;
        nloc0 = n0 - (kw - 1) * p.cpb;
        nlim = p.cpb;
    }
    const int pl = (kw - 1) / 2;
    const int ksteps = (p.K + BK - 1) / BK;
    const int total = kw * ksteps;
    const int per = (total + p.splitk - 1) / p.splitk;
    const int s_begin = blockIdx.z * per;
    const int nsteps = min(total, s_begin + per) - s_begin;
    if (nsteps <= 0) return;

    const int rl = tid >> 3, ck = tid & 7;                 // A: rows rl + 32 v, 4-float chunk ck
    int tpos[NV];
    unsigned arow[NV];
    bool arok[NV];
    const int klast = (ksteps - 1) * BK;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int r = rl + 32 * v;
        tpos[v] = seqpos(p, m0 + r);
        arow[v] = (unsigned)rowmap(p, m0 + r);
        arok[v] = m0 + r < p.M;
    }
    const int bn = tid & 127, bq = tid >> 7;               // B: column bn, k quads bq + 2 v
    const unsigned vob_col = nloc0 + bn < nlim ? (unsigned)(nloc0 + bn) * 4u : TACO_OOB;
    unsigned voa[NV], voa_last[NV];
    auto tap_offsets = [&](int j) {
        const int shift = j - pl;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool ok = arok[v] && (unsigned)(tpos[v] + shift) < (unsigned)p.T;
            voa[v] = ok ? ((arow[v] + shift) * (unsigned)p.lda + ck * 4) * 4u : TACO_OOB;
            voa_last[v] = klast + ck * 4 < p.K ? voa[v] : TACO_OOB;
        }
    };
    const int bstep = BK * ldb * 4, btap = p.K * ldb * 4;
    int i_j = s_begin / ksteps, i_kc = s_begin - i_j * ksteps, issued = 0;
    int soa = i_kc * BK * 4, sob = __builtin_amdgcn_readfirstlane(i_j * btap + i_kc * bstep);
    bool need_tap = true;
    u32x4_t ra[NV];
    float rb[NV][4];
    unsigned vob[NV][4], vob_last[NV][4];        // loop-invariant lane offsets of the 16 weight loads (the k walk is the scalar soffset)
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kl = (bq + 2 * v) * 4 + i;
            vob[v][i] = vob_col == TACO_OOB ? TACO_OOB : vob_col + (unsigned)(kl * ldb) * 4u;
            vob_last[v][i] = klast + kl < p.K ? vob[v][i] : TACO_OOB;
        }
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, 0x7FFFFFFF, 0x00020000);
    auto load = [&]() {
        if (need_tap) { tap_offsets(i_j); need_tap = false; }
        const bool last = KTAIL && i_kc == ksteps - 1;
#pragma unroll
        for (int v = 0; v < NV; ++v) ra[v] = buf_load16(p.A, last ? voa_last[v] : voa[v], soa);
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rb[v][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, last ? vob_last[v][i] : vob[v][i], sob, 0));
        ++i_kc; ++issued; soa += BK * 4; sob += bstep;
        if (i_kc == ksteps) { i_kc = 0; ++i_j; soa = 0; sob = __builtin_amdgcn_readfirstlane(i_j * btap); need_tap = true; }
    };
    auto stage_write = [&](unsigned char* st) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            uint2 hi, lo;
            split_bf16x2(ra[v], hi, lo);
            const int wa = (rl + 32 * v) * RS + ck * 8;
            *reinterpret_cast<uint2*>(st + wa) = hi;
            *reinterpret_cast<uint2*>(st + PLANE + wa) = lo;
            const u32x4_t bv = {__float_as_uint(rb[v][0]), __float_as_uint(rb[v][1]), __float_as_uint(rb[v][2]), __float_as_uint(rb[v][3])};
            split_bf16x2(bv, hi, lo);
            const int wb = bn * RS + (bq + 2 * v) * 8;
            *reinterpret_cast<uint2*>(st + 2 * PLANE + wb) = hi;
            *reinterpret_cast<uint2*>(st + 3 * PLANE + wb) = lo;
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    int fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            fa[i][kk] = (wm * 64 + i * 32 + fr) * RS + (2 * kk + fh) * 16;
            fb[i][kk] = (wn * 64 + i * 32 + fr) * RS + (2 * kk + fh) * 16;
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    auto mma = [&](const unsigned char* st) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8_t*>(st + fa[i][kk]);
                al[i] = *reinterpret_cast<const bf16x8_t*>(st + PLANE + fa[i][kk]);
                bh[i] = *reinterpret_cast<const bf16x8_t*>(st + 2 * PLANE + fb[i][kk]);
                bl[i] = *reinterpret_cast<const bf16x8_t*>(st + 3 * PLANE + fb[i][kk]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        }
    };
    load();
    for (int t = 0; t < nsteps; ++t) {
        unsigned char* st = smem + (t & 1) * STAGE;
        stage_write(st);
        __syncthreads();
        if (issued < nsteps) load();
        mma(st);
    }
    if (p.splitk == 1) { epilogue_store<BM, BN>(p, acc, m0, n0, wm, wn, lane); return; }
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + ni * 32 + i;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.M) atomicAdd(p.C + rowmap(p, row) * p.ldc + col, acc[mi][ni][r]);
            }
    }
}

// =====================================================================================================
// TN: dW_kw[j][c][n] += sum_m X[m + j - pl, c] * dY[m, aoff(kw) + n]                     (weight gradient)
//     p.M = reduction rows, p.K = Cin (output rows), p.N = Cout per conv (output cols);
//     grid.z = segment(kw,j) * splitk + split; partial sums are atomically added into pre-zeroed C.
// =====================================================================================================
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void conv_gemm_tn(ConvGemm p) {
    using TA = KS<BM, BK>;
    using TB = KS<BN, BK>;
    __shared__ __attribute__((aligned(16))) float smem[2][BK * (BM + BN)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int c0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    int seg = blockIdx.z / p.splitk;
    const int split = blockIdx.z - seg * p.splitk;
    int kw = p.kw_lo, j = seg;
    if (p.bank) { kw = 1; while (seg >= kw) { seg -= kw; ++kw; } j = seg; }
    const int shift = j - (kw - 1) / 2 + p.shift0;
    const int aoff = p.bank ? (kw - 1) * p.cpb : 0;
    const int ldc = p.bank ? p.cpb : p.ldc;
    float* Cw = p.C + (p.bank ? (long)p.K * p.cpb * ((kw - 1) * kw / 2) : 0) + (long)j * p.K * ldc;

    const int ktiles = (p.M + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = split * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;

    f32x16 acc[BM / 64][BN / 64];
#pragma unroll
    for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
        for (int ni = 0; ni < BN / 64; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    float4 ra[TA::NV], rb[TB::NV];
    auto gload = [&](int kt) {
#pragma unroll
        for (int v = 0; v < TA::NV; ++v) {
            const int m = kt * BK + TA::row(tid, v);
            const int c = c0 + TA::c4(tid, v) * 4;
            const bool ok = m < p.M && c < p.K && (unsigned)(m % p.T + shift) < (unsigned)p.T;
            ra[v] = ld4(p.A + (long)(m + shift) * p.lda + c, ok);
        }
#pragma unroll
        for (int v = 0; v < TB::NV; ++v) {
            const int m = kt * BK + TB::row(tid, v);
            const int n = n0 + TB::c4(tid, v) * 4;
            rb[v] = ld4(p.B + (long)m * p.ldb + aoff + n, m < p.M && n < p.N);
        }
    };

    gload(kt0);
    TA::store(smem[0], tid, ra);
    TB::store(smem[0] + BK * BM, tid, rb);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) gload(kt + 1);
        mma_tile<BM, BN, BK, false, false>(smem[cur], smem[cur] + BK * BM, acc, wm, wn, lane);
        if (kt + 1 < kt1) {
            TA::store(smem[cur ^ 1], tid, ra);
            TB::store(smem[cur ^ 1] + BK * BM, tid, rb);
        }
        __syncthreads();
    }
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < BN / 64; ++ni) {
        const int col = n0 + wn * (BN / 2) + ni * 32 + i;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.K) atomicAdd(Cw + (long)row * ldc + col, acc[mi][ni][r]);
            }
    }
}

// =====================================================================================================
// TN v2 (weight gradient): reduction over rows; X tile [BK rows][BM channels] and dY tile [BK rows][BN channels], both
// k-strided in LDS, loaded with buffer_load ... lds.  Row masks (sequence edges for a shifted tap, ragged M) only matter
// for a minority of row tiles: a scalar test picks precomputed lane offsets for interior tiles and computes masked
// offsets (VALU) only for edge tiles, so the steady-state K loop issues no VALU work.  The shift is folded into the
// buffer base so that lane offsets stay non-negative.
// =====================================================================================================
template <int BM, int BN, int BK>
struct MmaKs {
    static constexpr int MI = BM / 64, NI = BN / 64, KK = BK / 8;
    int aaddr[MI], baddr[NI];
    __device__ __forceinline__ void init(int wm, int wn, int lane) {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aaddr[mi] = 4 * h * BM + wm * (BM / 2) + mi * 32 + i;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) baddr[ni] = 4 * h * BN + wn * (BN / 2) + ni * 32 + i;
    }
    __device__ __forceinline__ void run(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[MI][NI]) const {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            float a[MI][4], b[NI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[mi][q] = As[aaddr[mi] + (kk * 8 + q) * BM];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) b[ni][q] = Bs[baddr[ni] + (kk * 8 + q) * BN];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][q], b[ni][q], acc[mi][ni], 0, 0, 0);
        }
    }
};

template <int BM, int BN, int BK, int STAGES>
__device__ __forceinline__ void tn2_body(const ConvGemm& p, const int bx, const int by, const int bz) {
    constexpr int ASZ = BK * BM, BSZ = BK * BN, SSZ = ASZ + BSZ;
    constexpr int RA = 256 / BM, RBn = 256 / BN;            // rows per wave-instruction (1 KiB)
    constexpr int NVA = BK / (RA * 4), NVB = BK / (RBn * 4), LPS = NVA + NVB;
    constexpr int LPRA = BM / 4, LPRB = BN / 4;             // lanes per row
    static_assert(NVA >= 1 && NVB >= 1 && (STAGES == 2 || STAGES == 3), "tile config");
    __shared__ __attribute__((aligned(16))) float smem[STAGES][SSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = by * BN;
    // bz = split * nseg + segment: the (conv width, tap) segments of one row range are neighbours in the tile order, so the row
    // range's X and dY tiles are re-read from L2 for every tap instead of from memory (conv bank: sum k = 36 / 136 taps)
    // bank == 2 ("flat" bank, Cin not a multiple of BM -- the post-net's 80 mel channels): the X-column tiles of one conv width walk the
    // FLATTENED (tap, channel) rows of that width's kernel, kw * Cin of them, instead of cdiv(Cin, BM) tiles per tap: a tile then holds
    // channels of up to two neighbouring taps (each lane group of 4 channels carries its own row shift), and the 136 taps of a K = 16
    // bank over 80 channels take 176 tiles instead of 272 (the second tile of every tap was 3/4 padding).
    const bool flat = p.bank == 2;
    int nseg = p.bank ? p.kw_hi * (p.kw_hi + 1) / 2 : p.kw_lo;
    if (flat) { nseg = 0; for (int k = 1; k <= p.kw_hi; ++k) nseg += (k * p.K + BM - 1) / BM; }
    const int split = bz / nseg;
    int seg = bz - split * nseg;
    int kw = p.kw_lo, j = seg;
    if (flat) { kw = 1; while (seg >= (kw * p.K + BM - 1) / BM) { seg -= (kw * p.K + BM - 1) / BM; ++kw; } j = 0; }
    else if (p.bank) { kw = 1; while (seg >= kw) { seg -= kw; ++kw; } j = seg; }
    const int c0 = flat ? seg * BM : bx * BM;                // first output row of this tile: flat (tap, channel) index / channel
    const int klim = flat ? kw * p.K : p.K;                  // rows of this output block
    const int shift = (flat ? c0 / p.K : j) - (kw - 1) / 2 + p.shift0;                 // row shift of the tile's first tap ...
    const int shift_hi = flat ? shift + (min(c0 + BM, klim) - 1) / p.K - c0 / p.K : shift;     // ... and of its last one
    const int aoff = p.bank ? (kw - 1) * p.cpb : 0;
    const int ldc = p.bank ? p.cpb : p.ldc;
    float* Cw = p.C + (p.bank ? (long)p.K * p.cpb * ((kw - 1) * kw / 2) : 0) + (long)j * p.K * ldc;

    const int ktiles = (p.M + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = split * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;
    const int nsteps = kt1 - kt0;
    const float* Ab = p.A + (long)shift * p.lda;             // row m of this view is X[m + shift]
    const float* Bbase = p.B + aoff;

    int mla[NVA], mlb[NVB];
    unsigned voa_in[NVA], vob_in[NVB];
    const int cl = c0 + (lane % LPRA) * 4;                   // this lane's 4 output rows (Cin % 4 == 0: one tap)
    const int jl = flat ? cl / p.K - c0 / p.K : 0;           // its tap, relative to the tile's first
    const int ccl = flat ? cl - (cl / p.K) * p.K : cl;       // its channel
#pragma unroll
    for (int v = 0; v < NVA; ++v) {
        mla[v] = (v * 4 + wave) * RA + lane / LPRA;
        voa_in[v] = cl < klim ? (unsigned)((mla[v] + jl) * p.lda + ccl) * 4u : TACO_OOB;
    }
#pragma unroll
    for (int v = 0; v < NVB; ++v) {
        mlb[v] = (v * 4 + wave) * RBn + lane / LPRB;
        const int n = n0 + (lane % LPRB) * 4;
        vob_in[v] = n < p.N ? (unsigned)(mlb[v] * p.ldb + n) * 4u : TACO_OOB;
    }
    int i_kt = kt0, issued = 0;
    int t0 = (kt0 * BK) % p.T;                               // position of the tile's first row inside its sequence
    const int astep = BK * p.lda * 4, bstep = BK * p.ldb * 4;
    int soa = __builtin_amdgcn_readfirstlane(kt0 * astep), sob = __builtin_amdgcn_readfirstlane(kt0 * bstep);
    auto issue = [&](float* stage) {
        const int mbase = i_kt * BK;
        const bool full = mbase + BK <= p.M;
        const bool interior = full && ((shift == 0 && shift_hi == 0) || (t0 + shift >= 0 && t0 + BK - 1 + shift_hi < p.T && t0 + BK - 1 < p.T));
        if (interior) {
#pragma unroll
            for (int v = 0; v < NVA; ++v) buf_load_lds16(Ab, stage + (v * 4 + wave) * RA * BM, voa_in[v], soa);
        } else {
#pragma unroll
            for (int v = 0; v < NVA; ++v) {
                const int m = mbase + mla[v];
                const bool ok = m < p.M && (unsigned)(m % p.T + shift + jl) < (unsigned)p.T;
                buf_load_lds16(Ab, stage + (v * 4 + wave) * RA * BM, ok ? voa_in[v] : TACO_OOB, soa);
            }
        }
        if (full) {
#pragma unroll
            for (int v = 0; v < NVB; ++v) buf_load_lds16(Bbase, stage + ASZ + (v * 4 + wave) * RBn * BN, vob_in[v], sob);
        } else {
#pragma unroll
            for (int v = 0; v < NVB; ++v)
                buf_load_lds16(Bbase, stage + ASZ + (v * 4 + wave) * RBn * BN, mbase + mlb[v] < p.M ? vob_in[v] : TACO_OOB, sob);
        }
        ++i_kt; ++issued; soa += astep; sob += bstep;
        t0 += BK; if (t0 >= p.T) t0 %= p.T;
    };

    f32x16 acc[BM / 64][BN / 64];
#pragma unroll
    for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
        for (int ni = 0; ni < BN / 64; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    MmaKs<BM, BN, BK> mm;
    mm.init(wm, wn, lane);
    // bias gradient of a dense layer = column sums of dY: the workgroups of the first X-column tile add up the dY tiles they stage
    // anyway (thread = column tid % BN, rows (tid / BN) * (BK * BN / 256) ... of every tile; rows beyond M were loaded as zeros)
    const bool do_bias = p.dbias != nullptr && bx == 0 && seg == 0;
    float bsum = 0.f;
    auto bias_tile = [&](const float* Bs) {
        constexpr int RPT = BK * BN / 256;           // rows per thread
        const int col = tid % BN, r0 = (tid / BN) * RPT;
#pragma unroll
        for (int k = 0; k < RPT; ++k) bsum += Bs[(r0 + k) * BN + col];
    };

#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nsteps) issue(smem[t]);
    if (STAGES == 3 && nsteps >= 2) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
    __syncthreads();
    const int ngroups = nsteps / STAGES;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int u = 0; u < STAGES; ++u) {
            const bool more = issued < nsteps;
            if (more) issue(smem[(u + STAGES - 1) % STAGES]);
            mm.run(smem[u], smem[u] + ASZ, acc);
            if (do_bias) bias_tile(smem[u] + ASZ);
            if (STAGES == 3 && more) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
            __syncthreads();
        }
    }
    const int rem = nsteps - ngroups * STAGES;
    if (rem >= 1) {
        mm.run(smem[0], smem[0] + ASZ, acc);
        if (do_bias) bias_tile(smem[0] + ASZ);
        if (STAGES == 3 && rem == 2) {
            wait_vmcnt<0>();
            __syncthreads();
            mm.run(smem[1], smem[1] + ASZ, acc);
            if (do_bias) bias_tile(smem[1] + ASZ);
        }
    }
    if (do_bias && n0 + tid % BN < p.N) atomicAdd(p.dbias + n0 + tid % BN, bsum);
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < BN / 64; ++ni) {
        const int col = n0 + wn * (BN / 2) + ni * 32 + i;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < BM / 64; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < klim) atomicAdd(Cw + (long)row * ldc + col, acc[mi][ni][r]);
            }
    }
}

template <int BM, int BN, int BK, int STAGES>
__global__ __launch_bounds__(256) void conv_gemm_tn2(ConvGemm p) {
    tn2_body<BM, BN, BK, STAGES>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// =====================================================================================================
// TN v3 (weight gradients of the large layers), bf16x3 products like conv_gemm_nt3 / nn3.  Both operands are k-strided here (the
// reduction runs over the ROWS of X and dY), so both tiles are fetched column-wise like the weights of conv_gemm_nn3: a thread owns
// one X channel / dY column (tid % 128) and the row quads tid / 128 + 2 v of a 32-row reduction tile, packs its four rows in order
// and writes 8 bytes of the [column][32 rows] bf16 image (80-byte rows).  Interior row tiles use loop-invariant lane offsets and a
// scalar row walk like tn2; edge tiles (sequence ends under a tap shift, ragged M) compute their row masks.  128 x 128 output tiles,
// split over the rows; partial sums are atomically added.  Flat bank rows (bank == 2) as in tn2_body: the tap is a per-thread constant.
// =====================================================================================================
__device__ __forceinline__ void tn3_body(const ConvGemm& p, const int bx, const int by, const int bz, unsigned char* smem) {
    constexpr int BM = 128, BN = 128, BK = 32, RS = 80;
    constexpr int PLANE = BM * RS, STAGE = 4 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = by * BN;
    const bool flat = p.bank == 2;
    int nseg = p.bank ? p.kw_hi * (p.kw_hi + 1) / 2 : p.kw_lo;
    if (flat) { nseg = 0; for (int k = 1; k <= p.kw_hi; ++k) nseg += (k * p.K + BM - 1) / BM; }
    const int split = bz / nseg;
    int seg = bz - split * nseg;
    int kw = p.kw_lo, j = seg;
    if (flat) { kw = 1; while (seg >= (kw * p.K + BM - 1) / BM) { seg -= (kw * p.K + BM - 1) / BM; ++kw; } j = 0; }
    else if (p.bank) { kw = 1; while (seg >= kw) { seg -= kw; ++kw; } j = seg; }
    const int c0 = flat ? seg * BM : bx * BM;
    const int klim = flat ? kw * p.K : p.K;
    const int shift = (flat ? c0 / p.K : j) - (kw - 1) / 2 + p.shift0;
    const int shift_hi = flat ? shift + (min(c0 + BM, klim) - 1) / p.K - c0 / p.K : shift;
    const int aoff = p.bank ? (kw - 1) * p.cpb : 0;
    const int ldc = p.bank ? p.cpb : p.ldc;
    float* Cw = p.C + (p.bank ? (long)p.K * p.cpb * ((kw - 1) * kw / 2) : 0) + (long)j * p.K * ldc;

    const int ktiles = (p.M + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = split * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;
    const int nsteps = kt1 - kt0;
    const float* Ab = p.A + (long)shift * p.lda;             // row m of this view is X[m + shift]
    const float* Bbase = p.B + aoff;

    const int col = tid & 127, q0 = tid >> 7;                // this thread's column of both tiles, row quads q0 + 2 v
    const int cl = c0 + col;
    const int jl = flat ? cl / p.K - c0 / p.K : 0;
    const int ccl = flat ? cl - (cl / p.K) * p.K : cl;
    const unsigned voa = cl < klim ? (unsigned)(jl * p.lda + ccl) * 4u : TACO_OOB;
    const unsigned vob = n0 + col < p.N ? (unsigned)(n0 + col) * 4u : TACO_OOB;
    int i_kt = kt0, issued = 0;
    int t0 = (kt0 * BK) % p.T;
    const int astep = BK * p.lda * 4, bstep = BK * p.ldb * 4;
    int soa = __builtin_amdgcn_readfirstlane(kt0 * astep), sob = __builtin_amdgcn_readfirstlane(kt0 * bstep);
    // the 2 x 16 lane offsets of a row tile are loop-invariant (the row walk is the scalar soffset): kept in registers, so that an interior
    // tile issues its 32 loads with no address arithmetic (rocprofv3 counted 9 non-MFMA vector and 10 scalar instructions per MFMA in the
    // first version of this kernel, a third of them these offsets and their selects)
    unsigned oa0[4][4], ob0[4][4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = (q0 + 2 * v) * 4 + i;
            oa0[v][i] = voa == TACO_OOB ? TACO_OOB : voa + (unsigned)(ml * p.lda) * 4u;
            ob0[v][i] = vob == TACO_OOB ? TACO_OOB : vob + (unsigned)(ml * p.ldb) * 4u;
        }
    const auto ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, 0x7FFFFFFF, 0x00020000);
    const auto rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bbase), 0, 0x7FFFFFFF, 0x00020000);
    float ra[4][4], rb[4][4];
    auto load = [&]() {
        const int mbase = i_kt * BK;
        const bool full = mbase + BK <= p.M;
        const bool interior = full && ((shift == 0 && shift_hi == 0) || (t0 + shift >= 0 && t0 + BK - 1 + shift_hi < p.T && t0 + BK - 1 < p.T));
        if (interior) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ra[v][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra_rsrc, oa0[v][i], soa, 0));
                    rb[v][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb_rsrc, ob0[v][i], sob, 0));
                }
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mbase + (q0 + 2 * v) * 4 + i;
                    const bool oka = m < p.M && (unsigned)(m % p.T + shift + jl) < (unsigned)p.T, okb = m < p.M;
                    ra[v][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra_rsrc, oka ? oa0[v][i] : TACO_OOB, soa, 0));
                    rb[v][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb_rsrc, okb ? ob0[v][i] : TACO_OOB, sob, 0));
                }
        }
        ++i_kt; ++issued; soa += astep; sob += bstep;
        t0 += BK; if (t0 >= p.T) t0 %= p.T;
    };
    auto stage_write = [&](unsigned char* st) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            uint2 hi, lo;
            const int w = col * RS + (q0 + 2 * v) * 8;
            const u32x4_t av = {__float_as_uint(ra[v][0]), __float_as_uint(ra[v][1]), __float_as_uint(ra[v][2]), __float_as_uint(ra[v][3])};
            split_bf16x2(av, hi, lo);
            *reinterpret_cast<uint2*>(st + w) = hi;
            *reinterpret_cast<uint2*>(st + PLANE + w) = lo;
            const u32x4_t bv = {__float_as_uint(rb[v][0]), __float_as_uint(rb[v][1]), __float_as_uint(rb[v][2]), __float_as_uint(rb[v][3])};
            split_bf16x2(bv, hi, lo);
            *reinterpret_cast<uint2*>(st + 2 * PLANE + w) = hi;
            *reinterpret_cast<uint2*>(st + 3 * PLANE + w) = lo;
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    int fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            fa[i][kk] = (wm * 64 + i * 32 + fr) * RS + (2 * kk + fh) * 16;
            fb[i][kk] = (wn * 64 + i * 32 + fr) * RS + (2 * kk + fh) * 16;
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    auto mma = [&](const unsigned char* st) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8_t*>(st + fa[i][kk]);
                al[i] = *reinterpret_cast<const bf16x8_t*>(st + PLANE + fa[i][kk]);
                bh[i] = *reinterpret_cast<const bf16x8_t*>(st + 2 * PLANE + fb[i][kk]);
                bl[i] = *reinterpret_cast<const bf16x8_t*>(st + 3 * PLANE + fb[i][kk]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        }
    };
    // bias gradient of a dense layer = column sums of dY: the workgroups of the first X-column tile add up the dY values they stage
    // anyway (this thread: its column, 16 of the 32 rows of every tile; rows beyond M were loaded as zeros)
    const bool do_bias = p.dbias != nullptr && bx == 0 && seg == 0;
    float bsum = 0.f;
    load();
    for (int t = 0; t < nsteps; ++t) {
        unsigned char* st = smem + (t & 1) * STAGE;
        if (do_bias) {
#pragma unroll
            for (int v = 0; v < 4; ++v) bsum += (rb[v][0] + rb[v][1]) + (rb[v][2] + rb[v][3]);
        }
        stage_write(st);
        __syncthreads();
        if (issued < nsteps) load();
        mma(st);
    }
    if (do_bias && n0 + col < p.N) atomicAdd(p.dbias + n0 + col, bsum);
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int cn = n0 + wn * 64 + ni * 32 + i;
        if (cn >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < klim) atomicAdd(Cw + (long)row * ldc + cn, acc[mi][ni][r]);
            }
    }
}

__global__ __launch_bounds__(256) void conv_gemm_tn3(ConvGemm p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 4 * 128 * 80];
    tn3_body(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Grouped weight-gradient launch: ONE grid walks the tiles of up to TACO_WG_MAX independent dW problems (largest first).
// Backward produces 44 weight-gradient GEMMs per step, most of them 10-40 us: launched one by one on the side stream their
// launch gaps and ramp-up/drain phases are a third of the stream's time and every launch is a new burst of workgroups
// competing with the latency-bound recurrence kernels.  The problem table travels in the kernel arguments (no device table,
// no host->device copy, HIP-graph capturable).
#define TACO_WG_MAX 30
struct WgradGroup {
    int count;
    int first[TACO_WG_MAX + 1];                 // first[i] = index of problem i's first workgroup; first[count] = grid size
    unsigned short gx[TACO_WG_MAX], gy[TACO_WG_MAX];
    ConvGemm p[TACO_WG_MAX];
};
static_assert(sizeof(WgradGroup) <= 4096, "kernel argument block");
template <int BM, int BN, int BK, int STAGES>
__global__ __launch_bounds__(256) void conv_gemm_tn2_group(WgradGroup g) {
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < g.count && b >= g.first[i + 1]) ++i;      // wave-uniform scan of <= 32 kernel-argument words
    const int first = g.first[i], end = g.first[i + 1], gx = g.gx[i], gy = g.gy[i];
    // XCD-aware tile order inside a problem: the workgroups that land on one XCD (global index mod 8 under round-robin placement;
    // speed only) take a CONTIGUOUS run of the problem's tiles, ordered (row segment, dY column tile, X column tile): they share
    // the dY tile and walk the X tiles of one row segment, so an operand is fetched into one L2 instead of all eight (rocprofv3
    // FETCH_SIZE of the grouped launches before this remap: 315 MB per launch for ~55 MB of operands).  Bijective for any sizes.
    auto upto = [](int n, int x) { return (n + 7 - x) >> 3; };       // blocks in [0, n) with index mod 8 == x
    const int x = b & 7;
    int base = 0;
    for (int y = 0; y < x; ++y) base += upto(end, y) - upto(first, y);
    const int r = base + upto(b, x) - upto(first, x);
    tn2_body<BM, BN, BK, STAGES>(g.p[i], r % gx, (r / gx) % gy, r / (gx * gy));
}

__global__ __launch_bounds__(256) void conv_gemm_tn3_group(WgradGroup g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 4 * 128 * 80];
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < g.count && b >= g.first[i + 1]) ++i;
    const int first = g.first[i], end = g.first[i + 1], gx = g.gx[i], gy = g.gy[i];
    auto upto = [](int n, int x) { return (n + 7 - x) >> 3; };       // same XCD-aware tile order as conv_gemm_tn2_group
    const int x = b & 7;
    int base = 0;
    for (int y = 0; y < x; ++y) base += upto(end, y) - upto(first, y);
    const int r = base + upto(b, x) - upto(first, x);
    tn3_body(g.p[i], r % gx, (r / gx) % gy, r / (gx * gy), smem);
}

// ---- host-side dispatch --------------------------------------------------------------------------------
// TACO_GEMM_TILE=128|64 forces the tile configuration (tuning aid); default: by tile count
static int forced_tile() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TACO_GEMM_TILE"); v = e ? atoi(e) : 0; }
    return v;
}
// Measured on MI355X (scripts/dev_gemm.py): 64x64x32 tiles (4+ workgroups per CU, finer tail) match or beat
// 128x128x16 on every C2 shape except very wide outputs (N >= 1024 with >= 1024 big tiles, e.g. the post-net bank).
static bool use128(long tiles128, int N) {
    const int f = forced_tile();
    if (f == 128) return true;
    if (f == 64) return false;
    return tiles128 >= 1024 && N >= 1024;
}

static bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int check_common(const ConvGemm& p) {
    if (!p.A || !p.B || !p.C) return TACO_EINVAL;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.T <= 0) return TACO_EINVAL;
    if (!aligned4(p.A) || !aligned4(p.B)) return TACO_EINVAL;
    if ((p.lda & 3) || (p.ldb & 3)) return TACO_EINVAL;
    return TACO_OK;
}

// ---- NN dispatch ---------------------------------------------------------------------------------------------------
// v2 (buffer_load ... lds, VALU-free K loop) whenever every operand offset fits 31 bits; otherwise the generic v1 kernel.
// Tile configurations measured with scripts/gemm_bench.hip on MI355X (us, post-net shapes at C2):
//   64x64x32, 3 stages : best general choice (3 WGs/CU, a lone workgroup still hides the load latency)
//   64x64x32, 2 stages : short reductions (<= 16 K-steps: prologue of the deeper ring is not amortised)
//   64x64x16, 3 stages : Cin with a large remainder mod 32 (post-net bank, Cin = 80)
//   128x128x16, 3 stages: >= 2048 big tiles (143 TFLOP/s on a plain 8192 x 8192 x 2048 GEMM)
// TACO_NN2_TILE=<0..3> forces a configuration, TACO_NN_V1=1 forces the v1 kernel (tuning aids).
static long max_phys_row(const ConvGemm& p) {
    const long m = p.M - 1;
    return p.rb_len ? (m / p.rb_len) * p.rb_stride + p.rb_off + (m % p.rb_len) : m;
}
static bool fits31(const ConvGemm& p) {
    const long abytes = (max_phys_row(p) + 1 + p.kw_hi) * (long)p.lda * 4;
    long wrows = 0;
    for (int k = p.kw_lo; k <= p.kw_hi; ++k) wrows += (long)k * p.K;
    if (!p.bank) wrows = (long)p.kw_hi * p.K;
    const long bbytes = (wrows + 64) * (long)p.ldb * 4;
    return abytes < (1L << 31) && bbytes < (1L << 31);
}
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// TACO_X3: which GEMMs multiply on the BF16 matrix pipe with three MFMAs per fp32 product (conv_gemm_nt3 / tn3 / nn3): 0 = none (exact
// fp32 products everywhere), 1 = the backward pass (input and weight gradients; default), 2 = the forward pass too.  Read at every launch
// (not cached: bench.py and the tests time / check both settings in one process).
static int x3_mode() { return env_int("TACO_X3", 1); }

#define NN2_LAUNCH(BM_, BN_, BK_, ST_) do { dim3 g2(cdiv(p.M, BM_), cdiv(p.N, BN_), p.splitk); \
        if (p.K % BK_) hipLaunchKernelGGL((conv_gemm_nn2<BM_, BN_, BK_, ST_, true>), g2, dim3(256), 0, stream, p); \
        else hipLaunchKernelGGL((conv_gemm_nn2<BM_, BN_, BK_, ST_, false>), g2, dim3(256), 0, stream, p); } while (0)

// returns the split count used (> 1: partial sums were added atomically, the epilogue never saw final values)
static int launch_nn(ConvGemm p, hipStream_t stream) {
    static const int force_v1 = env_int("TACO_NN_V1", 0), force_cfg = env_int("TACO_NN2_TILE", -1);
    static const int no_split = env_int("TACO_NN_NOSPLIT", 0), split_min_m = env_int("TACO_NN_SPLIT_MINM", 1024);
    p.splitk = 1;
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
    if (force_v1 || !fits31(p)) {
        if (use128(tiles128, p.N)) {
            dim3 g(cdiv(p.M, 128), cdiv(p.N, 128));
            hipLaunchKernelGGL((conv_gemm_nn<128, 128, 16>), g, dim3(256), 0, stream, p);
        } else {
            dim3 g(cdiv(p.M, 64), cdiv(p.N, 64));
            hipLaunchKernelGGL((conv_gemm_nn<64, 64, 32>), g, dim3(256), 0, stream, p);
        }
        return 1;
    }
    // few output tiles but a long reduction (encoder proj_1: 128 tiles x 192 steps): split the steps over blockIdx.z
    {
        const long tiles = (long)cdiv(p.M, 64) * cdiv(p.N, 64);
        const int nsteps = p.kw_hi * cdiv(p.K, 32);
        if (!no_split && !p.bank && !p.accumulate && !p.rb_len && p.M >= split_min_m && tiles <= 128 && nsteps >= 64) {
            int sk = (int)(1024 / tiles);
            if (sk > nsteps / 16) sk = nsteps / 16;
            if (sk > 16) sk = 16;
            if (sk > 1 && hipMemset2DAsync(p.C, (size_t)p.ldc * sizeof(float), 0, (size_t)p.N * sizeof(float), p.M, stream) == hipSuccess)
                p.splitk = sk;
        }
    }
    // bf16x3 products (conv_gemm_nn3) for the large problems, only on request (TACO_X3=2): the forward pass takes the ReLU / max-pool
    // decisions, and products good to 2^-17 flip ~6x more near-ties against the float64 oracle than fp32 products do
    static const int x3_min_tiles = env_int("TACO_X3_MIN_TILES", 128);
    if (x3_mode() >= 2 && force_cfg < 0 && tiles128 * p.splitk >= x3_min_tiles && p.N >= 64 && (!p.bank || p.cpb == 128)) {
        dim3 g3(cdiv(p.M, 128), cdiv(p.N, 128), p.splitk);
        if (p.K % 32) hipLaunchKernelGGL((conv_gemm_nn3<true>), g3, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_nn3<false>), g3, dim3(256), 0, stream, p);
        if (p.splitk > 1 && (p.bias || p.act != ACT_NONE)) {
            const long work = (long)p.M * p.N;
            hipLaunchKernelGGL(bias_act_k, dim3((int)((work + 1023) / 1024 > 2048 ? 2048 : (work + 1023) / 1024)), dim3(256), 0, stream, p);
        }
        return p.splitk;
    }
    int cfg = force_cfg;
    if (cfg < 0) {
        const int k32 = cdiv(p.K, 32) * 32;
        if (tiles128 >= 2048) cfg = 3;
        else if ((k32 - p.K) * 8 > p.K) cfg = 2;
        else if (p.kw_hi * (k32 / 32) <= 16) cfg = 1;
        else cfg = 0;
    }
    if (cfg == 1) NN2_LAUNCH(64, 64, 32, 2);
    else if (cfg == 2) NN2_LAUNCH(64, 64, 16, 3);
    else if (cfg == 3) NN2_LAUNCH(128, 128, 16, 3);
    else NN2_LAUNCH(64, 64, 32, 3);
    if (p.splitk > 1 && (p.bias || p.act != ACT_NONE)) {
        const long work = (long)p.M * p.N;
        hipLaunchKernelGGL(bias_act_k, dim3((int)((work + 1023) / 1024 > 2048 ? 2048 : (work + 1023) / 1024)), dim3(256), 0, stream, p);
    }
    return p.splitk;
}

static int conv_gemm_fwd_impl(const float* X, const float* W, const float* bias, float* Y, int M, int T, int Cin,
                              int Cout, int kw, int bank_K, int ldx, int ldw, int ldy, int act, int accumulate, double* bn_stat,
                              hipStream_t stream);
extern "C" int taco_conv_gemm_fwd(const float* X, const float* W, const float* bias, float* Y, int M, int T, int Cin,
                                  int Cout, int kw, int bank_K, int ldx, int ldw, int ldy, int act, int accumulate,
                                  hipStream_t stream) {
    return conv_gemm_fwd_impl(X, W, bias, Y, M, T, Cin, Cout, kw, bank_K, ldx, ldw, ldy, act, accumulate, nullptr, stream);
}
// conv + activation with the batch-norm sums of the output accumulated in the epilogue (no extra pass over Y); bn_stat: TACO_BN_DSTAT(C)
// zeroed doubles, finalised by taco_bn_finalize.  (A split-K launch cannot see final values: the sums are then taken by a pass over Y.)
extern "C" int taco_conv_gemm_bn_fwd(const float* X, const float* W, const float* bias, float* Y, int M, int T, int Cin,
                                     int Cout, int kw, int bank_K, int ldx, int ldw, int ldy, int act, double* bn_stat,
                                     hipStream_t stream) {
    if (!bn_stat) return TACO_EINVAL;
    return conv_gemm_fwd_impl(X, W, bias, Y, M, T, Cin, Cout, kw, bank_K, ldx, ldw, ldy, act, 0, bn_stat, stream);
}
static int conv_gemm_fwd_impl(const float* X, const float* W, const float* bias, float* Y, int M, int T, int Cin,
                              int Cout, int kw, int bank_K, int ldx, int ldw, int ldy, int act, int accumulate, double* bn_stat,
                              hipStream_t stream) {
    // bank_K > 0: fused conv bank (widths 1..bank_K, 128 channels each, outputs concatenated: Cout = bank_K*128)
    ConvGemm p{};
    p.bn_stat = bn_stat;
    p.A = X; p.B = W; p.C = Y; p.bias = bias;
    p.M = M; p.K = Cin; p.T = T; p.lda = ldx; p.ldc = ldy; p.act = act; p.accumulate = accumulate; p.splitk = 1;
    if (bank_K > 0) { p.bank = 1; p.cpb = 128; p.N = bank_K * 128; p.ldb = 128; p.kw_lo = 1; p.kw_hi = bank_K; }
    else { p.bank = 0; p.cpb = 0; p.N = Cout; p.ldb = ldw; p.kw_lo = p.kw_hi = kw; }
    if (int e = check_common(p)) return e;
    if ((p.K & 3) || M % T != 0 || kw < 1) return TACO_EINVAL;
    if (launch_nn(p, stream) > 1 && bn_stat)      // split-K: the sums come from a pass over Y
        return taco_bn_stats_rows(Y, ldy, bn_stat, M / T, T, 0, T, p.N, stream);
    TACO_RETURN_LAST();
}

// conv1d / conv bank over the FRAMES [t0, t1) of every length-T sequence (all taps read the full sequences: rows outside [t0, t1) must
// already hold their final values where a tap reaches them).  Lets the post-net's bank run chunk by chunk behind the decoder pipeline.
extern "C" int taco_conv_rows_fwd(const float* X, const float* W, const float* bias, float* Y, int N, int T, int t0, int t1, int Cin,
                                  int Cout, int kw, int bank_K, int ldx, int ldw, int ldy, int act, double* bn_stat, hipStream_t stream) {
    ConvGemm p{};
    p.bn_stat = bn_stat;                 // optional: batch-norm sums of the rows written (no split-K on this path)
    const int len = t1 - t0;
    if (N <= 0 || T <= 0 || len <= 0 || t0 < 0 || t1 > T || kw < 1) return TACO_EINVAL;
    p.A = X; p.B = W; p.C = Y; p.bias = bias;
    p.M = N * len; p.K = Cin; p.T = T; p.lda = ldx; p.ldc = ldy; p.act = act; p.accumulate = 0; p.splitk = 1;
    p.rb_len = len; p.rb_stride = T; p.rb_off = t0;
    if (bank_K > 0) { p.bank = 1; p.cpb = 128; p.N = bank_K * 128; p.ldb = 128; p.kw_lo = 1; p.kw_hi = bank_K; }
    else { p.bank = 0; p.cpb = 0; p.N = Cout; p.ldb = ldw; p.kw_lo = p.kw_hi = kw; }
    if (int e = check_common(p)) return e;
    if (p.K & 3) return TACO_EINVAL;
    launch_nn(p, stream);
    TACO_RETURN_LAST();
}

// ---- NT dispatch (input gradient) ------------------------------------------------------------------------------------
#define NT2_LAUNCH(BM_, BN_, BK_, ST_) do { dim3 g2(cdiv(p.M, BM_), cdiv(p.N, BN_), p.splitk); \
        if (p.K % BK_) hipLaunchKernelGGL((conv_gemm_nt2<BM_, BN_, BK_, ST_, true>), g2, dim3(256), 0, stream, p); \
        else hipLaunchKernelGGL((conv_gemm_nt2<BM_, BN_, BK_, ST_, false>), g2, dim3(256), 0, stream, p); } while (0)

static bool fits31_nt(const ConvGemm& p) {
    const long abytes = (max_phys_row(p) + 1 + p.kw_hi) * (long)p.lda * 4;
    long wrows = 0;                                                      // W rows of length ldb: [taps][Cin]
    for (int k = p.kw_lo; k <= p.kw_hi; ++k) wrows += (long)k * p.N;
    const long bbytes = (wrows + 64) * (long)(p.bank ? p.cpb : p.ldb) * 4;
    return abytes < (1L << 31) && bbytes < (1L << 31);
}

// returns the split count chosen (the caller zero-fills C when > 1 and not accumulating)
static int plan_nt(ConvGemm& p) {
    const long tiles = (long)cdiv(p.M, 64) * cdiv(p.N, 64);
    int nsteps = 0;
    for (int k = p.kw_lo; k <= p.kw_hi; ++k) nsteps += k * cdiv(p.K, 32);
    int splitk = 1;
    if (tiles < 256 && nsteps >= 64) {      // few output tiles but a long reduction (conv bank dX): split the taps
        splitk = (int)((768 + tiles - 1) / tiles);
        if (splitk > nsteps / 16) splitk = nsteps / 16;
        if (splitk < 1) splitk = 1;
    }
    p.splitk = splitk;
    return splitk;
}

static void launch_nt(const ConvGemm& p, hipStream_t stream) {
    static const int force_v1 = env_int("TACO_NT_V1", 0), force_cfg = env_int("TACO_NT2_TILE", -1);
    if (force_v1 || !fits31_nt(p)) {
        dim3 g(cdiv(p.M, 64), cdiv(p.N, 64), p.splitk);
        hipLaunchKernelGGL((conv_gemm_nt<64, 64, 32>), g, dim3(256), 0, stream, p);
        return;
    }
    // bf16x3 products (conv_gemm_nt3) for the large problems: >= 256 tiles of 128 x 128.  TACO_X3=0: exact fp32 products everywhere
    static const int x3_min_tiles = env_int("TACO_X3_MIN_TILES", 128);
    if (x3_mode() >= 1 && force_cfg < 0 && (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.splitk >= x3_min_tiles && p.N >= 64) {
        dim3 g3(cdiv(p.M, 128), cdiv(p.N, 128), p.splitk);
        if (p.K % 32) hipLaunchKernelGGL((conv_gemm_nt3<true>), g3, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_nt3<false>), g3, dim3(256), 0, stream, p);
        return;
    }
    int cfg = force_cfg;
    if (cfg < 0) {
        const int k32 = cdiv(p.K, 32) * 32;
        int nsteps = 0;
        for (int k = p.kw_lo; k <= p.kw_hi; ++k) nsteps += k * (k32 / 32);
        const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
        if (tiles128 >= 2048 && p.splitk == 1) cfg = 3;
        else if ((k32 - p.K) * 8 > p.K) cfg = 2;
        else if (nsteps / p.splitk <= 16) cfg = 1;
        else cfg = 0;
    }
    if (cfg == 1) NT2_LAUNCH(64, 64, 32, 2);
    else if (cfg == 2) NT2_LAUNCH(64, 64, 16, 3);
    else if (cfg == 3) NT2_LAUNCH(128, 128, 16, 3);
    else NT2_LAUNCH(64, 64, 32, 3);
}

extern "C" int taco_conv_gemm_bwd_data(const float* dY, const float* W, float* dX, int M, int T, int Cin, int Cout,
                                       int kw, int bank_K, int lddy, int ldw, int lddx, int accumulate, hipStream_t stream) {
    ConvGemm p{};
    p.A = dY; p.B = W; p.C = dX; p.bias = nullptr;
    p.M = M; p.N = Cin; p.T = T; p.lda = lddy; p.ldc = lddx; p.act = ACT_NONE; p.accumulate = accumulate; p.splitk = 1;
    if (bank_K > 0) { p.bank = 1; p.cpb = 128; p.K = 128; p.ldb = 128; p.kw_lo = 1; p.kw_hi = bank_K; }
    else { p.bank = 0; p.K = Cout; p.ldb = ldw; p.kw_lo = p.kw_hi = kw; }
    if (int e = check_common(p)) return e;
    if (M % T != 0 || kw < 1) return TACO_EINVAL;
    const int splitk = plan_nt(p);
    if (splitk > 1 && !accumulate)
        if (hipMemset2DAsync(dX, (size_t)lddx * sizeof(float), 0, (size_t)Cin * sizeof(float), M, stream) != hipSuccess) return TACO_EINVAL;
    launch_nt(p, stream);
    TACO_RETURN_LAST();
}

// input gradient over the FRAMES [t0, t1) of every length-T sequence (reads dY rows of the whole sequences).  accumulate: dX += ...;
// a piece small enough to be split over the taps adds its partial sums atomically, so it REQUIRES accumulate (dX pre-initialised).
// Lets the post-net bank's input gradient run chunk by chunk under the lead-in of the decoder backward.
extern "C" int taco_conv_rows_bwd_data(const float* dY, const float* W, float* dX, int N, int T, int t0, int t1, int Cin, int Cout,
                                       int kw, int bank_K, int lddy, int ldw, int lddx, int accumulate, hipStream_t stream) {
    ConvGemm p{};
    const int len = t1 - t0;
    if (N <= 0 || T <= 0 || len <= 0 || t0 < 0 || t1 > T || kw < 1) return TACO_EINVAL;
    p.A = dY; p.B = W; p.C = dX; p.bias = nullptr;
    p.M = N * len; p.N = Cin; p.T = T; p.lda = lddy; p.ldc = lddx; p.act = ACT_NONE; p.accumulate = accumulate; p.splitk = 1;
    p.rb_len = len; p.rb_stride = T; p.rb_off = t0;
    if (bank_K > 0) { p.bank = 1; p.cpb = 128; p.K = 128; p.ldb = 128; p.kw_lo = 1; p.kw_hi = bank_K; }
    else { p.bank = 0; p.K = Cout; p.ldb = ldw; p.kw_lo = p.kw_hi = kw; }
    if (int e = check_common(p)) return e;
    if (plan_nt(p) > 1 && !accumulate) return TACO_EINVAL;
    launch_nt(p, stream);
    TACO_RETURN_LAST();
}

// plans one weight-gradient problem: fills p and the launch grid; cfg: 0 = v2 64x64x32 (3 stages; the grouped kernel's
// configuration), 1 = v2 64x64x32 2 stages (short loops), 2 = v2 128x128x16, 3 = v1 64, 4 = v1 128
static inline int ktiles32_for(int M) { return (M + 31) / 32; }
static int plan_bwd_weight(ConvGemm& p, dim3& g, int& cfg, const float* X, const float* dY, float* dW, int M, int T, int Cin,
                           int Cout, int kw, int bank_K, int ldx, int lddy, int ldw, int shift0, int wgs_target, float* dbias = nullptr) {
    // dW must be zero-initialised (or hold a running sum): partial sums are atomically ADDED.
    p = ConvGemm{};
    p.shift0 = shift0;
    p.A = X; p.B = dY; p.C = dW; p.bias = nullptr;
    p.M = M; p.K = Cin; p.T = T; p.lda = ldx; p.ldb = lddy;
    int nseg;
    if (bank_K > 0) { p.bank = 1; p.cpb = 128; p.N = 128; p.ldc = 128; p.kw_lo = 1; p.kw_hi = bank_K; nseg = bank_K * (bank_K + 1) / 2; }
    else { p.bank = 0; p.N = Cout; p.ldc = ldw; p.kw_lo = p.kw_hi = kw; nseg = kw; }
    if (int e = check_common(p)) return e;
    if ((p.K & 3) || M % T != 0 || kw < 1) return TACO_EINVAL;
    // Measured with scripts/gemm_bench.hip (MODE=bwd_weight) on MI355X: with the v2 kernel 64x64x32 tiles match or beat
    // 128x128x16 on every model shape (fewer atomics per output element at equal workgroup count); big tiles only pay
    // for >= 512 of them.  The reduction is split until ~wgs_target workgroups exist, keeping >= 12 row tiles per workgroup.
    static const int tn_big = env_int("TACO_TN_BIG", -1);
    const bool big = tn_big >= 0 ? tn_big != 0
                                 : (p.K >= 128 && p.N >= 128 && (long)cdiv(p.K, 128) * cdiv(p.N, 128) * nseg >= 512);
    const int bm = big ? 128 : 64, bk = big ? 16 : 32;
    const long tiles = (long)cdiv(p.K, bm) * cdiv(p.N, bm) * nseg;
    const int ktiles = cdiv(M, bk);
    int splitk = (int)((wgs_target + tiles - 1) / tiles);
    const int max_split = ktiles / 12 > 0 ? ktiles / 12 : 1;
    if (splitk > max_split) splitk = max_split;
    if (splitk < 1) splitk = 1;
    p.splitk = splitk;
    g = dim3(cdiv(p.K, bm), cdiv(p.N, bm), nseg * splitk);
    static const int force_v1 = env_int("TACO_TN_V1", 0);
    // v2 needs 31-bit byte offsets into X (plus the shifted rows) and dY
    const bool v2 = !force_v1 && ((long)M + 64) * ldx * 4 < (1L << 31) && ((long)M + 64) * lddy * 4 < (1L << 31);
    cfg = !v2 ? (big ? 4 : 3) : big ? 2 : (cdiv(ktiles, splitk) <= 8 ? 1 : 0);
    p.dbias = (cfg <= 1 && kw == 1 && bank_K == 0) ? dbias : nullptr;      // fused bias gradient: dense problems on the 64x64x32 v2 kernel
    // bf16x3 products (tn3_body, 128 x 128 x 32 tiles; cfg 5) for the large problems: >= 2 GFLOP, both output dimensions >= 64
    static const int x3_min_mflop = env_int("TACO_X3_MIN_MFLOP", 2000);
    if (x3_mode() >= 1 && v2 && tn_big < 0 && p.N >= 64 && (bank_K > 0 ? p.K * bank_K : p.K) >= 64 &&
        2.0 * M * p.K * p.N * nseg >= 1e6 * x3_min_mflop) {
        int segs = nseg;
        p.bank = bank_K > 0 ? 1 : 0;
        if (bank_K > 0 && p.K % 128 != 0) {
            segs = 0;
            for (int k = 1; k <= bank_K; ++k) segs += cdiv(k * p.K, 128);
            p.bank = 2;
        }
        const long t3 = (long)(p.bank == 2 ? 1 : cdiv(p.K, 128)) * cdiv(p.N, 128) * segs;
        int sk = (int)((wgs_target / 2 + t3 - 1) / t3);
        const int ms = ktiles32_for(M) / 8 > 0 ? ktiles32_for(M) / 8 : 1;
        if (sk > ms) sk = ms;
        if (sk < 1) sk = 1;
        p.splitk = sk;
        g = dim3(p.bank == 2 ? 1 : cdiv(p.K, 128), cdiv(p.N, 128), segs * sk);
        cfg = 5;
        p.dbias = (kw == 1 && bank_K == 0) ? dbias : nullptr;
        return TACO_OK;
    }
    // flat bank (tn2_body): the X-column tiles walk the (tap, channel) rows of a conv width without per-tap padding.  TACO_BANK_FLAT=0: off
    static const int bank_flat = env_int("TACO_BANK_FLAT", 1);
    if (bank_K > 0 && cfg <= 1 && bank_flat && p.K % bm != 0) {
        int segs = 0;
        for (int k = 1; k <= bank_K; ++k) segs += cdiv(k * p.K, bm);
        // (the split was planned on the padded tile count: re-plan it on the real one)
        splitk = (int)((wgs_target + (long)segs * cdiv(p.N, bm) - 1) / ((long)segs * cdiv(p.N, bm)));
        if (splitk > max_split) splitk = max_split;
        if (splitk < 1) splitk = 1;
        p.bank = 2; p.splitk = splitk;
        g = dim3(1, cdiv(p.N, bm), segs * splitk);
        cfg = cdiv(ktiles, splitk) <= 8 ? 1 : 0;
    }
    return TACO_OK;
}

static void launch_bwd_weight(const ConvGemm& p, dim3 g, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 5: hipLaunchKernelGGL(conv_gemm_tn3, g, dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((conv_gemm_tn<128, 128, 16>), g, dim3(256), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((conv_gemm_tn<64, 64, 32>), g, dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((conv_gemm_tn2<128, 128, 16, 3>), g, dim3(256), 0, stream, p); break;
        case 1: hipLaunchKernelGGL((conv_gemm_tn2<64, 64, 32, 2>), g, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((conv_gemm_tn2<64, 64, 32, 3>), g, dim3(256), 0, stream, p);
    }
}

static int conv_gemm_bwd_weight_impl(const float* X, const float* dY, float* dW, int M, int T, int Cin, int Cout,
                                     int kw, int bank_K, int ldx, int lddy, int ldw, int shift0, hipStream_t stream) {
    static const int tn_wgs = env_int("TACO_TN_WGS", 2048);
    ConvGemm p; dim3 g; int cfg;
    if (int e = plan_bwd_weight(p, g, cfg, X, dY, dW, M, T, Cin, Cout, kw, bank_K, ldx, lddy, ldw, shift0, tn_wgs)) return e;
    launch_bwd_weight(p, g, cfg, stream);
    TACO_RETURN_LAST();
}

// Grouped form (see conv_gemm_tn2_group): every problem that fits the 64x64x32 v2 configuration joins a group of <=
// TACO_WG_MAX problems, ordered by work (largest first, so the long tiles start early and the short ones fill the tail);
// the others (128-wide tiles, > 2 GiB operands) are launched on their own.
extern "C" int taco_wgrad_group(const TacoWgrad* items, int count, hipStream_t stream) {
    if (!items || count < 0) return TACO_EINVAL;
    // a group shares the chip: the per-problem split targets fewer workgroups than a lone launch would
    static const int grp_wgs = env_int("TACO_TN_GROUP_WGS", 1024);
    struct Planned { ConvGemm p; dim3 g; double work; };
    Planned pl[TACO_WG_MAX], pl3[TACO_WG_MAX];
    int n = 0, n3 = 0;
    auto flush3 = [&]() {                                   // the bf16x3 problems of the run: one grouped launch of their own
        if (!n3) return;
        for (int a = 1; a < n3; ++a) {
            Planned t = pl3[a]; int b = a - 1;
            while (b >= 0 && pl3[b].work < t.work) { pl3[b + 1] = pl3[b]; --b; }
            pl3[b + 1] = t;
        }
        WgradGroup grp;
        grp.count = n3;
        int first = 0;
        for (int a = 0; a < n3; ++a) {
            grp.first[a] = first; grp.gx[a] = (unsigned short)pl3[a].g.x; grp.gy[a] = (unsigned short)pl3[a].g.y; grp.p[a] = pl3[a].p;
            first += (int)(pl3[a].g.x * pl3[a].g.y * pl3[a].g.z);
        }
        for (int a = n3; a <= TACO_WG_MAX; ++a) grp.first[a] = first;
        hipLaunchKernelGGL(conv_gemm_tn3_group, dim3(first), dim3(256), 0, stream, grp);
        n3 = 0;
    };
    auto flush = [&]() {
        flush3();
        if (!n) return;
        for (int a = 1; a < n; ++a) {                     // insertion sort, descending work
            Planned t = pl[a]; int b = a - 1;
            while (b >= 0 && pl[b].work < t.work) { pl[b + 1] = pl[b]; --b; }
            pl[b + 1] = t;
        }
        WgradGroup grp;
        grp.count = n;
        int first = 0;
        for (int a = 0; a < n; ++a) {
            grp.first[a] = first; grp.gx[a] = (unsigned short)pl[a].g.x; grp.gy[a] = (unsigned short)pl[a].g.y; grp.p[a] = pl[a].p;
            first += (int)(pl[a].g.x * pl[a].g.y * pl[a].g.z);
        }
        for (int a = n; a <= TACO_WG_MAX; ++a) grp.first[a] = first;
        hipLaunchKernelGGL((conv_gemm_tn2_group<64, 64, 32, 3>), dim3(first), dim3(256), 0, stream, grp);
        n = 0;
    };
    for (int i = 0; i < count; ++i) {
        const TacoWgrad& it = items[i];
        ConvGemm p; dim3 g; int cfg;
        if (int e = plan_bwd_weight(p, g, cfg, it.X, it.dY, it.dW, it.M, it.T, it.Cin, it.Cout, it.kw, it.bank_K, it.ldx, it.lddy,
                                    it.ldw, it.shift, grp_wgs, it.dbias)) return e;
        if (it.dbias && !p.dbias)            // a problem whose bias gradient cannot ride on its GEMM: column sums in a launch of their own
            if (int e = taco_col_sum(it.dY, it.lddy, it.dbias, it.M, it.Cout, stream)) return e;
        if (cfg == 5 && g.x <= 65535 && g.y <= 65535) {
            pl3[n3].p = p; pl3[n3].g = g;
            pl3[n3].work = (double)g.x * g.y * g.z * cdiv(cdiv(it.M, 32), p.splitk);
            if (++n3 == TACO_WG_MAX) flush3();
            continue;
        }
        if (cfg > 1 || g.x > 65535 || g.y > 65535) { launch_bwd_weight(p, g, cfg, stream); continue; }
        pl[n].p = p; pl[n].g = g;
        pl[n].work = (double)g.x * g.y * g.z * cdiv(cdiv(it.M, 32), p.splitk);
        if (++n == TACO_WG_MAX) flush();
    }
    flush();
    TACO_RETURN_LAST();
}

extern "C" int taco_conv_gemm_bwd_weight(const float* X, const float* dY, float* dW, int M, int T, int Cin, int Cout,
                                         int kw, int bank_K, int ldx, int lddy, int ldw, hipStream_t stream) {
    return conv_gemm_bwd_weight_impl(X, dY, dW, M, T, Cin, Cout, kw, bank_K, ldx, lddy, ldw, 0, stream);
}

// dW[c,n] += sum_m X[m + shift, c] * dY[m, n] with rows outside the length-T sequence of m contributing zero
// (shift = -1: X is a hidden-state sequence and dW the gradient of a recurrent weight)
extern "C" int taco_gemm_tn_shift(const float* X, const float* dY, float* dW, int M, int T, int K, int N, int ldx, int lddy,
                                  int ldw, int shift, hipStream_t stream) {
    return conv_gemm_bwd_weight_impl(X, dY, dW, M, T, K, N, 1, 0, ldx, lddy, ldw, shift, stream);
}

// Dense layers over a CHUNK of steps of [N,S,*] tensors: logical rows (n, s) with s in [s0, s1), physical row n*S + s.
// Used to pipeline the decoder recurrences chunk by chunk (engine.py); both X and Y / dY and dX use the [N,S,*] layout.
extern "C" int taco_dense_rows_fwd(const float* X, const float* W, const float* bias, float* Y, int N, int S, int s0, int s1,
                                   int Cin, int Cout, int ldx, int ldw, int ldy, int act, int accumulate, hipStream_t stream) {
    ConvGemm p{};
    const int ch = s1 - s0, M = N * ch;
    p.A = X; p.B = W; p.C = Y; p.bias = bias;
    p.M = M; p.N = Cout; p.K = Cin; p.T = M > 0 ? M : 1; p.lda = ldx; p.ldb = ldw; p.ldc = ldy; p.act = act; p.accumulate = accumulate;
    p.splitk = 1; p.kw_lo = p.kw_hi = 1; p.rb_len = ch; p.rb_stride = S; p.rb_off = s0;
    if (ch <= 0 || s0 < 0 || s1 > S) return TACO_EINVAL;
    if (int e = check_common(p)) return e;
    if (p.K & 3) return TACO_EINVAL;
    launch_nn(p, stream);
    TACO_RETURN_LAST();
}

extern "C" int taco_dense_rows_bwd_data(const float* dY, const float* W, float* dX, int N, int S, int s0, int s1, int Cin,
                                        int Cout, int lddy, int ldw, int lddx, int accumulate, hipStream_t stream) {
    ConvGemm p{};
    const int ch = s1 - s0, M = N * ch;
    p.A = dY; p.B = W; p.C = dX; p.bias = nullptr;
    p.M = M; p.N = Cin; p.K = Cout; p.T = M > 0 ? M : 1; p.lda = lddy; p.ldb = ldw; p.ldc = lddx; p.act = ACT_NONE;
    p.accumulate = accumulate; p.splitk = 1; p.kw_lo = p.kw_hi = 1; p.rb_len = ch; p.rb_stride = S; p.rb_off = s0;
    if (ch <= 0 || s0 < 0 || s1 > S) return TACO_EINVAL;
    if (int e = check_common(p)) return e;
    launch_nt(p, stream);
    TACO_RETURN_LAST();
}
