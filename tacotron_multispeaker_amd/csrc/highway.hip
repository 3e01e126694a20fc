// Fused highway stack of the CBHG (reference models/modules.py:63-64, 77-90: four layers of
//   H = relu(x W_H + b_H), T = sigmoid(x W_T + b_T), y = H T + x (1 - T)   on [M, 128] rows).
//
// One launch per direction instead of four (dense GEMM + gate kernel) pairs: a workgroup owns a tile of rows, keeps the
// activations of the tile in LDS for all four layers and streams the fused [W_H | W_T] kernels (128 x 256 fp32 = 128 KB per
// layer, L2 resident) through a two-stage LDS ring; the gating runs in the MFMA epilogue on the accumulator registers.
// The 128-wide GEMMs of one layer are only 4 K-steps deep as separate launches (56 TFLOP/s measured: prologue / epilogue and
// the gate kernel's extra pass dominate); fused, a layer never leaves the CU.
//   forward : saves per layer the gate activations [relu(H) | sigmoid(T)] ([M,256]) and the layer outputs ([M,128]) -- exactly the
//             tensors the unfused path saved -- so backward and the weight-gradient GEMMs are unchanged consumers.
//   backward: per layer dZ = [dy T (H>0) | dy (H - x) T (1-T)] (written out: operand of the deferred dW GEMM / bias sum),
//             dx = dy (1 - T) + dZ . W^T  (fp32 MFMA, K = 256), layer 4 -> 1 inside one launch.
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32); operand k order permuted identically for A and B (see gemm.hip mma_tile).
#include "common.hpp"

#define HW_D 128                 // highway width
#define HW_LDX (HW_D + 4)        // activation tile row stride (k-contiguous, padded)
#define HW_BK 16                 // K-chunk of the weight ring
// kernel-argument pointer arrays are read with constant indices only (a runtime index would copy the struct to scratch)
#define HW_SEL(arr, l) ((l) == 0 ? (arr)[0] : (l) == 1 ? (arr)[1] : (l) == 2 ? (arr)[2] : (arr)[3])

// ---------------------------------------------------------------------------------------------------------------------
// forward: tile = 32 rows, 4 waves side by side: wave w computes all 32 rows of the H columns [32 w, +32) together with the
// matching T columns [128 + 32 w, +32): H and T of one unit meet in one lane.  (A 64-row tile with 2 x 2 waves gave 320
// workgroups at the post-net's 20480 rows: 64 CUs held two of them and set the kernel time; 640 workgroups of 50 KB LDS are one
// round at three per CU.)
// ---------------------------------------------------------------------------------------------------------------------
struct Hw4 {
    const float* x0;             // [M,128] input of layer 1
    const float* W[4];           // [128,256] each ([W_H | W_T])
    const float* b[4];           // [256]
    float* Z[4];                 // [M,256] out: [relu(H) | sigmoid(T)]
    float* y[4];                 // [M,128] out: layer outputs
    int M;
    // frame band (taco_highway4_*_rows): the tiles cover frames [rb_off, rb_off + rb_len) of every length-rb_stride sequence;
    // rb_len = 0: the M rows as they lie
    int rb_len, rb_stride, rb_off;
};
// first physical row and number of valid rows of the 32-row tile b
__device__ __forceinline__ void hw_tile(int b, int M, int rb_len, int rb_stride, int rb_off, long& m0, int& valid) {
    if (rb_len) {
        const int tpb = (rb_len + 31) / 32, n = b / tpb, ft = b - n * tpb;
        m0 = (long)n * rb_stride + rb_off + 32 * ft;
        valid = min(32, rb_len - 32 * ft);
    } else {
        m0 = (long)b * 32;
        valid = (int)min(32L, (long)M - m0);
    }
}
#define HWF_ROWS 32

__global__ __launch_bounds__(256, 3) void highway4_fwd_k(Hw4 p) {
    extern __shared__ __attribute__((aligned(16))) float hw_smem[];
    float* xs = hw_smem;                                   // [32][HW_LDX] activations of the tile (in place)
    float (*ws)[HW_BK * 256] = reinterpret_cast<float (*)[HW_BK * 256]>(hw_smem + HWF_ROWS * HW_LDX);   // weight ring: [2][k][256 columns]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    long m0; int valid;
    hw_tile(blockIdx.x, p.M, p.rb_len, p.rb_stride, p.rb_off, m0, valid);
    // ---- stage the input tile
    for (int v = tid; v < HWF_ROWS * 32; v += 256) {
        const int r = v >> 5, c4 = v & 31;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < valid) a = *reinterpret_cast<const float4*>(p.x0 + (m0 + r) * HW_D + c4 * 4);
        *reinterpret_cast<float4*>(&xs[r * HW_LDX + c4 * 4]) = a;
    }
    // weight chunk c of layer l: rows k = 16 c .. +16 of W_l, all 256 columns = 16 KB = 4 float4 per thread
    // (macros, not lambdas: a register array captured by reference lands in scratch)
    float4 w0, w1, w2, w3;
#define wload(l, c) do { const float* wp_ = HW_SEL(p.W, l) + (long)((c) * HW_BK) * 256 + tid * 4; \
        w0 = *reinterpret_cast<const float4*>(wp_); w1 = *reinterpret_cast<const float4*>(wp_ + 1024); \
        w2 = *reinterpret_cast<const float4*>(wp_ + 2048); w3 = *reinterpret_cast<const float4*>(wp_ + 3072); } while (0)
#define wstore(dst) do { float* dp_ = (dst) + tid * 4; *reinterpret_cast<float4*>(dp_) = w0; *reinterpret_cast<float4*>(dp_ + 1024) = w1; \
        *reinterpret_cast<float4*>(dp_ + 2048) = w2; *reinterpret_cast<float4*>(dp_ + 3072) = w3; } while (0)
    wload(0, 0);
    wstore(ws[0]);
    __syncthreads();
    int stage = 0;
    const int colH = 32 * wave + i, colT = 128 + 32 * wave + i;
    for (int l = 0; l < 4; ++l) {
        f32x16 accH, accT;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accH[r] = 0.0f; accT[r] = 0.0f; }
        for (int c = 0; c < HW_D / HW_BK; ++c) {
            const bool last = (l == 3 && c == HW_D / HW_BK - 1);
            if (!last) { if (c + 1 < HW_D / HW_BK) wload(l, c + 1); else wload(l + 1, 0); }
            const float* wsb = ws[stage];
#pragma unroll
            for (int kk = 0; kk < HW_BK / 8; ++kk) {
                const float4 av = *reinterpret_cast<const float4*>(&xs[i * HW_LDX + c * HW_BK + kk * 8 + 4 * h]);
                const float a[4] = {av.x, av.y, av.z, av.w};
                float bh[4], bt[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { bh[q] = wsb[(kk * 8 + 4 * h + q) * 256 + colH]; bt[q] = wsb[(kk * 8 + 4 * h + q) * 256 + colT]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    accH = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bh[q], accH, 0, 0, 0);
                    accT = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bt[q], accT, 0, 0, 0);
                }
            }
            if (!last) wstore(ws[stage ^ 1]);
            __syncthreads();                              // chunk consumed by every wave; the next one is visible
            stage ^= 1;
        }
        // ---- gating epilogue on the accumulators (every wave is past its reads of xs: the last barrier of the chunk loop)
        const float* bl = HW_SEL(p.b, l);
        float* Zl = HW_SEL(p.Z, l);
        float* yl = HW_SEL(p.y, l);
        const float bhv = bl[colH], btv = bl[colT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float hv = fmaxf(accH[r] + bhv, 0.f);
            const float tv = sigmoidf_(accT[r] + btv);
            const float xv = xs[row * HW_LDX + colH];
            const float o = hv * tv + xv * (1.f - tv);
            xs[row * HW_LDX + colH] = o;
            if (row < valid) {
                Zl[(m0 + row) * 256 + colH] = hv;
                Zl[(m0 + row) * 256 + colT] = tv;
                yl[(m0 + row) * HW_D + colH] = o;
            }
        }
        __syncthreads();                                  // the layer's output tile is complete before the next layer reads it
    }
#undef wload
#undef wstore
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: tile = 32 rows; 4 waves, wave w computes the dx columns [32 w, +32) of all 32 rows (one accumulator).
// ---------------------------------------------------------------------------------------------------------------------
struct Hw4B {
    const float* dy;             // [M,128] gradient wrt the output of layer 4
    const float* HT[4];          // [M,256] saved gate activations
    const float* xin[4];         // [M,128] input of layer l (x0, y1, y2, y3)
    const float* W[4];           // [128,256]
    float* dZ[4];                // [M,256] out: gradient wrt the pre-activations [H | T] (dW / bias gradient operand)
    float* dx;                   // [M,128] out: gradient wrt the input of layer 1
    int M;
    int rb_len, rb_stride, rb_off;   // frame band, see Hw4
};
#define HWB_LDZ (256 + 4)
#define HWB_BK 16                // K-chunk (Z columns) of the backward weight ring: 70.6 KB of LDS = two workgroups per CU
                                 // (32: 87 KB, ONE workgroup per CU and 2.5 rounds of the 640 post-net tiles)
#define HWB_LDW (HWB_BK + 4)

__global__ __launch_bounds__(256, 2) void highway4_bwd_k(Hw4B p) {
    extern __shared__ __attribute__((aligned(16))) float hw_smem[];            // 70.6 KB
    float* gs = hw_smem;                                   // [32][HW_LDX] dy of the current layer (in place -> dx)
    float* zs = gs + 32 * HW_LDX;                          // [32][HWB_LDZ] dZ tile, k-contiguous (k = Z column)
    float (*ws)[HW_D * HWB_LDW] = reinterpret_cast<float (*)[HW_D * HWB_LDW]>(zs + 32 * HWB_LDZ);   // weight ring: [2][input unit n][16 Z columns]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    long m0; int valid;
    hw_tile(blockIdx.x, p.M, p.rb_len, p.rb_stride, p.rb_off, m0, valid);
    for (int v = tid; v < 32 * 32; v += 256) {
        const int r = v >> 5, c4 = v & 31;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < valid) a = *reinterpret_cast<const float4*>(p.dy + (m0 + r) * HW_D + c4 * 4);
        *reinterpret_cast<float4*>(&gs[r * HW_LDX + c4 * 4]) = a;
    }
    // weight chunk c of layer l: W_l[n][16 c .. +16] for all 128 rows n = 512 float4: thread covers rows (tid >> 2) + 64 v, v < 2
    // (named registers and macros: a float4 array staged through helper functions or lambdas lands in scratch)
    static_assert(HWB_BK == 16, "weight ring mapping");
    float4 w0, w1;
#define wload(l, c) do { const float* wp_ = HW_SEL(p.W, l) + (long)(tid >> 2) * 256 + (c) * HWB_BK + (tid & 3) * 4; \
        w0 = *reinterpret_cast<const float4*>(wp_); w1 = *reinterpret_cast<const float4*>(wp_ + 64 * 256); } while (0)
#define wstore(dst) do { float* dp_ = (dst) + (tid >> 2) * HWB_LDW + (tid & 3) * 4; *reinterpret_cast<float4*>(dp_) = w0; \
        *reinterpret_cast<float4*>(dp_ + 64 * HWB_LDW) = w1; } while (0)
    // saved activations of a layer ([H|T] 32 KB + layer input 16 KB per tile) are fetched into registers ONE LAYER AHEAD, while the
    // MFMA loop of the layer above runs: loaded at the point of use, every layer exposed an HBM round trip of 48 KB per workgroup
    // (plain unrolled loops on constant indices: staged through lambdas the arrays land in scratch)
    float4 ph[4], pt[4], px[4];
#define pfetch(l) do { const float* H_ = HW_SEL(p.HT, l); const float* X_ = HW_SEL(p.xin, l); \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) { \
            const long m_ = m0 + (tid >> 5) + 8 * v; const int c_ = (tid & 31) * 4; \
            ph[v] = pt[v] = px[v] = make_float4(0.f, 0.f, 0.f, 0.f); \
            if ((tid >> 5) + 8 * v < valid) { ph[v] = *reinterpret_cast<const float4*>(H_ + m_ * 256 + c_); \
                            pt[v] = *reinterpret_cast<const float4*>(H_ + m_ * 256 + 128 + c_); \
                            px[v] = *reinterpret_cast<const float4*>(X_ + m_ * HW_D + c_); } } } while (0)
    pfetch(3);
    __syncthreads();
    for (int l = 3; l >= 0; --l) {
        // ---- gate backward of the tile: thread = (row, 4 consecutive units); dZ to HBM + LDS, direct path into gs
        wload(l, 0);
        float* dZl = HW_SEL(p.dZ, l);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = (tid >> 5) + 8 * v, c = (tid & 31) * 4;
            const long m = m0 + r;
            const float4 hv = ph[v], tv = pt[v], xv = px[v];
            const float4 g = *reinterpret_cast<const float4*>(&gs[r * HW_LDX + c]);
            float4 dh, dt, dd;
            dh.x = hv.x > 0.f ? g.x * tv.x : 0.f; dh.y = hv.y > 0.f ? g.y * tv.y : 0.f;
            dh.z = hv.z > 0.f ? g.z * tv.z : 0.f; dh.w = hv.w > 0.f ? g.w * tv.w : 0.f;
            dt.x = g.x * (hv.x - xv.x) * tv.x * (1.f - tv.x); dt.y = g.y * (hv.y - xv.y) * tv.y * (1.f - tv.y);
            dt.z = g.z * (hv.z - xv.z) * tv.z * (1.f - tv.z); dt.w = g.w * (hv.w - xv.w) * tv.w * (1.f - tv.w);
            dd.x = g.x * (1.f - tv.x); dd.y = g.y * (1.f - tv.y); dd.z = g.z * (1.f - tv.z); dd.w = g.w * (1.f - tv.w);
            if (r < valid) {
                *reinterpret_cast<float4*>(dZl + m * 256 + c) = dh;
                *reinterpret_cast<float4*>(dZl + m * 256 + 128 + c) = dt;
            }
            *reinterpret_cast<float4*>(&zs[r * HWB_LDZ + c]) = dh;      // rows beyond M: hv = tv = 0 -> dh = dt = 0, dd = g = 0
            *reinterpret_cast<float4*>(&zs[r * HWB_LDZ + 128 + c]) = dt;
            *reinterpret_cast<float4*>(&gs[r * HW_LDX + c]) = dd;          // same thread read g above: in place
        }
        if (l > 0) pfetch(l - 1);                              // in flight during this layer's MFMA loop
        wstore(ws[0]);
        __syncthreads();
        // ---- dx[row][n] += sum_k dZ[row][k] W[n][k], k over the 256 Z columns
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        int stage = 0;
        for (int c = 0; c < 256 / HWB_BK; ++c) {
            if (c + 1 < 256 / HWB_BK) wload(l, c + 1);
            const float* wsb = ws[stage];
#pragma unroll
            for (int kk = 0; kk < HWB_BK / 8; ++kk) {
                const float4 av = *reinterpret_cast<const float4*>(&zs[i * HWB_LDZ + c * HWB_BK + kk * 8 + 4 * h]);
                const float4 bv = *reinterpret_cast<const float4*>(&wsb[(32 * wave + i) * HWB_LDW + kk * 8 + 4 * h]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
            if (c + 1 < 256 / HWB_BK) wstore(ws[stage ^ 1]);
            __syncthreads();
            stage ^= 1;
        }
        // ---- dx = direct path + product; becomes the dy of the layer below (or the kernel's output)
        {
            const int col = 32 * wave + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float o = gs[row * HW_LDX + col] + acc[r];
                gs[row * HW_LDX + col] = o;
                if (l == 0 && row < valid) p.dx[(m0 + row) * HW_D + col] = o;
            }
        }
        __syncthreads();
    }
#undef wload
#undef wstore
#undef pfetch
}

static int highway4_fwd_launch(const float* x0, const float* const* W4, const float* const* b4, float* const* Z4, float* const* y4,
                               int M, int rb_len, int rb_stride, int rb_off, int tiles, hipStream_t stream) {
    Hw4 p{};
    p.x0 = x0; p.M = M; p.rb_len = rb_len; p.rb_stride = rb_stride; p.rb_off = rb_off;
    for (int l = 0; l < 4; ++l) {
        if (!W4[l] || !b4[l] || !Z4[l] || !y4[l]) return TACO_EINVAL;
        p.W[l] = W4[l]; p.b[l] = b4[l]; p.Z[l] = Z4[l]; p.y[l] = y4[l];
    }
    constexpr size_t smem = (HWF_ROWS * HW_LDX + 2 * HW_BK * 256) * sizeof(float);      // 49.7 KB
    static DevMask attr{0};
    if (ensure_dyn_lds((const void*)highway4_fwd_k, (int)smem, attr) != TACO_OK) return TACO_EINVAL;
    hipLaunchKernelGGL(highway4_fwd_k, dim3(tiles), dim3(256), smem, stream, p);
    TACO_RETURN_LAST();
}

static int highway4_bwd_launch(const float* dy, const float* const* HT4, const float* const* xin4, const float* const* W4,
                               float* const* dZ4, float* dx, int M, int rb_len, int rb_stride, int rb_off, int tiles, hipStream_t stream) {
    Hw4B p{};
    p.dy = dy; p.dx = dx; p.M = M; p.rb_len = rb_len; p.rb_stride = rb_stride; p.rb_off = rb_off;
    for (int l = 0; l < 4; ++l) {
        if (!HT4[l] || !xin4[l] || !W4[l] || !dZ4[l]) return TACO_EINVAL;
        p.HT[l] = HT4[l]; p.xin[l] = xin4[l]; p.W[l] = W4[l]; p.dZ[l] = dZ4[l];
    }
    constexpr size_t smem = (32 * HW_LDX + 32 * HWB_LDZ + 2 * HW_D * HWB_LDW) * sizeof(float);      // 70.6 KB
    static DevMask attr{0};
    if (ensure_dyn_lds((const void*)highway4_bwd_k, (int)smem, attr) != TACO_OK) return TACO_EINVAL;
    hipLaunchKernelGGL(highway4_bwd_k, dim3(tiles), dim3(256), smem, stream, p);
    TACO_RETURN_LAST();
}

extern "C" int taco_highway4_fwd(const float* x0, const float* const* W4, const float* const* b4, float* const* Z4, float* const* y4,
                                 int M, hipStream_t stream) {
    if (!x0 || !W4 || !b4 || !Z4 || !y4 || M <= 0) return TACO_EINVAL;
    return highway4_fwd_launch(x0, W4, b4, Z4, y4, M, 0, 0, 0, cdiv(M, HWF_ROWS), stream);
}

extern "C" int taco_highway4_bwd(const float* dy, const float* const* HT4, const float* const* xin4, const float* const* W4,
                                 float* const* dZ4, float* dx, int M, hipStream_t stream) {
    if (!dy || !HT4 || !xin4 || !W4 || !dZ4 || !dx || M <= 0) return TACO_EINVAL;
    return highway4_bwd_launch(dy, HT4, xin4, W4, dZ4, dx, M, 0, 0, 0, cdiv(M, 32), stream);
}

// the same over the frames [f0, f1) of every length-T sequence of [N,T,*] tensors (rows n*T + f)
extern "C" int taco_highway4_fwd_rows(const float* x0, const float* const* W4, const float* const* b4, float* const* Z4,
                                      float* const* y4, int N, int T, int f0, int f1, hipStream_t stream) {
    if (!x0 || !W4 || !b4 || !Z4 || !y4 || N <= 0 || T <= 0 || f0 < 0 || f1 > T || f0 >= f1) return TACO_EINVAL;
    return highway4_fwd_launch(x0, W4, b4, Z4, y4, N * T, f1 - f0, T, f0, N * cdiv(f1 - f0, 32), stream);
}

extern "C" int taco_highway4_bwd_rows(const float* dy, const float* const* HT4, const float* const* xin4, const float* const* W4,
                                      float* const* dZ4, float* dx, int N, int T, int f0, int f1, hipStream_t stream) {
    if (!dy || !HT4 || !xin4 || !W4 || !dZ4 || !dx || N <= 0 || T <= 0 || f0 < 0 || f1 > T || f0 >= f1) return TACO_EINVAL;
    return highway4_bwd_launch(dy, HT4, xin4, W4, dZ4, dx, N * T, f1 - f0, T, f0, N * cdiv(f1 - f0, 32), stream);
}
