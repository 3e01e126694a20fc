// Persistent GRU(128) sequence kernels for gfx950: the encoder / post-net bidirectional GRUs of the CBHG
// (reference models/modules.py:68-74: tf.nn.bidirectional_dynamic_rnn(GRUCell(128), GRUCell(128), ...)).
//
// Design (MI355X-first): the recurrence is latency-bound (T = 128..800 dependent steps), so instead of a
// launch per step, ONE workgroup owns TWO batch rows for the whole sequence and never talks to another
// workgroup: the recurrent weights W_h ([128,256] gates + [128,128] candidate = 192 KB fp32) live in the
// workgroup's REGISTER FILE for the entire kernel (192 VGPRs/lane at one wave per SIMD), the hidden state
// is broadcast through LDS, and the fp32 VALU (same peak as the fp32 MFMA on CDNA4) does the two
// mat-vec products per step.  The input half of the GRU matmul (x.W_x + b) is hoisted out of the loop as
// one big MFMA GEMM over all time steps (gemm.hip).  32 batch rows x 2 directions = 32 workgroups.
//
// GRU semantics (tf.contrib.rnn.GRUCell):  [r,u] = sigmoid([x,h].Wg + bg);
// c = tanh([x, r*h].Wc + bc);  h' = u*h + (1-u)*c.   Sequence lengths (Appendix A.6): for t >= len the
// output row is zero and the state is carried; the backward direction runs t = len-1 .. 0 from zero state.
#include "common.hpp"

#define H 128

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
};

__global__ __launch_bounds__(256, 1) void gru128_seq_fwd_k(GruSeq p) {
    const int tid = threadIdx.x;
    const int dir = blockIdx.y;
    const int row0 = blockIdx.x * 2;
    const int bj = tid >> 7, j = tid & 127;     // stage-2 mapping: (batch row, column)
    __shared__ __attribute__((aligned(16))) float h_lds[2][H];
    __shared__ __attribute__((aligned(16))) float rh_lds[2][H];
    __shared__ float u_lds[2][H];

    // recurrent weights -> registers (column tid of Wg; column j of Wc)
    float wg[H], wc[H];
    {
        const float* Wg = p.wg[dir];
        const float* Wc = p.wc[dir];
#pragma unroll
        for (int k = 0; k < H; ++k) { wg[k] = Wg[k * 256 + tid]; wc[k] = Wc[k * H + j]; }
    }
    int len[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int row = row0 + b;
        len[b] = row < p.N ? (p.lengths ? min(max(p.lengths[row], 0), p.T) : p.T) : 0;
    }
    if (tid < 2 * H) h_lds[tid >> 7][tid & 127] = 0.0f;
    __syncthreads();

    const int xoff = dir * 3 * H;
    const long ruc_dir = (long)dir * p.N * p.T * 3 * H;
    const int myrow = row0 + bj;
    const bool myrow_ok = myrow < p.N;
    const int r0c = min(row0, p.N - 1), r1c = min(row0 + 1, p.N - 1);   // clamped rows for loads

    auto tstep = [&](int s) { return dir == 0 ? s : p.T - 1 - s; };
    // prefetch step 0
    float xg0, xg1, xc;
    {
        const int t = tstep(0);
        xg0 = p.xp[((long)r0c * p.T + t) * p.ldxp + xoff + tid];
        xg1 = p.xp[((long)r1c * p.T + t) * p.ldxp + xoff + tid];
        xc = p.xp[((long)min(myrow, p.N - 1) * p.T + t) * p.ldxp + xoff + 2 * H + j];
    }
    for (int s = 0; s < p.T; ++s) {
        const int t = tstep(s);
        float a0 = xg0, a1 = xg1, ac = xc;
        if (s + 1 < p.T) {
            const int tn = tstep(s + 1);
            xg0 = p.xp[((long)r0c * p.T + tn) * p.ldxp + xoff + tid];
            xg1 = p.xp[((long)r1c * p.T + tn) * p.ldxp + xoff + tid];
            xc = p.xp[((long)min(myrow, p.N - 1) * p.T + tn) * p.ldxp + xoff + 2 * H + j];
        }
        // ---- stage 1: gates for both rows, column tid
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 h0 = *reinterpret_cast<const float4*>(&h_lds[0][k4 * 4]);
            const float4 h1 = *reinterpret_cast<const float4*>(&h_lds[1][k4 * 4]);
            a0 = fmaf(h0.x, wg[k4 * 4 + 0], a0); a1 = fmaf(h1.x, wg[k4 * 4 + 0], a1);
            a0 = fmaf(h0.y, wg[k4 * 4 + 1], a0); a1 = fmaf(h1.y, wg[k4 * 4 + 1], a1);
            a0 = fmaf(h0.z, wg[k4 * 4 + 2], a0); a1 = fmaf(h1.z, wg[k4 * 4 + 2], a1);
            a0 = fmaf(h0.w, wg[k4 * 4 + 3], a0); a1 = fmaf(h1.w, wg[k4 * 4 + 3], a1);
        }
        const float g0 = sigmoidf_(a0), g1 = sigmoidf_(a1);
        if (tid < H) {
            rh_lds[0][tid] = g0 * h_lds[0][tid];
            rh_lds[1][tid] = g1 * h_lds[1][tid];
        } else {
            u_lds[0][tid - H] = g0;
            u_lds[1][tid - H] = g1;
        }
        // save r (threads < 128) / u (threads >= 128): ruc[..][tid] for both rows
        if (row0 < p.N) p.ruc[ruc_dir + ((long)row0 * p.T + t) * 3 * H + tid] = g0;
        if (row0 + 1 < p.N) p.ruc[ruc_dir + ((long)(row0 + 1) * p.T + t) * 3 * H + tid] = g1;
        __syncthreads();
        // ---- stage 2: candidate + state update for (bj, j)
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(&rh_lds[bj][k4 * 4]);
            ac = fmaf(v.x, wc[k4 * 4 + 0], ac); ac = fmaf(v.y, wc[k4 * 4 + 1], ac);
            ac = fmaf(v.z, wc[k4 * 4 + 2], ac); ac = fmaf(v.w, wc[k4 * 4 + 3], ac);
        }
        const float c = tanhf_(ac);
        const float hprev = h_lds[bj][j];
        const float u = u_lds[bj][j];
        const float hn = u * hprev + (1.0f - u) * c;
        const bool valid = t < len[bj];
        if (myrow_ok) {
            p.ruc[ruc_dir + ((long)myrow * p.T + t) * 3 * H + 2 * H + j] = c;
            p.out[((long)myrow * p.T + t) * p.ldo + dir * H + j] = valid ? hn : 0.0f;
        }
        h_lds[bj][j] = valid ? hn : hprev;
        __syncthreads();
    }
}

// BPTT twin.  Processing order is the reverse of the forward order of that direction.
__global__ __launch_bounds__(256, 1) void gru128_seq_bwd_k(GruSeq p) {
    const int tid = threadIdx.x;
    const int dir = blockIdx.y;
    const int row0 = blockIdx.x * 2;
    const int b = tid >> 7, k = tid & 127;       // (batch row, hidden index) owner mapping
    const int jh = tid >> 7;                     // stage-C mapping: (hidden index k, j-half jh), both rows
    __shared__ __attribute__((aligned(16))) float dcp_lds[2][H];
    __shared__ __attribute__((aligned(16))) float dg_lds[2][2 * H];
    __shared__ float part_lds[2][2][H];          // [jh][row][k]

    // transposed recurrent weights -> registers
    float wcT[H];     // Wc[k][j], j = 0..127           (stage B, owner (b,k))
    float wgT[H];     // Wg[k][jh*128 + j], j = 0..127  (stage C, thread (k, jh))
    {
        const float* Wg = p.wg[dir];
        const float* Wc = p.wc[dir];
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(Wc + k * H + q * 4);
            wcT[q * 4] = a.x; wcT[q * 4 + 1] = a.y; wcT[q * 4 + 2] = a.z; wcT[q * 4 + 3] = a.w;
            const float4 g = *reinterpret_cast<const float4*>(Wg + k * 256 + jh * H + q * 4);
            wgT[q * 4] = g.x; wgT[q * 4 + 1] = g.y; wgT[q * 4 + 2] = g.z; wgT[q * 4 + 3] = g.w;
        }
    }
    const int row = row0 + b;
    const bool row_ok = row < p.N;
    const int len = row_ok ? (p.lengths ? min(max(p.lengths[row], 0), p.T) : p.T) : 0;
    const int xoff = dir * 3 * H;
    const long dirN = (long)dir * p.N;
    float dh = 0.0f;

    for (int s = 0; s < p.T; ++s) {
        const int t = dir == 0 ? p.T - 1 - s : s;          // reverse of the forward order
        const bool valid = t < len;
        const long nt = (long)(row_ok ? row : 0) * p.T + t;
        float r = 0.f, u = 0.f, c = 0.f, hprev = 0.f, dhT = 0.f;
        if (valid) {
            const float* q = p.ruc + (dirN * p.T + nt) * 3 * H;
            r = q[k]; u = q[H + k]; c = q[2 * H + k];
            const int tp = dir == 0 ? t - 1 : t + 1;       // forward-order predecessor
            if (tp >= 0 && tp < len) hprev = p.out[((long)row * p.T + tp) * p.ldo + dir * H + k];
            dhT = dh + p.dout[nt * p.lddo + dir * H + k];
        }
        const float du = dhT * (hprev - c);
        const float dc = dhT * (1.0f - u);
        float dh_new = valid ? dhT * u : dh;
        const float dcp = dc * (1.0f - c * c);
        dcp_lds[b][k] = dcp;
        __syncthreads();
        // ---- stage B: drh[b][k] = sum_j dcp[b][j] * Wc[k][j]
        float drh = 0.0f;
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(&dcp_lds[b][q * 4]);
            drh = fmaf(v.x, wcT[q * 4], drh); drh = fmaf(v.y, wcT[q * 4 + 1], drh);
            drh = fmaf(v.z, wcT[q * 4 + 2], drh); drh = fmaf(v.w, wcT[q * 4 + 3], drh);
        }
        const float dr = drh * hprev;
        dh_new += drh * r;
        const float dgr = dr * r * (1.0f - r);
        const float dgu = du * u * (1.0f - u);
        dg_lds[b][k] = dgr;
        dg_lds[b][H + k] = dgu;
        if (row_ok) {
            float* dx = p.dxp + nt * p.ldxp + xoff;
            dx[k] = dgr; dx[H + k] = dgu; dx[2 * H + k] = dcp;
            p.hp[(dirN * p.T + nt) * H + k] = hprev;
            p.rh[(dirN * p.T + nt) * H + k] = r * hprev;
        }
        __syncthreads();
        // ---- stage C: thread (k, jh): partial[row][k] = sum_{j in half jh} dg[row][j] * Wg[k][j], both rows
        float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 v0 = *reinterpret_cast<const float4*>(&dg_lds[0][jh * H + q * 4]);
            const float4 v1 = *reinterpret_cast<const float4*>(&dg_lds[1][jh * H + q * 4]);
            p0 = fmaf(v0.x, wgT[q * 4], p0); p1 = fmaf(v1.x, wgT[q * 4], p1);
            p0 = fmaf(v0.y, wgT[q * 4 + 1], p0); p1 = fmaf(v1.y, wgT[q * 4 + 1], p1);
            p0 = fmaf(v0.z, wgT[q * 4 + 2], p0); p1 = fmaf(v1.z, wgT[q * 4 + 2], p1);
            p0 = fmaf(v0.w, wgT[q * 4 + 3], p0); p1 = fmaf(v1.w, wgT[q * 4 + 3], p1);
        }
        part_lds[jh][0][k] = p0;
        part_lds[jh][1][k] = p1;
        __syncthreads();
        dh = dh_new + part_lds[0][b][k] + part_lds[1][b][k];
        // (no trailing barrier needed: the next writes to dcp_lds/dg_lds/part_lds are separated from this
        //  step's reads by the next step's barriers, except part_lds, written only after two barriers)
    }
}

extern "C" int taco_gru128_seq_fwd(const float* xp, int ldxp, const float* wg_fw, const float* wc_fw, const float* wg_bw,
                                   const float* wc_bw, const int* lengths, float* out, int ldo, float* ruc, int N, int T,
                                   int ndir, hipStream_t stream) {
    if (!xp || !wg_fw || !wc_fw || !out || !ruc || N <= 0 || T <= 0 || ndir < 1 || ndir > 2) return TACO_EINVAL;
    if (ndir == 2 && (!wg_bw || !wc_bw)) return TACO_EINVAL;
    GruSeq p{};
    p.xp = xp; p.ldxp = ldxp; p.wg[0] = wg_fw; p.wc[0] = wc_fw; p.wg[1] = wg_bw; p.wc[1] = wc_bw;
    p.lengths = lengths; p.out = out; p.ldo = ldo; p.ruc = ruc; p.N = N; p.T = T;
    hipLaunchKernelGGL(gru128_seq_fwd_k, dim3((N + 1) / 2, ndir), dim3(256), 0, stream, p);
    TACO_RETURN_LAST();
}

extern "C" int taco_gru128_seq_bwd(const float* dout, int lddo, const float* wg_fw, const float* wc_fw, const float* wg_bw,
                                   const float* wc_bw, const int* lengths, const float* out, int ldo, const float* ruc,
                                   float* dxp, int ldxp, float* hp, float* rh, int N, int T, int ndir, hipStream_t stream) {
    if (!dout || !wg_fw || !wc_fw || !out || !ruc || !dxp || !hp || !rh || N <= 0 || T <= 0 || ndir < 1 || ndir > 2) return TACO_EINVAL;
    if (ndir == 2 && (!wg_bw || !wc_bw)) return TACO_EINVAL;
    GruSeq p{};
    p.ldxp = ldxp; p.wg[0] = wg_fw; p.wc[0] = wc_fw; p.wg[1] = wg_bw; p.wc[1] = wc_bw;
    p.lengths = lengths; p.out = const_cast<float*>(out); p.ldo = ldo; p.ruc = const_cast<float*>(ruc); p.N = N; p.T = T;
    p.dout = dout; p.lddo = lddo; p.dxp = dxp; p.hp = hp; p.rh = rh;
    hipLaunchKernelGGL(gru128_seq_bwd_k, dim3((N + 1) / 2, ndir), dim3(256), 0, stream, p);
    TACO_RETURN_LAST();
}
