/* taco_hip.h -- C-ABI of the MI355X-native (gfx950) Tacotron multispeaker training-step kernels.
 *
 * The reference (Jim-Song/tacotron_multispeaker) has NO native/FFI layer: its training step is a TF-1.x
 * graph built by models/tacotron.py:18-195 and executed by sess.run at train.py:142-146.  This header is
 * therefore the build's own boundary: each entry point below names the reference graph
 * op(s) it replaces.  The Python mirror of the reference construction API (hparams.py, models/, datasets/,
 * train.py at the repo root) drives these through ctypes (tacotron_multispeaker_amd/_lib.py parses THIS file).
 *
 * Conventions
 *   - every function returns 0 on success, -22 (EINVAL) on bad arguments, or the hipError_t of a failed launch;
 *     nothing throws, allocates, frees or synchronises: all work is enqueued asynchronously on `stream`
 *     (HIP-graph capturable); the caller owns every buffer.
 *   - tensors are row-major fp32, channel-last exactly like the reference tensors ([N,T,C] == [N*T, C] with a
 *     leading dimension `ld*` in floats); ids / lengths / global_step are int32; reduction scratch is double.
 *   - leading dimensions and channel counts must be multiples of 4 and base pointers 16-byte aligned
 *     (all kernels move 16 B per lane).
 */
#ifndef TACO_HIP_H
#define TACO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* ---- dense / conv1d('same') / CBHG conv bank as fp32-MFMA implicit GEMM ------------------------------------
 * replaces tf.layers.dense (models/modules.py:10,79-89; models/tacotron.py:101) and tf.layers.conv1d
 * (models/modules.py:95-100) incl. the K-wide conv bank (models/modules.py:39-42).
 * X [M=N*T, Cin]; W TF layout [kw, Cin, Cout] (ldw = row stride of the [Cin, Cout] slices); Y [M, Cout].
 * bank_K > 0: fused bank of widths 1..bank_K, 128 channels each, W packed [sum k][Cin][128], Y [M, bank_K*128].
 * act: 0 none, 1 relu, 2 sigmoid, 3 tanh.  Dense layer: kw = 1, T = M.
 * Arithmetic: fp32 in, fp32 accumulate.  Products are exact fp32 (v_mfma_f32_32x32x2_f32) except, by default, in the input- and
 * weight-gradient entry points on LARGE problems (>= 128 output tiles of 128 x 128, >= 2 GFLOP for weight gradients), where a product is
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi of bf16 parts on v_mfma_f32_32x32x16_bf16 (error ~4e-6 of the result norm).  Environment TACO_X3, read at
 * every launch: 0 = exact fp32 products everywhere, 1 = as described (default), 2 = the forward entry points too. */
int taco_conv_gemm_fwd(const float* X, const float* W, const float* bias, float* Y, int M, int T, int Cin, int Cout,
                       int kw, int bank_K, int ldx, int ldw, int ldy, int act, int accumulate, hipStream_t stream);
/* the same over the FRAMES [t0, t1) of every length-T sequence of X / Y [N*T, .] (taps read the full sequences: the rows a tap
 * reaches outside [t0, t1) must already be final): the post-net conv bank runs chunk by chunk behind the decoder pipeline */
int taco_conv_rows_fwd(const float* X, const float* W, const float* bias, float* Y, int N, int T, int t0, int t1, int Cin,
                       int Cout, int kw, int bank_K, int ldx, int ldw, int ldy, int act, double* bn_stat, hipStream_t stream);
/* conv + activation with the batch-norm sums of the output (tf.layers.batch_normalization after the conv, models/modules.py:95-101)
 * accumulated from the accumulators in the GEMM epilogue instead of by a pass over Y; bn_stat: TACO_BN_DSTAT(Cout) zeroed doubles
 * (taco_conv_rows_fwd: optional, may be NULL), finalised by taco_bn_finalize */
int taco_conv_gemm_bn_fwd(const float* X, const float* W, const float* bias, float* Y, int M, int T, int Cin, int Cout, int kw,
                          int bank_K, int ldx, int ldw, int ldy, int act, double* bn_stat, hipStream_t stream);
/* dX (+)= conv_transpose(dY, W)   (gradient of the above wrt X) */
int taco_conv_gemm_bwd_data(const float* dY, const float* W, float* dX, int M, int T, int Cin, int Cout, int kw,
                            int bank_K, int lddy, int ldw, int lddx, int accumulate, hipStream_t stream);
/* input gradient over the FRAMES [t0, t1) of every length-T sequence (reads dY rows of the whole sequences); a piece that is split
 * over the taps adds partial sums atomically and therefore requires accumulate != 0 (dX pre-initialised by the caller) */
int taco_conv_rows_bwd_data(const float* dY, const float* W, float* dX, int N, int T, int t0, int t1, int Cin, int Cout, int kw,
                            int bank_K, int lddy, int ldw, int lddx, int accumulate, hipStream_t stream);
/* dW += X^T (shifted per tap) . dY   (atomic accumulation into a zeroed / running gradient buffer) */
int taco_conv_gemm_bwd_weight(const float* X, const float* dY, float* dW, int M, int T, int Cin, int Cout, int kw,
                              int bank_K, int ldx, int lddy, int ldw, hipStream_t stream);
/* dW[K,N] += sum_m X[m+shift, :]^T dY[m, :], rows leaving their length-T sequence contribute 0 (recurrent weights) */
int taco_gemm_tn_shift(const float* X, const float* dY, float* dW, int M, int T, int K, int N, int ldx, int lddy,
                       int ldw, int shift, hipStream_t stream);

/* Grouped weight gradients: `count` independent problems of taco_conv_gemm_bwd_weight (shift = 0) / taco_gemm_tn_shift in ONE
 * launch (the problem table travels in the kernel arguments; problems that need another tile configuration are launched on
 * their own).  items is a HOST array, read before the call returns. */
typedef struct TacoWgrad {
    const float* X; const float* dY; float* dW;
    int M, T, Cin, Cout, kw, bank_K, ldx, lddy, ldw, shift;
    float* dbias;       /* optional: dbias[c] += sum_m dY[m, c] (bias gradient of a dense layer): summed from the dY tiles the weight-
                           gradient GEMM stages in LDS anyway, no pass of its own over dY */
} TacoWgrad;
int taco_wgrad_group(const TacoWgrad* items, int count, hipStream_t stream);
/* the same for bias gradients: out_i[c] += sum_m x_i[m, c] for `count` tensors in one launch (HOST array) */
typedef struct TacoColSum { const float* x; float* out; int ldx, M, C; } TacoColSum;
int taco_col_sum_group(const TacoColSum* items, int count, hipStream_t stream);

/* dense layer forward / input gradient over the step chunk [s0, s1) of [N,S,*] tensors (rows (n,s) live at n*S + s);
 * lets the hoisted decoder projections be issued chunk by chunk between pipelined recurrence launches */
int taco_dense_rows_fwd(const float* X, const float* W, const float* bias, float* Y, int N, int S, int s0, int s1, int Cin,
                        int Cout, int ldx, int ldw, int ldy, int act, int accumulate, hipStream_t stream);
int taco_dense_rows_bwd_data(const float* dY, const float* W, float* dX, int N, int S, int s0, int s1, int Cin, int Cout,
                             int lddy, int ldw, int lddx, int accumulate, hipStream_t stream);

/* ---- embeddings: tf.nn.embedding_lookup + speaker lookup/tile/concat (models/tacotron.py:42-55) -------------- */
int taco_embed_gather_fwd(const int* ids, const int* identities, const float* table, const float* spk_table, float* out,
                          int N, int Ti, int Et, int Es, int vocab, int id_num, hipStream_t stream);
/* scatter-add of dE [N*Ti, Et+Es] into the tables; sparse_sumsq[0] += sum of squares of the un-deduplicated
 * IndexedSlices rows (tf.global_norm quirk); may be NULL */
int taco_embed_scatter_bwd(const int* ids, const int* identities, const float* dE, float* dTable, float* dSpk,
                           double* sparse_sumsq, int N, int Ti, int Et, int Es, int vocab, int id_num, hipStream_t stream);

/* ---- batch norm (tf.layers.batch_normalization, models/modules.py:101), maxpool (modules.py:45-49), residual -- */
int taco_col_sum(const float* x, int ldx, float* out, int M, int C, hipStream_t stream);   /* out[c] += sum_m x[m,c] */
/* dstat_zeroed: TACO_BN_DSTAT(C) doubles of zeroed scratch (replicated per-column sums; same-address atomics from
 * hundreds of workgroups serialise in L2, so the reductions land in TACO_BN_REPL replicas that the consumers add up) */
#define TACO_BN_REPL 8
#define TACO_BN_DSTAT(C) (TACO_BN_REPL * 3 * (C))
int taco_bn_stats_fwd(const float* x, int ldx, const float* gamma, const float* beta, double* dstat_zeroed, float* mean,
                      float* var, float* rstd, float* scale, float* shift, int M, int C, float eps, hipStream_t stream);
/* in two parts, for a tensor produced in pieces: per-column sums over the frames [t0, t1) of every length-T sequence (any number of
 * calls into the same zeroed dstat), then the finalisation over all M = N*T rows */
int taco_bn_stats_rows(const float* x, int ldx, double* dstat, int N, int T, int t0, int t1, int C, hipStream_t stream);
int taco_bn_finalize(const double* dstat, const float* gamma, const float* beta, float* mean, float* var, float* rstd,
                     float* scale, float* shift, int M, int C, float eps, hipStream_t stream);
int taco_bn_infer_params(const float* moving_mean, const float* moving_var, const float* gamma, const float* beta,
                         float* scale, float* shift, int C, float eps, hipStream_t stream);
/* y = [max over (t, t+1)] (x*scale+shift) [+ res] */
int taco_bn_apply_fwd(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldr, float* y,
                      int ldy, int M, int C, int T, int pool, hipStream_t stream);
/* dy = gradient wrt the BN output (pool=1: wrt the pooled output); dx = gradient wrt the conv pre-activation
 * (relu=1 applies the conv's ReLU mask); dgamma/dbeta and the conv bias gradient dbias[c] = sum_m dx[m,c] are ADDED */
int taco_bn_bwd(const float* x, int ldx, const float* dy, int lddy, const float* mean, const float* rstd,
                const float* scale, const float* shift, const float* gamma, double* dstat_zeroed, float* dgamma,
                float* dbeta, float* dbias, float* dx, int lddx, int M, int C, int T, int pool, int relu, hipStream_t stream);

/* ---- highway gating (models/modules.py:77-90); Z [M,256] = x.[W_H|W_T]+b in, [relu(H), sigmoid(T)] out -------- */
int taco_highway_gate_fwd(float* Z, const float* x, float* y, int M, hipStream_t stream);
int taco_highway_gate_bwd(const float* HT, const float* x, const float* dy, float* dZ, float* dx, int M, hipStream_t stream);
/* the four highway layers of a CBHG in ONE launch per direction (csrc/highway.hip); W4 / b4 / Z4 / y4 ... are HOST arrays of four
 * device pointers (layer 1..4).  forward: Z4[l] [M,256] <- [relu(H) | sigmoid(T)], y4[l] [M,128] <- layer outputs.
 * backward: dy = gradient wrt y4[3]; HT4 = the saved Z4, xin4[l] = input of layer l (x0, y4[0..2]); dZ4[l] [M,256] <- gradient
 * wrt the pre-activations (operand of the dW / bias gradients), dx [M,128] <- gradient wrt x0. */
int taco_highway4_fwd(const float* x0, const float* const* W4, const float* const* b4, float* const* Z4, float* const* y4,
                      int M, hipStream_t stream);
int taco_highway4_bwd(const float* dy, const float* const* HT4, const float* const* xin4, const float* const* W4,
                      float* const* dZ4, float* dx, int M, hipStream_t stream);
/* the same over the FRAMES [f0, f1) of every length-T sequence of the [N,T,*] tensors (rows n*T + f; tiles never cross a sequence) */
int taco_highway4_fwd_rows(const float* x0, const float* const* W4, const float* const* b4, float* const* Z4, float* const* y4,
                           int N, int T, int f0, int f1, hipStream_t stream);
int taco_highway4_bwd_rows(const float* dy, const float* const* HT4, const float* const* xin4, const float* const* W4,
                           float* const* dZ4, float* dx, int N, int T, int f0, int f1, hipStream_t stream);
int taco_relu_bwd(const float* y, const float* dy, float* dpre, long n, hipStream_t stream);
int taco_add(const float* a, const float* b, float* y, long n, int accumulate, hipStream_t stream);

/* ---- L1 losses + sign gradients (models/tacotron.py:127-137) ----------------------------------------------------
 * sums2: 2 * TACO_L1_REPL zeroed doubles; replica r holds {sum |d| over all columns, sum |d| over columns < npri} of the
 * workgroups with index % TACO_L1_REPL == r -- the caller adds the replicas up */
#define TACO_L1_REPL 8
int taco_l1_loss(const float* out, int ldo, const float* tgt, int ldt, float* grad, int ldg, double* sums2, long rows,
                 int C, int npri, float w_all, float w_pri, hipStream_t stream);
/* the same over the FRAMES [f0, f1) of every length-T sequence of [N,T,*] tensors (rows n*T + f): lets the loss of the frames a
 * chunk of the post-net biGRU completes run beside the next chunk; the sums ADD into sums2 (zeroed once per step by the caller) */
int taco_l1_loss_rows(const float* out, int ldo, const float* tgt, int ldt, float* grad, int ldg, double* sums2, int N, int T,
                      int f0, int f1, int C, int npri, float w_all, float w_pri, hipStream_t stream);

/* ---- optional alignment regularisers (models/tacotron.py:140-171; hparams overwrought / oneorder_dynamic /
 * variance_between_row / alignment_entropy, all 0.0 by default).  align [N,S,Ti] (the layout the attention kernels save;
 * the reference tensor is its [N,Ti,S] transpose).  Adds the regulariser value to loss_sum[0] and WRITES its gradient wrt
 * the alignments to dalign [N,S,Ti], which taco_attn_rnn_bwd consumes through TACO_AP_DAEXT.  The reference applies a second
 * softmax over the decoder-step axis to the alignments (:142); `overwrought` needs S >= 40 (tf.slice at :159). */
int taco_align_regularity(const float* align, float* dalign, double* loss_sum, int N, int S, int Ti, float overwrought,
                          float oneorder_dynamic, float variance_between_row, float alignment_entropy, hipStream_t stream);

/* ---- persistent (bi)GRU(128) sequence kernels: tf.nn.bidirectional_dynamic_rnn (models/modules.py:68-74) -------
 * xp [N,T,ldxp]: hoisted x.W_x + b per direction d at columns [d*384, d*384+384) ordered r|u|c;
 * wg [128,256], wc [128,128]: recurrent halves of the GRUCell gates/candidate kernels;
 * out [N,T,ldo] direction d at columns [d*128, ..); ruc [ndir,N,T,384] saved gates for BPTT. */
/* [s0, s1): step range of this launch in processing order (forward pass: step s touches frame s of direction 0 and frame T-1-s of
 * direction 1; BPTT: the mirror image).  The recurrence may be cut into chunk launches issued in order, so that work on the frames a
 * chunk completes (both directions done) can run beside the next chunk; the recurrent state passes through state [ndir,N,128]
 * (required unless the launch covers [0, T)).  isolate_lds_bytes: see taco_gru256_seq_fwd. */
int taco_gru128_seq_fwd(const float* xp, int ldxp, const float* wg_fw, const float* wc_fw, const float* wg_bw,
                        const float* wc_bw, const int* lengths, float* out, int ldo, float* ruc, int N, int T, int ndir,
                        int s0, int s1, float* state, int isolate_lds_bytes, hipStream_t stream);
int taco_gru128_seq_bwd(const float* dout, int lddo, const float* wg_fw, const float* wc_fw, const float* wg_bw,
                        const float* wc_bw, const int* lengths, const float* out, int ldo, const float* ruc, float* dxp,
                        int ldxp, float* hp, float* rh, int N, int T, int ndir, int s0, int s1, float* state,
                        int isolate_lds_bytes, hipStream_t stream);

/* ---- attention decoder (models/tacotron.py:66-97, rnn_wrappers.py, helpers.py:41-82) ---------------------------- */
int taco_gather_frames(const float* mel, float* frames, int N, int S, int r, int num_mels, hipStream_t stream);
/* attention recurrence; ptrs = device pointer table indexed by enum TacoAttnPtr, dims = {N, S, Ti, s0, s1} (host arrays);
 * [s0, s1) = step range of this launch (cluster path only: chunks launched in order can be pipelined with the decoder GRUs;
 * state passes through HC (forward) and the DHCARRY / DCTXCARRY slots (backward)); whole sequence: s0 = 0, s1 = S */
int taco_attn_rnn_fwd(const void* const* ptrs, const int* dims, hipStream_t stream);
int taco_attn_rnn_bwd(const void* const* ptrs, const int* dims, hipStream_t stream);
/* The same for ONE chunk of a chunked BPTT with the chunk-to-chunk carries (dh, dctx at the boundary) handed over as granules
 * instead of through memory between two launches: TACO_ATTN_CARRY_POST -- this launch publishes its final carries as granules;
 * TACO_ATTN_CARRY_WAIT -- this launch (the chunk [s0, s1) with s1 < S) may be started on another stream BESIDE the launch of the
 * later chunk [s1, ..): it counts its workgroups in a residency counter as they start, runs its prologue and then polls for the
 * carries.  The caller uses that to have the last chunk resident before the one before it ends, so that no other work can take the
 * CUs between two chunk launches (engine.py, round 3); taco_wait_count on the residency counter gates such work.  A WAIT launch
 * needs a TACO_AP_XCHG buffer of its OWN (same size, zero-filled by the caller once per pass: two kernels running side by side
 * must not share step-exchange slots) and carry_xchg = the TACO_AP_XCHG buffer of the POST launch (its carry region and counter
 * are the ones used); carry_xchg = NULL otherwise.
 * TACO_ATTN_NO_REDUCE -- the launch of the chunk with s0 == 0 does NOT enqueue the reduction over the steps behind itself
 * (dKEYS / dMEM / dVPART from the per-step DE / DCTXS of ALL chunks): a WAIT launch ends after the carries of its predecessor
 * were published, which is before that predecessor's kernel -- on another stream -- has retired and its last stores are ordered;
 * the caller joins the two streams and then calls taco_attn_bwd_reduce. */
#define TACO_ATTN_CARRY_POST 1
#define TACO_ATTN_CARRY_WAIT 2
#define TACO_ATTN_NO_REDUCE 4
int taco_attn_rnn_bwd_chunk(const void* const* ptrs, const int* dims, int carry_flags, void* carry_xchg, hipStream_t stream);
int taco_attn_bwd_reduce(const void* const* ptrs, const int* dims, hipStream_t stream);
/* index (in 8-byte slots) of the residency counter (an int) inside the TACO_AP_XCHG buffer of an (N, Ti) launch; workgroups per launch */
int taco_attn_bwd_resident_slot(int N, int Ti);
int taco_attn_bwd_workgroups(int N);
/* one wave that waits until *counter >= target (bounded: sets err[0] after ~0.5 s): orders work on a stream behind an event
 * that only a running kernel can signal; enqueue it only after the launch that signals the counter has been issued */
int taco_wait_count(const int* counter, int target, int* err, hipStream_t stream);
/* free-running inference decoder (reference models/helpers.py:7-38 TacoTestHelper, models/tacotron.py:86-94):
 * ptrs indexed by enum TacoInferPtr (all required), dims = {N, S = max_iters (row stride of the [N,S,*] outputs), Ti, r,
 * num_mels, s0, s1}: enqueues decoder steps [s0, s1); state carries over between calls through HC / OUT / H1 / H2.
 * TACO_IP_STOP implements the reference's stop token: the caller sets stop[0] = S and stop[1..N] = 0 before step 0; after the
 * first step at which every row has produced an all-zero r-frame output (sticky per row) stop[0] = that step + 1. */
int taco_decoder_infer(const void* const* ptrs, const int* dims, hipStream_t stream);
/* persistent-cluster path of the attention recurrence: shape support (1/0) and granule scratch size in 8-byte slots */
int taco_attn_cluster_supported(int N, int Ti);
int taco_attn_cluster_xchg_slots(int N, int Ti);
int taco_attn_cluster_bwd_xchg_slots(int N, int Ti);
/* which attention-BPTT kernel taco_attn_rnn_bwd runs for (N, Ti): 0 per-step kernels, 1 cluster kernel with the prenet-gradient
 * weight slices in LDS (T_in <= ~152), 2 cluster kernel with all weight slices in registers (longer inputs) */
int taco_attn_cluster_bwd_variant(int N, int Ti);
/* the same for taco_attn_rnn_fwd: 1 = recurrent gate weights in LDS (T_in <= ~160), 2 = in registers */
int taco_attn_cluster_fwd_variant(int N, int Ti);
/* residual decoder GRU(256), whole recurrence in one persistent cluster launch (csrc/gru256.hip); hoisted input
 * projection xp [N,S,768]; d = res + h when d != NULL.  xchg: >= ceil(N/2)*6*256 8-byte granule slots of scratch,
 * err: device int[4]: [0] set to 1 if a bounded spin ever times out (results are then invalid), [1] / [2] placement statistics
 * (clusters on the agent-scope fallback / clusters checked, see TACO_AP_ERR).  N <= 128.
 * [s0, s1): step range of this launch -- the recurrence may be cut into chunks launched in order (forward: ascending,
 * backward: descending) so that chunks of different recurrences can be pipelined on different streams; the state passes
 * through h (forward) and carry [N,256] (backward).
 * isolate_lds_bytes: unused dynamic LDS per workgroup (0 = none, at most 155000): with ~150 KB a workgroup owns its CU and no GEMM
 * workgroup shares its SIMDs; the caller decides (it knows the batch, the other persistent launches in flight and whether
 * collective kernels need room on those CUs). */
int taco_gru256_seq_fwd(const float* xp, const float* whg, const float* whc, const float* res, float* r, float* u, float* c,
                        float* rh, float* h, float* d, void* xchg, int* err, int N, int S, int s0, int s1, int isolate_lds_bytes,
                        hipStream_t stream);
int taco_gru256_seq_bwd(const float* dout, const float* whg, const float* whc, const float* r, const float* u, const float* c,
                        const float* h, float* dxp, float* carry, void* xchg, int* err, int N, int S, int s0, int s1,
                        int isolate_lds_bytes, hipStream_t stream);

/* ---- optimizer: tf.clip_by_global_norm + tf.train.AdamOptimizer + Noam lr + BN UPDATE_OPS (tacotron.py:174-202) -- */
int taco_sumsq(const float* x, long n, double* acc, hipStream_t stream);
/* err (optional, all three): the device error word of the persistent cluster kernels (TACO_AP_ERR); when it is non-zero at
 * execution time the launch does nothing, so a step whose hand-off timed out never reaches the weights.
 * taco_bn_ema also increments *global_step when it is non-NULL (one launch for UPDATE_OPS + the step counter). */
/* grad_scale: grads (and gnorm2 = their sum of squares) hold 1/grad_scale times the gradient -- data parallel: the all-reduced
 * SUM over replicas with grad_scale = 1/world; the clip norm and the update use grad_scale * grads (no separate averaging pass) */
int taco_adam_step(float* params, const float* grads, float* m, float* v, long n, const double* gnorm2,
                   const int* global_step, float init_lr, int decay, float beta1, float beta2, float eps, float clip,
                   float grad_scale, float* info3, const int* err, hipStream_t stream);
int taco_bn_ema(float* moving, const float* batch, int n, float momentum, int* global_step, const int* err, hipStream_t stream);
int taco_step_inc(int* global_step, const int* err, hipStream_t stream);
/* What the step loop reads back (reference train.py:142-152 fetches global_step, loss, loss_regularity; :23-36 the summary
 * scalars), packed into out16 = 16 doubles: 0 mel L1 sum, 1 linear L1 sum over all columns, 2 over the priority columns (the sums2
 * replicas of taco_l1_loss added up), 3 loss_regularity, 4 global norm before the clip, 5 learning rate, 6 clip factor,
 * 7 global_step, 8 err4[0] (hand-off timeout), 9 err4[1] (clusters on the agent-scope fallback), 10 err4[2] (clusters that ran the
 * placement check), rest 0.  loss_sums = mel sums2 followed by the linear sums2 (2 * 2 * TACO_L1_REPL doubles); optional inputs may
 * be NULL.  One launch + one 128-byte device-to-host copy behind an event replace the blocking per-value reads. */
int taco_step_status(const double* loss_sums, const double* reg_sum, const float* info3, const int* err4, const int* global_step,
                     double* out16, hipStream_t stream);
int taco_scale(float* x, long n, float s, hipStream_t stream);
/* a one-wave kernel that idles for `us` microseconds (<= 100 ms): the host's probe of which streams share a hardware queue
 * (tacotron_multispeaker_amd/engine.py Engine._pick_streams); not part of a training step */
int taco_spin_us(int us, hipStream_t stream);
/* zero-fill (one memset node): bytes and p must be multiples of 16 */
int taco_zero(void* p, size_t bytes, hipStream_t stream);

/* pointer-table slots of taco_attn_rnn_fwd / taco_attn_rnn_bwd (all fp32 device pointers) */
enum TacoAttnPtr {
    TACO_AP_W1C = 0,   /* decoder_prenet dense_1 kernel rows 80:336 (context part)  [256,256] */
    TACO_AP_F1,        /* hoisted frame part of dense_1 (+bias)                     [N,S,256] */
    TACO_AP_W2, TACO_AP_B2,   /* decoder_prenet dense_2                             [256,128],[128] */
    TACO_AP_WX,        /* attention GRU input weights, gates|candidate              [128,768] */
    TACO_AP_WHG,       /* attention GRU recurrent gate weights                      [256,512] */
    TACO_AP_WHC,       /* attention GRU recurrent candidate weights                 [256,256] */
    TACO_AP_BG,        /* attention GRU biases gates|candidate                      [768] */
    TACO_AP_WQ, TACO_AP_V,    /* BahdanauAttention query_layer kernel, attention_v   [256,256],[256] */
    TACO_AP_KEYS,      /* memory_layer(encoder_outputs)                             [N,Ti,256] */
    TACO_AP_MEM,       /* encoder_outputs (attention values)                        [N,Ti,256] */
    TACO_AP_ZEROS,     /* >= N*256 zeros (initial states)                            */
    TACO_AP_P1, TACO_AP_P2,                      /* saved prenet activations  [N,S,256],[N,S,128] */
    TACO_AP_R, TACO_AP_U, TACO_AP_C, TACO_AP_RH, /* saved GRU gates           [N,S,256] each */
    TACO_AP_HC,        /* [h_s | ctx_s] (input of the concat projection)            [N,S,512] */
    TACO_AP_Q,         /* saved queries                                             [N,S,256] */
    TACO_AP_ALIGN,     /* alignments a_s                                            [N,S,Ti] */
    /* backward only */
    TACO_AP_DHC,       /* in:  gradient wrt [h_s | ctx_s] from the concat projection [N,S,512] */
    TACO_AP_DXP,       /* out: attention GRU pre-activation gradients               [N,S,768] */
    TACO_AP_DP2, TACO_AP_DP1, /* out: prenet pre-activation gradients         [N,S,128],[N,S,256] */
    TACO_AP_DQ,        /* out (pre-zeroed): query gradients                         [N,S,256] */
    TACO_AP_DKEYS, TACO_AP_DMEM, /* out (pre-zeroed accumulators)                   [N,Ti,256] each */
    TACO_AP_DVPART,    /* out: attention_v partial gradients; per-step path [N,ceil(Ti/16),256] pre-zeroed, cluster path [N*Ti,256] */
    TACO_AP_DA,        /* scratch [N,Ti] */
    TACO_AP_DHT, TACO_AP_DHPART, TACO_AP_DHCARRY, TACO_AP_DCTX, TACO_AP_DCTXCARRY, /* scratch [N,256] each */
    TACO_AP_XCHG,      /* optional: >= taco_attn_cluster_xchg_slots(N,Ti) 8-byte granule slots (enables the cluster path) */
    TACO_AP_ERR,       /* optional: device int[4]: [0] set to 1 if a bounded hand-off spin timed out, [1] clusters that kept the
                          agent-scope granule form although the L2-local form was allowed, [2] clusters that ran the placement check */
    TACO_AP_DE,        /* cluster bwd: out, softmax-input gradients de_s[t]           [N,S,Ti] */
    TACO_AP_DCTXS,     /* cluster bwd: out, total context gradients per step          [N,S,256] */
    TACO_AP_DAEXT,     /* optional in: extra gradient wrt the alignments (taco_align_regularity) [N,S,Ti]; may be NULL */
    TACO_AP_COUNT
};

/* pointer-table slots of taco_decoder_infer (fp32 device pointers; weights as in the flat parameter layout) */
enum TacoInferPtr {
    TACO_IP_W1 = 0, TACO_IP_B1,      /* decoder_prenet dense_1 kernel [num_mels+256,256] (frame rows first), bias */
    TACO_IP_W2, TACO_IP_B2,          /* decoder_prenet dense_2 */
    TACO_IP_WX, TACO_IP_WHG, TACO_IP_WHC, TACO_IP_BG,       /* attention GRU (as TacoAttnPtr) */
    TACO_IP_WQ, TACO_IP_V, TACO_IP_KEYS, TACO_IP_MEM, TACO_IP_ZEROS,
    TACO_IP_WP, TACO_IP_BP,          /* concat projection [512,256] */
    TACO_IP_G1WX, TACO_IP_G1B, TACO_IP_G1WHG, TACO_IP_G1WHC,  /* decoder GRU 1: wx [256,768], bias [768], whg, whc */
    TACO_IP_G2WX, TACO_IP_G2B, TACO_IP_G2WHG, TACO_IP_G2WHC,
    TACO_IP_WO, TACO_IP_BO,          /* output projection [256, num_mels*r] */
    TACO_IP_HC,                      /* out: [h_s | ctx_s]                    [N,S,512] */
    TACO_IP_ALIGN,                   /* out: alignments                       [N,S,Ti] */
    TACO_IP_OUT,                     /* out: decoder outputs                  [N,S,num_mels*r] */
    TACO_IP_H1, TACO_IP_H2,          /* scratch: GRU states, ping-pong        [2,N,256] each */
    TACO_IP_TMP,                     /* scratch                               [10,N,256] */
    TACO_IP_STOP,                    /* int32 [1+N]: steps to keep, per-row finished flags (see taco_decoder_infer) */
    TACO_IP_COUNT
};

#ifdef __cplusplus
}
#endif
#endif /* TACO_HIP_H */
