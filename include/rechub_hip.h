/*
 * rechub_hip.h — C ABI of librechub_hip.so, the MI355X (gfx950) CTR-training hot path.
 *
 * Every entry point replaces an ATen op chain of the reference (datawhalechina/torch-rechub
 * v0.8.0, 100 % Python on eager PyTorch — there is no FFI in the reference, so the
 * "interface replaced" is the Python call site cited per function, paths relative to
 * /root/reference).  Signatures carry plain pointers and sizes only: no torch types.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers unless the name starts with h_;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return value 0 = success, >0 = hipError_t of a failed launch, <0 = RH_E_* argument error
 *     (rh_last_error() returns a human-readable message for the calling thread);
 *   - kernels are asynchronous; nothing here synchronises, allocates or frees device memory,
 *     so every call may be captured into a hipGraph;
 *   - `err_flag` (device int32, may be NULL) is OR-ed with RH_FLAG_* bits by kernels that
 *     validate indices; the host reads it at its own sync points.
 *
 * Field descriptor tables (device int64 arrays, struct-of-arrays, F = number of fields):
 *   fdesc[0*F + f] = const float* table base of field f        (vocab_f x D, row-major, fp32)
 *   fdesc[1*F + f] = float*       dense gradient buffer of that table (same shape) or 0
 *   fdesc[2*F + f] = int64        vocab_f (rows)
 *   fdesc[3*F + f] = int64        padding_idx of field f, or -1 (rows whose gradient is dropped,
 *                                 nn.Embedding(padding_idx=...) semantics, basic/initializers.py:16-21)
 *   idesc[0*F + f] = const idx_t* index column of field f (one index per sample)
 *   idesc[1*F + f] = int64        stride between consecutive samples, in elements
 *   idesc[2*F + f] = int64        output slot of field f: its D values live at columns
 *                                 [slot*D, slot*D + D) of out / g_out / emb (slot = f unless sequence
 *                                 features are interleaved with sparse ones, basic/layers.py:80-99)
 * Two fields may name the same table (SparseFeature.shared_with, basic/layers.py:85,99).
 */
#ifndef RECHUB_HIP_H
#define RECHUB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_ABI_VERSION 1

/* argument errors */
#define RH_E_BADARG (-1)
#define RH_E_UNSUPPORTED (-2)

/* err_flag bits */
#define RH_FLAG_INDEX_OOB 1 /* an index was <0 or >= vocab (reference: IndexError / device assert) */

int rh_abi_version(void);
const char* rh_last_error(void);

/* tuning knobs (process-wide; defaults are the measured winners) */
#define RH_TUNE_WIDE_ATOMICS 1 /* 1: table-gradient atomics carry whole rows per request (default), 0: 16 B pieces */
#define RH_TUNE_SWEEP_GRID 2   /* workgroups of rh_adam_lazy_sweep (0 = default 8192) */
/* key 3 (an LDS-padding residency cap of the deferred sweep) was measured in round 3 and removed: rh_set_tuning(3, .) fails */
#define RH_TUNE_DEFERRED_GRID 8 /* persistent workgroups of a DEFERRED sweep (default 512 = 2 per CU; 0 = as RH_TUNE_SWEEP_GRID):
                                  the residency cap that lets the step's chain keep its wave slots and issue cycles */
#define RH_TUNE_SWEEP_STAGGER_NS 12 /* rh_adam_sweep_stagger: hold-back in nanoseconds (default 15000; 0 = no launch) */
#define RH_TUNE_SWEEP_GATE_NS 13 /* rh_adam_sweep_gate(fallback_ns = 0): hold-back behind the opening, ns (default 32000) */
#define RH_TUNE_SWEEP_WIDE 14 /* deferred window sweep of the lazy tables: float4 per lane at embed_dim >= 8 (2 = default: two
                                 independent float4 chains per lane, round 5; 1 = one float4 per lane, the round-4 kernel) */
#define RH_TUNE_WGRAD_RIDER_ORDER 15 /* rh_adam_lazy_step_ahead_wgrad: weight-gradient workgroups 2 = behind the optimizer's parts
                                      * (default), 0 = in front of them, 1 = dealt alternately with them */
#define RH_TUNE_WGRAD_BLOCKS 9 /* workgroups rh_linear_wgrad aims for when the reduction is >= 32768 rows (default 1024) */
#define RH_TUNE_WGRAD_ROWS_FORM 16 /* rh_linear_wgrad, reductions >= 32768 rows whose output is 2 .. 8 tiles of 64 x 64: value > 0 = ONE
                                      workgroup per row split computes the whole (N, K) slab, `value` workgroups aimed for (default 512), two
                                      tiles per wavefront where the tile counts pair up; value < 0 = -value workgroups, one tile per
                                      wavefront; 0 = one workgroup per tile and split (rounds 1-5) */
#define RH_TUNE_WGRAD_SHORT_FORM 11 /* rh_linear_wgrad at B < 32768: 0 = 206-register build, 1 = the 128-register build (default) */
#define RH_TUNE_DICE_VEC 10    /* bit mask of lanes-per-row (16 | 32 | 64) for which the Dice passes use the rows-per-wavefront
                                  kernel (C = 64 / 128 / 256); default 16 | 32 */
#define RH_TUNE_BWD_SPLIT 4     /* retired (accepted, ignored) */
#define RH_TUNE_BWD_SLABS 5     /* retired (accepted, ignored) */
#define RH_TUNE_FWD_PATH 7      /* rh_embed_fwd: 0 auto (by batch size), 1 lane-split kernel only, 2 field-uniform kernel only */
#define RH_TUNE_BWD_PATH 6      /* rh_embed_bwd experiments: 0 auto, 1 global atomics for every table, 3 no sink, 4 chunk-fastest
                                 * block order; TIMING ONLY (wrong sums on colliding rows, tools/bwd_ceiling_probe.py): 5 the
                                 * row-wide requests as plain stores, 6 one plain 16-byte store per lane */
int rh_set_tuning(int key, int value);

/* ---------------------------------------------------------------------------------------------
 * K1+K2+K3  fused multi-field gather + FM second order + LR first order (+ dense concat)
 * replaces: EmbeddingLayer.forward  torch_rechub/basic/layers.py:77-127  (26x nn.Embedding + cat)
 *           FM.forward              torch_rechub/basic/layers.py:313-319
 *           LR.forward              torch_rechub/basic/layers.py:185-189 on the flattened embeddings
 *           as composed by DeepFM.forward torch_rechub/models/ranking/deepfm.py:34-43
 * out   : (B, out_stride) fp32, rows only 4-byte aligned (e.g. a contiguous (B, 429) tensor); field f
 *         at columns [slot_f*D, slot_f*D + D), dense values at [dense_col, dense_col + n_dense)
 *         (sparse block first, dense last: layers.py:120), other columns untouched.
 * ddesc : device int64 [2*n_dense]: const float* column, sample stride (elements); NULL if n_dense = 0
 * lr_w  : (F*D,) indexed by FIELD (needs slot_f == f) or NULL; lr_b: (1,) or NULL;
 *         lr_out (B,) = emb . lr_w + lr_b
 * fm_out: (B,) or NULL = 0.5 * sum_d[(sum_f v)^2 - sum_f v^2]
 * s_out : (B, D) or NULL = sum_f v  (saved for the backward)
 * D must be a multiple of 4 and <= 128 (16-byte lanes); idx_is_i64: 1 = int64 indices, 0 = int32.
 * field_split: 0 = auto, else 1/2/4/8 lanes-groups per sample (tuning knob).
 */
int rh_embed_fwd(const int64_t* fdesc, const int64_t* idesc, int idx_is_i64, int B, int F, int D,
                 const int64_t* ddesc, int n_dense, int dense_col, float* out, int64_t out_stride,
                 const float* lr_w, const float* lr_b, float* lr_out, float* fm_out, float* s_out,
                 int field_split, int32_t* err_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4  fused backward of the above
 * replaces: autograd of the chain above (embedding_dense_backward x F, FM/LR backward),
 *           driven by loss.backward() at torch_rechub/trainers/ctr_trainer.py:97-98
 * per lookup (b,f):  g = scale * ( g_out[b, f*D:(f+1)*D] + g_lr[b]*lr_w[f*D:(f+1)*D]
 *                                  + g_fm[b] * (s_sum[b,:] - emb[b, f*D:(f+1)*D]) )
 * sink = 0: scatter-add g into the dense gradient buffer fdesc[1*F+f] (skipping padding_idx rows);
 * sink = 1: write g to rows_out (B, F, D) for the data-parallel exchange.
 * Any of g_out / g_fm / g_lr may be NULL (term dropped).  emb/s_sum are required when g_fm != NULL,
 * emb when lr_wgrad != NULL.
 * lr_wgrad : (nchunks, F*D) partial sums of g_lr[b]*emb[b,:] per sample chunk, nchunks =
 *            ceil(B / samples_per_block); the caller reduces over dim 0 (deterministic). NULL to skip.
 * samples_per_block: multiple of 64, 0 = auto (256).
 */
int rh_embed_bwd(const int64_t* fdesc, const int64_t* idesc, int idx_is_i64, int B, int F, int D,
                 const float* g_out, int64_t g_stride, const float* emb, int64_t emb_stride,
                 const float* s_sum, const float* g_fm, const float* g_lr, const float* lr_w,
                 float* lr_wgrad, float scale, int sink, float* rows_out, int samples_per_block,
                 int32_t* err_flag, void* stream);

/* number of sample chunks (= rows of lr_wgrad) rh_embed_bwd uses for this B / samples_per_block */
int rh_embed_bwd_nchunks(int B, int samples_per_block);

/* scatter-add precomputed gradient rows (B, F, D) (e.g. all-gathered from the other ranks)
 * into the dense gradient buffers.  Same padding_idx and small-vocab LDS aggregation as above. */
int rh_embed_scatter_rows(const int64_t* fdesc, const int64_t* idesc, int idx_is_i64, int B, int F,
                          int D, const float* rows, float scale, int samples_per_block,
                          int32_t* err_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FM on an arbitrary (B, F, D) tensor (row stride x_stride floats per sample, fields contiguous)
 * replaces: FM.forward torch_rechub/basic/layers.py:313-319 and its autograd
 * reduce_sum = 1 -> out (B,) ; 0 -> out (B, D)
 */
int rh_fm_fwd(const float* x, int64_t x_stride, int B, int F, int D, int reduce_sum, float* out,
              void* stream);
int rh_fm_bwd(const float* x, int64_t x_stride, int B, int F, int D, int reduce_sum,
              const float* g_out, float* g_x, int64_t gx_stride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sequence features: gather + masked pooling
 * replaces: InputMask.forward basic/layers.py:148-161 + Sum/Average/ConcatPooling :204-251
 *           as called from EmbeddingLayer.forward basic/layers.py:86-99
 * idx (B, L) with sample stride idx_stride_b and position stride idx_stride_l (elements)
 * mode 0 = sum, 1 = mean (divide by count + 1e-16), 2 = concat (no mask, out (B, L, D))
 * mask_sentinel = padding_idx if set else -1 (positions equal to it are excluded from sum/mean)
 * out row b at out + b*out_stride (D floats, or L*D floats for concat).
 */
int rh_seq_pool_fwd(const float* table, int64_t vocab, const void* idx, int idx_is_i64,
                    int64_t idx_stride_b, int64_t idx_stride_l, int B, int L, int D, int mode,
                    int64_t mask_sentinel, float* out, int64_t out_stride, int32_t* err_flag,
                    void* stream);
/* backward: scatter-add into grad_table; rows equal to padding_idx (or -1 = none) get no gradient */
int rh_seq_pool_bwd(float* grad_table, int64_t vocab, const void* idx, int idx_is_i64,
                    int64_t idx_stride_b, int64_t idx_stride_l, int B, int L, int D, int mode,
                    int64_t mask_sentinel, int64_t padding_idx, const float* g_out,
                    int64_t g_stride, float scale, int32_t* err_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5  CrossNetwork  x_{l+1} = x0 * (w_l . x_l) + b_l + x_l
 * replaces: CrossNetwork.forward torch_rechub/basic/layers.py:412-420 and its autograd
 * x0, x : (B, d) with row strides; w, b : (L, d) contiguous; L <= 4 per call (chain calls for more,
 * passing the original x0).  One wavefront per sample, d <= 2048.
 * bwd: g_x0 / g_x may alias-sum: if sum_into_gx != 0 the x0-gradient is added into g_x and g_x0 is
 * ignored (the x0 == x case of the first segment).
 * wb_partials: (nblocks, 2, L, d) per-block partial sums of (dW, dB); nblocks is returned by
 * rh_cross_bwd_nblocks(B); the caller reduces over dim 0.
 */
int rh_cross_fwd(const float* x0, int64_t x0_stride, const float* x, int64_t x_stride,
                 const float* w, const float* b, int B, int d, int L, float* out,
                 int64_t out_stride, void* stream);
int rh_cross_bwd_nblocks(int B);
int rh_cross_max_layers(int d); /* layers one call can take for width d (4 for d <= 512) */
int rh_cross_bwd(const float* x0, int64_t x0_stride, const float* x, int64_t x_stride,
                 const float* w, const float* b, int B, int d, int L, const float* g_out,
                 int64_t g_stride, float* g_x0, float* g_x, int64_t gx_stride, int sum_into_gx,
                 float* wb_partials, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DCN-v2 cross layers: the streaming epilogues around the library GEMMs (all tensors contiguous fp32)
 * rh_cross_v2_epilogue_* replaces: `x0 * self.w[i](x) + self.b[i] + x` of CrossNetV2.forward basic/layers.py:442-443
 *   fwd: out = x0*y + b + x   (y = W_l x from the GEMM);  bwd: g_x0 = g*y, g_y = g*x0  (g_x = g, g_b = colsum(g))
 * rh_cross_mix_epilogue_* replaces: the expert loop tail of CrossNetMix.forward basic/layers.py:491-503
 *   fwd: out = sum_e gate[b,e] * x0 * (uv[e,b,:] + bias) + xl     uv (E,B,d) = U_e v_e from the batched GEMM,
 *        gate (B,E) = softmax of the gating scores;
 *   bwd: g_x0 = g * sum_e gate_e (uv_e + bias), g_uv[e] = g * gate_e * x0, g_gate[b,e] = sum_d g*x0*(uv_e + bias)
 *        (g_xl = g, g_bias = colsum(g * x0 * sum_e gate_e) are the caller's).  d <= 2048, E <= 16.
 */
int rh_cross_v2_epilogue_fwd(const float* x0, const float* y, const float* b, const float* x, int B, int d, float* out,
                             void* stream);
int rh_cross_v2_epilogue_bwd(const float* x0, const float* y, const float* g, int B, int d, float* g_x0, float* g_y,
                             void* stream);
int rh_cross_mix_epilogue_fwd(const float* x0, const float* xl, const float* uv, const float* gate, const float* bias,
                              int B, int d, int E, float* out, void* stream);
int rh_cross_mix_epilogue_bwd(const float* x0, const float* uv, const float* gate, const float* bias, const float* g,
                              int B, int d, int E, float* g_x0, float* g_uv, float* g_gate, void* stream);
/* the same, also emitting the bias gradient as rh_cross_mix_nblocks(B) x d per-block partial rows (summed by the caller) */
int rh_cross_mix_nblocks(int B);
int rh_cross_mix_epilogue_bwd_b(const float* x0, const float* uv, const float* gate, const float* bias, const float* g, int B,
                                int d, int E, float* g_x0, float* g_uv, float* g_gate, float* g_bias_partial, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CrossNetMix (DCN-v2 mixture of low-rank experts) as two dense products per layer (csrc/moe.hip)
 * replaces: the layer x expert loops of CrossNetMix.forward torch_rechub/basic/layers.py:470-506 and their autograd.
 *   KP = rh_cross_moe_kp(E, r) = E r + E rounded up to a multiple of 4.  Per layer l:
 *     PG (B, KP) = x_l VgT_l^T;  mid pass: v1 = tanh(PG[:, :E r]), v2_e = tanh(C_e v1_e), gate = softmax(PG[:, E r:E r+E]),
 *     wp (B, KP) = [gate_e v2_e | sum_e gate_e | 0];  Y (B, d) = wp UTb_l^T;  x_{l+1} = x0 * Y + x_l.
 *   rh_cross_moe_pack: VgT (L, KP, d) rows e r + k = V_l[e][:, k], rows E r + e = gating_e.weight;
 *                      UTb (L, d, KP) columns e r + k = U_l[e][:, k], column E r = bias_l.  U / V / bias / Wg: HOST arrays
 *                      of device pointers (L, L, L, E entries).
 *   rh_cross_moe_mid_fwd / _bwd: the pass between the two products; the backward takes g_wp (B, KP), writes g_PG (B, KP)
 *                      and rh_cross_moe_mid_blocks(B, E, r) partial rows (E r r) of g_C.
 *   rh_cross_moe_res_bwd: g_Y = g * x0, acc = ((mode & 1) ? 0 : acc) + g * Y + ((mode & 2) ? g : 0)  (acc: running gradient of
 *                      x0; g, x0 with row strides; bit 0 = first layer of the backward, bit 1 = last: acc also takes the residual
 *                      path's gradient, the caller's closing product accumulates into it).
 *   rh_cross_moe_unpack: sums the weight-gradient slabs of the two products (rh_linear_wgrad_partial: slabV_l = S1_l slabs
 *                      (KP, d) of g_PG^T x_l, slabU_l = S2_l slabs (d, KP) of g_Y^T wp) and the g_C partials into g_U, g_V
 *                      (E, d, r), g_bias (d,), g_C (E, r, r) per layer and g_Wg (d,) per expert (summed over the layers).
 *   Supported: L <= 8, E <= 16, r in {4, 8, 16, 32, 64}, E r <= 256 (rh_cross_moe_supported); anything else runs the
 *   batched-GEMM formulation with rh_cross_mix_epilogue_*.
 */
int rh_cross_moe_kp(int E, int r);
int rh_cross_moe_supported(int L, int E, int d, int r);
int rh_cross_moe_mid_blocks(int B, int E, int r);
int rh_cross_moe_pack(const float* const* U, const float* const* V, const float* const* bias, const float* const* Wg, int L,
                      int E, int d, int r, float* VgT, float* UTb, void* stream);
int rh_cross_moe_mid_fwd(const float* PG, const float* C, int B, int E, int r, float* v1, float* v2, float* gate, float* wp,
                         void* stream);
int rh_cross_moe_mid_bwd(const float* g_wp, const float* v1, const float* v2, const float* gate, const float* C, int B, int E,
                         int r, float* g_PG, float* gC_partial, void* stream);
int rh_cross_moe_res_bwd(const float* g, int64_t ldg, const float* x0, int64_t ldx0, const float* Y, int B, int d, int mode,
                         float* g_Y, float* acc, void* stream);
int rh_cross_moe_unpack(const float* const* slabV, const int* S1, const float* const* slabU, const int* S2,
                        const float* const* gC, const int* NB, int L, int E, int d, int r, float* const* g_U,
                        float* const* g_V, float* const* g_bias, float* const* g_C, float* const* g_Wg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DIN: Dice activation and the memory-bound ends of the ActivationUnit
 * rh_dice_fwd/bwd replaces: Dice.forward torch_rechub/basic/activation.py:15-25 (row-wise statistics over the neurons)
 *   x (N, C) contiguous, alpha (1,), eps; bwd writes gx (N, C) and per-block partial sums of d/d alpha into
 *   alpha_partial (rh_dice_nblocks(N),) which the caller sums.
 * rh_din_att_input_fwd/bwd replaces: cat[t, h, t-h, t*h] of ActivationUnit.forward, models/ranking/din.py:80-81
 *   hist (B, L, D) with batch stride hist_stride (positions contiguous), tgt (B, D) with batch stride tgt_stride,
 *   out (B*L, 4D);  bwd: g (B*L, 4D) -> g_hist (B, L, D) contiguous, g_tgt (B, D)
 * rh_din_pool_fwd/bwd replaces: (att_weight.unsqueeze(-1) * history).sum(dim=1), models/ranking/din.py:92
 *   w (B, L) contiguous -> out (B, D);  bwd: g (B, D) -> g_hist (B, L, D), g_w (B, L)
 */
int rh_dice_nblocks(int64_t N);
int rh_dice_fwd(const float* x, const float* alpha, float eps, int64_t N, int C, const float* bn_scale,
                const float* bn_shift, float* out, void* stream);
int rh_dice_bwd(const float* x, const float* g, const float* alpha, float eps, int64_t N, int C, float* gx,
                float* alpha_partial, void* stream);
/* BatchNorm1d -> Dice of the ActivationUnit's MLP (basic/layers.py:281-287 with activation "dice"), the normalisation
 * folded into the Dice passes so that the normalised tensor is never materialised:
 *   forward : rh_bn_stats_fwd (batch statistics -> stat (6, C): mean, rstd, -, -, scale, shift; running stats,
 *             num_batches_tracked; eval: from the running statistics) then rh_dice_fwd(h, ..., stat + 4C, stat + 5C, out)
 *   backward: rh_bn_dice_bwd_stats (g = dL/d out -> per-block (sum g_x, sum g_x * xhat) rows, col_partial
 *             (rh_bn_dice_stats_blocks(N), 2, C), and alpha partial sums (same number of blocks)),
 *             rh_bn_finalize_bwd (-> stat rows 2, 3, dgamma, dbeta), rh_bn_dice_bwd_apply (-> dh).   C <= 512. */
int rh_bn_stats_fwd(const float* h, int B, int C, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int training,
                    float* partial, float* stat, void* stream);
int rh_bn_finalize_bwd(float* partial, int rows, int C, float* stat, float* dgamma, float* dbeta, void* stream);
/* ... and, in the same launch, the other per-block partials of the statistics pass: extra (rows, C) -> extra_out (C) (the
 * attention head's weight gradient; null = none) and nscal <= 8 scalar rows scal (nscal, rows) -> scal_out (nscal) (Dice's alpha,
 * the head's bias; 0 = none).  Replaces torch_rechub/basic/activation.py:15-25's autograd sums over the (B L)-row tensors.
 * Fixed summation order. */
int rh_bn_finalize_bwd_tail(float* partial, int rows, int C, float* stat, float* dgamma, float* dbeta, const float* extra,
                            float* extra_out, const float* scal, int nscal, float* scal_out, void* stream);
int rh_bn_dice_stats_blocks(int64_t N);
int rh_bn_dice_bwd_stats(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                         const float* stat, const float* gamma, float* col_partial, float* alpha_partial, void* stream);
int rh_bn_dice_bwd_apply(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                         const float* stat, const float* gamma, float* dh, void* stream);
/* BatchNorm1d -> Dice -> [Dropout(0)] -> Linear(C, 1): the tail of the ActivationUnit's MLP (basic/layers.py:281-288, the
 * output layer of MLP(4 * emb_dim, dims=...), models/ranking/din.py:74).  The Dice output has ONE consumer, a C -> 1 dot
 * product, so it is never written: forward out (N,) = Dice(bn(h)) . head_w + head_b; backward g (N,) = dL/d out, the
 * gradient of the Dice output is g[r] * head_w[c] in registers.  rh_bn_dice_head_bwd_stats writes, besides col_partial
 * (blocks, 2, C), alpha_partial (2, blocks) = [d alpha partials | dL/d head_b partials] and head_partial (blocks, C) =
 * partial dL/d head_w (Dice output recomputed); blocks = rh_bn_dice_stats_blocks(N); the caller sums over blocks. */
int rh_bn_dice_head_fwd(const float* h, const float* alpha, float eps, int64_t N, int C, const float* bn_scale,
                        const float* bn_shift, const float* head_w, const float* head_b, float* out, void* stream);
int rh_bn_dice_head_bwd_stats(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                              const float* stat, const float* gamma, const float* head_w, float* col_partial,
                              float* alpha_partial, float* head_partial, void* stream);
int rh_bn_dice_head_bwd_apply(const float* h, const float* g, const float* alpha, float eps, int64_t N, int C,
                              const float* stat, const float* gamma, const float* head_w, float* dh, void* stream);
int rh_din_att_input_fwd(const float* hist, int64_t hist_stride, const float* tgt, int64_t tgt_stride, int B, int L,
                         int D, float* out, void* stream);
int rh_din_att_input_bwd(const float* hist, int64_t hist_stride, const float* tgt, int64_t tgt_stride, const float* g,
                         int B, int L, int D, float* g_hist, float* g_tgt, void* stream);
int rh_din_pool_fwd(const float* hist, int64_t hist_stride, const float* w, int B, int L, int D, float* out,
                    void* stream);
int rh_din_pool_bwd(const float* hist, int64_t hist_stride, const float* w, const float* g, int B, int L, int D,
                    float* g_hist, float* g_w, void* stream);
/* First layer of the ActivationUnit's MLP with the operand built in registers (csrc/dinmlp.hip); replaces
 * models/ranking/din.py:81-85 (expand + cat + view) and the Linear(4 D, N) of basic/layers.py:279 on it:
 *   z (B L, N) = [t, h, t - h, t * h] W^T + bias;  the (B L, 4 D) operand never exists in memory.
 *   partial (optional): ceil(B L / rh_din_att_l1_chunk_rows(B L)) x 2 x N per-chunk (sum, M2 about the chunk mean) of z --
 *   the BatchNorm statistics, combined by rh_bn_stats_from_partial (no pass over z).
 *   Supported (rh_din_att_l1_supported): D in {4, 8, 16}, N a multiple of 64 up to 256. */
/* nn.PReLU() with one slope (activation_layer("prelu"), torch_rechub/basic/activation.py:40-41; the DSSM towers): x contiguous,
 * n elements; bwd writes gx and rh_prelu_nblocks(n) per-block partial sums of d / d slope. */
int rh_prelu_nblocks(int64_t n);
int rh_prelu_fwd(const float* x, const float* slope, int64_t n, float* out, void* stream);
int rh_prelu_bwd(const float* x, const float* g, const float* slope, int64_t n, float* gx, float* partial, void* stream);
/* In-batch negatives without the (B, C) score matrix (csrc/match.hip); replaces, for RANDOM negatives,
 * `scores = user @ item.T` + `gather_inbatch_logits(scores, neg_indices)` of trainers/match_trainer.py:118-138 /
 * utils/match.py:148-153 and their autograd:  logits[i, 0] = u_i . v_(row0 + i),  logits[i, 1 + k] = u_i . v_neg[i, k];
 * bwd: g_u (B, D) written, g_v (C, D) accumulated with float atomics into a buffer the caller zeroed.  D <= 1024. */
int rh_inbatch_logits_fwd(const float* u, int64_t ldu, const float* v, int64_t ldv, const int64_t* neg, int B, int C, int D,
                          int K, int row0, float* logits, int32_t* err_flag, void* stream);
int rh_inbatch_logits_bwd(const float* u, int64_t ldu, const float* v, int64_t ldv, const int64_t* neg, const float* g, int B,
                          int C, int D, int K, int row0, float* g_u, float* g_v, void* stream);

/* Two-tower glue as single launches.
 * rh_l2norm_fwd/bwd: F.normalize(x, p=2, dim=1) of the towers' outputs (torch_rechub/models/matching/dssm.py:56,66):
 *   y = x / max(||x||_2, eps) row-wise, nrm (B,) = ||x|| saved; gx = (g - y (g . y)) / ||x||, or g / eps where the norm was
 *   clamped.  x / g row stride ldx / ldg (multiples of 4), d % 4 == 0, d <= 4096; y, gx contiguous (B, d).
 * rh_ce_fwd/bwd: torch.nn.CrossEntropyLoss() (mean) over the (B, C) in-batch logits (trainers/match_trainer.py:60,136;
 *   target int64 (B,), null = class 0 for every row as the trainer's zero targets): lse (B,) = logsumexp per row, loss_partial
 *   (rh_ce_nblocks(B),) per-block sums of lse - x[target] (the caller sums them and divides by B: rh_colsum);
 *   g_logits = g_loss[0] / B * (softmax - onehot).  C <= 1024; an out-of-range target sets RH_FLAG_INDEX_OOB. */
int rh_l2norm_fwd(const float* x, int64_t ldx, int B, int d, float eps, float* y, float* nrm, void* stream);
int rh_l2norm_bwd(const float* y, const float* nrm, const float* g, int64_t ldg, int B, int d, float eps, float* gx, void* stream);
int rh_ce_nblocks(int B);
int rh_ce_fwd(const float* logits, const int64_t* target, int B, int C, float* lse, float* loss_partial, int32_t* err_flag,
              void* stream);
int rh_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* g_loss, int B, int C, float* g_logits,
              void* stream);
int rh_din_att_l1_supported(int D, int N);
int rh_din_att_l1_chunk_rows(int64_t rows);
int rh_din_att_l1_fwd(const float* hist, int64_t hist_stride, const float* tgt, int64_t tgt_stride, const float* W,
                      const float* bias, int B, int L, int D, int N, float* z, float* partial, void* stream);
/* rh_bn_stats_fwd (training) from per-chunk partials that a producer already emitted: finalize only. */
int rh_bn_stats_from_partial(const float* partial, int rows_per_chunk, int B, int C, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                             float eps, float* stat, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MLP Linear weight gradient, the (., 1) output head and the BCE loss (csrc/linear.hip)
 * rh_linear_wgrad replaces: the weight/bias gradient of every nn.Linear of MLP, torch_rechub/basic/layers.py:279,290
 *   (autograd's mm(g^T, x) + sum(g, 0)):  dW (N, K) = g^T x,  db (N,) = column sums of g (db may be NULL)
 *   g (B, N) row stride ldg, x (B, K) row stride ldx; batch split over blocks, f32 MFMA, partial tiles summed in split
 *   order by a second small launch (deterministic).  partial: rh_linear_wgrad_workspace(B, N, K) floats.
 * rh_head_fwd/bwd replaces: MLP's output Linear(K, 1) + `y_linear + y_fm + y_deep` + torch.sigmoid(y.squeeze(1)),
 *   models/ranking/deepfm.py:39-43, widedeep.py:35-39:  y (B,) = sigmoid(h w^T + bias + e0 + e1)  (bias, e0, e1 optional)
 *   bwd: g_z = g_y y (1 - y) (also the gradient of e0 / e1), g_h (B, K) = g_z w, g_w (K,) = g_z^T h, g_b = sum g_z.
 *   1 <= K <= 1024 (rows of h may be only 4-byte aligned).  partial: rh_head_nblocks(B) * (K + 1) floats.
 * rh_bce_fwd/bwd replaces: torch.nn.BCELoss() (mean) of trainers/ctr_trainer.py:62,:93-95: log terms clamped at -100;
 *   loss (1,), g_loss (1,) device scalars.
 */
/* rh_linear_fwd / rh_linear_dgrad replace: the forward y = x W^T + b and the input gradient g_x = g W of the same
 *   nn.Linear modules (aten::addmm / mm) at CTR batch sizes, where one 64x64 f32-MFMA tile per workgroup fills the chip
 *   exactly once (csrc/gemm.hip).  x (M, K) row stride ldx, w (N, K) row stride ldw, bias (N,) or NULL, y (M, N).
 *   stats (NULL or (ceil(M / R), 2, N) floats, R = rh_gemm_stats_rows(M, N) = 64 or 32): per R-row slab of y and
 *   column, the slab sum and M2 = sum (y - slab mean)^2 -- the input rh_bn_relu_dropout_fwd takes with partial_rows = R.
 *   bn_rng / bn_saved_ctr / bn_batches (optional, with stats): the consuming rh_bn_relu_dropout_fwd call's dropout
 *   state (rng[1] = call counter) and num_batches_tracked; the GEMM then performs that call's bookkeeping
 *   (saved_ctr = rng[1]++, batches += 1) and rh_bn_relu_dropout_fwd (partial_rows > 0) then advances nothing.
 *   rh_linear_dgrad: g (M, N), w (N, K) -> gx (M, K).  Exact f32 (MFMA f32 == fmaf chain); summation order over k is
 *   permuted inside each 32-wide K tile. */
int rh_gemm_stats_rows(int M, int N);
int rh_gemm_chain_stats_rows(int M); /* slab height of the `stats` rh_linear_bnact_fwd writes (64, or 32 when M <= 32) */
int rh_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N, int K,
                  float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr, int64_t* bn_batches,
                  void* stream);
/* rh_linear_fwd that also counts a CHAIN START in the deferred sweep's gate words (chain_gate[2], RH_GATE_WORDS below) when its
 * last workgroup starts, i.e. when every workgroup of the launch has been placed: captured as the first own GEMM of a step's
 * hipGraph it releases the optimizer's sweep of the step before (rh_adam_sweep_gate) beside a chain that is already running. */
int rh_linear_fwd_gate(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N, int K,
                       float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr, int64_t* bn_batches,
                       int64_t* chain_gate, void* stream);
int rh_linear_dgrad(const float* g, int64_t ldg, const float* w, int64_t ldw, int M, int N, int K, float* gx,
                    int64_t ldgx, void* stream);
/* One CrossNetV2 layer (torch_rechub/basic/layers.py:440-444: x <- x0 * (W_l x) + b_l + x) on the tile GEMM with the Hadamard +
 * bias + residual as its epilogue (round 5; rounds 1-4: library GEMM + rh_cross_v2_epilogue_fwd as a second pass).
 * rh_cross_v2_fwd: x0, x (M, d), w (d, d) row-major as nn.Linear.weight, b (d,) -> y (M, d) = x w^T (kept for the backward) and
 * out (M, d) = x0 * y + b + x.  rh_cross_v2_dgrad: gx (M, d) = g_y w + g with g_y = g * x0 (rh_cross_v2_epilogue_bwd forms g_y
 * and g_x0 = g * y): the layer's gradient with respect to x, residual path included.  Every operand contiguous (row stride d);
 * 1 <= M <= 16384, 1 <= d <= 1024, else RH_E_UNSUPPORTED. */
int rh_cross_v2_fwd(const float* x0, const float* x, const float* w, const float* b, int M, int d, float* y, float* out,
                    void* stream);
int rh_cross_v2_dgrad(const float* g_y, const float* w, const float* g, int M, int d, float* gx, void* stream);
/* The fused MLP chain (round 4): the reference's hidden-layer tail  BatchNorm1d -> ReLU -> Dropout
 * (torch_rechub/basic/layers.py:283-286) is never run as a pass of its own -- it is applied where its output is consumed.
 * rh_linear_bnact_fwd: layer l + 1 as ONE launch, y = dropout(relu(batch_norm(h))) W^T + b with h (M, K) the PRE-BatchNorm
 *   output of layer l.  pro_stats (ceil(M / pro_rows), 2, K): the per-slab (sum, M2) the producing GEMM wrote as `stats`
 *   (pro_rows = rh_gemm_stats_rows(M, K)); gamma / beta / running_* / momentum / eps: layer l's nn.BatchNorm1d; p_drop, rng,
 *   ctr: its nn.Dropout and the counter the producing GEMM drew for it (that call's bn_saved_ctr).  Written besides y:
 *   stat_out (>= 2, K) = mean, rstd of layer l; its running statistics; act_out (M, K) or NULL = the activations (this
 *   layer's weight gradient reads them); stats / bn_rng / bn_saved_ctr / bn_batches: as rh_linear_fwd, for y and the NEXT
 *   BatchNorm.  K % 4 == 0, K <= 1024, ldh % 4 == 0.  Same per-element arithmetic as rh_bn_relu_dropout_fwd.
 * rh_linear_dgrad_bnbwd: rh_linear_dgrad whose epilogue also forms the BatchNorm-backward column sums of the hidden layer
 *   that produced this Linear's input: gx (M, K) = gradient of a = dropout(relu(batch_norm(h))), h (M, K) that layer's
 *   pre-BatchNorm activations, stat (>= 2, K) its mean / rstd, ctr its dropout counter.  bwd_partial (ceil(M / R), 2, K),
 *   R = rh_gemm_stats_rows(M, K): per R-row slab (sum g1, sum g1 * xhat) -- rh_bn_relu_dropout_bwd_pre's `partial` with
 *   nchunks_pre = ceil(M / R) (replaces the statistics launch of the BatchNorm backward).
 * rh_head_bnact_fwd: the output head of such a chain, y = sigmoid(dropout(relu(batch_norm(z))) . w + b + e0 + e1), z (B, K)
 *   pre-BatchNorm, stats / stats_rows / ctr as above; t / loss_partial as rh_head_loss_fwd (both NULL: no loss terms).
 *   K % 4 == 0, K <= 256.  Its backward is rh_head_bwd_bn with h = NULL (the head's input is recomputed from z). */
int rh_linear_bnact_fwd(const float* h, int64_t ldh, int M, int K, const float* pro_stats, int pro_rows, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps, float p_drop,
                        const int64_t* rng, const int64_t* ctr, float* stat_out, float* act_out, const float* w, int64_t ldw,
                        const float* bias, int N, float* y, int64_t ldy, float* stats, int64_t* bn_rng, int64_t* bn_saved_ctr,
                        int64_t* bn_batches, void* stream);
int rh_linear_dgrad_bnbwd(const float* g, int64_t ldg, const float* w, int64_t ldw, int M, int N, int K, float* gx,
                          int64_t ldgx, const float* h, int64_t ldh, const float* stat, const float* gamma, const float* beta,
                          float p_drop, const int64_t* rng, const int64_t* ctr, float* bwd_partial, void* stream);
int rh_head_bnact_fwd(const float* z, int64_t ldz, const float* stats, int stats_rows, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float p_drop, const int64_t* rng,
                      const int64_t* ctr, float* stat_out, const float* w, const float* bias, const float* e0, const float* e1,
                      int B, int K, float* y, const float* t, float* loss_partial, void* stream);
int64_t rh_linear_wgrad_workspace(int B, int N, int K);
int rh_linear_wgrad_tiles(int N, int K);
int rh_linear_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx, int B, int N, int K, float* dW, float* db,
                    float* partial, void* stream);
/* The same without the reduction launch: partial = S slabs (N, K) followed by S slabs (N,), S = rh_linear_wgrad_splits;
 * the caller sums them (rh_pack_grads folds that into the packing of the step's dense gradients). */
int rh_linear_wgrad_splits(int B, int N, int K);
int rh_linear_wgrad_partial(const float* g, int64_t ldg, const float* x, int64_t ldx, int B, int N, int K, float* partial,
                            void* stream);
/* n <= 8 independent rh_linear_wgrad_partial problems as ONE launch (the two weight gradients per layer of CrossNetMix's
 * backward, torch_rechub/basic/layers.py:470-506: nothing else in that backward waits for them).  Host arrays of n entries. */
int rh_linear_wgrad_partial_group(int n, const float* const* g, const int64_t* ldg, const float* const* x, const int64_t* ldx,
                                  const int* B, const int* N, const int* K, float* const* partial, void* stream);
int rh_head_nblocks(int B);
int rh_head_fwd(const float* h, int64_t ldh, const float* w, const float* bias, const float* e0, const float* e1, int B,
                int K, float* y, void* stream);
int rh_head_bwd(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, int B, int K, float* g_h,
                float* g_z, float* g_w, float* g_b, float* partial, void* stream);
/* Column sums of the per-block partial buffers the backward kernels emit (replaces the trailing `.sum(0)` / `.sum()`
 * launches of autograd): out (cols,) = sum over rows of a (rows, cols); optionally vsum (1,) = sum of v (n,). */
int rh_colsum(const float* a, int rows, int cols, float* out, const float* v, int64_t n, float* vsum, void* stream);
/* Fused head + loss for the trainer's step (models/ranking/deepfm.py:39-43 + torch.nn.BCELoss trainers/ctr_trainer.py:62,88):
 * rh_head_loss_fwd = rh_head_fwd that also emits, per block, the sum of the BCE terms of its rows against the labels t
 *   (loss_partial: rh_head_loss_nblocks(B) floats; the mean is finished by rh_step_scalars);
 * rh_head_loss_bwd = rh_bce_bwd + rh_head_bwd in one pass: g_y = g_loss[0] / B * (y - t) / max((1 - y) y, 1e-12) is
 *   formed per row with the arithmetic of rh_bce_bwd (bit-identical to the two separate launches).
 *   rh_head_bwd_ex / rh_head_bwd_bn take g_y, (t, g_loss) or both (a prediction with a second consumer besides the
 *   loss: the inline BCE gradient and g_y add per row).
 * rh_step_scalars: the scalar work of one training step in ONE single-block launch (each part optional, null = skip):
 *   loss[0] = sum(loss_partial[0..n_partial)) / B;  the bias corrections of the next Adam step (== rh_adam_prepare);
 *   two device counters c = (c + inc) % mod (mod 0 = no wrap): the batch position of the device loader (==
 *   rh_batch_advance), the call counter of the in-batch sampler. */
int rh_head_bwd_ex(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, const float* t,
                   const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b, float* partial,
                   int reduce, void* stream);
/* rh_head_bwd_ex + the BatchNorm-backward column sums of the hidden layer below the head, when h = dropout(relu(bn(bn_z)))
 * and the head is its only consumer: bn_partial (rh_head_nblocks(B), 2, K) per-block (sum g1, sum g1 * xhat) with the mask
 * arithmetic of rh_bn_relu_dropout_bwd; rh_bn_relu_dropout_bwd_pre then needs no statistics pass.  K % 4 == 0, K <= 256. */
int rh_head_bwd_bn(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, const float* t,
                   const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b, float* partial,
                   int reduce, const float* bn_z, const float* bn_stat, const float* bn_gamma, const float* bn_beta,
                   float bn_p, const int64_t* bn_rng, const int64_t* bn_ctr, int bn_relu, float* bn_partial, void* stream);
/* rh_head_bwd_bn + rh_step_scalars as ONE launch (one extra workgroup does the scalar work; arguments as rh_step_scalars):
 * nothing between the head's forward and its backward reads the loss mean / Adam corrections / device counters. */
int rh_head_bwd_bn_scalars(const float* h, int64_t ldh, const float* w, const float* y, const float* g_y, const float* t,
                           const float* g_loss, int B, int K, float* g_h, float* g_z, float* g_w, float* g_b, float* partial,
                           int reduce, const float* bn_z, const float* bn_stat, const float* bn_gamma, const float* bn_beta,
                           float bn_p, const int64_t* bn_rng, const int64_t* bn_ctr, int bn_relu, float* bn_partial,
                           const float* loss_partial, int n_partial, float* loss, double* hyper, int64_t* step, float* ring,
                           int ring_size, int64_t* c0, int64_t inc0, int64_t mod0, int64_t* c1, int64_t inc1, int64_t mod1,
                           void* stream);
int rh_head_loss_nblocks(int B);
int rh_head_loss_fwd(const float* h, int64_t ldh, const float* w, const float* bias, const float* e0, const float* e1, int B,
                     int K, float* y, const float* t, float* loss_partial, void* stream);
int rh_head_loss_bwd(const float* h, int64_t ldh, const float* w, const float* y, const float* t, const float* g_loss, int B,
                     int K, float* g_h, float* g_z, float* g_w, float* g_b, float* partial, void* stream);
int rh_step_scalars(const float* loss_partial, int n_partial, int64_t B, float* loss, double* hyper, int64_t* step,
                    float* ring, int ring_size, int64_t* c0, int64_t inc0, int64_t mod0, int64_t* c1, int64_t inc1,
                    int64_t mod1, void* stream);
int rh_bce_fwd(const float* y, const float* t, int64_t B, float* loss, void* stream);
int rh_bce_bwd(const float* y, const float* t, const float* g_loss, int64_t B, float* g_y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MLP hidden-layer epilogue: BatchNorm1d + ReLU + Dropout fused (the Linear in front stays a library GEMM)
 * replaces: nn.BatchNorm1d -> ReLU -> nn.Dropout of MLP, torch_rechub/basic/layers.py:281-287, and their autograd
 * relu = 1: the MLP hidden layer BatchNorm1d -> ReLU -> Dropout; relu = 0: BatchNorm1d (-> Dropout) only, for the layers
 * whose activation is Dice / PReLU / ... (MLP of DIN's ActivationUnit, the DSSM towers).
 * h (B,C) pre-BN activations; training: batch statistics (biased variance), running stats updated with `momentum`
 * (unbiased variance), num_batches_tracked += 1; eval: running statistics, no dropout.
 * rng (device int64 [4]): seed, call counter (bumped by the forward), block ticket (zero on entry / exit), spare;
 * saved_ctr (device int64 [1]): the counter this call used — the backward recomputes the same dropout mask from it
 * (nothing is stored).  B <= 8192: two launches per direction (partial sums; finalize folded into apply), else three.
 * partial: (rh_bn_act_nchunks(B), 2, C) floats; stat: (4, C) floats (mean, rstd kept for the backward).
 * partial_rows > 0 (forward): `partial` ALREADY holds, per partial_rows-row slab and column, the slab's sum and its
 * M2 = sum (h - slab mean)^2 -- rh_linear_fwd writes exactly that (64- or 32-row slabs) from its accumulators, so the
 * statistics pass over h disappears; the slabs are combined with Chan's parallel-variance formula in slab order.
 * rng / num_batches_tracked were then already advanced by rh_linear_fwd (it was given the same pointers): this call
 * reads saved_ctr and advances nothing.
 */
int rh_bn_act_nchunks(int B);
int rh_bn_relu_dropout_fwd(const float* h, int B, int C, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float p_drop,
                           int training, int64_t* rng, int64_t* saved_ctr, float* partial, int partial_rows, float* stat,
                           float* out, int relu, void* stream);
/* the same with the column sums already formed by the producer of dy (rh_head_bwd_bn): nchunks_pre <= 128 partial rows */
int rh_bn_relu_dropout_bwd_pre(const float* h, const float* dy, int B, int C, const float* gamma, const float* beta,
                               float p_drop, const int64_t* rng, const int64_t* saved_ctr, const float* partial,
                               int nchunks_pre, float* stat, float* dx, float* dgamma, float* dbeta, int relu, void* stream);
int rh_bn_relu_dropout_bwd(const float* h, const float* dy, int B, int C, const float* gamma, const float* beta,
                           float p_drop, const int64_t* rng, const int64_t* saved_ctr, float* partial, float* stat,
                           float* dx, float* dgamma, float* dbeta, int relu, void* stream);
/* The same two passes with nn.PReLU() (ONE slope) in place of the ReLU: Linear -> BatchNorm1d -> PReLU -> Dropout of the
 * two-tower MLPs (reference MLP(activation="prelu"), examples/matching/run_ml_dssm.py:69-80; basic/layers.py:276-292).
 * y = bn > 0 ? bn : slope[0] * bn.  The backward also writes slope_partial (rh_bn_prelu_nblocks(B, C) floats): per-workgroup
 * sums of dy * min(bn, 0), whose total is the slope's gradient.  Training mode only (eval: BatchNorm pass + rh_prelu_fwd). */
int rh_bn_prelu_dropout_fwd(const float* h, int B, int C, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float p_drop,
                            int training, int64_t* rng, int64_t* saved_ctr, float* partial, int partial_rows, float* stat,
                            float* out, const float* slope, void* stream);
int rh_bn_prelu_nblocks(int B, int C);
int rh_bn_prelu_dropout_bwd(const float* h, const float* dy, int B, int C, const float* gamma, const float* beta, float p_drop,
                            const int64_t* rng, const int64_t* saved_ctr, float* partial, float* stat, float* dx,
                            float* dgamma, float* dbeta, const float* slope, float* slope_partial, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense Adam with coupled L2, torch.optim.Adam semantics over every row of every table
 * replaces: optimizer.step() torch_rechub/trainers/ctr_trainer.py:59-61,99 for embedding tables
 * tdesc (device int64 [5*T]): p, g, m, v pointers and numel per tensor (T <= 128)
 * h_numel (HOST int64 [T]): the same numel values (multiples of 4), used to size the launch
 * hyper (device double [16]): inputs  [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=weight_decay
 *                             derived [8]=step_size=lr/(1-beta1^t) [9]=sqrt(1-beta2^t)
 *                                     [10]=1-beta1 [11]=1-beta2 [12]=t
 * rh_adam_prepare increments *step (device int64) and fills hyper[8..12] on the device in fp64
 * (as torch does on the host), so prepare + dense can be replayed from a hipGraph.
 * per element (fp32): g' = g + wd*p; m += (1-b1)*(g'-m); v = v*b2 + (1-b2)*g'^2;
 *                     p -= step_size * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * zero_grad != 0: g <- 0 where it was non-zero (replaces model.zero_grad(), ctr_trainer.py:97)
 */
int rh_adam_prepare(double* hyper, int64_t* step, float* ring, int ring_size, void* stream);
int rh_adam_dense(const int64_t* tdesc, int T, const int64_t* h_numel, const double* hyper,
                  int zero_grad, void* stream);

/* Adam for the small dense parameters (MLP / LR / cross weights) in one launch; gradients are read from the packed
 * flat bucket (the buffer the RCCL all-reduce runs on).  Any numel, 4-byte alignment.  Uses hyper[] of rh_adam_prepare.
 * sdesc (device int64 [5*T]): p, m, v pointers, numel, offset of the parameter's gradient inside flat_g (T <= 128).
 * replaces: optimizer.step() for the non-embedding parameters, trainers/ctr_trainer.py:99 */
int rh_adam_small(const int64_t* sdesc, int T, const int64_t* h_numel, const float* flat_g, const double* hyper,
                  void* stream);

/* Pack the dense gradients of one step into the flat bucket (replaces: torch.cat over the parameter gradients + the
 * trailing partial-sum launch of every backward kernel; autograd's `.sum(0)` / accumulate kernels in the reference,
 * trainers/ctr_trainer.py:97-98):  flat[dst_offset + i] = sum_{r < nparts} src[r * stride + i] (+ add[i]),  i < numel,
 * summed in the fixed order r = 0, 1, ... (deterministic).  nparts == 0 writes zeros.  `items` is a HOST array; its
 * entries travel by value in the kernel arguments (hipGraph-capturable although the slabs are temporaries). */
typedef struct RhPackItem {
  uint64_t src;       /* device pointer to the partial rows (float) */
  uint64_t add;       /* device pointer to a plain (numel,) term added on top, or 0 */
  int64_t nparts;     /* rows to sum */
  int64_t stride;     /* floats between consecutive rows */
  int64_t numel;      /* elements of the parameter */
  int64_t dst_offset; /* offset of the parameter inside flat */
} RhPackItem;
int rh_pack_grads(const RhPackItem* items, int n, float* flat, void* stream);
/* the same + the Adam step of every packed parameter on the element just summed (= rh_pack_grads then rh_adam_small over the
 * same n parameters, sdesc in rh_adam_small's layout, hyper already holding this step's scalars): one launch */
int rh_pack_grads_adam(const RhPackItem* items, int n, float* flat, const int64_t* sdesc, const double* hyper, void* stream);
/* rh_pack_grads_adam whose launch also counts an opening of `gate` (as rh_adam_sweep_gate_open) when it STARTS: captured
 * behind the end-of-step table launch it announces that launch's end without a launch of its own.  gate may be NULL. */
int rh_pack_grads_adam_gate(const RhPackItem* items, int n, float* flat, const int64_t* sdesc, const double* hyper,
                            int64_t* gate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Blocked-lazy EXACT Adam (same arithmetic as rh_adam_dense, bit-identical results, ~1/K of its HBM traffic)
 * replaces: the same optimizer.step() (trainers/ctr_trainer.py:99).  Adam is element-wise and a row that is not in
 * the batch has the known gradient wd*p, so rows are brought up to date lazily by replaying the skipped steps in
 * registers; every row is refreshed at least every K steps (sweep window) and immediately when the batch touches it.
 * ring  (device float [2*ring_size], ring_size a power of two > K): per-step (A_t, E_t) written by rh_adam_prepare
 *       (hyper[13] = A_t = step_size*sqrt(1-b2^t), hyper[14] = E_t = eps*sqrt(1-b2^t))
 * ldesc (device int64 [8*T]): p, g, m, v, last (int32 per row: step the row is up to date with) pointers,
 *       rows, K_t (1 = dense table, stepped with its gradient by the sweep), window rows w_t = ceil(rows/K_t)
 * rh_adam_lazy_touched: for every lookup of the batch (index columns in idesc, as rh_embed_bwd) claim the row,
 *       replay it to step t-1, apply step t with its gradient row, re-zero the gradient row.
 *       field_table (device int64 [2*F]): table index of field f (-1 = skip), padding_idx (-1 = none)
 *       refresh != 0: the pre-gather pass (rows about to be READ are brought to the hyper step; their gradient rows
 *       are zero by construction and are neither read nor written)
 * rh_adam_lazy_sweep: bring window (t-1) mod K_t of every table up to date.  mode: RH_SWEEP_WINDOW = every table,
 *       RH_SWEEP_FLUSH = all rows of every table, RH_SWEEP_LAZY_TABLES = windows of the K_t > 1 tables only,
 *       RH_SWEEP_DENSE_TABLES = the K_t == 1 tables only (these take their gradient in the sweep).
 * Call order per step: rh_adam_prepare, rh_adam_lazy_touched (once per index batch), rh_adam_lazy_sweep.
 * t_value < 0: the sweep belongs to the step in hyper (in line, before the next rh_adam_prepare).  t_value = s >= 0:
 * deferred sweep of step s, step number by value and (A_s, E_s) from the ring -- it only touches rows that are NOT at
 * step >= s, so it may run on another stream concurrently with the whole of step s + 1 (its rh_adam_prepare,
 * forward, backward, rh_adam_lazy_touched), PROVIDED step s + 1 brought its own rows up to step s before the sweep
 * was launched (rh_adam_lazy_touched under the hyper of step s = the pre-gather refresh) and the sweep has finished
 * before step s + 2 refreshes its rows.
 */
#define RH_SWEEP_WINDOW 0
#define RH_SWEEP_FLUSH 1
#define RH_SWEEP_LAZY_TABLES 2
#define RH_SWEEP_DENSE_TABLES 3
/* A one-lane launch that occupies `stream` for RH_TUNE_SWEEP_STAGGER_NS: placed in front of a deferred sweep whose release
 * coincides with a kernel launch of the step's chain, so that the two are not dispatched together (csrc/optim.hip). */
int rh_adam_sweep_stagger(void* stream);
/* The sweep's release by device words instead of an event.  gate: RH_GATE_WORDS int64, zero-initialised by the caller:
 * [0] count of openings, [1] wall clock of the last one, [2] count of chain starts (rh_linear_fwd_gate), [4 + 2 (i & 3)],
 * [5 + 2 (i & 3)] chain-start count and wall clock at opening i.  rh_adam_sweep_gate_open: a one-lane launch that counts an
 * opening -- captured as the last launch of a step's hipGraph it announces on every replay that the step has finished.
 * rh_adam_sweep_gate occupies `stream` until the count has reached `expected` AND a chain start has been counted since that
 * opening (the next step's first own GEMM has placed its workgroups) -- or, failing that, fallback_ns (0: RH_TUNE_SWEEP_GATE_NS)
 * have passed since the opening (round 4: that hold-back WAS the release; it still is for graphs that count no chain start).  A gate not opened within 2 s gives up and sets bit
 * RH_ERR_GATE_TIMEOUT of *err_flag (the error word of rh_embed_fwd). */
#define RH_GATE_WORDS 16
#define RH_ERR_GATE_TIMEOUT 64
int rh_adam_sweep_gate(const int64_t* gate, int64_t expected, int64_t fallback_ns, int32_t* err_flag, void* stream);
/* rh_adam_sweep_gate that first stores done_value into *done_host -- the DEVICE address (rh_host_device_pointer) of a word of
 * pinned, mapped host memory: by stream order "everything enqueued on `stream` before this launch has completed", i.e. the
 * caller's count of finished deferred sweeps, readable by the host without an event record between the sweeps (round 6). */
int rh_adam_sweep_gate_done(const int64_t* gate, int64_t expected, int64_t fallback_ns, int32_t* err_flag, int64_t* done_host,
                            int64_t done_value, void* stream);
int rh_host_device_pointer(void* host, void** device);
int rh_adam_sweep_gate_open(int64_t* gate, void* stream);
/* Releases a deferred sweep that rh_adam_sweep_gate holds back for the NEXT step's chain start when the host knows that no
 * further step follows the ones it has enqueued (end of an epoch, a synchronisation): counts one chain start in gate[2], so the
 * sweep starts at once instead of after its fallback (round 6; enqueue on the stream the step's graph was launched on). */
int rh_adam_sweep_release(int64_t* gate, void* stream);
int rh_adam_lazy_touched(const int64_t* ldesc, int T, const int64_t* field_table, const int64_t* idesc,
                         int idx_is_i64, int B, int F, int D, const double* hyper, const float* ring,
                         int ring_size, int samples_per_block, int refresh, int32_t* err_flag, void* stream);
/* n <= 4 rh_adam_lazy_touched passes over ONE table group as one launch (round 6): the refreshes in front of the first gather of a
 * step with several gathers (two-tower / sequence models: user, item, history lookups; reference: one nn.Embedding call each,
 * basic/layers.py:83-99), or the touched-rows steps at its end.  field_table / idesc: HOST arrays of n device pointers;
 * idx_is_i64, B, F: host arrays of n entries. */
int rh_adam_lazy_touched_group(const int64_t* ldesc, int T, int n, const int64_t* const* field_table, const int64_t* const* idesc,
                               const int* idx_is_i64, const int* B, const int* F, int D, const double* hyper, const float* ring,
                               int ring_size, int samples_per_block, int refresh, int32_t* err_flag, void* stream);
/* rh_adam_lazy_touched, refresh argument: 0 = the touched-rows step (rows take their gradient); 1 = pre-gather refresh of
 * every row of the batch (no gradient traffic). */
/* rh_batch_gather + rh_adam_lazy_touched (refresh = 1, int64 indices) as ONE launch (round 4): the batch is assembled into the
 * static buffers (sparse_out (B, Fd), dense_out (B, ND), label_out (B)) from dataset rows perm[(pos + b) mod N] -- exactly
 * rh_batch_gather's arguments -- while the refresh part reads the same indices straight from the dataset.  idesc must
 * describe index columns INSIDE sparse_out (pointer = sparse_out + column, stride = Fd): the gather that follows reads them.
 * lookahead != 0: further workgroups also refresh those lookups of the NEXT batch (dataset rows perm[(pos + B + b) mod N],
 * b < B) whose table rows lie in the window ((t - 1) mod K) * w .. + w of the coming deferred sweep (t = hyper[12]): that sweep
 * then skips every row the next batch reads, so the next call of this function may run while it is still in flight. */
int rh_adam_lazy_refresh_assemble(const int64_t* ldesc, int T, const int64_t* field_table, const int64_t* idesc, int B, int F,
                                  int D, const double* hyper, const float* ring, int ring_size, int samples_per_block,
                                  int32_t* err_flag, const int64_t* perm, const int64_t* pos, int64_t N, const int64_t* sparse,
                                  int Fd, const float* dense, int ND, const float* label, int64_t* sparse_out, float* dense_out,
                                  float* label_out, int lookahead, void* stream);
/* The end of step t and the head of step t + 1 as ONE launch (round 4): rh_adam_lazy_step_mode(RH_SWEEP_DENSE_TABLES) of the
 * batch just trained on + rh_adam_lazy_refresh_assemble of the NEXT batch (arguments as there; B is the size of both; *pos
 * must already point at the next batch, i.e. the step's scalar launch has advanced it -- the finished batch is read back
 * from dataset positions pos - B ..; look_depth = batches after the next one whose lookups inside the coming deferred sweep's
 * window are refreshed too, 0..4).  Every part claims a row with atomicMax on its last-step word and the claimant applies the
 * row's gradient, so a row both batches look up is stepped exactly once.  The caller's hipGraph of a step then starts with
 * the gather; optim.TableAdam ("step ahead") keeps the bookkeeping. */
int rh_adam_lazy_step_ahead(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                            const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                            const int64_t* idesc, int B, int F, int32_t* err_flag, const int64_t* perm, const int64_t* pos,
                            int64_t N, const int64_t* sparse, int Fd, const float* dense, int ND, const float* label,
                            int64_t* sparse_out, float* dense_out, float* label_out, int look_depth, void* stream);
/* rh_adam_lazy_step_ahead whose touched-rows part walks touched_B rows of ANOTHER int64 index matrix (touched_idesc: F column
 * pointers, then F strides) instead of the batch before the one it assembles: the GATHERED lookups of every rank's batch under
 * data parallelism with replicated tables (round 6; reference: nn.DataParallel applies the global batch's update on every
 * replica, trainers/ctr_trainer.py:53-55, optimizer.step() :99).  The refresh / assembly / look-ahead parts work on the LOCAL
 * batch as before; a row both index sets hold is claimed by one of the two passes, as in rh_adam_lazy_step_ahead. */
int rh_adam_lazy_step_ahead_touched(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                    const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                                    const int64_t* idesc, int B, int F, int32_t* err_flag, const int64_t* perm,
                                    const int64_t* pos, int64_t N, const int64_t* sparse, int Fd, const float* dense, int ND,
                                    const float* label, int64_t* sparse_out, float* dense_out, float* label_out, int look_depth,
                                    const int64_t* touched_idesc, int touched_B, void* stream);
/* rh_adam_lazy_step_ahead whose launch ALSO carries the weight gradients of the step's nn.Linear layers (round 6; reference:
 * the Linear backward inside loss.backward(), trainers/ctr_trainer.py:98, dW = g^T x of basic/layers.py:279,290): wn <= 8
 * problems, arrays of wn host entries as rh_linear_wgrad_partial_group (problem i writes its split slabs to wpartial[i],
 * rh_linear_wgrad_workspace floats, for rh_pack_grads).  Nothing between the last input-gradient GEMM of the backward and
 * the packing launch behind the optimizer reads those slabs, so they leave the step's critical chain: their MFMA workgroups
 * run beside the replay arithmetic of the optimizer's parts.  Same workgroup body and split plan as the grouped launch --
 * the same slabs bit for bit. */
int rh_adam_lazy_step_ahead_wgrad(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                                  const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                                  const int64_t* idesc, int B, int F, int32_t* err_flag, const int64_t* perm, const int64_t* pos,
                                  int64_t N, const int64_t* sparse, int Fd, const float* dense, int ND, const float* label,
                                  int64_t* sparse_out, float* dense_out, float* label_out, int look_depth, int wn,
                                  const float* const* wg, const int64_t* wldg, const float* const* wx, const int64_t* wldx,
                                  const int* wB, const int* wN, const int* wK, float* const* wpartial, void* stream);
/* rh_adam_lazy_touched (refresh = 0, int64 indices) + rh_adam_lazy_sweep (RH_SWEEP_WINDOW) of the same step as ONE launch:
 * both parts claim a lazy row with atomicMax on its last-step word and the claimant applies the row's gradient. */
int rh_adam_lazy_step(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                      const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                      const int64_t* idesc, int B, int F, int samples_per_block, int32_t* err_flag, void* stream);
/* rh_adam_lazy_step with the sweep part restricted as rh_adam_lazy_sweep's `mode` (RH_SWEEP_WINDOW or RH_SWEEP_DENSE_TABLES:
 * touched rows + the dense K = 1 tables only, for a step whose lazy-table window sweep is deferred to a side stream). */
int rh_adam_lazy_step_mode(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                           const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                           const int64_t* idesc, int B, int F, int samples_per_block, int32_t* err_flag, int sweep_mode,
                           void* stream);
/* ... for int32 or int64 index columns (idx_is_i64 as rh_adam_lazy_touched): the row-sharded step's localised indices are int32
 * (round 6: its touched-rows step and the dense tables' step as one launch). */
int rh_adam_lazy_step_mode_idx(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                               const double* hyper, const float* ring, int ring_size, const int64_t* field_table,
                               const int64_t* idesc, int idx_is_i64, int B, int F, int samples_per_block, int32_t* err_flag,
                               int sweep_mode, void* stream);
int rh_adam_lazy_sweep(const int64_t* ldesc, int T, const int64_t* h_rows, const int64_t* h_window, int D,
                       const double* hyper, const float* ring, int ring_size, int mode, int64_t t_value, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Device-resident minibatch assembly (columnar dataset already in HBM)
 * replaces: TorchDataset.__getitem__ + default_collate  torch_rechub/utils/data.py:14-25,61-83
 *           and the 39 host->device copies at trainers/ctr_trainer.py:84-85
 * perm (N,) int64 sample order; *pos (device int64) = first position of this batch;
 * gathers rows perm[(pos + b) % N], b < B, of sparse (N,F) int64, dense (N,ND) fp32, label (N,) fp32
 * into the static batch buffers.  rh_batch_advance sets *pos = (*pos + B) % N (separate launch so
 * that every reader of *pos has finished).
 */
int rh_batch_gather(const int64_t* perm, const int64_t* pos, int64_t N, int B, const int64_t* sparse,
                    int F, const float* dense, int ND, const float* label, int64_t* sparse_out,
                    float* dense_out, float* label_out, void* stream);
int rh_batch_advance(int64_t* pos, int64_t B, int64_t N, void* stream);

/* In-batch negative sampling: out (B, K) int64, row i = K distinct columns drawn uniformly from {0..B-1} \ {i}.
 * replaces: the per-row randperm loop of inbatch_negative_sampling, torch_rechub/utils/match.py:136-145
 * rng (device int64 [2]): seed, call counter; bump the counter after the call with rh_batch_advance(rng + 1, 1, 0). */
int rh_inbatch_sample(const int64_t* rng, int B, int K, int64_t* out, void* stream);

/* The same draw for rows [row0, row0 + B) of a (cols x cols) problem: out (B, K), row r = K distinct columns from
 * {0..cols-1} \ {row0 + r}.  The random stream is keyed by the global row, so ranks holding slices of one global batch
 * (cross-rank in-batch negatives: their item embeddings all-gathered into `cols` candidates) draw what one process
 * would draw.  rh_inbatch_sample(rng, B, K, ...) == rh_inbatch_sample_rows(rng, B, B, 0, K, ...).
 * replaces: torch_rechub/utils/match.py:136-145 on the single-device branch of match_trainer.py:118-138 */
int rh_inbatch_sample_rows(const int64_t* rng, int B, int cols, int row0, int K, int64_t* out, void* stream);

/* ---- gated recurrences of DIEN: AUGRU (interest evolving) and GRU (interest extractor) ------------------------------
 * replaces: the per-step Python loop of AUGRU.forward over AUGRU_Cell.forward, torch_rechub/models/ranking/dien.py:30-36,
 *           60-66 (6 matmuls + ~12 elementwise kernels per step and as many again under autograd); and the
 *           nn.GRU(batch_first=True) of the interest extractor, dien.py:101,138-143.
 * xw   (B, T, 3D): input halves of the three gates for every step, x_t [Wu | Wr | Wh] + [bu | br | bh] (one GEMM by
 *                  the caller)
 * attn (B, T) or null: the step's attention weight a_t (0 on padded steps); null = 1 everywhere
 * U    (D, 3D):    [Uu | Ur | Uh];   state_bias (3D) or null: added to h U (nn.GRU's bias_hh)
 * forward, h_0 = 0:  s = h U + state_bias;  u = sigmoid(xw_u + s_u), r = sigmoid(xw_r + s_r), c = tanh(xw_h + r * s_h),
 *                    h_t = (1 - a_t u) h_{t-1} + a_t u c;   h_all (B, T, D) receives every h_t.
 *   nn.GRU is this with a = 1, u = 1 - z (negate the z rows of weight_ih / weight_hh / both biases) and r, n as r, c.
 * backward: g_hall (B, T, D) = upstream gradient of every h_t (may be null: zeros);
 *           d_xw (B, T, 3D) = gradient of xw;  d_huh (B, T, D) = gradient of s_h (so that
 *           dU = [h_0 .. h_{T-1}]^T [d_xw_u | d_xw_r | d_huh] is one GEMM by the caller and d state_bias its column sum);
 *           d_attn (B, T) (may be null when attn is).
 * D in {4, 8, 16, 32} (rh_augru_max_dim()); D / 4 lanes per sample, state in registers, U in LDS. */
int rh_augru_max_dim(void);
int rh_augru_fwd(const float* xw, const float* attn, const float* U, const float* state_bias, int B, int T, int D,
                 float* h_all, void* stream);
int rh_augru_bwd(const float* xw, const float* attn, const float* U, const float* state_bias, const float* h_all,
                 const float* g_hall, int B, int T, int D, float* d_xw, float* d_huh, float* d_attn, void* stream);

/* ---- row-sharded tables (one shard per rank) -----------------------------------------------------------------------
 * Global row g of a table lives on rank g % world as local row g / world.  rh_shard_localize rewrites an index matrix
 * idx (n_rows, F) (int64 / int32, contiguous: the all-gathered indices of the global batch) for this rank's shards:
 *   local[i, f] = g / world   if g % world == rank and g != pad[f]
 *               = sink[f]     otherwise (the shard's all-zero row, passed to the gather / scatter / optimizer kernels as
 *                             the field's padding_idx: read as zeros, never updated)
 * desc (device int64 [3F]): vocab[F] | pad[F] (-1: none) | sink[F].  g outside [0, vocab) sets RH_FLAG_INDEX_OOB
 * (and maps to the sink row).  The gathers of all ranks over `local` sum to the reference's lookup, one non-zero
 * contributor per element.
 * replaces: nn.Embedding lookup on a replicated table, torch_rechub/basic/layers.py:83-99 under
 *           nn.DataParallel (trainers/ctr_trainer.py:53-55), which broadcasts every table every step. */
int rh_shard_localize(const void* idx, int idx_is_i64, int64_t n_rows, int F, const int64_t* desc, int world, int rank,
                      int32_t* local, int32_t* err_flag, void* stream);
/* The (n_rows, F) int64 index matrix (row stride ld) as contiguous int32 for the wire of the index all-gather, saturating (ids
 * beyond int32 stay out of range, below -1 -> -1); written into the caller's slice of the gather buffer (in-place collective). */
int rh_shard_narrow(const int64_t* idx, int64_t ld, int64_t n_rows, int F, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECHUB_HIP_H */
