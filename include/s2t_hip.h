/*
 * libs2t_hip.so — C-ABI of the MI355X (gfx950) S2T hot path.
 *
 * The reference (xuchennlp/S2T, a fairseq fork) has no C ABI on this path: every FLOP goes through
 * torch.nn / ATen.  The seam this library replaces is therefore the set of ATen calls
 * made by the reference modules listed beside each entry point (paths relative to
 * /root/reference/fairseq).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain device pointers + explicit sizes/strides (in ELEMENTS unless a name ends in _bytes);
 *   - no allocation, no ownership transfer, stateless, stream-ordered (last argument: hipStream_t as void*);
 *   - return value: 0 = ok, negative = argument error (S2T_ERR_*), positive = hipError_t;
 *   - dtype enum: S2T_F32 (parity mode, exact-f32 MFMA) or S2T_BF16 (bf16 storage, fp32 accumulate,
 *     fp32 softmax / norm statistics / losses — the reference's mixed-precision contract, trainer.py:85-90).
 *   - activations are batch-major row matrices: row m = b * T + t, columns = channels.
 *
 * Packed rows (padding-free batches)
 *   The reference pads every utterance of a batch to the longest one (data/audio/speech_to_text_dataset.py:411-485) and
 *   computes on the padded frames; here the frames of utterance b may instead sit at rows cu[b] .. cu[b] + lens[b] - 1,
 *   followed by its HALO rows (at most (k-1)/2 frames behind the utterance's end, never beyond the padded length T: the
 *   frames of the padded batch whose depthwise-convolution output is not zero and therefore enters the BatchNorm batch
 *   statistics, modules/convolution.py:94-104) up to cu[b + 1].  Halo rows are treated exactly like padded frames (masked,
 *   no gradient); the padded frames behind them — all zero in the reference after the masks — are not stored at all.
 *   Buffers keep their padded size (B * T rows); rows >= cu[B] are never touched.
 *   - every (row_lens / lens, row_T / T) pair of this header also takes a ROW MAP: T = S2T_ROWS_PACKED, lens[m] >= 0
 *     ((b << 16) | t) on rows that hold a frame, < 0 on halo rows, lens[-1] = cu[B] (the live row count, read on the
 *     device so that one captured hipGraph serves batches of any fill).  T = S2T_ROWS_BOUND applies the bound without the
 *     mask (entry points that have no mask of their own document where they take it).
 *   - per-utterance entry points take `cu` ([B + 1] int32, NULL = uniform layout b * T) next to lens; T stays the padded
 *     length (position tables, BatchNorm divisor B * T).
 */
#ifndef S2T_HIP_H
#define S2T_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { S2T_OK = 0, S2T_ERR_ARG = -1, S2T_ERR_DTYPE = -2, S2T_ERR_ALIGN = -3, S2T_ERR_UNSUPPORTED = -4 };
enum { S2T_F32 = 0, S2T_BF16 = 1 };
enum { S2T_ACT_NONE = 0, S2T_ACT_RELU = 1, S2T_ACT_SWISH = 2, S2T_ACT_GLU = 3 };
enum { S2T_ROWS_PACKED = -1, S2T_ROWS_BOUND = -2 }; /* values of a row_T / T argument that make its lens pointer a row map */

int s2t_version(void);
/* number of compute units of the current device (for host-side grid heuristics) */
int s2t_device_cu_count(void);
/* Test / rehearsal utility: hold n compute units (one workgroup with 96 KiB of LDS each: no large-LDS workgroup of this library
 * fits beside it) until *stop != 0 or `ms` milliseconds (<= 2000) have passed, whichever comes first; *arrived (optional) counts
 * the workgroups that started.  Stands in for kernels that share the GPU with this path (the RCCL all-reduce beside backward). */
int s2t_occupy_cus(int n, int ms, const uint32_t* stop, uint32_t* arrived, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   C = epilogue( A_op[M,K] * B_op[K,N] )
 *
 * Replaces F.linear / nn.Conv1d(k=1) / nn.Conv1d(k=5,s=2) (as an overlapping-row GEMM) / torch.bmm in
 *   modules/s2t_transformer_layer.py:55-66 (FFN), modules/multihead_attention.py:239-263,367,420 (projections, bmm),
 *   modules/espnet_multihead_attention.py:88-106,331-347, modules/convolution.py:91,106 (pointwise convs),
 *   modules/speech_to_text/subsampling.py:145-159, modules/speech_to_text/ctc.py:59, models/transformer.py:1442
 * and their autograd backward (dgrad = NN form, wgrad = TN form with split-K accumulate).
 *
 * Operand storage:  a_kmajor = 0: A[m*lda + k]     a_kmajor = 1: A[k*lda + m]
 *                   b_kmajor = 0: B[n*ldb + k]     b_kmajor = 1: B[k*ldb + n]     (weights are [N][K])
 * Batching: batch index z in [0,batch) is split as z0 = z / zdiv, z1 = z % zdiv and each operand X is
 * offset by z0*X_s0 + z1*X_s1 elements (two-level (utterance, head) batching).
 * Epilogue, in order (v = fp32 accumulator):
 *   v += bias[n]; [GLU: out column c pairs accumulator columns c (value) and c + N/2 (gate),
 *   v = value*sigmoid(gate), C has N/2 columns]; [preact: store v before the activation];
 *   v = act(v); [dact_z: v *= act'(dact_z[m,n])]; [drop_p: inverted dropout, FairseqDropout of fairseq_dropout.py]; v *= alpha;
 *   [row_lens: v = 0 on rows with (global_row % row_T) >= row_lens[global_row / row_T]  (padded frames;
 *    the reference masks the branch output, not the residual: modules/convolution.py:109-116)];
 *   [residual: v += residual[m,n]]; store (c_dtype).   global_row = z*M + m.
 * split_k > 1 (wgrad): K is split over blockIdx.y and alpha*acc is ADDED to an fp32 C (no other epilogue stage
 *   allowed): through the fp32 workspace `ws` when it holds at least s2t_gemm_ws_floats() floats (every split stores
 *   its partial tiles with plain coalesced stores, a second kernel sums them into C — float atomics execute at the
 *   memory side on a multi-XCD part and are several times slower), with float atomics otherwise.  colsum_a (a_kmajor): the column sums of dY (= the bias gradient) are taken
 *   from the staged A tiles by the workgroups of the first tile column and added atomically (fp32).
 *   c_atomic == 2 (overwrite through the workspace; fp32 or bf16 C) also carries every epilogue stage above except GLU:
 *   the second kernel runs them on the summed tiles (same dropout keys as the one-pass kernel).  Without a workspace
 *   that form runs as one pass (split_k = 1), the same result.
 * Alignment: A, B base pointers and strides must keep 16-byte alignment of every row start.
 * ------------------------------------------------------------------------------------------------ */
typedef struct s2t_gemm_args {
  int32_t dtype;   /* A/B storage: S2T_F32 | S2T_BF16 */
  int32_t c_dtype; /* C / residual / preact storage */
  int32_t M, N, K;
  int32_t a_kmajor, b_kmajor;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C;       int64_t ldc;
  int32_t batch, zdiv;
  int64_t a_s0, a_s1, b_s0, b_s1, c_s0, c_s1;
  const void* bias;   /* [N] in bias_dtype, may be NULL */
  int32_t bias_dtype;
  int32_t act;        /* S2T_ACT_* */
  float alpha;
  const void* residual; int64_t ldr; /* c_dtype, same batch strides as C; may alias C */
  void* preact;         int64_t ldp; int64_t p_s0, p_s1; /* c_dtype; [M][N] pre-activation copy (GLU: value | gate halves); own batch strides */
  const void* dact_z;   int64_t ldz; int32_t dact; /* multiply by act'(z) with act = dact; z is c_dtype, same batch strides as C */
  const int32_t* row_lens; int32_t row_T;
  int32_t split_k;
  int32_t c_atomic; /* 1: add alpha*acc to fp32 C with atomics even when split_k == 1 (several batches share one C);
                       2: split-K through the workspace with C = alpha*sum (overwrite: C need not be zeroed; split_k > 1) */
  float* colsum_a;  /* optional, a_kmajor only: colsum_a[m] += alpha * sum_k A_op[m][k]  (bias gradient fused into wgrad) */
  float drop_p;     /* > 0: v = keep(seed, site, global_row*Nout + n) ? v/(1-p) : 0 after the activation / act' stage */
  uint32_t drop_site;
  const uint64_t* drop_seed; /* device pointer (a captured hipGraph replays with a fresh seed) */
  float* ws;         /* optional split-K workspace (see above); contents are scratch */
  int64_t ws_floats; /* its size in floats */
} s2t_gemm_args;

int s2t_gemm(const s2t_gemm_args* args, void* stream);
/* floats of workspace a split-K call needs for the two-phase (non-atomic) reduction; 0 when not applicable */
int64_t s2t_gemm_ws_floats(const s2t_gemm_args* args);
/* writes the (demangled) name of the kernel s2t_gemm would launch for args, as a profiler prints it */
int s2t_gemm_describe(const s2t_gemm_args* args, char* buf, int buflen);
/* Process-wide switch of s2t_gemm's large-tile path (256 x 256 x 64 tiles through LDS-DMA, gemm256.hip: bf16, both operands
 * row-major, K % 64 == 0, one batch, no split-K; results equal the 128 x 128 path's bit for bit — the same MFMA, the same order
 * over K).  mode: 0 never, 1 where it measured faster (default; first read from the environment, S2T_GEMM256: 256-row tiles from
 * 150 tiles, 128-row tiles when those reach 150), 2 / 3 whenever the arguments allow with 256- / 128-row tiles; negative: leave
 * as it is.  Returns the mode in force. */
int s2t_gemm_configure(int large_tile_mode);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (modules/layer_norm.py:30-35 -> torch.nn.LayerNorm, eps 1e-5).  x,y,dy,dx: [rows][cols] in
 * `dtype`; gamma/beta and the saved statistics are fp32.  row_lens/row_T (optional): rows of padded
 * frames are written as 0 in forward and carry no gradient in backward (the reference's per-layer
 * masked_fill, s2t_transformer.py:1828-1836, and the conv-module input mask, convolution.py:86-88).
 * dgamma/dbeta are ACCUMULATED (fp32).  ws: fp32 workspace [replicas][2][cols], zero on entry and left zero on exit
 * (the column sums are first spread over `replicas` copies so that no address sees more than gridDim/replicas
 * atomics, then folded and cleared), so one zero-initialised buffer can be reused by every call on the stream.
 * ------------------------------------------------------------------------------------------------ */
int s2t_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int64_t rows, int cols, float eps, const int32_t* row_lens, int row_T, void* stream);
int s2t_layernorm_bwd(int dtype, const void* x, const float* gamma, const void* dy, const float* mean,
                      const float* rstd, void* dx, float* dgamma, float* dbeta, float* ws, int replicas, int64_t rows,
                      int cols, const int32_t* row_lens, int row_T,
                      const void* dres /* optional [rows][cols]: dx += dres (gradient of the residual branch of a pre-LN block) */,
                      void* dx_drop /* optional second output (bf16, cols == 256 only): s2t_dropout(dx) under the mask
                                       (drop_p, drop_seed, drop_site) — the branch gradient of the block in front */,
                      float drop_p, const uint64_t* drop_seed, uint32_t drop_site, void* stream);
/* dgamma == dbeta == NULL in s2t_layernorm_bwd leaves the per-replica partial sums in ws ([replicas][2][cols], zeroed
 * before); s2t_layernorm_fold then adds them into the parameter gradients of MANY LayerNorms in one launch and zeroes
 * the workspaces again.  `entries` is a host array (passed to the kernel by value). */
typedef struct s2t_ln_fold_entry {
  float* ws;
  float* dgamma;
  float* dbeta;
  int32_t cols;
  int32_t reserved;
} s2t_ln_fold_entry;
int s2t_layernorm_fold(const s2t_ln_fold_entry* entries, int n, int replicas, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention probabilities from raw scores (fp32 in, `p_dtype` out), one row per (z, query):
 *   s[j] = (S[j] + (BD ? BD[Tq-1-i+j] : 0)) * scale ; key j >= key_lens[z / H] or (causal && j > i) -> -inf ;
 *   clamp != 0: s = clamp(s, -1e8, 1e8) (ESPnet flavour; -inf becomes -1e8) ; P = softmax_fp32(s).
 * Replaces modules/multihead_attention.py:367-403 and modules/espnet_multihead_attention.py:108-129,292-311,343-350
 * (rel_shift is an index computation here, not a pad/reshape/slice).  Backward:
 *   dS = P * (dP - sum_j P*dP) * scale ; dBD (optional) [row][n] = dS[n-(Tq-1-i)] inside the band, 0 outside.
 * ------------------------------------------------------------------------------------------------ */
int s2t_attn_softmax_fwd(int p_dtype, const float* S, int64_t ldS, const float* BD, int64_t ldBD, void* P, int64_t ldP,
                         int Z, int H, int Tq, int Tk, float scale, const int32_t* key_lens, int causal, int clamp,
                         void* Pdrop /* optional: dropout(P), same layout */, float drop_p, const uint64_t* drop_seed,
                         uint32_t drop_site, void* stream);
/* dBD rows are laid out HEAD-major ((h*B + b)*Tq + i) so that per head the (b,i) rows form one matrix for the
 * linear_pos weight-gradient GEMM; S/P/dP/dS rows are z-major (z = b*H + h). */
int s2t_attn_softmax_bwd(int dtype, const void* P, int64_t ldP, const float* dP, int64_t ldDP, void* dS, int64_t ldDS,
                         void* dBD, int64_t ldDBD, int Z, int H, int Tq, int Tk, float scale, float drop_p,
                         const uint64_t* drop_seed, uint32_t drop_site, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused (flash-style) attention, bf16, head dim 64: O = softmax(mask(scale * (Q K^T [+ rel-shift((Q+v) P^T)]))) V with the
 * score matrix kept on chip (same reference lines as above).  q/k/v/o element (b, t, h, c) at X[b*X_sb + t*X_sr + h*64 + c].
 * pos_p != NULL selects the relative-position form: Q+pos_u multiplies K, Q+pos_v multiplies pos_p[n][h*64 + c] with
 * n = Tq-1-i+j.  lse[(b*H+h)*Tq + i] = log-sum-exp of the scaled, masked scores (for the backward).  Dropout is applied to
 * the probabilities that multiply V (mask index ((b*H+h)*Tq + i)*Tk + j, as in s2t_attn_softmax_fwd).
 * Backward: delta[z][i] = sum_c dO*O; dq/dk/dv in the layouts of q/k/v; for the relative form dq receives only the
 * (Q+u) K^T part and `dbd` ([H][B][Tq][ldb], row n = Tq-1-i+j) receives the skewed dS for the position projections;
 * dbd_band_only != 0: only the band Tq-1-i <= n < Tq-1-i+Tk of each row is written — the caller keeps the rest of the
 * buffer zero (zero-filled once and reused: every call overwrites exactly the band).  delta is an OUTPUT (written by the
 * dQ kernel, read by the dK / dV kernel).
 * pos_pt != NULL (relative form): the TRANSPOSED projected positions, element (h*64 + c, n) at pos_pt[(h*64+c)*pt_ld + n],
 * zero-padded so that n in [-16, 2Tq-2+96] is readable (pt_ld % 8 == 0, 16-byte aligned).  The dQ kernel then adds the
 * (Q+v) P^T branch itself — dq is the complete gradient w.r.t. Q — and accumulates the pos_bias_u / pos_bias_v gradients
 * (column sums of the two branches, espnet_multihead_attention.py:339-345) into dpos_u / dpos_v ([H*64] fp32, atomics).
 * qv_out != NULL (relative form): the dQ kernel also writes Q + pos_v ([B*Tq][H*64] bf16, the operand of the position-table
 * gradient GEMM) instead of a separate s2t_bias_add_rows pass.
 * Limits: dk = 64; Tq, Tk and every row stride (q_sr, k_sr, v_sr, o_sr, p_sr; elements) below 65 536 — the tile loads form
 * row x stride with 24-bit multiplies — and, with relative positions, (2 Tq - 1) * p_sr below 2^32 (the position rows run to
 * 2 Tq - 2), else S2T_ERR_UNSUPPORTED.  An utterance of a packed batch may hold no row (cu[b] == cu[b + 1]): nothing of it is
 * read or stored.
 * ------------------------------------------------------------------------------------------------ */
int s2t_attn_fused_fwd(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr, const void* v,
                       int64_t v_sb, int64_t v_sr, void* o, int64_t o_sb, int64_t o_sr, float* lse, int B, int H, int Tq,
                       int Tk, int dk, const int32_t* key_lens, int causal, float scale, const void* pos_p, int64_t p_sr,
                       const float* pos_u, const float* pos_v, float drop_p, const uint64_t* drop_seed,
                       uint32_t drop_site,
                       const int32_t* cu_q, const int32_t* cu_k /* packed batch: rows of utterance b on the query / key side =
                       cu[b] .. cu[b+1] (X_sb is then unused on that side), or NULL */,
                       void* o_lo /* optional, layout of o (bf16): what the rounding of the fp32 output to bf16 dropped, itself
                       rounded to bf16.  The backward's delta = rowsum(dO * O) is taken on o + o_lo when it is given the same
                       buffer: dS = P (dP - delta) cancels where the probabilities are nearly uniform (encoder-decoder attention
                       over a few hundred keys, multihead_attention.py:367-420), and delta from the rounded o alone then carries
                       an error of the size of the difference */, void* stream);
int s2t_attn_fused_bwd(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr, const void* v,
                       int64_t v_sb, int64_t v_sr, const void* o, const void* dO, int64_t o_sb, int64_t o_sr,
                       const float* lse, float* delta, void* dq, void* dk, void* dv, void* dbd, int64_t ldb, int B, int H,
                       int Tq, int Tk, int dk_dim, const int32_t* key_lens, int causal, float scale, const void* pos_p,
                       int64_t p_sr, const float* pos_u, const float* pos_v, float drop_p, const uint64_t* drop_seed,
                       uint32_t drop_site, int dbd_band_only,
                       const void* pos_pt, int64_t pt_ld, float* dpos_u, float* dpos_v, void* qv_out,
                       const int32_t* cu_q, const int32_t* cu_k /* as s2t_attn_fused_fwd; lse / delta / dbd / qv_out keep their
                       padded strides (row b*Tq + i), of which only the utterance's own rows are touched */,
                       const void* o_lo /* optional: the forward's o_lo of the same call */, void* stream);

/* Relative-position attention backward, the (Q + pos_bias_v) branch behind s2t_attn_fused_bwd
 * (espnet_multihead_attention.py:331-337 backward; replaces a batched GEMM over the half-empty skewed dbd rows, the
 * element-wise add into dq and two column-sum passes):
 *   dqv[b,i,h,:] = sum_n dbd[h][b][i][n] * p[n][h*64 : h*64+64]  over the band n in [Tq-1-i, 2Tq-2-i] (dbd is zero elsewhere),
 *   dq[b,i,h,:] += dqv  (bf16, in place; dq_sb / dq_sr = batch / row strides in elements),
 *   dpos_u[h*64+c] += sum_{b,i} dq_before[b,i,h,c],   dpos_v[h*64+c] += sum_{b,i} dqv[b,i,h,c].
 * pos_pt: the TRANSPOSED projected positions as s2t_attn_fused_bwd takes them (element (c, n) at pos_pt[(h*64+c)*pt_ld + n],
 * zeros readable up to n = 2Tq-2+96).  The column sums go to one of `replicas` copies of dpos_u / dpos_v, `replica_stride`
 * floats apart (1 and 0 to accumulate in place; with the [replicas][2][cols] workspace of s2t_layernorm_fold the adds of the
 * thousand workgroups no longer serialise on 2 x 64 addresses per head).
 * bf16, dk == 64, ldb % 8 == 0, 16-byte aligned dbd / pos_pt, 8-byte aligned dq rows. */
/* The same branch AND this call's share of the position-table gradient in one pass over dbd (csrc/relpos_glue.hip; any Tq:
 * beyond 256 frames the position rows are walked in chunks of 512, every piece of a row's band still read once):
 *   dq += dqv and the two column sums as s2t_relpos_dqv (pos_p: the projected positions [2Tq-1][p_sr] as the forward takes
 *   them, not the transposed table),
 *   dp[n][h*64+c] = sum_{b,i} dbd[h][b][i][n] * qv[b*Tq+i][h*64+c]   (fp32 [2Tq-1][H*64], OVERWRITTEN: the gradient w.r.t. the
 *   projected positions, espnet_multihead_attention.py:313-331 backward — what the split-K GEMM over K = B*Tq produced);
 *   dp_part: scratch of B * (2Tq-1) * H*64 bf16 (per-utterance partial sums, summed in fp32 by a second kernel).
 *   dp == NULL: the partial table is left to the caller, who sums the tables of several calls (the layers of one backward
 *   pass, each with its own dp_part) in ONE launch with s2t_relpos_dp_reduce(dp_parts[n], dps[n] — host arrays of device
 *   pointers) — same sums, same order. */
int s2t_relpos_dp_reduce(const void* const* dp_parts, float* const* dps, int n, int B, int H, int Tq, int dk, void* stream);
int s2t_relpos_glue(const void* dbd, int64_t ldb, const void* pos_p, int64_t p_sr, const void* qv, void* dq, int64_t dq_sb,
                    int64_t dq_sr, float* dpos_u, float* dpos_v, int replicas, int64_t replica_stride, void* dp_part, float* dp,
                    int B, int H, int Tq, int dk,
                    const int32_t* cu /* packed batch (dq rows of utterance b from cu[b]; dbd and qv as s2t_attn_fused_bwd left
                    them), or NULL */,
                    void* dq_lo /* Tq > 256 only, optional: scratch [B*Tq][H*64] bf16 (need not be initialised).  The 2Tq - 1
                    position rows are then walked in chunks of 512 and dq takes every chunk's share in turn; with dq_lo the
                    rounding remainder of the running bf16 sum travels from chunk to chunk, so that dq is rounded ONCE as in the
                    one-chunk form (without it: once per chunk a row's band meets, 2 at 502 frames, 3 at 1004) */,
                    void* stream);
int s2t_relpos_dqv(const void* dbd, int64_t ldb, const void* pos_pt, int64_t pt_ld, void* dq, int64_t dq_sb, int64_t dq_sr,
                   float* dpos_u, float* dpos_v, int replicas, int64_t replica_stride, int B, int H, int Tq, int dk,
                   void* stream);
/* The whole backward of the relative-position SELF-attention in one pass, sequences of up to 256 frames
 * (csrc/relpos_bwd.hip; espnet_multihead_attention.py:292-356 backward): what s2t_attn_fused_bwd (relative form, dbd output)
 * followed by s2t_relpos_glue computes, with the scores, the position band and the exponentials formed ONCE and the skewed
 * score gradient dbd kept on the chip (one workgroup per (utterance, head); no dbd, delta or Q + pos_bias_v buffer):
 *   dq (complete: both the (Q+u) K^T and the (Q+v) P^T branch), dk, dv in the layouts of q, k, v (bf16);
 *   dpos_u / dpos_v += column sums of the two dq branches (replicated workspace as s2t_relpos_glue);
 *   dp_part [B][2T-1][H*64] bf16: this call's per-utterance partial tables of the gradient w.r.t. the projected positions
 *   (summed by s2t_relpos_dp_reduce, several calls per launch).
 * o, dO: the forward's output and its gradient (layout o_sb / o_sr): delta = rowsum(dO * o) is taken inside.  lse: the
 * forward's [B*H][T].  key_lens (optional) masks keys at and beyond key_lens[b]; cu: packed batch (rows of utterance b of q, k,
 * v, o, dO and their gradients = cu[b] .. cu[b+1]; T stays the padded length: the centre of the position table, the stride of
 * lse and the index space of the dropout mask).  Dropout as s2t_attn_fused_fwd (mask index ((b*H+h)*T + i)*T + j).
 * Limits: bf16, dk = 64, T <= 256 (S2T_ERR_UNSUPPORTED beyond: use the three-kernel route), row strides % 8 == 0, 16-byte
 * aligned operands. */
int s2t_relpos_attn_bwd(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr, const void* v,
                        int64_t v_sb, int64_t v_sr, const void* o, const void* dO, int64_t o_sb, int64_t o_sr, const float* lse,
                        void* dq, void* dk, void* dv, const void* pos_p, int64_t p_sr, const float* pos_u, const float* pos_v,
                        float* dpos_u, float* dpos_v, int replicas, int64_t replica_stride, void* dp_part, int B, int H, int T,
                        int dk_dim, const int32_t* key_lens, float scale, float drop_p, const uint64_t* drop_seed,
                        uint32_t drop_site, const int32_t* cu, void* stream);
/* The same schedule for PLAIN scaled-dot-product attention (multihead_attention.py:161-431 backward; csrc/relpos_bwd.hip, the
 * kernel without its position terms): what the two kernels of s2t_attn_fused_bwd compute — dq, dk, dv in the layouts of q, k, v —
 * from ONE launch, scores and probabilities formed once, delta = rowsum(dO * (o + o_lo)) taken inside (no delta buffer).
 * Queries and keys may differ in number (Tq, Tk) and packing (cu_q, cu_k); causal != 0 masks key j > query i; key_lens masks
 * keys at and beyond key_lens[b]; dropout as s2t_attn_fused_fwd (mask index ((b*H+h)*Tq + i)*Tk + j).  An utterance without
 * queries gives zero dk / dv rows, one without keys zero dq rows.
 * Limits: bf16, dk = 64, Tk <= 256 (S2T_ERR_UNSUPPORTED beyond: s2t_attn_fused_bwd), row strides % 8 == 0, 16-byte aligned. */
int s2t_attn_bwd_one_pass(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr, const void* v,
                          int64_t v_sb, int64_t v_sr, const void* o, const void* dO, int64_t o_sb, int64_t o_sr, const float* lse,
                          void* dq, void* dk, void* dv, int B, int H, int Tq, int Tk, int dk_dim, const int32_t* key_lens,
                          int causal, float scale, float drop_p, const uint64_t* drop_seed, uint32_t drop_site,
                          const int32_t* cu_q, const int32_t* cu_k, const void* o_lo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Grouped weight gradients: all dW[M=Nout][N=Kin] += alpha * dY[K=rows][M]^T @ X[K][N] (bf16 in, fp32 accumulate) of one
 * backward pass in one persistent launch + one reduce launch (csrc/gemm_grouped.hip).  The host cuts the problems into
 * work items (one 128x128 output tile x `ksteps` K-steps of 64 rows) and lists the output tiles for the reduction;
 * all three tables live in device memory.  colsum (optional): colsum[m] += alpha * sum_k dY[k][m] (bias gradient).
 *   items_dev: [n_items][4] int32 = (problem, tile row, tile column, K split)
 *   tiles_dev: [n_tiles][4] int32 = (problem, tile row, tile column, 0)
 *   ws: fp32 scratch, problem p uses [ws_base, ws_base + tiles_p * nsplit * 16384)
 * ------------------------------------------------------------------------------------------------ */
typedef struct s2t_wgrad_problem {
  const void* A;  /* dY: element (k, m) at A[k*lda + m], bf16 */
  const void* B;  /* X:  element (k, n) at B[k*ldb + n], bf16 */
  float* C;       /* dW [M][ldc], accumulated into */
  float* colsum;  /* [M] or NULL */
  int64_t lda, ldb, ldc, ws_base;
  int32_t M, N, K, tiles_n, ksteps, nsplit;
  float alpha;
  int32_t next;   /* index of the next problem accumulating into the same C (tied weights; same M, N), or -1:
                     only the first problem of such a chain appears in tiles_dev */
  const int32_t* k_live; /* optional device scalar (s2t_wgrad_grouped256 only): K = min(K, *k_live), read by the kernel, and the
                     K-steps are re-cut into nsplit balanced parts — a packed batch's live rows ("Packed rows" above) */
} s2t_wgrad_problem;

int s2t_wgrad_grouped(const s2t_wgrad_problem* problems_dev, int n_problems, const int32_t* items_dev, int n_items,
                      const int32_t* tiles_dev, int n_tiles, float* ws, int any_k_tail, void* stream);
/* The same tables cut for 256 x 256 output tiles and K-steps of 32 rows (`ksteps` counts those; ws: 65536 floats per
 * tile and split), executed by the LDS-DMA fed kernel: both operands bf16 with lda % 8 == 0, ldb % 8 == 0, 16-byte aligned
 * bases and spans below 2 GiB (the caller checks; s2t_wgrad_grouped takes everything else). */
int s2t_wgrad_grouped256(const s2t_wgrad_problem* problems_dev, int n_problems, const int32_t* items_dev, int n_items,
                         const int32_t* tiles_dev, int n_tiles, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Feature front-end (dataloader stage of the reference, here on the device)
 *   s2t_fbank: data/audio/audio_utils.py:59-79 (_get_torchaudio_fbank -> torchaudio.compliance.kaldi.fbank with its
 *     defaults: snip_edges, dither 0, DC removal, pre-emphasis, window, zero-pad to nfft, power spectrum, mel, log).
 *     wave: [B][wave_stride] fp32 in int16 range (audio_utils.py:31-32), n_samples[B]; feat: [B][max_frames][n_mel]
 *     fp32, rows >= 1 + (n_samples - win) / shift are zero-filled (collater padding).  window: [win]; mel_t:
 *     [nfft/2+1][n_mel] (transposed mel bank matrix); both built by the host (s2t_amd/audio.py).
 *   s2t_utterance_cmvn: data/audio/feature_transforms/utterance_cmvn.py:31-45 over the first n_frames[b] rows of
 *     each utterance; x, y: [B][stride_b] fp32 with rows of C features; may alias.
 * ------------------------------------------------------------------------------------------------ */
int s2t_fbank(const float* wave, int64_t wave_stride, const int32_t* n_samples, float* feat, int64_t feat_stride_b,
              int max_frames, int B, int win, int shift, int nfft, const float* window, const float* mel_t, int n_mel,
              float preemph, int remove_dc, float log_floor, void* stream);
int s2t_utterance_cmvn(const float* x, float* y, const int32_t* n_frames, int64_t stride_b, int B, int C, int norm_means,
                       int norm_vars, void* stream);
/* SpecAugment time warp (data/audio/feature_transforms/specaugment.py:97-112: two cv2.resize(INTER_LINEAR) calls along
 * time; OpenCV is a third-party dependency absent from the reference tree and this image, its published row mapping is
 * restated — parity unpinned).  x, y: [B][max_frames][C] fp32, y != x; warp: [B][2] int32 = (w0, w) drawn by the host in
 * the reference's numpy order, w0 <= 0 = no warp for that utterance; rows >= n_frames[b] are copied.
 * mean_out (optional, [B]): mean of the un-warped utterance (the fill value of mask_value = None). */
int s2t_time_warp(const float* x, float* y, const int32_t* n_frames, int64_t stride_b, int B, int max_frames, int C,
                  const int32_t* warp, float* mean_out, void* stream);

/* SpecAugment frequency / time masking (data/audio/feature_transforms/specaugment.py:114-131; the time warp in front is s2t_time_warp), in place on
 * x [B][stride_b] (rows of C features, first n_frames[b] rows): masks [B][n_freq + n_time][2] int32 = (start, width),
 * frequency intervals first; masked cells := value[b], or the utterance mean when value_is_mean (value is then output). */
int s2t_specaugment(float* x, const int32_t* n_frames, int64_t stride_b, int B, int max_frames, int C, const int32_t* masks,
                    int n_freq, int n_time, float* value, int value_is_mean, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / gather pieces of S2TTransformerEncoder.forward and TransformerDecoder
 *   s2t_add_positions : x = scale*x + (t < lens[b] ? tab[t+pos_offset] : 0)       s2t_transformer.py:1773-1787
 *   s2t_mask_rows     : zero padded frames in place                                s2t_transformer.py:1765,1828-1836
 *   s2t_embedding_fwd : out = scale*E[tok] + tab[pos]  ;  _bwd: dE[tok] += scale*dOut (pad row skipped)
 *                                                                                  models/transformer.py:1304-1323
 *   s2t_glu_bwd       : backward of GLU over channels (Z = [value | gate])         subsampling.py:131-144, convolution.py:92
 *   s2t_colsum_accum  : db[n] += sum_m dY[m,n] (bias gradients)
 * ------------------------------------------------------------------------------------------------ */
int s2t_add_positions(int dtype, void* x, const float* tab, const int32_t* lens, int64_t rows, int T, int d,
                      float scale, int pos_offset, void* stream);
int s2t_mask_rows(int dtype, void* x, const int32_t* lens, int64_t rows, int T, int d, void* stream);
/* out[m, 0:n] = x[m, 0:n] + bias[0:n]  (q + pos_bias_u / q + pos_bias_v, espnet_multihead_attention.py:335-337) */
int s2t_bias_add_rows(int dtype, const void* x, int64_t ldx, const float* bias, void* out, int64_t ldo, int64_t rows,
                      int n, void* stream);
int s2t_embedding_fwd(int dtype, const int64_t* tokens, const int32_t* pos, const void* E, const float* tab, void* out,
                      int64_t n, int d, float scale, void* stream);
int s2t_embedding_bwd(int dtype, const int64_t* tokens, const void* dout, float* dE, int64_t n, int d, float scale,
                      int64_t pad_idx, void* stream);
int s2t_glu_bwd(int dtype, const void* Z, const void* dY, void* dZ, int64_t rows, int n, const int32_t* lens, int T,
                int out_pad /* dZ has out_pad extra rows behind every T rows, not written (T > 0 required) */, void* stream);
int s2t_colsum_accum(int dtype, const void* dY, int64_t ld, float* db, int64_t rows, int n, void* stream);
/* relative-position attention backward glue: a[row][0:n] += b[row][0:n] in place, du[c] += column sums of the OLD a,
 * dv[c] += column sums of b (pos_bias_u / pos_bias_v gradients, espnet_multihead_attention.py:313-356); bf16, n = 256 */
int s2t_add_colsum2(int dtype, void* a, int64_t lda, const void* b, int64_t ldb, float* du, float* dv, int64_t rows, int n,
                    void* stream);
int s2t_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst = scale * src (bf16 -> fp32; n % 4 == 0): the way back of a gradient bucket reduced in bf16 */
int s2t_cast_bf16_to_f32(const void* src, float* dst, int64_t n, float scale, void* stream);
/* out[r, c] = keep(seed, site, r*cols + c) ? x[r, c] / (1-p) : 0   (FairseqDropout, modules/fairseq_dropout.py; the same
 * call on a gradient is its backward) */
int s2t_dropout(int dtype, const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int cols, float p,
                const uint64_t* seed, uint32_t site, void* stream);
int s2t_axpy(int dtype, const void* a, const void* b, void* y, float alpha, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer / gradient bookkeeping on FLAT fp32 buffers (one launch for the whole model)
 *   s2t_adam_step   : fairseq Adam (optim/adam.py:146-226) reading {lr, step_size, grad_scale} from the
 *                     device array `hyper` (so a captured hipGraph replays with fresh values); g is multiplied
 *                     by grad_scale first; refreshes the bf16 weight shadow when given.
 *   s2t_sumsq_accum : *out += sum g^2 (gradient norm, utils.py:328-369)
 *   s2t_clip_coef   : hyper[2] = mult*min(1, max_norm/(sqrt(sumsq)*mult+1e-6)), hyper[3] = grad norm
 *                     (trainer.py:729-741: multiply_grads(world/sample_size) then clip_grad_norm);
 *                     mult <= 0: the multiplier is read from hyper[2] (device-resident, so a captured update can
 *                     be replayed on a batch with another sample size)
 * ------------------------------------------------------------------------------------------------ */
int s2t_adam_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, int64_t n, float beta1, float beta2,
                  float eps, float weight_decay, const float* hyper, void* stream);
int s2t_clip_coef(const float* sumsq, float max_norm, float mult, float* hyper, void* stream);
int s2t_sumsq_accum(const float* g, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Conformer convolution module core (modules/convolution.py:94-104), channels-last (B,T,C):
 *   s2t_dwconv_fwd : y[b,t,c] = sum_k x[b,t+k-(K-1)/2,c] * w[c, flip ? K-1-k : k]   (zero outside [0,T)).
 *        scale/shift != NULL : y = act(y*scale[c] + shift[c]), frames t >= lens[b] -> 0   (eval: BN folded)
 *        stats != NULL       : one row [2][C] of partial sums (sum y | sum y^2) per workgroup, s2t_dwconv_stat_partials(B,T)
 *                              rows in all (train: BN batch stats; no atomics — s2t_bn_finalize adds the rows in a fixed order)
 *        flip = 1 gives the input gradient of the same convolution.
 *   s2t_dwconv_bwd_weight : dw[c,k] += sum_{b,t} dD[b,t,c] * G[b,t+k-(K-1)/2,c]
 *   s2t_bn_finalize : stats -> scale = gamma*rstd, shift = beta - mean*scale (+ running-stat update, momentum,
 *                     unbiased variance) ; training = 0 uses the running statistics.
 *   s2t_bn_act_fwd  : out = act(D*scale + shift), padded frames -> 0
 *   s2t_bn_act_bwd  : dD from dOut (reduce, fixed-order fold, apply); sums[0:C] = sum du (= dbeta), sums[C:2C] = sum du*xhat
 *                     (= dgamma); ws: s2t_bn_bwd_partials(rows) x 2C floats of scratch; dgamma / dbeta (optional): the sums are also
 *                     added to these parameter-gradient vectors
 * ------------------------------------------------------------------------------------------------ */
int s2t_dwconv_fwd(int dtype, const void* x, const float* w, void* y, int B, int T, int C, int K, int flip,
                   const float* scale, const float* shift, int act, const int32_t* lens,
                   const int32_t* cu /* packed batch: rows of utterance b = cu[b] .. cu[b+1] (frames, then halo rows), or NULL */,
                   float* stats, void* stream);
/* eval: depthwise conv + BatchNorm (running statistics) + activation + padded-frame mask in one launch — the affine
 * gamma * rsqrt(running_var + eps), beta - running_mean * that is folded inside the kernel (convolution.py:94-104, eval) */
int s2t_dwconv_bn_eval_fwd(int dtype, const void* x, const float* w, void* y, int B, int T, int C, int K, const float* gamma,
                           const float* beta, const float* running_mean, const float* running_var, float eps, int act,
                           const int32_t* lens, const int32_t* cu, void* stream);
int s2t_dwconv_bwd_weight(int dtype, const void* G, const void* dD, float* dw, float* ws /* [replicas][C][K]: zero in, zero out */,
                          int replicas, int B, int T, int C, int K, void* stream);
int s2t_dwconv_wgrad_partials(int B, int T); /* rows of C*K floats s2t_dwconv_bwd_weight needs in ws (pass as `replicas`) */
/* s2t_conv_bwd_fused (bf16): the backward of the convolution module's middle in one launch + one fold —
 *   dD = BatchNorm backward pass 2 on (D, dA) from the folded sums of s2t_bn_act_bwd (called with dD = NULL: reduce + fold
 *   only), dG = depthwise conv of dD with the flipped kernel, dZ [B*T][2C] = GLU backward of dG on the saved value | gate
 *   Z, dw[c][k] += sum_t dD[t][c] * G[t + k - pad][c] (G = the GLU output the forward convolved).  Replaces the apply pass of
 *   s2t_bn_act_bwd, s2t_dwconv_fwd(flip), s2t_glu_bwd and s2t_dwconv_bwd_weight (convolution.py:92-104 backward) without
 *   passing dD and dG through HBM.  ws: B * ceil(T/32) rows of C*K floats.
 *   dw == NULL: the partial rows stay in ws for the caller to fold, several modules per launch, with
 *   s2t_rows_fold_add(partials[count], outs[count] — host arrays of device pointers; outs[i][0:n] += the fixed-order sum of the
 *   `rows` rows of n floats of partials[i]) — the same fold. */
int s2t_rows_fold_add(const float* const* partials, float* const* outs, int count, int rows, int64_t n, void* stream);
int s2t_conv_bwd_fused(const void* D, const void* dA, const void* G, const void* Z, const float* w, const float* scale,
                       const float* shift, const float* mean, const float* rstd, const float* sums, float count, int act,
                       const int32_t* lens, const int32_t* cu /* packed batch, as s2t_dwconv_fwd */, void* dZ, float* dw,
                       float* ws, int B, int T, int C, int K, void* stream);
int s2t_dwconv_stat_partials(int B, int T);
int s2t_bn_bwd_partials(int64_t rows);
int s2t_bn_finalize(const float* stats, int partials, float count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                    float* mean, float* rstd, int C, void* stream);
int s2t_bn_act_fwd(int dtype, const void* D, void* out, const float* scale, const float* shift, int act, int64_t rows,
                   int C, const int32_t* lens, int T, void* stream);
int s2t_bn_act_bwd(int dtype, const void* D, const void* dOut, void* dD, const float* scale, const float* shift,
                   const float* mean, const float* rstd, float* sums, float* ws, float* dgamma, float* dbeta, float count,
                   int act, int64_t rows, int C, const int32_t* lens, int T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Vocabulary-wide kernels.  logits are row matrices [rows][ld] (batch-major: row = b*T + t).
 *   s2t_argmax_lse     : per row first arg-max, its log-prob, logsumexp        s2t_ctc.py:312-324, utils.py:470-481
 *   s2t_ctc_collapse   : pad->blank, unique_consecutive, drop blank, score      s2t_ctc.py:326-347
 *   s2t_ls_cross_entropy : label-smoothed CE summed over non-pad rows + unit gradient
 *                          sums[0..3] += loss, nll, n_correct, n_total          label_smoothed_cross_entropy.py:42-60
 *   s2t_ctc_loss_fwd/_bwd : CTC alpha/beta (log space) and d nll / d logits     criterions/ctc.py:243-245,435-474
 *                          (torch.nn.CTCLoss(blank, reduction="none", zero_infinity=True)); Lmax = 2*max(S)+1 (odd)
 * ------------------------------------------------------------------------------------------------ */
int s2t_argmax_lse(int dtype, const void* logits, int64_t ld, int64_t rows, int V, int32_t* idx, float* top_lp,
                   float* lse, const int32_t* live /* optional device scalar: only rows < *live (a packed batch's live rows) */,
                   void* stream);
/* CTC head + greedy arg-max in one launch for d = 256 (csrc/ctc_head.hip): idx[r] = first arg-max of x[r] . w^T + bias over the V
 * vocabulary entries, top_lp[r] = its log-probability (max - logsumexp), lse[r] = the logsumexp; fp32 accumulation of the bf16
 * products, fp32 bias — the arithmetic of s2t_gemm with an fp32 output followed by s2t_argmax_lse, without the [M][V] fp32 logits
 * ever reaching HBM.  Replaces, for greedy decoding, modules/speech_to_text/ctc.py:60-63 (ctc_projection) +
 * models/speech_to_text/s2t_ctc.py:312-328 (log-softmax, arg-max and its log-probability per frame).
 * x: bf16 rows [M][ldx] (ldx >= 256, % 8 == 0), w: bf16 [V][256], bias: fp32 [V] or NULL, any of idx / top_lp / lse may be NULL
 * (not all).  live: optional device scalar, only rows < *live are computed (a packed batch's live rows).  Limits: V >= 128,
 * V * 512 < 2^32 (S2T_ERR_UNSUPPORTED beyond), 16-byte aligned x and w. */
int s2t_ctc_head_greedy(const void* x, int64_t ldx, const void* w, const float* bias, int64_t M, int V, int32_t* idx,
                        float* top_lp, float* lse, const int32_t* live, void* stream);
/* SATE adapter distribution (modules/speech_to_text/adapter.py:214-217): P = softmax(x * inv_tau) per row, and its
 * backward dx = P * (dP - sum P dP) * inv_tau */
int s2t_row_softmax_fwd(int dtype, const void* x, int64_t ldx, void* p, int64_t ldp, int64_t rows, int V, float inv_tau,
                        const int32_t* live /* optional device scalar: only rows < *live (a packed batch's live rows) */,
                        void* stream);
int s2t_row_softmax_bwd(int dtype, const void* p, int64_t ldp, const void* dp, int64_t lddp, void* dx, int64_t lddx,
                        int64_t rows, int V, float inv_tau, const int32_t* live /* as above */, void* stream);
int s2t_ctc_collapse(const int32_t* idx, const float* top_lp, const int32_t* lens,
                     const int32_t* cu /* packed batch: idx / top_lp rows of utterance b from cu[b], or NULL; outputs stay [B][T] */,
                     int B, int T, int blank, int64_t* out_tokens, int32_t* out_lens, float* out_scores, void* stream);
int s2t_ls_cross_entropy(int dtype, const void* logits, int64_t ld, int64_t rows, int V, const int64_t* target,
                         int64_t pad_idx, float eps, void* dlogits, int64_t ldd, float* sums,
                         float* ws /* rows x 4 floats of scratch: per-row terms, folded into sums in a fixed order */,
                         const int32_t* live /* optional device scalar: only rows < *live (a packed batch's live rows) */,
                         void* stream);
/* force_emits (optional, [B][T] int64, -1 = free): imputer loss of torch_imputer/imputer.cu:57-215,339-477 (a frame
 * pinned to one extended-label state).  paths (optional, [B][T][Lmax] int32): max-product (Viterbi) recursion with
 * back-pointers of torch_imputer/best_alignment.cu:57-201 instead of log-sum-exp (then beta may be NULL).
 * wrt_logprobs = 1: gradient w.r.t. log-probabilities (what torch_imputer returns) instead of w.r.t. logits. */
int s2t_ctc_loss_fwd(int dtype, const void* logits, int64_t ld, int B, int T, int V, const float* lse,
                     const int64_t* targets, int ldt, const int32_t* tgt_lens, const int32_t* in_lens, int blank,
                     float* alpha, float* beta, int Lmax, float* nll, const int64_t* force_emits, int32_t* paths,
                     const int32_t* cu /* packed batch: logits / lse rows of utterance b from cu[b], or NULL; alpha, beta, paths,
                     force_emits keep the padded [B][T] strides */, void* stream);
int s2t_ctc_loss_bwd(int dtype, const void* logits, int64_t ld, int B, int T, int V, const float* lse,
                     const int64_t* targets, int ldt, const int32_t* tgt_lens, const int32_t* in_lens, int blank,
                     const float* alpha, const float* beta, int Lmax, const float* nll, float gscale,
                     const float* gscale_dev /* optional device scalar multiplied into gscale (upstream gradient) */,
                     void* grad, int64_t ldg, int wrt_logprobs,
                     const int32_t* cu /* packed batch: logits / lse / grad rows as in s2t_ctc_loss_fwd, or NULL */, void* stream);
/* best-alignment backtrace (torch_imputer/imputer.py:245-259,311-323) on the device: states[b][t], -1 beyond the length */
int s2t_ctc_backtrace(const float* alpha, const int32_t* paths, const int32_t* tgt_lens, const int32_t* in_lens, int B,
                      int T, int Lmax, int32_t* states, void* stream);

/* ---- CTC prefix scores for joint attention/CTC beam search ------------------------------------
 * Replaces the host-side scorer the reference drives from fairseq/sequence_generator.py:255-388
 * (espnet.nets.ctc_prefix_score.CTCPrefixScore: initial_state / __call__; third-party, numpy, one hypothesis at a time).
 * lp: fp32 log-probabilities, frame t of sentence b at lp + (b*T + t)*ld; sent[r]: sentence of hypothesis row r;
 * state r: [T][2] fp32 per hypothesis (log-prob of the prefix over frames 0..t ending in its last label | in blank).
 * s2t_ctc_prefix_init : the empty prefix's state for R rows.
 * s2t_ctc_prefix_score: for R rows x K candidate tokens: psi[r][k] = log prefix probability of (prefix_r + cand[r][k])
 *                       (</s>: probability of the prefix ending there; blank: -1e10), and, if r_new != NULL, the extended
 *                       prefix's state r_new[r][k][T][2].  out_len = labels already in the prefix (the decoding step),
 *                       last[r] = the prefix's last token. */
int s2t_ctc_prefix_init(const float* lp, int64_t ld, int T, const int32_t* in_lens, const int32_t* sent, int R, int blank,
                        float* r0, void* stream);
int s2t_ctc_prefix_score(const float* lp, int64_t ld, int T, const int32_t* in_lens, const int32_t* sent,
                         const float* r_prev, const int64_t* last, int out_len, const int64_t* cand, int R, int K, int blank,
                         int eos, float* psi, float* r_new, void* stream);

/* ---- CTC-guided compression of the frame axis -------------------------------------------------
 * Replaces the per-utterance boolean indexing of S2TTransformerEncoder.forward, s2t_transformer.py:1948-1986
 * (--compression-metric threshold, --compression-mode create).
 * s2t_ctc_compress_plan : keep[b][t] = t < lens[b] && exp(logits[b*T+t][blank] - lse[b*T+t]) < threshold;
 *                         src[b][j] = frame index of the j-th kept frame, new_lens[b] = number kept (lse from s2t_argmax_lse)
 * s2t_compress_rows     : scatter = 0: out[b][j][:] = j < new_lens[b] ? in[b][src[b][j]][:] : 0   (in [B][T][C], out [B][Tn][C])
 *                         scatter = 1: out[b][src[b][j]][:] = in[b][j][:] for j < new_lens[b]      (in [B][Tn][C], out [B][T][C],
 *                         the backward pass; the caller zero-fills out).  Rows must be multiples of 16 bytes. */
int s2t_ctc_compress_plan(int dtype, const void* logits, int64_t ld, const float* lse, const int32_t* lens, int B, int T,
                          int blank, float threshold, int32_t* src, int32_t* new_lens, void* stream);
int s2t_compress_rows(int dtype, const void* in, void* out, const int32_t* src, const int32_t* new_lens, int B, int T,
                      int Tn, int C, int scatter, void* stream);

/* ---- Packed rows <-> padded rows (see "Packed rows" at the top).  map: the row map of the batch (map[-1] = live rows);
 * padded: [B*T][C], packed: [rows >= live][C]; rows must be multiples of 16 bytes.
 *   to_packed = 1: packed[m] = padded[b*T + t] for map[m] = (b << 16 | t) >= 0, halo rows := 0     (m < live)
 *   to_packed = 0: padded[b*T + t] = packed[m] on the rows that hold a frame; the caller zero-fills `out` first (padded
 *                  frames are zero after the reference's masks, s2t_transformer.py:1765,1828-1836).
 * Each direction is the other's backward pass. */
int s2t_pack_rows(int dtype, const void* in, void* out, const int32_t* map, int64_t rows, int T, int C, int to_packed,
                  void* stream);
/* The geometry itself, from the batch's lengths (one launch per batch, outside a captured step): cu [B + 1]; map_buf: 4 header
 * words (the last one = cu[B], the live row count) followed by the B * T row-map entries — the row map pointer the other entry
 * points take is map_buf + 4.  halo: rows kept behind every utterance (never beyond T).  B <= 1024, T <= 65535. */
int s2t_rows_geometry(const int32_t* lens, int B, int T, int halo, int32_t* cu, int32_t* map_buf, void* stream);

/* ---- Per-batch bookkeeping of a training step, one launch per item, written INTO tensors a captured step reads (round 5).
 * The reference derives these inside forward() from the collater's fields (data/audio/speech_to_text_dataset.py:411-485):
 * s2t_subsampled_lengths: modules/speech_to_text/subsampling.py:150-154 (get_out_seq_lens_tensor: n_layers stride-2
 *   convolutions, l -> floor((l - 1) / 2) + 1) and data/data_utils.py:518-522 (lengths_to_padding_mask): len64 / len32 [B],
 *   mask_u8 [B][Tp] = (t >= l_b) as bytes (a torch.bool tensor); any output may be NULL.
 * s2t_token_positions: utils.py:240-250 (make_positions): positions[b][u] = (# non-pad tokens up to and including u) * nonpad
 *   + pad_idx, counts[b] = non-pad tokens of row b (the target-side key lengths, models/transformer.py:1340-1342).
 * s2t_ctc_targets: criterions/ctc.py:516-540: tmat [B][U] = the labels of row b (neither pad nor eos) in order, followed by the
 *   dropped tokens in order; counts[b] = labels.  tmat must not alias target.
 * s2t_gather_rows_i64: out[m] = map[m] >= 0 ? src[(map[m] >> 16) * U + (map[m] & 0xffff)] : fill for every row m < rows of a
 *   packed batch's row map (the flattened targets of the cross-entropy on packed target rows). */
int s2t_subsampled_lengths(const int64_t* src_lengths, int B, int Tp, int n_layers, int64_t* len64, int32_t* len32, void* mask_u8,
                           void* stream);
int s2t_token_positions(const int64_t* tokens, int B, int U, int64_t pad_idx, int32_t* positions, int32_t* counts, void* stream);
int s2t_ctc_targets(const int64_t* target, int B, int U, int64_t pad_idx, int64_t eos_idx, int64_t* tmat, int32_t* counts, void* stream);
int s2t_gather_rows_i64(const int64_t* src, const int32_t* map, int64_t rows, int U, int64_t fill, int64_t* out, void* stream);

/* ---- PDS multi-scale fusion: depthwise convolution with kernel = stride = r, no padding --------
 * The depthwise stage of DownSampleConvolutionModule (fairseq/modules/downsample_convolution.py:45-54,97-100) as used by
 * PDSS2TTransformerEncoder.forward, pdss2t_transformer.py:1187-1233.  x [B][Tin][C] channels-last, w [C][r] fp32,
 * y / dy [B][Tout = Tin / r][C]; r <= 8.
 * s2t_dwpool_fwd : y = bias + sum_k x[t*r + k] * w[k]; stats != NULL: one row [2][C] of partial (sum y | sum y^2) per
 *                  workgroup, s2t_dwpool_stat_partials(B, Tout) rows, for s2t_bn_finalize.
 * s2t_dwpool_bwd : dx (rows t < Tout*r; the caller zero-fills a ragged tail), dw[c][k] += sum dy*x, db[c] += sum dy. */
int s2t_dwpool_stat_partials(int B, int Tout);
int s2t_dwpool_fwd(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int Tin, int C, int r,
                   float* stats, void* stream);
int s2t_dwpool_bwd(int dtype, const void* x, const float* w, const void* dy, void* dx, float* dw, float* db, int B, int Tin,
                   int C, int r, void* stream);

/* ---- Row-block kernels (csrc/rowblock.hip): a workgroup owns 64 complete rows of a [rows][256] bf16 activation -------
 * s2t_ffn_fused_fwd: one launch for the whole position-wise feed-forward block of a pre-LN layer,
 *     xn  = ln_gamma ? LayerNorm(x; ln_gamma, ln_beta, ln_eps) : x                      (modules/layer_norm.py:30-35)
 *     h   = drop_h(act(xn W1^T + b1))          z = the pre-activation                    (s2t_transformer_layer.py:55-66)
 *     y   = residual + alpha * drop_o(h W2^T + b2)            (the half-step residual of :258-265 / :311-317; bf16)
 *     y_ln = eln_gamma ? LayerNorm(y; eln_gamma, eln_beta, ln_eps), rows of padded frames (eln_lens / eln_T) zeroed
 *                                                             (final_norm, s2t_transformer_layer.py:318-320)
 * replacing F.layer_norm + two F.linear + activation + two dropouts + the residual add (+ F.layer_norm) of the reference.
 * The [rows][F] hidden activation stays on the chip; z / h / x_ln / ln_mean / ln_rstd are written only when given
 * (what the backward pass of the unfused kernels needs: dgrad through s2t_gemm with dact_z = z, wgrad operands h and x_ln,
 * s2t_layernorm_bwd statistics).  Dropout masks are those of s2t_gemm's epilogue for the same (seed, site, element):
 * element index row*F + f for drop_h, row*256 + n for drop_o.
 * Constraints: d == 256, F % 64 == 0, bf16 activations and weights (W1 [F][256], W2 [256][F], row-major as
 * nn.Linear stores them), fp32 biases / LayerNorm parameters / statistics, every pointer 16-byte aligned.
 * y may be NULL when y_ln is given (eval). */
typedef struct s2t_ffn_args {
  const void* x;          /* [M][256] bf16 */
  const float* ln_gamma;  /* NULL: x is used as it is */
  const float* ln_beta;
  float ln_eps;
  int32_t d;              /* must be 256 */
  const void* w1; const float* b1;
  const void* w2; const float* b2;
  const void* residual;   /* [M][256] bf16 or NULL */
  void* y;                /* [M][256] bf16 */
  const float* eln_gamma; const float* eln_beta; /* optional LayerNorm behind the block */
  void* y_ln;             /* [M][256] bf16, required iff eln_gamma */
  float* eln_mean; float* eln_rstd; /* optional [M] fp32 statistics of that LayerNorm */
  const int32_t* eln_lens; int32_t eln_T; /* optional padded-frame mask on y_ln */
  void* x_ln; float* ln_mean; float* ln_rstd; /* optional saves of the leading LayerNorm */
  void* z; void* h;       /* optional [M][F] bf16 saves */
  int32_t M, F;
  int32_t act;            /* S2T_ACT_NONE | RELU | SWISH */
  float alpha;
  float drop_h_p; uint32_t drop_h_site;
  float drop_o_p; uint32_t drop_o_site;
  const uint64_t* drop_seed;
  /* optional exchange workspace of at least s2t_ffn_pair_ws_bytes(M) bytes, ZERO before its first use and owned by one
   * stream: with it, row counts that would leave most CUs idle run two, four or eight workgroups per 128-row block, each on
   * its share of the hidden units, which swap fp32 partial rows through it (csrc/ffn_pc.hip).  The flag words sit in its
   * first 16 KiB whatever M is and the kernels leave them zero again: one workspace (sized for the largest M) serves every
   * row count a caller runs. */
  void* pair_ws; int64_t pair_ws_bytes;
  /* z_tiled_ok != 0: the caller accepts z in the TILED layout of the 128-row kernel (s2t_ffn_z_tiled tells whether this call
   * writes it so) and has allocated s2t_ffn_z_elems(M, F) elements for it.  Tiled z: the 16-byte piece of units
   * 64 cg + 16 s + 8 hh .. +7 of row 128 p + 32 wi + m sits at element ((((p * (F/64) + cg) * 4 + wi) * 4 + s) * 64 + 32 hh + m) * 8:
   * a lane of the kernel owns one row and eight consecutive units, so its store (and the backward kernel's load) of a
   * k-step is one contiguous 1 KiB per wave instead of 32-byte pieces of 32 rows.  z is private to the fused kernels
   * (only s2t_ffn_fused_bwd reads it: pass z_tiled there); h keeps the row-major layout the weight gradient reads. */
  int32_t z_tiled_ok;
} s2t_ffn_args;
int s2t_ffn_fused_fwd(const s2t_ffn_args* args, void* stream);
int64_t s2t_ffn_pair_ws_bytes(int32_t M);
/* Exchange health.  A workgroup whose partner's flag does not arrive within the spin limit gives up (it never happens on a
 * grid that is resident at once; it CAN when another stream's kernels hold CUs) and COUNTS the event in one 32-bit word of
 * the workspace, s2t_ffn_exchange_error_offset() bytes from its start (inside the first s2t_ffn_exchange_flag_bytes()
 * bytes, the same place for every row count).  That launch's results are invalid: the caller reads the word (the bundled
 * Trainer does after every update, kernels.ffn_exchange_poll), raises, and zero-fills the flag area before the next launch
 * (a late partner's flag would otherwise stay raised). */
int64_t s2t_ffn_exchange_error_offset(void);
int64_t s2t_ffn_exchange_flag_bytes(void);
/* Process-wide switches of the fused feed-forward kernels (first read from the environment: S2T_FFN_PC, S2T_FFN_PC_SPLIT);
 * a negative argument leaves that switch as it is.  pc_mask: bit 0 eval, 1 training forward, 2 backward on the 128-row
 * kernel; split_force: 0 automatic, 1 one workgroup per row block, 2 / 4 at most that many; fault (TEST HOOK): part 1 of
 * every row block never raises its exchange flag and the spin limit is short, so that the time-out path can be exercised.
 * Returns the settings in force (mask | split_force << 4 | fault << 12). */
int s2t_ffn_configure(int pc_mask, int split_force, int fault);
/* Compute units the fused feed-forward launches may count on being free AT ONCE (the parts of a row block wait for each other):
 * cus > 0 sets the budget, 0 restores the device's count, < 0 only queries; returns the budget in force.  The split (parts per
 * 128-row block) and s2t_ffn_pair_ws_bytes follow it.  The data-parallel wrapper lowers it by what its all-reduce kernels
 * occupy beside backward (legacy_distributed_data_parallel.py:76-160); S2T_FFN_CU_BUDGET presets it. */
int s2t_ffn_cu_budget(int cus);
/* the kernel symbol s2t_ffn_fused_fwd launches for these arguments, as a profiler prints it (buf: >= 96 bytes) */
int s2t_ffn_fused_describe(const s2t_ffn_args* args, char* buf, int32_t buf_bytes);
int s2t_ffn_z_tiled(const s2t_ffn_args* args);     /* 1: s2t_ffn_fused_fwd(args) writes z tiled */
int64_t s2t_ffn_z_elems(int32_t M, int32_t F);     /* bf16 elements of a z buffer that either layout fits */

/* s2t_ffn_fused_bwd: the input gradient of the same block's two products in one launch (what autograd derives from the
 * two F.linear, the activation and the hidden dropout of s2t_transformer_layer.py:55-66):
 *     dH  = dy W2                                  dy = gradient w.r.t. (h W2^T + b2), output dropout already undone
 *     dz  = alpha * drop_h(dH * act'(z))           same mask as the forward (element row*F + f); written: it is the
 *                                                  operand of the W1 / b1 weight gradients
 *     dxn = dz W1                                  gradient w.r.t. the block's (normalised) input
 * The products run on the forward kernel's schedule and therefore take the weights TRANSPOSED: w2t = W2^T [F][256],
 * w1t = W1^T [256][F] (s2t_transpose_bf16_batched keeps such copies current).  The [rows][F] dH never leaves the chip.
 * ln_x != NULL adds the backward of the block's leading LayerNorm (modules/layer_norm.py:30-35) on the fp32 dxn rows, with
 * s2t_layernorm_bwd's arithmetic and outputs:  dx = LayerNorm'(dxn; ln_x, ln_gamma, ln_mean, ln_rstd) + dres,  dx_drop =
 * dropout(dx) under (up_drop_p, up_drop_site) when given, and the partial sums of dgamma | dbeta added into
 * ln_ws [ln_replicas][2][256] (fold them with s2t_layernorm_fold); dxn is then optional and not written.
 * Constraints: as s2t_ffn_fused_fwd, and M * F * 2 < 2^32. */
typedef struct s2t_ffn_bwd_args {
  const void* dy;         /* [M][256] bf16 */
  const void* w2t;        /* [F][256] bf16 */
  const void* w1t;        /* [256][F] bf16 */
  const void* z;          /* [M][F] bf16, saved by the forward */
  void* dz;               /* [M][F] bf16 out */
  void* dxn;              /* [M][256] bf16 out (required unless ln_x is given) */
  int32_t d;              /* must be 256 */
  int32_t M, F;
  int32_t act;            /* S2T_ACT_NONE | RELU | SWISH */
  float alpha;
  float drop_h_p; uint32_t drop_h_site;
  const uint64_t* drop_seed;
  const void* ln_x;       /* [M][256] bf16 input of the leading LayerNorm, or NULL */
  const float* ln_gamma; const float* ln_mean; const float* ln_rstd;
  const void* dres;       /* [M][256] bf16 residual-branch gradient added to dx, or NULL */
  float* ln_ws; int32_t ln_replicas;
  void* dx;               /* [M][256] bf16 out */
  void* dx_drop;          /* optional [M][256] bf16 */
  float up_drop_p; uint32_t up_drop_site;
  /* end_y != NULL: the block ended in a LayerNorm (final_norm).  `dy` is then the gradient w.r.t. that LayerNorm's OUTPUT;
   * the prologue applies its backward (rows of padded frames, end_lens / end_T, carry no gradient): dres_out = gradient
   * w.r.t. the block output y, dy_out = dropout(dres_out) under (drop_o_p, drop_o_site) when given — the products' input,
   * also the operand of the W2 weight gradient — and the dgamma | dbeta partial sums go to end_ws. */
  const void* end_y; const float* end_gamma; const float* end_mean; const float* end_rstd;
  const int32_t* end_lens; int32_t end_T;
  float* end_ws; int32_t end_replicas;
  void* dres_out; void* dy_out;
  float drop_o_p; uint32_t drop_o_site;
  void* pair_ws; int64_t pair_ws_bytes; /* as in s2t_ffn_args */
  int32_t z_tiled;        /* z is in the tiled layout (s2t_ffn_args.z_tiled_ok): only the 128-row kernel reads it */
} s2t_ffn_bwd_args;
int s2t_ffn_fused_bwd(const s2t_ffn_bwd_args* args, void* stream);
int s2t_ffn_fused_bwd_describe(const s2t_ffn_bwd_args* args, char* buf, int32_t buf_bytes);

/* dst_i [cols_i][rows_i] = src_i [rows_i][cols_i]^T for n bf16 matrices in one launch; items_dev: device array;
 * tiles_dev: device array of n_tiles (matrix index, row tile, column tile) int32 triples, 64 x 64 tiles, covering every
 * matrix (one workgroup each). */
typedef struct s2t_transpose_item {
  const void* src; void* dst;
  int32_t rows, cols;
} s2t_transpose_item;
int s2t_transpose_bf16_batched(const s2t_transpose_item* items_dev, int n, const int32_t* tiles_dev, int n_tiles, void* stream);

/* s2t_rowblock_gemm: the K = 256 projections of an encoder layer on the same 64-row blocks, with the LayerNorm in front
 * folded in:   out[M][Nout] = epilogue( xn[M][256] W[N][256]^T ),   xn = ln_gamma ? LayerNorm(x) (rows of padded frames
 * zeroed when ln_lens is given: the conv-module input mask, convolution.py:86-88) : x.
 * Replaces F.layer_norm + F.linear / Conv1d(k=1) of modules/multihead_attention.py:239-263, espnet_multihead_attention.py:
 * 88-106, convolution.py:91,106 (+ GLU, :92).  Epilogue = s2t_gemm's, in its order: bias; act == S2T_ACT_GLU (out column c
 * pairs weight rows c and N/2 + c, Nout = N/2, optional pre-activation copy [M][N] value | gate); dropout on element
 * row*Nout + n; alpha; row_lens mask; residual.  x_ln / ln_mean / ln_rstd: optional saves of the LayerNorm for backward.
 * Constraints: d == 256, bf16 x / W / out / residual / preact, fp32 bias and LayerNorm parameters, 16-byte aligned pointers,
 * row strides multiples of 8 elements, N % 8 == 0 (GLU: N % 64 == 0), N <= 4096 (S2T_ERR_UNSUPPORTED beyond). */
typedef struct s2t_rowblock_args {
  const void* x;          /* [M][256] bf16 */
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  int32_t d;              /* must be 256 */
  const int32_t* ln_lens; int32_t ln_T; /* optional padded-frame mask on the LayerNorm output */
  void* x_ln; float* ln_mean; float* ln_rstd;
  const void* w;          /* [N][256] bf16 */
  const float* bias;      /* [N] fp32 or NULL */
  int32_t M, N;
  int32_t act;            /* S2T_ACT_NONE | S2T_ACT_GLU */
  void* preact; int64_t ldp;
  void* out; int64_t ldc;
  float alpha;
  const int32_t* row_lens; int32_t row_T;
  const void* residual; int64_t ldr;
  float drop_p; uint32_t drop_site; const uint64_t* drop_seed;
  /* instead of the LayerNorm (ln_gamma == NULL): x_in = pre_act(x * pre_scale[c] + pre_shift[c]) per column, rows masked by
   * (ln_lens, ln_T) set to zero — the BatchNorm affine + activation between the depthwise conv and pointwise conv 2
   * (convolution.py:102-106, s2t_bn_act_fwd's arithmetic); x_ln then receives x_in (the operand of pointwise conv 2's
   * weight gradient).  pre_act: S2T_ACT_* */
  const float* pre_scale; const float* pre_shift; int32_t pre_act;
  /* conv_w != NULL (needs pre_scale / pre_shift, no GLU, no dropout, conv_T >= 18): x is the GLU output G of the
   * convolution module and x_in = pre_act(dwconv15(G)[m, c] * pre_scale[c] + pre_shift[c]) — the depthwise convolution over
   * time (conv_w: fp32 [256][15], pad 7, zero padding per utterance of conv_T frames; modules/convolution.py:100-112 with
   * the eval-mode BatchNorm folded into the affine), rows masked by ln_lens / ln_T; x_ln is not written in this mode */
  const float* conv_w; int32_t conv_T;
  /* conv mode only, bn_mean != NULL: pre_scale / pre_shift hold the BatchNorm's gamma / beta and the affine is folded in the
   * kernel from the running statistics: scale = gamma * rsqrt(bn_var + bn_eps), shift = beta - bn_mean * scale */
  const float* bn_mean; const float* bn_var; float bn_eps;
} s2t_rowblock_args;
int s2t_rowblock_gemm(const s2t_rowblock_args* args, void* stream);
/* s2t_rowblock_chain: two projections of the same rows in ONE launch (round 5) — `first`: a plain N = 256 projection (no
 * LayerNorm / affine in front, no GLU; bias, dropout, row mask, residual as in s2t_rowblock_gemm): the attention output
 * projection of a layer (multihead_attention.py:420, espnet_multihead_attention.py:155 + the layer's residual,
 * s2t_transformer_layer.py:283-288); `second`: a projection of LayerNorm(first's output) without dropout and residual:
 * conv_norm + pointwise conv 1 + GLU (convolution.py:86-92).  second->x is ignored (the rows stay on the chip); every output
 * of both — first->out included — is written exactly as by two s2t_rowblock_gemm launches.  S2T_ERR_UNSUPPORTED outside
 * that shape (callers then launch the two separately). */
int s2t_rowblock_chain(const s2t_rowblock_args* first, const s2t_rowblock_args* second, void* stream);

/* s2t_rowblock_dgrad: the input gradient of a 256 -> K projection that sits BEHIND a LayerNorm, and that
 * LayerNorm's backward, in one launch on the same 64-row blocks (the autograd backward of F.layer_norm + F.linear /
 * Conv1d(k=1): fused QKV projection, K = 768, multihead_attention.py:239-263 / espnet_multihead_attention.py:88-106;
 * pointwise conv 1, K = 512, convolution.py:86-91):
 *     dxn = dy W            dy [M][K] bf16 (gradient w.r.t. the projection's output), wt = W^T [256][K] bf16
 *     dx  = LayerNorm'(dxn; ln_x, ln_gamma, ln_mean, ln_rstd) + dres      (rows t >= ln_lens[b] of each T block: dxn = 0,
 *           the LayerNorm's output was masked there); dx_drop = dropout(dx) under (up_drop_p, up_drop_site) when given;
 *           dgamma | dbeta partial sums into ln_ws [ln_replicas][2][256] (s2t_layernorm_fold)
 * with s2t_layernorm_bwd's arithmetic on the fp32 dxn rows.  ln_x == NULL: dxn is stored as it is.
 * Constraints: d == 256, K % 256 == 0, K <= 2048, bf16, 16-byte aligned pointers, (M + 64) * K * 2 < 2^32. */
typedef struct s2t_rowblock_dgrad_args {
  const void* dy; const void* wt;
  int32_t d;              /* must be 256 */
  int32_t M, K;
  void* dxn;              /* [M][256] bf16 out, required iff ln_x == NULL */
  const void* ln_x; const float* ln_gamma; const float* ln_mean; const float* ln_rstd;
  const int32_t* ln_lens; int32_t ln_T;
  const void* dres;
  float* ln_ws; int32_t ln_replicas;
  void* dx; void* dx_drop;
  float up_drop_p; uint32_t up_drop_site; const uint64_t* drop_seed;
} s2t_rowblock_dgrad_args;
int s2t_rowblock_dgrad(const s2t_rowblock_dgrad_args* args, void* stream);

/* ---- Gradient all-reduce over RCCL / xGMI (csrc/comm.hip) ------------------------------------------------
 * Replaces torch.distributed.all_reduce in LegacyDistributedDataParallel.all_reduce_grads
 * (distributed/legacy_distributed_data_parallel.py:107-120): one process-global communicator (one process per GPU),
 * created from a 128-byte RCCL unique id that rank 0 draws with s2t_comm_unique_id and hands to the other ranks over any
 * host channel (the Python side uses torch.distributed's store / gloo).  s2t_allreduce_bucket is an in-place,
 * stream-ordered all-reduce of `count` elements (S2T_F32 or S2T_BF16); average != 0 divides by the world size inside
 * the collective (the reference's pre-division, :107-110).  No thread watches the communicator, so the call can be
 * captured into a hipGraph and issued on a side stream beside backward.
 * RCCL is bound at run time; S2T_ERR_UNSUPPORTED = no librccl, no communicator, or a second s2t_comm_init.
 * RCCL errors come back as 10000 + ncclResult_t. */
int s2t_comm_unique_id(void* out128);
int s2t_comm_init(int rank, int world, const void* unique_id128);
int s2t_comm_world(void);
int s2t_allreduce_bucket(void* ptr, int64_t count, int dtype, int average, void* stream);
int s2t_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif
