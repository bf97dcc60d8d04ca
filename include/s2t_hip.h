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

int s2t_version(void);
/* number of compute units of the current device (for host-side grid heuristics) */
int s2t_device_cu_count(void);

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
 *   v = act(v); [dact_z: v *= act'(dact_z[m,n])]; v *= alpha; [residual: v += residual[m,n]];
 *   [row_lens: rows with (global_row % row_T) >= row_lens[global_row / row_T] are stored as 0];
 *   store (c_dtype).   global_row = z*M + m.
 * split_k > 1 (wgrad): K is split over blockIdx.y and alpha*acc is atomically added to an fp32 C
 *   (no other epilogue stage allowed).
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
} s2t_gemm_args;

int s2t_gemm(const s2t_gemm_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif
