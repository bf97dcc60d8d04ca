// Attention score -> probability kernels (forward and backward) for the GEMM-composed attention path.
//
// Reference semantics:
//   modules/multihead_attention.py:367-403   scores (+causal mask) ; key-pad -> -inf ; softmax in fp32
//   modules/espnet_multihead_attention.py:108-129,292-311,343-350
//       scores = (ac + rel_shift(bd)) / sqrt(dk) ; key-pad -> -inf ; clamp(+-1e8) ; softmax in fp32
//   rel_shift is indexed directly:  bd_shifted[i][j] = bd[i][Tq-1-i+j]   (SURVEY.md §7 "hard parts")
//
// One wavefront per (z, query-row); keys are spread over the 64 lanes, fp32 statistics via wave shuffles.
#include "common.h"

namespace {

template <typename TP, int NPL>
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(
    const float* __restrict__ S, int64_t ldS, const float* __restrict__ BD, int64_t ldBD, TP* __restrict__ P,
    int64_t ldP, int64_t rows_total, int H, int Tq, int Tk, float scale, const int32_t* __restrict__ key_lens,
    int causal, int clamp, TP* __restrict__ Pdrop, float drop_p, const uint64_t* __restrict__ drop_seed,
    uint32_t drop_site) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_total) return;
  const int64_t z = row / Tq;
  const int i = (int)(row % Tq);
  const int b = (int)(z / H);
  const int klen = key_lens ? min(key_lens[b], Tk) : Tk;
  const float* s = S + row * ldS;
  const float* bd = BD ? BD + row * ldBD + (Tq - 1 - i) : nullptr;
  float v[NPL];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int j = lane + 64 * k;
    float t = -INFINITY;
    if (j < Tk) {
      t = s[j];
      if (bd) t += bd[j];
      t *= scale;
      if (j >= klen || (causal && j > i)) t = -INFINITY;
      if (clamp) t = fminf(fmaxf(t, -1e8f), 1e8f);
    }
    v[k] = t;
    mx = fmaxf(mx, t);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const float e = (mx == -INFINITY) ? 0.f : __expf(v[k] - mx);
    v[k] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  TP* p = P + row * ldP;
  TP* pd = Pdrop ? Pdrop + row * ldP : nullptr;
  const uint64_t key = pd ? s2t_drop_key(drop_seed, drop_site) : 0ull;
  const uint32_t th = s2t_drop_thresh(drop_p);
  const float dinv = s2t_drop_scale(drop_p);
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int j = lane + 64 * k;
    if (j < ldP) {
      const float pv = j < Tk ? v[k] * inv : 0.f;
      st_from_f32<TP>(p + j, pv);
      if (pd) st_from_f32<TP>(pd + j, (j < Tk && s2t_rand_u32(key, (uint64_t)row * Tk + j) >= th) ? pv * dinv : 0.f);
    }
  }
}

// dS = P * (dP - sum_j P*dP) * scale ; optional un-shift into dBD[i][n], n = Tq-1-i+j (zero elsewhere)
template <typename TP, int NPL>
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const TP* __restrict__ P, int64_t ldP,
                                                               const float* __restrict__ dP, int64_t ldDP,
                                                               TP* __restrict__ dS, int64_t ldDS,
                                                               TP* __restrict__ dBD, int64_t ldDBD,
                                                               int64_t rows_total, int H, int Tq, int Tk, float scale,
                                                               float drop_p, const uint64_t* __restrict__ drop_seed,
                                                               uint32_t drop_site) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_total) return;
  const int i = (int)(row % Tq);
  const TP* p = P + row * ldP;
  const float* dp = dP + row * ldDP;
  float pv[NPL], dv[NPL];
  float dot = 0.f;
  const uint64_t key = drop_p > 0.f ? s2t_drop_key(drop_seed, drop_site) : 0ull;
  const uint32_t th = s2t_drop_thresh(drop_p);
  const float dinv = s2t_drop_scale(drop_p);
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int j = lane + 64 * k;
    pv[k] = j < Tk ? ld_as_f32<TP>(p + j) : 0.f;
    dv[k] = j < Tk ? dp[j] : 0.f;
    // dP arrives as the gradient of dropout(P): route it through the regenerated mask
    if (drop_p > 0.f && j < Tk) dv[k] = s2t_rand_u32(key, (uint64_t)row * Tk + j) >= th ? dv[k] * dinv : 0.f;
    dot += pv[k] * dv[k];
  }
  dot = wave_sum(dot);
  TP* ds = dS + row * ldDS;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int j = lane + 64 * k;
    const float g = pv[k] * (dv[k] - dot) * scale;
    if (j < ldDS) st_from_f32<TP>(ds + j, j < Tk ? g : 0.f);
    pv[k] = g;
  }
  if (dBD) {
    // row of width ldDBD: entries n in [Tq-1-i, Tq-1-i+Tk) carry dS[j], everything else is zero
    const int64_t z = row / Tq;
    const int64_t nb = rows_total / Tq / H;  // utterances
    const int64_t orow = ((z % H) * nb + z / H) * Tq + i;  // head-major row
    TP* o = dBD + orow * ldDBD;
    const int n0 = Tq - 1 - i;
    for (int n = lane; n < n0; n += 64) st_from_f32<TP>(o + n, 0.f);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int j = lane + 64 * k;
      if (j < Tk) st_from_f32<TP>(o + n0 + j, pv[k]);
    }
    for (int n = n0 + Tk + lane; n < ldDBD; n += 64) st_from_f32<TP>(o + n, 0.f);
  }
}

}  // namespace

#define DISPATCH_NPL(T_, KERN, ...)                                                            \
  do {                                                                                         \
    if (Tk <= 256) hipLaunchKernelGGL((KERN<T_, 4>), grid, block, 0, s, __VA_ARGS__);          \
    else if (Tk <= 1024) hipLaunchKernelGGL((KERN<T_, 16>), grid, block, 0, s, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERN<T_, 48>), grid, block, 0, s, __VA_ARGS__);                   \
  } while (0)

extern "C" int s2t_attn_softmax_fwd(int p_dtype, const float* S, int64_t ldS, const float* BD, int64_t ldBD, void* P,
                                    int64_t ldP, int Z, int H, int Tq, int Tk, float scale, const int32_t* key_lens,
                                    int causal, int clamp, void* Pdrop, float drop_p, const uint64_t* drop_seed,
                                    uint32_t drop_site, void* stream) {
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !Pdrop)) return S2T_ERR_ARG;
  if (drop_p == 0.f) Pdrop = nullptr;
  if (!S || !P || Z <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return S2T_ERR_ARG;
  if (Tk > 3072 || ldP < Tk || ldS < Tk) return S2T_ERR_UNSUPPORTED;
  if (BD && (Tq != Tk || ldBD < 2 * Tq - 1)) return S2T_ERR_ARG;
  const int64_t rows = (int64_t)Z * Tq;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (p_dtype == S2T_F32)
    DISPATCH_NPL(float, attn_softmax_fwd_kernel, S, ldS, BD, ldBD, (float*)P, ldP, rows, H, Tq, Tk, scale, key_lens,
                 causal, clamp, (float*)Pdrop, drop_p, drop_seed, drop_site);
  else if (p_dtype == S2T_BF16)
    DISPATCH_NPL(bf16_t, attn_softmax_fwd_kernel, S, ldS, BD, ldBD, (bf16_t*)P, ldP, rows, H, Tq, Tk, scale, key_lens,
                 causal, clamp, (bf16_t*)Pdrop, drop_p, drop_seed, drop_site);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_attn_softmax_bwd(int dtype, const void* P, int64_t ldP, const float* dP, int64_t ldDP, void* dS,
                                    int64_t ldDS, void* dBD, int64_t ldDBD, int Z, int H, int Tq, int Tk, float scale,
                                    float drop_p, const uint64_t* drop_seed, uint32_t drop_site, void* stream) {
  if (drop_p < 0.f || drop_p >= 1.f) return S2T_ERR_ARG;
  if (!P || !dP || !dS || Z <= 0 || H <= 0 || Z % H || Tq <= 0 || Tk <= 0) return S2T_ERR_ARG;
  if (Tk > 3072 || ldP < Tk || ldDP < Tk || ldDS < Tk) return S2T_ERR_UNSUPPORTED;
  if (dBD && (Tq != Tk || ldDBD < 2 * Tq - 1)) return S2T_ERR_ARG;
  const int64_t rows = (int64_t)Z * Tq;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    DISPATCH_NPL(float, attn_softmax_bwd_kernel, (const float*)P, ldP, dP, ldDP, (float*)dS, ldDS, (float*)dBD, ldDBD,
                 rows, H, Tq, Tk, scale, drop_p, drop_seed, drop_site);
  else if (dtype == S2T_BF16)
    DISPATCH_NPL(bf16_t, attn_softmax_bwd_kernel, (const bf16_t*)P, ldP, dP, ldDP, (bf16_t*)dS, ldDS, (bf16_t*)dBD,
                 ldDBD, rows, H, Tq, Tk, scale, drop_p, drop_seed, drop_site);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}
