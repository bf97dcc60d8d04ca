// CTC prefix scores for joint attention/CTC beam search.
//
// The reference rescores the beam on the HOST, one hypothesis at a time, with ESPnet's numpy scorer
// (fairseq/sequence_generator.py:255-388 -> espnet.nets.ctc_prefix_score.CTCPrefixScore, a third-party package that is not
// part of the reference tree; algorithm: Watanabe et al. 2017, "Hybrid CTC/attention architecture for end-to-end speech
// recognition", eq. 51-54).  Here every (hypothesis, candidate token) pair is one thread that walks the utterance's frames:
//
//   state of a prefix g: r[t][0] / r[t][1] = log-probability that frames 0..t emit g and end in its last label / in blank
//   candidate c:   phi[t]   = (c == last(g)) ? r[t][1] : logaddexp(r[t][0], r[t][1])
//                  r'[t][0] = logaddexp(r'[t-1][0], phi[t-1]) + x[t][c]
//                  r'[t][1] = logaddexp(r'[t-1][0], r'[t-1][1]) + x[t][blank]
//                  psi      = logsumexp_t (phi[t-1] + x[t][c])  (+ r'[start-1][0])
//   </s>:  psi = logaddexp(r[T-1][0], r[T-1][1]);   blank: psi = logzero (-1e10, the scorer's "minus infinity")
//
// fp32 throughout, like the scorer's float32 numpy arrays.
#include "common.h"

namespace {

constexpr float kLogZero = -1.0e10f;

__device__ __forceinline__ float logaddexp_(float a, float b) {
  const float m = fmaxf(a, b);
  return m + log1pf(__expf(-fabsf(a - b)));
}

__global__ __launch_bounds__(64) void ctc_prefix_kernel(const float* __restrict__ lp, int64_t ld, int T,
                                                        const int32_t* __restrict__ in_lens,
                                                        const int32_t* __restrict__ sent,
                                                        const float* __restrict__ r_prev,
                                                        const int64_t* __restrict__ last, int out_len,
                                                        const int64_t* __restrict__ cand, int R, int K, int blank, int eos,
                                                        float* __restrict__ psi, float* __restrict__ r_new) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= R * K) return;
  const int r = idx / K;
  const int b = sent[r];
  const int len = min(in_lens[b], T);
  const int64_t c = min(max(cand[idx], (int64_t)0), ld - 1);  // a bad id must not become a wild read
  const float* x = lp + (int64_t)b * T * ld;  // frame t of this utterance: x + t * ld
  const float* rp = r_prev + (int64_t)r * T * 2;
  float* rn = r_new ? r_new + (int64_t)idx * T * 2 : nullptr;
  const bool same = out_len > 0 && c == last[r];

  if (out_len > len || len <= 0) {  // more labels than frames: no path (the caller ignores such rows)
    psi[idx] = kLogZero;
    if (rn)
      for (int t = 0; t < T; ++t) rn[2 * t] = rn[2 * t + 1] = kLogZero;
    return;
  }
  const int start = max(out_len, 1);
  float rn0, rn1;  // r'[t-1]
  if (out_len == 0) {
    rn0 = x[c];
    rn1 = kLogZero;
  } else {
    rn0 = rn1 = kLogZero;
  }
  if (rn) {
    for (int t = 0; t < start - 1; ++t) rn[2 * t] = rn[2 * t + 1] = kLogZero;
    rn[2 * (start - 1)] = rn0;
    rn[2 * (start - 1) + 1] = rn1;
  }
  float lpsi = rn0;
  constexpr int CH = 8;
  for (int t0 = start; t0 < len; t0 += CH) {
    float xc[CH], xb[CH], p0[CH], p1[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {  // loads first: none depends on the recurrence
      const int t = min(t0 + i, len - 1);
      xc[i] = x[(int64_t)t * ld + c];
      xb[i] = x[(int64_t)t * ld + blank];
      p0[i] = rp[2 * (t - 1)];
      p1[i] = rp[2 * (t - 1) + 1];
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int t = t0 + i;
      if (t < len) {
        const float phi = same ? p1[i] : logaddexp_(p0[i], p1[i]);
        const float n0 = logaddexp_(rn0, phi) + xc[i];
        const float n1 = logaddexp_(rn0, rn1) + xb[i];
        lpsi = logaddexp_(lpsi, phi + xc[i]);
        rn0 = n0;
        rn1 = n1;
        if (rn) {
          rn[2 * t] = n0;
          rn[2 * t + 1] = n1;
        }
      }
    }
  }
  if (rn)
    for (int t = len; t < T; ++t) rn[2 * t] = rn[2 * t + 1] = kLogZero;
  if (c == eos) lpsi = logaddexp_(rp[2 * (len - 1)], rp[2 * (len - 1) + 1]);
  if (c == blank) lpsi = kLogZero;
  psi[idx] = lpsi;
}

// initial state (the empty prefix): r[t][0] = logzero, r[t][1] = sum_{u <= t} x[u][blank]
__global__ __launch_bounds__(64) void ctc_prefix_init_kernel(const float* __restrict__ lp, int64_t ld, int T,
                                                             const int32_t* __restrict__ in_lens,
                                                             const int32_t* __restrict__ sent, int R, int blank,
                                                             float* __restrict__ r0) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= R) return;
  const int b = sent[r];
  const int len = min(in_lens[b], T);
  const float* x = lp + (int64_t)b * T * ld;
  float* o = r0 + (int64_t)r * T * 2;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) {
    if (t < len) acc += x[(int64_t)t * ld + blank];
    o[2 * t] = kLogZero;
    o[2 * t + 1] = t < len ? acc : kLogZero;
  }
}

}  // namespace

extern "C" int s2t_ctc_prefix_init(const float* lp, int64_t ld, int T, const int32_t* in_lens, const int32_t* sent, int R,
                                   int blank, float* r0, void* stream) {
  if (!lp || !in_lens || !sent || !r0 || R <= 0 || T <= 0 || blank < 0 || blank >= ld) return S2T_ERR_ARG;
  hipLaunchKernelGGL(ctc_prefix_init_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, lp, ld, T, in_lens, sent,
                     R, blank, r0);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ctc_prefix_score(const float* lp, int64_t ld, int T, const int32_t* in_lens, const int32_t* sent,
                                    const float* r_prev, const int64_t* last, int out_len, const int64_t* cand, int R, int K,
                                    int blank, int eos, float* psi, float* r_new, void* stream) {
  if (!lp || !in_lens || !sent || !r_prev || !last || !cand || !psi || R <= 0 || K <= 0 || T <= 0 || out_len < 0 ||
      blank < 0 || blank >= ld || eos < 0 || eos >= ld)
    return S2T_ERR_ARG;
  const int n = R * K;
  hipLaunchKernelGGL(ctc_prefix_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, lp, ld, T, in_lens, sent,
                     r_prev, last, out_len, cand, R, K, blank, eos, psi, r_new);
  return S2T_LAUNCH_CHECK();
}
