// Vocabulary-wide row kernels: log-softmax statistics / arg-max (CTC greedy decode), label-smoothed
// cross-entropy (+gradient), CTC loss (alpha/beta in log space, +gradient).  All statistics in fp32,
// whatever the logits' storage type (reference contract: utils.py:470-481 log_softmax in fp32).
#include "common.h"

namespace {

struct MaxIdx {
  float v;
  int i;
};
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {
  // larger value wins; ties -> LOWER index (torch.topk(1)/argmax on CPU return the first maximal element)
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

// visit every element of a row with 8/16-byte vector loads when the row start is 16-byte aligned
template <typename T, typename F>
__device__ __forceinline__ void row_foreach(const T* __restrict__ row, int V, F&& f) {
  constexpr int EPV = 16 / (int)sizeof(T);
  if ((((uintptr_t)row) & 15) == 0) {
    const int nvec = V / EPV;
    for (int i = threadIdx.x; i < nvec; i += 256) {
      const uint4 t = *reinterpret_cast<const uint4*>(row + (int64_t)i * EPV);
      const uint32_t w[4] = {t.x, t.y, t.z, t.w};
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f(i * 8 + 2 * q, __uint_as_float(w[q] << 16));
          f(i * 8 + 2 * q + 1, __uint_as_float(w[q] & 0xffff0000u));
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) f(i * 4 + q, __uint_as_float(w[q]));
      }
    }
    for (int c = nvec * EPV + threadIdx.x; c < V; c += 256) f(c, ld_as_f32<T>(row + c));
  } else {
    for (int c = threadIdx.x; c < V; c += 256) f(c, ld_as_f32<T>(row + c));
  }
}

// one workgroup per row: max, argmax (first), logsumexp — ONE pass over the row: every thread keeps a running (max, sum of
// exp relative to that max) and rescales the sum when the max moves, once per 16-byte vector (one extra exp per 8 values,
// no branch); partial (max, sum) pairs merge the same way across lanes and waves.  (The two-pass form read every row twice
// and paid two workgroup reductions: 113 us for 16 000 x 10 000 bf16 logits, 2.8 TB/s.)
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  const float ref = (M == -INFINITY) ? 0.f : M;  // all -inf so far: exp(-inf - 0) = 0, never exp(nan)
  s = s * __expf(m - ref) + s2 * __expf(m2 - ref);
  m = M;
}
// ARG = false: the caller wants the logsumexp only (CTC loss in training) — the arg-max bookkeeping (two compares and two
// selects per element) is half of the vector work of this kernel
template <typename T, bool ARG = true>
__device__ __forceinline__ void row_stats(const T* __restrict__ row, int V, float& mx, int& arg, float& lse) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float ss[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  MaxIdx m{-INFINITY, 0x7fffffff};
  float rm = -INFINITY, rs = 0.f;  // running max of the sum's reference, sum of exp(x - rm)
  constexpr int EPV = 16 / (int)sizeof(T);
  auto one = [&](int c, float x) __attribute__((always_inline)) {
    if constexpr (ARG) m = better(m, MaxIdx{x, c});
    lse_merge(rm, rs, x, 1.f);
  };
  if ((((uintptr_t)row) & 15) == 0) {
    const int nvec = V / EPV;
    constexpr int NB = 4;  // vectors in flight per thread (one load per iteration left the kernel waiting on memory)
    for (int i0 = threadIdx.x; i0 < nvec; i0 += 256 * NB) {
      uint4 tv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) tv[u] = *reinterpret_cast<const uint4*>(row + (int64_t)min(i0 + 256 * u, nvec - 1) * EPV);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + 256 * u;
        if (i < nvec) {
          const uint32_t wd[4] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
          float x[EPV];
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x[2 * q] = __uint_as_float(wd[q] << 16);
              x[2 * q + 1] = __uint_as_float(wd[q] & 0xffff0000u);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = __uint_as_float(wd[q]);
          }
          float vm = x[0];
#pragma unroll
          for (int e = 0; e < EPV; ++e) {
            if constexpr (ARG) m = better(m, MaxIdx{x[e], i * EPV + e});
            vm = fmaxf(vm, x[e]);
          }
          const float M = fmaxf(rm, vm);
          const float ref = (M == -INFINITY) ? 0.f : M;
          float acc = rs * __expf(rm - ref);
#pragma unroll
          for (int e = 0; e < EPV; ++e) acc += __expf(x[e] - ref);
          rs = acc;
          rm = M;
        }
      }
    }
    for (int c = nvec * EPV + threadIdx.x; c < V; c += 256) one(c, ld_as_f32<T>(row + c));
  } else {
    for (int c = threadIdx.x; c < V; c += 256) one(c, ld_as_f32<T>(row + c));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    if constexpr (ARG) {
      MaxIdx t{__shfl_xor(m.v, o, 64), __shfl_xor(m.i, o, 64)};
      m = better(m, t);
    }
    lse_merge(rm, rs, __shfl_xor(rm, o, 64), __shfl_xor(rs, o, 64));
  }
  if constexpr (!ARG) m = MaxIdx{rm, 0};  // the running max of the sum is the row's maximum of this wave
  if (lane == 0) {
    sv[w] = m.v;
    si[w] = m.i;
    ss[w] = rs;
  }
  __syncthreads();
  // (every wave's running max equals its arg-max value: sv doubles as the sums' reference)
  m = MaxIdx{sv[0], si[0]};
  float s = ss[0], sm = sv[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    m = better(m, MaxIdx{sv[k], si[k]});
    lse_merge(sm, s, sv[k], ss[k]);
  }
  mx = m.v;
  arg = m.i;
  lse = m.v + __logf(s);
  __syncthreads();
}

// write 8 (bf16) / 4 (f32) consecutive outputs of a row at once when aligned: out[c] = g(c, x[c])
template <typename T, typename G>
__device__ __forceinline__ void row_map(const T* __restrict__ row, T* __restrict__ out, int V, G&& g) {
  constexpr int EPV = 16 / (int)sizeof(T);
  if (((((uintptr_t)row) | ((uintptr_t)out)) & 15) == 0) {
    const int nvec = V / EPV;
    for (int i = threadIdx.x; i < nvec; i += 256) {
      const uint4 t = *reinterpret_cast<const uint4*>(row + (int64_t)i * EPV);
      const uint32_t w[4] = {t.x, t.y, t.z, t.w};
      uint32_t o[4];
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = g(i * 8 + 2 * q, __uint_as_float(w[q] << 16));
          const float b = g(i * 8 + 2 * q + 1, __uint_as_float(w[q] & 0xffff0000u));
          o[q] = bf16pack(a, b);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = __float_as_uint(g(i * 4 + q, __uint_as_float(w[q])));
      }
      *reinterpret_cast<uint4*>(out + (int64_t)i * EPV) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    for (int c = nvec * EPV + threadIdx.x; c < V; c += 256) st_from_f32<T>(out + c, g(c, ld_as_f32<T>(row + c)));
  } else {
    for (int c = threadIdx.x; c < V; c += 256) st_from_f32<T>(out + c, g(c, ld_as_f32<T>(row + c)));
  }
}

// ---- CTC greedy, stage 1: per frame arg-max + its log-probability (s2t_ctc.py:312-328) ----
constexpr int ARGMAX_ROWS = 8;
template <typename T, bool ARG = true>
__global__ __launch_bounds__(256) void argmax_lse_kernel(const T* __restrict__ logits, int64_t ld, int V, int64_t rows,
                                                         int32_t* __restrict__ idx, float* __restrict__ top_lp,
                                                         float* __restrict__ lse_out, const int32_t* __restrict__ live) {
  if (live) rows = min(rows, (int64_t)live[0]);  // packed batch: only the live rows
  // ARGMAX_ROWS rows per workgroup: with one row each, 16 000 workgroups of a few microseconds are bound by the dispatch
  // rate (~170 workgroups per microsecond), not by the 320 MB they read
  for (int64_t row = (int64_t)blockIdx.x * ARGMAX_ROWS; row < min(rows, ((int64_t)blockIdx.x + 1) * ARGMAX_ROWS); ++row) {
    float mx, lse;
    int arg;
    row_stats<T, ARG>(logits + row * ld, V, mx, arg, lse);
    if (threadIdx.x == 0) {
      if (idx) idx[row] = arg;
      if (top_lp) top_lp[row] = mx - lse;
      if (lse_out) lse_out[row] = lse;
    }
  }
}

// ---- CTC greedy, stage 2: pad->blank, unique_consecutive, drop blank (s2t_ctc.py:329-347); one WG / utterance ----
__global__ __launch_bounds__(1024) void ctc_collapse_kernel(const int32_t* __restrict__ idx,
                                                            const float* __restrict__ top_lp,
                                                            const int32_t* __restrict__ lens,
                                                            const int32_t* __restrict__ cu, int T, int blank,
                                                            int64_t* __restrict__ out_tokens,
                                                            int32_t* __restrict__ out_lens,
                                                            float* __restrict__ out_scores) {
  __shared__ int scan[1024];
  __shared__ float sred[16];
  __shared__ int carry;
  const int b = blockIdx.x;
  const int len = lens[b];
  const int64_t r0 = s2t_utt_row0(cu, b, T);  // first row of the utterance in idx / top_lp (frames >= len are not read)
  if (threadIdx.x == 0) carry = 0;
  float score = 0.f;
  __syncthreads();
  for (int base = 0; base < T; base += 1024) {
    const int t = base + threadIdx.x;
    int cur = blank, prev = -1;
    if (t < T) {
      // packed batch: the rows behind an utterance's frames belong to the next one — the padded frames they stand for are not
      // stored.  Tokens: the reference maps padded frames to the blank (s2t_ctc.py:324-326), so nothing is lost.  SCORE: the
      // reference adds the top-1 log-probability of every frame whose UNMASKED arg-max is not the blank, padded frames included
      // (:327-329; their logits are LayerNorm bias -> projection, not zero rows) — a packed batch cannot, so its score lacks that
      // input-independent term (INTEGRATION.md "Hypothesis scores"; CTCDecoder.exact_scores takes the padded layout instead)
      const bool have = !cu || t < len;
      const int raw = have ? idx[r0 + t] : blank;
      cur = t < len ? raw : blank;
      if (t > 0) {
        const int rawp = (!cu || t - 1 < len) ? idx[r0 + t - 1] : blank;
        prev = (t - 1) < len ? rawp : blank;
      }
      // the score uses the UNMASKED arg-max (s2t_ctc.py:327-328)
      if (have && raw != blank) score += top_lp[r0 + t];
    }
    const int keep = (t < T && cur != blank && cur != prev) ? 1 : 0;
    scan[threadIdx.x] = keep;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int v = threadIdx.x >= o ? scan[threadIdx.x - o] : 0;
      __syncthreads();
      scan[threadIdx.x] += v;
      __syncthreads();
    }
    const int pos = carry + scan[threadIdx.x] - keep;
    if (keep) out_tokens[(int64_t)b * T + pos] = cur;
    __syncthreads();
    if (threadIdx.x == 1023) carry += scan[1023];
    __syncthreads();
  }
  score = wave_sum(score);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = score;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += sred[k];
    out_scores[b] = -s;
    out_lens[b] = carry;
  }
}

// ---- label-smoothed cross entropy, forward + unit gradient (label_smoothed_cross_entropy.py:42-60) ----
// per row r (token): lp = log_softmax(x); nll = -lp[y]; smooth = -sum(lp)
//   loss += (1-eps-eps_i)*nll + eps_i*smooth  with eps_i = eps/(V-1);  pad rows contribute nothing.
//   d loss / d x[c] = (1-eps-eps_i)*(p[c] - [c==y]) + eps_i*(V*p[c] - 1)
// per row: part[row][0..3] = loss, nll, correct, counted (pad rows: zeros); ls_ce_fold_kernel adds the rows into sums in a
// fixed order.  (Every row's workgroup adding into the same four floats with atomics took 160 us for 2 624 rows: same-address
// float atomics are executed one after the other at the memory side.)
template <typename T>
__global__ __launch_bounds__(256) void ls_ce_kernel(const T* __restrict__ logits, int64_t ld, int V,
                                                    const int64_t* __restrict__ target, int64_t pad_idx, float eps,
                                                    T* __restrict__ dlogits, int64_t ldd, float* __restrict__ part,
                                                    const int32_t* __restrict__ live) {
  __shared__ float ssum[4];
  const int64_t row = blockIdx.x;
  if (live && row >= live[0]) {  // packed batch: no such row (its partial terms are zero, its gradient row is never read)
    if (threadIdx.x < 4) part[row * 4 + threadIdx.x] = 0.f;
    return;
  }
  const int64_t y = target[row];
  const T* x = logits + row * ld;
  T* dx = dlogits ? dlogits + row * ldd : nullptr;
  if (y == pad_idx) {
    if (dx)
      for (int c = threadIdx.x; c < V; c += 256) st_from_f32<T>(dx + c, 0.f);
    if (threadIdx.x < 4) part[row * 4 + threadIdx.x] = 0.f;
    return;
  }
  float mx, lse;
  int arg;
  row_stats<T>(x, V, mx, arg, lse);
  const float eps_i = eps / (V - 1);
  const float wn = 1.f - eps - eps_i;
  float sx = 0.f;
  if (dx) {
    const int yi = (int)y;
    row_map<T>(x, dx, V, [&](int c, float xv) {
      sx += xv;
      const float p = __expf(xv - lse);
      return wn * (p - (c == yi ? 1.f : 0.f)) + eps_i * (V * p - 1.f);
    });
  } else {
    row_foreach<T>(x, V, [&](int c, float xv) { sx += xv; });
  }
  sx = wave_sum(sx);
  if ((threadIdx.x & 63) == 0) ssum[threadIdx.x >> 6] = sx;
  __syncthreads();
  if (threadIdx.x == 0) {
    sx = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    const float nll = lse - ld_as_f32<T>(x + y);
    const float smooth = V * lse - sx;
    part[row * 4 + 0] = wn * nll + eps_i * smooth;
    part[row * 4 + 1] = nll;
    part[row * 4 + 2] = arg == (int)y ? 1.f : 0.f;
    part[row * 4 + 3] = 1.f;
  }
}

// sums[k] += sum_rows part[row][k]: one workgroup, thread (g = tid / 4, k = tid % 4) adds rows g, g + 256, ... in order,
// then a fixed tree over the 256 groups
__global__ __launch_bounds__(1024) void ls_ce_fold_kernel(const float* __restrict__ part, int64_t rows,
                                                          float* __restrict__ sums) {
  __shared__ float red[256][4];
  const int k = threadIdx.x & 3, g = threadIdx.x >> 2;
  float acc = 0.f;
  int64_t r = g;
  for (; r + 768 < rows; r += 1024) {
    const float v0 = part[r * 4 + k], v1 = part[(r + 256) * 4 + k], v2 = part[(r + 512) * 4 + k],
                v3 = part[(r + 768) * 4 + k];
    acc = (((acc + v0) + v1) + v2) + v3;
  }
  for (; r < rows; r += 256) acc += part[r * 4 + k];
  red[g][k] = acc;
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if (g < half) red[g][k] += red[g + half][k];
    __syncthreads();
  }
  if (g == 0) sums[k] += red[0][k];
}

// ---- CTC loss (torch.nn.CTCLoss(blank, reduction='none', zero_infinity=True), criterions/ctc.py:243-245,467-472) ----
// logits are batch-major [B][T][V] (row = b*T + t); log-probs are formed on the fly as x - lse[row].
// One workgroup per utterance, one thread per extended-label state s in [0, 2S+1) (S <= 511).
__device__ __forceinline__ float lse2(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b);
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) { return lse2(lse2(a, b), c); }

// workgroup barrier that orders LDS traffic only (no s_waitcnt vmcnt: outstanding global stores / prefetches keep flying)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__global__ __launch_bounds__(1024) void ctc_alpha_beta_kernel(const T* __restrict__ logits, int64_t ld, int V, int T_,
                                                              const float* __restrict__ lse,
                                                              const int64_t* __restrict__ targets, int ldt,
                                                              const int32_t* __restrict__ tgt_lens,
                                                              const int32_t* __restrict__ in_lens, int blank,
                                                              float* __restrict__ alpha, float* __restrict__ beta,
                                                              int Lmax, float* __restrict__ nll_out,
                                                              const int64_t* __restrict__ force_emits,
                                                              int32_t* __restrict__ paths, const int32_t* __restrict__ cu) {
  // force_emits (imputer loss, torch_imputer/imputer.cu:114-152): fe[b,t] >= 0 pins frame t to state fe, every other
  // state gets -inf.  paths != NULL (best alignment, torch_imputer/best_alignment.cu:57-201): max-product recursion
  // with back-pointers instead of log-sum-exp; beta is not computed.
  extern __shared__ float sh[];  // [2][Lmax+2] ping-pong
  constexpr int CH = 16;
  const int b = blockIdx.x;
  const int S = tgt_lens[b];
  const int L = 2 * S + 1;
  const int Tb = min(in_lens[b], T_);
  const int s = threadIdx.x;
  const bool act = s < L;
  int lab = blank;
  bool skip = false;
  if (act && (s & 1)) {
    lab = (int)targets[(int64_t)b * ldt + (s >> 1)];
    if (s >= 3) skip = lab != (int)targets[(int64_t)b * ldt + (s >> 1) - 1];
  }
  bool skipn = false;  // transition s -> s+2 allowed (for beta)
  if (act && (s & 1) && s + 2 < L) skipn = lab != (int)targets[(int64_t)b * ldt + (s >> 1) + 1];
  float* bufA = sh;
  float* bufB = sh + (Lmax + 2);
  const int64_t row0 = s2t_utt_row0(cu, b, T_);  // logits / lse rows of the utterance (alpha, beta, paths: padded strides)
  float* al = alpha + (int64_t)b * T_ * Lmax;
  float* be = beta + (int64_t)b * T_ * Lmax;
  if (Tb <= 0) {
    if (s == 0) nll_out[b] = 0.f;
    return;
  }
  // alpha and beta are independent recursions: blockIdx.y = 0 runs alpha (+ nll), 1 runs beta
  const bool do_alpha = blockIdx.y == 0, do_beta = (gridDim.y == 1 || blockIdx.y == 1) && beta != nullptr;
  // ---- alpha
  if (do_alpha) {
    float a = -INFINITY;
    if (act && s < 2) a = ld_as_f32<T>(logits + row0 * ld + lab) - lse[row0];
    if (force_emits) {
      const int64_t fe = force_emits[(int64_t)b * T_];
      if (fe > -1 && fe != s) a = -INFINITY;
    }
    if (act) {
      al[s] = a;
      bufA[s] = a;
      if (paths) paths[(int64_t)b * T_ * Lmax + s] = s;
    }
    __syncthreads();
    float* cur = bufA;
    float* nxt = bufB;
    // the emission log-probabilities do not depend on the recursion: fetch CH frames at a time (CH independent loads
    // in flight) so that one memory round trip is paid per chunk instead of per frame
    for (int t0 = 1; t0 < Tb; t0 += CH) {
      float lpq[CH];
      int64_t feq[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int t = min(t0 + i, Tb - 1);
        lpq[i] = act ? ld_as_f32<T>(logits + (row0 + t) * ld + lab) - lse[row0 + t] : 0.f;
        feq[i] = force_emits ? force_emits[(int64_t)b * T_ + t] : -1;
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int t = t0 + i;
        if (t >= Tb) break;  // workgroup-uniform
        float v = -INFINITY;
        if (act) {
          const float a0 = cur[s];
          const float a1 = s >= 1 ? cur[s - 1] : -INFINITY;
          const float a2 = (s >= 2 && skip) ? cur[s - 2] : -INFINITY;
          const float lp = lpq[i];
          if (paths) {
            // Viterbi: first maximum in the order s, s-1, s-2 (strict > to move on), best_alignment.cu:141-160
            float m = a0;
            int arg = s;
            if (a1 > m) { m = a1; arg = s - 1; }
            if (a2 > m) { m = a2; arg = s - 2; }
            v = m + lp;
            paths[((int64_t)b * T_ + t) * Lmax + s] = arg;
          } else {
            const float m = lse3(a0, a1, a2);
            v = m == -INFINITY ? -INFINITY : m + lp;
          }
          if (feq[i] > -1 && feq[i] != s) v = -INFINITY;
          nxt[s] = v;
          al[(int64_t)t * Lmax + s] = v;
        }
        lds_barrier();  // LDS hand-over only: the alpha/beta stores to HBM drain in the background
        float* tmp = cur;
        cur = nxt;
        nxt = tmp;
      }
    }
    if (s == 0) {
      const float l1 = cur[L - 1];
      const float l2 = L >= 2 ? cur[L - 2] : -INFINITY;
      const float ll = lse2(l1, l2);
      nll_out[b] = -ll;  // +inf when no alignment exists; zero_infinity is applied by the gradient kernel / host
    }
    __syncthreads();
  }
  // ---- beta
  if (do_beta) {
    float v = -INFINITY;
    if (act && s >= L - 2) v = ld_as_f32<T>(logits + (row0 + Tb - 1) * ld + lab) - lse[row0 + Tb - 1];
    if (force_emits) {
      const int64_t fe = force_emits[(int64_t)b * T_ + Tb - 1];
      if (fe > -1 && fe != s) v = -INFINITY;
    }
    if (act) {
      be[(int64_t)(Tb - 1) * Lmax + s] = v;
      bufA[s] = v;
    }
    __syncthreads();
    float* cur = bufA;
    float* nxt = bufB;
    for (int t0 = Tb - 2; t0 >= 0; t0 -= CH) {
      float lpq[CH];
      int64_t feq[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int t = max(t0 - i, 0);
        lpq[i] = act ? ld_as_f32<T>(logits + (row0 + t) * ld + lab) - lse[row0 + t] : 0.f;
        feq[i] = force_emits ? force_emits[(int64_t)b * T_ + t] : -1;
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int t = t0 - i;
        if (t < 0) break;  // workgroup-uniform
        if (act) {
          const float b0 = cur[s];
          const float b1 = s + 1 < L ? cur[s + 1] : -INFINITY;
          const float b2 = (s + 2 < L && skipn) ? cur[s + 2] : -INFINITY;
          const float m = lse3(b0, b1, b2);
          float r = m == -INFINITY ? -INFINITY : m + lpq[i];
          if (feq[i] > -1 && feq[i] != s) r = -INFINITY;
          nxt[s] = r;
          be[(int64_t)t * Lmax + s] = r;
        }
        lds_barrier();  // LDS hand-over only: the alpha/beta stores to HBM drain in the background
        float* tmp = cur;
        cur = nxt;
        nxt = tmp;
      }
    }
  }
}

// Single-wave variant for 2S+1 <= 127 states, log-sum-exp recursion only (no force_emits / paths): lane l owns the
// blank state 2l and the label state 2l+1 and the neighbour's values arrive by a DPP wave shift, so a frame step has
// no LDS round trip and no barrier.  The step time is the latency of the dependent chain, so the chain is kept short:
// values live in the log2 domain (raw v_exp_f32 / v_log_f32, converted to natural logs only on the way to memory), the
// three-way sum is taken around one max3, "no predecessor" is -inf flowing through exp2 -> 0 instead of a branch, and
// inactive states are kept at -inf by an emission term of -inf.
__device__ __forceinline__ float wave_from_prev(float v) {  // lane l <- lane l-1, lane 0 <- -inf
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xff800000, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_from_next(float v) {  // lane l <- lane l+1, lane 63 <- -inf
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xff800000, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_value(float v, int l) {  // v of lane l (wave-uniform l), whatever the exec mask
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float l2se3(float a, float b, float c) {  // log2(2^a + 2^b + 2^c), -inf when all are -inf
  const float m = fmaxf(__builtin_fmaxf(__builtin_fmaxf(a, b), c), -3.0e38f);
  return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m));
}
__device__ __forceinline__ float l2se2(float a, float b) {
  const float m = fmaxf(__builtin_fmaxf(a, b), -3.0e38f);
  return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m));
}

template <typename T>
__global__ __launch_bounds__(64) void ctc_alpha_beta_wave_kernel(const T* __restrict__ logits, int64_t ld, int T_,
                                                                 const float* __restrict__ lse,
                                                                 const int64_t* __restrict__ targets, int ldt,
                                                                 const int32_t* __restrict__ tgt_lens,
                                                                 const int32_t* __restrict__ in_lens, int blank,
                                                                 float* __restrict__ alpha, float* __restrict__ beta,
                                                                 int Lmax, float* __restrict__ nll_out,
                                                                 const int32_t* __restrict__ cu) {
  constexpr int CH = 16;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const int b = blockIdx.x;
  const int S = tgt_lens[b];
  const int L = 2 * S + 1;
  const int Tb = min(in_lens[b], T_);
  const int lane = threadIdx.x;
  const int s0 = 2 * lane, s1 = s0 + 1;
  const bool act0 = s0 < L, act1 = s1 < L;
  const int64_t* tg = targets + (int64_t)b * ldt;
  const int lab = act1 ? (int)tg[lane] : blank;
  const bool skip = act1 && lane >= 1 && lab != (int)tg[lane - 1];
  const bool skipn = act1 && s1 + 2 < L && lab != (int)tg[lane + 1];
  const int64_t row0 = s2t_utt_row0(cu, b, T_);  // logits / lse rows of the utterance (alpha, beta: padded strides)
  if (Tb <= 0) {
    if (lane == 0 && blockIdx.y == 0) nll_out[b] = 0.f;
    return;
  }
  const T* lg = logits + row0 * ld;
  const float* ls = lse + row0;
  // Emission terms of CH frames at a time, fetched one chunk ahead of the recursion.  Every lane loads (lab is the
  // blank for a lane without a label state) so that the loads go out back to back; the per-frame log-sum-exp comes in
  // as one vector load, lane i holding frame i of the chunk.
  struct Chunk {
    float x0[CH], x1[CH], z;
  };
  auto fetch = [&](Chunk& c, int t0, int dir) {
    c.z = ls[min(max(t0 + dir * (lane & (CH - 1)), 0), Tb - 1)];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int t = min(max(t0 + dir * i, 0), Tb - 1);
      c.x0[i] = ld_as_f32<T>(lg + t * ld + blank);
      c.x1[i] = ld_as_f32<T>(lg + t * ld + lab);
    }
  };
  // log2-domain emission terms; -inf for a state this utterance does not have
  auto emit0 = [&](const Chunk& c, int i) {
    const float v = (c.x0[i] - lane_value(c.z, i)) * LOG2E;
    return act0 ? v : -INFINITY;
  };
  auto emit1 = [&](const Chunk& c, int i) {
    const float v = (c.x1[i] - lane_value(c.z, i)) * LOG2E;
    return act1 ? v : -INFINITY;
  };
  Chunk cur, nxt;
  if (blockIdx.y == 0) {
    float* al = alpha + (int64_t)b * T_ * Lmax;
    const __amdgpu_buffer_rsrc_t asrd = __builtin_amdgcn_make_buffer_rsrc(al, 0, (int)((int64_t)T_ * Lmax * 4), 0x00020000);
    fetch(cur, 0, 1);
    float a0 = lane == 0 ? emit0(cur, 0) : -INFINITY;
    float a1 = lane == 0 ? emit1(cur, 0) : -INFINITY;
    if (act0) al[s0] = a0 * LN2;
    if (act1) al[s1] = a1 * LN2;
    fetch(cur, 1, 1);
    for (int t0 = 1; t0 < Tb; t0 += CH) {
      fetch(nxt, t0 + CH, 1);
      float e0[CH], e1[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        e0[i] = emit0(cur, i);
        e1[i] = emit1(cur, i);
      }
      // (a wave-uniform `if` per step, not a `break`: the loop then unrolls completely and e0 / e1 stay in fixed registers;
      // the two stores of a step are bounds-checked buffer stores — a lane without the state gets an out-of-range offset —
      // instead of two exec-mask branches)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (t0 + i < Tb) {
          const float p1 = wave_from_prev(a1);  // alpha[t-1][2l-1]
          const float n0 = l2se2(a0, p1) + e0[i];
          const float n1 = l2se3(a1, a0, skip ? p1 : -INFINITY) + e1[i];
          a0 = n0;
          a1 = n1;
          const uint32_t ro = (uint32_t)(t0 + i) * (uint32_t)(Lmax * 4);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a0 * LN2), asrd, act0 ? ro + 4u * s0 : 0xFFFFFFFFu, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a1 * LN2), asrd, act1 ? ro + 4u * s1 : 0xFFFFFFFFu, 0, 0);
        }
      }
      cur = nxt;
    }
    const float l1 = lane_value(a0, S);                                   // state L-1 = 2S
    const float l2 = L >= 2 ? lane_value(a1, max(S - 1, 0)) : -INFINITY;  // state L-2 = 2(S-1)+1
    if (lane == 0) nll_out[b] = -(l2se2(l1, l2) * LN2);
  } else {
    float* be = beta + (int64_t)b * T_ * Lmax;
    const __amdgpu_buffer_rsrc_t bsrd = __builtin_amdgcn_make_buffer_rsrc(be, 0, (int)((int64_t)T_ * Lmax * 4), 0x00020000);
    fetch(cur, Tb - 1, -1);
    float b0 = s0 >= L - 2 ? emit0(cur, 0) : -INFINITY;
    float b1 = s1 >= L - 2 ? emit1(cur, 0) : -INFINITY;
    if (act0) be[(int64_t)(Tb - 1) * Lmax + s0] = b0 * LN2;
    if (act1) be[(int64_t)(Tb - 1) * Lmax + s1] = b1 * LN2;
    fetch(cur, Tb - 2, -1);
    for (int t0 = Tb - 2; t0 >= 0; t0 -= CH) {
      fetch(nxt, t0 - CH, -1);
      float e0[CH], e1[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        e0[i] = emit0(cur, i);
        e1[i] = emit1(cur, i);
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (t0 - i >= 0) {
          const float nb0 = wave_from_next(b0);  // beta[t+1][2l+2]  (-inf past the last state)
          const float nb1 = wave_from_next(b1);  // beta[t+1][2l+3]
          const float n0 = l2se2(b0, b1) + e0[i];
          const float n1 = l2se3(b1, nb0, skipn ? nb1 : -INFINITY) + e1[i];
          b0 = n0;
          b1 = n1;
          const uint32_t ro = (uint32_t)(t0 - i) * (uint32_t)(Lmax * 4);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, b0 * LN2), bsrd, act0 ? ro + 4u * s0 : 0xFFFFFFFFu, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, b1 * LN2), bsrd, act1 ? ro + 4u * s1 : 0xFFFFFFFFu, 0, 0);
        }
      }
      cur = nxt;
    }
  }
}

// grad[b,t,c] = gscale * ( p[c] - sum_{s: ext[s]==c} exp(alpha+beta + nll - lp[c]) )   for t < len, finite nll; else 0
// one workgroup per frame row; the <= 2S+1 states are merged per label in LDS.
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const T* __restrict__ logits, int64_t ld, int V, int T_,
                                                       const float* __restrict__ lse,
                                                       const int64_t* __restrict__ targets, int ldt,
                                                       const int32_t* __restrict__ tgt_lens,
                                                       const int32_t* __restrict__ in_lens, int blank,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       int Lmax, const float* __restrict__ nll, float gscale_host,
                                                       const float* __restrict__ gscale_dev, T* __restrict__ grad,
                                                       int64_t ldg, int wrt_logprobs, const int32_t* __restrict__ cu) {
  // the upstream gradient of the summed loss usually lives on the device (a 0-dim autograd tensor): multiply here
  // instead of in a separate pass over the [B*T, V] gradient
  const float gscale = gscale_dev ? gscale_host * gscale_dev[0] : gscale_host;
  // wrt_logprobs = 0: gradient w.r.t. the LOGITS (softmax - occupancy); 1: w.r.t. log-probabilities (- occupancy),
  // the quantity torch_imputer returns (imputer.cu:561-638)
  extern __shared__ float occ[];  // [Lmax] exp(alpha+beta+nll) per state, then merged per label
  const int b = (int)(blockIdx.x / T_), t = (int)(blockIdx.x % T_);
  if (cu && t >= cu[b + 1] - cu[b]) return;  // packed batch: the utterance has no such row
  const int64_t row = s2t_utt_row0(cu, b, T_) + t;
  T* g = grad + row * ldg;
  const float n = nll[b];
  const int Tb = min(in_lens[b], T_);
  if (t >= Tb || !(n < INFINITY)) {
    for (int c = threadIdx.x; c < V; c += 256) st_from_f32<T>(g + c, 0.f);
    return;
  }
  const int S = tgt_lens[b];
  const int L = 2 * S + 1;
  const T* x = logits + row * ld;
  const float l = lse[row];
  row_map<T>(x, g, V, [&](int c, float xv) { return wrt_logprobs ? 0.f : gscale * __expf(xv - l); });
  const float* al = alpha + ((int64_t)b * T_ + t) * Lmax;
  const float* be = beta + ((int64_t)b * T_ + t) * Lmax;
  for (int s = threadIdx.x; s < L; s += 256) {
    const float ab = al[s] + be[s];
    occ[s] = (ab == -INFINITY || ab != ab) ? 0.f : __expf(ab + n);  // = exp(alpha+beta)/P(y|x)
  }
  __syncthreads();
  // owner of a label = its first state; blanks (even s) are owned by s = 0
  for (int s = threadIdx.x; s < L; s += 256) {
    int lab;
    bool owner;
    if (!(s & 1)) {
      lab = blank;
      owner = s == 0;
    } else {
      lab = (int)targets[(int64_t)b * ldt + (s >> 1)];
      owner = true;
      for (int q = 0; q < (s >> 1); ++q)
        if ((int)targets[(int64_t)b * ldt + q] == lab) { owner = false; break; }
      if (lab == blank) owner = false;  // (labels never equal blank in valid input)
    }
    if (!owner) continue;
    float sum = 0.f;
    if (!(s & 1)) {
      for (int q = 0; q < L; q += 2) sum += occ[q];
    } else {
      for (int q = (s >> 1); q < S; ++q)
        if ((int)targets[(int64_t)b * ldt + q] == lab) sum += occ[2 * q + 1];
    }
    // occ holds exp(alpha+beta+nll); alpha*beta double counts lp[t,lab] once -> divide by p
    const float xv = ld_as_f32<T>(x + lab);
    const float lp = xv - l;
    const float p = __expf(lp);
    st_from_f32<T>(g + lab, gscale * ((wrt_logprobs ? 0.f : p) - sum * __expf(-lp)));
  }
}

// ---- row softmax over a wide vocabulary (SATE adapter: softmax(ctc_logit / tau), adapter.py:214-217) ----
template <typename T>
__global__ __launch_bounds__(256) void row_softmax_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ p,
                                                              int64_t ldp, int V, float inv_tau,
                                                              const int32_t* __restrict__ live, int) {
  const int64_t row = blockIdx.x;
  if (live && row >= *live) return;  // packed batch: beyond the live rows
  const T* xr = x + row * ldx;
  T* pr = p + row * ldp;
  float mx, lse;
  int arg;
  // statistics of x * inv_tau: max scales, logsumexp is recomputed on the scaled values
  __shared__ float ss2[4];
  row_stats<T>(xr, V, mx, arg, lse);
  float s = 0.f;
  const float m2 = mx * inv_tau;
  row_foreach<T>(xr, V, [&](int c, float v) { s += __expf(v * inv_tau - m2); });
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) ss2[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = 1.f / (ss2[0] + ss2[1] + ss2[2] + ss2[3]);
  row_map<T>(xr, pr, V, [&](int c, float v) { return __expf(v * inv_tau - m2) * inv; });
}

// The same for bf16 rows of at most 256 * 8 * NPT elements held in REGISTERS (one read of the row instead of three: the pass
// is bound by the L2 -> CU path, 3.7 TB/s of algorithmic bytes for the three-pass form on 64000 x 10000): V % 8 == 0, 16-byte
// aligned rows.
template <int NPT>
__global__ __launch_bounds__(256) void row_softmax_fwd_reg_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ p,
                                                                  int64_t ldp, int V, float inv_tau,
                                                                  const int32_t* __restrict__ live) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  if (live && (int)blockIdx.x >= *live) return;  // packed batch: beyond the live rows
  const bf16_t* xr = x + (int64_t)blockIdx.x * ldx;
  bf16_t* pr = p + (int64_t)blockIdx.x * ldp;
  uint4 raw[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int c = (i * 256 + tid) * 8;
    raw[i] = c < V ? *reinterpret_cast<const uint4*>(xr + c) : make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);  // -inf
  }
  float v[NPT][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[i][2 * q] = __uint_as_float(w[q] << 16);
      v[i][2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
      mx = fmaxf(mx, fmaxf(v[i][2 * q], v[i][2 * q + 1]));
    }
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[0][tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  const float m2 = mx * inv_tau;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NPT; ++i)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      v[i][r] = __expf(v[i][r] * inv_tau - m2);  // (exp(-inf) = 0 beyond V)
      sum += v[i][r];
    }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[1][tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[1][0] + red[1][1] + red[1][2] + red[1][3]);
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int c = (i * 256 + tid) * 8;
    if (c < V) {
      uint4 o;
      o.x = bf16pack(v[i][0] * inv, v[i][1] * inv);
      o.y = bf16pack(v[i][2] * inv, v[i][3] * inv);
      o.z = bf16pack(v[i][4] * inv, v[i][5] * inv);
      o.w = bf16pack(v[i][6] * inv, v[i][7] * inv);
      *reinterpret_cast<uint4*>(pr + c) = o;
    }
  }
}

// dx = P * (dP - sum_j P dP) * inv_tau
template <typename T>
__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const T* __restrict__ p, int64_t ldp,
                                                              const T* __restrict__ dp, int64_t lddp,
                                                              T* __restrict__ dx, int64_t lddx, int V, float inv_tau,
                                                              const int32_t* __restrict__ live) {
  __shared__ float sd[4];
  const int64_t row = blockIdx.x;
  if (live && row >= *live) return;
  const T* pr = p + row * ldp;
  const T* dr = dp + row * lddp;
  float dot = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) dot += ld_as_f32<T>(pr + c) * ld_as_f32<T>(dr + c);
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = sd[0] + sd[1] + sd[2] + sd[3];
  T* xr = dx + row * lddx;
  for (int c = threadIdx.x; c < V; c += 256)
    st_from_f32<T>(xr + c, ld_as_f32<T>(pr + c) * (ld_as_f32<T>(dr + c) - dot) * inv_tau);
}

// The same for bf16 rows of at most 256 * 8 * NPT elements with P and dP held in registers (one read of each: the two-pass form
// reads both twice in 2-byte scalars): V % 8 == 0, 16-byte aligned rows.
template <int NPT>
__global__ __launch_bounds__(256) void row_softmax_bwd_reg_kernel(const bf16_t* __restrict__ p, int64_t ldp,
                                                                  const bf16_t* __restrict__ dp, int64_t lddp,
                                                                  bf16_t* __restrict__ dx, int64_t lddx, int V, float inv_tau,
                                                                  const int32_t* __restrict__ live) {
  __shared__ float sd[4];
  const int tid = threadIdx.x;
  if (live && (int)blockIdx.x >= *live) return;
  const bf16_t* pr = p + (int64_t)blockIdx.x * ldp;
  const bf16_t* dr = dp + (int64_t)blockIdx.x * lddp;
  uint4 rp[NPT], rd[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int c = (i * 256 + tid) * 8;
    rp[i] = c < V ? *reinterpret_cast<const uint4*>(pr + c) : make_uint4(0, 0, 0, 0);
    rd[i] = c < V ? *reinterpret_cast<const uint4*>(dr + c) : make_uint4(0, 0, 0, 0);
  }
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const uint32_t a[4] = {rp[i].x, rp[i].y, rp[i].z, rp[i].w}, b[4] = {rd[i].x, rd[i].y, rd[i].z, rd[i].w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dot += __uint_as_float(a[q] << 16) * __uint_as_float(b[q] << 16) +
             __uint_as_float(a[q] & 0xffff0000u) * __uint_as_float(b[q] & 0xffff0000u);
  }
  dot = wave_sum(dot);
  if ((tid & 63) == 0) sd[tid >> 6] = dot;
  __syncthreads();
  dot = sd[0] + sd[1] + sd[2] + sd[3];
  bf16_t* xr = dx + (int64_t)blockIdx.x * lddx;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int c = (i * 256 + tid) * 8;
    if (c < V) {
      const uint32_t a[4] = {rp[i].x, rp[i].y, rp[i].z, rp[i].w}, b[4] = {rd[i].x, rd[i].y, rd[i].z, rd[i].w};
      uint32_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        o[q] = bf16pack(__uint_as_float(a[q] << 16) * (__uint_as_float(b[q] << 16) - dot) * inv_tau,
                        __uint_as_float(a[q] & 0xffff0000u) * (__uint_as_float(b[q] & 0xffff0000u) - dot) * inv_tau);
      *reinterpret_cast<uint4*>(xr + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// best-alignment backtrace (torch_imputer/imputer.py:245-259): start at argmax(alpha[T-1, L-2:]) + L-2 (state 0 when
// L == 1; first maximum on ties), follow the back-pointers; one thread per utterance, states[b, t] (-1 beyond the length)
__global__ void ctc_backtrace_kernel(const float* __restrict__ alpha, const int32_t* __restrict__ paths,
                                     const int32_t* __restrict__ tgt_lens, const int32_t* __restrict__ in_lens, int B,
                                     int T_, int Lmax, int32_t* __restrict__ states) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int L = 2 * tgt_lens[b] + 1;
  const int Tb = min(in_lens[b], T_);
  for (int t = Tb; t < T_; ++t) states[(int64_t)b * T_ + t] = -1;
  if (Tb <= 0) return;
  const float* al = alpha + ((int64_t)b * T_ + (Tb - 1)) * Lmax;
  int cur = 0;
  if (L > 1) cur = (al[L - 1] > al[L - 2]) ? L - 1 : L - 2;
  states[(int64_t)b * T_ + Tb - 1] = cur;
  for (int t = Tb - 1; t > 0; --t) {
    cur = paths[((int64_t)b * T_ + t) * Lmax + cur];
    states[(int64_t)b * T_ + t - 1] = cur;
  }
}

}  // namespace

extern "C" int s2t_argmax_lse(int dtype, const void* logits, int64_t ld, int64_t rows, int V, int32_t* idx,
                              float* top_lp, float* lse, const int32_t* live, void* stream) {
  if (!logits || rows < 0 || V <= 0 || ld < V) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows + ARGMAX_ROWS - 1) / ARGMAX_ROWS)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool arg = idx != nullptr || top_lp != nullptr;
  if (dtype == S2T_F32) {
    if (arg) hipLaunchKernelGGL((argmax_lse_kernel<float, true>), grid, block, 0, s, (const float*)logits, ld, V, rows, idx, top_lp, lse, live);
    else hipLaunchKernelGGL((argmax_lse_kernel<float, false>), grid, block, 0, s, (const float*)logits, ld, V, rows, idx, top_lp, lse, live);
  } else if (dtype == S2T_BF16) {
    if (arg) hipLaunchKernelGGL((argmax_lse_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)logits, ld, V, rows, idx, top_lp, lse, live);
    else hipLaunchKernelGGL((argmax_lse_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)logits, ld, V, rows, idx, top_lp, lse, live);
  } else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ctc_collapse(const int32_t* idx, const float* top_lp, const int32_t* lens, const int32_t* cu, int B, int T,
                                int blank, int64_t* out_tokens, int32_t* out_lens, float* out_scores, void* stream) {
  if (!idx || !top_lp || !lens || !out_tokens || !out_lens || !out_scores || B <= 0 || T <= 0) return S2T_ERR_ARG;
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, idx, top_lp, lens, cu, T, blank,
                     out_tokens, out_lens, out_scores);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ls_cross_entropy(int dtype, const void* logits, int64_t ld, int64_t rows, int V,
                                    const int64_t* target, int64_t pad_idx, float eps, void* dlogits, int64_t ldd,
                                    float* sums, float* ws /* [rows][4] scratch */, const int32_t* live, void* stream) {
  if (!logits || !target || !sums || !ws || rows < 0 || V <= 1 || ld < V) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)rows), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(ls_ce_kernel<float>, grid, block, 0, s, (const float*)logits, ld, V, target, pad_idx, eps, (float*)dlogits, ldd, ws, live);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(ls_ce_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)logits, ld, V, target, pad_idx, eps, (bf16_t*)dlogits, ldd, ws, live);
  else return S2T_ERR_DTYPE;
  hipLaunchKernelGGL(ls_ce_fold_kernel, dim3(1), dim3(1024), 0, s, ws, rows, sums);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ctc_loss_fwd(int dtype, const void* logits, int64_t ld, int B, int T, int V, const float* lse,
                                const int64_t* targets, int ldt, const int32_t* tgt_lens, const int32_t* in_lens,
                                int blank, float* alpha, float* beta, int Lmax, float* nll, const int64_t* force_emits,
                                int32_t* paths, const int32_t* cu, void* stream) {
  if (!logits || !lse || !targets || !tgt_lens || !in_lens || !alpha || !nll) return S2T_ERR_ARG;
  if (!beta && !paths) return S2T_ERR_ARG;
  if (B <= 0 || T <= 0 || V <= 0 || Lmax <= 0 || Lmax > 1023 || !(Lmax & 1)) return S2T_ERR_ARG;
  int threads = (Lmax + 63) / 64 * 64;
  const size_t shm = 2 * (size_t)(Lmax + 2) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (Lmax <= 127 && beta && !paths && !force_emits && (dtype == S2T_F32 || dtype == S2T_BF16)) {
    if (dtype == S2T_F32)
      hipLaunchKernelGGL(ctc_alpha_beta_wave_kernel<float>, dim3(B, 2), dim3(64), 0, s, (const float*)logits, ld, T, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, cu);
    else
      hipLaunchKernelGGL(ctc_alpha_beta_wave_kernel<bf16_t>, dim3(B, 2), dim3(64), 0, s, (const bf16_t*)logits, ld, T, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, cu);
    return S2T_LAUNCH_CHECK();
  }
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(ctc_alpha_beta_kernel<float>, dim3(B, (beta && !paths) ? 2 : 1), dim3(threads), shm, s, (const float*)logits, ld, V, T, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, force_emits, paths, cu);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(ctc_alpha_beta_kernel<bf16_t>, dim3(B, (beta && !paths) ? 2 : 1), dim3(threads), shm, s, (const bf16_t*)logits, ld, V, T, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, force_emits, paths, cu);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ctc_loss_bwd(int dtype, const void* logits, int64_t ld, int B, int T, int V, const float* lse,
                                const int64_t* targets, int ldt, const int32_t* tgt_lens, const int32_t* in_lens,
                                int blank, const float* alpha, const float* beta, int Lmax, const float* nll,
                                float gscale, const float* gscale_dev, void* grad, int64_t ldg, int wrt_logprobs,
                                const int32_t* cu, void* stream) {
  if (!logits || !lse || !targets || !tgt_lens || !in_lens || !alpha || !beta || !nll || !grad) return S2T_ERR_ARG;
  if (B <= 0 || T <= 0 || V <= 0 || Lmax <= 0) return S2T_ERR_ARG;
  const size_t shm = (size_t)Lmax * sizeof(float);
  dim3 grid((unsigned)((int64_t)B * T)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(ctc_grad_kernel<float>, grid, block, shm, s, (const float*)logits, ld, V, T, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, gscale, gscale_dev, (float*)grad, ldg, wrt_logprobs, cu);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(ctc_grad_kernel<bf16_t>, grid, block, shm, s, (const bf16_t*)logits, ld, V, T, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, gscale, gscale_dev, (bf16_t*)grad, ldg, wrt_logprobs, cu);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ctc_backtrace(const float* alpha, const int32_t* paths, const int32_t* tgt_lens, const int32_t* in_lens,
                                 int B, int T, int Lmax, int32_t* states, void* stream) {
  if (!alpha || !paths || !tgt_lens || !in_lens || !states || B <= 0 || T <= 0 || Lmax <= 0) return S2T_ERR_ARG;
  hipLaunchKernelGGL(ctc_backtrace_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, alpha, paths, tgt_lens,
                     in_lens, B, T, Lmax, states);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_row_softmax_fwd(int dtype, const void* x, int64_t ldx, void* p, int64_t ldp, int64_t rows, int V,
                                   float inv_tau, const int32_t* live, void* stream) {
  if (!x || !p || rows < 0 || V <= 0 || ldx < V || ldp < V) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)rows), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(row_softmax_fwd_kernel<float>, grid, block, 0, s, (const float*)x, ldx, (float*)p, ldp, V, inv_tau, live, 0);
  else if (dtype == S2T_BF16) {
    const bool reg = inv_tau > 0.f && V % 8 == 0 && V <= 256 * 8 * 5 && ldx % 8 == 0 && ldp % 8 == 0 && ((uintptr_t)x % 16) == 0 &&
                     ((uintptr_t)p % 16) == 0;
    const int npt = (V + 2047) / 2048;
#define GO(N) hipLaunchKernelGGL(row_softmax_fwd_reg_kernel<N>, grid, block, 0, s, (const bf16_t*)x, ldx, (bf16_t*)p, ldp, V, inv_tau, live)
    if (reg && npt <= 1) GO(1);
    else if (reg && npt == 2) GO(2);
    else if (reg && npt == 3) GO(3);
    else if (reg && npt == 4) GO(4);
    else if (reg && npt == 5) GO(5);
    else hipLaunchKernelGGL(row_softmax_fwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, ldx, (bf16_t*)p, ldp, V, inv_tau, live, 0);
#undef GO
  } else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_row_softmax_bwd(int dtype, const void* p, int64_t ldp, const void* dp, int64_t lddp, void* dx,
                                   int64_t lddx, int64_t rows, int V, float inv_tau, const int32_t* live, void* stream) {
  if (!p || !dp || !dx || rows < 0 || V <= 0) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)rows), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(row_softmax_bwd_kernel<float>, grid, block, 0, s, (const float*)p, ldp, (const float*)dp, lddp, (float*)dx, lddx, V, inv_tau, live);
  else if (dtype == S2T_BF16) {
    const bool reg = V % 8 == 0 && V <= 256 * 8 * 5 && ldp % 8 == 0 && lddp % 8 == 0 && lddx % 8 == 0 && ((uintptr_t)p % 16) == 0 &&
                     ((uintptr_t)dp % 16) == 0 && ((uintptr_t)dx % 16) == 0;
    const int npt = (V + 2047) / 2048;
#define GO(N) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<N>, grid, block, 0, s, (const bf16_t*)p, ldp, (const bf16_t*)dp, lddp, (bf16_t*)dx, lddx, V, inv_tau, live)
    if (reg && npt <= 1) GO(1);
    else if (reg && npt == 2) GO(2);
    else if (reg && npt == 3) GO(3);
    else if (reg && npt == 4) GO(4);
    else if (reg && npt == 5) GO(5);
    else hipLaunchKernelGGL(row_softmax_bwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)p, ldp, (const bf16_t*)dp, lddp, (bf16_t*)dx, lddx, V, inv_tau, live);
#undef GO
  } else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}
