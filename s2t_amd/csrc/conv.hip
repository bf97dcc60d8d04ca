// Conformer convolution-module core: depthwise Conv1d over time + BatchNorm1d + activation
// (reference: modules/convolution.py:94-104; BatchNorm statistics are taken over ALL (B,T) positions,
// padded frames included, exactly as nn.BatchNorm1d on the (B,C,T) tensor does).
//
// Layout is channels-last (B, T, C): a lane owns 4 consecutive channels (8/16-byte accesses), a 256-thread
// workgroup covers 256 channels x 32 time steps; the input window (32 + K - 1 rows) and the K x 256 weights
// are staged in LDS, each thread then produces 8 consecutive time steps for its 4 channels.
#include "common.h"

namespace {

constexpr int TT = 32;      // time steps per workgroup tile
constexpr int CCH = 256;    // channels per workgroup

// stage rows [t_first, t_first + nrows) of utterance b into LDS (fp32), zero outside [0,T).
// Loads are issued in batches of 6 independent requests before any LDS store so that their latencies overlap.
template <typename T>
__device__ __forceinline__ void stage_rows(float* lds, const T* __restrict__ x, int64_t base_row, int T_, int C, int c0,
                                           int t_first, int nrows) {
  constexpr int NB = 6;
  const int total = nrows * 64;
  for (int base = threadIdx.x; base < total; base += 256 * NB) {
    float v[NB][4];
    bool ok[NB];
    // The loads are UNCONDITIONAL (clamped addresses, results discarded by a select afterwards): a load under a branch
    // makes the compiler wait for it on the spot, which serialises the whole batch (one memory latency per load).
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int idx = min(base + q * 256, total - 1);
      const int r = idx >> 6, cq = idx & 63;
      const int t = t_first + r;
      const int c = c0 + cq * 4;
      ok[q] = base + q * 256 < total && t >= 0 && t < T_ && c < C;
      const int tc = min(max(t, 0), T_ - 1), cc = min(c, C - 4);
      ld4_as_f32<T>(x + (base_row + tc) * C + cc, v[q]);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int idx = base + q * 256;
      if (idx < total) {
        const int r = idx >> 6, cq = idx & 63;
        *reinterpret_cast<float4*>(lds + r * CCH + cq * 4) =
            make_float4(ok[q] ? v[q][0] : 0.f, ok[q] ? v[q][1] : 0.f, ok[q] ? v[q][2] : 0.f, ok[q] ? v[q][3] : 0.f);
      }
    }
  }
}

// y[b,t,c] = sum_k x[b, t + k - pad, c] * w[c, flip ? K-1-k : k]  ; optional affine+act+mask epilogue; optional stats
template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     T* __restrict__ y, int B, int T_, int C, int K, int flip,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int act, const int32_t* __restrict__ lens,
                                                     const int32_t* __restrict__ cu,
                                                     float* __restrict__ stats, const float* __restrict__ bn_mean,
                                                     const float* __restrict__ bn_var, float bn_eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = (K - 1) / 2;
  const int nrows = TT + K - 1;
  float* lx = smem;                 // [nrows][256]
  float* lw = smem + nrows * CCH;   // [K][256]
  float* lred = lw + K * CCH;       // [2][4][256] (stats only)
  const int t0 = blockIdx.x * TT;
  const int b = blockIdx.y;
  const int c0 = blockIdx.z * CCH;
  // packed batch: the utterance's rows (frames + halo rows) start at cu[b]; the grid still covers the padded length, tiles
  // beyond the utterance's rows leave at once (their partial statistics row is zero: the frames they stand for are zero)
  const int64_t row_b = s2t_utt_row0(cu, b, T_);
  const int Tfull = T_;
  T_ = s2t_utt_rows(cu, b, Tfull);
  if (t0 >= T_) {
    if (stats && c0 + (int)threadIdx.x < C) {
      float* row = stats + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
      row[c0 + threadIdx.x] = 0.f;
      row[C + c0 + threadIdx.x] = 0.f;
    }
    return;
  }
  stage_rows<T>(lx, x, row_b, T_, C, c0, t0 - pad, nrows);
  for (int idx = threadIdx.x; idx < K * CCH; idx += 256) {
    const int k = idx / CCH, c = idx % CCH;
    lw[idx] = (c0 + c < C) ? w[(int64_t)(c0 + c) * K + (flip ? K - 1 - k : k)] : 0.f;
  }
  __syncthreads();
  const int cq = threadIdx.x & 63, tg = threadIdx.x >> 6;
  float acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  // (the 8 window rows of tap k are those of tap k - 1 moved by one: they stay in registers, slot (i + k) & 7 — static under the
  // unroll by eight — and a tap fetches ONE row, K + 7 LDS reads per thread instead of 8 K; the same products in the same order)
  float4 win[8];
#pragma unroll
  for (int i = 0; i < 7; ++i) win[i] = *reinterpret_cast<const float4*>(lx + (tg * 8 + i) * CCH + cq * 4);
  for (int kb = 0; kb < K; kb += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb + u;
      if (k < K) {
        const float4 w4 = *reinterpret_cast<const float4*>(lw + k * CCH + cq * 4);
        win[(7 + u) & 7] = *reinterpret_cast<const float4*>(lx + (tg * 8 + 7 + k) * CCH + cq * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 x4 = win[(i + u) & 7];
          acc[i][0] += w4.x * x4.x;
          acc[i][1] += w4.y * x4.y;
          acc[i][2] += w4.z * x4.z;
          acc[i][3] += w4.w * x4.w;
        }
      }
    }
  }
  const int c = c0 + cq * 4;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (scale && c < C) {
    ld4_as_f32<float>(scale + c, sc);
    ld4_as_f32<float>(shift + c, sh);
    if (bn_mean) {  // scale / shift are BatchNorm's gamma / beta: fold the running statistics here (eval)
      float mu[4], var[4];
      ld4_as_f32<float>(bn_mean + c, mu);
      ld4_as_f32<float>(bn_var + c, var);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[r] *= rsqrtf(var[r] + bn_eps);
        sh[r] -= mu[r] * sc[r];
      }
    }
  }
  const int len = lens ? lens[b] : T_;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = t0 + tg * 8 + i;
    if (t < T_ && c < C) {
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[r] += acc[i][r];
        s2[r] += acc[i][r] * acc[i][r];
        o[r] = acc[i][r];
        if (scale) o[r] = (t < len) ? act_apply(act, o[r] * sc[r] + sh[r]) : 0.f;
      }
      st4_from_f32<T>(y + (row_b + t) * C + c, o);
    }
  }
  if (stats) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lred[(0 * 4 + tg) * CCH + cq * 4 + r] = s1[r];
      lred[(1 * 4 + tg) * CCH + cq * 4 + r] = s2[r];
    }
    __syncthreads();
    const int cc = threadIdx.x;
    if (c0 + cc < C) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        a += lred[(0 * 4 + g) * CCH + cc];
        q += lred[(1 * 4 + g) * CCH + cc];
      }
      // one row of partial sums per workgroup (no atomics: bn_finalize adds the rows in a fixed order, so the batch
      // statistics — and with them the whole forward pass — are reproducible run to run)
      float* row = stats + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
      row[c0 + cc] = a;
      row[C + c0 + cc] = q;
    }
  }
}

// dw[c,k] += sum_{b,t} dD[b,t,c] * G[b, t + k - pad, c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ G, const T* __restrict__ dD,
                                                           float* __restrict__ dw_ws, int replicas, int B, int T_,
                                                           int C, int K, int tiles_t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = (K - 1) / 2;
  const int nrows = TT + K - 1;
  float* lg = smem;                 // [nrows][256]
  float* ld_ = smem + nrows * CCH;  // [TT][256]
  const int c0 = blockIdx.z * CCH;
  const int cq = threadIdx.x & 63, tg = threadIdx.x >> 6;
  float acc[8][4];  // k = tg + 4*kk
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  const int total = B * tiles_t;
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int b = tile / tiles_t, t0 = (tile % tiles_t) * TT;
    __syncthreads();
    stage_rows<T>(lg, G, (int64_t)b * T_, T_, C, c0, t0 - pad, nrows);
    stage_rows<T>(ld_, dD, (int64_t)b * T_, T_, C, c0, t0, TT);
    __syncthreads();
    for (int t = 0; t < TT; ++t) {
      const float4 d4 = *reinterpret_cast<const float4*>(ld_ + t * CCH + cq * 4);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = tg + 4 * kk;
        if (k < K) {
          const float4 g4 = *reinterpret_cast<const float4*>(lg + (t + k) * CCH + cq * 4);
          acc[kk][0] += d4.x * g4.x;
          acc[kk][1] += d4.y * g4.y;
          acc[kk][2] += d4.z * g4.z;
          acc[kk][3] += d4.w * g4.w;
        }
      }
    }
  }
  // transpose the per-thread partials through LDS and store this workgroup's [channels][K] partial as one contiguous row
  // of the workspace (no atomics: dwconv_wgrad_fold_kernel adds the rows in a fixed order)
  __syncthreads();
  float* lt = smem;  // [256 channels][K]
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int k = tg + 4 * kk;
    if (k < K) {
#pragma unroll
      for (int r = 0; r < 4; ++r) lt[(cq * 4 + r) * K + k] = acc[kk][r];
    }
  }
  __syncthreads();
  float* dw = dw_ws + (int64_t)blockIdx.x * C * K + (int64_t)c0 * K;
  const int nvalid = min(CCH, C - c0) * K;
  for (int idx = threadIdx.x; idx < nvalid; idx += 256) dw[idx] = lt[idx];
}


// Fixed-order column sum of `rows` partial rows (row stride `ld` floats) for 16 adjacent columns per 1024-thread
// workgroup.  Thread (g = tid / 16, cl = tid % 16) adds rows g, g + 64, ... in order (loads issued four at a time), the
// four g of a wave are combined by two xor shuffles, the 16 waves through LDS in wave order: the summation tree depends on
// `rows` only, never on scheduling.  Every thread returns the total for its column.
__device__ __forceinline__ float fold_partial_rows(const float* __restrict__ col, int rows, int64_t ld, bool live,
                                                   float (*red)[16]) {
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4, wv = threadIdx.x >> 6;
  float acc = 0.f;
  if (live) {
    int r = g;
    for (; r + 192 < rows; r += 256) {
      const float v0 = col[(int64_t)r * ld], v1 = col[(int64_t)(r + 64) * ld], v2 = col[(int64_t)(r + 128) * ld],
                  v3 = col[(int64_t)(r + 192) * ld];
      acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; r < rows; r += 64) acc += col[(int64_t)r * ld];
  }
  acc = s2t_xadd<16>(acc);
  acc = s2t_xadd<32>(acc);
  __syncthreads();  // `red` may still be read from a previous call
  if ((threadIdx.x & 63) < 16) red[wv][cl] = acc;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += red[i][cl];
  return t;
}

// per-channel BatchNorm bookkeeping: partial rows -> (scale, shift, mean, rstd) and running-stat update
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ stats, int partials, float count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float momentum, float eps, int training,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o, int C) {
  __shared__ float red[16][16];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  float t1 = 0.f, t2 = 0.f;
  if (training) {
    t1 = fold_partial_rows(stats + c, partials, 2 * (int64_t)C, c < C, red);
    t2 = fold_partial_rows(stats + C + c, partials, 2 * (int64_t)C, c < C, red);
  }
  if (threadIdx.x >= 16 || c >= C) return;
  float mean, var;
  if (training) {
    mean = t1 / count;
    var = fmaxf(t2 / count - mean * mean, 0.f);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float rstd = rsqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  if (mean_o) {
    mean_o[c] = mean;
    rstd_o[c] = rstd;
  }
}

// out = act(D*scale + shift), padded rows -> 0
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ D, T* __restrict__ out,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, int64_t rows, int C,
                                                         const int32_t* __restrict__ lens, int Tn) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vpr = C / 4;
  rows = s2t_live_rows(lens, Tn, rows);
  if (idx >= rows * vpr) return;
  const int64_t row = idx / vpr;
  const int c = (int)(idx % vpr) * 4;
  float d[4], sc[4], sh[4], o[4];
  ld4_as_f32<T>(D + row * C + c, d);
  ld4_as_f32<float>(scale + c, sc);
  ld4_as_f32<float>(shift + c, sh);
  const bool masked = lens && s2t_row_masked(lens, Tn, row);
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = masked ? 0.f : act_apply(act, d[r] * sc[r] + sh[r]);
  st4_from_f32<T>(out + row * C + c, o);
}

// pass 1 of the backward: sums[c] += du, sums[C+c] += du * xhat with du = dOut * act'(u), u = D*scale+shift
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const T* __restrict__ D, const T* __restrict__ dOut,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, int act, int64_t rows,
                                                                int C, const int32_t* __restrict__ lens, int Tn,
                                                                float* __restrict__ sums) {
  __shared__ float red[2][4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  rows = s2t_live_rows(lens, Tn, rows);
  if (c < C) {
    float sc[4], sh[4], mu[4], rs[4];
    ld4_as_f32<float>(scale + c, sc);
    ld4_as_f32<float>(shift + c, sh);
    ld4_as_f32<float>(mean + c, mu);
    ld4_as_f32<float>(rstd + c, rs);
    // four rows per trip, all eight loads issued unconditionally (rows past the end re-read the last row, masked rows
    // are loaded anyway) and discarded by a select: loads under a branch are waited for one at a time
    const int64_t step = (int64_t)gridDim.y * 4;
    for (int64_t m0 = (int64_t)blockIdx.y * 4 + w; m0 < rows; m0 += 4 * step) {
      float d[4][4], g[4][4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t m = m0 + u * step;
        const int64_t mc = m < rows ? m : rows - 1;
        ok[u] = m < rows && !(lens && s2t_row_masked(lens, Tn, mc));
        ld4_as_f32<T>(D + mc * C + c, d[u]);
        ld4_as_f32<T>(dOut + mc * C + c, g[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float du = ok[u] ? g[u][r] * act_grad(act, d[u][r] * sc[r] + sh[r]) : 0.f;
          a1[r] += du;
          a2[r] += du * (d[u][r] - mu[r]) * rs[r];
        }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[0][w][lane * 4 + r] = a1[r];
    red[1][w][lane * 4 + r] = a2[r];
  }
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {  // `sums` is this pass's partial buffer [gridDim.y][2][C]; bn_bwd_fold_kernel adds the rows in order
    float* row = sums + (int64_t)blockIdx.y * 2 * C;
    row[cc] = ((red[0][0][threadIdx.x] + red[0][1][threadIdx.x]) + red[0][2][threadIdx.x]) + red[0][3][threadIdx.x];
    row[C + cc] = ((red[1][0][threadIdx.x] + red[1][1][threadIdx.x]) + red[1][2][threadIdx.x]) + red[1][3][threadIdx.x];
  }
}

// sums[0:2C] = fixed-order sum of the partial rows
__global__ __launch_bounds__(1024) void bn_bwd_fold_kernel(const float* __restrict__ partial, int rows, int C,
                                                           float* __restrict__ sums, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  __shared__ float red[16][16];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);  // over 2C columns
  const float t = fold_partial_rows(partial + c, rows, 2 * (int64_t)C, c < 2 * C, red);
  if (threadIdx.x < 16 && c < 2 * C) {
    sums[c] = t;
    // sums[0:C] = sum du = dbeta, sums[C:2C] = sum du*xhat = dgamma: optionally accumulated into the parameter gradients
    if (c < C) {
      if (dbeta) dbeta[c] += t;
    } else if (dgamma) {
      dgamma[c - C] += t;
    }
  }
}

// out[0:n] += fixed-order sum of the partial rows [rows][n]   (blockIdx.y: the entry of a batch)
constexpr int ROWS_BATCH = 16;
struct RowsBatch {
  const float* partial[ROWS_BATCH];
  float* out[ROWS_BATCH];
};
__global__ __launch_bounds__(1024) void rows_fold_add_kernel(const RowsBatch batch, int rows, int64_t n) {
  const float* __restrict__ partial = batch.partial[blockIdx.y];
  float* __restrict__ out = batch.out[blockIdx.y];
  __shared__ float red[16][16];
  const int64_t c = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const float t = fold_partial_rows(partial + c, rows, n, c < n, red);
  if (threadIdx.x < 16 && c < n) out[c] += t;
}

// pass 2: dD = gamma*rstd * (du - s1/n - xhat * s2/n)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ D, const T* __restrict__ dOut,
                                                               T* __restrict__ dD, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ sums, float count, int act,
                                                               int64_t rows, int C, const int32_t* __restrict__ lens,
                                                               int Tn) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vpr = C / 4;
  rows = s2t_live_rows(lens, Tn, rows);
  if (idx >= rows * vpr) return;
  const int64_t row = idx / vpr;
  const int c = (int)(idx % vpr) * 4;
  float d[4], g[4], sc[4], sh[4], mu[4], rs[4], s1[4], s2[4], o[4];
  ld4_as_f32<T>(D + row * C + c, d);
  ld4_as_f32<T>(dOut + row * C + c, g);
  ld4_as_f32<float>(scale + c, sc);
  ld4_as_f32<float>(shift + c, sh);
  ld4_as_f32<float>(mean + c, mu);
  ld4_as_f32<float>(rstd + c, rs);
  ld4_as_f32<float>(sums + c, s1);
  ld4_as_f32<float>(sums + C + c, s2);
  const bool masked = lens && s2t_row_masked(lens, Tn, row);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float du = masked ? 0.f : g[r] * act_grad(act, d[r] * sc[r] + sh[r]);
    const float xh = (d[r] - mu[r]) * rs[r];
    o[r] = sc[r] * (du - s1[r] / count - xh * s2[r] / count);  // scale = gamma * rstd
  }
  st4_from_f32<T>(dD + row * C + c, o);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the convolution module's middle (BatchNorm + activation -> depthwise conv -> GLU) in ONE launch, bf16:
//   dD  = gamma*rstd * (du - s1/n - xhat*s2/n),  du = dA * act'(D*scale + shift), padded frames: du = 0        (BatchNorm, pass 2)
//   dG  = depthwise conv of dD with the flipped kernel                                         (s2t_dwconv_fwd, flip)
//   dZ  = [dG * sigmoid(gate) | dG * value * sigmoid'(gate)]                                   (s2t_glu_bwd)
//   dw[c][k] partial = sum_t dD[t][c] * G[t + k - pad][c]                                      (s2t_dwconv_bwd_weight)
// replacing four launches that pass dD and dG through HBM (convolution.py:92-104 backward).  A workgroup owns 32 time
// steps x 256 channels of one utterance: the dD window (32 + K - 1 rows, computed while staging, rounded to bf16 as the
// stored tensor of the unfused path is) and the G window sit in LDS as bf16; thread (channel quad cq, time group tg)
// produces 8 time steps of dG / dZ and its share of the weight-gradient partial, which leaves as one plain row per
// workgroup (rows_fold_add_kernel adds them in a fixed order).
__global__ __launch_bounds__(256) void conv_bwd_fused_kernel(
    const bf16_t* __restrict__ D, const bf16_t* __restrict__ dA, const bf16_t* __restrict__ G, const bf16_t* __restrict__ Z,
    const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ sums, float inv_count,
    int act, const int32_t* __restrict__ lens, const int32_t* __restrict__ cu, bf16_t* __restrict__ dZ,
    float* __restrict__ dw_ws, int B, int T_, int C, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = (K - 1) / 2;
  const int nrows = TT + K - 1;
  bf16_t* ld_ = reinterpret_cast<bf16_t*>(smem);  // [nrows][256] dD window
  bf16_t* lg = ld_ + nrows * CCH;                 // [nrows][256] G window
  float* lw = reinterpret_cast<float*>(lg + nrows * CCH);  // [K][256] flipped weights; reused for the dw transpose
  const int t0 = blockIdx.x * TT;
  const int b = blockIdx.y;
  const int c0 = blockIdx.z * CCH;
  // packed batch: rows of utterance b (frames + halo rows) from cu[b]; a tile beyond them stands for padded frames whose dD
  // reaches no frame and whose G window is zero: nothing to store but a zero weight-gradient partial row
  const int64_t row_b = s2t_utt_row0(cu, b, T_);
  T_ = s2t_utt_rows(cu, b, T_);
  if (t0 >= T_) {
    float* dwr = dw_ws + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * C * K + (int64_t)c0 * K;
    const int nvalid = min(CCH, C - c0) * K;
    for (int idx = threadIdx.x; idx < nvalid; idx += 256) dwr[idx] = 0.f;
    return;
  }
  const int len = lens ? min(lens[b], T_) : T_;
  for (int idx = threadIdx.x; idx < K * CCH; idx += 256) {
    const int k = idx / CCH, c = idx % CCH;
    lw[idx] = (c0 + c < C) ? w[(int64_t)(c0 + c) * K + (K - 1 - k)] : 0.f;
  }
  {  // ---- staging: dD computed on the way in, G copied; 8 independent row pieces per thread and batch
    const int cq = threadIdx.x & 63;
    const int c = c0 + cq * 4;
    const int cc = min(c, C - 4);
    float sc[4], sh[4], mu[4], rs[4], m1[4], m2[4];
    ld4_as_f32<float>(scale + cc, sc);
    ld4_as_f32<float>(shift + cc, sh);
    ld4_as_f32<float>(mean + cc, mu);
    ld4_as_f32<float>(rstd + cc, rs);
    ld4_as_f32<float>(sums + cc, m1);
    ld4_as_f32<float>(sums + C + cc, m2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m1[r] *= inv_count;
      m2[r] *= inv_count;
    }
    constexpr int NB = 4;
    for (int r0 = threadIdx.x >> 6; r0 < nrows; r0 += 4 * NB) {
      float dv[NB][4], av[NB][4], gv[NB][4];
      bool ok[NB];
      int tt[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {  // unconditional, clamped loads (a load under a branch is waited for on the spot)
        const int r = min(r0 + 4 * q, nrows - 1);
        const int t = t0 - pad + r;
        tt[q] = t;
        ok[q] = r0 + 4 * q < nrows && t >= 0 && t < T_ && c < C;
        const int64_t row = row_b + min(max(t, 0), T_ - 1);
        ld4_as_f32<bf16_t>(D + row * C + cc, dv[q]);
        ld4_as_f32<bf16_t>(dA + row * C + cc, av[q]);
        ld4_as_f32<bf16_t>(G + row * C + cc, gv[q]);
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int r = r0 + 4 * q;
        if (r < nrows) {
          float o[4], g4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float du = tt[q] < len ? av[q][e] * act_grad(act, dv[q][e] * sc[e] + sh[e]) : 0.f;
            const float xh = (dv[q][e] - mu[e]) * rs[e];
            o[e] = ok[q] ? sc[e] * (du - m1[e] - xh * m2[e]) : 0.f;
            g4[e] = ok[q] ? gv[q][e] : 0.f;
          }
          st4_from_f32<bf16_t>(ld_ + r * CCH + cq * 4, o);
          st4_from_f32<bf16_t>(lg + r * CCH + cq * 4, g4);
        }
      }
    }
  }
  __syncthreads();
  const int cq = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c = c0 + cq * 4;
  {  // ---- dG = conv(dD, flipped w) for 8 time steps, GLU backward, dZ out
    float acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    // The kernel is bound by vector issue (two waves per SIMD, ~4.5 k VALU instructions per thread, half of them bf16 -> fp32
    // unpacking of LDS operands fetched again for every tap): the 8 window rows of tap k are those of tap k - 1 moved by one, so
    // the rows stay in registers — slot (i + k) & 7, static under the unroll by eight — and each tap fetches and unpacks ONE row
    // (K + 7 instead of 8 K).  Same products in the same order: the same bits.
    float win[8][4];
#pragma unroll
    for (int i = 0; i < 7; ++i) ld4_as_f32<bf16_t>(ld_ + (tg * 8 + i) * CCH + cq * 4, win[i]);
    for (int kb = 0; kb < K; kb += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + u;
        if (k < K) {
          const float4 w4 = *reinterpret_cast<const float4*>(lw + k * CCH + cq * 4);
          ld4_as_f32<bf16_t>(ld_ + (tg * 8 + 7 + k) * CCH + cq * 4, win[(7 + u) & 7]);  // row i + k of i = 7
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float (&x4)[4] = win[(i + u) & 7];
            acc[i][0] += w4.x * x4[0];
            acc[i][1] += w4.y * x4[1];
            acc[i][2] += w4.z * x4[2];
            acc[i][3] += w4.w * x4[3];
          }
        }
      }
    }
    float zv[8][4], zg[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // all sixteen loads first
      const int t = min(t0 + tg * 8 + i, T_ - 1);
      const int64_t row = row_b + t;
      const int cc = min(c, C - 4);
      ld4_as_f32<bf16_t>(Z + row * 2 * C + cc, zv[i]);
      ld4_as_f32<bf16_t>(Z + row * 2 * C + C + cc, zg[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + tg * 8 + i;
      if (t < T_ && c < C) {
        float da[4], dg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dy = bf2f(f2bf(acc[i][r]));  // the unfused path stores dG in bf16 before the GLU backward
          const float sg = sigmoidf_(zg[i][r]);
          da[r] = dy * sg;
          dg[r] = dy * zv[i][r] * sg * (1.f - sg);
        }
        const int64_t row = row_b + t;
        st4_from_f32<bf16_t>(dZ + row * 2 * C + c, da);
        st4_from_f32<bf16_t>(dZ + row * 2 * C + C + c, dg);
      }
    }
  }
  // ---- weight-gradient partial: dw[c][k] = sum over the tile's 32 time steps of dD[t][c] * G[t + k - pad][c]
  // Thread (cq, tg) owns EIGHT CONSECUTIVE taps k = 8 kg + j (kg = tg % NG, NG = ceil(K / 8) tap groups) over the time steps of its
  // share th = tg / NG of the tile (NS = 4 / NG shares: two halves at K = 15, the whole tile at K = 31): the eight G rows t + k of
  // step t are those of step t - 1 moved by one, so they too stay in registers (slot (j + t) & 7) and a step fetches and unpacks
  // two rows — dD[t] and the new G row — where taps strided by four made it nine.  The shares of a tap meet in the transpose buffer.
  const int NG = (K + 7) >> 3;            // 1, 2 or 4 (K <= 31; NG = 3 runs as 4 with an idle group)
  const int NGe = NG == 3 ? 4 : NG;
  const int NS = 4 / NGe;
  const int kg = tg % NGe, th = tg / NGe;
  const int k0 = 8 * kg;
  const int tlen = TT / NS, ts = th * tlen;   // (TT = 32: 32, 16 or 8 steps, multiples of 8)
  float accw[8][4];  // tap k0 + j
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) accw[i][r] = 0.f;
  {
    float gw[8][4];
    auto grow = [&](int row, float (&v)[4]) __attribute__((always_inline)) {
      ld4_as_f32<bf16_t>(lg + min(row, nrows - 1) * CCH + cq * 4, v);  // (rows beyond the window belong to taps >= K: never stored)
    };
#pragma unroll
    for (int j = 0; j < 7; ++j) grow(ts + k0 + j, gw[j]);
    for (int tb = 0; tb < tlen; tb += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = ts + tb + u;
        float d4[4];
        ld4_as_f32<bf16_t>(ld_ + (t + pad) * CCH + cq * 4, d4);
        grow(t + k0 + 7, gw[(7 + u) & 7]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float (&g4)[4] = gw[(j + u) & 7];
          accw[j][0] += d4[0] * g4[0];
          accw[j][1] += d4[1] * g4[1];
          accw[j][2] += d4[2] * g4[2];
          accw[j][3] += d4[3] * g4[3];
        }
      }
    }
  }
  __syncthreads();  // the flipped weights are no longer read: lw becomes the [256 channels][K] transpose buffer
  for (int sh = 0; sh < NS; ++sh) {  // the time shares of a tap one after the other (a fixed order)
    if (th == sh) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        if (k < K) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = lw + (cq * 4 + r) * K + k;
            *q = sh == 0 ? accw[j][r] : *q + accw[j][r];
          }
        }
      }
    }
    __syncthreads();
  }
  float* dwr = dw_ws + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * C * K + (int64_t)c0 * K;
  const int nvalid = min(CCH, C - c0) * K;
  for (int idx = threadIdx.x; idx < nvalid; idx += 256) dwr[idx] = lw[idx];
}

// dynamic LDS above the 64 KiB default needs a one-time opt-in per kernel (never inside a stream capture)
void ensure_lds_optin() {
  static bool done = false;
  if (done) return;
  hipFuncSetAttribute((const void*)dwconv_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute((const void*)dwconv_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute((const void*)dwconv_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute((const void*)dwconv_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute((const void*)conv_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  done = true;
}

}  // namespace

extern "C" int s2t_dwconv_fwd(int dtype, const void* x, const float* w, void* y, int B, int T, int C, int K, int flip,
                              const float* scale, const float* shift, int act, const int32_t* lens, const int32_t* cu,
                              float* stats, void* stream) {
  if (!x || !w || !y || B <= 0 || T <= 0 || C <= 0 || K <= 0 || !(K & 1) || C % 4) return S2T_ERR_ARG;
  if (K > 31) return S2T_ERR_UNSUPPORTED;
  if ((scale == nullptr) != (shift == nullptr)) return S2T_ERR_ARG;
  const size_t shm = (size_t)((TT + K - 1) * CCH + K * CCH + (stats ? 8 * CCH : 0)) * sizeof(float);
  dim3 grid((T + TT - 1) / TT, B, (C + CCH - 1) / CCH), block(256);
  hipStream_t s = (hipStream_t)stream;
  ensure_lds_optin();
  if (dtype == S2T_F32) {
    hipLaunchKernelGGL(dwconv_kernel<float>, grid, block, shm, s, (const float*)x, w, (float*)y, B, T, C, K, flip, scale,
                       shift, act, lens, cu, stats, (const float*)nullptr, (const float*)nullptr, 0.f);
  } else if (dtype == S2T_BF16) {
    hipLaunchKernelGGL(dwconv_kernel<bf16_t>, grid, block, shm, s, (const bf16_t*)x, w, (bf16_t*)y, B, T, C, K, flip,
                       scale, shift, act, lens, cu, stats, (const float*)nullptr, (const float*)nullptr, 0.f);
  } else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_dwconv_bn_eval_fwd(int dtype, const void* x, const float* w, void* y, int B, int T, int C, int K,
                                      const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, int act, const int32_t* lens, const int32_t* cu,
                                      void* stream) {
  if (!x || !w || !y || !gamma || !beta || !running_mean || !running_var) return S2T_ERR_ARG;
  if (B <= 0 || T <= 0 || C <= 0 || K <= 0 || !(K & 1) || C % 4) return S2T_ERR_ARG;
  if (K > 31) return S2T_ERR_UNSUPPORTED;
  const size_t shm = (size_t)((TT + K - 1) * CCH + K * CCH) * sizeof(float);
  dim3 grid((T + TT - 1) / TT, B, (C + CCH - 1) / CCH), block(256);
  hipStream_t s = (hipStream_t)stream;
  ensure_lds_optin();
  if (dtype == S2T_F32) {
    hipLaunchKernelGGL(dwconv_kernel<float>, grid, block, shm, s, (const float*)x, w, (float*)y, B, T, C, K, 0, gamma, beta,
                       act, lens, cu, (float*)nullptr, running_mean, running_var, eps);
  } else if (dtype == S2T_BF16) {
    hipLaunchKernelGGL(dwconv_kernel<bf16_t>, grid, block, shm, s, (const bf16_t*)x, w, (bf16_t*)y, B, T, C, K, 0, gamma,
                       beta, act, lens, cu, (float*)nullptr, running_mean, running_var, eps);
  } else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_dwconv_bwd_weight(int dtype, const void* G, const void* dD, float* dw, float* ws, int replicas, int B,
                                     int T, int C, int K, void* stream) {
  if (!G || !dD || !dw || !ws || replicas <= 0 || B <= 0 || T <= 0 || C <= 0 || K <= 0 || !(K & 1) || C % 4) return S2T_ERR_ARG;
  if (K > 31) return S2T_ERR_UNSUPPORTED;
  const size_t shm = (size_t)((TT + K - 1) * CCH + TT * CCH) * sizeof(float);
  const int tiles_t = (T + TT - 1) / TT;
  int nb = B * tiles_t;
  if (nb > 1024) nb = 1024;
  if (replicas < nb) return S2T_ERR_ARG;  // ws: one row of C*K floats per workgroup (s2t_dwconv_wgrad_partials)
  dim3 grid(nb, 1, (C + CCH - 1) / CCH), block(256);
  hipStream_t s = (hipStream_t)stream;
  ensure_lds_optin();
  if (dtype == S2T_F32) {
    hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, block, shm, s, (const float*)G, (const float*)dD, ws, replicas, B, T, C, K, tiles_t);
  } else if (dtype == S2T_BF16) {
    hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, grid, block, shm, s, (const bf16_t*)G, (const bf16_t*)dD, ws, replicas, B, T, C, K, tiles_t);
  } else return S2T_ERR_DTYPE;
  const int64_t n = (int64_t)C * K;
  RowsBatch bt = {};
  bt.partial[0] = ws;
  bt.out[0] = dw;
  hipLaunchKernelGGL(rows_fold_add_kernel, dim3((unsigned)((n + 15) / 16)), dim3(1024), 0, s, bt, nb, n);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_conv_bwd_fused(const void* D, const void* dA, const void* G, const void* Z, const float* w,
                                  const float* scale, const float* shift, const float* mean, const float* rstd,
                                  const float* sums, float count, int act, const int32_t* lens, const int32_t* cu, void* dZ,
                                  float* dw, float* ws, int B, int T, int C, int K, void* stream) {
  if (!D || !dA || !G || !Z || !w || !scale || !shift || !mean || !rstd || !sums || !dZ || !ws) return S2T_ERR_ARG;
  if (B <= 0 || T <= 0 || C <= 0 || K <= 0 || !(K & 1) || C % 4 || count <= 0.f) return S2T_ERR_ARG;
  if (K > 31) return S2T_ERR_UNSUPPORTED;
  const void* ptrs[] = {D, dA, G, Z, dZ};
  for (const void* q : ptrs)
    if ((uintptr_t)q % 8) return S2T_ERR_ALIGN;
  const int tiles_t = (T + TT - 1) / TT;
  const size_t shm = (size_t)(2 * (TT + K - 1) * CCH) * sizeof(bf16_t) + (size_t)(K * CCH) * sizeof(float);
  dim3 grid(tiles_t, B, (C + CCH - 1) / CCH), block(256);
  hipStream_t s = (hipStream_t)stream;
  ensure_lds_optin();
  hipLaunchKernelGGL(conv_bwd_fused_kernel, grid, block, shm, s, (const bf16_t*)D, (const bf16_t*)dA, (const bf16_t*)G,
                     (const bf16_t*)Z, w, scale, shift, mean, rstd, sums, 1.0f / count, act, lens, cu, (bf16_t*)dZ, ws, B, T, C, K);
  if (dw) {  // (dw == NULL: the caller folds the partial rows later, several layers per launch: s2t_rows_fold_add)
    const int64_t n = (int64_t)C * K;
    RowsBatch bt = {};
    bt.partial[0] = ws;
    bt.out[0] = dw;
    hipLaunchKernelGGL(rows_fold_add_kernel, dim3((unsigned)((n + 15) / 16)), dim3(1024), 0, s, bt, B * tiles_t, n);
  }
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_rows_fold_add(const float* const* partials, float* const* outs, int count, int rows, int64_t n, void* stream) {
  if (!partials || !outs || count <= 0 || rows <= 0 || n <= 0) return S2T_ERR_ARG;
  for (int i0 = 0; i0 < count; i0 += ROWS_BATCH) {
    const int cnt = count - i0 < ROWS_BATCH ? count - i0 : ROWS_BATCH;
    RowsBatch bt = {};
    for (int i = 0; i < cnt; ++i) {
      if (!partials[i0 + i] || !outs[i0 + i]) return S2T_ERR_ARG;
      bt.partial[i] = partials[i0 + i];
      bt.out[i] = outs[i0 + i];
    }
    hipLaunchKernelGGL(rows_fold_add_kernel, dim3((unsigned)((n + 15) / 16), cnt), dim3(1024), 0, (hipStream_t)stream, bt, rows, n);
  }
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_dwconv_wgrad_partials(int B, int T) {
  const int nb = B * ((T + TT - 1) / TT);
  return nb > 1024 ? 1024 : nb;
}

extern "C" int s2t_dwconv_stat_partials(int B, int T) { return B * ((T + TT - 1) / TT); }
extern "C" int s2t_bn_bwd_partials(int64_t rows) {
  int64_t slices = (rows + 31) / 32;  // 8 rows per wave: the pass is latency bound with fewer workgroups
  return (int)(slices > 512 ? 512 : slices);
}

extern "C" int s2t_bn_finalize(const float* stats, int partials, float count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, int training,
                               float* scale, float* shift, float* mean, float* rstd, int C, void* stream) {
  if (!gamma || !beta || !scale || !shift || C <= 0) return S2T_ERR_ARG;
  if (training && (!stats || partials <= 0)) return S2T_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return S2T_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, stats, partials, count,
                     gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift, mean, rstd, C);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_bn_act_fwd(int dtype, const void* D, void* out, const float* scale, const float* shift, int act,
                              int64_t rows, int C, const int32_t* lens, int T, void* stream) {
  if (!D || !out || !scale || !shift || rows < 0 || C <= 0 || C % 4) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows * (C / 4) + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(bn_act_fwd_kernel<float>, grid, block, 0, s, (const float*)D, (float*)out, scale, shift, act, rows, C, lens, T);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(bn_act_fwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)D, (bf16_t*)out, scale, shift, act, rows, C, lens, T);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_bn_act_bwd(int dtype, const void* D, const void* dOut, void* dD, const float* scale,
                              const float* shift, const float* mean, const float* rstd, float* sums /* [2C] out */,
                              float* ws /* [s2t_bn_bwd_partials(rows)][2C] scratch */, float* dgamma, float* dbeta,
                              float count, int act,
                              int64_t rows, int C, const int32_t* lens, int T, void* stream) {
  if (!D || !dOut || !scale || !shift || !mean || !rstd || !sums || !ws || rows <= 0 || C <= 0 || C % 4) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int slices = s2t_bn_bwd_partials(rows);
  dim3 rgrid((C + 255) / 256, (unsigned)slices), block(256);
  dim3 agrid((unsigned)((rows * (C / 4) + 255) / 256));
  dim3 fgrid((2 * C + 15) / 16), fblock(1024);
  if (dtype == S2T_F32) {
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<float>, rgrid, block, 0, s, (const float*)D, (const float*)dOut, scale, shift, mean, rstd, act, rows, C, lens, T, ws);
    hipLaunchKernelGGL(bn_bwd_fold_kernel, fgrid, fblock, 0, s, ws, slices, C, sums, dgamma, dbeta);
    if (dD) hipLaunchKernelGGL(bn_act_bwd_apply_kernel<float>, agrid, block, 0, s, (const float*)D, (const float*)dOut, (float*)dD, scale, shift, mean, rstd, sums, count, act, rows, C, lens, T);
  } else if (dtype == S2T_BF16) {
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<bf16_t>, rgrid, block, 0, s, (const bf16_t*)D, (const bf16_t*)dOut, scale, shift, mean, rstd, act, rows, C, lens, T, ws);
    hipLaunchKernelGGL(bn_bwd_fold_kernel, fgrid, fblock, 0, s, ws, slices, C, sums, dgamma, dbeta);
    if (dD) hipLaunchKernelGGL(bn_act_bwd_apply_kernel<bf16_t>, agrid, block, 0, s, (const bf16_t*)D, (const bf16_t*)dOut, (bf16_t*)dD, scale, shift, mean, rstd, sums, count, act, rows, C, lens, T);
  } else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}
