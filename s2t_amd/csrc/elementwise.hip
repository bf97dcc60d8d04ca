// HBM-bound elementwise / gather / reduction kernels of the S2T path.  All: fp32 arithmetic, 4-element
// (8-byte bf16 / 16-byte f32) vectors per lane, consecutive lanes on consecutive addresses.
#include "common.h"

namespace {

// ---- x[b,t,:] = scale * x[b,t,:] + (t < len[b] ? tab[t + pos_offset] : 0) ; optional row mask of x first ----
// Encoder position step (s2t_transformer.py:1765-1787): sinusoidal positions of NON-PAD frames start at
// padding_idx+1 = 2, padded frames get the zero row.
template <typename T>
__global__ __launch_bounds__(256) void add_pos_kernel(T* __restrict__ x, const float* __restrict__ tab,
                                                      const int32_t* __restrict__ lens, int64_t rows, int Tn, int d,
                                                      float scale, int pos_offset) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vec_per_row = d / 4;
  rows = s2t_live_rows(lens, Tn, rows);
  if (idx >= rows * vec_per_row) return;
  const int64_t row = idx / vec_per_row;
  const int c = (int)(idx % vec_per_row) * 4;
  int t;
  bool valid;
  if (Tn > 0) {
    const int b = (int)(row / Tn);
    t = (int)(row % Tn);
    valid = !lens || t < lens[b];
  } else {  // packed rows: the row map holds the frame index
    const int e = lens[row];
    t = e & 0xffff;
    valid = e >= 0;
  }
  float v[4];
  ld4_as_f32<T>(x + row * d + c, v);
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid && tab) ld4_as_f32<float>(tab + (int64_t)(t + pos_offset) * d + c, p);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = v[r] * scale + p[r];
  st4_from_f32<T>(x + row * d + c, v);
}

template <typename T>
__global__ __launch_bounds__(256) void mask_rows_kernel(T* __restrict__ x, const int32_t* __restrict__ lens,
                                                        int64_t rows, int Tn, int d) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vec_per_row = d / 4;
  rows = s2t_live_rows(lens, Tn, rows);
  if (idx >= rows * vec_per_row) return;
  const int64_t row = idx / vec_per_row;
  const int c = (int)(idx % vec_per_row) * 4;
  if (s2t_row_masked(lens, Tn, row)) {
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    st4_from_f32<T>(x + row * d + c, z);
  }
}

// ---- out[m, :n] = x[m, :n] + bias[:n] (strided rows) ----
template <typename T>
__global__ __launch_bounds__(256) void bias_add_rows_kernel(const T* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ bias, T* __restrict__ out,
                                                            int64_t ldo, int64_t rows, int n) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vpr = n / 4;
  if (idx >= rows * vpr) return;
  const int64_t row = idx / vpr;
  const int c = (int)(idx % vpr) * 4;
  float v[4], b[4];
  ld4_as_f32<T>(x + row * ldx + c, v);
  ld4_as_f32<float>(bias + c, b);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] += b[r];
  st4_from_f32<T>(out + row * ldo + c, v);
}

// ---- inverted dropout on a strided row matrix (mask index = row*cols + col, the GEMM epilogue's convention) ----
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ out,
                                                      int64_t ldo, int64_t rows, int cols, float p,
                                                      const uint64_t* __restrict__ seed, uint32_t site) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vpr = cols / 4;
  if (idx >= rows * vpr) return;
  const int64_t row = idx / vpr;
  const int c = (int)(idx % vpr) * 4;
  const uint64_t key = s2t_drop_key(seed, site);
  const uint32_t th = s2t_drop_thresh(p);
  const float inv = s2t_drop_scale(p);
  float v[4];
  ld4_as_f32<T>(x + row * ldx + c, v);
  uint32_t r16[4];
  s2t_rand_run<4>(key, (uint64_t)row * cols + c, r16);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = r16[r] >= th ? v[r] * inv : 0.f;
  st4_from_f32<T>(out + row * ldo + c, v);
}

// ---- decoder embedding: out[n,:] = scale * E[tok[n],:] + tab[pos[n],:]  (models/transformer.py:1304-1323) ----
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ tok, const int32_t* __restrict__ pos,
                                                        const T* __restrict__ E, const float* __restrict__ tab,
                                                        T* __restrict__ out, int64_t n, int d, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vec_per_row = d / 4;
  if (idx >= n * vec_per_row) return;
  const int64_t row = idx / vec_per_row;
  const int c = (int)(idx % vec_per_row) * 4;
  float e[4], p[4] = {0.f, 0.f, 0.f, 0.f};
  ld4_as_f32<T>(E + tok[row] * d + c, e);
  if (tab) ld4_as_f32<float>(tab + (int64_t)pos[row] * d + c, p);
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = e[r] * scale + p[r];
  st4_from_f32<T>(out + row * d + c, e);
}
// dE[tok[n],:] += scale * dOut[n,:]   (fp32 atomics; rows of one token collide only within a batch)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tok, const T* __restrict__ dout,
                                                        float* __restrict__ dE, int64_t n, int d, float scale,
                                                        int64_t pad_idx) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vec_per_row = d / 4;
  if (idx >= n * vec_per_row) return;
  const int64_t row = idx / vec_per_row;
  const int c = (int)(idx % vec_per_row) * 4;
  const int64_t t = tok[row];
  if (t == pad_idx) return;  // nn.Embedding(padding_idx): no gradient for the pad row
  float g[4];
  ld4_as_f32<T>(dout + row * d + c, g);
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(dE + t * d + c + r, g[r] * scale);
}

// ---- GLU backward: Z = [a | g] (M x 2n), dY (M x n) -> dZ (M x 2n); optional padded-row mask ----
template <typename T>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const T* __restrict__ Z, const T* __restrict__ dY,
                                                      T* __restrict__ dZ, int64_t rows, int n,
                                                      const int32_t* __restrict__ lens, int Tn, int out_pad) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int vec_per_row = n / 4;
  if (idx >= rows * vec_per_row) return;
  const int64_t row = idx / vec_per_row;
  const int c = (int)(idx % vec_per_row) * 4;
  float a[4], g[4], dy[4], da[4], dg[4];
  ld4_as_f32<T>(Z + row * 2 * n + c, a);
  ld4_as_f32<T>(Z + row * 2 * n + n + c, g);
  ld4_as_f32<T>(dY + row * n + c, dy);
  const bool masked = lens && s2t_row_masked(lens, Tn, row);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s = sigmoidf_(g[r]);
    const float d = masked ? 0.f : dy[r];
    da[r] = d * s;
    dg[r] = d * a[r] * s * (1.f - s);
  }
  // out_pad: dZ keeps `out_pad` extra rows behind every block of Tn rows (left untouched here)
  const int64_t orow = out_pad ? row + (row / Tn) * out_pad : row;
  st4_from_f32<T>(dZ + orow * 2 * n + c, da);
  st4_from_f32<T>(dZ + orow * 2 * n + n + c, dg);
}

// bf16 fast path for row widths of 256 / 512 / 1024 / 2048: one 16-byte load per lane (8 columns), TPR = n / 8 lanes per
// row, 256 / TPR rows per pass, eight passes in flight; the rows are cut into at most 128 slices so that the closing
// atomics (n per workgroup, executed at the memory side) stay few.
template <int TPR>
__global__ __launch_bounds__(256) void colsum_bf16x8_kernel(const bf16_t* __restrict__ dY, int64_t ld,
                                                            float* __restrict__ db, int64_t rows) {
  constexpr int RPP = 256 / TPR;  // rows per pass
  __shared__ float red[RPP][TPR * 8];
  const int cl = threadIdx.x % TPR, rg = threadIdx.x / TPR;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(rows, r0 + per);
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const bf16_t* base = dY + cl * 8;
  int64_t m = r0 + rg;
  for (; m + 7 * RPP < r1; m += 8 * RPP) {
    uint4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const uint4*>(base + (m + u * RPP) * ld);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t w4[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[2 * q] += __uint_as_float(w4[q] << 16);
        acc[2 * q + 1] += __uint_as_float(w4[q] & 0xffff0000u);
      }
    }
  }
  for (; m < r1; m += RPP) {
    const uint4 t = *reinterpret_cast<const uint4*>(base + m * ld);
    const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[2 * q] += __uint_as_float(w4[q] << 16);
      acc[2 * q + 1] += __uint_as_float(w4[q] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rg][cl * 8 + i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < TPR * 8; c += 256) {
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < RPP; ++g) sum += red[g][c];
    atomicAdd(db + c, sum);
  }
}

// Relative-position attention backward glue (espnet_multihead_attention.py:313-356, autograd of (q + pos_bias_u) and
// (q + pos_bias_v)): a[row][:] += b[row][:] in place, du[c] += sum_rows a_old, dv[c] += sum_rows b — one pass instead of two
// column-sum launches and an element-wise add.  bf16 rows of 256 columns (8 per lane, 32 lanes per row).
__global__ __launch_bounds__(256) void add_colsum2_bf16_kernel(bf16_t* __restrict__ a, int64_t lda,
                                                               const bf16_t* __restrict__ b, int64_t ldb,
                                                               float* __restrict__ du, float* __restrict__ dv, int64_t rows) {
  constexpr int TPR = 32, RPP = 8;
  __shared__ float red[2][RPP][256];
  const int cl = threadIdx.x % TPR, rg = threadIdx.x / TPR;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(rows, r0 + per);
  float su[8], sv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) su[i] = sv[i] = 0.f;
  auto one = [&](const uint4 ta, const uint4 tb, bf16_t* dst) __attribute__((always_inline)) {
    const uint32_t wa[4] = {ta.x, ta.y, ta.z, ta.w}, wb[4] = {tb.x, tb.y, tb.z, tb.w};
    float o[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a0 = __uint_as_float(wa[q] << 16), a1 = __uint_as_float(wa[q] & 0xffff0000u);
      const float b0 = __uint_as_float(wb[q] << 16), b1 = __uint_as_float(wb[q] & 0xffff0000u);
      su[2 * q] += a0;
      su[2 * q + 1] += a1;
      sv[2 * q] += b0;
      sv[2 * q + 1] += b1;
      o[2 * q] = a0 + b0;
      o[2 * q + 1] = a1 + b1;
    }
    uint4 t;
    t.x = bf16pack(o[0], o[1]);
    t.y = bf16pack(o[2], o[3]);
    t.z = bf16pack(o[4], o[5]);
    t.w = bf16pack(o[6], o[7]);
    *reinterpret_cast<uint4*>(dst) = t;
  };
  int64_t m = r0 + rg;
  for (; m + 3 * RPP < r1; m += 4 * RPP) {
    uint4 ta[4], tb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ta[u] = *reinterpret_cast<const uint4*>(a + (m + u * RPP) * lda + cl * 8);
      tb[u] = *reinterpret_cast<const uint4*>(b + (m + u * RPP) * ldb + cl * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(ta[u], tb[u], a + (m + u * RPP) * lda + cl * 8);
  }
  for (; m < r1; m += RPP)
    one(*reinterpret_cast<const uint4*>(a + m * lda + cl * 8), *reinterpret_cast<const uint4*>(b + m * ldb + cl * 8),
        a + m * lda + cl * 8);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    red[0][rg][cl * 8 + i] = su[i];
    red[1][rg][cl * 8 + i] = sv[i];
  }
  __syncthreads();
  const int c = threadIdx.x;
  float tu = 0.f, tv = 0.f;
#pragma unroll
  for (int g = 0; g < RPP; ++g) {
    tu += red[0][g][c];
    tv += red[1][g][c];
  }
  atomicAdd(du + c, tu);
  atomicAdd(dv + c, tv);
}

// ---- bias gradient: db[n] += sum_m dY[m,n]  (fp32 accumulate) ----
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dY, int64_t ld, float* __restrict__ db,
                                                     int64_t rows, int n) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < n) {
    const int64_t step = (int64_t)gridDim.y * 4;
    int64_t m = (int64_t)blockIdx.y * 4 + w;
    if (c + 3 < n) {
      // four rows per trip: four independent 8/16-byte loads in flight per lane (the pass is latency bound otherwise)
      float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f}, a3[4] = {0.f, 0.f, 0.f, 0.f};
      for (; m + 3 * step < rows; m += 4 * step) {
        float v0[4], v1[4], v2[4], v3[4];
        ld4_as_f32<T>(dY + m * ld + c, v0);
        ld4_as_f32<T>(dY + (m + step) * ld + c, v1);
        ld4_as_f32<T>(dY + (m + 2 * step) * ld + c, v2);
        ld4_as_f32<T>(dY + (m + 3 * step) * ld + c, v3);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[r] += v0[r];
          a1[r] += v1[r];
          a2[r] += v2[r];
          a3[r] += v3[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += a1[r] + a2[r] + a3[r];
    }
    for (; m < rows; m += step) {
      float v[4];
      if (c + 3 < n) ld4_as_f32<T>(dY + m * ld + c, v);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = c + r < n ? ld_as_f32<T>(dY + m * ld + c + r) : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += v[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][lane * 4 + r] = acc[r];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < n) atomicAdd(db + cc, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- fp32 -> bf16 cast of a flat buffer (bf16 shadow of the fp32 master weights) ----
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                        int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float v[4];
    ld4_as_f32<float>(src + i * 4, v);
    st4_from_f32<bf16_t>(dst + i * 4, v);
  }
}

// ---- y = a + alpha * b  (flat, same dtype) ----
template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                   float alpha, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float u[4], v[4];
    ld4_as_f32<T>(a + i * 4, u);
    ld4_as_f32<T>(b + i * 4, v);
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] += alpha * v[r];
    st4_from_f32<T>(y + i * 4, u);
  }
}

// ---- fused Adam on flat fp32 buffers (optim/adam.py:146-226) + bf16 shadow refresh ----
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p -= wd*lr*p ; p -= step_size * m / (sqrt(v) + eps)
//   g is pre-multiplied by grad_scale (clip coefficient x 1/sample_size).
#ifndef S2T_ADAM_NT
#define S2T_ADAM_NT 0  // 1: the optimizer state (p, m, v) streams past the caches (non-temporal loads and stores)
#endif
typedef float adam_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void adam_ld(const float* q, float (&v)[4]) {
#if S2T_ADAM_NT
  const adam_f4 t = __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(q));
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
#else
  ld4_as_f32<float>(q, v);
#endif
}
__device__ __forceinline__ void adam_st(float* q, const float (&v)[4]) {
#if S2T_ADAM_NT
  __builtin_nontemporal_store((adam_f4){v[0], v[1], v[2], v[3]}, reinterpret_cast<adam_f4*>(q));
#else
  st4_from_f32<float>(q, v);
#endif
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ shadow, int64_t n4, float b1, float b2,
                                                   float eps, float wd, const float* __restrict__ hyper) {
  // hyper[0] = lr, hyper[1] = step_size = lr*sqrt(1-b2^t)/(1-b1^t), hyper[2] = grad_scale (device-resident so a
  // captured hipGraph replays with fresh values)
  const float lr = hyper[0], step_size = hyper[1], grad_scale = hyper[2];
  // The pass is pure streaming (30 bytes per parameter, nothing re-read): TWO 16-byte pieces per tensor and trip, all eight
  // loads issued before the first use — one piece per trip left four loads in flight per lane and the pass at 3.6 TB/s.
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    const int64_t i1 = i0 + stride;
    const bool two = i1 < n4;
    const int64_t j1 = two ? i1 : i0;  // (clamped: the second piece's loads stay unconditional)
    float pv[2][4], gv[2][4], mv[2][4], vv[2][4];
    adam_ld(p + i0 * 4, pv[0]);
    ld4_as_f32<float>(g + i0 * 4, gv[0]);
    adam_ld(m + i0 * 4, mv[0]);
    adam_ld(v + i0 * 4, vv[0]);
    adam_ld(p + j1 * 4, pv[1]);
    ld4_as_f32<float>(g + j1 * 4, gv[1]);
    adam_ld(m + j1 * 4, mv[1]);
    adam_ld(v + j1 * 4, vv[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gr = gv[u][r] * grad_scale;
        mv[u][r] = b1 * mv[u][r] + (1.f - b1) * gr;
        vv[u][r] = b2 * vv[u][r] + (1.f - b2) * gr * gr;
        float pr = pv[u][r];
        if (wd != 0.f) pr -= wd * lr * pr;
        pr -= step_size * mv[u][r] / (sqrtf(vv[u][r]) + eps);
        pv[u][r] = pr;
      }
      if (u == 1 && !two) break;
      const int64_t i = u ? i1 : i0;
      adam_st(p + i * 4, pv[u]);
      adam_st(m + i * 4, mv[u]);
      adam_st(v + i * 4, vv[u]);
      if (shadow) st4_from_f32<bf16_t>(shadow + i * 4, pv[u]);
    }
  }
}

// ---- sum of squares of a flat fp32 buffer (grad-norm for clipping, utils.py:328-369) ----
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n4, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ld4_as_f32<float>(g + (i + u * stride) * 4, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
  }
  for (; i < n4; i += stride) {
    float v[4];
    ld4_as_f32<float>(g + i * 4, v);
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

inline unsigned flat_grid(int64_t work_items) {
  int64_t nb = (work_items + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

}  // namespace

extern "C" int s2t_add_positions(int dtype, void* x, const float* tab, const int32_t* lens, int64_t rows, int T, int d,
                                 float scale, int pos_offset, void* stream) {
  if (!x || rows < 0 || (T <= 0 && !(T == S2T_ROWS_PACKED && lens)) || d <= 0 || d % 4) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows * (d / 4) + 255) / 256));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(add_pos_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (float*)x, tab, lens, rows, T, d, scale, pos_offset);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(add_pos_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, tab, lens, rows, T, d, scale, pos_offset);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_mask_rows(int dtype, void* x, const int32_t* lens, int64_t rows, int T, int d, void* stream) {
  if (!x || !lens || rows < 0 || (T <= 0 && T != S2T_ROWS_PACKED) || d <= 0 || d % 4) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows * (d / 4) + 255) / 256));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(mask_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (float*)x, lens, rows, T, d);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(mask_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, lens, rows, T, d);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_bias_add_rows(int dtype, const void* x, int64_t ldx, const float* bias, void* out, int64_t ldo,
                                 int64_t rows, int n, void* stream) {
  if (!x || !bias || !out || rows < 0 || n <= 0 || n % 4 || ldx % 4 || ldo % 4) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows * (n / 4) + 255) / 256));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(bias_add_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, bias, (float*)out, ldo, rows, n);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(bias_add_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, bias, (bf16_t*)out, ldo, rows, n);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_dropout(int dtype, const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int cols, float p,
                           const uint64_t* seed, uint32_t site, void* stream) {
  if (!x || !out || rows < 0 || cols <= 0 || cols % 4 || ldx % 4 || ldo % 4 || p < 0.f || p >= 1.f) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows * (cols / 4) + 255) / 256));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)out, ldo, rows, cols, p, seed, site);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, rows, cols, p, seed, site);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_embedding_fwd(int dtype, const int64_t* tokens, const int32_t* pos, const void* E, const float* tab,
                                 void* out, int64_t n, int d, float scale, void* stream) {
  if (!tokens || !E || !out || n < 0 || d <= 0 || d % 4 || (tab && !pos)) return S2T_ERR_ARG;
  if (n == 0) return S2T_OK;
  dim3 grid((unsigned)((n * (d / 4) + 255) / 256));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(embed_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, tokens, pos, (const float*)E, tab, (float*)out, n, d, scale);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, tokens, pos, (const bf16_t*)E, tab, (bf16_t*)out, n, d, scale);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_embedding_bwd(int dtype, const int64_t* tokens, const void* dout, float* dE, int64_t n, int d,
                                 float scale, int64_t pad_idx, void* stream) {
  if (!tokens || !dout || !dE || n < 0 || d <= 0 || d % 4) return S2T_ERR_ARG;
  if (n == 0) return S2T_OK;
  dim3 grid((unsigned)((n * (d / 4) + 255) / 256));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, tokens, (const float*)dout, dE, n, d, scale, pad_idx);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, tokens, (const bf16_t*)dout, dE, n, d, scale, pad_idx);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_glu_bwd(int dtype, const void* Z, const void* dY, void* dZ, int64_t rows, int n,
                           const int32_t* lens, int T, int out_pad, void* stream) {
  if (!Z || !dY || !dZ || rows < 0 || n <= 0 || n % 4 || out_pad < 0 || (out_pad > 0 && T <= 0)) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows * (n / 4) + 255) / 256));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(glu_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)Z, (const float*)dY, (float*)dZ, rows, n, lens, T, out_pad);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(glu_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Z, (const bf16_t*)dY, (bf16_t*)dZ, rows, n, lens, T, out_pad);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_colsum_accum(int dtype, const void* dY, int64_t ld, float* db, int64_t rows, int n, void* stream) {
  if (!dY || !db || rows < 0 || n <= 0) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  if (ld % 4 || ((uintptr_t)dY % 16)) return S2T_ERR_ALIGN;
  if (dtype == S2T_BF16 && ld % 8 == 0 && (n == 256 || n == 512 || n == 1024 || n == 2048) && rows >= 512) {
    int64_t sl = (rows + 63) / 64;
    if (sl > 128) sl = 128;
    dim3 g((unsigned)sl);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* src = (const bf16_t*)dY;
    if (n == 256) hipLaunchKernelGGL(colsum_bf16x8_kernel<32>, g, dim3(256), 0, st, src, ld, db, rows);
    else if (n == 512) hipLaunchKernelGGL(colsum_bf16x8_kernel<64>, g, dim3(256), 0, st, src, ld, db, rows);
    else if (n == 1024) hipLaunchKernelGGL(colsum_bf16x8_kernel<128>, g, dim3(256), 0, st, src, ld, db, rows);
    else hipLaunchKernelGGL(colsum_bf16x8_kernel<256>, g, dim3(256), 0, st, src, ld, db, rows);
    return S2T_LAUNCH_CHECK();
  }
  int64_t slices = (rows + 31) / 32;  // 8 rows per wave
  if (slices > 1024) slices = 1024;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)slices);
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dY, ld, db, rows, n);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dY, ld, db, rows, n);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_add_colsum2(int dtype, void* a, int64_t lda, const void* b, int64_t ldb, float* du, float* dv,
                               int64_t rows, int n, void* stream) {
  if (!a || !b || !du || !dv || rows < 0 || n <= 0) return S2T_ERR_ARG;
  if (dtype != S2T_BF16 || n != 256) return S2T_ERR_UNSUPPORTED;
  if (lda % 8 || ldb % 8 || ((uintptr_t)a % 16) || ((uintptr_t)b % 16)) return S2T_ERR_ALIGN;
  if (rows == 0) return S2T_OK;
  int64_t sl = (rows + 63) / 64;
  if (sl > 128) sl = 128;
  hipLaunchKernelGGL(add_colsum2_bf16_kernel, dim3((unsigned)sl), dim3(256), 0, (hipStream_t)stream, (bf16_t*)a, lda,
                     (const bf16_t*)b, ldb, du, dv, rows);
  return S2T_LAUNCH_CHECK();
}

// dst[i] = scale * float(src[i]) for n4 groups of four bf16 values
__global__ __launch_bounds__(256) void cast_bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst,
                                                               int64_t n4, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint2 v = *reinterpret_cast<const uint2*>(src + 4 * i);
    *reinterpret_cast<float4*>(dst + 4 * i) = make_float4(scale * __uint_as_float(v.x << 16), scale * __uint_as_float(v.x & 0xffff0000u),
                                                          scale * __uint_as_float(v.y << 16), scale * __uint_as_float(v.y & 0xffff0000u));
  }
}
extern "C" int s2t_cast_bf16_to_f32(const void* src, float* dst, int64_t n, float scale, void* stream) {
  if (!src || !dst || n < 0 || n % 4) return S2T_ERR_ARG;
  if (n == 0) return S2T_OK;
  hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     dst, n / 4, scale);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0 || n % 4) return S2T_ERR_ARG;
  if (n == 0) return S2T_OK;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n / 4);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_axpy(int dtype, const void* a, const void* b, void* y, float alpha, int64_t n, void* stream) {
  if (!a || !b || !y || n < 0 || n % 4) return S2T_ERR_ARG;
  if (n == 0) return S2T_OK;
  dim3 grid(flat_grid(n / 4));
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(axpy_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)y, alpha, n / 4);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(axpy_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, alpha, n / 4);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_adam_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, int64_t n, float beta1,
                             float beta2, float eps, float weight_decay, const float* hyper, void* stream) {
  if (!p || !g || !m || !v || !hyper || n < 0 || n % 4) return S2T_ERR_ARG;
  if (n == 0) return S2T_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (bf16_t*)bf16_shadow, n / 4, beta1, beta2, eps, weight_decay, hyper);
  return S2T_LAUNCH_CHECK();
}

// hyper[2] = mult * min(1, max_norm / (sqrt(sumsq)*mult + 1e-6)) ; hyper[3] = sqrt(sumsq)*mult (the reported grad norm)
// (trainer.py:729-741: grads are multiplied by world/sample_size, then clipped by their global norm, utils.py:328-369)
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float mult, float* __restrict__ hyper) {
  if (mult <= 0.f) mult = hyper[2];  // the normaliser travels in the hyper row (a captured step replays with a new one)
  const float norm = sqrtf(sumsq[0]) * mult;
  float c = 1.f;
  if (max_norm > 0.f) c = fminf(1.f, max_norm / (norm + 1e-6f));
  hyper[2] = mult * c;
  hyper[3] = norm;
}
extern "C" int s2t_clip_coef(const float* sumsq, float max_norm, float mult, float* hyper, void* stream) {
  if (!sumsq || !hyper) return S2T_ERR_ARG;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, mult, hyper);
  return S2T_LAUNCH_CHECK();
}

// n bf16 matrices transposed by one launch: 64 x 64 tiles, one workgroup per entry of the tile list (matrix, row tile,
// column tile) the host made — a grid sized by the largest extents launched eight empty workgroups per busy one.
// Fast path (both extents multiples of 8, 16-byte aligned bases): every thread moves two 16-byte pieces each way — rows of
// the tile go into LDS as they are ([row][col], 136-byte pitch), the transposed rows are gathered from it eight 2-byte reads
// per output piece.  Anything else takes the 4-byte path (2 x 2 element blocks turned in registers).
__global__ __launch_bounds__(256) void transpose_batched_kernel(const s2t_transpose_item* __restrict__ items,
                                                                const int32_t* __restrict__ tiles) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 68];
  const int32_t* td = tiles + 3 * (int64_t)blockIdx.x;
  const s2t_transpose_item it = items[td[0]];
  const int r0 = td[1] * 64, c0 = td[2] * 64;
  if (r0 >= it.rows || c0 >= it.cols) return;
  const bf16_t* src = reinterpret_cast<const bf16_t*>(it.src);
  bf16_t* dst = reinterpret_cast<bf16_t*>(it.dst);
  const bool fast = (it.rows % 8) == 0 && (it.cols % 8) == 0 && (((uintptr_t)it.src | (uintptr_t)it.dst) & 15) == 0;
  if (fast) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int v = threadIdx.x + 256 * ps;   // 512 pieces: row v >> 3, 8-column piece v & 7
      const int r = v >> 3, pc = v & 7;
      uint4 q = make_uint4(0, 0, 0, 0);
      if (r0 + r < it.rows && c0 + 8 * pc < it.cols) q = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + r) * it.cols + c0 + 8 * pc);
      *reinterpret_cast<uint2*>(tile + r * 68 + 8 * pc) = make_uint2(q.x, q.y);
      *reinterpret_cast<uint2*>(tile + r * 68 + 8 * pc + 4) = make_uint2(q.z, q.w);
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int v = threadIdx.x + 256 * ps;   // output row (input column) v & 63, 8-row piece v >> 6
      const int oc = v & 63, pr = v >> 6;
      if (c0 + oc < it.cols && r0 + 8 * pr < it.rows) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          w[k] = (uint32_t)tile[(8 * pr + 2 * k) * 68 + oc] | ((uint32_t)tile[(8 * pr + 2 * k + 1) * 68 + oc] << 16);
        *reinterpret_cast<uint4*>(dst + (int64_t)(c0 + oc) * it.rows + r0 + 8 * pr) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return;
  }
  uint32_t* t32 = reinterpret_cast<uint32_t*>(tile);  // [64 output rows][33 pairs of output columns]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const bool src32 = (it.cols & 1) == 0, dst32 = (it.rows & 1) == 0;
  auto ld2 = [&](int r, int c) -> uint32_t {  // elements (r, c) | (r, c + 1) << 16, zero outside
    if (r >= it.rows || c >= it.cols) return 0u;
    const bf16_t* q = src + (int64_t)r * it.cols + c;
    if (src32 && c + 1 < it.cols) return *reinterpret_cast<const uint32_t*>(q);
    return (uint32_t)q[0] | (c + 1 < it.cols ? (uint32_t)q[1] << 16 : 0u);
  };
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int rp = 8 * ps + ty;
    const uint32_t a = ld2(r0 + 2 * rp, c0 + 2 * tx), b = ld2(r0 + 2 * rp + 1, c0 + 2 * tx);
    t32[(2 * tx) * 33 + rp] = (a & 0xffffu) | (b << 16);
    t32[(2 * tx + 1) * 33 + rp] = (a >> 16) | (b & 0xffff0000u);
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    const int orow = 8 * ps + ty;
    const int c = c0 + orow, r = r0 + 2 * tx;  // output row c, output columns r, r + 1
    if (c < it.cols && r < it.rows) {
      const uint32_t v = t32[orow * 33 + tx];
      bf16_t* q = dst + (int64_t)c * it.rows + r;
      if (dst32 && r + 1 < it.rows) *reinterpret_cast<uint32_t*>(q) = v;
      else {
        q[0] = (bf16_t)(v & 0xffffu);
        if (r + 1 < it.rows) q[1] = (bf16_t)(v >> 16);
      }
    }
  }
}
extern "C" int s2t_transpose_bf16_batched(const s2t_transpose_item* items_dev, int n, const int32_t* tiles_dev, int n_tiles,
                                          void* stream) {
  if (!items_dev || n <= 0 || !tiles_dev || n_tiles <= 0) return S2T_ERR_ARG;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, items_dev, tiles_dev);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_sumsq_accum(const float* g, int64_t n, float* out, void* stream) {
  if (!g || !out || n < 0 || n % 4) return S2T_ERR_ARG;
  if (n == 0) return S2T_OK;
  int64_t nb = (n / 4 + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, n / 4, out);
  return S2T_LAUNCH_CHECK();
}
