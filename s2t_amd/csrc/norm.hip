// LayerNorm forward / backward (reference: modules/layer_norm.py:30-35 -> torch.nn.LayerNorm, eps 1e-5).
// HBM-bound: one wavefront per row, 4 elements (8 B bf16 / 16 B f32) per lane per step, fp32 statistics by
// wave shuffles.  Optional fused row mask (rows t >= len[b] are written as zero) covers the reference's
// per-layer `masked_fill(pad, 0)` (s2t_transformer.py:1828-1836) when it directly follows a LayerNorm.
#include "common.h"

namespace {

constexpr int LN_MAX_VEC = 8;  // supports cols <= 64*4*8 = 2048 (kernels are instantiated for 1/2/4/8 vectors per lane)

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int64_t rows, int cols, float eps,
                                                     const int32_t* __restrict__ row_lens, int row_T) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  rows = s2t_live_rows(row_lens, row_T, rows);  // packed batch: rows beyond the live ones are not touched
  if (row >= rows) return;
  const T* xr = x + row * cols;
  float v[NV][4];
  constexpr int nvec = NV;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nvec) {
      const int c = i * 256 + lane * 4;
      if (c < cols) ld4_as_f32<T>(xr + c, v[i]);
      else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nvec) {
      const int c = i * 256 + lane * 4;
      if (c < cols) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = v[i][r] - mean;
          q += d * d;
        }
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / cols + eps);
  bool masked = false;
  if (row_lens) masked = s2t_row_masked(row_lens, row_T, row);
  T* yr = y + row * cols;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nvec) {
      const int c = i * 256 + lane * 4;
      if (c < cols) {
        float g[4], b[4], o[4];
        ld4_as_f32<float>(gamma + c, g);
        ld4_as_f32<float>(beta + c, b);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = masked ? 0.f : (v[i][r] - mean) * rstd * g[r] + b[r];
        st4_from_f32<T>(yr + c, o);
      }
    }
  }
  if (lane == 0 && mean_out) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// dx = rstd * (dy*g - mean_c(dy*g) - xhat * mean_c(dy*g*xhat)); dgamma += sum_rows dy*xhat; dbeta += sum_rows dy
// A fixed grid of workgroups strides over the rows so that the per-column partial sums stay in registers and
// each workgroup issues ONE set of atomics (avoids a chip-wide pile-up on the same 2*cols addresses).
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const T* __restrict__ dy, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                                     const T* __restrict__ dres, float* __restrict__ ws, int replicas,
                                                     int64_t rows, int cols, const int32_t* __restrict__ row_lens,
                                                     int row_T) {
  __shared__ float red[2][4][NV * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  rows = s2t_live_rows(row_lens, row_T, rows);
  constexpr int nvec = NV;
  float ag[NV][4], ab[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) ag[i][r] = ab[i][r] = 0.f;

  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
    bool masked = false;
    if (row_lens) masked = s2t_row_masked(row_lens, row_T, row);
    const float mu = mean[row], rs = rstd[row];
    float xh[NV][4], dg[NV][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nvec) {
        const int c = i * 256 + lane * 4;
        if (c < cols) {
          float xv[4], dv[4], g[4];
          ld4_as_f32<T>(x + row * cols + c, xv);
          ld4_as_f32<T>(dy + row * cols + c, dv);
          ld4_as_f32<float>(gamma + c, g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = masked ? 0.f : dv[r];
            xh[i][r] = (xv[r] - mu) * rs;
            dg[i][r] = d * g[r];
            s1 += dg[i][r];
            s2 += dg[i][r] * xh[i][r];
            ag[i][r] += d * xh[i][r];
            ab[i][r] += d;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) xh[i][r] = dg[i][r] = 0.f;
        }
      }
    }
    s1 = wave_sum(s1) / cols;
    s2 = wave_sum(s2) / cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nvec) {
        const int c = i * 256 + lane * 4;
        if (c < cols) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = rs * (dg[i][r] - s1 - xh[i][r] * s2);
          if (dres) {  // gradient arriving at x through the residual branch, added here instead of by a separate kernel
            float q[4];
            ld4_as_f32<T>(dres + row * cols + c, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += q[r];
          }
          st4_from_f32<T>(dx + row * cols + c, o);
        }
      }
    }
  }
  // block reduce over the 4 waves, then one atomic per column per workgroup
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nvec) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[0][w][i * 256 + lane * 4 + r] = ag[i][r];
        red[1][w][i * 256 + lane * 4 + r] = ab[i][r];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float g = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    // replica (blockIdx % replicas) of the [2][cols] partial sums: at most gridDim/replicas adders per address
    float* w = ws + (int64_t)(blockIdx.x % replicas) * 2 * cols;
    atomicAdd(w + c, g);
    atomicAdd(w + cols + c, b);
  }
}

// ---- cols == 256, bf16: half a wave per row (32 lanes x 8 elements = one 16-byte access per lane), two rows per
// wave: half the instructions per row and twice the bytes per memory instruction of the generic kernels, which are
// issue/latency bound on 512-byte rows (the model width of every recipe is 256).
__device__ __forceinline__ float half_sum(float v) {
  v = s2t_sum32(v);
  return v;
}
__device__ __forceinline__ void unpack8(const uint4 t, float (&v)[8]) {
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __uint_as_float(w[q] << 16);
    v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8f(const float (&o)[8]) {
  uint4 t;
  t.x = bf16pack(o[0], o[1]);
  t.y = bf16pack(o[2], o[3]);
  t.z = bf16pack(o[4], o[5]);
  t.w = bf16pack(o[6], o[7]);
  return t;
}

__global__ __launch_bounds__(256) void ln256_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        int64_t rows, float eps, const int32_t* __restrict__ row_lens,
                                                        int row_T) {
  // two rows per half-wave, both 16-byte loads issued before either is used (the pass is latency bound: one row per
  // half-wave leaves a single load in flight per lane)
  constexpr int RPH = 2;
  const int lane = threadIdx.x & 63, l = lane & 31;
  rows = s2t_live_rows(row_lens, row_T, rows);
  if ((int64_t)blockIdx.x * 16 >= rows) return;  // (workgroup-uniform: a packed batch fills only part of the grid)
  const int64_t row0 = ((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + (lane >> 5)) * RPH;
  float v[RPH][8];
  bool valid[RPH];
#pragma unroll
  for (int u = 0; u < RPH; ++u) {
    valid[u] = row0 + u < rows;
    const int64_t rr = valid[u] ? row0 + u : rows - 1;
    unpack8(*reinterpret_cast<const uint4*>(x + rr * 256 + l * 8), v[u]);
  }
  float g[8], b[8];
  ld4_as_f32<float>(gamma + l * 8, reinterpret_cast<float (&)[4]>(g[0]));
  ld4_as_f32<float>(gamma + l * 8 + 4, reinterpret_cast<float (&)[4]>(g[4]));
  ld4_as_f32<float>(beta + l * 8, reinterpret_cast<float (&)[4]>(b[0]));
  ld4_as_f32<float>(beta + l * 8 + 4, reinterpret_cast<float (&)[4]>(b[4]));
#pragma unroll
  for (int u = 0; u < RPH; ++u) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += v[u][r];
    const float mean = half_sum(s) * (1.f / 256.f);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float d = v[u][r] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(half_sum(q) * (1.f / 256.f) + eps);
    if (!valid[u]) continue;  // (after the wave-wide shuffles)
    const int64_t row = row0 + u;
    bool masked = false;
    if (row_lens) masked = s2t_row_masked(row_lens, row_T, row);
    float o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = masked ? 0.f : (v[u][r] - mean) * rstd * g[r] + b[r];
    *reinterpret_cast<uint4*>(y + row * 256 + l * 8) = pack8f(o);
    if (l == 0 && mean_out) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// cols = 512 (the NAST recipe's width): a row is one 16-byte piece per lane; four rows per wave, all four loads issued before
// any is used (the generic kernel's one row per wave in 8-byte pieces reads at 3.5 TB/s)
__global__ __launch_bounds__(256) void ln512_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        int64_t rows, float eps, const int32_t* __restrict__ row_lens,
                                                        int row_T) {
  constexpr int RPW = 4;
  const int lane = threadIdx.x & 63;
  rows = s2t_live_rows(row_lens, row_T, rows);
  if ((int64_t)blockIdx.x * 16 >= rows) return;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  float v[RPW][8];
  bool valid[RPW];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    valid[u] = row0 + u < rows;
    const int64_t rr = valid[u] ? row0 + u : rows - 1;
    unpack8(*reinterpret_cast<const uint4*>(x + rr * 512 + lane * 8), v[u]);
  }
  float g[8], b[8];
  ld4_as_f32<float>(gamma + lane * 8, reinterpret_cast<float (&)[4]>(g[0]));
  ld4_as_f32<float>(gamma + lane * 8 + 4, reinterpret_cast<float (&)[4]>(g[4]));
  ld4_as_f32<float>(beta + lane * 8, reinterpret_cast<float (&)[4]>(b[0]));
  ld4_as_f32<float>(beta + lane * 8 + 4, reinterpret_cast<float (&)[4]>(b[4]));
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += v[u][r];
    const float mean = wave_sum(s) * (1.f / 512.f);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float d = v[u][r] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / 512.f) + eps);
    if (!valid[u]) continue;  // (after the wave-wide shuffles)
    const int64_t row = row0 + u;
    bool masked = false;
    if (row_lens) masked = s2t_row_masked(row_lens, row_T, row);
    float o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = masked ? 0.f : (v[u][r] - mean) * rstd * g[r] + b[r];
    *reinterpret_cast<uint4*>(y + row * 512 + lane * 8) = pack8f(o);
    if (lane == 0 && mean_out) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

__global__ __launch_bounds__(256) void ln256_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const bf16_t* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                        const bf16_t* __restrict__ dres, float* __restrict__ ws,
                                                        int replicas, int64_t rows, const int32_t* __restrict__ row_lens,
                                                        int row_T, bf16_t* __restrict__ dx_drop, float drop_p,
                                                        const uint64_t* __restrict__ drop_seed, uint32_t drop_site) {
  __shared__ float red[2][8][256];
  const int lane = threadIdx.x & 63, l = lane & 31, hw = (threadIdx.x >> 6) * 2 + (lane >> 5);
  rows = s2t_live_rows(row_lens, row_T, rows);
  float g[8];
  ld4_as_f32<float>(gamma + l * 8, reinterpret_cast<float (&)[4]>(g[0]));
  ld4_as_f32<float>(gamma + l * 8 + 4, reinterpret_cast<float (&)[4]>(g[4]));
  float ag[8], ab[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) ag[r] = ab[r] = 0.f;
  // all 64 lanes of a wave take the same number of trips (shuffles are wave-wide): out-of-range halves work on a
  // clamped row and discard the result
  // two rows per half-wave and trip; x, dy and dres of both rows are requested before any of them is used
  constexpr int RPH = 2;
  const int64_t per_trip = (int64_t)gridDim.x * 8 * RPH;
  const int64_t trips = (rows + per_trip - 1) / per_trip;
  for (int64_t it = 0; it < trips; ++it) {
    const int64_t row0 = ((it * gridDim.x + blockIdx.x) * 8 + hw) * RPH;
    uint4 tx[RPH], td[RPH], tr[RPH];
    float mu[RPH], rs[RPH];
    bool valid[RPH], masked[RPH];
#pragma unroll
    for (int u = 0; u < RPH; ++u) {
      const int64_t row = row0 + u;
      valid[u] = row < rows;
      const int64_t rr = valid[u] ? row : rows - 1;
      tx[u] = *reinterpret_cast<const uint4*>(x + rr * 256 + l * 8);
      td[u] = *reinterpret_cast<const uint4*>(dy + rr * 256 + l * 8);
      tr[u] = dres ? *reinterpret_cast<const uint4*>(dres + rr * 256 + l * 8) : make_uint4(0, 0, 0, 0);
      mu[u] = mean[rr];
      rs[u] = rstd[rr];
      masked[u] = !valid[u];
      if (valid[u] && row_lens) masked[u] = s2t_row_masked(row_lens, row_T, row);
    }
#pragma unroll
    for (int u = 0; u < RPH; ++u) {
      float xv[8], dv[8], xh[8], dg[8];
      unpack8(tx[u], xv);
      unpack8(td[u], dv);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float d = masked[u] ? 0.f : dv[r];
        xh[r] = (xv[r] - mu[u]) * rs[u];
        dg[r] = d * g[r];
        s1 += dg[r];
        s2 += dg[r] * xh[r];
        ag[r] += d * xh[r];
        ab[r] += d;
      }
      s1 = half_sum(s1) * (1.f / 256.f);
      s2 = half_sum(s2) * (1.f / 256.f);
      if (valid[u]) {
        float o[8], q[8];
        unpack8(tr[u], q);
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = rs[u] * (dg[r] - s1 - xh[r] * s2) + q[r];
        const uint4 packed = pack8f(o);
        *reinterpret_cast<uint4*>(dx + (row0 + u) * 256 + l * 8) = packed;
        if (dx_drop) {
          // second output: the inverted-dropout image of dx under the mask of the block that PRODUCED this LayerNorm's
          // input (its branch gradient), bit-identical to s2t_dropout applied to the stored bf16 dx
          float v[8];
          unpack8(packed, v);
          const uint64_t key = s2t_drop_key(drop_seed, drop_site);
          const uint32_t th = s2t_drop_thresh(drop_p);
          const float inv = s2t_drop_scale(drop_p);
          uint32_t r16[8];
          s2t_rand_run<8>(key, (uint64_t)(row0 + u) * 256 + l * 8, r16);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = r16[r] >= th ? v[r] * inv : 0.f;
          *reinterpret_cast<uint4*>(dx_drop + (row0 + u) * 256 + l * 8) = pack8f(v);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    red[0][hw][l * 8 + r] = ag[r];
    red[1][hw][l * 8 + r] = ab[r];
  }
  __syncthreads();
  const int c = threadIdx.x;
  float sg = 0.f, sb = 0.f;
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    sg += red[0][h][c];
    sb += red[1][h][c];
  }
  float* w = ws + (int64_t)(blockIdx.x % replicas) * 512;
  atomicAdd(w + c, sg);
  atomicAdd(w + 256 + c, sb);
}

// cols = 512: a wave per row (16-byte pieces), two rows per wave and trip with every load of both requested first; the column
// sums of dgamma / dbeta stay in registers over the trips and leave as one set of atomics per workgroup (as above)
__global__ __launch_bounds__(256) void ln512_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const bf16_t* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                        const bf16_t* __restrict__ dres, float* __restrict__ ws,
                                                        int replicas, int64_t rows, const int32_t* __restrict__ row_lens,
                                                        int row_T) {
  __shared__ float red[2][4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  rows = s2t_live_rows(row_lens, row_T, rows);
  float g[8];
  ld4_as_f32<float>(gamma + lane * 8, reinterpret_cast<float (&)[4]>(g[0]));
  ld4_as_f32<float>(gamma + lane * 8 + 4, reinterpret_cast<float (&)[4]>(g[4]));
  float ag[8], ab[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) ag[r] = ab[r] = 0.f;
  constexpr int RPW = 2;
  const int64_t per_trip = (int64_t)gridDim.x * 4 * RPW;
  const int64_t trips = (rows + per_trip - 1) / per_trip;
  for (int64_t it = 0; it < trips; ++it) {
    const int64_t row0 = ((it * gridDim.x + blockIdx.x) * 4 + wv) * RPW;
    uint4 tx[RPW], td[RPW], tr[RPW];
    float mu[RPW], rs[RPW];
    bool valid[RPW], masked[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int64_t row = row0 + u;
      valid[u] = row < rows;
      const int64_t rr = valid[u] ? row : rows - 1;
      tx[u] = *reinterpret_cast<const uint4*>(x + rr * 512 + lane * 8);
      td[u] = *reinterpret_cast<const uint4*>(dy + rr * 512 + lane * 8);
      tr[u] = dres ? *reinterpret_cast<const uint4*>(dres + rr * 512 + lane * 8) : make_uint4(0, 0, 0, 0);
      mu[u] = mean[rr];
      rs[u] = rstd[rr];
      masked[u] = !valid[u];
      if (valid[u] && row_lens) masked[u] = s2t_row_masked(row_lens, row_T, row);
    }
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      float xv[8], dv[8], xh[8], dg[8];
      unpack8(tx[u], xv);
      unpack8(td[u], dv);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float d = masked[u] ? 0.f : dv[r];
        xh[r] = (xv[r] - mu[u]) * rs[u];
        dg[r] = d * g[r];
        s1 += dg[r];
        s2 += dg[r] * xh[r];
        ag[r] += d * xh[r];
        ab[r] += d;
      }
      s1 = wave_sum(s1) * (1.f / 512.f);
      s2 = wave_sum(s2) * (1.f / 512.f);
      if (valid[u]) {
        float o[8], q[8];
        unpack8(tr[u], q);
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = rs[u] * (dg[r] - s1 - xh[r] * s2) + q[r];
        *reinterpret_cast<uint4*>(dx + (row0 + u) * 512 + lane * 8) = pack8f(o);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    red[0][wv][lane * 8 + r] = ag[r];
    red[1][wv][lane * 8 + r] = ab[r];
  }
  __syncthreads();
  float* w = ws + (int64_t)(blockIdx.x % replicas) * 1024;
  for (int c = threadIdx.x; c < 512; c += 256) {
    atomicAdd(w + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(w + 512 + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// dgamma += sum_r ws[r][0][:] ; dbeta += sum_r ws[r][1][:]
__global__ void ln_bwd_finalize_kernel(float* __restrict__ ws, int replicas, int cols, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float g = 0.f, b = 0.f;
#pragma unroll 8
  for (int r = 0; r < replicas; ++r) {
    g += ws[(int64_t)r * 2 * cols + c];
    b += ws[(int64_t)r * 2 * cols + cols + c];
  }
  for (int r = 0; r < replicas; ++r) {  // leave the workspace zeroed for the next call (stream-ordered reuse)
    ws[(int64_t)r * 2 * cols + c] = 0.f;
    ws[(int64_t)r * 2 * cols + cols + c] = 0.f;
  }
  dgamma[c] += g;
  dbeta[c] += b;
}

}  // namespace

#define LN_DISPATCH(KERN, T_, ...)                                                        \
  do {                                                                                    \
    const int nv = (cols + 255) / 256;                                                    \
    if (nv <= 1) hipLaunchKernelGGL((KERN<T_, 1>), grid, block, 0, s, __VA_ARGS__);       \
    else if (nv <= 2) hipLaunchKernelGGL((KERN<T_, 2>), grid, block, 0, s, __VA_ARGS__);  \
    else if (nv <= 4) hipLaunchKernelGGL((KERN<T_, 4>), grid, block, 0, s, __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERN<T_, 8>), grid, block, 0, s, __VA_ARGS__);               \
  } while (0)

extern "C" int s2t_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                                 float* mean, float* rstd, int64_t rows, int cols, float eps,
                                 const int32_t* row_lens, int row_T, void* stream) {
  if (!x || !gamma || !beta || !y || rows < 0 || cols <= 0) return S2T_ERR_ARG;
  if (cols % 4 || cols > LN_MAX_VEC * 256) return S2T_ERR_UNSUPPORTED;
  if (rows == 0) return S2T_OK;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    LN_DISPATCH(ln_fwd_kernel, float, (const float*)x, gamma, beta, (float*)y, mean, rstd, rows, cols, eps, row_lens, row_T);
  else if (dtype == S2T_BF16 && cols == 256 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0)
    hipLaunchKernelGGL(ln256_fwd_kernel, dim3((unsigned)((rows + 15) / 16)), block, 0, s, (const bf16_t*)x, gamma, beta,
                       (bf16_t*)y, mean, rstd, rows, eps, row_lens, row_T);
  else if (dtype == S2T_BF16 && cols == 512 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0)
    hipLaunchKernelGGL(ln512_fwd_kernel, dim3((unsigned)((rows + 15) / 16)), block, 0, s, (const bf16_t*)x, gamma, beta,
                       (bf16_t*)y, mean, rstd, rows, eps, row_lens, row_T);
  else if (dtype == S2T_BF16)
    LN_DISPATCH(ln_fwd_kernel, bf16_t, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, cols, eps, row_lens, row_T);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_layernorm_bwd(int dtype, const void* x, const float* gamma, const void* dy, const float* mean,
                                 const float* rstd, void* dx, float* dgamma, float* dbeta, float* ws, int replicas,
                                 int64_t rows, int cols, const int32_t* row_lens, int row_T, const void* dres,
                                 void* dx_drop, float drop_p, const uint64_t* drop_seed, uint32_t drop_site,
                                 void* stream) {
  if (!x || !gamma || !dy || !mean || !rstd || !dx || (!dgamma != !dbeta) || !ws || replicas <= 0 || rows < 0 || cols <= 0)
    return S2T_ERR_ARG;
  if (cols % 4 || cols > LN_MAX_VEC * 256) return S2T_ERR_UNSUPPORTED;
  if (rows == 0) return S2T_OK;
  int64_t nb = (rows + 3) / 4;
  if (nb > 2048) nb = 2048;
  dim3 grid((unsigned)nb), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool fast256 = dtype == S2T_BF16 && cols == 256 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 &&
                       ((uintptr_t)dx % 16) == 0 && ((uintptr_t)dres % 16) == 0;
  if (dx_drop && (!fast256 || ((uintptr_t)dx_drop % 16) || !(drop_p > 0.f && drop_p < 1.f))) return S2T_ERR_UNSUPPORTED;
  if (dtype == S2T_F32)
    LN_DISPATCH(ln_bwd_kernel, float, (const float*)x, gamma, (const float*)dy, mean, rstd, (float*)dx, (const float*)dres, ws, replicas, rows, cols, row_lens, row_T);
  else if (fast256) {
    // 32 rows per workgroup and trip; at most 512 workgroups: every workgroup closes with 512 float atomics into the
    // replica workspace, which are executed at the memory side (2000 workgroups made that a million per call)
    int64_t nb8 = (rows + 31) / 32;
    if (nb8 > 512) nb8 = 512;
    hipLaunchKernelGGL(ln256_bwd_kernel, dim3((unsigned)nb8), block, 0, s, (const bf16_t*)x, gamma, (const bf16_t*)dy, mean,
                       rstd, (bf16_t*)dx, (const bf16_t*)dres, ws, replicas, rows, row_lens, row_T, (bf16_t*)dx_drop, drop_p,
                       drop_seed, drop_site);
  } else if (dtype == S2T_BF16 && cols == 512 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
             ((uintptr_t)dres % 16) == 0) {
    int64_t nb8 = (rows + 7) / 8;
    if (nb8 > 512) nb8 = 512;
    hipLaunchKernelGGL(ln512_bwd_kernel, dim3((unsigned)nb8), block, 0, s, (const bf16_t*)x, gamma, (const bf16_t*)dy, mean, rstd,
                       (bf16_t*)dx, (const bf16_t*)dres, ws, replicas, rows, row_lens, row_T);
  } else if (dtype == S2T_BF16)
    LN_DISPATCH(ln_bwd_kernel, bf16_t, (const bf16_t*)x, gamma, (const bf16_t*)dy, mean, rstd, (bf16_t*)dx, (const bf16_t*)dres, ws, replicas, rows, cols, row_lens, row_T);
  else return S2T_ERR_DTYPE;
  if (dgamma)  // NULL: the partial sums stay in ws, the caller folds many LayerNorms at once (s2t_layernorm_fold)
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((cols + 63) / 64), dim3(64), 0, s, ws, replicas, cols, dgamma, dbeta);
  return S2T_LAUNCH_CHECK();
}

namespace {
constexpr int FOLD_MAX = 96;  // entries per launch (passed by value: 3 KiB of kernel arguments)
struct FoldBatch {
  s2t_ln_fold_entry e[FOLD_MAX];
};
__global__ void ln_fold_kernel(const FoldBatch batch, int replicas) {
  const s2t_ln_fold_entry en = batch.e[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= en.cols) return;
  float g = 0.f, b = 0.f;
#pragma unroll 8
  for (int r = 0; r < replicas; ++r) {
    g += en.ws[(int64_t)r * 2 * en.cols + c];
    b += en.ws[(int64_t)r * 2 * en.cols + en.cols + c];
  }
  for (int r = 0; r < replicas; ++r) {  // leave the workspace zeroed for its next use
    en.ws[(int64_t)r * 2 * en.cols + c] = 0.f;
    en.ws[(int64_t)r * 2 * en.cols + en.cols + c] = 0.f;
  }
  en.dgamma[c] += g;
  en.dbeta[c] += b;
}
}  // namespace

extern "C" int s2t_layernorm_fold(const s2t_ln_fold_entry* entries, int n, int replicas, void* stream) {
  if (!entries || n < 0 || replicas <= 0) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += FOLD_MAX) {
    FoldBatch batch;
    const int m = n - i0 < FOLD_MAX ? n - i0 : FOLD_MAX;
    int maxc = 0;
    for (int i = 0; i < m; ++i) {
      batch.e[i] = entries[i0 + i];
      if (!batch.e[i].ws || !batch.e[i].dgamma || !batch.e[i].dbeta || batch.e[i].cols <= 0) return S2T_ERR_ARG;
      if (batch.e[i].cols > maxc) maxc = batch.e[i].cols;
    }
    hipLaunchKernelGGL(ln_fold_kernel, dim3((maxc + 63) / 64, m), dim3(64), 0, s, batch, replicas);
  }
  return S2T_LAUNCH_CHECK();
}
