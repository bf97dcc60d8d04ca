// Strided depthwise "pooling" convolution with kernel = stride = r and no padding — the depthwise stage of the
// DownSampleConvolutionModule that PDS multi-scale fusion applies to earlier stage outputs
// (reference: fairseq/modules/downsample_convolution.py:45-54,97-100, used from
// models/speech_to_text/pdss2t_transformer.py:1187-1233 with r = the remaining down-sampling ratio, 1..8):
//
//     y[b][t'][c] = bias[c] + sum_{k<r} x[b][t'*r + k][c] * w[c][k]          t' < T' = floor(T / r)
//
// Channels-last rows, a lane owns 4 consecutive channels (8/16-byte accesses).  The forward also emits the per-workgroup
// partial sums (sum y | sum y^2 per channel, one row per workgroup, fixed order) that s2t_bn_finalize folds into the
// BatchNorm batch statistics, exactly like s2t_dwconv_fwd.  HBM-bound; no LDS tiling beyond the partial-sum exchange.
#include "common.h"

namespace {

constexpr int PT = 32;    // output frames per workgroup
constexpr int PCH = 256;  // channels per workgroup
constexpr int RMAX = 8;

template <typename T>
__global__ __launch_bounds__(256) void dwpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ y, int B, int Tin,
                                                         int Tout, int C, int r, float* __restrict__ stats) {
  __shared__ float red[2][4][PCH];
  const int cq = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c0 = blockIdx.z * PCH, c = c0 + cq * 4;
  const int b = blockIdx.y, t0 = blockIdx.x * PT;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    float wk[RMAX][4], bs[4];
    for (int k = 0; k < r; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) wk[k][q] = w[(int64_t)(c + q) * r + k];
    ld4_as_f32<float>(bias + c, bs);
    for (int i = tg; i < PT; i += 4) {
      const int t = t0 + i;
      if (t >= Tout) break;
      float o[4] = {bs[0], bs[1], bs[2], bs[3]};
      for (int k = 0; k < r; ++k) {
        float v[4];
        ld4_as_f32<T>(x + ((int64_t)b * Tin + (int64_t)t * r + k) * C + c, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] += v[q] * wk[k][q];
      }
      st4_from_f32<T>(y + ((int64_t)b * Tout + t) * C + c, o);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1[q] += o[q];
        s2[q] += o[q] * o[q];
      }
    }
  }
  if (!stats) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    red[0][tg][cq * 4 + q] = s1[q];
    red[1][tg][cq * 4 + q] = s2[q];
  }
  __syncthreads();
  const int cc = threadIdx.x;
  if (c0 + cc < C) {
    float* row = stats + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
    row[c0 + cc] = ((red[0][0][cc] + red[0][1][cc]) + red[0][2][cc]) + red[0][3][cc];
    row[C + c0 + cc] = ((red[1][0][cc] + red[1][1][cc]) + red[1][2][cc]) + red[1][3][cc];
  }
}

// dx[b][t'*r + k][c] = dy[b][t'][c] * w[c][k];  dw[c][k] += sum dy * x;  db[c] += sum dy   (fp32 atomics per workgroup)
template <typename T>
__global__ __launch_bounds__(256) void dwpool_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const T* __restrict__ dy, T* __restrict__ dx,
                                                         float* __restrict__ dw, float* __restrict__ db, int B, int Tin,
                                                         int Tout, int C, int r) {
  __shared__ float red[4][PCH];
  const int cq = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c0 = blockIdx.z * PCH, c = c0 + cq * 4;
  const int b = blockIdx.y, t0 = blockIdx.x * PT;
  float aw[RMAX][4], ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < RMAX; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) aw[k][q] = 0.f;
  if (c < C) {
    float wk[RMAX][4];
    for (int k = 0; k < r; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) wk[k][q] = w[(int64_t)(c + q) * r + k];
    for (int i = tg; i < PT; i += 4) {
      const int t = t0 + i;
      if (t >= Tout) break;
      float g[4];
      ld4_as_f32<T>(dy + ((int64_t)b * Tout + t) * C + c, g);
#pragma unroll
      for (int q = 0; q < 4; ++q) ab[q] += g[q];
      for (int k = 0; k < r; ++k) {
        const int64_t row = ((int64_t)b * Tin + (int64_t)t * r + k) * C + c;
        float v[4], o[4];
        ld4_as_f32<T>(x + row, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          aw[k][q] += g[q] * v[q];
          o[q] = g[q] * wk[k][q];
        }
        st4_from_f32<T>(dx + row, o);
      }
    }
  }
  // workgroup sums of db and of each dw tap, then one atomic per (channel, tap)
  for (int k = -1; k < r; ++k) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) red[tg][cq * 4 + q] = k < 0 ? ab[q] : aw[k][q];
    __syncthreads();
    const int cc = threadIdx.x;
    if (c0 + cc < C) {
      const float sum = ((red[0][cc] + red[1][cc]) + red[2][cc]) + red[3][cc];
      if (k < 0) atomicAdd(db + c0 + cc, sum);
      else atomicAdd(dw + (int64_t)(c0 + cc) * r + k, sum);
    }
  }
}

}  // namespace

extern "C" int s2t_dwpool_stat_partials(int B, int Tout) { return B * ((Tout + PT - 1) / PT); }

extern "C" int s2t_dwpool_fwd(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int Tin, int C,
                              int r, float* stats, void* stream) {
  if (!x || !w || !bias || !y || B <= 0 || Tin <= 0 || C <= 0 || C % 4 || r <= 0) return S2T_ERR_ARG;
  if (r > RMAX) return S2T_ERR_UNSUPPORTED;
  const int Tout = Tin / r;
  if (Tout <= 0) return S2T_ERR_ARG;
  dim3 grid((Tout + PT - 1) / PT, B, (C + PCH - 1) / PCH), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(dwpool_fwd_kernel<float>, grid, block, 0, s, (const float*)x, w, bias, (float*)y, B, Tin, Tout, C, r, stats);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(dwpool_fwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, B, Tin, Tout, C, r, stats);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_dwpool_bwd(int dtype, const void* x, const float* w, const void* dy, void* dx, float* dw, float* db,
                              int B, int Tin, int C, int r, void* stream) {
  if (!x || !w || !dy || !dx || !dw || !db || B <= 0 || Tin <= 0 || C <= 0 || C % 4 || r <= 0) return S2T_ERR_ARG;
  if (r > RMAX) return S2T_ERR_UNSUPPORTED;
  const int Tout = Tin / r;
  if (Tout <= 0) return S2T_ERR_ARG;
  dim3 grid((Tout + PT - 1) / PT, B, (C + PCH - 1) / PCH), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(dwpool_bwd_kernel<float>, grid, block, 0, s, (const float*)x, w, (const float*)dy, (float*)dx, dw, db, B, Tin, Tout, C, r);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(dwpool_bwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, w, (const bf16_t*)dy, (bf16_t*)dx, dw, db, B, Tin, Tout, C, r);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}
