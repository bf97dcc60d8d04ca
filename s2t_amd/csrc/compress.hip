// CTC-guided compression of the frame axis (reference: S2TTransformerEncoder.forward,
// fairseq/models/speech_to_text/s2t_transformer.py:1948-1986, ``--compression-metric threshold --compression-mode create``):
// after an intermediate CTC head, frames whose blank posterior reaches the layer's threshold are dropped and every
// utterance's remaining frames are left-packed.  The reference loops over the batch on the host with boolean indexing;
// here one workgroup per utterance turns the keep flags into source indices with a ballot/popcount prefix sum, and a
// row-gather (forward) / row-scatter (backward) moves the d-wide frames.  HBM-bound byte movers: no LDS tiles, 16-byte
// accesses, one row segment per thread.
#include "common.h"

namespace {

// keep[t] = t < len && softmax(logit[t])[blank] < thr  ->  src[b][j] = t of the j-th kept frame, new_lens[b] = #kept
template <typename T>
__global__ __launch_bounds__(256) void compress_plan_kernel(const T* __restrict__ logits, int64_t ld,
                                                            const float* __restrict__ lse,
                                                            const int32_t* __restrict__ lens, int Tn, int blank, float thr,
                                                            int32_t* __restrict__ src, int32_t* __restrict__ new_lens) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = min(lens[b], Tn);
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < Tn; t0 += 256) {
    const int t = t0 + tid;
    bool keep = false;
    if (t < len) {
      const int64_t row = (int64_t)b * Tn + t;
      const float p = __expf(ld_as_f32<T>(logits + row * ld + blank) - lse[row]);
      keep = p < thr;
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (keep) src[(int64_t)b * Tn + off + before] = t;
    __syncthreads();
    if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) new_lens[b] = base_s;
}

// forward: y[b][j][:] = j < new_lens[b] ? x[b][src[b][j]][:] : 0      (x: [B][T][C], y: [B][Tn][C], 16-byte vectors)
// backward (scatter = true): dx[b][src[b][j]][:] = dy[b][j][:] for j < new_lens[b] (dx zero-filled by the caller)
template <bool SCATTER>
__global__ __launch_bounds__(256) void compress_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                            const int32_t* __restrict__ src,
                                                            const int32_t* __restrict__ new_lens, int B, int T, int Tn,
                                                            int vec_per_row) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Tn * vec_per_row;
  if (idx >= total) return;
  const int v = (int)(idx % vec_per_row);
  const int64_t r = idx / vec_per_row;
  const int j = (int)(r % Tn), b = (int)(r / Tn);
  const bool live = j < new_lens[b];
  if (SCATTER) {
    if (live) out[((int64_t)b * T + src[(int64_t)b * T + j]) * vec_per_row + v] = in[idx];
  } else {
    out[idx] = live ? in[((int64_t)b * T + src[(int64_t)b * T + j]) * vec_per_row + v] : make_uint4(0, 0, 0, 0);
  }
}

// packed rows <-> padded rows of one batch through its row map (include/s2t_hip.h, "Packed rows"); 16-byte pieces
template <bool TO_PACKED>
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                        const int32_t* __restrict__ map, int64_t rows, int T,
                                                        int vec_per_row) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  rows = s2t_live_rows(map, S2T_ROWS_PACKED, rows);
  if (idx >= rows * vec_per_row) return;
  const int64_t m = idx / vec_per_row;
  const int v = (int)(idx % vec_per_row);
  const int e = map[m];
  const int64_t padded = ((int64_t)(e >> 16) * T + (e & 0xffff)) * vec_per_row + v;
  if (TO_PACKED) out[idx] = e >= 0 ? in[padded] : make_uint4(0, 0, 0, 0);
  else if (e >= 0) out[padded] = in[idx];
}

// Geometry of a packed batch from its lengths, one launch: cu[b] = rows in front of utterance b (frames + halo rows each), the row
// map behind a 4-word header whose last word is the live row count.  Every workgroup scans the (at most 1024) lengths itself and
// fills its slice of the map: row m belongs to the utterance whose [cu[b], cu[b + 1]) holds it (binary search in LDS).
__global__ __launch_bounds__(1024) void rows_geometry_kernel(const int32_t* __restrict__ lens, int B, int T, int halo,
                                                             int32_t* __restrict__ cu, int32_t* __restrict__ buf, int M) {
  __shared__ int scu[1025];
  __shared__ int slen[1024];
  const int tid = threadIdx.x;
  int cap = 0, len = 0;
  if (tid < B) {
    len = lens[tid];
    const int rest = T - len;
    cap = len + (rest < 0 ? 0 : (rest < halo ? rest : halo));
    slen[tid] = len;
  }
  scu[tid + 1] = cap;  // inclusive scan (Hillis-Steele) over 1024 slots
  if (tid == 0) scu[0] = 0;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid + 1 > o ? scu[tid + 1 - o] : 0;
    __syncthreads();
    scu[tid + 1] += v;
    __syncthreads();
  }
  const int live = scu[B];
  if (blockIdx.x == 0) {
    if (tid < B) cu[tid] = scu[tid];
    if (tid == 0) cu[B] = live;  // (its own store: with B = 1024 no thread has tid == B)
    if (tid < 4) buf[tid] = tid == 3 ? live : 0;
  }
  for (int m = blockIdx.x * 1024 + tid; m < M; m += gridDim.x * 1024) {
    int e = -1;
    if (m < live) {
      int lo = 0, hi = B;  // the utterance b with scu[b] <= m < scu[b + 1]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (scu[mid] <= m) lo = mid; else hi = mid;
      }
      const int t = m - scu[lo];
      e = t < slen[lo] ? ((lo << 16) | t) : -1;
    }
    buf[4 + m] = e;
  }
}

}  // namespace

extern "C" int s2t_rows_geometry(const int32_t* lens, int B, int T, int halo, int32_t* cu, int32_t* map_buf, void* stream) {
  if (!lens || !cu || !map_buf || B <= 0 || B > 1024 || T <= 0 || T > 65535 || halo < 0) return S2T_ERR_ARG;
  const int M = B * T;
  int nb = (M + 4095) / 4096;
  if (nb > 64) nb = 64;
  hipLaunchKernelGGL(rows_geometry_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, lens, B, T, halo, cu, map_buf, M);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_pack_rows(int dtype, const void* in, void* out, const int32_t* map, int64_t rows, int T, int C,
                             int to_packed, void* stream) {
  if (!in || !out || !map || rows < 0 || T <= 0 || T > 65535 || C <= 0) return S2T_ERR_ARG;
  const int esz = dtype == S2T_F32 ? 4 : dtype == S2T_BF16 ? 2 : 0;
  if (!esz) return S2T_ERR_DTYPE;
  if ((C * esz) % 16 || ((uintptr_t)in % 16) || ((uintptr_t)out % 16)) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  const int vpr = C * esz / 16;
  dim3 grid((unsigned)((rows * vpr + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (to_packed)
    hipLaunchKernelGGL(pack_rows_kernel<true>, grid, dim3(256), 0, s, (const uint4*)in, (uint4*)out, map, rows, T, vpr);
  else
    hipLaunchKernelGGL(pack_rows_kernel<false>, grid, dim3(256), 0, s, (const uint4*)in, (uint4*)out, map, rows, T, vpr);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ctc_compress_plan(int dtype, const void* logits, int64_t ld, const float* lse, const int32_t* lens,
                                     int B, int T, int blank, float threshold, int32_t* src, int32_t* new_lens,
                                     void* stream) {
  if (!logits || !lse || !lens || !src || !new_lens || B <= 0 || T <= 0 || blank < 0 || blank >= ld) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2T_F32)
    hipLaunchKernelGGL(compress_plan_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)logits, ld, lse, lens, T, blank,
                       threshold, src, new_lens);
  else if (dtype == S2T_BF16)
    hipLaunchKernelGGL(compress_plan_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)logits, ld, lse, lens, T,
                       blank, threshold, src, new_lens);
  else return S2T_ERR_DTYPE;
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_compress_rows(int dtype, const void* in, void* out, const int32_t* src, const int32_t* new_lens, int B,
                                 int T, int Tn, int C, int scatter, void* stream) {
  if (!in || !out || !src || !new_lens || B <= 0 || T <= 0 || Tn <= 0 || Tn > T || C <= 0) return S2T_ERR_ARG;
  const int esz = dtype == S2T_F32 ? 4 : dtype == S2T_BF16 ? 2 : 0;
  if (!esz) return S2T_ERR_DTYPE;
  if ((C * esz) % 16 || ((uintptr_t)in % 16) || ((uintptr_t)out % 16)) return S2T_ERR_ARG;
  const int vpr = C * esz / 16;
  const int64_t total = (int64_t)B * Tn * vpr;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (scatter)
    hipLaunchKernelGGL(compress_rows_kernel<true>, grid, dim3(256), 0, s, (const uint4*)in, (uint4*)out, src, new_lens, B, T,
                       Tn, vpr);
  else
    hipLaunchKernelGGL(compress_rows_kernel<false>, grid, dim3(256), 0, s, (const uint4*)in, (uint4*)out, src, new_lens, B,
                       T, Tn, vpr);
  return S2T_LAUNCH_CHECK();
}
