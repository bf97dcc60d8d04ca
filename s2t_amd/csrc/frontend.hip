// Feature front-end on the device (§8(f) row 3): Kaldi-compatible log-mel filterbank and
// per-utterance CMVN, fed from raw (int16-range) audio already resident in HBM.
//
//   a1  data/audio/audio_utils.py:59-79 -> torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=80,
//       sample_frequency=sr) with torchaudio's defaults (third-party, not in the reference tree: parity unpinned).
//   a2  data/audio/feature_transforms/utterance_cmvn.py:31-45.
//
// Both are HBM-trivial (160 new samples in, 80 floats out per frame); the design goal is one launch per batch with no
// intermediate tensors: a wave owns a frame end to end (DC removal, pre-emphasis, window, FFT in LDS, power spectrum,
// mel projection, log).
#include "common.h"

namespace {

constexpr int MAX_NFFT = 1024;

// one wave per frame, 4 frames per workgroup
__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wave, int64_t wave_stride,
                                                    const int32_t* __restrict__ n_samples, float* __restrict__ feat,
                                                    int64_t feat_stride_b, int max_frames, int win, int shift, int nfft,
                                                    int log2n, const float* __restrict__ window,
                                                    const float* __restrict__ mel_t /* [nfft/2+1][n_mel] */, int n_mel,
                                                    float preemph, int remove_dc, float log_floor) {
  __shared__ float2 buf[4][2][MAX_NFFT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int t = blockIdx.x * 4 + wv;
  if (t >= max_frames) return;
  float* out = feat + (int64_t)b * feat_stride_b + (int64_t)t * n_mel;
  const int ns = n_samples[b];
  const int n_frames = ns >= win ? 1 + (ns - win) / shift : 0;
  if (t >= n_frames) {  // padded frame: zeros (the collater zero-pads, speech_to_text_dataset.py:267-285)
    for (int m = lane; m < n_mel; m += 64) out[m] = 0.f;
    return;
  }
  const float* x = wave + (int64_t)b * wave_stride + (int64_t)t * shift;
  float2* a = buf[wv][0];
  float2* c = buf[wv][1];
  // DC offset of the frame
  float mean = 0.f;
  if (remove_dc) {
    float s = 0.f;
    for (int i = lane; i < win; i += 64) s += x[i];
    mean = wave_sum(s) / (float)win;
  }
  // pre-emphasis against the previous sample of the DC-free frame (the first sample against itself), window, zero-pad
  for (int i = lane; i < nfft; i += 64) {
    float v = 0.f;
    if (i < win) {
      const float cur = x[i] - mean;
      const float prev = x[i > 0 ? i - 1 : 0] - mean;
      v = (cur - preemph * prev) * window[i];
    }
    a[i] = make_float2(v, 0.f);
  }
  __builtin_amdgcn_wave_barrier();
  // Stockham radix-2 decimation-in-frequency FFT, ping-pong between the two LDS buffers (wave-private: no barriers,
  // the LDS queue of a wave is in order)
  int n = nfft, s = 1;
  for (int st = 0; st < log2n; ++st) {
    const int m = n >> 1;
    for (int j = lane; j < (nfft >> 1); j += 64) {
      const int p = j / s, q = j - p * s;
      float sn, cs;
      sincospif(-2.0f * (float)p / (float)n, &sn, &cs);
      const float2 u = a[q + s * p];
      const float2 v = a[q + s * (p + m)];
      c[q + s * (2 * p)] = make_float2(u.x + v.x, u.y + v.y);
      const float dx = u.x - v.x, dy = u.y - v.y;
      c[q + s * (2 * p + 1)] = make_float2(dx * cs - dy * sn, dx * sn + dy * cs);
    }
    __builtin_amdgcn_wave_barrier();
    float2* tmp = a;
    a = c;
    c = tmp;
    n = m;
    s <<= 1;
  }
  // power spectrum into the real parts of the free buffer
  const int nbin = (nfft >> 1) + 1;
  float* pw = reinterpret_cast<float*>(c);
  for (int k = lane; k < nbin; k += 64) pw[k] = a[k].x * a[k].x + a[k].y * a[k].y;
  __builtin_amdgcn_wave_barrier();
  // mel projection (lane = mel bin; mel_t rows are contiguous over bins) + log
  for (int m = lane; m < n_mel; m += 64) {
    float e = 0.f;
    for (int k = 0; k < nbin; ++k) e = fmaf(pw[k], mel_t[(int64_t)k * n_mel + m], e);
    out[m] = logf(fmaxf(e, log_floor));
  }
}

// per-utterance CMVN: one workgroup per utterance, thread = (row group, column); double accumulators
__global__ __launch_bounds__(512) void cmvn_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   const int32_t* __restrict__ n_frames, int64_t stride_b, int C,
                                                   int norm_means, int norm_vars) {
  __shared__ double ssum[4][128], ssq[4][128];
  __shared__ float smean[128], sinv[128];
  const int b = blockIdx.x;
  const int cl = threadIdx.x & 127, rg = threadIdx.x >> 7;
  const int T = n_frames[b];
  const float* xb = x + (int64_t)b * stride_b;
  float* yb = y + (int64_t)b * stride_b;
  for (int c0 = 0; c0 < C; c0 += 128) {
    const int c = c0 + cl;
    double s = 0.0, q = 0.0;
    if (c < C)
      for (int r = rg; r < T; r += 4) {
        const double v = xb[(int64_t)r * C + c];
        s += v;
        q += v * v;
      }
    ssum[rg][cl] = s;
    ssq[rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < C) {
      const double S = ssum[0][cl] + ssum[1][cl] + ssum[2][cl] + ssum[3][cl];
      const double Q = ssq[0][cl] + ssq[1][cl] + ssq[2][cl] + ssq[3][cl];
      const double mean = T > 0 ? S / T : 0.0;
      const double var = T > 0 ? Q / T - mean * mean : 1.0;
      smean[cl] = norm_means ? (float)mean : 0.f;
      sinv[cl] = norm_vars ? (float)(1.0 / sqrt(var > 1e-10 ? var : 1e-10)) : 1.f;
    }
    __syncthreads();
    if (c < C)
      for (int r = rg; r < T; r += 4) yb[(int64_t)r * C + c] = (xb[(int64_t)r * C + c] - smean[cl]) * sinv[cl];
    __syncthreads();
  }
}

// SpecAugment masks (feature_transforms/specaugment.py:79-131): the host draws the intervals (numpy RandomState order of
// the reference), the device applies them to the whole batch in place.  masks: [B][n_masks][2] int32 = (start, width),
// the first n_freq of each utterance along the feature axis, the rest along time; value[b] fills the masked cells.
__global__ __launch_bounds__(256) void specaug_kernel(float* __restrict__ x, const int32_t* __restrict__ n_frames,
                                                      int64_t stride_b, int C, const int32_t* __restrict__ masks,
                                                      int n_freq, int n_time, const float* __restrict__ value) {
  const int b = blockIdx.y;
  const int T = n_frames[b];
  const int32_t* m = masks + (int64_t)b * (n_freq + n_time) * 2;
  const float val = value[b];
  float* xb = x + (int64_t)b * stride_b;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)T * C; idx += (int64_t)gridDim.x * 256) {
    const int t = (int)(idx / C), c = (int)(idx % C);
    bool hit = false;
    for (int i = 0; i < n_freq; ++i) hit |= (c >= m[2 * i] && c < m[2 * i] + m[2 * i + 1]);
    for (int i = n_freq; i < n_freq + n_time; ++i) hit |= (t >= m[2 * i] && t < m[2 * i] + m[2 * i + 1]);
    if (hit) xb[idx] = val;
  }
}

// SpecAugment time warp (feature_transforms/specaugment.py:97-112): the first w0 frames are resized to w0 + w frames and the
// remaining n - w0 to n - w0 - w, each by cv2.resize(..., interpolation=cv2.INTER_LINEAR) along the time axis only (the feature
// axis keeps its size, so its interpolation is the identity).  OpenCV (third-party, absent here: parity unpinned) maps output
// row dy of a segment to the source coordinate fy = (float)((dy + 0.5) * src/dst - 0.5), takes rows floor(fy) and floor(fy) + 1
// clamped into the segment and blends them with weights (1 - frac, frac) in fp32.  warp: [B][2] int32 = (w0, w); w0 <= 0
// leaves the utterance as it is; rows >= n_frames[b] are copied.
__global__ __launch_bounds__(256) void time_warp_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const int32_t* __restrict__ n_frames, int64_t stride_b,
                                                        int max_frames, int C, const int32_t* __restrict__ warp) {
  const int b = blockIdx.y;
  const int n = n_frames[b];
  const int w0 = warp[2 * b], w = warp[2 * b + 1];
  const float* xb = x + (int64_t)b * stride_b;
  float* yb = y + (int64_t)b * stride_b;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)max_frames * C; idx += (int64_t)gridDim.x * 256) {
    const int t = (int)(idx / C), c = (int)(idx % C);
    if (t >= n || w0 <= 0) {
      yb[idx] = xb[idx];
      continue;
    }
    int seg0, src, dst, dy;  // source segment [seg0, seg0 + src) -> dst output rows, dy the row inside the output segment
    if (t < w0 + w) {
      seg0 = 0; src = w0; dst = w0 + w; dy = t;
    } else {
      seg0 = w0; src = n - w0; dst = n - w0 - w; dy = t - (w0 + w);
    }
    float fy = (float)(((double)dy + 0.5) * ((double)src / (double)dst) - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int s0 = min(max(sy, 0), src - 1), s1 = min(max(sy + 1, 0), src - 1);
    const float a = xb[(int64_t)(seg0 + s0) * C + c], bb = xb[(int64_t)(seg0 + s1) * C + c];
    yb[idx] = __fadd_rn(__fmul_rn(a, 1.0f - fy), __fmul_rn(bb, fy));  // no contraction: the oracle's two products and a sum
  }
}

// mean over the first n_frames[b] rows of each utterance (mask_value = None: "use local mean")
__global__ __launch_bounds__(256) void utt_mean_kernel(const float* __restrict__ x, const int32_t* __restrict__ n_frames,
                                                       int64_t stride_b, int C, float* __restrict__ mean) {
  __shared__ double red[256];
  const int b = blockIdx.x;
  const int64_t n = (int64_t)n_frames[b] * C;
  const float* xb = x + (int64_t)b * stride_b;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += xb[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) mean[b] = n > 0 ? (float)(red[0] / (double)n) : 0.f;
}

}  // namespace

extern "C" int s2t_fbank(const float* wave, int64_t wave_stride, const int32_t* n_samples, float* feat,
                         int64_t feat_stride_b, int max_frames, int B, int win, int shift, int nfft, const float* window,
                         const float* mel_t, int n_mel, float preemph, int remove_dc, float log_floor, void* stream) {
  if (!wave || !n_samples || !feat || !window || !mel_t || B <= 0 || max_frames < 0 || n_mel <= 0) return S2T_ERR_ARG;
  if (win <= 0 || shift <= 0 || nfft < win || nfft > MAX_NFFT || (nfft & (nfft - 1))) return S2T_ERR_UNSUPPORTED;
  if (max_frames == 0) return S2T_OK;
  int log2n = 0;
  while ((1 << log2n) < nfft) ++log2n;
  dim3 grid((max_frames + 3) / 4, B), block(256);
  hipLaunchKernelGGL(fbank_kernel, grid, block, 0, (hipStream_t)stream, wave, wave_stride, n_samples, feat, feat_stride_b,
                     max_frames, win, shift, nfft, log2n, window, mel_t, n_mel, preemph, remove_dc, log_floor);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_utterance_cmvn(const float* x, float* y, const int32_t* n_frames, int64_t stride_b, int B, int C,
                                  int norm_means, int norm_vars, void* stream) {
  if (!x || !y || !n_frames || B <= 0 || C <= 0) return S2T_ERR_ARG;
  hipLaunchKernelGGL(cmvn_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, x, y, n_frames, stride_b, C, norm_means,
                     norm_vars);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_time_warp(const float* x, float* y, const int32_t* n_frames, int64_t stride_b, int B, int max_frames,
                             int C, const int32_t* warp, float* mean_out, void* stream) {
  if (!x || !y || x == y || !n_frames || !warp || B <= 0 || C <= 0 || max_frames < 0) return S2T_ERR_ARG;
  if (max_frames == 0) return S2T_OK;
  hipStream_t s = (hipStream_t)stream;
  // (mask_value = None fills with the mean of the spectrogram BEFORE the warp, specaugment.py:89-90)
  if (mean_out) hipLaunchKernelGGL(utt_mean_kernel, dim3(B), dim3(256), 0, s, x, n_frames, stride_b, C, mean_out);
  int64_t nb = ((int64_t)max_frames * C + 255) / 256;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(time_warp_kernel, dim3((unsigned)nb, B), dim3(256), 0, s, x, y, n_frames, stride_b, max_frames, C, warp);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_specaugment(float* x, const int32_t* n_frames, int64_t stride_b, int B, int max_frames, int C,
                               const int32_t* masks, int n_freq, int n_time, float* value, int value_is_mean,
                               void* stream) {
  if (!x || !n_frames || !value || B <= 0 || C <= 0 || n_freq < 0 || n_time < 0) return S2T_ERR_ARG;
  if (n_freq + n_time == 0 || max_frames <= 0) return S2T_OK;
  if (!masks) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (value_is_mean) hipLaunchKernelGGL(utt_mean_kernel, dim3(B), dim3(256), 0, s, x, n_frames, stride_b, C, value);
  int64_t nb = ((int64_t)max_frames * C + 255) / 256;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(specaug_kernel, dim3((unsigned)nb, B), dim3(256), 0, s, x, n_frames, stride_b, C, masks, n_freq, n_time,
                     value);
  return S2T_LAUNCH_CHECK();
}
