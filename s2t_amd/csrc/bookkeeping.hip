// Per-batch bookkeeping of a training step, one launch each, written INTO tensors a captured step reads (gfx950).
//
// The reference derives these inside forward() with a dozen small ATen launches per item (sort / scan / gather / compare);
// a captured step keeps them outside the graph (s2t_amd/functional.py: batch_memo) and refreshes them when a new batch is
// copied into the static one — 45 launches and 0.23 ms per step of the headline bench before these entries.
//   s2t_subsampled_lengths  data/data_utils.py:518-522 (lengths_to_padding_mask) on the subsampler's output lengths
//                           (modules/speech_to_text/subsampling.py:150-154: two stride-2 convolutions)
//   s2t_token_positions     utils.py:240-250 (make_positions) + the target-side key lengths
//   s2t_ctc_targets         criterions/ctc.py:516-540 (pad and eos dropped, labels left-packed, label counts)
//   s2t_gather_rows_i64     the flattened targets through a packed batch's row map (rows that hold no token: pad)
#include "common.h"

namespace {

// lengths after `n_layers` stride-2 convolutions (kernel 5, padding 2): l -> floor((l - 1) / 2) + 1; mask[b][t] = t >= l_b
__global__ __launch_bounds__(256) void subsampled_lengths_kernel(const int64_t* __restrict__ src_len, int B, int Tp, int n_layers,
                                                                  int64_t* __restrict__ len64, int32_t* __restrict__ len32,
                                                                  uint8_t* __restrict__ mask) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * Tp) return;
  const int b = idx / Tp, t = idx - b * Tp;
  int64_t l = src_len[b];
  for (int i = 0; i < n_layers; ++i) {
    const int64_t a = l - 1;
    l = (a >= 0 ? a / 2 : -((-a + 1) / 2)) + 1;  // floor division, as torch.div(..., rounding_mode="floor")
  }
  if (t == 0) {
    if (len64) len64[b] = l;
    if (len32) len32[b] = (int32_t)l;
  }
  if (mask) mask[idx] = (int64_t)t >= l ? 1 : 0;
}

// one wave per target row: pos = (inclusive count of non-pad tokens) * nonpad + pad_idx, count = non-pad tokens
__global__ __launch_bounds__(64) void token_positions_kernel(const int64_t* __restrict__ tok, int U, int64_t pad_idx,
                                                             int32_t* __restrict__ pos, int32_t* __restrict__ count) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* row = tok + (int64_t)b * U;
  int carry = 0;
  for (int u0 = 0; u0 < U; u0 += 64) {
    const int u = u0 + lane;
    const bool np = u < U && row[u] != pad_idx;
    const unsigned long long m = __ballot(np);
    const int incl = carry + __popcll(m & ((2ull << lane) - 1ull));
    if (u < U && pos) pos[(int64_t)b * U + u] = np ? incl + (int32_t)pad_idx : (int32_t)pad_idx;
    carry += __popcll(m);
  }
  if (lane == 0 && count) count[b] = carry;
}

// one wave per target row: the labels (neither pad nor eos) first, in order, the dropped tokens behind them, in order
// (torch.argsort of the drop flag, stable, then gather)
__global__ __launch_bounds__(64) void ctc_targets_kernel(const int64_t* __restrict__ target, int U, int64_t pad_idx, int64_t eos_idx,
                                                         int64_t* __restrict__ tmat, int32_t* __restrict__ count) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* row = target + (int64_t)b * U;
  int kept = 0;
  for (int u0 = 0; u0 < U; u0 += 64) {
    const int u = u0 + lane;
    const int64_t v = u < U ? row[u] : pad_idx;
    kept += __popcll(__ballot(u < U && v != pad_idx && v != eos_idx));
  }
  int ck = 0, cd = 0;
  for (int u0 = 0; u0 < U; u0 += 64) {
    const int u = u0 + lane;
    const bool in = u < U;
    const int64_t v = in ? row[u] : pad_idx;
    const bool keep = in && v != pad_idx && v != eos_idx, drop = in && !keep;
    const unsigned long long mk = __ballot(keep), md = __ballot(drop);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (keep) tmat[(int64_t)b * U + ck + __popcll(mk & below)] = v;
    if (drop) tmat[(int64_t)b * U + kept + cd + __popcll(md & below)] = v;
    ck += __popcll(mk);
    cd += __popcll(md);
  }
  if (lane == 0 && count) count[b] = kept;
}

// out[m] = map[m] >= 0 ? src[(map[m] >> 16) * U + (map[m] & 0xffff)] : fill      (every row of the buffer, live or not)
__global__ __launch_bounds__(256) void gather_rows_i64_kernel(const int64_t* __restrict__ src, const int32_t* __restrict__ map,
                                                              int64_t rows, int U, int64_t fill, int64_t* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= rows) return;
  const int32_t e = map[m];
  out[m] = e >= 0 ? src[(int64_t)(e >> 16) * U + (e & 0xffff)] : fill;
}

}  // namespace

extern "C" int s2t_subsampled_lengths(const int64_t* src_lengths, int B, int Tp, int n_layers, int64_t* len64, int32_t* len32,
                                      void* mask_u8, void* stream) {
  if (!src_lengths || B <= 0 || Tp <= 0 || n_layers < 0 || (int64_t)B * Tp >= ((int64_t)1 << 31)) return S2T_ERR_ARG;
  hipLaunchKernelGGL(subsampled_lengths_kernel, dim3((unsigned)(((int64_t)B * Tp + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src_lengths, B, Tp, n_layers, len64, len32, (uint8_t*)mask_u8);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_token_positions(const int64_t* tokens, int B, int U, int64_t pad_idx, int32_t* positions, int32_t* counts,
                                   void* stream) {
  if (!tokens || B <= 0 || U <= 0) return S2T_ERR_ARG;
  hipLaunchKernelGGL(token_positions_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, tokens, U, pad_idx, positions, counts);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_ctc_targets(const int64_t* target, int B, int U, int64_t pad_idx, int64_t eos_idx, int64_t* tmat, int32_t* counts,
                               void* stream) {
  if (!target || !tmat || tmat == target || B <= 0 || U <= 0) return S2T_ERR_ARG;
  hipLaunchKernelGGL(ctc_targets_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, target, U, pad_idx, eos_idx, tmat, counts);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_gather_rows_i64(const int64_t* src, const int32_t* map, int64_t rows, int U, int64_t fill, int64_t* out,
                                   void* stream) {
  if (!src || !map || !out || rows < 0 || U <= 0 || U > 65536) return S2T_ERR_ARG;
  if (rows == 0) return S2T_OK;
  hipLaunchKernelGGL(gather_rows_i64_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, map, rows, U,
                     fill, out);
  return S2T_LAUNCH_CHECK();
}
