// CTC head + greedy arg-max in ONE kernel for d = 256 (round 6): idx[row] = first arg-max of  x[row] W^T + b,  top_lp[row] = its
// log-probability, lse[row] = logsumexp of the row — the fp32 logits [rows][V] are never written.
//   fairseq/modules/speech_to_text/ctc.py:60-63 (ctc_projection), fairseq/models/speech_to_text/s2t_ctc.py:312-328
//   (logits -> log-softmax -> arg-max and its log-probability per frame: CTCDecoder.generate's first stage).
// Replaces, for greedy decoding, s2t_gemm with an fp32 output (51 000 x 10 000 x 4 B = 2 GB written at configuration 5a: 625 us) +
// s2t_argmax_lse (the same 2 GB read back: 450 us).
//
// Structure: the row-panel loop of tools/ubench/rowpanel_proj.hip.  A workgroup owns 64 rows (their B fragments in 128 registers
// per lane), W streams through two 64 KiB LDS stages by LDS-DMA in chunks of 128 vocabulary entries (the chunk's 128 bias values
// ride along as a ninth piece of wave 0), wave w multiplies columns 16 w .. + 15 of every chunk against all 64 rows (32 MFMAs
// 16x16x32) and folds the 16 fp32 logits a lane holds per chunk (4 row tiles x 4 columns) into a running (max, sum of
// exponentials, first arg-max) per row tile: 12 registers.  One barrier per chunk: behind it every wave's pieces of the chunk
// have landed and every wave has read the fragments of the chunk before, so the stage of chunk c + 1 is free; its DMA goes out
// between the two halves of the wave's MFMAs (tools/ubench/stream_mfma.hip: the cheapest place).  At the end the 32 partial
// states of a row (8 waves x 4 lane groups) meet in LDS and one thread per row merges them in a fixed order (ties: the lowest
// vocabulary index, as torch.max / the reference's arg-max return it).
#include "common.h"
#include "lds_dma.h"

namespace {

constexpr int D = 256, TM = 64, CW = 128;
constexpr int WBYTES = CW * 512;          // 64 KiB: one chunk of W rows
constexpr int SSTRIDE = WBYTES + 1024;    // + the chunk's bias slot (128 floats used)
constexpr int L_BYTES = 2 * SSTRIDE;      // 130 KiB; the prologue's row image (32 KiB) borrows stage 1

struct HeadArgs {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  int32_t* idx;
  float* top_lp;
  float* lse;
  const int32_t* live;   // optional device scalar: only rows < *live (a packed batch's live row count)
  int M, V;
  int64_t ldx;
};

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int N_>
__device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N_) : "memory");
}

__global__ __launch_bounds__(512, 2) void ctc_head_greedy_kernel(const HeadArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[L_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TM;
  int M = p.M;
  if (p.live) M = min(M, p.live[0]);
  if (row0 >= M) return;   // (workgroup-uniform, before any barrier)
  const int V = p.V;
  const int NC = (V + CW - 1) / CW;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const i32x4 srd = make_srd(p.w, (uint32_t)V * D * 2u);                 // rows beyond V read as zero
  const i32x4 bsrd = make_srd(p.bias, p.bias ? (uint32_t)V * 4u : 0u);  // no bias: zeros

  // DMA plan (rowpanel_proj.hip): wave w loads W rows 16 w .. + 15 of a chunk in 8 pieces of 2 rows; row u at u * 512, its 16-byte
  // k-piece s at slot s ^ (u & 15)
  uint32_t ve;
  {
    const int hi = lane >> 5, s_ = lane & 31;
    const int u = 16 * wave + hi;
    ve = (uint32_t)(u * 512 + 16 * (s_ ^ (u & 15)));
  }
  auto issue = [&](int c) __attribute__((always_inline)) {
    const uint32_t base = lds0 + (uint32_t)((c & 1) * SSTRIDE + wave * 8192);
    const uint32_t soff = (uint32_t)c * (uint32_t)WBYTES;
    dma16_off<0>(base, ve, srd, soff);
    dma16_off<1024>(base, ve ^ 32u, srd, soff);
    dma16_off<2048>(base, ve ^ 64u, srd, soff);
    dma16_off<3072>(base, ve ^ 96u, srd, soff);
    dma16_off<0>(base + 4096, ve ^ 128u, srd, soff + 4096);
    dma16_off<1024>(base + 4096, ve ^ 160u, srd, soff + 4096);
    dma16_off<2048>(base + 4096, ve ^ 192u, srd, soff + 4096);
    dma16_off<3072>(base + 4096, ve ^ 224u, srd, soff + 4096);
    if (wave == 0)   // the chunk's bias: 128 floats (lanes 32 - 63 fetch the following 128: unused)
      dma16(lds0 + (uint32_t)((c & 1) * SSTRIDE + WBYTES), (uint32_t)(lane * 16), bsrd, (uint32_t)c * (uint32_t)(CW * 4));
  };
  issue(0);
  // ---- prologue: the 64 rows -> an LDS image in stage 1 (row r at r * 512, piece c at slot c ^ (r & 15)) -> B fragments
  {
    char* xs = smem + SSTRIDE;
    const int cch = tid & 31;
    uint4 raw[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int mc = min(row0 + 16 * ps + (tid >> 5), M - 1);
      raw[ps] = *reinterpret_cast<const uint4*>(p.x + (int64_t)mc * p.ldx + 8 * cch);
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rl = 16 * ps + (tid >> 5);
      *reinterpret_cast<uint4*>(xs + rl * 512 + 16 * (cch ^ (rl & 15))) = raw[ps];
    }
  }
  __syncthreads();
  bf16x8 xn[4][8];
  {
    const char* xs = smem + SSTRIDE;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int rl = 16 * mt + x;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) xn[mt][ks] = as_frag(*reinterpret_cast<const uint4*>(xs + rl * 512 + 16 * ((4 * ks + g) ^ x)));
    }
  }
  // running state of row 16 mt + x over this lane's columns (16 wave + 4 g + r of every chunk)
  float rm[4], rs[4];
  int ri[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    rm[mt] = -INFINITY;
    rs[mt] = 0.f;
    ri[mt] = 0x7fffffff;
  }
  const int cw0 = 16 * wave;

  for (int c = 0; c < NC; ++c) {
    // chunk c has landed everywhere (the only DMA in flight here: chunk c + 1 goes out below) and every wave has read the fragments
    // of chunk c - 1 (at c = 0: of the row image, which sits in the stage chunk 1 will land in)
    wait_vm_barrier<0>();
    const char* lw = smem + (c & 1) * SSTRIDE;
    uint4 af[8];
    {
      const char* rowp = lw + (cw0 + x) * 512;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const uint4*>(rowp + 16 * ((4 * ks + g) ^ x));
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lw + WBYTES + (cw0 + 4 * g) * 4);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[ks]), xn[mt][ks], acc[mt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) issue(c + 1);   // between the halves of the MFMAs; its stage was read a chunk ago
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 4; ks < 8; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[ks]), xn[mt][ks], acc[mt], 0, 0, 0);
    // fold the chunk's 16 logits of this lane into the running states (columns beyond V do not exist: -inf)
    const int col0 = c * CW + cw0 + 4 * g;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (col0 + r < V) ? acc[mt][r] + b4[r] : -INFINITY;
      float vm = v[0];
      int am = 0;
#pragma unroll
      for (int r = 1; r < 4; ++r)
        if (v[r] > vm) {   // strictly greater: the lowest column of a tie stays
          vm = v[r];
          am = r;
        }
      const float old = rm[mt];
      if (vm > old) ri[mt] = col0 + am;   // (columns ascend inside a lane: an equal later value never replaces an earlier one)
      const float nm = fmaxf(old, vm);    // finite from chunk 0 on: a lane's columns of chunk 0 are below 128 <= V
      float sum = rs[mt] * __expf(old - nm);
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += __expf(v[r] - nm);
      rs[mt] = sum;
      rm[mt] = nm;
    }
  }
  // ---- the 32 partial states of a row (8 waves x 4 lane groups) meet in the stage the last chunk did NOT use: read for the last
  // time a chunk ago (every wave has passed the last barrier since), no DMA pending into it
  float* scr = reinterpret_cast<float*>(smem + (NC & 1) * SSTRIDE);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    float* e = scr + ((16 * mt + x) * 32 + 4 * wave + g) * 3;
    e[0] = rm[mt];
    e[1] = rs[mt];
    e[2] = __int_as_float(ri[mt]);
  }
  __syncthreads();
  if (tid < TM) {
    const float* e = scr + tid * 32 * 3;
    float Mx = -INFINITY, S = 0.f;
    int I = 0x7fffffff;
    for (int k = 0; k < 32; ++k) {   // fixed order: the merge is reproducible; ties go to the lowest vocabulary index
      const float m = e[3 * k], s_ = e[3 * k + 1];
      const int i = __float_as_int(e[3 * k + 2]);
      if (m > Mx || (m == Mx && i < I)) I = i;
      const float nm = fmaxf(Mx, m);
      S = S * __expf(Mx - nm) + s_ * __expf(m - nm);
      Mx = nm;
    }
    const int gm = row0 + tid;
    if (gm < M) {
      const float lg = __logf(S);
      if (p.idx) p.idx[gm] = I;
      if (p.top_lp) p.top_lp[gm] = -lg;       // max - logsumexp
      if (p.lse) p.lse[gm] = Mx + lg;
    }
  }
}

}  // namespace

// x: bf16 rows [M][ldx] with ldx >= 256 (the encoder output), w: bf16 [V][256] (ctc_projection.weight), bias: fp32 [V] or null.
extern "C" int s2t_ctc_head_greedy(const void* x, int64_t ldx, const void* w, const float* bias, int64_t M, int V, int32_t* idx,
                                   float* top_lp, float* lse, const int32_t* live, void* stream) {
  if (!x || !w || M < 0 || V < CW || ldx < D) return S2T_ERR_ARG;
  if (!idx && !top_lp && !lse) return S2T_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)w) & 15 || (ldx % 8)) return S2T_ERR_ALIGN;
  if ((int64_t)V * D * 2 >= ((int64_t)1 << 32) || M >= ((int64_t)1 << 31)) return S2T_ERR_UNSUPPORTED;   // 32-bit descriptor offsets
  if (M == 0) return S2T_OK;
  HeadArgs a{(const bf16_t*)x, (const bf16_t*)w, bias, idx, top_lp, lse, live, (int)M, V, ldx};
  hipLaunchKernelGGL(ctc_head_greedy_kernel, dim3((unsigned)((M + TM - 1) / TM)), dim3(512), 0, (hipStream_t)stream, a);
  return S2T_LAUNCH_CHECK();
}
