// Row-block kernels for d = 256 activations on gfx950: a workgroup owns 64 complete rows of the [B*T, 256] activation
// matrix, so everything row-wise (LayerNorm in front, bias / activation / dropout / residual / LayerNorm behind) is
// fused around the MFMA products and a chain of two products keeps its intermediate on the chip.
//
//   s2t_ffn_fused_fwd   out = residual + alpha * drop_o( W2 drop_h(act(W1 LN(x) + b1)) + b2 ) [-> LayerNorm]
//                       modules/s2t_transformer_layer.py:55-66 (FeedForwardModule), :258-265, :311-317 (macaron / final
//                       half-step residuals), :318-320 (final_norm); modules/layer_norm.py:30-35.
//
// The 16000 x 2048 hidden activation of the headline configuration (65 MB in bf16) is never written in eval mode and
// never re-read in training mode (training stores the pre-activation and the dropped activation once, for backward).
//
// Decomposition (one workgroup = 4 waves = 64 rows; 16000 rows -> 250 workgroups on 256 CUs):
//   wave (mp, fh): rows 32*mp .. +32 (two 16-row MFMA column tiles), hidden units 32*fh .. +32 of every 64-unit chunk.
//   All products are "swapped" 16x16x32 bf16 MFMAs that produce TRANSPOSED tiles (D row = output feature, D column =
//   activation row), so a lane owns ONE activation row (lane & 15):
//     G1 (chunk c):  H^T[f][m] = sum_k W1[f][k] Xn[m][k]      A = W1 rows from LDS, B = Xn fragments in registers
//                    The two 16-unit tiles of a wave interleave the hidden units in groups of four (tile ft, MFMA row
//                    4g + r  <->  unit 32 fh + 8g + 4 ft + r), so that lane (x, g) ends up with EIGHT CONSECUTIVE
//                    hidden units of activation row x in its two accumulators:
//     E1          :  bias (the initial accumulator), activation, dropout, bf16 pack — lane-local; the packed registers
//                    ARE the B fragment (k = 8g + j) of the next product, and one 16-byte store saves them for backward
//     G2 (chunk c):  Y^T[n][m] += sum_f W2[n][f] H^T[f][m]    A = W2 rows from LDS (one ds_read_b128), B = H fragment
//   Weights stream through LDS by LDS-DMA (buffer_load ... lds, no staging registers, no ds_write): per chunk 32 KiB of
//   W1 (64 rows x 512 B) and 32 KiB of W2 (256 rows x 128 B), double buffered, issued one chunk ahead.
//   LDS images are lane-linear per DMA instruction; the XOR swizzles that make the fragment reads conflict-free are
//   applied to the per-lane SOURCE address and to the read address.
//   The two fh halves of a row block hold partial Y sums; they meet in LDS (the weight buffers, free by then) where
//   the final row-wise epilogue runs on whole 512-byte rows (coalesced stores, LayerNorm statistics by shuffles).
#include "common.h"
#include "lds_dma.h"
#include "ffn_args.h"

#ifndef S2T_RB_SAVE_AUX
#define S2T_RB_SAVE_AUX 0  // cache-policy bits of the stores of the saves (z, h, dZ): 2 = nt (streaming)
#endif
#ifndef S2T_RB_SPREAD
#define S2T_RB_SPREAD 1  // LDS-DMA pieces spread over the iteration's MFMA groups (0: all at the head of the iteration)
#endif
#ifndef S2T_RBG_ORDER
#define S2T_RBG_ORDER 1  // row-block projections: 1 the DMA of chunk c+2 in front of read-out c-1's stores with a counted wait that
#endif                   // leaves those stores in flight; 0 the order of rounds 2 - 4 (stores, DMA, vmcnt(4): every store acknowledged)
#ifndef S2T_RBG_SKEW
#define S2T_RBG_SKEW 0  // experiment (measured: no gain, 19.2 against 18.4 us): waves 4 - 7 run product then read-out, waves 0 - 3 read-out then product
#endif
#ifndef S2T_RBG_STAGGER
#define S2T_RBG_STAGGER 1  // row-block projections: the workgroups of an XCD start their walk over W's chunks at different chunks
#endif
#ifndef S2T_RB_DBG
#define S2T_RB_DBG 0  // kernel-experiment switches (tools/rb_dbg_build.sh): 1 no DMA inside the loop, 2 no MFMAs, 4 no E1,
                      // 8 no z / h saves, 16 s_memtime stamps, 32 no L2 warm-up loads
#endif

namespace {

constexpr int D = 256;    // model width (fixed)
constexpr int TM = 64;    // rows per workgroup
constexpr int FC = 64;    // hidden units per chunk
constexpr int STAGE = 32768;             // bytes of one W1 or W2 chunk image
constexpr int LDS_W1 = 0;                // two W1 stages
constexpr int LDS_W2 = 2 * STAGE;        // two W2 stages
constexpr int LDS_MBOX = 4 * STAGE;      // mailbox: 2 parities x 8 waves x 2 row tiles x 64 lanes x 16 B (h | z, packed bf16)
constexpr int LDS_Z = LDS_MBOX + 16384;  // backward: two 8 KiB stages of the pre-activation tile (the mailbox slots are 8 bytes there)
constexpr int MAXF = 1 << 20;
constexpr int LDS_BYTES = LDS_MBOX + 32768;   // 160 KiB in all

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
#if S2T_RB_DBG & 2
  asm volatile("" :: "v"(a), "v"(b));
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ uint32_t pack2(float a, float b) { return bf16pack(a, b); }

// 8 fp32 -> 8 bf16, one 16-byte store
__device__ __forceinline__ void st8row(bf16_t* ptr, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(ptr) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}
#ifndef S2T_RB_SAVE_NT
#define S2T_RB_SAVE_NT 0  // experiment: the row-block projections' saves for backward (x_ln, the pre-GLU values) stored non-temporally
#endif
typedef uint32_t rb_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_save(void* ptr, uint4 o) {
#if S2T_RB_SAVE_NT
  __builtin_nontemporal_store((rb_u4){o.x, o.y, o.z, o.w}, reinterpret_cast<rb_u4*>(ptr));
#else
  *reinterpret_cast<uint4*>(ptr) = o;
#endif
}
__device__ __forceinline__ void st8row_save(bf16_t* ptr, const float (&v)[8]) {
  st16_save(ptr, make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])));
}

// swizzle key of row r (0..63) of a W1 chunk image: the 16 rows one fragment read touches (r = 32 fh + 8 (x>>2) + 4 ft +
// (x&3), x = 0..15) get 16 different keys
__device__ __forceinline__ int w1key(int r) { return (r & 3) | (((r >> 3) & 3) << 2); }

// ---------------------------------------------------------------------------------------------------------------
// Eight waves per workgroup, two per SIMD (<= 256 registers each), so that one wave's LDS latency, DMA issue and E1
// arithmetic sit beside its SIMD partner's MFMAs.  wave = (mp, fh, nh):
//   rows 32*mp .. +32 (two 16-row tiles mt), hidden units 32*fh .. +32 of every chunk, and
//   G1: the nh-th of the two interleaved 16-unit tiles of that hidden range (MFMA row 4g + r <-> unit 32 fh + 8g + 4 nh + r),
//   G2: output columns 128*nh .. +128 (8 tiles), K = the full 32 hidden units of (fh): lane (x, g)'s B fragment is
//       [its own four packed values | its partner wave's four] — the SAME lane of wave (mp, fh, 1-nh) — exchanged through
//       a 16 KiB LDS mailbox (double buffered by chunk parity; the chunk barrier orders it).
// Schedule (one barrier per chunk, the second product lags the first by one chunk):
//   iteration c:  [stores of chunk c-1's saves]  DMA W1(c+1), W2(c)
//                 G1(c)   : 16 MFMAs per wave         G2(c-1) : 16 MFMAs per wave
//                 E1(c)   : bias (initial accumulator), activation, dropout, pack; mailbox write
//                 s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier
//
// MODE 2 is the BACKWARD of the two products on the same schedule (s2t_ffn_fused_bwd):
//   dH = dY W2, dZ = alpha * drop_h(dH * act'(Z)), dXn = dZ W1.  With the TRANSPOSED weight copies W2^T [F][256] in the place
//   of W1 and W1^T [256][F] in the place of W2 the two products are the forward ones; E1 becomes the activation derivative
//   (the Z tile of the chunk arrives by one LDS-DMA instruction per wave, a chunk ahead), the single save is dZ (operand of
//   the W1 weight gradient), there is no LayerNorm prologue, bias or output dropout.
template <int MODE, int ACT, bool DROP>
__global__ __launch_bounds__(512, 2) void ffn_fused_fwd_kernel(const FfnK p) {
  constexpr bool TRAIN = MODE == 1, BWD = MODE == 2;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mp = wave & 1, fh = (wave >> 1) & 1, nh = wave >> 2;
  const int x = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TM;
  // packed batch: the live row count comes from the row map given as the mask of the trailing LayerNorm (forward: eln_lens,
  // backward: pl_lens); row blocks beyond it leave at once (workgroup-uniform, before any barrier)
  const int M = (int)s2t_live_rows(BWD ? p.pl_lens : p.eln_lens, BWD ? p.pl_T : p.eln_T, p.M), F = p.F;
  if (row0 >= M) return;
  const int nchunks = F / FC;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
#if S2T_RB_DBG & 16
  const unsigned long long k_t0 = __builtin_amdgcn_s_memtime(), k_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long k_t1 = 0, k_t2 = 0;
#endif

  const i32x4 srd1 = make_srd(p.w1, (uint32_t)F * D * 2u);
  const i32x4 srd2 = make_srd(p.w2, (uint32_t)F * D * 2u);

  // ---- DMA plans.  W1 chunk: wave w loads hidden rows 8w .. 8w+7 of the chunk, instruction i rows 8w+2i, +1
  // (512 B each): lane l -> row r = 8w + 2i + (l>>5), LDS slot s = l&31 holds 16-byte k-chunk s ^ w1key(r).
  // W2 chunk: instruction i of wave w covers output rows 32w + 8i .. +7 (128 B each): lane l -> row r = 32w+8i+(l>>3),
  // slot s = l&7 holds 16-byte f-chunk s ^ ((r>>1)&7).
  // Two per-lane offsets per weight serve all four instructions: rows 8w + 2i + hi differ between even and odd i only in
  // bit 1 of the swizzle key (byte offset ^ 32), the rest is the instruction offset i*1024 (global and LDS alike);
  // W2 rows 32w + 8i + (l>>3): odd i flips bit 2 of the key (byte offset ^ 64), i*8 rows go into the scalar offset.
  // Training flavour (SPLIT): the vector-memory work is divided by ROLE — waves 0-3 issue all the DMAs (their own share and
  // that of wave + 4), waves 4-7 all the stores of the saves.  vmcnt counts a wave's loads and stores in one order: a wave
  // that does both cannot wait for this chunk's weights without also waiting for the write acknowledgement of the stores it
  // issued a chunk earlier.  Split, the DMA waves wait for DMAs only and the store waves never wait for memory inside the
  // loop.  (Measured: 86.0 -> 83.6 us with two saves per chunk; nothing for the backward flavour's single save, which keeps
  // the symmetric form with a counted wait.)
  constexpr bool SPLIT = TRAIN;
  const bool dma_wave = !SPLIT || wave < 4;
  const int dw = SPLIT ? (wave & 3) : wave;  // the wave whose DMA share the per-lane plans describe
  constexpr int NSET = SPLIT ? 2 : 1;
  uint32_t v1e, v1o, v2e, v2o;
  {
    const int hi = lane >> 5, s = lane & 31;
    const int r = 8 * dw + hi;
    v1e = (uint32_t)(r * 512 + 16 * (s ^ w1key(r)));
    v1o = v1e ^ 32u;
    const int r2 = 32 * dw + (lane >> 3), s2 = lane & 7;
    v2e = (uint32_t)r2 * (uint32_t)(F * 2) + (uint32_t)(16 * (s2 ^ ((r2 >> 1) & 7)));
    v2o = v2e ^ 64u;
  }
  const uint32_t w2step = (uint32_t)(8 * F * 2);
  // (set 1 = the share of wave dw + 4: W1 rows + 32, W2 rows + 128 — the swizzle keys do not change)
  auto issue_w1_half = [&](int c, int half) __attribute__((always_inline)) {
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
      const uint32_t base = lds0 + LDS_W1 + (c & 1) * STAGE + (dw + 4 * set) * 4096;
      const uint32_t soff = (uint32_t)c * (FC * 512) + (uint32_t)set * (32 * 512);
      if (half == 0) {
        dma16_off<0>(base, v1e, srd1, soff);
        dma16_off<1024>(base, v1o, srd1, soff);
      } else {
        dma16_off<2048>(base, v1e, srd1, soff);
        dma16_off<3072>(base, v1o, srd1, soff);
      }
    }
  };
  auto issue_w2_half = [&](int c, int half) __attribute__((always_inline)) {
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
      const uint32_t base = lds0 + LDS_W2 + (c & 1) * STAGE + (dw + 4 * set) * 4096;
      const uint32_t soff = (uint32_t)c * (FC * 2) + (uint32_t)set * (uint32_t)(128 * F * 2);
      if (half == 0) {
        dma16(base, v2e, srd2, soff);
        dma16(base + 1024, v2o, srd2, soff + w2step);
      } else {
        dma16(base + 2048, v2e, srd2, soff + 2 * w2step);
        dma16(base + 3072, v2o, srd2, soff + 3 * w2step);
      }
    }
  };
  auto issue_w1 = [&](int c) __attribute__((always_inline)) {
    if (dma_wave) {
      issue_w1_half(c, 0);
      issue_w1_half(c, 1);
    }
  };
  auto issue_w2 = [&](int c) __attribute__((always_inline)) {
    if (dma_wave) {
      issue_w2_half(c, 0);
      issue_w2_half(c, 1);
    }
  };
  // backward: the [64 rows][64 units] bf16 tile of Z of chunk c, 16-byte piece pp of row r at cell r*8 + (pp ^ ((r>>1)&7));
  // lane l of wave w fills cell 64 w + l.  Rows >= M read as zero (descriptor bounds).
  const i32x4 srdz = make_srd(p.z, BWD ? (uint32_t)M * (uint32_t)F * 2u : 0u);
  uint32_t vz = 0;
  if constexpr (BWD) {
    const int zr = 8 * dw + (lane >> 3);
    vz = (uint32_t)(row0 + zr) * (uint32_t)(F * 2) + (uint32_t)(16 * ((lane & 7) ^ ((zr >> 1) & 7)));
  }
  auto issue_z = [&](int c) __attribute__((always_inline)) {
    if constexpr (BWD) {
      dma16(lds0 + LDS_Z + (c & 1) * 8192 + wave * 1024, vz, srdz, (uint32_t)c * (FC * 2));
    }
  };
  issue_w1(0);
  issue_w2(0);
  issue_z(0);

  // ---- prologue: LayerNorm of the 64 rows, one 16-byte piece (8 columns) per thread and pass: 32 lanes per row, 16 rows
  // per pass.  The normalised bf16 tile is staged in the (still unused) second W1 buffer with the W1 image's swizzle:
  // 16-byte chunk c of row r at r*512 + 16*(c ^ (r & 15)).
  {
    const bf16_t* X = reinterpret_cast<const bf16_t*>(p.x);
    char* stage = smem + LDS_W1 + STAGE;
    const int cch = tid & 31;  // 16-byte column chunk
    float gm[8], bt[8];
    if (p.ln_gamma) {
      const float4 g0 = *reinterpret_cast<const float4*>(p.ln_gamma + 8 * cch);
      const float4 g1 = *reinterpret_cast<const float4*>(p.ln_gamma + 8 * cch + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.ln_beta + 8 * cch);
      const float4 b1v = *reinterpret_cast<const float4*>(p.ln_beta + 8 * cch + 4);
      gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
      bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1v.x; bt[5] = b1v.y; bt[6] = b1v.z; bt[7] = b1v.w;
    }
    uint4 raw[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int mc = min(row0 + 16 * ps + (tid >> 5), M - 1);
      raw[ps] = *reinterpret_cast<const uint4*>(X + (int64_t)mc * D + 8 * cch);
    }
    if constexpr (BWD) {
      if (p.pl_y) {
        // ---- backward of the trailing LayerNorm on the way in (s2t_layernorm_bwd's arithmetic, 32 lanes per row):
        //   d = masked ? 0 : dout;  dres = rstd * (d*gamma - mean(d*gamma) - xhat * mean(d*gamma*xhat));  staged tile =
        //   dropout_o(dres).  dgamma / dbeta: per-thread sums over its 4 rows, the 16 row groups meet in the (idle)
        //   mailbox region after the prologue barrier.
        const bf16_t* Yp = reinterpret_cast<const bf16_t*>(p.pl_y);
        bf16_t* DR = reinterpret_cast<bf16_t*>(p.pl_dres);
        bf16_t* DY = reinterpret_cast<bf16_t*>(p.pl_dy);
        const uint64_t key_e = DY ? s2t_drop_key(p.drop_seed, p.drop_o_site) : 0ull;
        const uint32_t th_e = s2t_drop_thresh(p.drop_o_p);
        const float inv_e = s2t_drop_scale(p.drop_o_p);
        float pg[8], ag[8], ab[8];
        {
          const float4 g0 = *reinterpret_cast<const float4*>(p.pl_gamma + 8 * cch);
          const float4 g1 = *reinterpret_cast<const float4*>(p.pl_gamma + 8 * cch + 4);
          pg[0] = g0.x; pg[1] = g0.y; pg[2] = g0.z; pg[3] = g0.w; pg[4] = g1.x; pg[5] = g1.y; pg[6] = g1.z; pg[7] = g1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ag[j] = ab[j] = 0.f;
        uint4 yraw[4];
        float mu4[4], rs4[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int mc = min(row0 + 16 * ps + (tid >> 5), M - 1);
          yraw[ps] = *reinterpret_cast<const uint4*>(Yp + (int64_t)mc * D + 8 * cch);
          mu4[ps] = p.pl_mean[mc];
          rs4[ps] = p.pl_rstd[mc];
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int rl = 16 * ps + (tid >> 5);
          const int m = row0 + rl;
          const bool live = m < M;
          const bool masked = !live || (p.pl_lens && s2t_row_masked32(p.pl_lens, p.pl_T, (uint32_t)m));
          const uint32_t dw4[4] = {raw[ps].x, raw[ps].y, raw[ps].z, raw[ps].w};
          const uint32_t yw4[4] = {yraw[ps].x, yraw[ps].y, yraw[ps].z, yraw[ps].w};
          float dgv[8], xh[8];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int j = 2 * q + e;
              const float dv = masked ? 0.f : __uint_as_float(e ? (dw4[q] & 0xffff0000u) : (dw4[q] << 16));
              const float yv = __uint_as_float(e ? (yw4[q] & 0xffff0000u) : (yw4[q] << 16));
              xh[j] = (yv - mu4[ps]) * rs4[ps];
              dgv[j] = dv * pg[j];
              s1 += dgv[j];
              s2 += dgv[j] * xh[j];
              ag[j] += dv * xh[j];
              ab[j] += dv;
            }
          }
          s1 = s2t_sum32(s1);
          s2 = s2t_sum32(s2);
          s1 *= 1.0f / D;
          s2 *= 1.0f / D;
          uint32_t rw[4], yw[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            rw[q] = pack2(rs4[ps] * (dgv[2 * q] - s1 - xh[2 * q] * s2), rs4[ps] * (dgv[2 * q + 1] - s1 - xh[2 * q + 1] * s2));
          uint4 o = make_uint4(rw[0], rw[1], rw[2], rw[3]);
          if (live) *reinterpret_cast<uint4*>(DR + (int64_t)m * D + 8 * cch) = o;
          if (DY) {  // the dropped image of the STORED bf16 dres, as s2t_dropout would make it
            uint32_t r16[8];
            s2t_rand_run_even32<8>(key_e, (uint32_t)m * D + (uint32_t)(8 * cch), r16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float lo = __uint_as_float(rw[q] << 16), hi = __uint_as_float(rw[q] & 0xffff0000u);
              yw[q] = pack2(r16[2 * q] >= th_e ? lo * inv_e : 0.f, r16[2 * q + 1] >= th_e ? hi * inv_e : 0.f);
            }
            o = make_uint4(yw[0], yw[1], yw[2], yw[3]);
            if (live) *reinterpret_cast<uint4*>(DY + (int64_t)m * D + 8 * cch) = o;
          }
          raw[ps] = o;  // what the plain path below stages
        }
        // [2][16][256] fp32 = 32 KiB in the second W2 stage: idle until chunk 1's W2 DMA (the mailbox region is not: the
        // Z tile of chunk 0 is landing in its upper half)
        float* red = reinterpret_cast<float*>(smem + LDS_W2 + STAGE);
        const int grp = tid >> 5;
        *reinterpret_cast<float4*>(red + (0 * 16 + grp) * 256 + 8 * cch) = make_float4(ag[0], ag[1], ag[2], ag[3]);
        *reinterpret_cast<float4*>(red + (0 * 16 + grp) * 256 + 8 * cch + 4) = make_float4(ag[4], ag[5], ag[6], ag[7]);
        *reinterpret_cast<float4*>(red + (1 * 16 + grp) * 256 + 8 * cch) = make_float4(ab[0], ab[1], ab[2], ab[3]);
        *reinterpret_cast<float4*>(red + (1 * 16 + grp) * 256 + 8 * cch + 4) = make_float4(ab[4], ab[5], ab[6], ab[7]);
      }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rl = 16 * ps + (tid >> 5);
      const int m = row0 + rl;
      uint4 o = raw[ps];
      if (p.ln_gamma) {
        const uint32_t w4[4] = {o.x, o.y, o.z, o.w};
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[2 * q] = __uint_as_float(w4[q] << 16);
          v[2 * q + 1] = __uint_as_float(w4[q] & 0xffff0000u);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[j];
        sum = s2t_sum32(sum);
        const float mean = sum * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dd = v[j] - mean;
          sq += dd * dd;
        }
        sq = s2t_sum32(sq);
        const float rstd = rsqrtf(sq * (1.0f / D) + p.ln_eps);
        uint32_t ow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          ow[q] = pack2((v[2 * q] - mean) * rstd * gm[2 * q] + bt[2 * q], (v[2 * q + 1] - mean) * rstd * gm[2 * q + 1] + bt[2 * q + 1]);
        o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        if constexpr (TRAIN) {
          if (m < M) {
            if (p.x_ln) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.x_ln) + (int64_t)m * D + 8 * cch) = o;
            if (cch == 0) {
              if (p.ln_mean) p.ln_mean[m] = mean;
              if (p.ln_rstd) p.ln_rstd[m] = rstd;
            }
          }
        }
      }
      *reinterpret_cast<uint4*>(stage + rl * 512 + 16 * (cch ^ (rl & 15))) = o;
    }
  }
  // chunk 0 of both weights, b1, the staged tile and the compiler's own prologue loads / stores have landed (the builtin
  // form also tells the compiler that its loads are complete: no vmcnt waits of its own inside the loop)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  if constexpr (BWD) {
    if (p.pl_y) {  // column sums of the trailing LayerNorm's parameter gradients: 512 atomics into one replica
      const float* red = reinterpret_cast<const float*>(smem + LDS_W2 + STAGE);
      const int which = tid >> 8, c = tid & 255;
      float sum = 0.f;
#pragma unroll
      for (int gI = 0; gI < 16; ++gI) sum += red[(which * 16 + gI) * 256 + c];
      atomicAdd(p.pl_ws + (int64_t)(blockIdx.x % p.pl_replicas) * 512 + which * 256 + c, sum);
    }
  }
  // this wave's 32 rows as B fragments: lane (x, g) owns k = 32*ks + 8g .. +8 of row 32 mp + 16 mt + x
  bf16x8 xn[2][8];
  {
    const char* stage = smem + LDS_W1 + STAGE;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int rl = 32 * mp + 16 * mt + x;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        xn[mt][ks] = as_frag(*reinterpret_cast<const uint4*>(stage + rl * 512 + 16 * ((4 * ks + g) ^ x)));
    }
  }
  // every wave holds its fragments before the DMA of chunk 1 overwrites the staging buffer
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  f32x4 yacc[8][2];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) yacc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint64_t key_h = DROP ? s2t_drop_key(p.drop_seed, p.drop_h_site) : 0ull;
  const uint32_t th_h = s2t_drop_thresh(p.drop_h_p);
  const float inv_h = s2t_drop_scale(p.drop_h_p);

  // mailbox: [parity][wave][mt][lane] 16 bytes: packed h (8 B, what the SIMD partner's G2 needs) | packed z (8 B, training)
  char* mbox = smem + LDS_MBOX;
  const int partner = wave ^ 4;
  constexpr int MSLOT = BWD ? 8 : 16;
  auto mslot = [&](int c, int w, int mt, int ln) __attribute__((always_inline)) {
    return mbox + ((((c & 1) * 8 + w) * 2 + mt) * 64 + ln) * MSLOT;
  };

  // G1 of chunk c: this wave's 16-unit tile for both row tiles (fragments in two groups of four k-steps)
  const int g1row = 32 * fh + 8 * (x >> 2) + 4 * nh + (x & 3);
  auto g1_read = [&](int c, int k0, uint4 (&af)[4]) __attribute__((always_inline)) {
    const char* l1 = smem + LDS_W1 + (c & 1) * STAGE;
    const int key = w1key(g1row);
#pragma unroll
    for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const uint4*>(l1 + g1row * 512 + 16 * ((4 * (k0 + j) + g) ^ key));
  };
  auto g1_mma = [&](int k0, const uint4 (&af)[4], f32x4 (&hacc)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hacc[0] = mfma16(as_frag(af[j]), xn[0][k0 + j], hacc[0]);
      hacc[1] = mfma16(as_frag(af[j]), xn[1][k0 + j], hacc[1]);
    }
  };
  // b1[c*64 + 32 fh + 8 g + 4 nh + r]: four wave-uniform 16-byte loads per chunk through the scalar memory path (no LDS,
  // no vector memory operation that a counted wait would have to know about), issued ONE CHUNK AHEAD of their use; the
  // lane picks the one of its g.  The bias rides into the product as the initial accumulator.
  struct Bias4 {
    f32x4 q[4];
  };
  auto bias_load = [&](int c) __attribute__((always_inline)) -> Bias4 {
    if constexpr (BWD) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      return Bias4{{zero, zero, zero, zero}};
    }
    const uint64_t ba = (uint64_t)(p.b1 + c * FC + 32 * fh + 4 * nh);  // wave-uniform; made provably so for the compiler
    const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ba >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ba);
    typedef const __attribute__((address_space(4))) f32x4* cptr4;  // constant address space: s_load_dwordx4
    const cptr4 bp = (cptr4)bu;
    return Bias4{{bp[0], bp[2], bp[4], bp[6]}};
  };
  auto g1_bias = [&](const Bias4& b, f32x4 (&hacc)[2]) __attribute__((always_inline)) {
    const f32x4 bb = g == 0 ? b.q[0] : (g == 1 ? b.q[1] : (g == 2 ? b.q[2] : b.q[3]));
    hacc[0] = bb;
    hacc[1] = bb;
  };
  // backward: this lane's four pre-activation values of chunk c, row tile mt (8 bytes of the staged Z tile)
  auto z_read = [&](int c, uint2 (&zq)[2]) __attribute__((always_inline)) {
    if constexpr (BWD) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rl = 32 * mp + 16 * mt + x;
        const int cell = rl * 8 + ((4 * fh + g) ^ ((rl >> 1) & 7));
        zq[mt] = *reinterpret_cast<const uint2*>(smem + LDS_Z + (c & 1) * 8192 + cell * 16 + 8 * nh);
      }
    }
  };
  // lane (x, g): v[r] = H[row x of tile mt][unit c*64 + 32 fh + 8 g + 4 nh + r]; packed halves go to the mailbox
  // (backward: zp holds the lane's Z values on entry, v = dH, the packed result is dZ)
  // dropout multipliers (scale or 0) of this lane's 2 x 4 hidden values of chunk c: they depend on indices only, so they are
  // drawn EARLY in the iteration (beside the first G1 products) and only applied in E1 — the hash's multiply chain leaves the
  // dependent chain G1 -> E1 -> mailbox -> barrier that paces the loop
  auto drop_masks = [&](int c, float (&ms)[2][4]) __attribute__((always_inline)) {
    if constexpr (DROP) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = row0 + 32 * mp + 16 * mt + x;
        // (element index m*F + f: even and, by the launcher's check, below 2^32)
        const uint32_t base = (uint32_t)m * (uint32_t)F + (uint32_t)(c * FC + 32 * fh + 8 * g + 4 * nh);
        uint32_t r16[4];
        s2t_rand_run_even32<4>(key_h, base, r16);
#pragma unroll
        for (int r = 0; r < 4; ++r) ms[mt][r] = r16[r] >= th_h ? inv_h : 0.f;
      }
    }
  };
  auto e1 = [&](int c, const f32x4 (&hacc)[2], uint2 (&zp)[2], uint2 (&hp)[2], const float (&ms)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = hacc[mt][r];
      if constexpr (TRAIN) zp[mt] = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
      if constexpr (BWD) {
        const float zf[4] = {__uint_as_float(zp[mt].x << 16), __uint_as_float(zp[mt].x & 0xffff0000u),
                             __uint_as_float(zp[mt].y << 16), __uint_as_float(zp[mt].y & 0xffff0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= act_grad(ACT, zf[r]);
      } else if constexpr ((S2T_RB_DBG & 4) != 0) {
      } else if constexpr (ACT == S2T_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if constexpr (ACT == S2T_ACT_SWISH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * sigmoidf_(v[r]);
      }
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= ms[mt][r];
      }
      if constexpr (BWD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
      }
      hp[mt] = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
      if constexpr (TRAIN) *reinterpret_cast<uint4*>(mslot(c, wave, mt, lane)) = make_uint4(hp[mt].x, hp[mt].y, zp[mt].x, zp[mt].y);
      else *reinterpret_cast<uint2*>(mslot(c, wave, mt, lane)) = hp[mt];
    }
  };
  // G2 of chunk c: own packed values hp + the partner's from the mailbox (B fragment: k = 8g + j <-> unit 32 fh + 8g + j)
  auto g2_hb = [&](int c, const uint2 (&hp)[2], bf16x8 (&hb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const uint2 o = *reinterpret_cast<const uint2*>(mslot(c, partner, mt, lane));
      hb[mt] = nh == 0 ? as_frag(make_uint4(hp[mt].x, hp[mt].y, o.x, o.y)) : as_frag(make_uint4(o.x, o.y, hp[mt].x, hp[mt].y));
    }
  };
  // row n = 128 nh + 16 nt + x: 16-byte f-chunk 4 fh + g at slot (4 fh + g) ^ ((n >> 1) & 7), (n >> 1) & 7 == (x >> 1) & 7
  const int g2off = (128 * nh + x) * 128 + 16 * ((4 * fh + g) ^ ((x >> 1) & 7));
  auto g2_read = [&](int c, int n0, uint4 (&af)[4]) __attribute__((always_inline)) {
    const char* pa = smem + LDS_W2 + (c & 1) * STAGE + g2off;
#pragma unroll
    for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const uint4*>(pa + (n0 + j) * 2048);
  };
  auto g2_mma = [&](int n0, const uint4 (&af)[4], const bf16x8 (&hb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      yacc[n0 + j][0] = mfma16(as_frag(af[j]), hb[0], yacc[n0 + j][0]);
      yacc[n0 + j][1] = mfma16(as_frag(af[j]), hb[1], yacc[n0 + j][1]);
    }
  };
  // Saves for backward of chunk c, one chunk after its mailbox slots were written (the chunk barrier lies between): thread
  // (row rr = tid >> 3, piece pp = tid & 7) assembles the 8 hidden units 8 pp .. 8 pp + 7 of its row from the two SIMD
  // partners' slots and stores 16 bytes of z and 16 bytes of h: a wave instruction covers 8 whole 128-byte lines.
  // ALWAYS exactly two store instructions per wave and chunk (the closing wait of the loop counts on it): buffer stores
  // whose descriptor drops rows >= M (and everything when the tensor was not asked for).
  const __amdgpu_buffer_rsrc_t zsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.z), 0, p.z ? (int)((uint32_t)M * (uint32_t)F * 2u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t hsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.h), 0, p.h ? (int)((uint32_t)M * (uint32_t)F * 2u) : 0, 0x00020000);
  // (SPLIT: only the store waves 4-7 execute this, thread st = tid - 256 takes rows st >> 3 and (st >> 3) + 32; otherwise
  // thread tid takes row tid >> 3)
  auto save = [&](int c) __attribute__((always_inline)) {
    if (SPLIT && dma_wave) return;
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    const int pp = tid & 7;
#pragma unroll
    for (int half = 0; half < (SPLIT ? 2 : 1); ++half) {
      const int rr = SPLIT ? ((tid - 256) >> 3) + 32 * half : (tid >> 3);
      const int w0 = (rr >> 5) + 2 * (pp >> 2);  // wave (mp, fh, nh = 0); its partner is w0 + 4
      const int smt = (rr >> 4) & 1, sl = 16 * (pp & 3) + (rr & 15);
      const uint32_t o = ((uint32_t)(row0 + rr) * (uint32_t)F + (uint32_t)(c * FC + 8 * pp)) * 2u;
      if constexpr (BWD) {  // 16 bytes of dZ (units 8 pp .. 8 pp + 7 of row rr) into p.h
        const uint2 lo = *reinterpret_cast<const uint2*>(mslot(c, w0, smt, sl));
        const uint2 hi = *reinterpret_cast<const uint2*>(mslot(c, w0 + 4, smt, sl));
        __builtin_amdgcn_raw_buffer_store_b128((u32x4s){lo.x, lo.y, hi.x, hi.y}, hsrd, o, 0, S2T_RB_SAVE_AUX);
      }
      if constexpr (TRAIN && !(S2T_RB_DBG & 8)) {
        const uint4 lo = *reinterpret_cast<const uint4*>(mslot(c, w0, smt, sl));      // units 8 pp + 0..3: h | z
        const uint4 hi = *reinterpret_cast<const uint4*>(mslot(c, w0 + 4, smt, sl));  // units 8 pp + 4..7
        __builtin_amdgcn_raw_buffer_store_b128((u32x4s){lo.z, lo.w, hi.z, hi.w}, zsrd, o, 0, S2T_RB_SAVE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4s){lo.x, lo.y, hi.x, hi.y}, hsrd, o, 0, S2T_RB_SAVE_AUX);
      }
    }
  };

  // L2 warm-up two chunks ahead: inside a model every FFN brings its own 2 MiB of weights, whose first touch per XCD is
  // an HBM miss (~2 us) that a DMA issued one chunk ahead cannot hide.  One 4-byte load per lane touches each 128-byte line
  // of chunk c+2 (256 lines of W1, 256 of W2); its value is never used.  The load is the wave's YOUNGEST vector-memory
  // operation at the closing wait of the iteration and stays in flight across it (counted vmcnt).
  uint32_t pf_sink = 0;
  const char* pf_base = tid < 256 ? reinterpret_cast<const char*>(p.w1) + (size_t)tid * 128
                                  : reinterpret_cast<const char*>(p.w2) + (size_t)(tid - 256) * ((size_t)F * 2);
  const uint32_t pf_step = tid < 256 ? (uint32_t)(FC * 512) : (uint32_t)(FC * 2);
  auto l2_prefetch = [&](int c) __attribute__((always_inline)) {
    const char* a = pf_base + (size_t)c * pf_step;
    asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(a) : "memory");
  };
#if S2T_RB_DBG & 16
  unsigned long long stamp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                   \
  do {                                             \
    __builtin_amdgcn_sched_barrier(0);             \
    stamp[i] = __builtin_amdgcn_s_memtime();       \
    __builtin_amdgcn_sched_barrier(0);             \
  } while (0)
#else
#define STAMP(i)
#endif
#if S2T_RB_DBG & 16
  __builtin_amdgcn_sched_barrier(0);
  k_t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#endif
  uint2 zp[2], hp[2];
  {
    f32x4 hacc[2];
    uint4 a0[4], a1[4];
    if (nchunks > 1) {
      issue_w1(1);
      issue_z(1);
    }
    g1_bias(bias_load(0), hacc);
    g1_read(0, 0, a0);
    g1_read(0, 4, a1);
    z_read(0, zp);
    float ms0[2][4];
    drop_masks(0, ms0);
    g1_mma(0, a0, hacc);
    g1_mma(4, a1, hacc);
    e1(0, hacc, zp, hp, ms0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  Bias4 bcur = bias_load(nchunks > 1 ? 1 : 0);
  for (int c = 1; c < nchunks; ++c) {
    // The first fragment reads go out before anything else; the DMAs of the next chunks follow (they must be OLDER than
    // the chunk's saves, see the closing wait), then the MFMA groups with the next group's reads behind them.
    // sched_barrier pins the order of the groups.
    const bool more = c + 1 < nchunks;
    f32x4 hacc[2];
    uint2 zn[2], hn[2];
    bf16x8 hb[2];
    uint4 a0[4], a1[4];
    STAMP(0);
    g1_bias(bcur, hacc);
    bcur = bias_load(c + 1 < nchunks ? c + 1 : c);
    g2_hb(c - 1, hp, hb);
    g1_read(c, 0, a0);
    g1_read(c, 4, a1);
    z_read(c, zn);
    __builtin_amdgcn_sched_barrier(0);
    // The 8 (SPLIT: 16) LDS-DMA pieces of the iteration are spread over its four MFMA groups: a piece issued into a phase
    // that already carries the fragment reads and other pieces costs the wave 100+ cycles of issue, one issued behind a
    // group of MFMAs a fraction of that.  The saves and the warm-up load follow the last piece (they must be the wave's
    // YOUNGEST vector-memory operations at the closing wait).
#if S2T_RB_SPREAD
#if !(S2T_RB_DBG & 1)
    if (more) {
      if (dma_wave && S2T_RB_SPREAD != 2) issue_w1_half(c + 1, 0);
      issue_z(c + 1);
    }
#endif
#else
#if !(S2T_RB_DBG & 1)
    if (more) {
      issue_w1(c + 1);
      issue_z(c + 1);
    }
    issue_w2(c);
#endif
    save(c - 1);
#endif
#if !(S2T_RB_DBG & 32)
    const bool pf = c + 2 < nchunks && blockIdx.x < 8;  // one workgroup per XCD group warms the XCD's L2 for all
#else
    const bool pf = false;
#endif
#if !S2T_RB_SPREAD
    if (pf) l2_prefetch(c + 2);
#endif
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    float ms[2][4];
    drop_masks(c, ms);
    g1_mma(0, a0, hacc);
    g2_read(c - 1, 0, a0);
#if S2T_RB_SPREAD && !(S2T_RB_DBG & 1)
    if (more && dma_wave) {
      if (S2T_RB_SPREAD == 2) issue_w1_half(c + 1, 0);
      issue_w1_half(c + 1, 1);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    STAMP(2);
    g1_mma(4, a1, hacc);
    g2_read(c - 1, 4, a1);
#if S2T_RB_SPREAD && !(S2T_RB_DBG & 1)
    if (dma_wave) {
      issue_w2_half(c, 0);
      if (S2T_RB_SPREAD == 3) issue_w2_half(c, 1);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    g2_mma(0, a0, hb);
#if S2T_RB_SPREAD
#if !(S2T_RB_DBG & 1)
    if (dma_wave && S2T_RB_SPREAD != 3) issue_w2_half(c, 1);
#endif
    save(c - 1);
    if (pf) l2_prefetch(c + 2);
    __builtin_amdgcn_sched_barrier(0);
#endif
    e1(c, hacc, zn, hn, ms);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(4);
    g2_mma(4, a1, hb);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      zp[mt] = zn[mt];
      hp[mt] = hn[mt];
    }
    STAMP(7);
    // what must have landed are this wave's DMAs (the warm-up load, its youngest operation, may stay in flight); the
    // store waves have none and do not wait for memory
    if constexpr (BWD) {  // symmetric: every wave has its DMAs, then one dZ store (+ the warm-up load) that may stay in flight
      if (pf) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (dma_wave) {
      if (pf) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    STAMP(8);
#if S2T_RB_DBG & 16
    if (tid == 0 && p.eln_mean && !p.eln_gamma && (blockIdx.x == 0 || blockIdx.x == 100))
      (reinterpret_cast<unsigned long long*>(p.eln_mean) + 256 + (blockIdx.x ? 64 : 0))[c] = stamp[8] - stamp[0];
#endif
  }
#if S2T_RB_DBG & 16
  __builtin_amdgcn_sched_barrier(0);
  k_t2 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) :: "memory");  // the warm-up loads may no longer touch the register
  save(nchunks - 1);
  {
    bf16x8 hb[2];
    uint4 b0[4], b1[4];
    g2_hb(nchunks - 1, hp, hb);
    g2_read(nchunks - 1, 0, b0);
    g2_read(nchunks - 1, 4, b1);
    g2_mma(0, b0, hb);
    g2_mma(4, b1, hb);
  }
  __syncthreads();

  // the residual rows of this thread's epilogue passes travel during the LDS exchange below (issued inside the pass loop they
  // were one global round trip per pass): thread (wave, hi, s) handles rows 8 wave + 2 ps + hi, columns 4s.. and 128 + 4s..
  uint2 rpre[4][2];
  uint2 xpre[4][2];     // backward flavour: the LayerNorm input rows (lb_x) and the residual-branch gradient (lb_dres)
  float mupre[4], rspre[4];
  if constexpr (BWD) {
    if (p.lb_x) {
      const bf16_t* Xp = reinterpret_cast<const bf16_t*>(p.lb_x);
      const bf16_t* Dp = reinterpret_cast<const bf16_t*>(p.lb_dres);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int mc = min(row0 + 8 * wave + 2 * ps + (lane >> 5), M - 1);
        mupre[ps] = p.lb_mean[mc];
        rspre[ps] = p.lb_rstd[mc];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          xpre[ps][q] = *reinterpret_cast<const uint2*>(Xp + (int64_t)mc * D + 128 * q + 4 * (lane & 31));
          rpre[ps][q] = Dp ? *reinterpret_cast<const uint2*>(Dp + (int64_t)mc * D + 128 * q + 4 * (lane & 31)) : make_uint2(0, 0);
        }
      }
    }
  }
  if constexpr (!BWD) {
    if (p.residual) {
      const bf16_t* Rp = reinterpret_cast<const bf16_t*>(p.residual);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int mc = min(row0 + 8 * wave + 2 * ps + (lane >> 5), M - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) rpre[ps][q] = *reinterpret_cast<const uint2*>(Rp + (int64_t)mc * D + 128 * q + 4 * (lane & 31));
      }
    }
  }
  // ---- partial sums of the two hidden halves meet in LDS: region fh, fp32 [64 rows][256], 16-byte chunk cc of row m at
  // m*1024 + 16*(cc ^ (m & 7))
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int ml = 32 * mp + 16 * mt + x;
    char* rowp = smem + fh * 65536 + ml * 1024;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<f32x4*>(rowp + 16 * ((32 * nh + 4 * nt + g) ^ (ml & 7))) = yacc[nt][mt];
  }
  __syncthreads();

  // ---- row epilogue: wave w, pass ps: rows 8w + 2ps + (lane >> 5); lane s = lane & 31 owns columns 4s..4s+3 and
  // 128+4s..+3 of its row
  {
    const int hi = lane >> 5, s = lane & 31;
    const uint64_t key_o = DROP ? s2t_drop_key(p.drop_seed, p.drop_o_site) : 0ull;
    const uint32_t th_o = s2t_drop_thresh(p.drop_o_p);
    const float inv_o = s2t_drop_scale(p.drop_o_p);
    const bf16_t* R = reinterpret_cast<const bf16_t*>(p.residual);
    bf16_t* Y = reinterpret_cast<bf16_t*>(p.y);
    bf16_t* YL = reinterpret_cast<bf16_t*>(p.y_ln);
    float b2v[2][4], eg[2][4], eb[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 t = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + 128 * q + 4 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
      b2v[q][0] = t.x; b2v[q][1] = t.y; b2v[q][2] = t.z; b2v[q][3] = t.w;
      if (p.eln_gamma) {
        const float4 a = *reinterpret_cast<const float4*>(p.eln_gamma + 128 * q + 4 * s);
        const float4 b = *reinterpret_cast<const float4*>(p.eln_beta + 128 * q + 4 * s);
        eg[q][0] = a.x; eg[q][1] = a.y; eg[q][2] = a.z; eg[q][3] = a.w;
        eb[q][0] = b.x; eb[q][1] = b.y; eb[q][2] = b.z; eb[q][3] = b.w;
      }
    }
    if constexpr (BWD) {
      if (p.lb_x) {
        // ---- backward of the block's leading LayerNorm on the fp32 dXn rows (s2t_layernorm_bwd's arithmetic):
        //   dx = rstd * (dxn*gamma - mean(dxn*gamma) - xhat * mean(dxn*gamma*xhat)) + dres   [+ its dropped copy]
        //   dgamma += sum_rows dxn * xhat, dbeta += sum_rows dxn: lane sums over its 4 rows, 16 (wave, half) groups meet
        //   in LDS (the mailbox region, idle by now), 512 atomics per workgroup into one replica of the workspace
        const bf16_t* X = reinterpret_cast<const bf16_t*>(p.lb_x);
        const bf16_t* DR = reinterpret_cast<const bf16_t*>(p.lb_dres);
        bf16_t* DX = reinterpret_cast<bf16_t*>(p.lb_dx);
        bf16_t* DXD = reinterpret_cast<bf16_t*>(p.lb_dx_drop);
        const uint64_t key_u = DXD ? s2t_drop_key(p.drop_seed, p.lb_drop_site) : 0ull;
        const uint32_t th_u = s2t_drop_thresh(p.lb_drop_p);
        const float inv_u = s2t_drop_scale(p.lb_drop_p);
        float gmm[2][4], ag[2][4], ab[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(p.lb_gamma + 128 * q + 4 * s);
          gmm[q][0] = t.x; gmm[q][1] = t.y; gmm[q][2] = t.z; gmm[q][3] = t.w;
#pragma unroll
          for (int r = 0; r < 4; ++r) ag[q][r] = ab[q][r] = 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int ml = 8 * wave + 2 * ps + hi;
          const int m = row0 + ml;
          const bool live = m < M;
          const float mu = mupre[ps], rs = rspre[ps];
          float dv[2][4], xh[2][4], dg[2][4], rr[2][4];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int cc = 32 * q + s;
            const f32x4 a = *reinterpret_cast<const f32x4*>(smem + ml * 1024 + 16 * (cc ^ (ml & 7)));
            const f32x4 b = *reinterpret_cast<const f32x4*>(smem + 65536 + ml * 1024 + 16 * (cc ^ (ml & 7)));
            const uint2 tx = xpre[ps][q], tr = rpre[ps][q];
            const float xv[4] = {__uint_as_float(tx.x << 16), __uint_as_float(tx.x & 0xffff0000u),
                                 __uint_as_float(tx.y << 16), __uint_as_float(tx.y & 0xffff0000u)};
            rr[q][0] = __uint_as_float(tr.x << 16); rr[q][1] = __uint_as_float(tr.x & 0xffff0000u);
            rr[q][2] = __uint_as_float(tr.y << 16); rr[q][3] = __uint_as_float(tr.y & 0xffff0000u);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dv[q][r] = live ? a[r] + b[r] : 0.f;
              xh[q][r] = (xv[r] - mu) * rs;
              dg[q][r] = dv[q][r] * gmm[q][r];
              s1 += dg[q][r];
              s2 += dg[q][r] * xh[q][r];
              ag[q][r] += dv[q][r] * xh[q][r];
              ab[q][r] += dv[q][r];
            }
          }
          s1 = s2t_sum32(s1);
          s2 = s2t_sum32(s2);
          s1 *= 1.0f / D;
          s2 *= 1.0f / D;
          if (live) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              float o4[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) o4[r] = bf2f(f2bf(rs * (dg[q][r] - s1 - xh[q][r] * s2) + rr[q][r]));
              st4_from_f32<bf16_t>(DX + (int64_t)m * D + 128 * q + 4 * s, o4);
              if (DXD) {  // the dropped image of the STORED bf16 dx, as s2t_dropout would make it
                uint32_t r16[4];
                s2t_rand_run_even32<4>(key_u, (uint32_t)m * D + (uint32_t)(128 * q + 4 * s), r16);
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = r16[r] >= th_u ? o4[r] * inv_u : 0.f;
                st4_from_f32<bf16_t>(DXD + (int64_t)m * D + 128 * q + 4 * s, o4);
              }
            }
          }
        }
        float* red = reinterpret_cast<float*>(smem + LDS_MBOX);  // [2][16][256] fp32 = 32 KiB
        const int grp = 2 * wave + hi;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          *reinterpret_cast<float4*>(red + (0 * 16 + grp) * 256 + 128 * q + 4 * s) = make_float4(ag[q][0], ag[q][1], ag[q][2], ag[q][3]);
          *reinterpret_cast<float4*>(red + (1 * 16 + grp) * 256 + 128 * q + 4 * s) = make_float4(ab[q][0], ab[q][1], ab[q][2], ab[q][3]);
        }
        __syncthreads();
        {
          const int which = tid >> 8, c = tid & 255;
          float sum = 0.f;
#pragma unroll
          for (int gI = 0; gI < 16; ++gI) sum += red[(which * 16 + gI) * 256 + c];
          atomicAdd(p.lb_ws + (int64_t)(blockIdx.x % p.lb_replicas) * 512 + which * 256 + c, sum);
        }
        return;
      }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int ml = 8 * wave + 2 * ps + hi;
      const int m = row0 + ml;
      const bool live = m < M;
      float v[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int cc = 32 * q + s;
        const f32x4 a = *reinterpret_cast<const f32x4*>(smem + ml * 1024 + 16 * (cc ^ (ml & 7)));
        const f32x4 b = *reinterpret_cast<const f32x4*>(smem + 65536 + ml * 1024 + 16 * (cc ^ (ml & 7)));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[q][r] = a[r] + b[r] + b2v[q][r];
        if (DROP && p.drop_o_p > 0.f) {
          uint32_t r16[4];
          s2t_rand_run_even32<4>(key_o, (uint32_t)m * D + (uint32_t)(128 * q + 4 * s), r16);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[q][r] = r16[r] >= th_o ? v[q][r] * inv_o : 0.f;
        }
        if constexpr (!BWD) {  // (backward: alpha went into dZ)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[q][r] *= p.alpha;
        }
        if (R) {
          float rr[4];
          if constexpr (BWD) {
            const int mc = live ? m : M - 1;
            ld4_as_f32<bf16_t>(R + (int64_t)mc * D + 128 * q + 4 * s, rr);
          } else {
            const uint2 t = rpre[ps][q];
            rr[0] = __uint_as_float(t.x << 16); rr[1] = __uint_as_float(t.x & 0xffff0000u);
            rr[2] = __uint_as_float(t.y << 16); rr[3] = __uint_as_float(t.y & 0xffff0000u);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[q][r] += rr[r];
        }
        // the block output is a bf16 tensor: a LayerNorm behind it sees the rounded values
#pragma unroll
        for (int r = 0; r < 4; ++r) v[q][r] = bf2f(f2bf(v[q][r]));
        if (Y && live) st4_from_f32<bf16_t>(Y + (int64_t)m * D + 128 * q + 4 * s, v[q]);
      }
      if (p.eln_gamma) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) sum += v[q][r];
        sum = s2t_sum32(sum);
        const float mean = sum * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = v[q][r] - mean;
            sq += d * d;
          }
        sq = s2t_sum32(sq);
        const float rstd = rsqrtf(sq * (1.0f / D) + p.ln_eps);
        const bool masked = p.eln_lens && live && s2t_row_masked32(p.eln_lens, p.eln_T, (uint32_t)m);
        if (live) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float o4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = masked ? 0.f : (v[q][r] - mean) * rstd * eg[q][r] + eb[q][r];
            st4_from_f32<bf16_t>(YL + (int64_t)m * D + 128 * q + 4 * s, o4);
          }
          if (s == 0) {
            if (p.eln_mean) p.eln_mean[m] = mean;
            if (p.eln_rstd) p.eln_rstd[m] = rstd;
          }
        }
      }
    }
  }
#if S2T_RB_DBG & 16
  if (lane == 0 && p.eln_mean && !p.eln_gamma && (blockIdx.x == 0 || blockIdx.x == 100)) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long k_t3 = __builtin_amdgcn_s_memtime(), k_r1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.eln_mean) + (blockIdx.x ? 128 : 0) + wave * 16;
    for (int i = 0; i < 9; ++i) dbg[i] = stamp[i];
    dbg[9] = k_t1 - k_t0;
    dbg[10] = k_t2 - k_t1;
    dbg[11] = k_t3 - k_t2;
    dbg[12] = k_r1 - k_r0;
  }
#endif
}

// ===============================================================================================================
// s2t_rowblock_gemm: out = epilogue( LN(x)[M,256] W[N,256]^T ) for the K = 256 projections of an encoder layer (fused
// QKV, pointwise conv 1 with GLU, attention output projection, pointwise conv 2): the same 64-row blocks, the LayerNorm
// in front folded into the prologue, the epilogue of s2t_gemm (bias, GLU, pre-activation copy, dropout, alpha, padded-row
// mask, residual) applied to whole 128-byte output rows.
//   wave (mp, q): rows 32 mp .. +32 (two MFMA column tiles), output units 16 q .. +16 of every 64-unit chunk of W
//   chunk pipeline: three 32 KiB LDS stages, the DMA runs two chunks ahead (counted vmcnt(4)), one barrier per chunk;
//   the fp32 result tile of chunk c ([64 rows][64 units], 16 KiB, double buffered) is written to LDS by the MFMA owners
//   and read back one chunk later by all 512 threads as (row, 8 consecutive columns): 16-byte residual loads and stores.
//   GLU: the chunk holds 32 value rows (weight rows 32c .. +32) and the 32 gate rows of the SAME output columns (weight
//   rows N/2 + 32c ..), so value and gate meet in one result tile.
#if S2T_RB_DBG & 64
__device__ unsigned long long s2t_rbg_dbg_buf[2 * 32];
#define RBG_STAMP(i) do { if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0) { __builtin_amdgcn_sched_barrier(0); \
    s2t_rbg_dbg_buf[(blockIdx.x ? 32 : 0) + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define RBG_STAMP(i)
#endif
constexpr int PJ_STAGES = 3;
constexpr int PJ_W = 0;                          // three weight stages
constexpr int PJ_TILE = PJ_STAGES * STAGE;       // two fp32 result tiles of 16 KiB
constexpr int PJ_BIAS = PJ_TILE + 2 * 16384;     // fp32 bias[N] (N <= PJ_MAXN)
constexpr int PJ_MAXN = 4096;
constexpr int PJ_BYTES = PJ_BIAS + PJ_MAXN * 4;  // 144 KiB

// The body of one projection on the workgroup's 64 rows.  SRC_LDS: the rows come from the bf16 image [64][512 B] a previous
// body of the same workgroup left in the result-tile region (rowblock_chain_kernel) instead of from p.x; KEEP: the packed
// output pieces of the thread (N = 256: four chunks) are also handed back in `keep` for that image.
template <bool GLU, bool DROP, bool CONV, bool SRC_LDS, bool KEEP>
__device__ __forceinline__ void rb_body(const s2t_rowblock_args& p, char* smem, const int M, uint4 (&keep)[4]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mp = wave & 1, q = wave >> 1;
  const int x = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TM;
  const int N = p.N;
  RBG_STAMP(1);
  const int nout = GLU ? N / 2 : N;
  const int ncols = GLU ? 32 : 64;            // output columns per chunk
  const int nchunks = (nout + ncols - 1) / ncols;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const i32x4 srd = make_srd(p.w, (uint32_t)N * D * 2u);  // rows beyond N read as zero (descriptor bounds)
  // (the seed is read HERE: requested in front of the loop its round trip stood between the prologue and the first chunk)
  const uint64_t key = DROP ? s2t_drop_key(p.drop_seed, p.drop_site) : 0ull;
  const uint32_t th = s2t_drop_thresh(p.drop_p);
  const float inv = s2t_drop_scale(p.drop_p);

  // DMA plan: chunk row u = 8 wave + 2 i + hi (512 B each) at LDS slot s = l & 31 holding k-chunk s ^ (u & 15);
  // weight row of chunk row u: plain 64 c + u; GLU u < 32: 32 c + u (value), u >= 32: N/2 + 32 c + u - 32 (gate)
  // (instruction i covers rows u + 2i: key (u + 2i) & 15 = (u & 15) ^ 2i, i.e. byte offset ^ 32 i, and + 1024 i through the
  // instruction offset, global and LDS alike)
  uint32_t ve;
  {
    const int hi = lane >> 5, s_ = lane & 31;
    const int u = 8 * wave + hi;
    const int wrow = GLU ? (u < 32 ? u : N / 2 + u - 32) : u;
    ve = (uint32_t)(wrow * 512 + 16 * (s_ ^ (u & 15)));
  }
  // The workgroups of an XCD (blocks b, b + 8, ...) walk the chunks of W from different starting points: they all stream the
  // same 128 - 384 KiB from that XCD's L2 at the same time, and in lockstep they would all ask the same channels
  // (chunk = output columns: the order changes no result)
#if S2T_RBG_STAGGER
  const int crot = (int)((blockIdx.x >> 3) % (unsigned)nchunks);
#else
  const int crot = 0;
#endif
  auto cmap = [&](int c) __attribute__((always_inline)) -> int { const int t = c + crot; return t >= nchunks ? t - nchunks : t; };
  auto issue = [&](int c) __attribute__((always_inline)) {
    const uint32_t base = lds0 + PJ_W + (c % PJ_STAGES) * STAGE + wave * 4096;
    const uint32_t soff = (uint32_t)cmap(c) * (uint32_t)(ncols * 512);
    dma16_off<0>(base, ve, srd, soff);
    dma16_off<1024>(base, ve ^ 32u, srd, soff);
    dma16_off<2048>(base, ve ^ 64u, srd, soff);
    dma16_off<3072>(base, ve ^ 96u, srd, soff);
  };
  issue(0);
  if (nchunks > 1) issue(1);
  {  // bias -> LDS (zeros when absent): the read-out then never waits for a global load
    float* lb = reinterpret_cast<float*>(smem + PJ_BIAS);
    for (int i = tid; i < N; i += 512) lb[i] = p.bias ? p.bias[i] : 0.f;
  }
  // read-out of a chunk's result tile: thread (row er = tid >> 3, ej = tid & 7) owns 8 consecutive units 8 ej .. 8 ej + 7
  const bf16_t* R = reinterpret_cast<const bf16_t*>(p.residual);
  const int er = tid >> 3, ej = tid & 7;
  const int em = row0 + er;
  const bool elive = em < M && (!GLU || ej < 4);
  // (the row's mask entry is read ONCE: a global load inside the loop would make the compiler wait for the DMAs in flight)
  const int emask_e = s2t_row_mask_entry(p.row_lens, p.row_T, (uint32_t)min(em, M - 1));   // (tested behind the prologue)
  auto res_load = [&](int c) __attribute__((always_inline)) -> uint4 {
    const int n0 = ncols * cmap(c) + 8 * ej;
    if (R && elive && n0 < nout) return *reinterpret_cast<const uint4*>(R + (int64_t)em * p.ldr + n0);
    return make_uint4(0, 0, 0, 0);
  };
  // Four chunks (the residual projections of the layers: N = 256): every residual piece of the thread is requested HERE, beside
  // the first weight chunks and the rows (the wait that closes the prologue covers them: no wait inside the loop), and the
  // chunk loop is unrolled (static registers).  Other widths with a residual request the piece of a read-out inside the loop
  // (slow form).
  const bool res4 = R && nchunks == 4;
  uint4 rp[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rp[c] = res4 ? res_load(c) : make_uint4(0, 0, 0, 0);

  // ---- prologue: (LayerNorm of) the 64 rows staged in the third weight stage, then this wave's B fragments --------
  if constexpr (CONV) {
    // Depthwise convolution over time (15 taps, pad 7, per-utterance zero padding) + per-channel affine (the folded
    // eval-mode BatchNorm) + activation + padded-frame mask of the 64 rows, formed here instead of in a launch of its own
    // (modules/convolution.py:100-112 in eval mode).  A wave owns 4 consecutive rows, a lane 4 channels of them: 18 input
    // rows of 8 bytes and the 60 taps of its channels per lane, all in registers.
    const bf16_t* X = reinterpret_cast<const bf16_t*>(p.x);
    char* stage = smem + PJ_W + 2 * STAGE;
    const int c4 = lane;
    float wr[60];  // this lane's taps: channel 4 c4 + c, tap k at [15 c + k] (60 contiguous floats of conv_w)
    {
      const float4* wp = reinterpret_cast<const float4*>(p.conv_w + 60 * c4);
#pragma unroll
      for (int q4 = 0; q4 < 15; ++q4) {
        const float4 t = wp[q4];
        wr[4 * q4] = t.x; wr[4 * q4 + 1] = t.y; wr[4 * q4 + 2] = t.z; wr[4 * q4 + 3] = t.w;
      }
    }
    float4 gm = *reinterpret_cast<const float4*>(p.pre_scale + 4 * c4);
    float4 bt = *reinterpret_cast<const float4*>(p.pre_shift + 4 * c4);
    if (p.bn_mean) {  // fold the running statistics: scale = gamma * rsqrt(var + eps), shift = beta - mean * scale
      const float4 mu = *reinterpret_cast<const float4*>(p.bn_mean + 4 * c4);
      const float4 va = *reinterpret_cast<const float4*>(p.bn_var + 4 * c4);
      gm.x *= rsqrtf(va.x + p.bn_eps); gm.y *= rsqrtf(va.y + p.bn_eps);
      gm.z *= rsqrtf(va.z + p.bn_eps); gm.w *= rsqrtf(va.w + p.bn_eps);
      bt.x -= mu.x * gm.x; bt.y -= mu.y * gm.y; bt.z -= mu.z * gm.z; bt.w -= mu.w * gm.w;
    }
    const int T = p.conv_T;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int rg = 8 * pass + wave;          // row group: rows 4 rg .. 4 rg + 3 of the block (wave-uniform)
      const int m0 = row0 + 4 * rg;
      const int b0 = T > 0 ? min(m0, M - 1) / T : 0;
      const int t0 = m0 - b0 * T;              // frame of the group's first row (>= T only beyond the last row)
      uint2 raw[18];
#pragma unroll
      for (int j = 0; j < 18; ++j) {
        const int mj = min(max(m0 - 7 + j, 0), M - 1);
        raw[j] = *reinterpret_cast<const uint2*>(X + (int64_t)mj * D + 4 * c4);
      }
      // rows of a neighbouring utterance (and beyond either end of the batch) read as zero padding.  ut[j]: which utterance
      // window row j belongs to — relative to the group's first row (uniform layout) or its index from the row map (packed
      // batch; -1 on rows that hold no frame: halo rows are zero in G, rows beyond the live ones are not read at all)
      int ut[18];
#pragma unroll
      for (int j = 0; j < 18; ++j) {
        if (T > 0) {
          const int tj = t0 - 7 + j;
          ut[j] = tj < 0 ? -1 : (tj >= T ? 1 : 0);
        } else {
          const int mj = m0 - 7 + j;
          const int e = (mj >= 0 && mj < M) ? p.ln_lens[mj] : -1;
          ut[j] = e >= 0 ? (e >> 16) : -1;
        }
      }
      float acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        const float4 w = make_float4(wr[k], wr[15 + k], wr[30 + k], wr[45 + k]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ui = T > 0 ? ((t0 + i >= T) ? 1 : 0) : ut[i + 7];  // wave-uniform: the group straddles an utterance boundary
          const bool ok = ut[i + k] == ui;
          const uint2 r = raw[i + k];
          const float x0 = ok ? __uint_as_float(r.x << 16) : 0.f, x1 = ok ? __uint_as_float(r.x & 0xffff0000u) : 0.f;
          const float x2 = ok ? __uint_as_float(r.y << 16) : 0.f, x3 = ok ? __uint_as_float(r.y & 0xffff0000u) : 0.f;
          acc[i][0] = fmaf(w.x, x0, acc[i][0]);
          acc[i][1] = fmaf(w.y, x1, acc[i][1]);
          acc[i][2] = fmaf(w.z, x2, acc[i][2]);
          acc[i][3] = fmaf(w.w, x3, acc[i][3]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 4 * rg + i, m = row0 + rl;
        const bool masked = m >= M || (p.ln_lens && s2t_row_masked32(p.ln_lens, p.ln_T, (uint32_t)m));
        uint2 o;
        o.x = masked ? 0u : pack2(act_apply(p.pre_act, acc[i][0] * gm.x + bt.x), act_apply(p.pre_act, acc[i][1] * gm.y + bt.y));
        o.y = masked ? 0u : pack2(act_apply(p.pre_act, acc[i][2] * gm.z + bt.z), act_apply(p.pre_act, acc[i][3] * gm.w + bt.w));
        *reinterpret_cast<uint2*>(stage + rl * 512 + 16 * ((c4 >> 1) ^ (rl & 15)) + 8 * (c4 & 1)) = o;
      }
    }
  } else {
    const bf16_t* X = reinterpret_cast<const bf16_t*>(p.x);
    char* stage = smem + PJ_W + 2 * STAGE;
    const int cch = tid & 31;
    float gm[8], bt[8];
    if (p.ln_gamma || p.pre_scale) {
      const float* pg = p.ln_gamma ? p.ln_gamma : p.pre_scale;
      const float* pb = p.ln_gamma ? p.ln_beta : p.pre_shift;
      const float4 g0 = *reinterpret_cast<const float4*>(pg + 8 * cch);
      const float4 g1 = *reinterpret_cast<const float4*>(pg + 8 * cch + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(pb + 8 * cch);
      const float4 b1v = *reinterpret_cast<const float4*>(pb + 8 * cch + 4);
      gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
      bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1v.x; bt[5] = b1v.y; bt[6] = b1v.z; bt[7] = b1v.w;
    }
    uint4 raw[4];
    int pmask[4];   // the rows' mask entries, requested together with the rows (one by one behind each pass's stores they cost a
                     // memory round trip per pass)
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int m = row0 + 16 * ps + (tid >> 5);
      const int mc = min(m, M - 1);
      if constexpr (SRC_LDS) {
        const int rl = 16 * ps + (tid >> 5);
        raw[ps] = *reinterpret_cast<const uint4*>(smem + PJ_TILE + rl * 512 + 16 * (cch ^ (rl & 15)));
      } else {
        raw[ps] = *reinterpret_cast<const uint4*>(X + (int64_t)mc * D + 8 * cch);
      }
      pmask[ps] = s2t_row_mask_entry(p.ln_lens, p.ln_T, (uint32_t)mc);
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rl = 16 * ps + (tid >> 5);
      const int m = row0 + rl;
      uint4 o = raw[ps];
      if (p.ln_gamma) {
        const uint32_t w4[4] = {o.x, o.y, o.z, o.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[2 * k] = __uint_as_float(w4[k] << 16);
          v[2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[j];
        sum = s2t_sum32(sum);
        const float mean = sum * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dd = v[j] - mean;
          sq += dd * dd;
        }
        sq = s2t_sum32(sq);
        const float rstd = rsqrtf(sq * (1.0f / D) + p.ln_eps);
        const bool masked = m < M && s2t_row_mask_test(p.ln_T, (uint32_t)m, pmask[ps]);
        uint32_t ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          ow[k] = masked ? 0u : pack2((v[2 * k] - mean) * rstd * gm[2 * k] + bt[2 * k],
                                      (v[2 * k + 1] - mean) * rstd * gm[2 * k + 1] + bt[2 * k + 1]);
        o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        if (m < M) {
          if (p.x_ln) st16_save(reinterpret_cast<bf16_t*>(p.x_ln) + (int64_t)m * D + 8 * cch, o);
          if (cch == 0) {
            if (p.ln_mean) p.ln_mean[m] = mean;
            if (p.ln_rstd) p.ln_rstd[m] = rstd;
          }
        }
      } else if (p.pre_scale) {  // per-column affine + activation (BatchNorm apply), masked rows to zero
        const uint32_t w4[4] = {o.x, o.y, o.z, o.w};
        const bool masked = m < M && s2t_row_mask_test(p.ln_T, (uint32_t)m, pmask[ps]);
        uint32_t ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v0 = __uint_as_float(w4[k] << 16) * gm[2 * k] + bt[2 * k];
          const float v1 = __uint_as_float(w4[k] & 0xffff0000u) * gm[2 * k + 1] + bt[2 * k + 1];
          ow[k] = masked ? 0u : pack2(act_apply(p.pre_act, v0), act_apply(p.pre_act, v1));
        }
        o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        if (m < M && p.x_ln) st16_save(reinterpret_cast<bf16_t*>(p.x_ln) + (int64_t)m * D + 8 * cch, o);
      }
      *reinterpret_cast<uint4*>(stage + rl * 512 + 16 * (cch ^ (rl & 15))) = o;
    }
  }
  RBG_STAMP(2);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): chunks 0 and 1, the staged rows and the compiler's own loads / stores
  RBG_STAMP(3);
  __syncthreads();
  RBG_STAMP(4);
  bf16x8 xn[2][8];
  {
    const char* stage = smem + PJ_W + 2 * STAGE;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int rl = 32 * mp + 16 * mt + x;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        xn[mt][ks] = as_frag(*reinterpret_cast<const uint4*>(stage + rl * 512 + 16 * ((4 * ks + g) ^ x)));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // fragments held before chunk 2 lands in that stage
  RBG_STAMP(5);

  const bool Z = p.preact != nullptr;
  const bool emasked = elive && s2t_row_mask_test(p.row_T, (uint32_t)em, emask_e);
  const float* lbias = reinterpret_cast<const float*>(smem + PJ_BIAS);
  // Stores go through buffer descriptors: a lane without an output (row beyond the live ones, GLU's idle half, a column tail)
  // offers an offset beyond the descriptor and the hardware drops it, so EVERY wave issues the same number of store
  // instructions per read-out — the counted waits below depend on it.  (Extents below 2 GiB: checked at the entry point.)
  constexpr uint32_t DROPPED = 0x80000000u;
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((uint32_t)p.M * (uint32_t)p.ldc * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t zsrd = __builtin_amdgcn_make_buffer_rsrc(p.preact, 0, p.preact ? (int)((uint32_t)p.M * (uint32_t)p.ldp * 2u) : 0, 0x00020000);
  auto bst8 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t off, const float (&v)[8], int aux) __attribute__((always_inline)) -> uint4 {
    const rb_u4 t = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    if (aux) __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, 0, 0);
    return make_uint4(t.x, t.y, t.z, t.w);
  };

  // result tile: fp32 [64 rows][64 units], 16-byte piece pc (4 units) of row r at r*256 + 16*(pc ^ (r & 15))
  auto read_a = [&](int c, uint4 (&af)[8]) __attribute__((always_inline)) {
    const char* lw = smem + PJ_W + (c % PJ_STAGES) * STAGE + (16 * q + x) * 512;
#pragma unroll
#if S2T_RB_DBG & 128
    for (int ks = 0; ks < 8; ++ks) af[ks] = make_uint4(ks, lane, c, 1);
#else
    for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const uint4*>(lw + 16 * ((4 * ks + g) ^ x));
#endif
  };
  auto mma_store = [&](int c, const uint4 (&af)[8]) __attribute__((always_inline)) {
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      acc[0] = mfma16(as_frag(af[ks]), xn[0][ks], acc[0]);
      acc[1] = mfma16(as_frag(af[ks]), xn[1][ks], acc[1]);
    }
    char* tile = smem + PJ_TILE + (c & 1) * 16384;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int r = 32 * mp + 16 * mt + x;
      *reinterpret_cast<f32x4*>(tile + r * 256 + 16 * ((4 * q + g) ^ x)) = acc[mt];  // r & 15 == x
    }
  };
  auto emit = [&](int c, const uint4 rres, auto&& pre_store) __attribute__((always_inline)) -> uint4 {
    const char* tile = smem + PJ_TILE + (c & 1) * 16384;
    const int r = er, j = ej, m = em;
    auto ld8t = [&](int j8, float (&v)[8]) __attribute__((always_inline)) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(tile + r * 256 + 16 * ((2 * j8) ^ (r & 15)));
      const f32x4 b = *reinterpret_cast<const f32x4*>(tile + r * 256 + 16 * ((2 * j8 + 1) ^ (r & 15)));
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    };
    auto add_bias = [&](int n0, float (&v)[8]) __attribute__((always_inline)) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(lbias + n0);
      const f32x4 b1v = *reinterpret_cast<const f32x4*>(lbias + n0 + 4);
      v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3]; v[4] += b1v[0]; v[5] += b1v[1]; v[6] += b1v[2]; v[7] += b1v[3];
    };
    const int n0 = ncols * cmap(c) + 8 * j;
    const bool live = elive && n0 < nout;
    const int n0c = live ? n0 : 0;   // (idle lanes compute on column 0's bias and drop the store)
    float v[8];
    if constexpr (GLU) {
      float gt[8];
      ld8t(j & 3, v);
      ld8t(4 + (j & 3), gt);
      add_bias(n0c, v);
      add_bias(nout + n0c, gt);
      if (Z) {
        pre_store();
        const uint32_t zo = live ? ((uint32_t)m * (uint32_t)p.ldp + (uint32_t)n0) * 2u : DROPPED;
        bst8(zsrd, zo, v, S2T_RB_SAVE_NT);
        bst8(zsrd, live ? zo + (uint32_t)nout * 2u : DROPPED, gt, S2T_RB_SAVE_NT);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= sigmoidf_(gt[e]);
    } else {
      ld8t(j, v);
      add_bias(n0c, v);
    }
    if constexpr (DROP) {
      uint32_t r16[8];
      s2t_rand_run<8>(key, (uint64_t)m * (uint64_t)nout + (uint64_t)n0, r16);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = r16[e] >= th ? v[e] * inv : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
    if (emasked) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    if (R) {
      const uint32_t w4[4] = {rres.x, rres.y, rres.z, rres.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += __uint_as_float(w4[k] << 16);
        v[2 * k + 1] += __uint_as_float(w4[k] & 0xffff0000u);
      }
    }
    if (!(GLU && Z)) pre_store();
    return bst8(osrd, live ? ((uint32_t)m * (uint32_t)p.ldc + (uint32_t)n0) * 2u : DROPPED, v, 0);
  };
  auto nothing = []() __attribute__((always_inline)) {};
  const int nst = 1 + ((GLU && Z) ? 2 : 0);   // store instructions of one read-out (every wave, see above)
  auto wait_vm = [&](int n) __attribute__((always_inline)) {   // s_waitcnt vmcnt(n) lgkmcnt(0) + barrier, n uniform
    switch (n) {
#define S2T_RB_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      S2T_RB_W(1) S2T_RB_W(2) S2T_RB_W(3) S2T_RB_W(4) S2T_RB_W(5) S2T_RB_W(6) S2T_RB_W(7) S2T_RB_W(8) S2T_RB_W(9) S2T_RB_W(10)
#undef S2T_RB_W
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
  };

  // iteration c: the fragment reads of chunk c go out first; the DMA of chunk c+2 (its stage held chunk c-1, read before the
  // last barrier); the read-out of tile (c-1) & 1 fills the time the reads take; product of chunk c into tile c & 1.
  // The closing wait needs chunk c+1 in LDS and nothing else.  In issue order the wave's outstanding operations are then
  //   DMA(c+1) | stores of read-out c-2 | DMA(c+2) | stores of read-out c-1
  // so everything behind DMA(c+1) may stay in flight: a store's acknowledgement (about a microsecond) is never waited for
  // (the order stores -> DMA -> vmcnt(4) of rounds 2 - 4 paid it once per chunk: 1.3 us per chunk for 0.3 us of work).
  auto step = [&](int c, const uint4 rres) __attribute__((always_inline)) {
    const bool more = c + 2 < nchunks;
    uint4 af[8];
    read_a(c, af);
#if S2T_RBG_ORDER == 1
#if !(S2T_RB_DBG & 1)
    if (more) issue(c + 2);
#endif
#endif
#if S2T_RBG_ORDER == 2
    // the DMA of chunk c + 2 goes out INSIDE the read-out, in front of its first store (still DMA before stores: the counted
    // wait below is unchanged) — behind the read-out's LDS reads and arithmetic instead of beside the fragment reads just
    // requested: a piece issued with LDS reads in flight parks the wave 100 - 185 clocks, one in a vector-only stretch 25 - 60
    // (MI355X_MICROARCH.md; tools/ubench/stream_mfma.hip: "burst mid-step" against "burst")
    if (c == 0 && more) issue(c + 2);
#endif
#if S2T_RBG_ORDER
#if S2T_RBG_SKEW
    // the two waves of a SIMD (w and w + 4) take the read-out and the product in opposite orders: in the same order both sit
    // in vector arithmetic at the same time and then both queue at the matrix pipe
    if (wave < 4) {
      if (c > 0) emit(c - 1, rres, nothing);
      mma_store(c, af);
    } else {
      mma_store(c, af);
      if (c > 0) emit(c - 1, rres, nothing);
    }
#else
#if !(S2T_RB_DBG & 4)
    if (c > 0) {
#if S2T_RBG_ORDER == 2
      const uint4 kp = emit(c - 1, rres, [&]() __attribute__((always_inline)) { if (more) issue(c + 2); });
#else
      const uint4 kp = emit(c - 1, rres, nothing);
#endif
      if constexpr (KEEP) keep[(c - 1) & 3] = kp;
    }
#endif
    mma_store(c, af);
#endif
    if (c + 1 < nchunks) wait_vm((c >= 2 ? nst : 0) + (more ? 4 : 0) + (c >= 1 ? nst : 0));
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    if (c > 0) emit(c - 1, rres, nothing);
    if (more) issue(c + 2);
    mma_store(c, af);
    if (more) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    RBG_STAMP(6 + (c < 20 ? c : 20));
  };
  if (res4 || KEEP) {   // (KEEP: N = 256, checked at the entry point; without a residual the pieces are zero)
#pragma unroll
    for (int c = 0; c < 4; ++c) step(c, rp[c > 0 ? c - 1 : 0]);
    const uint4 kp = emit(3, rp[3], nothing);
    if constexpr (KEEP) keep[3] = kp;
  } else if (!R) {
    for (int c = 0; c < nchunks; ++c) step(c, make_uint4(0, 0, 0, 0));
    emit(nchunks - 1, make_uint4(0, 0, 0, 0), nothing);
  } else {
    // (a wait for the residual load also waits for the OLDER DMAs of chunk c+1, which have had an iteration; the DMAs of chunk
    // c+2 are issued behind the read-out, so no compiler-inserted wait ever covers them)
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 2 < nchunks;
      const uint4 rres = c > 0 ? res_load(c - 1) : make_uint4(0, 0, 0, 0);
      uint4 af[8];
      read_a(c, af);
      if (c > 0) emit(c - 1, rres, nothing);
      if (more) issue(c + 2);
      mma_store(c, af);
      if (more) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    emit(nchunks - 1, res_load(nchunks - 1), nothing);
  }
#if S2T_RB_DBG & 64
  RBG_STAMP(28);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RBG_STAMP(29);
#endif
}

template <bool GLU, bool DROP, bool CONV = false>
__global__ __launch_bounds__(512, 2) void rowblock_gemm_kernel(const s2t_rowblock_args p) {
  __shared__ __attribute__((aligned(16))) char smem[PJ_BYTES];
  const int tid = threadIdx.x;
  RBG_STAMP(0);
  const int M = (int)s2t_live_rows(p.row_lens, p.row_T, s2t_live_rows(p.ln_lens, p.ln_T, p.M));
  if ((int)blockIdx.x * TM >= M) return;  // packed batch: this row block holds no live row
  uint4 keep[4];
  rb_body<GLU, DROP, CONV, false, false>(p, smem, M, keep);
}

// s2t_rowblock_chain: two projections of the same 64 rows in one launch — first: a plain N = 256 projection (the attention
// output projection: bias, dropout, residual); second: a projection of LayerNorm(first's output) (pointwise conv 1 + GLU,
// convolution.py:86-92).  The first body's bf16 output rows stay in the workgroup (registers -> the result-tile region as a row
// image) and the second body's prologue normalises them from there: one launch, one row load and one kernel head fewer per
// layer; every output — including the first projection's, which the layer needs as residual and for backward — is written as
// by the two separate launches, bit for bit.
struct RbChain {
  s2t_rowblock_args a, b;
};
template <bool DROP1, bool GLU2>
__global__ __launch_bounds__(512, 2) void rowblock_chain_kernel(const RbChain p) {
  __shared__ __attribute__((aligned(16))) char smem[PJ_BYTES];
  const int tid = threadIdx.x;
  RBG_STAMP(0);
  const int M = (int)s2t_live_rows(p.a.row_lens, p.a.row_T, s2t_live_rows(p.b.ln_lens, p.b.ln_T, s2t_live_rows(p.b.row_lens, p.b.row_T, p.a.M)));
  if ((int)blockIdx.x * TM >= M) return;
  uint4 keep[4];
  rb_body<false, DROP1, false, false, true>(p.a, smem, M, keep);
  __syncthreads();   // every wave is through its read-outs (the tile region) and its products
  {
    // the read-out's ownership: row er, the c-th chunk of this workgroup's walk = column block (c + crot) & 3 (the staggered
    // walk of rb_body), columns 8 ej .. + 7 of it -> 16-byte piece 8 * block + ej of the row
    const int er = tid >> 3, ej = tid & 7;
#if S2T_RBG_STAGGER
    const int crot = (int)((blockIdx.x >> 3) & 3u);
#else
    const int crot = 0;
#endif
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<uint4*>(smem + PJ_TILE + er * 512 + 16 * ((8 * ((c + crot) & 3) + ej) ^ (er & 15))) = keep[c];
  }
  __syncthreads();
  rb_body<GLU2, false, false, true, false>(p.b, smem, M, keep);
}

// ===============================================================================================================
// s2t_rowblock_dgrad: dx = LayerNorm'( dY[M,K] W ) + dres for the projections BEHIND a LayerNorm (fused QKV, K = 768;
// pointwise conv 1, K = 512): the input-gradient GEMM of the projection and the backward of the LayerNorm in front of it
// in one launch on the same 64-row blocks.  W arrives transposed (Wt [256][K], K contiguous: the forward kernel's weight
// layout for a K -> 256 projection), the reduction runs in K-blocks of 256:
//   K-block kb: the dY tile [64 rows][256] of the block lands in LDS by DMA (one block ahead), every wave takes its rows as
//   B fragments; the four 64-column chunks of Wt's K-block stream through three 32 KiB stages (DMA two steps ahead) and
//   accumulate into the wave's 4 x 2 result tiles, which stay in registers across the K-blocks;
//   epilogue: the fp32 rows meet in LDS ([64][256], the free weight stages) and go through s2t_layernorm_bwd's arithmetic
//   (mask, dgamma / dbeta partial sums into the fold workspace, residual gradient, dropped copy) as in the fused FFN backward.
#ifndef S2T_DG_A_NT
#define S2T_DG_A_NT 0  // experiment: the dY tile of s2t_rowblock_dgrad (read once) fetched non-temporally
#endif
constexpr int DG_W = 0;                 // three weight stages
constexpr int DG_A = 3 * STAGE;         // the dY tile of one K-block
constexpr int DG_BYTES = 4 * STAGE;     // 128 KiB

struct DgradK {
  const void* dy;          // [M][K] bf16
  const void* wt;          // [256][K] bf16
  int M, K;
  void* dxn;               // [M][256] bf16 out when ln_x == NULL
  const void* ln_x; const float* ln_gamma; const float* ln_mean; const float* ln_rstd;
  const int32_t* ln_lens; int ln_T;   // rows of padded frames carry no gradient into the LayerNorm (its output was masked)
  const void* dres;
  float* ws; int replicas;
  void* dx; void* dx_drop;
  float drop_p; uint32_t drop_site; const uint64_t* drop_seed;
};

__global__ __launch_bounds__(512, 2) void rowblock_dgrad_kernel(const DgradK p) {
  __shared__ __attribute__((aligned(16))) char smem[DG_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mp = wave & 1, q = wave >> 1;
  const int x = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TM;
  const int M = (int)s2t_live_rows(p.ln_lens, p.ln_T, p.M), K = p.K;
  if (row0 >= M) return;  // packed batch: this row block holds no live row
  const int KB = K / D;
  const uint32_t K2 = (uint32_t)K * 2u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const i32x4 srdw = make_srd(p.wt, (uint32_t)D * K2);
  const i32x4 srda = make_srd(p.dy, (uint32_t)M * K2);  // rows >= M read as zero

  // DMA plans: instruction i of wave w covers tile rows u = 8w + 2i + hi (512 B of one K-block each), LDS slot s = l & 31
  // holds 16-byte k-chunk s ^ (u & 15); (u + 2i) & 15 = (u & 15) ^ 2i.  Weight chunk c: Wt rows 64c + u; dY tile: rows row0 + u.
  uint32_t vw[4], va[4];
  {
    const int hi = lane >> 5, s_ = lane & 31;
    const int u0 = 8 * wave + hi;
    const uint32_t cp0 = (uint32_t)(16 * (s_ ^ (u0 & 15)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vw[i] = (uint32_t)(u0 + 2 * i) * K2 + (cp0 ^ (uint32_t)(32 * i));
      va[i] = (uint32_t)(row0 + u0 + 2 * i) * K2 + (cp0 ^ (uint32_t)(32 * i));
    }
  }
  auto issue_w = [&](int s) __attribute__((always_inline)) {  // step s = 4 kb + c
    const uint32_t base = lds0 + DG_W + (s % 3) * STAGE + wave * 4096;
    const uint32_t soff = (uint32_t)(s & 3) * (64u * K2) + (uint32_t)(s >> 2) * 512u;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(base + 1024 * i, vw[i], srdw, soff);
  };
  auto issue_a = [&](int kb) __attribute__((always_inline)) {
    const uint32_t base = lds0 + DG_A + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#if S2T_DG_A_NT
      dma16_nt(base + 1024 * i, va[i], srda, (uint32_t)kb * 512u);
#else
      dma16(base + 1024 * i, va[i], srda, (uint32_t)kb * 512u);
#endif
    }
  };
  const int S = 4 * KB;
  issue_a(0);
  issue_w(0);
  issue_w(1);
  // the LayerNorm-backward operands of this thread's epilogue rows are requested HERE, beside the first weight chunks and
  // the dY tile, and arrive under the product loop (inside the pass loop of the epilogue they were one global round trip
  // per pass; requested behind the loop they stood between the loop and the LDS exchange)
  // (so do the rows' mask entries, the LayerNorm's gamma and the dropout seed: requested behind the exchange they were a
  // round trip each, the mask entries one per pass)
  uint2 xpre[4][2], rpre[4][2];
  float mupre[4], rspre[4];
  int mpre[4] = {0, 0, 0, 0};
  float gmm[2][4];
  uint64_t key_u = 0ull;
  if (p.ln_x) {
    const bf16_t* Xp = reinterpret_cast<const bf16_t*>(p.ln_x);
    const bf16_t* Dp = reinterpret_cast<const bf16_t*>(p.dres);
    key_u = p.dx_drop ? s2t_drop_key(p.drop_seed, p.drop_site) : 0ull;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const float4 t = *reinterpret_cast<const float4*>(p.ln_gamma + 128 * qq + 4 * (lane & 31));
      gmm[qq][0] = t.x; gmm[qq][1] = t.y; gmm[qq][2] = t.z; gmm[qq][3] = t.w;
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int mr = row0 + 8 * wave + 2 * ps + (lane >> 5);
      const int mc = min(mr, M - 1);
      mpre[ps] = s2t_row_mask_entry(p.ln_lens, p.ln_T, (uint32_t)mc);
      mupre[ps] = p.ln_mean[mc];
      rspre[ps] = p.ln_rstd[mc];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        xpre[ps][qq] = *reinterpret_cast<const uint2*>(Xp + (int64_t)mc * D + 128 * qq + 4 * (lane & 31));
        rpre[ps][qq] = Dp ? *reinterpret_cast<const uint2*>(Dp + (int64_t)mc * D + 128 * qq + 4 * (lane & 31)) : make_uint2(0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  f32x4 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 xn[2][8];
  auto read_xn = [&]() __attribute__((always_inline)) {
    const char* stage = smem + DG_A;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int rl = 32 * mp + 16 * mt + x;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        xn[mt][ks] = as_frag(*reinterpret_cast<const uint4*>(stage + rl * 512 + 16 * ((4 * ks + g) ^ x)));
    }
  };
  for (int kb = 0; kb < KB; ++kb) {
    read_xn();  // (the tile of this K-block landed before the previous closing barrier)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int s = 4 * kb + c;
      uint4 af[8];
      {
        const char* lw = smem + DG_W + (s % 3) * STAGE + (16 * q + x) * 512;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const uint4*>(lw + 16 * ((4 * ks + g) ^ x));
      }
      const bool w_more = s + 2 < S;
      // the next K-block's dY tile goes out one step after this block's fragments were read by every wave (the closing
      // barrier of step c = 0 lies between), behind this step's weight group
      const bool a_more = c == 1 && kb + 1 < KB;
      if (w_more) issue_w(s + 2);
      if (a_more) issue_a(kb + 1);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        acc[c][0] = mfma16(as_frag(af[ks]), xn[0][ks], acc[c][0]);
        acc[c][1] = mfma16(as_frag(af[ks]), xn[1][ks], acc[c][1]);
      }
      // counted wait: the groups issued in this step may stay in flight
      if (w_more && a_more) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else if (w_more || a_more) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  // ---- fp32 rows into LDS: row m at m*1024 + 16*(cc ^ (m & 7)), cc = 4-column piece 16c + 4q + g
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int ml = 32 * mp + 16 * mt + x;
      *reinterpret_cast<f32x4*>(smem + ml * 1024 + 16 * ((16 * c + 4 * q + g) ^ (ml & 7))) = acc[c][mt];
    }
  __syncthreads();
  // ---- row epilogue: wave w, pass ps: rows 8w + 2ps + (lane >> 5); lane s = lane & 31 owns columns 4s.. and 128 + 4s..
  const int hi = lane >> 5, sl = lane & 31;
  if (!p.ln_x) {
    bf16_t* Y = reinterpret_cast<bf16_t*>(p.dxn);
#pragma unroll 2
    for (int ps = 0; ps < 4; ++ps) {
      const int ml = 8 * wave + 2 * ps + hi;
      const int m = row0 + ml;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int cc = 32 * qq + sl;
        const f32x4 a = *reinterpret_cast<const f32x4*>(smem + ml * 1024 + 16 * (cc ^ (ml & 7)));
        float v4[4] = {a[0], a[1], a[2], a[3]};
        if (m < M) st4_from_f32<bf16_t>(Y + (int64_t)m * D + 128 * qq + 4 * sl, v4);
      }
    }
    return;
  }
  const bf16_t* X = reinterpret_cast<const bf16_t*>(p.ln_x);
  const bf16_t* DR = reinterpret_cast<const bf16_t*>(p.dres);
  bf16_t* DX = reinterpret_cast<bf16_t*>(p.dx);
  bf16_t* DXD = reinterpret_cast<bf16_t*>(p.dx_drop);
  const uint32_t th_u = s2t_drop_thresh(p.drop_p);
  const float inv_u = s2t_drop_scale(p.drop_p);
  float ag[2][4], ab[2][4];
#pragma unroll
  for (int qq = 0; qq < 2; ++qq)
#pragma unroll
    for (int r = 0; r < 4; ++r) ag[qq][r] = ab[qq][r] = 0.f;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int ml = 8 * wave + 2 * ps + hi;
    const int m = row0 + ml;
    const bool live = m < M;
    const bool masked = !live || s2t_row_mask_test(p.ln_T, (uint32_t)m, mpre[ps]);
    const float mu = mupre[ps], rs = rspre[ps];
    float dv[2][4], xh[2][4], dg[2][4], rr[2][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int cc = 32 * qq + sl;
      const f32x4 a = *reinterpret_cast<const f32x4*>(smem + ml * 1024 + 16 * (cc ^ (ml & 7)));
      const uint2 tx = xpre[ps][qq], tr = rpre[ps][qq];
      const float xv[4] = {__uint_as_float(tx.x << 16), __uint_as_float(tx.x & 0xffff0000u),
                           __uint_as_float(tx.y << 16), __uint_as_float(tx.y & 0xffff0000u)};
      rr[qq][0] = __uint_as_float(tr.x << 16); rr[qq][1] = __uint_as_float(tr.x & 0xffff0000u);
      rr[qq][2] = __uint_as_float(tr.y << 16); rr[qq][3] = __uint_as_float(tr.y & 0xffff0000u);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (the separate kernels round the GEMM result to bf16 before the LayerNorm backward; fp32 here)
        dv[qq][r] = masked ? 0.f : a[r];
        xh[qq][r] = (xv[r] - mu) * rs;
        dg[qq][r] = dv[qq][r] * gmm[qq][r];
        s1 += dg[qq][r];
        s2 += dg[qq][r] * xh[qq][r];
        ag[qq][r] += dv[qq][r] * xh[qq][r];
        ab[qq][r] += dv[qq][r];
      }
    }
    s1 = s2t_sum32(s1);
    s2 = s2t_sum32(s2);
    s1 *= 1.0f / D;
    s2 *= 1.0f / D;
    if (live) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        float o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = bf2f(f2bf(rs * (dg[qq][r] - s1 - xh[qq][r] * s2) + rr[qq][r]));
        st4_from_f32<bf16_t>(DX + (int64_t)m * D + 128 * qq + 4 * sl, o4);
        if (DXD) {
          uint32_t r16[4];
          s2t_rand_run_even32<4>(key_u, (uint32_t)m * D + (uint32_t)(128 * qq + 4 * sl), r16);
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = r16[r] >= th_u ? o4[r] * inv_u : 0.f;
          st4_from_f32<bf16_t>(DXD + (int64_t)m * D + 128 * qq + 4 * sl, o4);
        }
      }
    }
  }
  float* red = reinterpret_cast<float*>(smem + DG_A);  // [2][16][256] fp32 = 32 KiB (the dY tile region, idle now)
  const int grp = 2 * wave + hi;
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    *reinterpret_cast<float4*>(red + (0 * 16 + grp) * 256 + 128 * qq + 4 * sl) = make_float4(ag[qq][0], ag[qq][1], ag[qq][2], ag[qq][3]);
    *reinterpret_cast<float4*>(red + (1 * 16 + grp) * 256 + 128 * qq + 4 * sl) = make_float4(ab[qq][0], ab[qq][1], ab[qq][2], ab[qq][3]);
  }
  __syncthreads();
  {
    const int which = tid >> 8, c = tid & 255;
    float sum = 0.f;
#pragma unroll
    for (int gI = 0; gI < 16; ++gI) sum += red[(which * 16 + gI) * 256 + c];
    atomicAdd(p.ws + (int64_t)(blockIdx.x % p.replicas) * 512 + which * 256 + c, sum);
  }
}

}  // namespace

// csrc/ffn_pc.hip: the 128-row producer / consumer form (mode 0 eval, 1 training forward, 2 backward)
int s2t_ffn_pc_launch(const void* kargs, int mode, int split, int drop, void* stream);

#ifndef S2T_FFN_PC_DEFAULT
#define S2T_FFN_PC_DEFAULT 7
#endif
namespace {
constexpr int PC_RB = 128;
// S2T_FFN_PC = bit mask of the flavours that run the 128-row producer / consumer kernel instead of the 64-row kernels of this
// file: 1 eval, 2 training forward, 4 backward.  Default: what measured faster on MI355X at the headline shape.
// Both switches are read from the environment ONCE (first use) and can be changed afterwards through s2t_ffn_configure
// (tests and tools switch flavours inside one process; a getenv per launch is host time on the critical path of eager steps).
struct PcConfig {
  int mask, split_force, fault, cu_budget;
};
PcConfig& pc_config() {
  static PcConfig c = [] {
    const char* e = getenv("S2T_FFN_PC");
    const char* fe = getenv("S2T_FFN_PC_SPLIT");
    const char* ce = getenv("S2T_FFN_CU_BUDGET");
    return PcConfig{e ? atoi(e) : S2T_FFN_PC_DEFAULT, fe ? atoi(fe) : 0, 0, ce ? atoi(ce) : 0};
  }();
  return c;
}
bool pc_enabled(int mode) { return (pc_config().mask >> mode) & 1; }
// training forward on the 128-row kernel only when the caller takes z tiled (its lanes cannot store row-major z efficiently)
bool pc_train(const s2t_ffn_args* a) { return pc_enabled(1) && (a->z_tiled_ok || !a->z); }
int pc_num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256;
    return v;
  }();
  return n;
}
// workgroups per 128-row block by the row count alone: 8 (each on an eighth of the hidden units) for few rows (the decoder's
// 3 904: 31 blocks), 4 or 2 while the whole grid is then resident at once — the partners wait for each other — else 1.
// "At once" is measured against the CU BUDGET: the device's compute units, or what s2t_ffn_cu_budget / S2T_FFN_CU_BUDGET leaves
// of them when other kernels hold some (the data-parallel wrapper takes off what its all-reduce kernels occupy beside backward).
int pc_split_rows(int M) {
  const int P = (M + PC_RB - 1) / PC_RB;
  const int b = pc_config().cu_budget;
  const int cus = b > 0 ? b : pc_num_cus();
  return 8 * P <= cus ? 8 : (4 * P <= cus ? 4 : (2 * P <= cus ? 2 : 1));
}
// bytes of the fp32 partial-row slabs of the exchange: split 2: the 64 rows the partner finishes; split 8: all 128 rows
int64_t pc_slab_bytes(int M, int split) {
  const int64_t P = (M + PC_RB - 1) / PC_RB;
  return split == 2 ? P * 2 * 64 * 256 * 4 : (split > 2 ? P * split * PC_RB * 256 * 4 : 0);
}
// ... and what the caller gave: the exchange workspace, hidden units in whole chunks per part
// mode 0 eval, 1 training forward, 2 backward.  S2T_FFN_PC_TRAIN_MAX_SPLIT caps the parts per block of the training flavours
// (default 8: no cap).  Round 6 measured the cap at four: at the decoder's 3 904 rows (31 blocks) eight parts exchange 8 x 128 KiB of
// fp32 partial rows per block — 3.2 x the launch's algorithmic bytes — and the ISOLATED kernels are as fast or faster with four
// (tools/ffn_split_probe.py, MI355X: training forward 30.5 / 29.8 us, backward 29.4 / 27.8 us for 8 / 4 parts; eval 24.6 / 25.3),
// but the STEP is slower with four: 10.52 / 10.53 / 10.50 ms against 10.46 / 10.44 / 10.43 with eight (tools/ab_env.sh, three
// alternations on one box) — so eight stays.
int pc_train_max_split() {
  static const int v = [] {
    const char* e = getenv("S2T_FFN_PC_TRAIN_MAX_SPLIT");
    const int n = e ? atoi(e) : 8;
    return (n == 1 || n == 2 || n == 4 || n == 8) ? n : 8;
  }();
  return v;
}
int pc_split(int M, int F, const void* ws, int64_t ws_bytes, int mode) {
  // S2T_FFN_PC_SPLIT / s2t_ffn_configure pins (1) or caps (2, 4) the workgroups per block: the forms add the hidden units'
  // products in different orders, so results that must match bit for bit across DIFFERENT row counts need one of them pinned
  const int force = pc_config().split_force;
  if (force == 1 || !ws || ws_bytes < s2t_ffn_pair_ws_bytes(M) || ((uintptr_t)ws % 16) || (F % 128)) return 1;
  int split = pc_split_rows(M);
  while (split > 2 && ((force >= 2 && split > force) || F % (split * 128))) split >>= 1;  // whole pairs of chunks per part
  if (mode != 0 && force == 0) while (split > 2 && split > pc_train_max_split()) split >>= 1;
  return split;
}
// Layout of the exchange workspace: PC_FLAG_BYTES of flags first (split x split words per block for 4 or 8 parts, 2 per block
// for 2 — at most 8 x 256 words; word S2T_PC_ERR_WORD counts exchange time-outs, the same place for every shape), the slabs behind.  The flags sit at the same place for every
// row count and are zero between launches (their readers clear them), so ONE workspace serves every shape a caller runs.
constexpr int64_t PC_FLAG_BYTES = 16384;
uint32_t* pc_flags(void* ws) { return reinterpret_cast<uint32_t*>(ws); }
float* pc_slabs(void* ws) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + PC_FLAG_BYTES); }
}  // namespace

static_assert((S2T_PC_ERR_WORD + 1) * 4 <= PC_FLAG_BYTES && S2T_PC_ERR_WORD >= 2560, "error word inside the flag area, above every flag");

extern "C" int s2t_ffn_configure(int pc_mask, int split_force, int fault) {
  PcConfig& c = pc_config();
  if (pc_mask >= 0) c.mask = pc_mask;
  if (split_force >= 0) c.split_force = split_force;
  if (fault >= 0) c.fault = fault;
  return (c.mask & 7) | (c.split_force << 4) | (c.fault << 12);
}
extern "C" int s2t_ffn_cu_budget(int cus) {
  PcConfig& c = pc_config();
  if (cus >= 0) c.cu_budget = cus;  // 0: back to the device's count
  return c.cu_budget > 0 ? c.cu_budget : pc_num_cus();
}
extern "C" int64_t s2t_ffn_exchange_error_offset(void) { return (int64_t)S2T_PC_ERR_WORD * 4; }
extern "C" int64_t s2t_ffn_exchange_flag_bytes(void) { return PC_FLAG_BYTES; }

extern "C" int64_t s2t_ffn_pair_ws_bytes(int32_t M) {
  const int split = pc_split_rows(M);  // (a launch may use fewer parts than this: it then needs less)
  const int64_t two = pc_slab_bytes(M, 2), many = pc_slab_bytes(M, split);
  return PC_FLAG_BYTES + (many > two ? many : two);
}

extern "C" int s2t_ffn_fused_fwd(const s2t_ffn_args* a, void* stream) {
  if (!a || !a->x || !a->w1 || !a->b1 || !a->w2 || !a->b2) return S2T_ERR_ARG;
  if (a->M <= 0 || a->F <= 0) return S2T_ERR_ARG;
  if (a->d != D) return S2T_ERR_UNSUPPORTED;
  if (a->F % FC || a->F > MAXF) return S2T_ERR_UNSUPPORTED;
  if ((int64_t)(a->M + TM) * a->F >= ((int64_t)1 << 32)) return S2T_ERR_UNSUPPORTED;  // 32-bit element indices (dropout, saves)
  if (a->act != S2T_ACT_NONE && a->act != S2T_ACT_RELU && a->act != S2T_ACT_SWISH) return S2T_ERR_ARG;
  if (!a->y && !a->y_ln) return S2T_ERR_ARG;
  if ((a->eln_gamma != nullptr) != (a->y_ln != nullptr) || (a->eln_gamma && !a->eln_beta)) return S2T_ERR_ARG;
  if ((a->ln_gamma != nullptr) != (a->ln_beta != nullptr)) return S2T_ERR_ARG;
  if (s2t_rows_arg_bad(a->eln_lens, a->eln_T)) return S2T_ERR_ARG;
  if (a->drop_h_p < 0.f || a->drop_h_p >= 1.f || a->drop_o_p < 0.f || a->drop_o_p >= 1.f) return S2T_ERR_ARG;
  if ((a->drop_h_p > 0.f || a->drop_o_p > 0.f) && !a->drop_seed) return S2T_ERR_ARG;
  const void* ptrs[] = {a->x, a->w1, a->w2, a->residual, a->y, a->y_ln, a->x_ln, a->z, a->h, a->b1, a->b2,
                        a->ln_gamma, a->ln_beta, a->eln_gamma, a->eln_beta};
  for (const void* q : ptrs)
    if (q && ((uintptr_t)q % 16)) return S2T_ERR_ALIGN;
  const bool train = a->z || a->h || a->x_ln || a->ln_mean || a->ln_rstd;
  const dim3 grid((a->M + TM - 1) / TM), block(512);
  hipStream_t s = (hipStream_t)stream;
  const bool drop = a->drop_h_p > 0.f || a->drop_o_p > 0.f;
  FfnK k = {};
  static_cast<s2t_ffn_args&>(k) = *a;
  if (train ? pc_train(a) : pc_enabled(0)) {
    const int split = pc_split(a->M, a->F, a->pair_ws, a->pair_ws_bytes, train ? 1 : 0);
    k.z_tiled = 1;
    if (split >= 2) {
      k.xws = pc_slabs(a->pair_ws);
      k.xflags = pc_flags(a->pair_ws);
      k.xfault = pc_config().fault;
    }
    return s2t_ffn_pc_launch(&k, train ? 1 : 0, split, drop ? 1 : 0, stream);
  }
#define GO(T, A, DR) hipLaunchKernelGGL((ffn_fused_fwd_kernel<T, A, DR>), grid, block, 0, s, k)
#define GO_A(T, DR)                                     \
  do {                                                  \
    if (a->act == S2T_ACT_RELU) GO(T, S2T_ACT_RELU, DR); \
    else if (a->act == S2T_ACT_SWISH) GO(T, S2T_ACT_SWISH, DR); \
    else GO(T, S2T_ACT_NONE, DR);                       \
  } while (0)
  if (train) {
    if (drop) GO_A(1, true); else GO_A(1, false);
  } else {
    if (drop) GO_A(0, true); else GO_A(0, false);
  }
#undef GO_A
#undef GO
  return S2T_LAUNCH_CHECK();
}

namespace {
int ffn_describe(int mode, int act, bool drop, int M, int F, const void* ws, int64_t ws_bytes, char* buf, int n, bool force_pc = false) {
  if (!buf || n < 96) return S2T_ERR_ARG;
  if (pc_enabled(mode) || (mode == 2 && force_pc)) snprintf(buf, n, "ffn_pc_kernel<%d, %d, %s, %d>", mode, act, drop ? "true" : "false", pc_split(M, F, ws, ws_bytes, mode));
  else snprintf(buf, n, "ffn_fused_fwd_kernel<%d, %d, %s>", mode, act, drop ? "true" : "false");
  return S2T_OK;
}
}  // namespace

extern "C" int s2t_ffn_z_tiled(const s2t_ffn_args* a) {
  if (!a) return 0;
  const bool train = a->z || a->h || a->x_ln || a->ln_mean || a->ln_rstd;
  return train && a->z && pc_train(a) ? 1 : 0;
}

extern "C" int64_t s2t_ffn_z_elems(int32_t M, int32_t F) { return (int64_t)((M + PC_RB - 1) / PC_RB) * PC_RB * F; }

extern "C" int s2t_ffn_fused_describe(const s2t_ffn_args* a, char* buf, int32_t n) {
  if (!a) return S2T_ERR_ARG;
  const bool train = a->z || a->h || a->x_ln || a->ln_mean || a->ln_rstd;
  if (train && !pc_train(a)) {
    if (!buf || n < 96) return S2T_ERR_ARG;
    snprintf(buf, n, "ffn_fused_fwd_kernel<1, %d, %s>", a->act, (a->drop_h_p > 0.f || a->drop_o_p > 0.f) ? "true" : "false");
    return S2T_OK;
  }
  return ffn_describe(train ? 1 : 0, a->act, a->drop_h_p > 0.f || a->drop_o_p > 0.f, a->M, a->F, a->pair_ws, a->pair_ws_bytes, buf, n);
}

extern "C" int s2t_ffn_fused_bwd_describe(const s2t_ffn_bwd_args* b, char* buf, int32_t n) {
  if (!b) return S2T_ERR_ARG;
  return ffn_describe(2, b->act, b->drop_h_p > 0.f, b->M, b->F, b->pair_ws, b->pair_ws_bytes, buf, n, b->z_tiled != 0);
}

extern "C" int s2t_ffn_fused_bwd(const s2t_ffn_bwd_args* b, void* stream) {
  if (!b || !b->dy || !b->w2t || !b->w1t || !b->z || !b->dz) return S2T_ERR_ARG;
  if (b->end_y) {
    if (!b->end_gamma || !b->end_mean || !b->end_rstd || !b->end_ws || b->end_replicas <= 0 || !b->dres_out) return S2T_ERR_ARG;
    if (s2t_rows_arg_bad(b->end_lens, b->end_T)) return S2T_ERR_ARG;
    if (b->dy_out && (b->drop_o_p <= 0.f || b->drop_o_p >= 1.f || !b->drop_seed)) return S2T_ERR_ARG;
    if (((uintptr_t)b->end_y % 16) || ((uintptr_t)b->dres_out % 16) || ((uintptr_t)b->dy_out % 16)) return S2T_ERR_ALIGN;
  }
  if (b->ln_x) {
    if (!b->ln_gamma || !b->ln_mean || !b->ln_rstd || !b->ln_ws || b->ln_replicas <= 0 || !b->dx) return S2T_ERR_ARG;
    if (b->dx_drop && (b->up_drop_p <= 0.f || b->up_drop_p >= 1.f || !b->drop_seed)) return S2T_ERR_ARG;
  } else if (!b->dxn) return S2T_ERR_ARG;
  if (b->M <= 0 || b->F <= 0) return S2T_ERR_ARG;
  if (b->d != D) return S2T_ERR_UNSUPPORTED;
  if (b->F % FC || b->F > MAXF) return S2T_ERR_UNSUPPORTED;
  if ((int64_t)b->M * b->F * 2 >= ((int64_t)1 << 32)) return S2T_ERR_UNSUPPORTED;  // 32-bit byte offsets into Z / dZ
  if (b->act != S2T_ACT_NONE && b->act != S2T_ACT_RELU && b->act != S2T_ACT_SWISH) return S2T_ERR_ARG;
  if (b->drop_h_p < 0.f || b->drop_h_p >= 1.f || (b->drop_h_p > 0.f && !b->drop_seed)) return S2T_ERR_ARG;
  const void* ptrs[] = {b->dy, b->w2t, b->w1t, b->z, b->dz, b->dxn, b->ln_x, b->ln_gamma, b->dres, b->dx, b->dx_drop};
  for (const void* q : ptrs)
    if (q && ((uintptr_t)q % 16)) return S2T_ERR_ALIGN;
  // the kernel is the forward one with the transposed weights in the places of W1 / W2 (see MODE 2 there)
  FfnK a = {};
  a.lb_x = b->ln_x;
  a.lb_gamma = b->ln_gamma;
  a.lb_mean = b->ln_mean;
  a.lb_rstd = b->ln_rstd;
  a.lb_dres = b->dres;
  a.lb_ws = b->ln_ws;
  a.lb_replicas = b->ln_replicas;
  a.lb_dx = b->dx;
  a.lb_dx_drop = b->dx_drop;
  a.lb_drop_p = b->up_drop_p;
  a.lb_drop_site = b->up_drop_site;
  a.pl_y = b->end_y;
  a.pl_gamma = b->end_gamma;
  a.pl_mean = b->end_mean;
  a.pl_rstd = b->end_rstd;
  a.pl_lens = b->end_lens;
  a.pl_T = b->end_T;
  a.pl_ws = b->end_ws;
  a.pl_replicas = b->end_replicas;
  a.pl_dres = b->dres_out;
  a.pl_dy = b->dy_out;
  a.drop_o_p = b->dy_out ? b->drop_o_p : 0.f;
  a.drop_o_site = b->drop_o_site;
  a.x = b->dy;
  a.d = D;
  a.w1 = b->w2t;
  a.w2 = b->w1t;
  a.y = b->dxn;
  a.z = const_cast<void*>(b->z);
  a.h = b->dz;
  a.M = b->M;
  a.F = b->F;
  a.act = b->act;
  a.alpha = b->alpha;
  a.drop_h_p = b->drop_h_p;
  a.drop_h_site = b->drop_h_site;
  a.drop_seed = b->drop_seed;
  const dim3 grid((a.M + TM - 1) / TM), block(512);
  hipStream_t s = (hipStream_t)stream;
  const bool drop = a.drop_h_p > 0.f;
  a.z_tiled = b->z_tiled;
  if (pc_enabled(2) || b->z_tiled) {
    const int split = pc_split(b->M, b->F, b->pair_ws, b->pair_ws_bytes, 2);
    if (split >= 2) {
      a.xws = pc_slabs(b->pair_ws);
      a.xflags = pc_flags(b->pair_ws);
      a.xfault = pc_config().fault;
    }
    return s2t_ffn_pc_launch(&a, 2, split, drop ? 1 : 0, stream);
  }
#define GO(A, DR) hipLaunchKernelGGL((ffn_fused_fwd_kernel<2, A, DR>), grid, block, 0, s, a)
  if (a.act == S2T_ACT_RELU) { if (drop) GO(S2T_ACT_RELU, true); else GO(S2T_ACT_RELU, false); }
  else if (a.act == S2T_ACT_SWISH) { if (drop) GO(S2T_ACT_SWISH, true); else GO(S2T_ACT_SWISH, false); }
  else { if (drop) GO(S2T_ACT_NONE, true); else GO(S2T_ACT_NONE, false); }
#undef GO
  return S2T_LAUNCH_CHECK();
}

#if S2T_RB_DBG & 64
extern "C" int s2t_rbg_dbg_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(s2t_rbg_dbg_buf), sizeof(unsigned long long) * 64);
}
#endif
static int rowblock_args_check(const s2t_rowblock_args* a) {
  if (!a || !a->x || !a->w || !a->out) return S2T_ERR_ARG;
  if (a->M <= 0 || a->N <= 0) return S2T_ERR_ARG;
  if (a->d != D) return S2T_ERR_UNSUPPORTED;
  const bool glu = a->act == S2T_ACT_GLU;
  if (a->act != S2T_ACT_NONE && !glu) return S2T_ERR_UNSUPPORTED;
  const int nout = glu ? a->N / 2 : a->N;
  if (glu ? (a->N % 64) : (a->N % 8)) return S2T_ERR_UNSUPPORTED;  // GLU: whole 32-column chunks of value and gate rows
  if (a->N > PJ_MAXN) return S2T_ERR_UNSUPPORTED;                  // the bias row staged in LDS holds PJ_MAXN floats
  if (a->preact && !glu) return S2T_ERR_UNSUPPORTED;
  if ((a->ln_gamma != nullptr) != (a->ln_beta != nullptr)) return S2T_ERR_ARG;
  if ((a->pre_scale != nullptr) != (a->pre_shift != nullptr) || (a->pre_scale && a->ln_gamma)) return S2T_ERR_ARG;
  if ((a->ln_mean || a->ln_rstd) && !a->ln_gamma) return S2T_ERR_ARG;
  if ((a->x_ln || a->ln_lens) && !a->ln_gamma && !a->pre_scale) return S2T_ERR_ARG;
  if (a->pre_scale && (((uintptr_t)a->pre_scale % 16) || ((uintptr_t)a->pre_shift % 16))) return S2T_ERR_ALIGN;
  if (s2t_rows_arg_bad(a->ln_lens, a->ln_T) || s2t_rows_arg_bad(a->row_lens, a->row_T)) return S2T_ERR_ARG;
  if (a->drop_p < 0.f || a->drop_p >= 1.f || (a->drop_p > 0.f && !a->drop_seed)) return S2T_ERR_ARG;
  if (a->ldc < nout || a->ldc % 8 || (a->residual && (a->ldr < nout || a->ldr % 8)) || (a->preact && (a->ldp < a->N || a->ldp % 8)))
    return S2T_ERR_ALIGN;
  // (the kernel's stores address out / preact through 32-bit byte offsets of buffer descriptors)
  if ((int64_t)a->M * a->ldc * 2 >= ((int64_t)1 << 31) || (a->preact && (int64_t)a->M * a->ldp * 2 >= ((int64_t)1 << 31))) return S2T_ERR_UNSUPPORTED;
  if (a->conv_w) {
    const bool conv_packed = a->conv_T == S2T_ROWS_PACKED && a->ln_lens && a->ln_T == S2T_ROWS_PACKED;  // utterances from the row map
    if (!a->pre_scale || glu || a->drop_p > 0.f || a->x_ln || (!conv_packed && a->conv_T < 18) || (a->ln_lens && a->ln_T != a->conv_T)) return S2T_ERR_ARG;
    if (!conv_packed && a->M % a->conv_T) return S2T_ERR_ARG;
    if ((uintptr_t)a->conv_w % 16) return S2T_ERR_ALIGN;
    if (a->bn_mean && (!a->bn_var || ((uintptr_t)a->bn_mean % 16) || ((uintptr_t)a->bn_var % 16))) return S2T_ERR_ARG;
  }
  const void* ptrs[] = {a->x, a->w, a->out, a->residual, a->preact, a->x_ln, a->bias, a->ln_gamma, a->ln_beta};
  for (const void* q : ptrs)
    if (q && ((uintptr_t)q % 16)) return S2T_ERR_ALIGN;
  return S2T_OK;
}

extern "C" int s2t_rowblock_gemm(const s2t_rowblock_args* a, void* stream) {
  const int chk = rowblock_args_check(a);
  if (chk != S2T_OK) return chk;
  const bool glu = a->act == S2T_ACT_GLU;
  const dim3 grid((a->M + TM - 1) / TM), block(512);
  hipStream_t s = (hipStream_t)stream;
  const bool drop = a->drop_p > 0.f;
  if (glu) {
    if (drop) hipLaunchKernelGGL((rowblock_gemm_kernel<true, true>), grid, block, 0, s, *a);
    else hipLaunchKernelGGL((rowblock_gemm_kernel<true, false>), grid, block, 0, s, *a);
  } else if (a->conv_w) {
    hipLaunchKernelGGL((rowblock_gemm_kernel<false, false, true>), grid, block, 0, s, *a);
  } else {
    if (drop) hipLaunchKernelGGL((rowblock_gemm_kernel<false, true>), grid, block, 0, s, *a);
    else hipLaunchKernelGGL((rowblock_gemm_kernel<false, false>), grid, block, 0, s, *a);
  }
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_rowblock_chain(const s2t_rowblock_args* first, const s2t_rowblock_args* second, void* stream) {
  if (!first || !second) return S2T_ERR_ARG;
  int e = rowblock_args_check(first);
  if (e != S2T_OK) return e;
  s2t_rowblock_args b = *second;
  if (!b.x) b.x = first->out;   // (the second projection reads the first one's rows from the chip; x is not dereferenced)
  e = rowblock_args_check(&b);
  if (e != S2T_OK) return e;
  // first: plain 256-column projection of rows given as they are; second: behind a LayerNorm of the first one's output
  if (first->act != S2T_ACT_NONE || first->N != D || first->ln_gamma || first->pre_scale || first->conv_w || first->preact ||
      first->ldc != D)
    return S2T_ERR_UNSUPPORTED;
  if (!b.ln_gamma || b.pre_scale || b.conv_w || b.x != first->out || b.M != first->M || b.drop_p > 0.f || b.residual)
    return S2T_ERR_UNSUPPORTED;
  RbChain k;
  k.a = *first;
  k.b = b;
  const dim3 grid((first->M + TM - 1) / TM), block(512);
  hipStream_t s = (hipStream_t)stream;
  const bool drop = first->drop_p > 0.f, glu = b.act == S2T_ACT_GLU;
  if (drop) {
    if (glu) hipLaunchKernelGGL((rowblock_chain_kernel<true, true>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((rowblock_chain_kernel<true, false>), grid, block, 0, s, k);
  } else {
    if (glu) hipLaunchKernelGGL((rowblock_chain_kernel<false, true>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((rowblock_chain_kernel<false, false>), grid, block, 0, s, k);
  }
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_rowblock_dgrad(const s2t_rowblock_dgrad_args* a, void* stream) {
  if (!a || !a->dy || !a->wt) return S2T_ERR_ARG;
  if (a->M <= 0 || a->K <= 0) return S2T_ERR_ARG;
  if (a->d != D || a->K % D || a->K > 8 * D) return S2T_ERR_UNSUPPORTED;
  if ((int64_t)(a->M + TM) * a->K * 2 >= ((int64_t)1 << 32)) return S2T_ERR_UNSUPPORTED;  // 32-bit byte offsets into dY
  if (a->ln_x) {
    if (!a->ln_gamma || !a->ln_mean || !a->ln_rstd || !a->ln_ws || a->ln_replicas <= 0 || !a->dx) return S2T_ERR_ARG;
    if (s2t_rows_arg_bad(a->ln_lens, a->ln_T)) return S2T_ERR_ARG;
    if (a->dx_drop && (a->up_drop_p <= 0.f || a->up_drop_p >= 1.f || !a->drop_seed)) return S2T_ERR_ARG;
  } else if (!a->dxn) return S2T_ERR_ARG;
  const void* ptrs[] = {a->dy, a->wt, a->dxn, a->ln_x, a->ln_gamma, a->dres, a->dx, a->dx_drop};
  for (const void* q : ptrs)
    if (q && ((uintptr_t)q % 16)) return S2T_ERR_ALIGN;
  DgradK k = {};
  k.dy = a->dy; k.wt = a->wt; k.M = a->M; k.K = a->K; k.dxn = a->dxn;
  k.ln_x = a->ln_x; k.ln_gamma = a->ln_gamma; k.ln_mean = a->ln_mean; k.ln_rstd = a->ln_rstd;
  k.ln_lens = a->ln_lens; k.ln_T = a->ln_T; k.dres = a->dres; k.ws = a->ln_ws; k.replicas = a->ln_replicas;
  k.dx = a->dx; k.dx_drop = a->dx_drop; k.drop_p = a->up_drop_p; k.drop_site = a->up_drop_site; k.drop_seed = a->drop_seed;
  hipLaunchKernelGGL(rowblock_dgrad_kernel, dim3((a->M + TM - 1) / TM), dim3(512), 0, (hipStream_t)stream, k);
  return S2T_LAUNCH_CHECK();
}
