// Fused feed-forward block for d = 256 on gfx950, producer / consumer form (round 3): a workgroup owns 128 complete rows
// of the [B*T, 256] activation and, when the grid would otherwise leave most of the chip idle, HALF of the hidden
// dimension — the two workgroups of a pair exchange fp32 partial rows through L2 at the end and each finishes 64 rows.
//
//   s2t_ffn_fused_fwd   out = residual + alpha * drop_o( W2 drop_h(act(W1 LN(x) + b1)) + b2 ) [-> LayerNorm]
//                       modules/s2t_transformer_layer.py:55-66 (FeedForwardModule), :258-265, :311-317 (macaron / final
//                       half-step residuals), :318-320 (final_norm); modules/layer_norm.py:30-35.
//   s2t_ffn_fused_bwd   the input gradient of the same block on the same schedule (MODE 2, transposed weights).
//
// Why this shape (measured on the 64-row kernel of rowblock.hip, tools/rb_stamps.py): a 64-row workgroup streams all 2 MiB of
// W1 | W2 through the CU's 64 B/clk vector-memory path — 1024 cycles per 64-unit chunk, exactly the time of the chunk's MFMAs —
// and every wave both issued those LDS-DMAs and computed, so a wave parked on a DMA issue (~68 cycles a piece) issued no
// MFMA: 2280 cycles per chunk for 1024 cycles of matrix work.  Here
//   * 128 rows per workgroup halve the weight bytes per flop (the ratio of a 256 x 256 GEMM tile);
//   * the eight waves have ROLES: waves 0-3 (producers) hold the normalised rows as B fragments and compute
//       G1: H^T[f][m] = sum_k W1[f][k] Xn[m][k]  (v_mfma_f32_32x32x16_bf16, A = W1 rows from LDS), then bias / activation /
//       dropout / bf16 pack on their own accumulators (lane-local: a lane owns ONE activation row);
//     waves 4-7 (consumers) issue every LDS-DMA and compute
//       G2: Y^T[n][m] += sum_f W2[n][f] H^T[f][m] with the packed H registers of their SIMD partner as B fragments,
//     handed over lane for lane through a 4 KiB LDS mailbox (double buffered by chunk parity, one barrier per chunk, G2 lags
//     G1 by a chunk).  Producer w and consumer w + 4 sit on one SIMD: the VALU-heavy half of one overlaps the MFMAs of the
//     other, and a wave parked on a DMA issue blocks nobody's matrix work;
//   * 32x32x16 MFMAs hold the vector issue port for 8 of 32 cycles (16x16x32: 8 of 16).
//   The G1 tile's MFMA row rho is fed with hidden unit pi(rho) (bits 2 and 3 swapped), so that accumulator registers
//   8s .. 8s+7 of lane (m, h) are the eight CONSECUTIVE units 16s + 8h .. +7: packed pairwise they are the B fragment of
//   G2's k-step s as they stand, and one 16-byte piece of a saved row.
// LDS (160 KiB): two 32 KiB stages of W1 chunks, two of W2 chunks (XOR-swizzled through the DMA's per-lane source address),
// 32 KiB of mailboxes; the prologue stages the 128 normalised rows in the W2 stages, the epilogue the fp32 result rows in
// all four.
#include "common.h"
#include "lds_dma.h"
#include "ffn_args.h"

#ifndef S2T_PC_SGB
#define S2T_PC_SGB 8  // VALU / transcendental instructions dealt behind each MFMA of the producers' second tile (0: scheduler's choice)
#endif
#ifndef S2T_PC_SIDE_AT
#define S2T_PC_SIDE_AT 5  // producers' MFMA group behind which the next chunk's bias / pre-activation loads are issued
#endif
#ifndef S2T_PC_PIECE_ORDER
#define S2T_PC_PIECE_ORDER 0  // consumers' LDS-DMA pieces: 0 in front of each pair of MFMAs, 1 behind them with the LDS queue drained
#endif
#ifndef S2T_PC_BURST_AT
#define S2T_PC_BURST_AT -1  // experiment (round 6, tools/ubench/stream_mfma.hip): the consumers' 16 LDS-DMA pieces of a chunk as ONE burst in
#endif                      // front of MFMA pair BURST_AT (0 .. 12) instead of dealt over the first thirteen pairs (-1)
#ifndef S2T_PC_SAVE_AUX
#define S2T_PC_SAVE_AUX 2  // cache policy of the training saves (z, h / dZ): 0 default, 2 non-temporal.  They are written once and
#endif                     // read by the backward pass a whole model later; left to the default policy they push the weights (which every
                           // workgroup re-reads) out of L2 / the Infinity Cache: the isolated training forward 64.6 -> 58.6 us at
                           // 12 950 rows (tools/ffn_probe_cold.py); inside the step the kernels AROUND the FFN gain: bench step
                           // 11.65 -> 11.59 ms (tools/ab_bench.sh, four alternations, same box)
#ifndef S2T_PC_ZLOAD_AUX
#define S2T_PC_ZLOAD_AUX 0  // cache policy of the backward's pre-activation loads (read once)
#endif
#ifndef S2T_PC_PRIO
#define S2T_PC_PRIO 0  // experiment: s_setprio of the producers (bits 0-1) and the consumers (bits 2-3) for the chunk loop
#endif
#ifndef S2T_PC_DBG
#define S2T_PC_DBG 0  // experiment switches: 1 no DMA in the loop, 2 no MFMAs, 4 no E1 arithmetic, 16 stamps
#endif

namespace {

// The LayerNorm-backward reductions of MODE 2 keep the LDS-crossbar form (__shfl_xor): with the DPP form (csrc/common.h) hipcc
// spills five registers INSIDE the consumers' chunk loop (scratch loads there make the compiler wait for the DMA stream:
// 72 -> 106 us per launch); S2T_PC_BWD_DPP=1 selects the DPP form for a newer compiler to try
#ifndef S2T_PC_XCHG_SC1
#define S2T_PC_XCHG_SC1 1  // the partial-row exchange reads the partners' slabs with sc1 loads (L2-served, past this CU's L1) instead of
#endif                     // an agent-scope acquire (buffer_inv sc1 + the wait for it: ~1.7 us per launch) followed by plain loads: MI355X_MICROARCH.md,
                           // "Valid forms", first row of the sc1 table — every byte stored sc1 and drained, ONE lane's sc1 flag store behind the
                           // workgroup barrier, an sc1 poll, the other waves behind a barrier the polling wave joins, 16-byte sc1 loads, one
                           // workgroup per CU (160 KiB of LDS), torch's (hipMalloc) memory.  0: round 5's acquire + plain loads.
#ifndef S2T_PC_BWD_DPP
#define S2T_PC_BWD_DPP 0
#endif
__device__ __forceinline__ float pc_shfl_sum32(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#if S2T_PC_BWD_DPP
#define PC_BWD_SUM32(v) s2t_sum32(v)
#else
#define PC_BWD_SUM32(v) pc_shfl_sum32(v)
#endif

constexpr int D = 256;
constexpr int RB = 128;             // rows per workgroup
constexpr int FC = 64;              // hidden units per chunk
constexpr int STAGE = 32768;
constexpr int L_W1 = 0;
constexpr int L_W2 = 2 * STAGE;
constexpr int L_MB = 4 * STAGE;     // mailboxes: [parity][row group][k-step s][64 lanes] 16 B
constexpr int LDS_BYTES = 5 * STAGE;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
#if S2T_PC_DBG & 2
  asm volatile("" :: "v"(a), "v"(b));
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ uint32_t pk2(float a, float b) { return bf16pack(a, b); }
__device__ __forceinline__ float lo16(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// s_waitcnt immediate of gfx9: vmcnt = bits 3:0 | 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

template <int MODE, int ACT, bool DROP, int SPLIT>
__global__ __launch_bounds__(512, 2) void ffn_pc_kernel(const FfnK p) {
  constexpr bool TRAIN = MODE == 1, BWD = MODE == 2;
  constexpr bool SAVE = TRAIN || BWD;   // the consumers store one [rows][F] tensor (h / dZ) chunk by chunk
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int wi = wave & 3;                 // row group: rows 32 wi .. +32 of the block
  const int r32 = lane & 31, hh = lane >> 5;
  // packed batch: the live row count comes from the row map given as the mask of the trailing LayerNorm (forward: eln_lens,
  // backward: pl_lens, S2T_ROWS_BOUND when there is no mask); the block -> (row block, part) dealing follows the HOST's M
  const int Mh = p.M, F = p.F;
  const int M = (int)s2t_live_rows(BWD ? p.pl_lens : p.eln_lens, BWD ? p.pl_T : p.eln_T, Mh);
  // ---- which rows, which half of the hidden units.  Pairs are dealt so that the two workgroups of a pair are blocks
  // b and b + 8 (one XCD under round-robin placement: speed only, the exchange is placement independent).
  int pair, half;
  {
    const int b = blockIdx.x;
    if constexpr (SPLIT == 1) {
      pair = b;
      half = 0;
    } else {
      // groups of 8 row blocks x SPLIT parts: the parts of a row block are blocks b, b + 8, b + 16, ...
      const int P = (Mh + RB - 1) / RB;
      constexpr int G = 8 * SPLIT;
      const int full = (P >> 3) * G;
      if (b < full) {
        pair = (b / G) * 8 + (b & 7);
        half = (b >> 3) % SPLIT;
      } else {
        pair = ((P >> 3) << 3) + (b - full) / SPLIT;
        half = (b - full) % SPLIT;
      }
    }
  }
  const int row0 = pair * RB;
  if (row0 >= M) return;  // no live row in this block: every part of it leaves (no flag is raised, none is awaited)
  const int FH = F / SPLIT;                // hidden units of this workgroup
  const int fbase = half * FH;
  const int nchunks = FH / FC;
  constexpr int KR = RB / SPLIT;           // rows this workgroup finishes
  const int krow0 = KR * half;             // (`half`: this workgroup's part, 0 .. SPLIT-1)
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
#if S2T_PC_DBG & 16
  unsigned long long stp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long seg[5] = {0, 0, 0, 0, 0}, lt[5] = {0, 0, 0, 0, 0};
#define LSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); lt[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LACC() do { for (int i_ = 0; i_ < 4; ++i_) seg[i_] += lt[i_ + 1] - lt[i_]; } while (0)
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); stp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(i)
#define LSTAMP(i)
#define LACC()
#endif
  PSTAMP(0);
  // forward flavours: the dropout seed is read ONCE, here (a scalar load in front of every store; requested where a key is first
  // used it was a vector load whose round trip stood in front of the producers' chunk loop and in front of the row epilogue).
  // (Not in the backward flavour: there the two scalar registers it holds across the kernel tip hipcc into spilling inside the
  // producers' loop — tools/spill_sites.sh.)
  uint64_t seed_top = 0ull;
  if constexpr (!BWD) seed_top = (DROP && p.drop_seed) ? *p.drop_seed : 0ull;

  const i32x4 srd1 = make_srd(p.w1, (uint32_t)F * D * 2u);
  const i32x4 srd2 = make_srd(p.w2, (uint32_t)F * D * 2u);

  // ---- DMA plans (consumer wave wi; 16 one-KiB pieces per chunk each).
  // W1 chunk image [64 units][512 B]: 16-byte k-piece q of unit row u at u*512 + 16*(q ^ (u & 15)).  Piece i of wave wi covers
  // rows 16 wi + 2 i + (lane >> 5): key 2i + hi -> source offset (row 16 wi + hi, key hi) ^ 32 i, + 1024 i.
  // W2 chunk image [256 outputs][128 B]: f-piece q of output row n at n*128 + 16*(q ^ ((n >> 1) & 7)).  Piece i covers rows
  // 64 wi + 8 i + (lane >> 3): key (lane >> 4) for even i, ^ 4 for odd i.
  uint32_t v1, v2e, v2o;
  {
    const int r = 16 * wi + hh;
    v1 = (uint32_t)(r * 512 + 16 * (r32 ^ hh));
    const int r2 = 64 * wi + (lane >> 3);
    v2e = (uint32_t)r2 * (uint32_t)(F * 2) + (uint32_t)(16 * ((lane & 7) ^ (lane >> 4)));
    v2o = v2e ^ 64u;
  }
  const uint32_t w2step = (uint32_t)(8 * F * 2);
  auto dma_w1 = [&](int c, int i0, int i1) __attribute__((always_inline)) {
#if !(S2T_PC_DBG & 1)
    const uint32_t base = lds0 + L_W1 + (c & 1) * STAGE + wi * 8192;
    const uint32_t soff = (uint32_t)(fbase + c * FC) * 512u;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const uint32_t vo = v1 ^ (uint32_t)(32 * i);
      const uint32_t b = base + (i >> 2) * 4096, so = soff + (i >> 2) * 4096;
      if ((i & 3) == 0) dma16_off<0>(b, vo, srd1, so);
      else if ((i & 3) == 1) dma16_off<1024>(b, vo, srd1, so);
      else if ((i & 3) == 2) dma16_off<2048>(b, vo, srd1, so);
      else dma16_off<3072>(b, vo, srd1, so);
    }
#endif
  };
  auto dma_w2 = [&](int c, int i0, int i1) __attribute__((always_inline)) {
#if !(S2T_PC_DBG & 1)
    const uint32_t base = lds0 + L_W2 + (c & 1) * STAGE + wi * 8192;
    const uint32_t soff = (uint32_t)(fbase + c * FC) * 2u;
#pragma unroll
    for (int i = i0; i < i1; ++i) dma16(base + i * 1024, (i & 1) ? v2o : v2e, srd2, soff + (uint32_t)i * w2step);
#endif
  };
  if (!producer) dma_w1(0, 0, 8);

  // ---- prologue: (LayerNorm of) the 128 rows, one 16-byte piece per thread and pass (32 lanes per row, 16 rows per pass),
  // staged in the W2 stages: piece q of row r at r*512 + 16*(q ^ (r & 15)).
  {
    const bf16_t* X = reinterpret_cast<const bf16_t*>(p.x);
    char* stage = smem + L_W2;
    const int cch = tid & 31;
    float gm[8], bt[8];
    if (p.ln_gamma) {
      const float4 g0 = *reinterpret_cast<const float4*>(p.ln_gamma + 8 * cch);
      const float4 g1 = *reinterpret_cast<const float4*>(p.ln_gamma + 8 * cch + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.ln_beta + 8 * cch);
      const float4 b1v = *reinterpret_cast<const float4*>(p.ln_beta + 8 * cch + 4);
      gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
      bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1v.x; bt[5] = b1v.y; bt[6] = b1v.z; bt[7] = b1v.w;
    }
    uint4 raw[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int mc = min(row0 + 16 * ps + (tid >> 5), M - 1);
      raw[ps] = *reinterpret_cast<const uint4*>(X + (int64_t)mc * D + 8 * cch);
    }
    if constexpr (BWD) {
      if (p.pl_y) {
        // ---- backward of the trailing LayerNorm on the way in (s2t_layernorm_bwd's arithmetic, 32 lanes per row):
        //   d = masked ? 0 : dout;  dres = rstd * (d*gamma - mean(d*gamma) - xhat * mean(d*gamma*xhat));  staged tile =
        //   dropout_o(dres).  Both workgroups of a pair need all 128 rows; each writes and sums only the rows it finishes.
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
        // The operands of FOUR passes are requested together before the first of them computes: requested pass by pass (behind
        // the previous pass's stores, which the compiler will not move a load across) each pass paid a memory round trip AND
        // the acknowledgement of those stores, eight times one after the other at the head of every backward launch.  (All
        // eight at once cost registers that hipcc took from the producers' chunk loop: spills inside it.)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
        uint4 yrs[4];
        float mus[4], rss[4];
        int mke[4];
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
          const int m = row0 + 16 * (4 * ph + pq) + (tid >> 5);
          const int mc = min(m, M - 1);
          yrs[pq] = *reinterpret_cast<const uint4*>(Yp + (int64_t)mc * D + 8 * cch);
          mus[pq] = p.pl_mean[mc];
          rss[pq] = p.pl_rstd[mc];
          mke[pq] = s2t_row_mask_entry(p.pl_lens, p.pl_T, (uint32_t)mc);   // (the CLAMPED row: a tail block reads no entry beyond the map)
        }
        bool mks[4];   // (lane masks in scalar registers)
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
          const int m = row0 + 16 * (4 * ph + pq) + (tid >> 5);
          mks[pq] = m >= M || s2t_row_mask_test(p.pl_T, (uint32_t)m, mke[pq]);
        }
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
          const int ps = 4 * ph + pq;
          const int rl = 16 * ps + (tid >> 5);
          const int m = row0 + rl;
          const uint4 yr = yrs[pq];
          const float mu = mus[pq], rs = rss[pq];
          const bool live = m < M;
          const bool own = live && (SPLIT == 1 || rl / KR == half);
          const bool masked = mks[pq];
          const uint32_t dw4[4] = {raw[ps].x, raw[ps].y, raw[ps].z, raw[ps].w};
          const uint32_t yw4[4] = {yr.x, yr.y, yr.z, yr.w};
          float dgv[8], xh[8];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int j = 2 * q + e;
              const float dv = masked ? 0.f : (e ? hi16(dw4[q]) : lo16(dw4[q]));
              const float yv = e ? hi16(yw4[q]) : lo16(yw4[q]);
              xh[j] = (yv - mu) * rs;
              dgv[j] = dv * pg[j];
              s1 += dgv[j];
              s2 += dgv[j] * xh[j];
              if (own) {
                ag[j] += dv * xh[j];
                ab[j] += dv;
              }
            }
          }
          s1 = PC_BWD_SUM32(s1);
          s2 = PC_BWD_SUM32(s2);
          s1 *= 1.0f / D;
          s2 *= 1.0f / D;
          uint32_t rw[4], yw[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            rw[q] = pk2(rs * (dgv[2 * q] - s1 - xh[2 * q] * s2), rs * (dgv[2 * q + 1] - s1 - xh[2 * q + 1] * s2));
          uint4 o = make_uint4(rw[0], rw[1], rw[2], rw[3]);
          if (own) *reinterpret_cast<uint4*>(DR + (int64_t)m * D + 8 * cch) = o;
          if (DY) {  // the dropped image of the STORED bf16 dres, as s2t_dropout would make it
            uint32_t r16[8];
            s2t_rand_run_even32<8>(key_e, (uint32_t)m * D + (uint32_t)(8 * cch), r16);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              yw[q] = pk2(r16[2 * q] >= th_e ? lo16(rw[q]) * inv_e : 0.f, r16[2 * q + 1] >= th_e ? hi16(rw[q]) * inv_e : 0.f);
            o = make_uint4(yw[0], yw[1], yw[2], yw[3]);
            if (own) *reinterpret_cast<uint4*>(DY + (int64_t)m * D + 8 * cch) = o;
          }
          raw[ps] = o;
        }
        }
        // [2][16][256] fp32 = 32 KiB in the mailbox region (idle until the first chunk)
        float* red = reinterpret_cast<float*>(smem + L_MB);
        const int grp = tid >> 5;
        *reinterpret_cast<float4*>(red + (0 * 16 + grp) * 256 + 8 * cch) = make_float4(ag[0], ag[1], ag[2], ag[3]);
        *reinterpret_cast<float4*>(red + (0 * 16 + grp) * 256 + 8 * cch + 4) = make_float4(ag[4], ag[5], ag[6], ag[7]);
        *reinterpret_cast<float4*>(red + (1 * 16 + grp) * 256 + 8 * cch) = make_float4(ab[0], ab[1], ab[2], ab[3]);
        *reinterpret_cast<float4*>(red + (1 * 16 + grp) * 256 + 8 * cch + 4) = make_float4(ab[4], ab[5], ab[6], ab[7]);
      }
    }
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int rl = 16 * ps + (tid >> 5);
      const int m = row0 + rl;
      uint4 o = raw[ps];
      if (p.ln_gamma) {
        const uint32_t w4[4] = {o.x, o.y, o.z, o.w};
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[2 * q] = lo16(w4[q]);
          v[2 * q + 1] = hi16(w4[q]);
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
          ow[q] = pk2((v[2 * q] - mean) * rstd * gm[2 * q] + bt[2 * q], (v[2 * q + 1] - mean) * rstd * gm[2 * q + 1] + bt[2 * q + 1]);
        o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        if constexpr (TRAIN) {
          if (m < M && (SPLIT == 1 || rl / KR == half)) {
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
  // chunk 0 of W1, the staged tile and the compiler's own prologue loads / stores have landed
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
  __syncthreads();
  if constexpr (BWD) {
    if (p.pl_y) {  // column sums of the trailing LayerNorm's parameter gradients: 512 atomics into one replica
      const float* red = reinterpret_cast<const float*>(smem + L_MB);
      const int which = tid >> 8, c = tid & 255;
      float sum = 0.f;
#pragma unroll
      for (int gI = 0; gI < 16; ++gI) sum += red[(which * 16 + gI) * 256 + c];
      atomicAdd(p.pl_ws + (int64_t)(blockIdx.x % p.pl_replicas) * 512 + which * 256 + c, sum);
    }
  }
  // producers: this wave's 32 rows as B fragments of the 32x32x16 product: lane (m, h) owns k = 16 ks + 8 h .. +8 of row m
  bf16x8 xn[16];
  if (producer) {
    const char* stage = smem + L_W2;
    const int rl = 32 * wi + r32;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      xn[ks] = frag(*reinterpret_cast<const uint4*>(stage + rl * 512 + 16 * ((2 * ks + hh) ^ (rl & 15))));
  }
  // every producer holds its fragments (and the mailbox region's partial sums were read) before chunk 0 of W2 overwrites the
  // staging buffer / the first mailbox is written
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const int mrow = row0 + 32 * wi + r32;   // the activation row of this lane (both roles)
  char* const mbox = smem + L_MB;
  // mailbox cell of (chunk parity, row group wi, k-step s, lane (m, h)): the XOR spreads the eight 16-byte pieces of one row
  // over eight bank groups for the row-major read-back of the saves
  auto mcell = [&](int c, int s, int h, int m) __attribute__((always_inline)) -> char* {
    return mbox + (c & 1) * 16384 + wi * 4096 + s * 1024 + h * 512 + ((m ^ (2 * (2 * s + h))) & 31) * 16;
  };
  // z: row-major [M][F], or (z_tiled) the tiled layout of include/s2t_hip.h over row blocks of 128 (rows padded)
  const uint32_t zrows = p.z_tiled ? (uint32_t)((M + RB - 1) / RB) * RB : (uint32_t)M;
  const __amdgpu_buffer_rsrc_t zsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.z), 0, p.z ? (int)(zrows * (uint32_t)F * 2u) : 0, 0x00020000);
  // byte offset of this lane's 16-byte piece of k-step s of (global) chunk cg in the tiled layout
  const uint32_t ztl = (uint32_t)(((pair * (F / FC)) * 4 + wi) * 4) * 1024u + (uint32_t)(hh * 32 + r32) * 16u;
  auto ztile_off = [&](int cg, int s) __attribute__((always_inline)) -> uint32_t {
    return ztl + (uint32_t)cg * 16384u + (uint32_t)s * 1024u;
  };
  const __amdgpu_buffer_rsrc_t hsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.h), 0, p.h ? (int)((uint32_t)M * (uint32_t)F * 2u) : 0, 0x00020000);

  f32x16 yacc[8];
  PSTAMP(1);
#if S2T_PC_PRIO
  // wave priority: the producers' chain (G1 -> E1 -> mailbox) is the critical path of a chunk, the consumers have slack
  if (producer) __builtin_amdgcn_s_setprio(S2T_PC_PRIO & 3);
  else __builtin_amdgcn_s_setprio((S2T_PC_PRIO >> 2) & 3);
#endif
  if (producer) {
    // =========================================== producers: G1 + E1 ===================================================
    const uint64_t key_h = DROP ? (BWD ? s2t_drop_key(p.drop_seed, p.drop_h_site) : s2t_drop_key_of(seed_top, p.drop_h_site)) : 0ull;
    const uint32_t th_h = s2t_drop_thresh(p.drop_h_p);
    const float inv_h = s2t_drop_scale(p.drop_h_p);
    // MFMA row rho = r32 of a 32-unit tile is fed with unit pi(rho) (bits 2 and 3 swapped)
    const int pr = (r32 & 19) | ((r32 & 4) << 1) | ((r32 & 8) >> 1);
    const int akey = pr & 15;
    const uint32_t abase = (uint32_t)(pr * 512);
    // what rides in from memory per chunk: forward the 32 bias values of the lane (initial accumulators), backward its 32
    // pre-activation values (4 x 16 bytes)
    struct Side {
      f32x16 b[2];
      uint4 z[4];
    };
    const uint32_t rowF = (uint32_t)mrow * (uint32_t)F;
    auto side_load = [&](int c) __attribute__((always_inline)) -> Side {
      Side sd;
      if constexpr (BWD) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint32_t off = p.z_tiled ? ztile_off(fbase / FC + c, s) : (rowF + (uint32_t)(fbase + c * FC + 16 * s + 8 * hh)) * 2u;
          const u32x4s t = __builtin_amdgcn_raw_buffer_load_b128(zsrd, off, 0, S2T_PC_ZLOAD_AUX);
          sd.z[s] = make_uint4(t.x, t.y, t.z, t.w);
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        sd.b[0] = zero;
        sd.b[1] = zero;
      } else {
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
          const float* bp = p.b1 + fbase + c * FC + 32 * ut + 8 * hh;
          const float4 q0 = *reinterpret_cast<const float4*>(bp), q1 = *reinterpret_cast<const float4*>(bp + 4);
          const float4 q2 = *reinterpret_cast<const float4*>(bp + 16), q3 = *reinterpret_cast<const float4*>(bp + 20);
          sd.b[ut] = (f32x16){q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        }
      }
      return sd;
    };
    // E1 on four of the lane's values: accumulator registers 8 sl + 4 jh .. +3 of `acc` = units 16 s + 8 h + 4 jh .. +3 of
    // the chunk (s = 2 ut + sl); returns the two packed bf16 pairs (the second product's operand) and, training, the packed
    // pre-activation
    auto e1_quad = [&](int c, int s, int jh, const f32x16& acc, const Side& sd, uint32_t (&hp)[2], uint32_t (&zp)[2]) __attribute__((always_inline)) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[8 * (s & 1) + 4 * jh + j];
      if constexpr (TRAIN) {
        zp[0] = pk2(v[0], v[1]);
        zp[1] = pk2(v[2], v[3]);
      }
#if !(S2T_PC_DBG & 4)
      if constexpr (BWD) {
        const uint32_t z0 = jh ? sd.z[s].z : sd.z[s].x, z1 = jh ? sd.z[s].w : sd.z[s].y;
        v[0] *= act_grad(ACT, lo16(z0));
        v[1] *= act_grad(ACT, hi16(z0));
        v[2] *= act_grad(ACT, lo16(z1));
        v[3] *= act_grad(ACT, hi16(z1));
      } else if constexpr (ACT == S2T_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if constexpr (ACT == S2T_ACT_SWISH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * sigmoidf_(v[j]);
      }
      if constexpr (DROP) {
        uint32_t r16[4];
        s2t_rand_run_even32<4>(key_h, rowF + (uint32_t)(fbase + c * FC + 16 * s + 8 * hh + 4 * jh), r16);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = r16[j] >= th_h ? v[j] * inv_h : 0.f;
      }
      if constexpr (BWD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
      }
#endif
      hp[0] = pk2(v[0], v[1]);
      hp[1] = pk2(v[2], v[3]);
    };
    // the finished eight values of k-step s: mailbox cell (the consumers' B fragment) and, training, 16 bytes of z
    auto e1_out = [&](int c, int s, const uint32_t (&hp)[4], const uint32_t (&zp)[4]) __attribute__((always_inline)) {
      if constexpr (TRAIN) {
        // (tiled: the wave's 64 pieces of this k-step are 1 KiB contiguous; row-major they would be 32-byte pieces of 32 rows)
        __builtin_amdgcn_raw_buffer_store_b128((u32x4s){zp[0], zp[1], zp[2], zp[3]}, zsrd, ztile_off(fbase / FC + c, s), 0, S2T_PC_SAVE_AUX);
      }
      *reinterpret_cast<uint4*>(mcell(c, s, hh, r32)) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
    };
    // the A fragments of four k-steps (group g) of unit tile ut
    auto rd_a = [&](const char* l1, int ut, int g, uint4 (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const uint4*>(l1 + ut * (32 * 512) + 16 * ((2 * (4 * g + j) + hh) ^ akey));
    };
    Side cur = side_load(0);
    for (int c = 0; c < nchunks; ++c) {
      Side nxt;
      const char* l1 = smem + L_W1 + (c & 1) * STAGE + abase;
      f32x16 acc0 = cur.b[0], acc1 = cur.b[1];
      // Software pipeline over eight groups of four MFMAs (tile 0: groups 0-3, tile 1: groups 4-7): the fragment reads of
      // group t + 1 are issued BEFORE the MFMAs of group t (hipcc otherwise sinks every read to just in front of its
      // MFMA: one exposed LDS round trip per pair), and tile 0's bias / activation / dropout / pack runs between the
      // MFMAs of tile 1.  sched_barrier pins the groups; inside a group of tile 1 the VALU work is dealt behind the MFMAs.
      uint4 R[2][4];
      uint32_t hp[4], zp[4];
      LSTAMP(0);
      rd_a(l1, 0, 0, R[0]);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < 7) rd_a(l1, (t + 1) >> 2, (t + 1) & 3, R[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (t < 4) acc0 = mfma32(frag(R[t & 1][j]), xn[4 * t + j], acc0);
          else acc1 = mfma32(frag(R[t & 1][j]), xn[4 * (t - 4) + j], acc1);
        }
        if (t >= 4) {  // quarter t - 4 of tile 0's E1: k-step s = (t - 4) >> 1, values 4 ((t - 4) & 1) .. +3
          const int q = t - 4;
          uint32_t h2[2], z2[2];
          e1_quad(c, q >> 1, q & 1, acc0, cur, h2, z2);
          hp[2 * (q & 1)] = h2[0]; hp[2 * (q & 1) + 1] = h2[1];
          zp[2 * (q & 1)] = z2[0]; zp[2 * (q & 1) + 1] = z2[1];
          if (q & 1) e1_out(c, q >> 1, hp, zp);
#if S2T_PC_SGB
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x402, S2T_PC_SGB, 0);  // VALU | TRANS behind it
          }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t == 3) LSTAMP(1);
        // what rides in for the next chunk is fetched late: it then shares no registers with the first tile's fragments
        if (t == S2T_PC_SIDE_AT) {
          nxt = side_load(c + 1 < nchunks ? c + 1 : c);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      LSTAMP(2);
      // tile 1's E1 has no MFMAs of its own left to hide behind: it runs beside the consumer's on this SIMD
#pragma unroll
      for (int s = 2; s < 4; ++s) {
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
          uint32_t h2[2], z2[2];
          e1_quad(c, s, jh, acc1, cur, h2, z2);
          hp[2 * jh] = h2[0]; hp[2 * jh + 1] = h2[1];
          zp[2 * jh] = z2[0]; zp[2 * jh + 1] = z2[1];
        }
        e1_out(c, s, hp, zp);
      }
      cur = nxt;
      LSTAMP(3);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      LSTAMP(4);
      LACC();
    }
  } else {
    // =========================================== consumers: DMA + G2 + saves ===========================================
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      yacc[nt] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // G2 of chunk c with the DMA pieces of the running iteration spread over its MFMA groups, and (training / backward) the
    // saved tensor of the same chunk row-major out of the mailbox: lane (row 8 q + (l >> 3), piece l & 7) -> a wave
    // instruction covers 8 whole 128-byte lines.  ALWAYS four store instructions behind the last DMA piece (the closing wait
    // counts on it); the descriptor drops rows >= M and everything when the tensor was not asked for.
    // The fragment reads of output tile nt + 1 go out BEFORE the MFMAs of tile nt (see the producers' loop).
    auto g2 = [&](int c, auto&& piece) __attribute__((always_inline)) {
      // Every per-lane address of the chunk is formed HERE from the lane id, behind an asm the compiler cannot hoist: carried
      // across the loop (four mailbox cells, four fragment offsets, the row offset, the save cells) they cost the registers
      // hipcc then spilled INSIDE the loop — a scratch reload and its vmcnt(0), which also waits for the DMA stream and the
      // saves in flight, at the head of every chunk.  ~30 vector instructions per chunk instead.
      uint32_t ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int r32 = (int)(ln & 31u), hh = (int)(ln >> 5), lane = (int)ln;
      const int g2key = (r32 >> 1) & 7;
      bf16x8 hb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) hb[s] = frag(*reinterpret_cast<const uint4*>(mcell(c, s, hh, r32)));
      const char* l2 = smem + L_W2 + (c & 1) * STAGE + r32 * 128;
      auto rd_w = [&](int nt, uint4 (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const uint4*>(l2 + nt * 4096 + 16 * ((2 * s + hh) ^ g2key));
      };
      uint4 R[2][4];
      uint4 sv[2];
      const int pc = lane & 7;
      rd_w(0, R[0]);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        if (nt < 7) rd_w(nt + 1, R[(nt + 1) & 1]);
        if constexpr (SAVE) {
          if (nt >= 6) {
#pragma unroll
            for (int k = 0; k < 2; ++k) sv[k] = *reinterpret_cast<const uint4*>(mcell(c, pc >> 1, pc & 1, 8 * (2 * (nt - 6) + k) + (lane >> 3)));
          }
        }
#if S2T_PC_PIECE_ORDER == 0
        piece(nt, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) yacc[nt] = mfma32(frag(R[nt & 1][s]), hb[s], yacc[nt]);
        __builtin_amdgcn_sched_barrier(0);
        piece(nt, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 2; s < 4; ++s) yacc[nt] = mfma32(frag(R[nt & 1][s]), hb[s], yacc[nt]);
        __builtin_amdgcn_sched_barrier(0);
#else
        // the pieces go out once this wave's fragment reads have landed (a piece issued beside LDS reads in flight parks the
        // wave for 100+ cycles, one issued behind MFMAs with the LDS queue drained about half of that)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) yacc[nt] = mfma32(frag(R[nt & 1][s]), hb[s], yacc[nt]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        piece(nt, 0);
        __builtin_amdgcn_sched_barrier(0);
        yacc[nt] = mfma32(frag(R[nt & 1][2]), hb[2], yacc[nt]);
        __builtin_amdgcn_sched_barrier(0);
        piece(nt, 1);
        __builtin_amdgcn_sched_barrier(0);
        yacc[nt] = mfma32(frag(R[nt & 1][3]), hb[3], yacc[nt]);
        __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr (SAVE) {
          if (nt >= 6) {  // behind the last DMA piece of the iteration (the pieces end with group 6's first pair)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int rr = 8 * (2 * (nt - 6) + k) + (lane >> 3);
              const uint32_t off = ((uint32_t)(row0 + 32 * wi + rr) * (uint32_t)F + (uint32_t)(fbase + c * FC + 8 * pc)) * 2u;
              __builtin_amdgcn_raw_buffer_store_b128((u32x4s){sv[k].x, sv[k].y, sv[k].z, sv[k].w}, hsrd, off, 0, S2T_PC_SAVE_AUX);
            }
          }
        }
      }
    };
    // L2 warm-up two chunks ahead by the first 16 workgroups (one per XCD and half under round-robin placement): the first
    // touch of a chunk per XCD is an HBM miss that a DMA issued one chunk ahead cannot hide.  256 threads, one 4-byte load per
    // 128-byte line of W1's chunk, another per line of W2's; the value is never used and the load may stay in flight.
    uint32_t pf_sink = 0;
    const int ct = tid - 256;
    const uint32_t pfo1 = (uint32_t)ct * 128u, pfo2 = (uint32_t)ct * (uint32_t)(F * 2);
    const char* pfb1 = reinterpret_cast<const char*>(p.w1) + (size_t)fbase * 512;
    const char* pfb2 = reinterpret_cast<const char*>(p.w2) + (size_t)fbase * 2;
    const bool pf_wg = blockIdx.x < 16;
    // iteration 0: nothing to multiply yet
    {
      if (nchunks > 1) dma_w1(1, 0, 8);
      dma_w2(0, 0, 8);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    for (int c = 1; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      LSTAMP(0);
      g2(c - 1, [&](int nt, int half) __attribute__((always_inline)) {
        // 16 pieces dealt over the first thirteen pairs of MFMAs, the first three pairs carrying two (a burst of pieces
        // queues behind the other waves' at the memory pipe and parks this wave, MFMAs unissued, for over a hundred cycles a
        // piece); the saves of groups 6 and 7 follow the last piece
        const int k = 2 * nt + half;       // pair 0 .. 15
#if S2T_PC_BURST_AT >= 0
        const int p0 = k == S2T_PC_BURST_AT ? 0 : 16, p1 = 16;
#else
        const int p0 = k < 3 ? 2 * k : k + 3, p1 = k < 3 ? 2 * k + 2 : k + 4;   // pieces p0 .. p1-1 of 16
#endif
#pragma unroll
        for (int q = p0; q < p1 && q < 16; ++q) {
          if (q < 8) { if (more) dma_w1(c + 1, q, q + 1); }
          else dma_w2(c, q - 8, q - 7);
        }
        if (nt == 4 && half == 0) LSTAMP(1);
      });
      LSTAMP(2);
      const bool pf = pf_wg && c + 2 < nchunks;
      if (pf) {
        const char* a1 = pfb1 + (size_t)(c + 2) * (FC * 512);   // wave-uniform bases, 32-bit lane offsets
        const char* a2 = pfb2 + (size_t)(c + 2) * (FC * 2);
        asm volatile("global_load_dword %0, %1, %3\n\tglobal_load_dword %0, %2, %4"
                     : "+v"(pf_sink) : "v"(pfo1), "v"(pfo2), "s"(a1), "s"(a2) : "memory");
        __builtin_amdgcn_s_waitcnt(waitcnt_imm((SAVE ? 4 : 0) + 2, 0));
      } else {
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(SAVE ? 4 : 0, 0));
      }
      LSTAMP(3);
      asm volatile("s_barrier" ::: "memory");
      LSTAMP(4);
      LACC();
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) :: "memory");
    g2(nchunks - 1, [&](int, int) __attribute__((always_inline)) {});
  }
  PSTAMP(2);
  __syncthreads();
  PSTAMP(3);

  // ---- the fp32 result rows meet in LDS: [128 rows][256] fp32, 16-byte piece cc of row r at r*1024 + 16*(cc ^ (r & 7)).
  // Consumer lane (m, h), tile nt, register group q holds columns 32 nt + 8 q + 4 h .. +3: piece 8 nt + 2 q + h.
  if (!producer) {
    const int rl = 32 * wi + r32;
    char* rowp = smem + rl * 1024;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(rowp + 16 * ((8 * nt + 2 * q + hh) ^ (rl & 7))) =
            (f32x4){yacc[nt][4 * q], yacc[nt][4 * q + 1], yacc[nt][4 * q + 2], yacc[nt][4 * q + 3]};
  }
  // Row epilogue geometry: wave w finishes rows krow0 + (KR/8) w + 2 ps + hi; lane s = lane & 31 owns columns 4s..4s+3 and
  // 128+4s..+3 of its row.  The residual / LayerNorm-backward operand rows travel during the exchange.
  constexpr int NPS = KR / 16;
  const int hi = lane >> 5, s = lane & 31;
  uint2 rpre[NPS][2];
  uint2 xpre[NPS][2];
  float mupre[NPS], rspre[NPS];
  if constexpr (BWD) {
    if (p.lb_x) {
      const bf16_t* Xp = reinterpret_cast<const bf16_t*>(p.lb_x);
      const bf16_t* Dp = reinterpret_cast<const bf16_t*>(p.lb_dres);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int mc = min(row0 + krow0 + (KR / 8) * wave + 2 * ps + hi, M - 1);
        mupre[ps] = p.lb_mean[mc];
        rspre[ps] = p.lb_rstd[mc];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          xpre[ps][q] = *reinterpret_cast<const uint2*>(Xp + (int64_t)mc * D + 128 * q + 4 * s);
          rpre[ps][q] = Dp ? *reinterpret_cast<const uint2*>(Dp + (int64_t)mc * D + 128 * q + 4 * s) : make_uint2(0, 0);
        }
      }
    } else if (p.residual) {
      const bf16_t* Rp = reinterpret_cast<const bf16_t*>(p.residual);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int mc = min(row0 + krow0 + (KR / 8) * wave + 2 * ps + hi, M - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) rpre[ps][q] = *reinterpret_cast<const uint2*>(Rp + (int64_t)mc * D + 128 * q + 4 * s);
      }
    }
  } else {
    if (p.residual) {
      const bf16_t* Rp = reinterpret_cast<const bf16_t*>(p.residual);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int mc = min(row0 + krow0 + (KR / 8) * wave + 2 * ps + hi, M - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) rpre[ps][q] = *reinterpret_cast<const uint2*>(Rp + (int64_t)mc * D + 128 * q + 4 * s);
      }
    }
  }
  // So do the column parameters of the row epilogue, the mask entries of its rows and (backward) the dropout seed: requested
  // behind the exchange each was a memory round trip on the tail of the launch — the mask entries one per pass, each behind that
  // pass's stores and their acknowledgement.
  float b2v[2][4], eg[2][4], eb[2][4], gmm[2][4];
  int emk[NPS];
  uint64_t seed_tail = 0ull;
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
    if constexpr (BWD) {
      if (p.lb_x) {
        const float4 g4 = *reinterpret_cast<const float4*>(p.lb_gamma + 128 * q + 4 * s);
        gmm[q][0] = g4.x; gmm[q][1] = g4.y; gmm[q][2] = g4.z; gmm[q][3] = g4.w;
      }
    }
  }
  if constexpr (BWD) seed_tail = p.drop_seed ? *p.drop_seed : 0ull;
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps)
    emk[ps] = p.eln_gamma ? s2t_row_mask_entry(p.eln_lens, p.eln_T, (uint32_t)min(row0 + krow0 + (KR / 8) * wave + 2 * ps + hi, M - 1)) : 0;
  __syncthreads();
  PSTAMP(4);

  // ---- pair exchange (SPLIT == 2): the 64 rows the OTHER workgroup finishes leave as write-through (sc1) 16-byte stores,
  // every storing wave drains them, one lane raises this workgroup's flag; one wave polls the partner's flag, one agent-scope
  // acquire drops this CU's stale lines, then plain loads (cdna_hip_programming.md, Guideline 16, R1).  Flags are zero
  // between launches: the reader of a flag clears it.
  f32x4 peer[NPS][2];
  // (the four-part backward flavour keeps the acquire form: with the descriptor loads hipcc moves three fragment registers of its
  // chunk loop to scratch — tools/spill_sites.sh)
  constexpr bool XSC1 = S2T_PC_XCHG_SC1 != 0 && !(BWD && SPLIT == 4);
  if constexpr (SPLIT == 2) {
    {
      float* slab = p.xws + (size_t)(pair * 2 + half) * (64 * 256);
      const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, 64 * 256 * 4, 0x00020000);
      const int srow0 = 64 * (1 - half);
      const int cc = tid & 63;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rr = 8 * q + (tid >> 6);
        const int ml = srow0 + rr;
        const u32x4s t = *reinterpret_cast<const u32x4s*>(smem + ml * 1024 + 16 * (cc ^ (ml & 7)));
        __builtin_amdgcn_raw_buffer_store_b128(t, xs, (uint32_t)(rr * 1024 + 16 * cc), 0, 16 /* sc1 */);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PSTAMP(5);
    typedef __attribute__((address_space(1))) uint32_t gu32;
    gu32* flags = (gu32*)p.xflags;
    if (tid == 0 && !(p.xfault && half == 1))
      __hip_atomic_store(flags + pair * 2 + half, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      gu32* pfl = flags + pair * 2 + (1 - half);
      uint32_t spins = 0;
      const uint32_t spin_limit = p.xfault ? (1u << 8) : SPIN_LIMIT;
      while (__hip_atomic_load(pfl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > spin_limit) {
          // never seen on a resident grid.  The launch is COUNTED in the error word (fixed place, read by the host, which raises
          // and re-zeroes the flag area: kernels.ffn_exchange_check) instead of hanging the chip; this block's rows are invalid.
          if (lane == 0) __hip_atomic_fetch_add(flags + S2T_PC_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      if constexpr (!XSC1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (lane == 0) __hip_atomic_store(pfl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    PSTAMP(6);
    const float* pslab = p.xws + (size_t)(pair * 2 + (1 - half)) * (64 * 256);
    const __amdgpu_buffer_rsrc_t ps_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pslab), 0, 64 * 256 * 4, 0x00020000);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int rr = (KR / 8) * wave + 2 * ps + hi;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if constexpr (XSC1) {
          const u32x4s t = __builtin_amdgcn_raw_buffer_load_b128(ps_rs, (uint32_t)((rr * 256 + 128 * q + 4 * s) * 4), 0, 16 /* sc1 */);
          peer[ps][q] = __builtin_bit_cast(f32x4, t);
        } else {
          peer[ps][q] = *reinterpret_cast<const f32x4*>(pslab + rr * 256 + 128 * q + 4 * s);
        }
      }
    }
  }
  // ---- SPLIT > 2 (few rows: the decoder's 3 904 on 8 x 31 workgroups): every workgroup leaves ALL the rows it does not finish
  // in its own [128][256] fp32 slab, raises one flag per partner, waits for the SPLIT - 1 flags addressed to it (one lane of
  // wave 0 per partner; each reader clears the flags it read) and sums its KR rows over the partners' slabs.
  if constexpr (SPLIT > 2) {
    {
      float* slab = p.xws + (size_t)(pair * SPLIT + half) * (RB * 256);
      const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, RB * 256 * 4, 0x00020000);
      const int cc = tid & 63;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ml = 8 * q + (tid >> 6);
        if (ml / KR == half) continue;  // (wave-uniform)
        const u32x4s t = *reinterpret_cast<const u32x4s*>(smem + ml * 1024 + 16 * (cc ^ (ml & 7)));
        __builtin_amdgcn_raw_buffer_store_b128(t, xs, (uint32_t)(ml * 1024 + 16 * cc), 0, 16 /* sc1 */);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PSTAMP(5);
    typedef __attribute__((address_space(1))) uint32_t gu32;
    gu32* flags = (gu32*)p.xflags;   // [row block][source part][destination part]
    if (tid < SPLIT && tid != half && !(p.xfault && half == 1))
      __hip_atomic_store(flags + (pair * SPLIT + half) * SPLIT + tid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      const bool mine = lane < SPLIT && lane != half;
      gu32* pfl = flags + (pair * SPLIT + (mine ? lane : 0)) * SPLIT + half;
      bool ok = !mine;
      uint32_t spins = 0;
      const uint32_t spin_limit = p.xfault ? (1u << 8) : SPIN_LIMIT;
      while (!__all(ok)) {
        if (!ok) ok = __hip_atomic_load(pfl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u;
        __builtin_amdgcn_s_sleep(4);
        if (++spins > spin_limit) {  // never seen on a resident grid: counted in the error word (see the two-part form above)
          if (lane == 0) __hip_atomic_fetch_add(flags + S2T_PC_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      if constexpr (!XSC1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (mine) __hip_atomic_store(pfl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    PSTAMP(6);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
      for (int q = 0; q < 2; ++q) peer[ps][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int src = 0; src < SPLIT; ++src) {
      if (src == half) continue;
      const float* pslab = p.xws + (size_t)(pair * SPLIT + src) * (RB * 256);
      const __amdgpu_buffer_rsrc_t ps_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pslab), 0, RB * 256 * 4, 0x00020000);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int ml = krow0 + (KR / 8) * wave + 2 * ps + hi;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if constexpr (XSC1) {
            const u32x4s t = __builtin_amdgcn_raw_buffer_load_b128(ps_rs, (uint32_t)((ml * 256 + 128 * q + 4 * s) * 4), 0, 16 /* sc1 */);
            peer[ps][q] += __builtin_bit_cast(f32x4, t);
          } else {
            peer[ps][q] += *reinterpret_cast<const f32x4*>(pslab + ml * 256 + 128 * q + 4 * s);
          }
        }
      }
    }
  }
  auto ysum = [&](int ps, int q, int ml) __attribute__((always_inline)) -> f32x4 {
    const int cc = 32 * q + s;
    f32x4 a = *reinterpret_cast<const f32x4*>(smem + ml * 1024 + 16 * (cc ^ (ml & 7)));
    if constexpr (SPLIT >= 2) a += peer[ps][q];
    return a;
  };

  // ---- row epilogue ----------------------------------------------------------------------------------------------
  const uint64_t key_o = DROP ? s2t_drop_key_of(BWD ? seed_tail : seed_top, p.drop_o_site) : 0ull;
  const uint32_t th_o = s2t_drop_thresh(p.drop_o_p);
  const float inv_o = s2t_drop_scale(p.drop_o_p);
  const bf16_t* R = reinterpret_cast<const bf16_t*>(p.residual);
  bf16_t* Y = reinterpret_cast<bf16_t*>(p.y);
  bf16_t* YL = reinterpret_cast<bf16_t*>(p.y_ln);
  if constexpr (BWD) {
    if (p.lb_x) {
      // ---- backward of the block's leading LayerNorm on the fp32 dXn rows (s2t_layernorm_bwd's arithmetic):
      //   dx = rstd * (dxn*gamma - mean(dxn*gamma) - xhat * mean(dxn*gamma*xhat)) + dres   [+ its dropped copy]
      //   dgamma += sum_rows dxn * xhat, dbeta += sum_rows dxn: lane sums over its rows, 16 (wave, half) groups meet in LDS
      //   (the mailbox region, idle by now), 512 atomics per workgroup into one replica of the workspace
      bf16_t* DX = reinterpret_cast<bf16_t*>(p.lb_dx);
      bf16_t* DXD = reinterpret_cast<bf16_t*>(p.lb_dx_drop);
      const uint64_t key_u = DXD ? s2t_drop_key_of(seed_tail, p.lb_drop_site) : 0ull;
      const uint32_t th_u = s2t_drop_thresh(p.lb_drop_p);
      const float inv_u = s2t_drop_scale(p.lb_drop_p);
      float ag[2][4], ab[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) ag[q][r] = ab[q][r] = 0.f;
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int ml = krow0 + (KR / 8) * wave + 2 * ps + hi;
        const int m = row0 + ml;
        const bool live = m < M;
        const float mu = mupre[ps], rs = rspre[ps];
        float dv[2][4], xh[2][4], dg[2][4], rr[2][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 a = ysum(ps, q, ml);
          const uint2 tx = xpre[ps][q], tr = rpre[ps][q];
          const float xv[4] = {lo16(tx.x), hi16(tx.x), lo16(tx.y), hi16(tx.y)};
          rr[q][0] = lo16(tr.x); rr[q][1] = hi16(tr.x); rr[q][2] = lo16(tr.y); rr[q][3] = hi16(tr.y);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dv[q][r] = live ? a[r] : 0.f;
            xh[q][r] = (xv[r] - mu) * rs;
            dg[q][r] = dv[q][r] * gmm[q][r];
            s1 += dg[q][r];
            s2 += dg[q][r] * xh[q][r];
            ag[q][r] += dv[q][r] * xh[q][r];
            ab[q][r] += dv[q][r];
          }
        }
        s1 = PC_BWD_SUM32(s1);
        s2 = PC_BWD_SUM32(s2);
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
      float* red = reinterpret_cast<float*>(smem + L_MB);  // [2][16][256] fp32 = 32 KiB
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
  for (int ps = 0; ps < NPS; ++ps) {
    const int ml = krow0 + (KR / 8) * wave + 2 * ps + hi;
    const int m = row0 + ml;
    const bool live = m < M;
    float v[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 a = ysum(ps, q, ml);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[q][r] = a[r] + b2v[q][r];
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
        const uint2 t = rpre[ps][q];
        v[q][0] += lo16(t.x); v[q][1] += hi16(t.x); v[q][2] += lo16(t.y); v[q][3] += hi16(t.y);
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
      const bool masked = live && s2t_row_mask_test(p.eln_T, (uint32_t)m, emk[ps]);
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
#if S2T_PC_DBG & 16
  if (lane == 0 && p.eln_mean && !p.eln_gamma && (blockIdx.x == 0 || blockIdx.x == 100)) {
    __builtin_amdgcn_s_waitcnt(0);
    stp[7] = __builtin_amdgcn_s_memtime();
    const unsigned long long real1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.eln_mean) + (blockIdx.x ? 128 : 0) + wave * 16;
    for (int i = 0; i < 8; ++i) dbg[i] = stp[i];
    dbg[8] = real1 - real0;
    for (int i = 0; i < 4; ++i) dbg[9 + i] = seg[i];
  }
#endif
}

template <int MODE>
int launch_pc(const FfnK& k, int split, bool drop, hipStream_t s) {
  const int P = (k.M + RB - 1) / RB;
  const dim3 grid(P * split), block(512);
#define GO(A, DR, SP) hipLaunchKernelGGL((ffn_pc_kernel<MODE, A, DR, SP>), grid, block, 0, s, k)
#define GO_S(A, DR) do { if (split == 8) GO(A, DR, 8); else if (split == 4) GO(A, DR, 4); else if (split == 2) GO(A, DR, 2); else GO(A, DR, 1); } while (0)
#define GO_D(A) do { if (drop) GO_S(A, true); else GO_S(A, false); } while (0)
  if (k.act == S2T_ACT_RELU) GO_D(S2T_ACT_RELU);
  else if (k.act == S2T_ACT_SWISH) GO_D(S2T_ACT_SWISH);
  else GO_D(S2T_ACT_NONE);
#undef GO_D
#undef GO_S
#undef GO
  return S2T_LAUNCH_CHECK();
}

}  // namespace

// Entry points for rowblock.hip's public launchers (the C-ABI stays s2t_ffn_fused_fwd / _bwd).  mode: 0 eval, 1 training
// forward, 2 backward; split: 1, 2, 4 or 8 workgroups per 128-row block.
int s2t_ffn_pc_launch(const void* kargs, int mode, int split, int drop, void* stream) {
  const FfnK& k = *static_cast<const FfnK*>(kargs);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) return launch_pc<0>(k, split, drop != 0, s);
  if (mode == 1) return launch_pc<1>(k, split, drop != 0, s);
  return launch_pc<2>(k, split, drop != 0, s);
}
