// s2t_gemm's large-tile path for gfx950: C[M][N] = epilogue(A[M][K] . B[N][K]^T), bf16 operands, both row-major (the layout of
// every Linear forward: modules/s2t_transformer_layer.py:55-66, modules/multihead_attention.py:161-300, modules/speech_to_text/
// ctc.py:55-66), K a multiple of 64.
//
// Why a second kernel (measured on the 128 x 128, register-staged kernel of gemm.hip): a K-step there moves 32 KB through
// ds_write_b128 (13 LDS cycles per wave-instruction) and re-fetches every operand slab once per 128-wide tile; the structure
// tops out near 700 TFLOP/s.  Here
//   * the tile is 256 x 256 x 64 per 512-thread workgroup (one per CU, 128 KiB of LDS): half the operand bytes per flop;
//   * both operands go global -> LDS by LDS-DMA (buffer_load ... lds, no staging registers, no ds_write), double buffered:
//     step s + 1 is in flight while step s is multiplied — across tile boundaries, so the next tile's first step lands during
//     this tile's epilogue; ONE raw barrier per K-step, counted waits (the compiler does not see the DMAs);
//   * the LDS image of an operand is [256 rows][128 B], 16-byte piece c of row r at r*128 + 16*(c ^ ((r >> 1) & 7)): the
//     swizzle is applied to the DMA's per-lane SOURCE address (the LDS destination of a DMA is lane-linear) and again to the
//     fragment reads, which are then conflict-free ds_read_b128;
//   * eight waves as 2 (M) x 4 (N), 128 x 64 of C per wave = 8 x 4 v_mfma_f32_16x16x32_bf16 tiles (128 accumulator registers).
//     The MFMA is issued with B as its first operand (D rows = n), and the B image holds its rows PERMUTED inside groups of
//     32 (LDS row 16 t + i of a group = B row 8 (i >> 2) + 4 t + (i & 3)), so that the accumulators of an even / odd tile pair
//     of a lane are 8 CONSECUTIVE output columns of one row: the fused epilogue (gemm_common.h, Epi) runs on the registers as
//     they stand, 16-byte bf16 stores, no transposition through LDS;
//   * persistent walk, XCD aware: workgroup w runs on XCD w & 7 (round-robin placement: speed only); an XCD owns whole row
//     blocks and its 32 workgroups take the column tiles of a row block side by side, so an A slab is fetched into ONE L2.
#include <stdio.h>
#include <stdlib.h>

#include "gemm_common.h"
#include "lds_dma.h"
#include "gemm256.h"

#include <type_traits>

#ifndef S2T_G256_DBG
#define S2T_G256_DBG 0  // experiment switches (never in a shipped build): 1 no DMA in the loop, 2 no MFMAs, 4 no epilogue, 8 every C store dropped
#endif

namespace {

constexpr int TN = 256, TK = 64;  // (rows per tile: the kernel's TMR, 256 or 128)
constexpr int OP_BYTES = 32768;          // one operand image of a K-step
constexpr int STAGE_BYTES = 2 * OP_BYTES;
constexpr int G256_LDS = 2 * STAGE_BYTES;

// s_waitcnt immediate of gfx9: vmcnt = bits 3:0 | 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8 (15 = no wait)
constexpr int wait_vm(int vm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | (15 << 8); }

struct Tile {
  int ord, tm, tn, z;  // z: batch index (s2t_gemm's two-level batch, z0 = z / zdiv, z1 = z % zdiv)
  int sp;              // K split (SPK), else 0
};

// VEC: every tensor of the epilogue 16-byte aligned, N % 8 == 0.  PLAIN (needs VEC): the epilogue is bias / activation / alpha /
// residual only (two forms in one kernel made the compiler spill the accumulators where they merge).
// BKM: B is k-major (B_op[k][n] at B[k*ldb + n]: the weight of a Linear seen from its input gradient, the embedding matrix of the
// PAE product).  Its K-step image is [64 k][256 columns] (512 B per k-row, 16-byte piece c of k-row r at r*512 + 16*(c ^ 4*(r & 3))),
// filled by the same DMA pieces (two k-rows each) and read through ds_read_b64_tr_b16, which hands lane x the operand row
// (= B column) quad x >> 2, element x & 3 of the four 4-column quads the 16 lanes address: the quads of tile t of a pair are
// columns 8 p + 4 t .. + 3, which is the column permutation the row-major image carries in its rows.
// GLU (row-major B only): N = 2 * nout weight rows, value rows then gate rows; out = (a + bias_a) * sigmoid(g + bias_g)
// (modules/speech_to_text/subsampling.py:106-159, modules/convolution.py:94-98).  A wave's 64 image rows are the value rows and the
// gate rows of the SAME 32 output columns (a tile is 256 x 128 outputs), so a lane's two pieces are value | gate of its columns.
// TMR: rows per tile.  256, or 128 for problems whose 256-row tiles would fill less than 0.6 of a round (M of the order of 16 000
// with N <= 1024: the Linears of a 64 x 1000 batch at d = 512): the same kernel with 64 x 64 of C per wave (64 accumulator
// registers), a 16 KiB A image and twice the tiles.
// SPK: one K split of a two-phase split-K product (s2t_gemm's c_atomic == 2 with a workspace): the (tile, split) pairs are the
// work items — the split is the slowest index of the row-block walk — and the fp32 partial tile goes to the workspace in the
// 128 x 128 kernel's register-native order, so that gemm.hip's second phase (splitk_reduce / splitk_epilogue) sums either
// kernel's partials.  Long reductions over few output tiles: the input gradients of the vocabulary projections (K = 10 000).
template <typename TC, bool VEC, bool PLAIN, bool BKM, bool GLU = false, int TMR = 256, bool SPK = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const s2t_gemm_args p0) {
  static_assert(!(GLU && (BKM || PLAIN)), "GLU: row-major B, general epilogue");
  static_assert(!SPK || (sizeof(TC) == 4 && VEC && PLAIN && !GLU), "split-K partials: fp32, no epilogue");
  static_assert(TMR == 256 || TMR == 128, "tile rows");
  constexpr int TM = TMR;
  constexpr int TMI = TM / 32;   // 16-row blocks of a wave's TM / 2 rows
  constexpr int APC = TM / 8;    // one-KiB pieces of the A image of a K-step
  __shared__ __attribute__((aligned(16))) char smem[G256_LDS];
  s2t_gemm_args p = p0;
  p.M = (int)s2t_live_rows(p0.row_lens, p0.row_T, p0.M);  // packed batch: the row blocks beyond the live rows are never walked
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int x = lane & 15, y = lane >> 4;

  const int nout = GLU ? p.N / 2 : p.N;             // output columns
  constexpr int TNO = GLU ? TN / 2 : TN;            // ... per tile
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (nout + TNO - 1) / TNO;
  // ---- which tiles: XCD xc owns row blocks xc, xc + 8, ...; its workgroups (slots) take that list's tiles column-fastest
  const int G = gridDim.x;
  int xc = 0, slot = blockIdx.x, nslots = G, nx = 1;
  if ((G & 7) == 0) {
    xc = blockIdx.x & 7;
    slot = blockIdx.x >> 3;
    nslots = G >> 3;
    nx = 8;
  }
  // (a batch is folded into the row-block index: row block rb = z * tiles_m + tm)
  const int nsp = SPK ? p.split_k : 1;
  const int rbs = nsp * p.batch * tiles_m;
  const int my_rows = rbs > xc ? (rbs - xc + nx - 1) / nx : 0;
  const int local_tiles = my_rows * tiles_n;
  if (slot >= local_tiles) return;
  const int my_tiles = (local_tiles - slot + nslots - 1) / nslots;
  const int nkT = (p.K + TK - 1) / TK;         // K-steps of the product
  const int nk = SPK ? (nkT + nsp - 1) / nsp : nkT;  // ... of a work item (the last split's steps at and beyond nkT fetch zeros)
  const int krem = p.K - (nkT - 1) * TK;  // k of the last step: 64, or the tail (a multiple of 8; the pieces beyond it are fetched
                                         // from beyond the descriptors' ranges, which reads as zero)
  const int S = my_tiles * nk;
  auto tile_at = [&](int ord) __attribute__((always_inline)) {
    const int l = slot + ord * nslots;
    const int r = l / tiles_n;
    const int rb = r * nx + xc;
    const int zs = rb / tiles_m;
    const int sp = SPK ? zs / p.batch : 0;
    return Tile{ord, rb - zs * tiles_m, l - r * tiles_n, zs - sp * p.batch, sp};
  };

  // ---- DMA plan.  A K-step's operand image is 32 one-KiB pieces (8 rows each); wave w issues pieces 4 w .. 4 w + 3 of A and of
  // B.  Lane l of piece q lands at row 8 q + (l >> 3), slot l & 7, and therefore FETCHES k-piece (l & 7) ^ swz(row) of the
  // operand row that belongs there.
  const int64_t z0max = (p.batch - 1) / p.zdiv, z1max = p.batch > 1 ? p.zdiv - 1 : 0;
  const i32x4 srdA = make_srd(p.A, (uint32_t)((z0max * p.a_s0 + z1max * p.a_s1 + (int64_t)(p0.M - 1) * p.lda + p.K) * 2));
  const i32x4 srdB = make_srd(p.B, (uint32_t)((z0max * p.b_s0 + z1max * p.b_s1 + (int64_t)((BKM ? p.K : p.N) - 1) * p.ldb + (BKM ? p.N : p.K)) * 2));
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t lda2 = (uint32_t)(p.lda * 2), ldb2 = (uint32_t)(p.ldb * 2);
  // WHO issues: a wave is parked on each piece until the CU's DMA path takes it (tools/ubench/dma_stream.hip: the 64 pieces of a
  // step take ~1.1 k cycles — 60 B/clk/CU from L2, whoever issues them and however many CUs stream — and a wave that issues any of
  // them sits through most of that), and a parked wave issues no MFMA: with all eight waves issuing, every SIMD idles for about
  // that long per step.  FOUR loader waves (0-3: one per SIMD, 16 pieces each, S2T_G256_LOADERS = 4) let the SIMD's other wave
  // multiply meanwhile: 3.26 k -> 2.8 k cycles per step in the micro-benchmark with this kernel's reads and MFMAs; in the kernel,
  // together with the hand-pipelined fragment reads below (a wave must run well WITHOUT a partner hiding its LDS latency), in-kernel
  // stamps read 3.8 k -> 3.0 k cycles per step — and 3-4 % of wall time: the chip is at its power limit in this loop and gives the
  // saved cycles back as clock (1.55 -> 1.77 GHz; MI355X_MICROARCH.md, DVFS give-back).
#ifndef S2T_G256_LOADERS
#define S2T_G256_LOADERS 4
#endif
  // (the general vectorised epilogue at 256-row tiles — GLU included — sits at the register limit: there every wave issues its
  // own pieces, dealt behind the MFMA groups of the step's first half; on the GLU shapes of the subsampler that form measured
  // 12 % FASTER than four loaders — 563 against 641 us at 256 x 1000 x 80 -> 2048 — while the plain-epilogue shapes are equal
  // or up to 4 % better with loaders)
  constexpr int NLD = (TM == 256 && VEC && !PLAIN) ? 8 : S2T_G256_LOADERS;
  constexpr int PPW = 32 / NLD;           // B pieces per issuing wave
  constexpr int PPA = APC / NLD;          // A pieces per issuing wave
  const bool loader = wave < NLD;
  uint32_t va[PPA], vb[PPW];
  int lk0 = 0;  // (SPK) first K-step of the split being fetched
  constexpr uint32_t OOB = 0xfffffff0u;
  auto plan = [&](const Tile& t) __attribute__((always_inline)) {
    if constexpr (SPK) lk0 = t.sp * nk;
    const int z0 = t.z / p.zdiv, z1 = t.z - z0 * p.zdiv;
    const uint32_t abase = (uint32_t)((z0 * p.a_s0 + z1 * p.a_s1) * 2), bbase = (uint32_t)((z0 * p.b_s0 + z1 * p.b_s1) * 2);
#pragma unroll
    for (int q = 0; q < PPA; ++q) {
      const int rho = 8 * (PPA * wave + q) + (lane >> 3);
      const uint32_t piece = (uint32_t)(16 * ((lane & 7) ^ ((rho >> 1) & 7)));
      const int ga = min(t.tm * TM + rho, p.M - 1);  // rows / columns beyond the problem: a clamped duplicate, never stored
      va[q] = abase + (uint32_t)ga * lda2 + piece;
    }
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int pg = PPW * wave + q;       // piece of the B image (8 rows, or 2 k-rows of a k-major B)
      const int rho = 8 * pg + (lane >> 3);
      const uint32_t piece = (uint32_t)(16 * ((lane & 7) ^ ((rho >> 1) & 7)));
      if constexpr (BKM) {
        // piece 4 w + q = k-rows 2 (4 w + q) + (lane >> 5) of the step, slot lane & 31 -> columns 8 (slot ^ swz) of the tile
        const int kr = 2 * pg + (lane >> 5);
        const int col = min(t.tn * TN + 8 * ((lane & 31) ^ (4 * (kr & 3))), p.N - 8);
        vb[q] = bbase + (uint32_t)kr * ldb2 + (uint32_t)(col * 2);
      } else {
        const int i_ = rho & 15, tau = (rho >> 4) & 1;
        int gb;
        if constexpr (GLU) {  // image rows 64 w' + 32 g + ..: g = 0 value, 1 gate rows of output columns 32 w' ..
          gb = ((rho >> 5) & 1) * nout + min(t.tn * TNO + (rho >> 6) * 32 + 8 * (i_ >> 2) + 4 * tau + (i_ & 3), nout - 1);
        } else {
          gb = min(t.tn * TN + (rho & ~31) + 8 * (i_ >> 2) + 4 * tau + (i_ & 3), p.N - 1);
        }
        vb[q] = bbase + (uint32_t)gb * ldb2 + piece;
      }
    }
  };
  auto piece_out = [&](int q, int kt, int stage) __attribute__((always_inline)) {
#if !(S2T_G256_DBG & 1)
    // q: 0 .. PPA-1 the wave's A pieces, PPA .. PPA + PPW - 1 its B pieces
    const bool isa = q < PPA;
    const int qq = isa ? q : q - PPA;
    const uint32_t dst = lds0 + (uint32_t)(stage * STAGE_BYTES) + (uint32_t)(((isa ? PPA : PPW) * wave + qq) * 1024);
    // row-major operand in the tail step: this lane's k-piece is (lane & 7) ^ swz(row) — dropped when it starts at or beyond K
    if constexpr (SPK) kt += lk0;
    const bool tail = kt == nkT - 1 && krem < TK;
    const bool dead = SPK && kt >= nkT;  // (a row-major operand's bytes behind its row end are the next row's, not the range's end)
    const int kpiece = (lane & 7) ^ ((4 * (qq & 1) + (lane >> 4)) & 7);  // swz(row) = (row >> 1) & 7, row = 8 pg + (lane >> 3)
    if (isa) {
      const uint32_t v = (dead || (tail && 8 * kpiece >= krem)) ? OOB : va[qq];
      dma16(dst, v, srdA, (uint32_t)(kt * (TK * 2)));
    } else if constexpr (BKM) {
      // (k-rows at and beyond K lie beyond the descriptor's range by themselves)
      dma16(dst + OP_BYTES, vb[qq], srdB, (uint32_t)kt * (uint32_t)TK * ldb2);
    } else {
      const uint32_t v = (dead || (tail && 8 * kpiece >= krem)) ? OOB : vb[qq];
      dma16(dst + OP_BYTES, v, srdB, (uint32_t)(kt * (TK * 2)));
    }
#endif
  };

  f32x4 acc[TMI][4];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // fragment addresses: row 16 i + x of the wave's rows, k-piece 4 ks + y -> slot (4 ks + y) ^ (x >> 1)  ((row >> 1) & 7 = x >> 1)
  const uint32_t lo0 = (uint32_t)(x * 128 + 16 * (y ^ (x >> 1)));
  const char* const fa0 = smem + wm * ((TM / 2) * 128) + lo0;
  const char* const fb0 = smem + OP_BYTES + wn * (64 * 128) + lo0;
  // k-major B: lane (x = 4 q + p, y) addresses k-row 8 y + 4 half + q (+ 32 ks), the quad at columns 64 wn + 32 g + 8 p + 4 t
  // of tile j = 2 g + t: piece 8 wn + 4 g + p, its half t
  const uint32_t tb0 = (uint32_t)((8 * y + (x >> 2)) * 512);
  const int tq = x >> 2, tp = x & 3;

  // Fragment reads of one k-sub-step (32 k): the wave's four B fragments / two of its eight A fragments
  auto rd_b = [&](int stage, int ks, uint4 (&fb)[4]) __attribute__((always_inline)) {
    if constexpr (BKM) {
      const char* img = smem + stage * STAGE_BYTES + OP_BYTES + ks * (32 * 512);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t w[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          // swizzle key of k-row 8 y + 4 half + q (+ 32 ks): 4 * (row & 3) = 4 q
          const char* a = img + tb0 + half * (4 * 512) + 16 * ((8 * wn + 4 * (j >> 1) + tp) ^ (4 * tq)) + 8 * (j & 1);
          const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
          const uint2 tt = __builtin_bit_cast(uint2, t);
          w[2 * half] = tt.x;
          w[2 * half + 1] = tt.y;
        }
        fb[j] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else {
      // (x*128 + 16*((4+y) ^ (x>>1))) = lo0 ^ 64
      const char* lb = fb0 + stage * STAGE_BYTES + (ks ? (int)((lo0 ^ 64u) - lo0) : 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(lb + j * 2048);
    }
  };
  auto mma = [&](int i, int j, const uint4& b, const uint4& a) __attribute__((always_inline)) {
#if S2T_G256_DBG & 2
    asm volatile("" :: "v"(b.x), "v"(b.w), "v"(a.x), "v"(a.w));
#else
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc[i][j], 0, 0, 0);
#endif
  };
  // The step as TMI groups of eight MFMAs (k-sub-step ks = g / (TMI / 2), A rows 32 (g % (TMI / 2)) .. + 32 of the wave's): the two
  // A fragments of group g + 1 (and, early in the first sub-step, the B fragments of the second) are requested BEFORE the MFMAs of
  // group g are issued, the groups pinned by sched_barrier — left to itself hipcc requests a sub-step's fragments together and
  // lets its first MFMA wait for them, which two waves sharing a SIMD hide for each other but a wave whose partner is parked on
  // DMA issue does not.  The MFMA order (hence every result bit) is the one of plain i / j loops inside each sub-step.
  auto multiply = [&](int stage, auto&& side) __attribute__((always_inline)) {
    const char* la = fa0 + stage * STAGE_BYTES;
    constexpr int GH = TMI / 2;  // groups per sub-step
    uint4 fb[2][4], fa[2][2];
    auto rd_a = [&](int g, uint4 (&f)[2]) __attribute__((always_inline)) {
      const char* pa = la + ((g / GH) ? (int)((lo0 ^ 64u) - lo0) : 0) + (g % GH) * 4096;
      f[0] = *reinterpret_cast<const uint4*>(pa);
      f[1] = *reinterpret_cast<const uint4*>(pa + 2048);
    };
    rd_b(stage, 0, fb[0]);
    rd_a(0, fa[0]);
#pragma unroll
    for (int g = 0; g < TMI; ++g) {
      if (g + 1 < TMI) rd_a(g + 1, fa[(g + 1) & 1]);
      if (g == (GH > 1 ? 1 : 0)) rd_b(stage, 1, fb[1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(2 * (g % GH) + ii, j, fb[g / GH][j], fa[g & 1][ii]);
      __builtin_amdgcn_sched_barrier(0);
      if (g < GH) side(g);
    }
  };

  // ---- epilogue of tile (tm, tn): lane (x, y) holds, for row block i and tile pair jp, columns 32 jp + 8 y .. + 7 of row
  // 16 i + x of the wave's 128 x 64.  Same arithmetic, in the same order, as Epi::finish (gemm_common.h) — the results of the two
  // GEMM paths are equal bit for bit.  The epilogue is VALU-issue bound (in-kernel stamps: 8.5 k cycles per tile for the general
  // form, against 2 k for a K-step), so the common case — bias, activation, alpha, residual — has its own fully unrolled form on
  // static accumulator indices; the general form (act' input, dropout, pre-activation copy, padded-frame mask) walks two row
  // blocks per trip and shifts the remaining accumulators down (one copy of the long code).
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
#if S2T_G256_DBG & 4
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
#else
    if constexpr (SPK) {
      // acc[i][2c + tau] = row 16 i + x of the wave's rows, columns 64 wn + 32 c + 8 y + 4 tau .. + 3 of the tile; in the 128 x 128
      // kernel's order (gemm.hip, splitk_reduce_kernel) element (m, n) sits in tile (m >> 7, n >> 7) at
      // ((i' * 4 + j') * 256 + (wm' * 2 + wn') * 64 + y' * 16 + x') * 4 + r with m & 127 = 64 wm' + 16 i' + x', n & 127 = 64 wn' + 16 j' + 4 y' + r
      const int t128n = (p.N + 127) >> 7, t128m = (p0.M + 127) >> 7;
      const int64_t slice = (int64_t)t128m * t128n * 16384;  // floats of one (batch, split)
      const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
          p.ws + ((int64_t)t.z * p.split_k + t.sp) * slice, 0, (int)(uint32_t)(slice * 4), 0x00020000);
      const int nt = 2 * t.tn + (wn >> 1);
#pragma unroll
      for (int i = 0; i < TMI; ++i) {
        const int mt = TM == 256 ? 2 * t.tm + wm : t.tm;
        const int wq = TM == 256 ? (i >> 2) : wm, iq = i & 3;
        const bool ok = mt < t128m && nt < t128n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int jq = 2 * (j >> 1) + (y >> 1), yq = 2 * (y & 1) + (j & 1);
          const uint32_t off = ok ? (uint32_t)(((mt * t128n + nt) * 16384 + ((iq * 4 + jq) * 256 + (wq * 2 + (wn & 1)) * 64 + yq * 16 + x) * 4) * 4)
                                  : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){__float_as_uint(acc[i][j][0]), __float_as_uint(acc[i][j][1]),
                                                          __float_as_uint(acc[i][j][2]), __float_as_uint(acc[i][j][3])}, wsrd, off, 0, 0);
        }
      }
      return;
    }
    const int ez0 = t.z / p.zdiv, ez1 = t.z - ez0 * p.zdiv;
    const int64_t coff = ez0 * p.c_s0 + ez1 * p.c_s1;
    const int64_t grow0 = (int64_t)t.z * p.M;  // global row of the batch's row 0 (mask, dropout index)
    Epi<TC, VEC> e{p,
                   reinterpret_cast<TC*>(p.C) + coff,
                   p.residual ? reinterpret_cast<const TC*>(p.residual) + coff : nullptr,
                   p.preact ? reinterpret_cast<TC*>(p.preact) + (ez0 * p.p_s0 + ez1 * p.p_s1) : nullptr,
                   p.dact_z ? reinterpret_cast<const TC*>(p.dact_z) + coff : nullptr,
                   nout,
                   false, false, false, false};
    if constexpr (!VEC) {
      auto vec_ok = [](const void* ptr, int64_t ld) { return ((ld * (int64_t)sizeof(TC)) % 16 == 0) && (((uintptr_t)ptr) % 16 == 0); };
      e.vec_c = vec_ok(e.C, p.ldc);
      e.vec_r = e.R && vec_ok(e.R, p.ldr);
      e.vec_p = e.P && vec_ok(e.P, p.ldp);
      e.vec_z = e.Z && vec_ok(e.Z, p.ldz);
    }
    float bpre[2][8];
    int ncol[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      ncol[jp] = GLU ? t.tn * TNO + wn * 32 + 8 * y : t.tn * TN + wn * 64 + 32 * jp + 8 * y;
      const int nvb = max(0, min(8, nout - ncol[jp]));
      e.bias8(nvb > 0 ? (GLU ? jp * nout : 0) + ncol[jp] : 0, nvb, bpre[jp]);
    }
    // C leaves through a raw buffer descriptor: a piece outside the problem gets an offset beyond the descriptor's range and is
    // dropped by the hardware, so every lane ISSUES the same number of stores — the wait at the top of the next step can then
    // leave exactly those in flight
    const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(
        e.C, 0, (int)(uint32_t)(((int64_t)(p.M - 1) * p.ldc + nout) * (int64_t)sizeof(TC)), 0x00020000);
    auto store8 = [&](bool ok, int m, int n, const float (&v)[8]) __attribute__((always_inline)) {
#if S2T_G256_DBG & 8
      const uint32_t off = 0xfffffff0u;
      asm volatile("" :: "v"(ok), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
#else
      const uint32_t off = ok ? (uint32_t)(((int64_t)m * p.ldc + n) * (int64_t)sizeof(TC)) : 0xfffffff0u;
#endif
      if constexpr (sizeof(TC) == 2) {
        __builtin_amdgcn_raw_buffer_store_b128(
            (u32x4_t){bf16pack(v[0], v[1]), bf16pack(v[2], v[3]), bf16pack(v[4], v[5]), bf16pack(v[6], v[7])}, csrd, off, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                        __float_as_uint(v[3])}, csrd, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]),
                                                        __float_as_uint(v[7])}, csrd, ok ? off + 16u : 0xfffffff0u, 0, 0);
      }
    };
    const bool has_act = p.act == S2T_ACT_RELU || p.act == S2T_ACT_SWISH;
    const bool drop = p.drop_p > 0.f;
    if constexpr (PLAIN) {
      const bool scale = p.alpha != 1.0f;
#pragma unroll
      for (int i = 0; i < TMI; ++i) {
        const int m = t.tm * TM + wm * (TM / 2) + i * 16 + x;
        bool ok[2];
        float v[2][8], q[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c) ok[c] = m < p.M && ncol[c] < nout;
        if (e.R) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
            if (ok[c]) ld8<TC>(e.R + (int64_t)m * p.ldr + ncol[c], true, 8, q[c]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[c][r] = acc[i][2 * c][r] + bpre[c][r];
            v[c][4 + r] = acc[i][2 * c + 1][r] + bpre[c][4 + r];
          }
        if (p.act == S2T_ACT_RELU) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[c][r] = v[c][r] > 0.f ? v[c][r] : 0.f;
        } else if (p.act == S2T_ACT_SWISH) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[c][r] = v[c][r] * sigmoidf_(v[c][r]);
        }
        if (scale) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[c][r] *= p.alpha;
        }
        if (e.R) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[c][r] += ok[c] ? q[c][r] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) store8(ok[c], m, ncol[c], v[c]);
      }
    } else {
      const uint64_t dkey = drop ? s2t_drop_key(p.drop_seed, p.drop_site) : 0ull;
      const uint32_t dth = s2t_drop_thresh(p.drop_p);
      const float dinv = s2t_drop_scale(p.drop_p);
      f32x4 (&af)[4 * TMI] = reinterpret_cast<f32x4 (&)[4 * TMI]>(acc);
#pragma unroll 1
      for (int i2 = 0; i2 < TMI / 2; ++i2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = t.tm * TM + wm * (TM / 2) + (2 * i2 + h) * 16 + x;
          if constexpr (GLU) {
            const f32x4 a0 = af[4 * h], a1 = af[4 * h + 1], g0 = af[4 * h + 2], g1 = af[4 * h + 3];
            if (m < p.M && ncol[0] < nout) {
              float a[8], gt[8], v[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                a[r] = a0[r] + bpre[0][r];
                a[4 + r] = a1[r] + bpre[0][4 + r];
                gt[r] = g0[r] + bpre[1][r];
                gt[4 + r] = g1[r] + bpre[1][4 + r];
              }
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] = a[r] * sigmoidf_(gt[r]);
              if (e.P) {
                const int nv = min(8, nout - ncol[0]);
                st8<TC>(e.P + (int64_t)m * p.ldp + ncol[0], VEC || e.vec_p, VEC ? 8 : nv, a);
                st8<TC>(e.P + (int64_t)m * p.ldp + nout + ncol[0], VEC || (e.vec_p && ((nout * (int)sizeof(TC)) % 16 == 0)), VEC ? 8 : nv, gt);
              }
              e.finish(m, ncol[0], grow0 + m, v);
            }
          } else if constexpr (VEC) {
            bool ok[2];
            float v[2][8], q[2][8], zz[2][8];
#pragma unroll
            for (int c = 0; c < 2; ++c) ok[c] = m < p.M && ncol[c] < nout;
            if (e.R) {
#pragma unroll
              for (int c = 0; c < 2; ++c)
                if (ok[c]) ld8<TC>(e.R + (int64_t)m * p.ldr + ncol[c], true, 8, q[c]);
            }
            if (e.Z) {
#pragma unroll
              for (int c = 0; c < 2; ++c)
                if (ok[c]) ld8<TC>(e.Z + (int64_t)m * p.ldz + ncol[c], true, 8, zz[c]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const f32x4 t0 = af[4 * h + 2 * c], t1 = af[4 * h + 2 * c + 1];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[c][r] = t0[r] + bpre[c][r];
                v[c][4 + r] = t1[r] + bpre[c][4 + r];
              }
            }
            if (has_act) {
              if (e.P) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  if (ok[c]) st8<TC>(e.P + (int64_t)m * p.ldp + ncol[c], true, 8, v[c]);
              }
              if (p.act == S2T_ACT_RELU) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                  for (int r = 0; r < 8; ++r) v[c][r] = v[c][r] > 0.f ? v[c][r] : 0.f;
              } else {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                  for (int r = 0; r < 8; ++r) v[c][r] = v[c][r] * sigmoidf_(v[c][r]);
              }
            }
            if (e.Z) {
              if (p.dact == S2T_ACT_RELU) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                  for (int r = 0; r < 8; ++r) v[c][r] *= zz[c][r] > 0.f ? 1.f : 0.f;
              } else if (p.dact == S2T_ACT_SWISH) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                  for (int r = 0; r < 8; ++r) v[c][r] *= act_grad(S2T_ACT_SWISH, zz[c][r]);
              }
            }
            if (drop) {
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                uint32_t r16[8];
                s2t_rand_run<8>(dkey, (uint64_t)(grow0 + m) * (uint64_t)nout + (uint64_t)ncol[c], r16);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[c][r] = r16[r] >= dth ? v[c][r] * dinv : 0.f;
              }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int r = 0; r < 8; ++r) v[c][r] *= p.alpha;
            if (p.row_lens && m < p.M && s2t_row_masked32(p.row_lens, p.row_T, (uint32_t)(grow0 + m))) {
#pragma unroll
              for (int r = 0; r < 8; ++r) v[0][r] = v[1][r] = 0.f;
            }
            if (e.R) {
#pragma unroll
              for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 8; ++r) v[c][r] += ok[c] ? q[c][r] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) store8(ok[c], m, ncol[c], v[c]);
          } else {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
              const f32x4 t0 = af[4 * h + 2 * jp], t1 = af[4 * h + 2 * jp + 1];
              float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
              if (m < p.M && ncol[jp] < nout) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += bpre[jp][r];
                e.finish(m, ncol[jp], grow0 + m, v);
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4 * TMI - 8; ++k) af[k] = af[k + 8];
      }
    }
#endif
  };

  // ---- the walk: step s multiplies LDS stage s & 1 while step s + 1 lands in the other.
  // (S2T_G256_EARLY2, an experiment that LOST: at the end of a tile the stage just multiplied is free as well, so behind one more
  // barrier the next tile's SECOND step can be sent into it before the epilogue starts and the next tile's first step issues
  // nothing.  64000 x 2048 x 512: 185-193 us without, 193-196 us with; 16000 x 10000 x 256: 131 / 139.)
  Tile L = tile_at(0);  // tile / K-step of the next step to FETCH (step fs)
  int lkt = 0;
  plan(L);
  if (loader) {
#pragma unroll
    for (int q = 0; q < PPA + PPW; ++q) piece_out(q, 0, 0);
  }
  auto advance_fetch = [&]() __attribute__((always_inline)) {
    if (lkt + 1 < nk) {
      ++lkt;
    } else {
      L = tile_at(L.ord + 1);
      lkt = 0;
      plan(L);
    }
  };
  int fs = 1;           // steps fetched (or in flight) so far
  if (S > 1) advance_fetch();
  Tile C = tile_at(0);  // tile of the step being multiplied
  int ckt = 0;
  // stores a VEC epilogue issues per lane (none of them skipped: see csrd): 8 row blocks x 2 pieces x (1 | 2) 16-byte stores
  constexpr int NST = 2 * TMI * (int)(sizeof(TC) / 2);
  const bool counted = VEC && !GLU && !p.preact;   // (the pre-activation copy and the GLU form go out by ordinary conditional stores)
  int since = 99;       // steps since a tile end: 0 with the next step AND the one after in flight, 1 with one step in flight
#ifndef S2T_G256_EARLY2
#define S2T_G256_EARLY2 0
#endif
#if S2T_G256_DBG & 16
  unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
  int since_epi = 99;
#define G256_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G256_STAMP(v)
#endif
  for (int s = 0; s < S; ++s) {
#if S2T_G256_DBG & 16
    unsigned long long t0, t1, t2, t3;
#endif
    G256_STAMP(t0);
    // This wave's pieces of step s have landed; behind the barrier everybody's have, and everybody has finished reading the
    // other stage.  What may stay in flight is YOUNGER than those pieces: behind a tile end the (eight) pieces of step s + 1
    // and the epilogue's stores, one step later the stores alone.
    if (since == 0) {
      if (counted) __builtin_amdgcn_s_waitcnt(wait_vm(PPA + PPW + NST));
      else __builtin_amdgcn_s_waitcnt(wait_vm(PPA + PPW));
    } else if (since == 1 && counted) {
      __builtin_amdgcn_s_waitcnt(wait_vm(NST));
    } else {
      __builtin_amdgcn_s_waitcnt(wait_vm(0));
    }
    asm volatile("s_barrier" ::: "memory");
    G256_STAMP(t1);
    const bool issue = fs == s + 1 && fs < S;
    if constexpr (NLD == 8) {
      // every wave issues: its PPA + PPW pieces dealt behind the MFMA groups of the first sub-step
      constexpr int PER = (PPA + PPW + TMI / 2 - 1) / (TMI / 2);
      multiply(s & 1, [&](int g) __attribute__((always_inline)) {
        if (issue) {
#pragma unroll
          for (int q = PER * g; q < PER * (g + 1) && q < PPA + PPW; ++q) piece_out(q, lkt, (s & 1) ^ 1);
        }
      });
    } else {
      // loader waves: the whole step's pieces at its top
      if (issue && loader) {
#pragma unroll
        for (int q = 0; q < PPA + PPW; ++q) piece_out(q, lkt, (s & 1) ^ 1);
      }
      multiply(s & 1, [&](int) __attribute__((always_inline)) {});
    }
    if (issue) {
      ++fs;
      if (fs < S) advance_fetch();
    }
    ++since;
    G256_STAMP(t2);
#if S2T_G256_DBG & 16
    seg[since_epi == 0 ? 0 : since_epi == 1 ? 1 : 2] += t1 - t0;
    seg[3] += t2 - t1;
    seg[5] += 1;
    ++since_epi;
#endif
    if (ckt + 1 == nk) {
      const bool more = s + 1 < S;
      since = 1;
      if (S2T_G256_EARLY2 && fs == s + 2 && fs < S) {
        asm volatile("s_barrier" ::: "memory");  // every wave has finished reading stage s & 1
        if (loader) {
#pragma unroll
          for (int q = 0; q < PPA + PPW; ++q) piece_out(q, lkt, s & 1);
        }
        ++fs;
        if (fs < S) advance_fetch();
        since = 0;
      }
      epilogue(C);
#if S2T_G256_DBG & 16
      G256_STAMP(t3);
      seg[4] += t3 - t2;
      since_epi = 0;
      if (!more && p.colsum_a && tid == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.colsum_a[blockIdx.x * 8 + k] = (float)seg[k];
      }
#endif
      if (!more) break;
      zero_acc();
      C = tile_at(C.ord + 1);
      ckt = 0;
    } else {
      ++ckt;
    }
  }
  __builtin_amdgcn_s_waitcnt(wait_vm(0));  // no DMA may still be writing this workgroup's LDS when it is handed on
}

}  // namespace

// ---- host side (called by s2t_gemm) ---------------------------------------------------------------------------------
// S2T_GEMM256 / s2t_gemm_configure: 0 never, 1 (default) where it measured faster, 2 whenever the arguments allow (256-row
// tiles), 3 the same with 128-row tiles
static int& g256_mode_ref() {
  static int mode = [] { const char* e = getenv("S2T_GEMM256"); return e ? atoi(e) : 1; }();
  return mode;
}
static int g256_mode() { return g256_mode_ref(); }

extern "C" int s2t_gemm_configure(int large_tile_mode) {
  if (large_tile_mode >= 0) g256_mode_ref() = large_tile_mode > 3 ? 3 : large_tile_mode;
  return g256_mode_ref();
}

// rows per tile of the large-tile path for these arguments: 256, 128, or 0 (the 128 x 128 kernel of gemm.hip)
int s2t_gemm256_tile_rows(const s2t_gemm_args& p) {
  const int mode = g256_mode();
  if (mode <= 0) return 0;
  if (p.dtype != S2T_BF16 || p.a_kmajor || (p.act == S2T_ACT_GLU && p.b_kmajor)) return 0;
#if S2T_G256_DBG & 16
  if (p.split_k > 1 || p.c_atomic || p.ws) return 0;  // (p.colsum_a receives the stamps)
#else
  // split-K: the two-phase (workspace) form only — s2t_gemm has normalised the arguments: p.ws is set iff that form runs
  const bool spk = p.split_k > 1 && p.c_atomic == 2 && p.ws;
  if (!spk && (p.split_k > 1 || p.c_atomic || p.ws)) return 0;
  if (p.colsum_a) return 0;
  if (spk) {
    if (p.act == S2T_ACT_GLU) return 0;
    const int64_t slice = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) * 16384 * 4;  // bytes of one (batch, split) of partials
    if (slice >= (1ll << 32) - 64) return 0;
    if (p.K < 64 * 8 * p.split_k) return 0;  // (at least eight K-steps per split: below that the partial tiles outweigh the product)
  }
  if (p.batch > 1) {
    // a batch is folded into the row-block walk: every operand of the whole batch inside one 32-bit byte range, and no
    // k-major B (its K tail relies on the descriptor's end)
    if (p.b_kmajor) return 0;
    const int64_t z0max = (p.batch - 1) / p.zdiv, z1max = p.zdiv - 1;
    const int64_t aspan = (z0max * p.a_s0 + z1max * p.a_s1 + (int64_t)(p.M - 1) * p.lda + p.K) * 2;
    const int64_t bspan = (z0max * p.b_s0 + z1max * p.b_s1 + (int64_t)(p.N - 1) * p.ldb + p.K) * 2;
    if (p.a_s0 < 0 || p.a_s1 < 0 || p.b_s0 < 0 || p.b_s1 < 0 || aspan >= (1ll << 32) - 64 || bspan >= (1ll << 32) - 64) return 0;
  }
#endif
  if (p.K < 128 || (p.K % 8)) return 0;          // (a K tail is dropped in whole 16-byte pieces)
  {
    // C leaves through a buffer descriptor with 32-bit byte offsets (per batch)
    const int nout = p.act == S2T_ACT_GLU ? p.N / 2 : p.N;
    if (((int64_t)(p.M - 1) * p.ldc + nout) * (p.c_dtype == S2T_F32 ? 4 : 2) >= (1ll << 32) - 64) return 0;
  }
  if (p.b_kmajor && (p.N % 8)) return 0;         // (a k-major piece is 8 columns)
  if (mode == 2) return 256;
  if (mode >= 3) return 128;
  // (tools/gemm256_probe.py border: 156 tiles of 256 x 256 1.39x, 189 1.38x, 250 1.3x; 126-128 tiles 0.95-1.05x, 88 0.92x, 64 0.78x
  // — below about 0.6 of a round the 128 x 128 path's 2 x 256 slots fill the chip better; 128-row tiles when THOSE reach 0.6)
  const int64_t cols = (int64_t)(spk ? p.split_k : 1) * ((p.N + TN - 1) / TN);  // (GLU: N / 2 outputs in 128-column tiles: the same count)
  if ((int64_t)p.batch * ((p.M + 255) / 256) * cols >= 150) return 256;
  // (not the GLU form: the subsampler's second convolution — 64 x 250 rows, K = 2560 — measured 85 us on 128-row tiles against
  // 64 on the 128 x 128 kernel)
  if (p.act != S2T_ACT_GLU && (int64_t)p.batch * ((p.M + 127) / 128) * cols >= 150) return 128;
  return 0;
}

bool s2t_gemm256_eligible(const s2t_gemm_args& p) { return s2t_gemm256_tile_rows(p) != 0; }

static bool g256_plain(const s2t_gemm_args& p, bool vec) {
  // (a row map that only BOUNDS the rows — S2T_ROWS_BOUND — is no mask: the live row count is read in every instantiation)
  return vec && p.act != S2T_ACT_GLU && !p.dact_z && !(p.drop_p > 0.f) && !p.preact && !(p.row_lens && p.row_T != S2T_ROWS_BOUND);
}

int s2t_gemm256_launch(const s2t_gemm_args& p, bool vec, hipStream_t s) {
  const dim3 grid(s2t_device_cu_count()), block(512);
  const bool plain = g256_plain(p, vec);
  const int rows = s2t_gemm256_tile_rows(p);
  if (p.ws) {  // one split of a two-phase split-K product: fp32 partial tiles, the caller (s2t_gemm) runs the second phase
    if (p.b_kmajor) {
      if (rows == 128) hipLaunchKernelGGL((gemm256_kernel<float, true, true, true, false, 128, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((gemm256_kernel<float, true, true, true, false, 256, true>), grid, block, 0, s, p);
    } else {
      if (rows == 128) hipLaunchKernelGGL((gemm256_kernel<float, true, true, false, false, 128, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((gemm256_kernel<float, true, true, false, false, 256, true>), grid, block, 0, s, p);
    }
    return S2T_LAUNCH_CHECK();
  }
#define GO3(TC, BK, R) \
  do { \
    if (plain) hipLaunchKernelGGL((gemm256_kernel<TC, true, true, BK, false, R>), grid, block, 0, s, p); \
    else if (vec) hipLaunchKernelGGL((gemm256_kernel<TC, true, false, BK, false, R>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((gemm256_kernel<TC, false, false, BK, false, R>), grid, block, 0, s, p); \
  } while (0)
#define GO2(TC, BK) \
  do { \
    if (rows == 128) GO3(TC, BK, 128); \
    else GO3(TC, BK, 256); \
  } while (0)
#define GO(TC) \
  do { \
    if (p.act == S2T_ACT_GLU) { \
      if (rows == 128) { \
        if (vec) hipLaunchKernelGGL((gemm256_kernel<TC, true, false, false, true, 128>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((gemm256_kernel<TC, false, false, false, true, 128>), grid, block, 0, s, p); \
      } else { \
        if (vec) hipLaunchKernelGGL((gemm256_kernel<TC, true, false, false, true>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((gemm256_kernel<TC, false, false, false, true>), grid, block, 0, s, p); \
      } \
    } else if (p.b_kmajor) GO2(TC, true); \
    else GO2(TC, false); \
  } while (0)
  if (p.c_dtype == S2T_F32) GO(float);
  else GO(bf16_t);
#undef GO3
#undef GO2
#undef GO
  return S2T_LAUNCH_CHECK();
}

int s2t_gemm256_describe(const s2t_gemm_args& p, bool vec, char* buf, int buflen) {
  int n;
  if (p.ws)
    n = snprintf(buf, buflen, "gemm256_kernel<float, true, true, %s, false, %d, true>", p.b_kmajor ? "true" : "false",
                 s2t_gemm256_tile_rows(p) == 128 ? 128 : 256);
  else
    n = snprintf(buf, buflen, "gemm256_kernel<%s, %s, %s, %s, %s, %d, false>", p.c_dtype == S2T_F32 ? "float" : "unsigned short",
                 vec ? "true" : "false", g256_plain(p, vec) ? "true" : "false", p.b_kmajor ? "true" : "false",
                 p.act == S2T_ACT_GLU ? "true" : "false", s2t_gemm256_tile_rows(p) == 128 ? 128 : 256);
  return (n > 0 && n < buflen) ? S2T_OK : S2T_ERR_ARG;
}
