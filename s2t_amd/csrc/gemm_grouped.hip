// Grouped weight-gradient GEMM: every dW[Nout, Kin] += alpha * dY[rows, Nout]^T @ X[rows, Kin] of one backward pass in
// ONE persistent launch (+ one reduce launch), instead of one small split-K GEMM per parameter.
//
// Why: the ~160 weight-gradient GEMMs of a training step are individually small (4 to 32 output tiles, K = B*T' = 16 000):
// each launch pays its own ramp-up, tail and split-K reduction, and the small ones run at 50-150 TFLOP/s.  Nothing in
// backward consumes a weight gradient before the optimizer, so the host queues (dY, X, dW) triples and flushes them
// at the end of backward (functional._wgrad / flush_wgrads).  The work list is cut into items of one 128x128 output tile
// x up to `ksteps` K-steps; persistent workgroups (2 per CU) walk the items back to back with the same register
// prefetch across item boundaries as gemm.hip, store fp32 partial tiles to a workspace in register-native order, and a
// second kernel sums the partials of each tile into dW.  Bias gradients (column sums of dY) are taken from the staged
// dY tiles as in gemm.hip.
//
// Reference semantics replaced: the autograd backward of every F.linear / pointwise Conv1d on the path
// (modules/s2t_transformer_layer.py:55-66, multihead_attention.py:239-263, espnet_multihead_attention.py:88-106,
// convolution.py:91,106, speech_to_text/ctc.py:59, models/transformer.py:1442).
#include <type_traits>
#include <utility>

#include "gemm_common.h"
#include "lds_dma.h"

namespace {

constexpr int GD = 2;  // register sets (K-steps of global loads in flight per workgroup)

struct Item {  // one work item, 16 bytes
  int32_t prob, tm, tn, split;
};

// position of a K-step in the walk (all workgroup-uniform)
struct GCursor {
  int ord;             // item ordinal of this workgroup
  int kt, kt1, kt_end; // current K-tile, end of the item's real K-tiles, end incl. padding to a multiple of GD
};

template <bool KT>
__global__ __launch_bounds__(256, 2) void wgrad_grouped_kernel(const s2t_wgrad_problem* __restrict__ probs,
                                                               const Item* __restrict__ items, int n_items,
                                                               float* __restrict__ ws) {
  typedef bf16_t T;
  constexpr int BKE = 64, EPB = 8;
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int x = lane & 15, y = lane >> 4;

  const int G = gridDim.x;
  int w = blockIdx.x;
  if ((G & 7) == 0) w = (w & 7) * (G >> 3) + (w >> 3);  // XCD-contiguous item ranges (see gemm.hip)
  if (w >= n_items) return;
  const int my_items = (n_items - w + G - 1) / G;

  // ---- load side state
  Item li;
  const s2t_wgrad_problem* lp;
  LoadPlan pa, pb;
  auto load_item = [&](int ord, GCursor& c) __attribute__((always_inline)) {
    li = items[w + ord * G];
    lp = probs + li.prob;
    const int ktiles = (lp->K + BKE - 1) / BKE;
    c.ord = ord;
    c.kt = li.split * lp->ksteps;
    c.kt1 = min(ktiles, c.kt + lp->ksteps);
    c.kt_end = c.kt + (c.kt1 - c.kt + GD - 1) / GD * GD;
    plan_kmajor<T>(pa, lp->lda, li.tm * BM, lp->M, tid);
    plan_kmajor<T>(pb, lp->ldb, li.tn * BN, lp->N, tid);
  };
  uint4 ra[GD][4], rb[GD][4];
  auto gload = [&](uint4 (&qa)[4], uint4 (&qb)[4], const GCursor& c) __attribute__((always_inline)) {
    const int k0 = min(c.kt, c.kt1 - 1) * BKE;
    load_kmajor<T, KT>(qa, reinterpret_cast<const char*>(lp->A), pa, lp->lda, k0, lp->K, tid);
    load_kmajor<T, KT>(qb, reinterpret_cast<const char*>(lp->B), pb, lp->ldb, k0, lp->K, tid);
  };
  auto advance_load = [&](GCursor& c) __attribute__((always_inline)) {
    if (c.kt + 1 < c.kt_end) ++c.kt;
    else if (c.ord + 1 < my_items) load_item(c.ord + 1, c);
  };

  // ---- compute side state (the step being multiplied and the one landing in LDS next)
  struct CItem {
    Item it;
    const s2t_wgrad_problem* p;
    int kt, kt1, kt_end, K;
  };
  auto citem_at = [&](int ord) __attribute__((always_inline)) {
    CItem c;
    c.it = items[w + ord * G];
    c.p = probs + c.it.prob;
    c.K = c.p->K;
    const int ktiles = (c.K + BKE - 1) / BKE;
    c.kt = c.it.split * c.p->ksteps;
    c.kt1 = min(ktiles, c.kt + c.p->ksteps);
    c.kt_end = c.kt + (c.kt1 - c.kt + GD - 1) / GD * GD;
    return c;
  };

  f32x4 acc[4][4];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  float csum[EPB];
#pragma unroll
  for (int e = 0; e < EPB; ++e) csum[e] = 0.f;

  auto gfix = [&](uint4 (&qa)[4], uint4 (&qb)[4], int kt, int kt1, int K) __attribute__((always_inline)) {
    if (kt >= kt1) {  // padding step
#pragma unroll
      for (int u = 0; u < 4; ++u) qa[u] = qb[u] = make_uint4(0, 0, 0, 0);
    } else if constexpr (KT) {
      const int k0 = kt * BKE;
      if (k0 + BKE > K) {
        fix_kmajor<T>(qa, k0, K, tid);
        fix_kmajor<T>(qb, k0, K, tid);
      }
    }
  };
  auto colsum_acc = [&](const uint4 (&qa)[4], bool on) __attribute__((always_inline)) {
    if (on) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t w4[4] = {qa[u].x, qa[u].y, qa[u].z, qa[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          csum[2 * q] += __uint_as_float(w4[q] << 16);
          csum[2 * q + 1] += __uint_as_float(w4[q] & 0xffff0000u);
        }
      }
    }
  };
  auto lstore = [&](int buf, const uint4 (&qa)[4], const uint4 (&qb)[4]) __attribute__((always_inline)) {
    char* la = smem + buf * 32768;
    store_kmajor<T>(la, qa, tid);
    store_kmajor<T>(la + 16384, qb, tid);
  };

  // ---- epilogue of one item: partial tile -> workspace (register-native order), bias-gradient partial -> atomics
  auto epilogue = [&](const CItem& c) __attribute__((always_inline)) {
    const s2t_wgrad_problem* p = c.p;
    const int tile = c.it.tm * p->tiles_n + c.it.tn;
    float* wt = ws + p->ws_base + ((int64_t)tile * p->nsplit + c.it.split) * (int64_t)(BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(wt + ((i * 4 + j) * 256 + tid) * 4) = acc[i][j];
    if (p->colsum && c.it.tn == 0) {  // workgroup-uniform
      constexpr int CPR_A = 128 / EPB;
      __syncthreads();
      float* lc = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int e = 0; e < EPB; ++e) {
        lc[tid * EPB + e] = csum[e];
        csum[e] = 0.f;
      }
      __syncthreads();
      if (tid < 128) {
        const int ch = tid / EPB, e = tid % EPB;
        float sum = 0.f;
        for (int j = 0; j < 256 / CPR_A; ++j) sum += lc[(ch + CPR_A * j) * EPB + e];
        const int m = c.it.tm * BM + tid;
        if (m < p->M) atomicAdd(p->colsum + m, p->alpha * sum);
      }
      __syncthreads();
    }
  };

  // ---- the walk
  GCursor L;
  load_item(0, L);
#pragma unroll
  for (int u = 0; u < GD; ++u) {
    gload(ra[u], rb[u], L);
    advance_load(L);
  }
  CItem C = citem_at(0);
  gfix(ra[0], rb[0], C.kt, C.kt1, C.K);
  colsum_acc(ra[0], C.p->colsum != nullptr && C.it.tn == 0);
  lstore(0, ra[0], rb[0]);
  __syncthreads();
  int c_ord = 0;
  int s = 0;  // global step counter of this workgroup (LDS buffer parity)

  auto multiply = [&](auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    gload(ra[u], rb[u], L);
    advance_load(L);
    const char* la = smem + (s & 1) * 32768;
    const char* lb = la + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<T, true>(la, wm * 64 + i * 16, ks, x, y);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<T, true>(lb, wn * 64 + j * 16, ks, x, y);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma<T>(acc[i][j], fb[j], fa[i]);
    }
  };
  // land the step after (C.kt) in the other LDS buffer; `nx` describes that step's item
  auto land_next = [&](auto uc, const CItem& nx, int nkt) __attribute__((always_inline)) {
    constexpr int un = (decltype(uc)::value + 1) % GD;
    gfix(ra[un], rb[un], nkt, nx.kt1, nx.K);
    colsum_acc(ra[un], nx.p->colsum != nullptr && nx.it.tn == 0 && nkt < nx.kt1);
    lstore((s & 1) ^ 1, ra[un], rb[un]);
    __syncthreads();
    ++s;
  };

  for (;;) {
    // GD steps of the current item (items are padded to multiples of GD steps, so they end on iteration boundaries)
    static_assert(GD == 2, "the iteration below is written for two register sets");
    multiply(std::integral_constant<int, 0>{});
    land_next(std::integral_constant<int, 0>{}, C, C.kt + 1);
    ++C.kt;
    multiply(std::integral_constant<int, 1>{});
    if (C.kt + 1 == C.kt_end) {  // workgroup-uniform: last step of the item
      epilogue(C);
      if (c_ord + 1 == my_items) break;
      zero_acc();
      ++c_ord;
      C = citem_at(c_ord);
      land_next(std::integral_constant<int, 1>{}, C, C.kt);
    } else {
      land_next(std::integral_constant<int, 1>{}, C, C.kt + 1);
      ++C.kt;
    }
  }
}

// dW[m][n] += alpha * sum_split partial[tile][split]: one workgroup per (output tile, accumulator fragment)
struct TileRef {
  int32_t prob, tm, tn, pad;
};
__global__ __launch_bounds__(256) void wgrad_grouped_reduce_kernel(const s2t_wgrad_problem* __restrict__ probs,
                                                                   const TileRef* __restrict__ tiles,
                                                                   const float* __restrict__ ws) {
  const TileRef t = tiles[blockIdx.x];
  const s2t_wgrad_problem* p = probs + t.prob;
  const int f = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, x = lane & 15, y = lane >> 4;
  const int i = f >> 2, j = f & 3;
  const int m = t.tm * BM + wm * 64 + i * 16 + x;
  const int n = t.tn * BN + wn * 64 + j * 16 + 4 * y;
  const int tile = t.tm * p->tiles_n + t.tn;
  // problems that accumulate into the SAME dW (tied weights) are chained through `next` and summed by this one
  // workgroup, so that every dW element has exactly one (non-atomic) writer
  f32x4 total = {0.f, 0.f, 0.f, 0.f};
  for (const s2t_wgrad_problem* q = p;;) {
    const float* src = ws + q->ws_base + (int64_t)tile * q->nsplit * (BM * BN) + (f * 256 + tid) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int s = 0;
    const int ns = q->nsplit;
    for (; s + 4 <= ns; s += 4) {
      s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)s * (BM * BN));
      s1 += *reinterpret_cast<const f32x4*>(src + (int64_t)(s + 1) * (BM * BN));
      s2 += *reinterpret_cast<const f32x4*>(src + (int64_t)(s + 2) * (BM * BN));
      s3 += *reinterpret_cast<const f32x4*>(src + (int64_t)(s + 3) * (BM * BN));
    }
    for (; s < ns; ++s) s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)s * (BM * BN));
    total += ((s0 + s1) + (s2 + s3)) * q->alpha;
    if (q->next < 0) break;
    q = probs + q->next;
  }
  if (m < p->M && n < p->N) {
    float* dst = p->C + (int64_t)m * p->ldc + n;
    if (n + 3 < p->N && (p->ldc & 3) == 0 && (((uintptr_t)p->C) & 15) == 0) {
      f32x4 c = *reinterpret_cast<f32x4*>(dst);
      c += total;
      *reinterpret_cast<f32x4*>(dst) = c;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p->N) dst[r] += total[r];
    }
  }
}

// ===============================================================================================================
// The same work list on 256 x 256 output tiles fed by LDS-DMA (s2t_wgrad_grouped256).
//
// Why a second kernel: at 128 x 128 per workgroup the register-staged loop above moves 32 KiB per 64-row K-step per
// workgroup through the vector-memory path (64 B/clk/CU with two workgroups per CU = all of it) and writes it to LDS with
// ds_write_b128 (~79 B/clk/CU), both at the rate the MFMAs need it: three saturated pipes, 16-22 % of the MFMA peak.
// Here one workgroup of 8 waves owns a 256 x 256 tile (wave: 128 dW rows x 64 dW columns, 8 x 4 accumulator tiles =
// 128 VGPRs): half the operand bytes per flop, no staging registers and no LDS stores — both operands are K-major
// ([k][column] rows, exactly the LDS image the transposing ds_read_b64_tr_b16 fragment reads want), so buffer_load ... lds
// drops 512-byte k-rows of a tile straight into place (the XOR swizzle of the 16-byte chunks rides on the per-lane SOURCE
// address).  Four 32 KiB stages of 32 k-rows each (A part 16 KiB | B part 16 KiB): the DMA runs three steps ahead,
// counted vmcnt + one barrier per step.  Bias gradients: column sums of the A part read back from LDS (two 16-byte reads
// per thread and step, only for tile column 0).  Operand rows beyond K and everything beyond the end of an operand read as
// zero through the buffer descriptor's bounds; columns beyond M / N only feed accumulators that are never stored.
// Requirements (checked by the host): bf16, lda % 8 == 0, ldb % 8 == 0, 16-byte aligned operands spanning < 2 GiB.
constexpr int T2 = 256;
constexpr int BK2 = 32;
#ifndef S2T_WG_DBG
#define S2T_WG_DBG 0
#endif
#ifndef S2T_WG_PRIO
#define S2T_WG_PRIO 0  // s_setprio 1 around each MFMA group
#endif
#ifndef S2T_WG_NST
#define S2T_WG_NST 4
#endif
constexpr int NST = S2T_WG_NST;  // LDS stages of 32 KiB (5 = all 160 KiB measured no faster than 4)
constexpr int PART2 = BK2 * T2 * 2;   // 16 KiB: one operand's 32 k-rows of 512 B
constexpr int STAGE2 = 2 * PART2;     // 32 KiB

__global__ __launch_bounds__(512) void wgrad256_kernel(const s2t_wgrad_problem* __restrict__ probs,
                                                          const Item* __restrict__ items, int n_items,
                                                          float* __restrict__ ws, int nt_mode, int stagger) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int x = lane & 15, y = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  const int G = gridDim.x;
  int w = blockIdx.x;
  if ((G & 7) == 0) w = (w & 7) * (G >> 3) + (w >> 3);  // XCD-contiguous item ranges
  if (w >= n_items) return;
  const int my_items = (n_items - w + G - 1) / G;

  // ---- fragment read offsets inside a part (bytes).  16-column block blk of the tile, lane (x, y): k-rows 8y + 4 half + q
  // (q = x >> 2), columns 16 blk + 4 (x & 3) .. +4; chunk (2 blk + ((x&3) >> 1)) ^ kswz(row), kswz depends on q and y & 1
  // only, so the second half is the first + 4 rows (2048 B).
  uint32_t offA[8], offB[4];
  {
    const int q = x >> 2, pb = x & 3;
    const int key = 2 * (q | ((y & 1) << 2));
    const uint32_t base = (uint32_t)((8 * y + q) * 512 + 16 * (pb >> 1) + 8 * (pb & 1));
#pragma unroll
    for (int i = 0; i < 8; ++i) offA[i] = base + 16u * (uint32_t)((16 * wm + 2 * i) ^ key);
#pragma unroll
    for (int j = 0; j < 4; ++j) offB[j] = base + 16u * (uint32_t)((8 * wn + 2 * j) ^ key);
  }
  // ---- DMA plan: instruction i of wave w covers k-rows 4w + 2i, +1 of a part (lane l: row + (l >> 5), slot l & 31 holds
  // chunk (l & 31) ^ kswz(row)); kswz(4w + 2i + hi) = 2 * ((2i + hi) | (((w >> 1) & 1) << 2))
  const int hi = lane >> 5, sl = lane & 31;
  const int key0 = 2 * (hi | (((wave >> 1) & 1) << 2));
  const uint32_t cp0 = 16u * (uint32_t)(sl ^ key0), cp1 = cp0 ^ 64u;  // i = 1: key ^ 4
  const int r0 = 4 * wave + hi;
  const uint32_t ldsw = lds0 + (uint32_t)(4 * wave) * 512u;
  // ---- bias-gradient read: thread (c = tid & 31, rr = tid >> 5) sums chunk c of k-rows rr and rr + 16
  const uint32_t offC = (uint32_t)((tid >> 5) * 512 + 16 * ((tid & 31) ^ kswz(tid >> 5)));

  f32x4 acc[8][4];
  float csum[8];
  int sc = 0;  // running step counter (stage rotation)
  for (int ord = 0; ord < my_items; ++ord) {
    const Item it = items[w + ord * G];
    const s2t_wgrad_problem* p = probs + it.prob;
    // k_live (packed batch: the rows are the reduction dimension): K is read here, and the item's share of the K-steps is
    // re-cut from it in nsplit balanced parts — every work item of the problem shrinks with the fill of the batch
    const int K = p->k_live ? min(p->K, __builtin_amdgcn_readfirstlane(p->k_live[0])) : p->K;  // (workgroup-uniform: keep it scalar)
    const uint32_t lda2 = (uint32_t)(p->lda * 2), ldb2 = (uint32_t)(p->ldb * 2);
    const int ktiles = (K + BK2 - 1) / BK2;
    const int kt0 = p->k_live ? (int)((int64_t)it.split * ktiles / p->nsplit) : it.split * p->ksteps;
    const int nst = p->k_live ? (int)((int64_t)(it.split + 1) * ktiles / p->nsplit) - kt0 : min(ktiles, kt0 + p->ksteps) - kt0;
    const bool do_cs = p->colsum != nullptr && it.tn == 0;
    // descriptors: base moved to the tile's first column, bounds at the end of the operand (K rows)
    const int64_t a_bytes = (int64_t)K * lda2 - (int64_t)it.tm * (T2 * 2);
    const int64_t b_bytes = (int64_t)K * ldb2 - (int64_t)it.tn * (T2 * 2);
    const i32x4 srdA = make_srd(reinterpret_cast<const char*>(p->A) + (int64_t)it.tm * (T2 * 2), (uint32_t)(a_bytes > 0 ? a_bytes : 0));
    const i32x4 srdB = make_srd(reinterpret_cast<const char*>(p->B) + (int64_t)it.tn * (T2 * 2), (uint32_t)(b_bytes > 0 ? b_bytes : 0));
    const uint32_t vA0 = (uint32_t)r0 * lda2 + cp0, vA1 = (uint32_t)(r0 + 2) * lda2 + cp1;
    const uint32_t vB0 = (uint32_t)r0 * ldb2 + cp0, vB1 = (uint32_t)(r0 + 2) * ldb2 + cp1;
    // an operand tile that no other tile of this weight reads is fetched non-temporally: its lines would only push the
    // SHARED operand's (xn of a W1 gradient: read by all eight 256-column tiles) out of the XCD's L2
    const bool ntA = nt_mode == 2 || (nt_mode == 1 && p->tiles_n == 1);
    const bool ntB = nt_mode == 2 || (nt_mode == 1 && p->M <= T2);
    auto issue = [&](int kt, int st) __attribute__((always_inline)) {
#if S2T_WG_DBG & 1
      return;  // experiment: no operand traffic (the products run on whatever the stages hold)
#endif
      const uint32_t la = ldsw + (uint32_t)st * STAGE2;
      const uint32_t sa = (uint32_t)kt * (BK2 * lda2), sb = (uint32_t)kt * (BK2 * ldb2);
      if (ntA) {
        dma16_nt(la, vA0, srdA, sa);
        dma16_nt(la + 1024, vA1, srdA, sa);
      } else {
        dma16(la, vA0, srdA, sa);
        dma16(la + 1024, vA1, srdA, sa);
      }
      if (ntB) {
        dma16_nt(la + PART2, vB0, srdB, sb);
        dma16_nt(la + PART2 + 1024, vB1, srdB, sb);
      } else {
        dma16(la + PART2, vB0, srdB, sb);
        dma16(la + PART2 + 1024, vB1, srdB, sb);
      }
    };
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    // every wave is done with the previous item's stages (and its LDS scratch) before new rows land in them
    __syncthreads();
    // the tiles of one weight walk their K-range from different starting steps (wrapping around)
    const int rot = nst > 0 ? ((it.tm + it.tn) * stagger) % nst : 0;
    auto kstep = [&](int t) __attribute__((always_inline)) { const int r = t + rot; return kt0 + (r >= nst ? r - nst : r); };
    // Pipeline (round 3): the fragments of step t+1 are read from LDS BETWEEN the MFMAs of step t (hipcc would sink every
    // read to just in front of its first use: 24 transposed reads of exposed latency per step, every wave in the same phase
    // behind the step's barrier — 3 300 cycles per step against 1 024 of MFMA).  Ring of NST stages: while step t computes
    // from registers, stage t+1 is being read and stages t+2 .. t+NST are in flight (stage t+NST reuses the slot of stage t,
    // whose reads were retired before this step's barrier).
    auto tr8 = [&](const char* a) __attribute__((always_inline)) -> Frag {
      typedef __attribute__((address_space(3))) s16x4* lptr;
      const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a)));
      const uint2 hi2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a + 2048)));
      Frag f;
      f.v = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
      return f;
    };
    auto colsum = [&](const char* pa) __attribute__((always_inline)) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const uint4 v = *reinterpret_cast<const uint4*>(pa + offC + h2 * 8192);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          csum[2 * q4] += __uint_as_float(w4[q4] << 16);
          csum[2 * q4 + 1] += __uint_as_float(w4[q4] & 0xffff0000u);
        }
      }
    };
    // registers: B fragments double buffered (all four feed every MFMA group of a step); A fragments ROLL — the pair a group
    // has just used is refilled with the next step's values behind it — except the last pair, which alternates with a spare
    // pair so that no read is issued right in front of the step's barrier
    Frag fA[10], fb[2][4];
    // wait until at most `stages` DMA groups (4 pieces each) of this wave are still in flight, then the workgroup barrier
    auto wait_bar = [&](int stages, bool lgkm) __attribute__((always_inline)) {
      // (the builtin, not inline assembly: hipcc's own wait insertion then knows that no LDS read is pending behind it —
      // left unaware it puts a full lgkmcnt(0) in front of every MFMA group, i.e. behind the prefetch reads just issued)
      (void)lgkm;
      if (stages >= 3) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
      else if (stages == 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      else if (stages == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int u = 0; u < NST - 1; ++u)
      if (u < nst) issue(kstep(u), (sc + u) % NST);
    if (nst > 0) {
      wait_bar(min(nst - 1, NST - 2), false);
      if (NST - 1 < nst) issue(kstep(NST - 1), (sc + NST - 1) % NST);
      const char* pa = smem + (sc % NST) * STAGE2;
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[0][j] = tr8(pa + PART2 + offB[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) fA[i] = tr8(pa + offA[i]);
      if (do_cs) colsum(pa);
    }
    auto step = [&](int t, auto cur_c) __attribute__((always_inline)) {
      constexpr int CUR = decltype(cur_c)::value, NXT = CUR ^ 1;
      const bool has_next = t + 1 < nst;
      const char* pn = smem + ((sc + t + 1) % NST) * STAGE2;
      // Every LDS read of the previous step is retired here, on every path (the reads of stage t were issued during it: its slot
      // is refilled behind the barrier below).  The builtin, not inline assembly: hipcc's own wait insertion then knows that
      // nothing is pending — left unaware, or with a path that skips this wait, it puts a full lgkmcnt(0) in front of every MFMA
      // group, i.e. behind the prefetch reads just issued.
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      if (has_next) {
        // steps behind t+1 issued so far: at most NST - 2 (lgkmcnt(0): the reads of stage t, issued during the previous
        // step, are retired before the barrier behind which its slot is refilled)
        wait_bar(min(nst - 2 - t, NST - 2), true);
        if (t + NST < nst) issue(kstep(t + NST), (sc + t + NST) % NST);
      }
      // A fragment i of the CURRENT step lives in fA[i] for i < 6 and in fA[6 + 2 CUR + (i - 6)] for the last pair
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        // (unconditional: behind the last step these read a stage nobody needs — a branch here costs the precise waits)
        fb[NXT][gq] = tr8(pn + PART2 + offB[gq]);
        if (gq > 0) {
          fA[2 * gq - 2] = tr8(pn + offA[2 * gq - 2]);
          fA[2 * gq - 1] = tr8(pn + offA[2 * gq - 1]);
        }
        if (gq == 3) {
          fA[6 + 2 * NXT] = tr8(pn + offA[6]);
          fA[7 + 2 * NXT] = tr8(pn + offA[7]);
        }
        __builtin_amdgcn_sched_barrier(0);
#if S2T_WG_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int i = 2 * gq; i < 2 * gq + 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#if S2T_WG_DBG & 2
            asm volatile("" ::"v"(fb[CUR][j].v.x), "v"(fb[CUR][j].v.w), "v"(fA[i < 6 ? i : i + 2 * CUR].v.x),
                         "v"(fA[i < 6 ? i : i + 2 * CUR].v.w));  // experiment: no MFMAs, the reads stay
#else
            mma<T>(acc[i][j], fb[CUR][j], fA[i < 6 ? i : i + 2 * CUR]);
#endif
          }
#if S2T_WG_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      if (has_next && do_cs) colsum(pn);
    };
    for (int t = 0; t < nst; t += 2) {
      step(t, std::integral_constant<int, 0>{});
      if (t + 1 < nst) step(t + 1, std::integral_constant<int, 1>{});
    }
    sc += nst;
    // ---- partial tile -> workspace in register-native order; bias-gradient partial -> atomics
    const int tile = it.tm * p->tiles_n + it.tn;
    float* wt = ws + p->ws_base + ((int64_t)tile * p->nsplit + it.split) * (int64_t)(T2 * T2);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(wt + ((i * 4 + j) * 512 + tid) * 4) = acc[i][j];
    if (do_cs) {  // workgroup-uniform
      __syncthreads();  // all fragment reads of the last stage are done: the stages are scratch now
      float* lc = reinterpret_cast<float*>(smem);  // [16 row groups][256 columns]
#pragma unroll
      for (int e = 0; e < 8; ++e) lc[(tid >> 5) * 256 + 8 * (tid & 31) + e] = csum[e];
      __syncthreads();
      if (tid < 256) {
        float sum = 0.f;
#pragma unroll
        for (int gI = 0; gI < 16; ++gI) sum += lc[gI * 256 + tid];
        const int m = it.tm * T2 + tid;
        if (m < p->M) atomicAdd(p->colsum + m, p->alpha * sum);
      }
    }
  }
}

// dW[m][n] += alpha * sum_split partial[tile][split] for the 256 x 256 tiles: one workgroup per (tile, accumulator fragment)
__global__ __launch_bounds__(512) void wgrad256_reduce_kernel(const s2t_wgrad_problem* __restrict__ probs,
                                                              const TileRef* __restrict__ tiles,
                                                              const float* __restrict__ ws) {
  const TileRef t = tiles[blockIdx.x];
  const s2t_wgrad_problem* p = probs + t.prob;
  const int f = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, x = lane & 15, y = lane >> 4;
  const int i = f >> 2, j = f & 3;
  const int m = t.tm * T2 + wm * 128 + i * 16 + x;
  const int n = t.tn * T2 + wn * 64 + j * 16 + 4 * y;
  const int tile = t.tm * p->tiles_n + t.tn;
  f32x4 total = {0.f, 0.f, 0.f, 0.f};
  for (const s2t_wgrad_problem* q = p;;) {  // tied weights: the chain's problems are summed by this one workgroup
    const float* src = ws + q->ws_base + (int64_t)tile * q->nsplit * (T2 * T2) + (f * 512 + tid) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int s = 0;
    const int ns = q->nsplit;
    for (; s + 2 <= ns; s += 2) {
      s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)s * (T2 * T2));
      s1 += *reinterpret_cast<const f32x4*>(src + (int64_t)(s + 1) * (T2 * T2));
    }
    for (; s < ns; ++s) s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)s * (T2 * T2));
    total += (s0 + s1) * q->alpha;
    if (q->next < 0) break;
    q = probs + q->next;
  }
  if (m < p->M && n < p->N) {
    float* dst = p->C + (int64_t)m * p->ldc + n;
    if (n + 3 < p->N && (p->ldc & 3) == 0 && (((uintptr_t)p->C) & 15) == 0) {
      f32x4 c = *reinterpret_cast<f32x4*>(dst);
      c += total;
      *reinterpret_cast<f32x4*>(dst) = c;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p->N) dst[r] += total[r];
    }
  }
}

}  // namespace

extern "C" int s2t_wgrad_grouped256(const s2t_wgrad_problem* problems_dev, int n_problems, const int32_t* items_dev,
                                    int n_items, const int32_t* tiles_dev, int n_tiles, float* ws, void* stream) {
  if (!problems_dev || !items_dev || !tiles_dev || !ws || n_problems <= 0 || n_items <= 0 || n_tiles <= 0) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int slots = s2t_device_cu_count();  // 128 KiB of LDS per workgroup: one per CU
  dim3 grid(n_items < slots ? n_items : slots), block(512);
  // (experiment switches, read once: a getenv per launch is host time on the critical path of an eager step)
  static const int nt_mode = [] { const char* e = getenv("S2T_WG_NT"); return e ? atoi(e) : 1; }();
  static const int stagger = [] { const char* e = getenv("S2T_WG_STAG"); return e ? atoi(e) : 0; }();
  hipLaunchKernelGGL(wgrad256_kernel, grid, block, 0, s, problems_dev, reinterpret_cast<const Item*>(items_dev), n_items, ws,
                     nt_mode, stagger);
  hipLaunchKernelGGL(wgrad256_reduce_kernel, dim3(n_tiles, 32), dim3(512), 0, s, problems_dev,
                     reinterpret_cast<const TileRef*>(tiles_dev), ws);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_wgrad_grouped(const s2t_wgrad_problem* problems_dev, int n_problems, const int32_t* items_dev,
                                 int n_items, const int32_t* tiles_dev, int n_tiles, float* ws, int any_k_tail,
                                 void* stream) {
  if (!problems_dev || !items_dev || !tiles_dev || !ws || n_problems <= 0 || n_items <= 0 || n_tiles <= 0) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int slots = 2 * s2t_device_cu_count();
  dim3 grid(n_items < slots ? n_items : slots), block(256);
  if (any_k_tail)
    hipLaunchKernelGGL(wgrad_grouped_kernel<true>, grid, block, 0, s, problems_dev, reinterpret_cast<const Item*>(items_dev),
                       n_items, ws);
  else
    hipLaunchKernelGGL(wgrad_grouped_kernel<false>, grid, block, 0, s, problems_dev,
                       reinterpret_cast<const Item*>(items_dev), n_items, ws);
  hipLaunchKernelGGL(wgrad_grouped_reduce_kernel, dim3(n_tiles, 16), dim3(256), 0, s, problems_dev,
                     reinterpret_cast<const TileRef*>(tiles_dev), ws);
  return S2T_LAUNCH_CHECK();
}
