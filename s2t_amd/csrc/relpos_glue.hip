// Relative-position attention backward, everything behind the skewed score gradient dbd in ONE pass over it (round 3):
//   dqv[b,i,h,:]  = sum_n dbd[h][b][i][n] * p[n][h*64 : h*64+64]          the (Q + pos_bias_v) branch of dQ
//   dq[b,i,h,:]  += dqv                                                   (bf16, in place)
//   dpos_u[h*64+c] += sum_{b,i} dq_before[b,i,h,c],  dpos_v[h*64+c] += sum_{b,i} dqv[b,i,h,c]
//   dp_part[b][n][h*64+c] = sum_i dbd[h][b][i][n] * qv[b,i,h*64+c]        this utterance's share of the position-table gradient
//   (espnet_multihead_attention.py:313-337 backward: matrix_bd = (q + pos_bias_v) p^T, rel_shift; linear_pos).
// Replaces s2t_relpos_dqv (one wave per 64 queries reading 57 MB of dbd as strided fragments: 21 us per layer), the batched
// split-K GEMM over K = B * T that read dbd a second time (64 MB, 24 us) and its reduce: dbd is read ONCE, through LDS, by
// one workgroup per (utterance, head) that keeps the head's 2T - 1 projected position rows resident (64 KiB).
// Per 32-query tile of the slab (dbd rows are 2T - 1 <= 512 wide: a 32 x 512 bf16 tile = 32 KiB in LDS, double buffered, 16-byte
// chunk j of row q at q*1024 + ((j ^ (q & 15)) << 4)):
//   (1) dqv^T[c][q] = sum_n P^T[c][n] dbd^T[n][q]   A = position image read column-wise (ds_read_b64_tr_b16), B = tile rows
//   (2) dp^T[c][n] += sum_q qv^T[c][q] dbd[q][n]    A = the qv tile read column-wise, B = the dbd tile read column-wise
// 128 MFMAs (16x16x32) each per tile; wave w owns one (channel tile, query tile) result of (1) and position tiles 4w .. 4w+3
// of (2) (64 accumulator registers, kept across the query tiles).  The partial dp leaves as bf16 rows of 128 bytes; a second
// kernel sums the B partials of a head in fp32 (s2t_relpos_glue's dp output, what the linear_pos weight gradient reads).
// Round 5: sequences beyond 256 frames (the PDS stages: T' = 1004 / 502).  The 2T - 1 position rows no longer fit one image, so
// the workgroup walks them in CHUNKS of 512: per chunk it stages that part of the position image, visits only the query tiles
// whose band [T-1-i, T-1-i+len) meets the chunk (and only the chunk's 512 columns of their dbd rows: every (tile, chunk) piece
// of the band is still read exactly once), adds the chunk's share of dqv into dq (the same lane updates the same dq elements in
// every chunk: program order), and stores the chunk's rows of the partial table.  The pos_bias_u column sum takes a row's dq
// value in the chunk that holds the low end of its band (before any update).  T' <= 256 is one chunk: the round-3 kernel.
#include "common.h"

namespace {

constexpr int DK = 64;
#ifndef S2T_GLUE_DBG
#define S2T_GLUE_DBG 0  // experiment builds (tools/dbg_variant.sh): 1 no product (1), 2 no product (2), 4 one global tile load only,
#endif                  // 8 no dq update, 16 no partial-table store
constexpr int NP = 512;            // position rows / dbd columns held (2T - 1 <= 511)
constexpr int TQ = 32;             // queries per tile
constexpr int L_P = 0;             // [NP][128 B]
constexpr int L_D = NP * 128;      // two dbd tiles [TQ][1024 B]
constexpr int L_Q = L_D + 2 * TQ * 1024;  // two qv tiles [TQ][128 B]
constexpr int L_BYTES = L_Q + 2 * TQ * 128 + 2 * 8 * 64 * 4;  // + [2][8 waves][64] floats of column sums

typedef short s16x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ uint2 tr64(const char* a) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a)));
}

// Swizzle keys chosen for the NATURAL-order transposed reads of this kernel (a 32-lane half of ds_read_b64_tr_b16 touches rows
// {r, r+1, r+2, r+3, r+8, .. r+11} x two adjacent 16-byte chunks: sixteen different bank groups with these keys; the usual
// (r & 7) key puts rows r and r + 8 on the same one).  key128: images with 128-byte rows (two rows per 256-byte bank row, the
// row's parity picks the half); key1024: the dbd tile (every row starts a bank row; also conflict-free for its ds_read_b128
// row reads, whose 16-lane groups are rows {0-3, 12-15} at one chunk and {4-11} at the next).
__device__ __forceinline__ int key128(int r) { return 2 * (((r >> 1) & 1) | (((r >> 3) & 1) << 1)); }
__device__ __forceinline__ int key1024(int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); }

// column-wise fragment of an image with 128-byte rows (chunk c of row r at r*128 + ((c ^ key128(r)) << 4)): operand rows =
// image columns 16 cblk + x, k = image rows row0 + 32 s + 8 y + j (natural order)
__device__ __forceinline__ bf16x8 cols128(const char* img, int row0, int cblk, int s, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = row0 + 32 * s + 8 * y + 4 * h + qq;
    const uint2 t = tr64(img + R * 128 + (((2 * cblk + (p >> 1)) ^ key128(R)) << 4) + (p & 1) * 8);
    w[2 * h] = t.x;
    w[2 * h + 1] = t.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}
// the same of the dbd tile (1024-byte rows, chunk j of row q at q*1024 + ((j ^ key1024(q)) << 4)): operand columns = tile
// columns 16 nblk + x, k = tile rows 32 s + 8 y + j
__device__ __forceinline__ bf16x8 cols1024(const char* img, int nblk, int s, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = 32 * s + 8 * y + 4 * h + qq;
    const uint2 t = tr64(img + R * 1024 + (((2 * nblk + (p >> 1)) ^ key1024(R)) << 4) + (p & 1) * 8);
    w[2 * h] = t.x;
    w[2 * h + 1] = t.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}

struct GlueArgs {
  const bf16_t* dbd;   // [H][B][Tq][ldb]
  int64_t ldb;
  const bf16_t* pos_p; // [2Tq-1][p_sr], head h at column h*64
  int64_t p_sr;
  const bf16_t* qv;    // [B*Tq][H*64]
  bf16_t* dq;          // rows b*dq_sb + i*dq_sr, head h at column h*64
  int64_t dq_sb, dq_sr;
  float *du, *dv;      // column-sum targets (replicated)
  int replicas;
  int64_t replica_stride;
  bf16_t* dq_lo;       // optional [B*Tq][H*64] scratch (Tq > 256): what the bf16 rounding of a row's RUNNING dq sum dropped,
                       // carried from one position chunk to the next so that dq is rounded once, as in the one-chunk form
  bf16_t* dp_part;     // [B][2Tq-1][H*64]
  int B, H, Tq;
  const int32_t* cu;   // packed batch: rows of utterance b in dq = cu[b] .. cu[b+1] (dbd / qv keep their padded row strides)
};

// the first `rem` (0 < rem < 8) bf16 elements of a 16-byte piece, the rest zero
__device__ __forceinline__ uint4 keep_first(uint4 v, int rem) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = rem - 2 * t;
    w[t] = r >= 2 ? w[t] : (r == 1 ? (w[t] & 0xffffu) : 0u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// MULTI: more than one chunk of position rows (2Tq - 1 > 512); false compiles the chunk walk away (the round-3 kernel)
template <bool MULTI>
__global__ __launch_bounds__(512) void relpos_glue_kernel(const GlueArgs a_in) {
  __shared__ __attribute__((aligned(16))) char lds[L_BYTES];
  GlueArgs a = a_in;
  // packed batch: this utterance's rows.  nq rows are walked; of a dbd row only the columns its dQ kernel wrote in THIS pass
  // are taken — n < Tq-1-i+nq: the slab is shared by every batch, and what lies beyond may be the band of a longer utterance
  // of an earlier one.
  int nq = a.Tq;
  if (a.cu) {
    const int bb = (int)blockIdx.x / a.H, r0 = a.cu[bb];
    nq = a.cu[bb + 1] - r0;
    a.dq += (int64_t)r0 * a.dq_sr - (int64_t)bb * a.dq_sb;
  }
  char* lp = lds + L_P;
  char* ld = lds + L_D;
  char* lq = lds + L_Q;
  float* red = reinterpret_cast<float*>(lds + L_Q + 64 * 128);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, y = lane >> 4;
  const int z = blockIdx.x;
  const int b = z / a.H, h = z % a.H;
  const int npos = 2 * a.Tq - 1;
  const int d = a.H * DK;

  const bf16_t* slab = a.dbd + (((int64_t)h * a.B + b) * a.Tq) * a.ldb;
  const int nchunk = (int)(a.ldb / 8);  // 16-byte chunks per dbd row (<= 64)

  float su[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
  const int ct1 = w & 3, qh = w >> 2;  // product (1): channel tile ct1, query tile qh of the 32-query tile
  const int nck = MULTI ? (npos + NP - 1) / NP : 1;  // chunks of position rows / dbd columns
  bf16_t* out = a.dp_part + ((int64_t)b * npos) * d + h * DK;

  for (int ck = 0; ck < nck; ++ck) {
    const int n0 = MULTI ? ck * NP : 0;
    // query rows whose band [Tq-1-i, Tq-1-i+nq) meets the chunk's columns [n0, n0 + NP); whole 32-query tiles of them
    const int i_lo = MULTI ? max(0, a.Tq - NP - n0) : 0, i_hi = MULTI ? min(nq - 1, a.Tq + nq - 2 - n0) : nq - 1;
    const int q_lo = i_lo & ~(TQ - 1);
    const int jc0 = n0 / 8;
    // the tile ranges of the neighbouring chunks: a tile's FIRST visit starts its remainder at zero, its LAST one drops it
    const int p_lo = max(0, a.Tq - NP - (n0 - NP)) & ~(TQ - 1), p_hi = ck > 0 ? min(nq - 1, a.Tq + nq - 2 - (n0 - NP)) : -1;
    const int x_lo = max(0, a.Tq - NP - (n0 + NP)) & ~(TQ - 1), x_hi = ck + 1 < nck ? min(nq - 1, a.Tq + nq - 2 - (n0 + NP)) : -1;
    f32x4 dp[4][4];  // [position tile 4w + nt of the chunk][channel tile]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) dp[nt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // 32-query tiles, double buffered: the next tile's rows travel global -> registers during this tile's products and are
    // written to the other LDS buffer behind them (one barrier per tile)
    uint4 tr[4], tqv = make_uint4(0, 0, 0, 0);
    uint2 old_n = make_uint2(0, 0);  // the dq values this lane updates in the NEXT tile (a global round trip per tile otherwise)
    auto tile_load = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = tid + 512 * u;
        const int q = min(q0 + (c >> 6), nq - 1), jc = min(jc0 + (c & 63), nchunk - 1);
        tr[u] = *reinterpret_cast<const uint4*>(slab + (int64_t)q * a.ldb + jc * 8);
      }
      if (tid < 256)
        tqv = *reinterpret_cast<const uint4*>(a.qv + ((int64_t)b * a.Tq + min(q0 + (tid >> 3), nq - 1)) * d + h * DK + (tid & 7) * 8);
      old_n = *reinterpret_cast<const uint2*>(a.dq + (int64_t)b * a.dq_sb + (int64_t)min(q0 + 16 * qh + x, nq - 1) * a.dq_sr + h * DK +
                                              16 * ct1 + 4 * y);
    };
    auto tile_store = [&](int q0, int buf) __attribute__((always_inline)) {
      char* ldb_ = ld + buf * (TQ * 1024);
      char* lqb = lq + buf * (TQ * 128);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = tid + 512 * u;
        const int q = c >> 6, jc = c & 63;
        const bool ok = q0 + q < nq && jc0 + jc < nchunk;
        uint4 v = tr[u];
        if (a.cu) {  // (workgroup-uniform)
          const int rem = a.Tq - 1 - (q0 + q) + nq - (n0 + 8 * jc);  // columns of this piece below the row's written band end
          if (rem < 8) v = rem > 0 ? keep_first(v, rem) : make_uint4(0, 0, 0, 0);
        }
        *reinterpret_cast<uint4*>(ldb_ + q * 1024 + ((jc ^ key1024(q)) << 4)) = ok ? v : make_uint4(0, 0, 0, 0);
      }
      if (tid < 256) {
        const int qr = tid >> 3;
        *reinterpret_cast<uint4*>(lqb + qr * 128 + (((tid & 7) ^ key128(qr)) << 4)) = q0 + qr < nq ? tqv : make_uint4(0, 0, 0, 0);
      }
    };
    const bool any = i_hi >= i_lo;
    if (any) tile_load(q_lo);  // (in flight together with the position rows below: one global round trip for both)
    // ---- the chunk's projected position rows of the head (rows >= 2T-1 zero)
    if (any) {
      const bf16_t* pp = a.pos_p + h * DK;
      uint4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = tid + 512 * u;
        const int n = min(n0 + (c >> 3), npos - 1);
        t[u] = *reinterpret_cast<const uint4*>(pp + (int64_t)n * a.p_sr + (c & 7) * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = tid + 512 * u;
        const int n = c >> 3, ch = c & 7;
        *reinterpret_cast<uint4*>(lp + n * 128 + ((ch ^ key128(n)) << 4)) = n0 + n < npos ? t[u] : make_uint4(0, 0, 0, 0);
      }
      tile_store(q_lo, 0);
    }
    __syncthreads();  // (also orders the position image)
    int buf = 0;
    for (int q0 = q_lo; any && q0 <= i_hi; q0 += TQ, buf ^= 1) {
      const bool more = q0 + TQ <= i_hi;
      const uint2 old = old_n;
      const bool seen = MULTI && a.dq_lo && q0 >= p_lo && q0 <= p_hi;    // an earlier chunk visited this tile and left its remainder
      const bool again = MULTI && a.dq_lo && q0 >= x_lo && q0 <= x_hi;   // the next chunk visits it too
      if (more && !((S2T_GLUE_DBG & 4) && q0 > 0)) tile_load(q0 + TQ);
      const char* ldc = ld + buf * (TQ * 1024);
      const char* lqc = lq + buf * (TQ * 128);
      // ---- (1) dqv^T[c][q] over K = the chunk's 512 position columns
      // (hipcc sinks every fragment read to just in front of its MFMA — one exposed LDS round trip per MFMA of this dependent
      // chain: the reads run one group of four k-steps ahead by hand, two accumulators halve the chain; the fragments of
      // product (2) are read behind the last group's)
      f32x4 acc, acc1 = {0.f, 0.f, 0.f, 0.f};
      acc = acc1;
      const int i = q0 + 16 * qh + x;
      bf16x8 qa[4], dbn[4];
      {
        const int q = 16 * qh + x;
        bf16x8 PA[2][4], DB[2][4];
        auto rd1 = [&](int g, int sl) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ks = 4 * g + j;
            PA[sl][j] = cols128(lp, 0, ct1, ks, x, y);
            DB[sl][j] = as_frag(*reinterpret_cast<const uint4*>(ldc + q * 1024 + (((4 * ks + y) ^ key1024(q)) << 4)));
          }
        };
        rd1(0, 0);
#pragma unroll
        for (int g = 0; g < NP / 128; ++g) {
          if (g + 1 < NP / 128) {
            rd1(g + 1, (g + 1) & 1);
          } else {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) qa[ct] = cols128(lqc, 0, ct, 0, x, y);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) dbn[nt] = cols1024(ldc, 4 * w + nt, 0, x, y);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            if (S2T_GLUE_DBG & 1) continue;
            acc = mfma16(PA[g & 1][j], DB[g & 1][j], acc);
            acc1 = mfma16(PA[g & 1][j + 1], DB[g & 1][j + 1], acc1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        acc += acc1;
      }
      if (i < nq && !(S2T_GLUE_DBG & 8)) {
        const float o4[4] = {__uint_as_float(old.x << 16), __uint_as_float(old.x & 0xffff0000u), __uint_as_float(old.y << 16),
                             __uint_as_float(old.y & 0xffff0000u)};
        const bool first = !MULTI || (a.Tq - 1 - i) / NP == ck;  // the chunk that holds the low end of this row's band: dq is still untouched
        bf16_t* lop = (MULTI && a.dq_lo) ? a.dq_lo + ((int64_t)b * a.Tq + i) * d + h * DK + 16 * ct1 + 4 * y : nullptr;
        float l4[4] = {0.f, 0.f, 0.f, 0.f};
        if (seen) {
          const uint2 lv = *reinterpret_cast<const uint2*>(lop);
          l4[0] = __uint_as_float(lv.x << 16); l4[1] = __uint_as_float(lv.x & 0xffff0000u);
          l4[2] = __uint_as_float(lv.y << 16); l4[3] = __uint_as_float(lv.y & 0xffff0000u);
        }
        float n4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (first) su[r] += o4[r];
          sv[r] += acc[r];
          n4[r] = o4[r] + l4[r] + acc[r];
        }
        st4_from_f32<bf16_t>(a.dq + (int64_t)b * a.dq_sb + (int64_t)i * a.dq_sr + h * DK + 16 * ct1 + 4 * y, n4);
        if (again) {
          float r4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) r4[r] = n4[r] - bf2f(f2bf(n4[r]));
          st4_from_f32<bf16_t>(lop, r4);
        }
      }
      // ---- (2) dp^T[c][n] += qv^T[c][q] dbd[q][n] over the tile's 32 queries (one k-step)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          if (!(S2T_GLUE_DBG & 2)) dp[nt][ct] = mfma16(qa[ct], dbn[nt], dp[nt][ct]);
      if (more) tile_store(q0 + TQ, buf ^ 1);  // (the other buffer was last read a tile ago, behind the previous barrier)
      __syncthreads();
    }
    // ---- the chunk's dp^T -> [n][64 channels] bf16 rows in LDS (the position image's place: every product of the chunk is
    // done), then whole 128-byte rows to the partial table (zeros where no query row met the chunk: every row of the table is
    // written, the reduction sums all utterances).
    // lane (n = x, y) of tile nt holds channels 16 ct + 4 y + r of position n0 + 16 (4 w + nt) + x
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = 16 * (4 * w + nt) + x;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const uint2 v = make_uint2(bf16pack(dp[nt][ct][0], dp[nt][ct][1]), bf16pack(dp[nt][ct][2], dp[nt][ct][3]));
        *reinterpret_cast<uint2*>(lp + n * 128 + (((2 * ct + (y >> 1)) ^ (n & 7)) << 4) + (y & 1) * 8) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = tid + 512 * u;
      const int n = c >> 3, ch = c & 7;
      if (n0 + n < npos && !(S2T_GLUE_DBG & 16))
        *reinterpret_cast<uint4*>(out + (int64_t)(n0 + n) * d + ch * 8) = *reinterpret_cast<const uint4*>(lp + n * 128 + ((ch ^ (n & 7)) << 4));
    }
    if (MULTI) __syncthreads();  // (the next chunk's position image goes where these rows were read from)
  }
  // ---- column sums of the two branches: 16 query lanes by shuffles, the two waves of a channel tile through LDS, one atomic
  // per channel and branch into a replica of the workspace
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    su[r] = s2t_sum16_up(su[r]);
    sv[r] = s2t_sum16_up(sv[r]);
  }
  if (x == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[(0 * 8 + w) * 64 + 4 * y + r] = su[r];   // wave w: channels 16 ct1 + 4 y + r (slot 4 y + r of its 16)
      red[(1 * 8 + w) * 64 + 4 * y + r] = sv[r];
    }
  }
  __syncthreads();  // (also: every wave is done with the dbd tile)
  if (tid < 128) {
    const int br = tid >> 6, c = tid & 63;
    const int ct = c >> 4, sl = c & 15;
    const float sum = red[(br * 8 + ct) * 64 + sl] + red[(br * 8 + ct + 4) * 64 + sl];
    const int64_t ro = (int64_t)(z % a.replicas) * a.replica_stride;
    atomicAdd((br ? a.dv : a.du) + ro + h * DK + c, sum);
  }
}

// dp[n][col] = sum_b part[b][n][col]  (fp32 out, overwritten): 8 columns per thread, the utterances dealt to the four
// 64-thread rows of a workgroup (independent 16-byte loads in flight), the four partial sums meet in LDS
// (blockIdx.y: the entry of a batch — the layers of a backward pass reduced by one launch at its end)
constexpr int DP_BATCH = 16;
struct DpBatch {
  const bf16_t* part[DP_BATCH];
  float* dp[DP_BATCH];
};
__global__ __launch_bounds__(256) void relpos_dp_reduce_kernel(const DpBatch batch, int B, int64_t per_b /* npos * d */) {
  const bf16_t* __restrict__ part = batch.part[blockIdx.y];
  float* __restrict__ dp = batch.dp[blockIdx.y];
  __shared__ float red[3][64][8];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t e = ((int64_t)blockIdx.x * 64 + tx) * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < per_b) {
#pragma unroll 4
    for (int b = ty; b < B; b += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(part + (int64_t)b * per_b + e);
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[2 * t] += __uint_as_float(w4[t] << 16);
        s[2 * t + 1] += __uint_as_float(w4[t] & 0xffff0000u);
      }
    }
  }
  if (ty > 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) red[ty - 1][tx][t] = s[t];
  }
  __syncthreads();
  if (ty == 0 && e < per_b) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int t = 0; t < 8; ++t) s[t] += red[g][tx][t];
    *reinterpret_cast<float4*>(dp + e) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(dp + e + 4) = make_float4(s[4], s[5], s[6], s[7]);
  }
}

}  // namespace

extern "C" int s2t_relpos_glue(const void* dbd, int64_t ldb, const void* pos_p, int64_t p_sr, const void* qv, void* dq,
                               int64_t dq_sb, int64_t dq_sr, float* dpos_u, float* dpos_v, int replicas, int64_t replica_stride,
                               void* dp_part, float* dp, int B, int H, int Tq, int dk, const int32_t* cu, void* dq_lo,
                               void* stream) {
  if (!dbd || !pos_p || !qv || !dq || !dpos_u || !dpos_v || !dp_part || B <= 0 || H <= 0 || Tq <= 0 || replicas < 1)
    return S2T_ERR_ARG;
  if (dk != DK || Tq > 32768) return S2T_ERR_UNSUPPORTED;
  if (ldb < 2 * Tq - 1 || ldb % 8 || p_sr % 8 || dq_sr % 4 || dq_sb % 4) return S2T_ERR_ARG;
  if (((uintptr_t)dbd % 16) || ((uintptr_t)pos_p % 16) || ((uintptr_t)qv % 16) || ((uintptr_t)dq % 8) || ((uintptr_t)dp_part % 16) ||
      (dp && ((uintptr_t)dp % 16)))
    return S2T_ERR_ALIGN;
  GlueArgs a = {};
  a.dbd = (const bf16_t*)dbd; a.ldb = ldb; a.pos_p = (const bf16_t*)pos_p; a.p_sr = p_sr; a.qv = (const bf16_t*)qv;
  a.dq = (bf16_t*)dq; a.dq_sb = dq_sb; a.dq_sr = dq_sr; a.du = dpos_u; a.dv = dpos_v; a.replicas = replicas;
  a.replica_stride = replica_stride; a.dp_part = (bf16_t*)dp_part; a.B = B; a.H = H; a.Tq = Tq; a.cu = cu;
  if (dq_lo && ((uintptr_t)dq_lo % 8)) return S2T_ERR_ALIGN;
  a.dq_lo = 2 * Tq - 1 > NP ? (bf16_t*)dq_lo : nullptr;  // (one chunk: one rounding anyway)
  hipStream_t s = (hipStream_t)stream;
  if (2 * Tq - 1 > NP) hipLaunchKernelGGL(relpos_glue_kernel<true>, dim3(B * H), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(relpos_glue_kernel<false>, dim3(B * H), dim3(512), 0, s, a);
  if (dp) {  // (dp == NULL: the caller sums the partial tables later, several layers per launch: s2t_relpos_dp_reduce)
    const int64_t per_b = (int64_t)(2 * Tq - 1) * H * DK;
    DpBatch bt = {};
    bt.part[0] = (const bf16_t*)dp_part;
    bt.dp[0] = dp;
    hipLaunchKernelGGL(relpos_dp_reduce_kernel, dim3((unsigned)((per_b / 8 + 63) / 64)), dim3(256), 0, s, bt, B, per_b);
  }
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_relpos_dp_reduce(const void* const* dp_parts, float* const* dps, int n, int B, int H, int Tq, int dk,
                                    void* stream) {
  if (!dp_parts || !dps || n <= 0 || B <= 0 || H <= 0 || Tq <= 0) return S2T_ERR_ARG;
  if (dk != DK) return S2T_ERR_UNSUPPORTED;
  const int64_t per_b = (int64_t)(2 * Tq - 1) * H * DK;
  for (int i0 = 0; i0 < n; i0 += DP_BATCH) {
    const int cnt = n - i0 < DP_BATCH ? n - i0 : DP_BATCH;
    DpBatch bt = {};
    for (int i = 0; i < cnt; ++i) {
      if (!dp_parts[i0 + i] || !dps[i0 + i]) return S2T_ERR_ARG;
      if (((uintptr_t)dp_parts[i0 + i] % 16) || ((uintptr_t)dps[i0 + i] % 16)) return S2T_ERR_ALIGN;
      bt.part[i] = (const bf16_t*)dp_parts[i0 + i];
      bt.dp[i] = dps[i0 + i];
    }
    hipLaunchKernelGGL(relpos_dp_reduce_kernel, dim3((unsigned)((per_b / 8 + 63) / 64), cnt), dim3(256), 0, (hipStream_t)stream, bt, B,
                       per_b);
  }
  return S2T_LAUNCH_CHECK();
}
