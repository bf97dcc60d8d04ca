// Relative-position self-attention backward in ONE pass (round 5): dQ, dK, dV, both position-bias gradients and this call's
// share of the position-table gradient, for sequences of up to 256 frames (the Conformer encoder of the recipes: T' = 250).
//   fairseq/modules/espnet_multihead_attention.py:292-311 (rel_shift), :313-356 (forward), its autograd backward.
// Replaces, for T' <= 256, attn_bwd_dq_kernel<true> + attn_bwd_dkv_kernel<true> (csrc/attention_fused.hip: scores, position
// band, exponentials and dP computed TWICE, the skewed score gradient dbd [h][B][T][2T-1] written to HBM) + relpos_glue_kernel
// (csrc/relpos_glue.hip: dbd read back): the scores are formed once and dbd never leaves the chip.
//
// One workgroup of 8 waves per (utterance, head); resident in LDS for the whole kernel: the head's 2T-1 projected position rows
// (64 KiB) and its K rows (32 KiB).  Wave w keeps the V fragments of keys 32w .. 32w+31 and their dK / dV accumulators, and
// the accumulators of position tiles {w, w+8, w+16, w+24} of the position-table gradient (144 of its 256 registers).  Per
// tile of 32 queries:
//   phase A  (Q+u), (Q+v), dO rows -> LDS; delta = rowsum(dO * O); the tile's columns of the dbd image zeroed
//   phase B  every wave, its 32 keys x the 32 queries: S = (Q+u) K^T, the position band (Q+v) P^T re-indexed inside the
//            registers (the reference's rel_shift = a rotation of each 16-lane row: DPP), dP = dO V^T, P = exp(S - lse), dropout,
//            dS = P (dP - delta); dV += Pd^T dO, dK += dS^T (Q+u) straight from the registers; dS (bf16) written to LDS twice:
//            SKEWED into the dbd image [32 q][512 n], n = T-1-i+j, and by key [32 q][256 j] — the one exchange of the tile
//            between the three owners of its products
//   phase C  wave (channel tile, query half):  dQ^T = K^T dS^T (rows of the copy by key) + P^T dbd^T (rows of the image);
//            wave (position tiles):            dp^T += (Q+v)^T dbd (columns of the image)
// Three barriers per tile.  Products are 16x16x32 bf16 MFMAs (16x16x16 for dV / dK: one query tile at a time, so that a tile's
// Pd / dS die early); fragment layouts as in attention_fused.hip / relpos_glue.hip.
// Measured (tools/rpb_probe.py, 64 utterances x 4 heads, T' = 250, fill 0.8): 70 us; both waves of a SIMD are bound by vector
// issue in phase B (~870 vector instructions per wave and tile: exponentials, dropout hash, rel_shift, the addresses of 32
// two-byte LDS writes), the matrix work is ~7 us of it.  What it took to run without scratch spills (each cost 10 - 40 us):
// lane coordinates re-derived per phase (asm barrier) so that swizzled LDS addresses are not hoisted out of the tile loop; no
// branch around accumulating MFMAs (a join rotates the ~130 loop-carried registers); one kernel per dropout form; the next
// tile's rows fetched unconditionally (clamped) at the head of phase C.
#include "common.h"

#ifndef S2T_RPB_DBG
#define S2T_RPB_DBG 0  // experiment builds: 1 no product (1), 2 no product (2), 4 no product (3), 8 no dropout arithmetic
#endif

#ifndef S2T_RPB_PRIO
#define S2T_RPB_PRIO 0  // experiment: s_setprio of waves 4 - 7 in the score phase
#endif
#ifndef S2T_RPB_SHIFT
#define S2T_RPB_SHIFT 1  // the rel_shift of the position band: 0 ds_bpermute_b32 (12 per query tile, measured 86 us), 1 DPP row rotates (79 us: fewer live registers)
#endif

namespace {

constexpr int DK = 64;
constexpr int TQ = 32;    // queries per tile
constexpr int NP = 512;   // rows of the position image / columns of the dbd image (2T - 1 <= 511)
constexpr int KMAX = 256; // keys (= frames) of an utterance

constexpr int L_P = 0;                          // [NP][128 B]      projected position rows of the head, key128 swizzle
constexpr int L_K = L_P + NP * 128;             // [KMAX][128 B]    K rows of the (utterance, head), key128 swizzle
// The two dS images have PADDED rows instead of an XOR swizzle (+16 bytes: consecutive rows start four banks apart, so the
// sixteen rows of a row-wise ds_read_b128 cover the 64 banks once): an element's address is then LINEAR in (query, key), and the
// 32 two-byte writes of a wave per tile are one base register + an immediate offset each (with the swizzle: six vector
// instructions of address arithmetic per write, a quarter of the vector work of the phase that bounds the kernel).
constexpr int DROW = 1024 + 16;                 // row stride of the skewed image (512 columns + pad)
constexpr int AROW = 512 + 16;                  // row stride of the copy by key (256 keys + pad)
constexpr int L_D = L_K + KMAX * 128;           // [TQ][DROW]       skewed dS of the tile: row q, column n = T-1-i+j
constexpr int L_QU = L_D + TQ * DROW;           // [TQ][128 B]      Q + pos_bias_u, (r & 7) swizzle
constexpr int L_QV = L_QU + TQ * 128;           //                  Q + pos_bias_v
constexpr int L_DO = L_QV + TQ * 128;           //                  dO
constexpr int L_DA = L_DO + TQ * 128;           // [TQ][AROW]       dS of the tile by KEY (unskewed)
constexpr int L_ST = L_DA + TQ * AROW;          // lse[32], delta[32]
constexpr int L_BI = L_ST + 2 * TQ * 4;         // pos_bias_u[64], pos_bias_v[64] of the head (floats)
constexpr int L_BYTES = L_BI + 2 * DK * 4;
static_assert(TQ * AROW >= 2 * 8 * 64 * 4, "the column sums of the epilogue fit the copy by key");
static_assert(L_BYTES <= 160 * 1024, "LDS budget");

typedef short s16x4v __attribute__((ext_vector_type(4)));

struct RpbArgs {
  const bf16_t *q, *k, *v;
  int64_t q_sb, q_sr, k_sb, k_sr, v_sb, v_sr;
  const bf16_t *o, *dO;
  int64_t o_sb, o_sr;
  const float* lse;  // [B*H][T]
  bf16_t *dq, *dk, *dv;  // layouts of q, k, v
  const bf16_t* pos_p;   // [2T-1][p_sr], head h at column h*64
  int64_t p_sr;
  const float *pos_u, *pos_v;  // [H*64]
  float *du, *dv_;             // column-sum targets (replicated)
  int replicas;
  int64_t replica_stride;
  bf16_t* dp_part;  // [B][2T-1][H*64]
  int B, H, T;      // T: the (padded) number of queries — with positions also that of the keys
  int Tk;           // (plain form) the padded number of keys
  int causal;       // (plain form) key j > query i masked
  const bf16_t* o_lo;  // (plain form, optional) the forward's rounding remainder of o: delta is taken on o + o_lo
  const int32_t* cu_k; // (plain form) packed key side: rows of utterance b of k, v, dk, dv = cu_k[b] .. cu_k[b+1]
  const int32_t* key_lens;
  float scale;
  float drop_p;
  const uint64_t* drop_seed;
  uint32_t drop_site;
  const int32_t* cu;  // packed batch: rows of utterance b = cu[b] .. cu[b+1]
};

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint2 tr64(const char* a) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a)));
}
// swizzle keys of relpos_glue.hip (natural-order transposed reads without bank conflicts)
__device__ __forceinline__ int key128(int r) { return 2 * (((r >> 1) & 1) | (((r >> 3) & 1) << 1)); }

// 8 bf16 + per-column fp32 bias, rounded to bf16 (q + pos_bias_u / q + pos_bias_v as the forward rounds them)
__device__ __forceinline__ uint4 add_bias8(uint4 qv, const float* __restrict__ bias) {
  const uint32_t w[4] = {qv.x, qv.y, qv.z, qv.w};
  uint32_t o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    o[t] = bf16pack(__uint_as_float(w[t] << 16) + bias[2 * t], __uint_as_float(w[t] & 0xffff0000u) + bias[2 * t + 1]);
  return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  uint32_t o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = bf16pack(v[2 * t], v[2 * t + 1]);
  return as_frag(make_uint4(o[0], o[1], o[2], o[3]));
}

// ---- images with 128-byte rows and the (r & 7) swizzle (the query tiles; attention_fused.hip's layout) -----------------
// row-wise fragment: 8 consecutive k (chunk ks*4 + y) of row blk*16 + x
__device__ __forceinline__ bf16x8 rows7(const char* img, int blk, int ks, int x, int y) {
  const int r = blk * 16 + x, c = ks * 4 + y;
  return as_frag(*reinterpret_cast<const uint4*>(img + r * 128 + ((c ^ (r & 7)) << 4)));
}
// column-wise fragment, operand rows = image columns 16 cblk + x, k = image rows in the PERMUTED order
// kappa(y, j) = 16 (j >> 2) + 4 y + (j & 3)   (the order of a lane's accumulator registers over two 16-row tiles)
__device__ __forceinline__ bf16x8 cols7_perm(const char* img, int cblk, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = 16 * h + 4 * y + qq;
    const uint2 t = tr64(img + R * 128 + (((2 * cblk + (p >> 1)) ^ (R & 7)) << 4) + (p & 1) * 8);
    w[2 * h] = t.x;
    w[2 * h + 1] = t.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}
// the same in NATURAL order: k = image rows 8 y + j
__device__ __forceinline__ bf16x8 cols7_nat(const char* img, int cblk, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = 8 * y + 4 * h + qq;
    const uint2 t = tr64(img + R * 128 + (((2 * cblk + (p >> 1)) ^ (R & 7)) << 4) + (p & 1) * 8);
    w[2 * h] = t.x;
    w[2 * h + 1] = t.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}
// column-wise fragment of ONE 16-row tile for the 16x16x16 product: operand rows = image columns 16 cblk + x, k = image rows
// 16 blk + 4 y + j — the order of a lane's four accumulator registers of that tile
__device__ __forceinline__ s16x4v cols7_k16(const char* img, int blk, int cblk, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  const int R = 16 * blk + 4 * y + qq;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4v*)(img + R * 128 + (((2 * cblk + (p >> 1)) ^ (R & 7)) << 4) + (p & 1) * 8));
}
__device__ __forceinline__ f32x4 mfma16k16(s16x4v a, s16x4v b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// ---- images with 128-byte rows and the key128 swizzle (positions, K) ---------------------------------------------------
__device__ __forceinline__ bf16x8 rows128(const char* img, int r, int c) {
  return as_frag(*reinterpret_cast<const uint4*>(img + r * 128 + ((c ^ key128(r)) << 4)));
}
// column-wise: operand rows = image columns 16 cblk + x, k = image rows row0 + 8 y + j (natural order)
__device__ __forceinline__ bf16x8 cols128(const char* img, int row0, int cblk, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = row0 + 8 * y + 4 * h + qq;
    const uint2 t = tr64(img + R * 128 + (((2 * cblk + (p >> 1)) ^ key128(R)) << 4) + (p & 1) * 8);
    w[2 * h] = t.x;
    w[2 * h + 1] = t.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}
// ---- the skewed dS image (rows of DROW bytes, no swizzle) ------------------------------------------------------------------
// column-wise: operand columns = image columns 16 nblk + x, k = image rows 8 y + j
__device__ __forceinline__ bf16x8 colsD(const char* img, int nblk, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = 8 * y + 4 * h + qq;
    const uint2 t = tr64(img + R * DROW + nblk * 32 + p * 8);
    w[2 * h] = t.x;
    w[2 * h + 1] = t.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}

// DROP: 0 no dropout, 1 the 32-bit pair hash (even T, index space below 2^32), 2 the general 64-bit-indexed hash — one
// instantiation each: with both hashes compiled in, the one that never runs still costs registers in a kernel that has none to spare
// REL: the relative-position form.  false: plain scaled-dot-product attention (multihead_attention.py:161-431 backward; the decoder's
// self- and encoder-decoder attention, the encoder of the plain Transformer recipes) on the same schedule — no position image, no
// skewed image, products (1) and (2) and the bias sums compiled away; queries and keys may differ in number and packing, the
// causal mask and the o_lo form of delta exist only here.
template <int DROP, bool REL>
__global__ __launch_bounds__(512) void relpos_attn_bwd_kernel(const RpbArgs a_in) {
  __shared__ __attribute__((aligned(16))) char lds[L_BYTES];
  RpbArgs a = a_in;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 15, y = lane >> 4;
  const int z = blockIdx.x;
  const int b = z / a.H, h = z % a.H;
  const int T = a.T;
  const int npos = 2 * T - 1;
  const int d = a.H * DK;
  const int Tk = REL ? T : a.Tk;
  int nq = T, nk = Tk;
  if (a.cu) {
    const int r0 = a.cu[b];
    nq = a.cu[b + 1] - r0;
    const int64_t oq = (int64_t)r0 * a.q_sr - (int64_t)b * a.q_sb, oo = (int64_t)r0 * a.o_sr - (int64_t)b * a.o_sb;
    a.q += oq; a.dq += oq; a.o += oo; a.dO += oo;
    if (!REL && a.o_lo) a.o_lo += oo;
  }
  const int32_t* cuk = REL ? a.cu : a.cu_k;
  if (REL) nk = nq;
  if (cuk) {
    const int r0 = cuk[b];
    nk = cuk[b + 1] - r0;
    const int64_t ok = (int64_t)r0 * a.k_sr - (int64_t)b * a.k_sb, ov = (int64_t)r0 * a.v_sr - (int64_t)b * a.v_sb;
    a.k += ok; a.dk += ok; a.v += ov; a.dv += ov;
  }
  const int klen = a.key_lens ? min(a.key_lens[b], nk) : nk;
  char* lp = lds + L_P;
  char* lk = lds + L_K;
  char* ld = lds + L_D;
  char* lqu = lds + L_QU;
  char* lqv = lds + L_QV;
  char* ldo = lds + L_DO;
  char* lda = lds + L_DA;
  float* lse_s = reinterpret_cast<float*>(lds + L_ST);
  float* del_s = lse_s + TQ;
  bf16_t* out = REL ? a.dp_part + ((int64_t)b * npos) * d + h * DK : nullptr;

  if (REL && nq <= 0) {  // an utterance without rows: its table of partial sums is zero (the reduction adds every utterance's)
    for (int c = tid; c < npos * 8; c += 512)
      *reinterpret_cast<uint4*>(out + (int64_t)(c >> 3) * d + (c & 7) * 8) = make_uint4(0, 0, 0, 0);
    return;
  }
  if (!REL && (nq <= 0 || nk <= 0)) {  // no query of the utterance: its keys get zero gradients; no key: its queries do
    for (int c = tid; c < max(nk, 0) * 8; c += 512) {
      *reinterpret_cast<uint4*>(a.dk + (int64_t)b * a.k_sb + (int64_t)(c >> 3) * a.k_sr + h * DK + (c & 7) * 8) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(a.dv + (int64_t)b * a.v_sb + (int64_t)(c >> 3) * a.v_sr + h * DK + (c & 7) * 8) = make_uint4(0, 0, 0, 0);
    }
    for (int c = tid; c < max(nq, 0) * 8; c += 512)
      *reinterpret_cast<uint4*>(a.dq + (int64_t)b * a.q_sb + (int64_t)(c >> 3) * a.q_sr + h * DK + (c & 7) * 8) = make_uint4(0, 0, 0, 0);
    return;
  }
  const bf16_t* qb = a.q + (int64_t)b * a.q_sb + h * DK;
  const bf16_t* kb = a.k + (int64_t)b * a.k_sb + h * DK;
  const bf16_t* vb = a.v + (int64_t)b * a.v_sb + h * DK;
  const bf16_t* ob = a.o + (int64_t)b * a.o_sb + h * DK;
  const bf16_t* dob = a.dO + (int64_t)b * a.o_sb + h * DK;
  const bf16_t* olb = (!REL && a.o_lo) ? a.o_lo + (int64_t)b * a.o_sb + h * DK : nullptr;

  // ---- tile operands global -> registers (one tile ahead): threads 0..255 the Q pieces, 256..511 the dO and O pieces
  uint4 t0 = make_uint4(0, 0, 0, 0), t1 = make_uint4(0, 0, 0, 0), t2 = make_uint4(0, 0, 0, 0);
  float nlse = 0.f;
  const int prow = (tid & 255) >> 3, pch = tid & 7;
  auto tile_load = [&](int q0) __attribute__((always_inline)) {
    const int r = min(q0 + prow, nq - 1);
    if (tid < 256) {
      t0 = ldg16(qb + (int64_t)r * a.q_sr + pch * 8);
    } else {
      t0 = ldg16(dob + (int64_t)r * a.o_sr + pch * 8);
      t1 = ldg16(ob + (int64_t)r * a.o_sr + pch * 8);
      if (!REL && olb) t2 = ldg16(olb + (int64_t)r * a.o_sr + pch * 8);
    }
    if (tid < TQ) nlse = a.lse[(int64_t)z * T + min(q0 + tid, nq - 1)];
  };
  tile_load(0);

  // ---- resident images: the head's position rows (rows >= 2T-1 zero) and the K rows (rows >= nq zero)
  {
    uint4 t[8];
    if constexpr (REL) {
      const bf16_t* pp = a.pos_p + h * DK;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = tid + 512 * u;
        t[u] = ldg16(pp + (int64_t)min(c >> 3, npos - 1) * a.p_sr + (c & 7) * 8);
      }
    }
    uint4 tk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = tid + 512 * u;
      tk[u] = ldg16(kb + (int64_t)min(c >> 3, nk - 1) * a.k_sr + (c & 7) * 8);
    }
    if constexpr (REL) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = tid + 512 * u;
        const int n = c >> 3, ch = c & 7;
        *reinterpret_cast<uint4*>(lp + n * 128 + ((ch ^ key128(n)) << 4)) = n < npos ? t[u] : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = tid + 512 * u;
      const int n = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(lk + n * 128 + ((ch ^ key128(n)) << 4)) = n < nk ? tk[u] : make_uint4(0, 0, 0, 0);
    }
  }
  // ---- this wave's keys: fragments (B operands: lane x = key, k = 32 ks + 8 y + j) and accumulators
  const int nkw = (nk + 31) >> 5;            // waves that hold keys
  bf16x8 vf[2][2];  // (the K fragments are row reads of the resident image: 16 registers fewer across the tile loop)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const bf16_t* vp = vb + (int64_t)min(32 * w + 16 * kt + x, nk - 1) * a.v_sr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) vf[kt][ks] = as_frag(ldg16(vp + (ks * 4 + y) * 8));
  }
  f32x4 dk[2][4], dv[2][4], dp[4][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) dk[u][ct] = dv[u][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) dp[nt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float su[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
  const int ct1 = w & 3, qh = w >> 2;  // phase C: channel tile and query half of this wave's dQ block

  const uint64_t dkey = DROP ? s2t_drop_key(a.drop_seed, a.drop_site) : 0ull;
  const uint32_t dth = s2t_drop_thresh(a.drop_p);
  const float dinv = s2t_drop_scale(a.drop_p);
  float* bias_s = reinterpret_cast<float*>(lds + L_BI);
  if (REL && tid < 2 * DK) bias_s[tid] = tid < DK ? a.pos_u[h * DK + tid] : a.pos_v[h * DK + tid - DK];

#if S2T_RPB_DBG & 64
  unsigned long long stamp[16];  // kernel start .. loop, the nine stamps of the second tile, loop end
  int nstamp = 0, stamp_on = 1;
#define RSTAMP()                                                                   \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    if (stamp_on && nstamp < 16) stamp[nstamp++] = __builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define RSTAMP()
#endif
  RSTAMP();
  for (int q0 = 0; q0 < nq; q0 += TQ) {
    // columns of the dbd image this tile touches: the bands of its 32 rows over the keys of the nkw waves
    const int nlo = max(0, T - 1 - (q0 + TQ - 1)), nhi = min(NP - 1, T - 1 - q0 + 32 * nkw - 1);
    const int c_lo = (nlo >> 5) << 2, c_hi = ((nhi >> 5) << 2) + 3;  // 16-byte chunks of the k-steps (32 columns) the products read
#if S2T_RPB_DBG & 64
    stamp_on = q0 == TQ;
#endif
    __syncthreads();
    RSTAMP();
    // ---- phase A
    // (lane coordinates are re-derived at the head of every phase: the LDS addresses formed from them are loop invariants that the
    // compiler otherwise computes once, in front of the tile loop, and keeps — or spills — across all three phases)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int prow_ = (tl & 255) >> 3, pch_ = tl & 7;
    {
      const bool live = q0 + prow_ < nq;
      if (tl < 256) {
        const int r = prow_;
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        if constexpr (REL) {
          *reinterpret_cast<uint4*>(lqu + r * 128 + ((pch_ ^ (r & 7)) << 4)) = live ? add_bias8(t0, bias_s + pch_ * 8) : z4;
          *reinterpret_cast<uint4*>(lqv + r * 128 + ((pch_ ^ (r & 7)) << 4)) = live ? add_bias8(t0, bias_s + DK + pch_ * 8) : z4;
        } else {
          *reinterpret_cast<uint4*>(lqu + r * 128 + ((pch_ ^ (r & 7)) << 4)) = live ? t0 : z4;
        }
      } else {
        const int r = prow_;
        *reinterpret_cast<uint4*>(ldo + r * 128 + ((pch_ ^ (r & 7)) << 4)) = live ? t0 : make_uint4(0, 0, 0, 0);
        const uint32_t dw[4] = {t0.x, t0.y, t0.z, t0.w}, ow[4] = {t1.x, t1.y, t1.z, t1.w};
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          part += __uint_as_float(ow[t] << 16) * __uint_as_float(dw[t] << 16) +
                  __uint_as_float(ow[t] & 0xffff0000u) * __uint_as_float(dw[t] & 0xffff0000u);
        if (!REL && olb) {  // (workgroup-uniform) delta on o + o_lo: 16 mantissa bits of the forward's fp32 output
          const uint32_t lw[4] = {t2.x, t2.y, t2.z, t2.w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
            part += __uint_as_float(lw[t] << 16) * __uint_as_float(dw[t] << 16) +
                    __uint_as_float(lw[t] & 0xffff0000u) * __uint_as_float(dw[t] & 0xffff0000u);
        }
        part = s2t_xadd<1>(part);
        part = s2t_xadd<2>(part);
        part = s2t_xadd<4>(part);
        if (pch_ == 0) del_s[r] = part;
      }
      if (tl < TQ) lse_s[tl] = nlse;
      // zero the touched columns of the image (32 rows x 64 chunks, 4 pieces per thread)
      if constexpr (REL) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = tl + 512 * u;
          const int r = c >> 6, ch = c & 63;
          if (ch >= c_lo && ch <= c_hi) *reinterpret_cast<uint4*>(ld + r * DROW + ch * 16) = make_uint4(0, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    RSTAMP();
    // ---- phase B: scores of this wave's 32 keys against the tile's 32 queries
    int xl = x, yl = y;
    asm volatile("" : "+v"(xl), "+v"(yl));
#if S2T_RPB_PRIO
    // the two waves of a SIMD (w, w + 4) enter the phase together and are bound by vector issue; the arbiter serves the older
    // one first, which then waits at the barrier while the younger runs alone with its dependency stalls exposed
    if (w >= 4) __builtin_amdgcn_s_setprio(S2T_RPB_PRIO);
#endif
    {  // (waves whose keys all lie beyond the utterance run along: they would wait at the barrier anyway, and a branch around the
       //  accumulating MFMAs costs a rotation of the ~130 loop-carried accumulator registers at its join; their dS are zeros)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // (three steps with short-lived operands — position band, scores, dP — fenced so that the scheduler does not hoist all the
        // fragment reads of a query tile to its top: the kernel lives on its last registers)
        f32x4 s4[2], dp4[2], bd[3];
        uint32_t pw[2][2], dw[2][2];  // Pd and dS of this query tile as bf16 pairs: [key tile][queries 4y + (0, 1) | (2, 3)]
        // position band of (16 q x 32 keys): rows n = nb0 + (0..46), nb0 = T-1-(q0+16qt+15) + 32w; lane x = n index of a tile
        if constexpr (REL) {
          bf16x8 qv[2];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) qv[ks] = rows7(lqv, qt, ks, xl, yl);
          const int nb0 = T - 1 - (q0 + 16 * qt + 15) + 32 * w;
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) {
            int n = nb0 + 16 * nt + xl;
            n = n < 0 ? 0 : (n > NP - 1 ? NP - 1 : n);
            bd[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bd[nt] = mfma16(qv[ks], rows128(lp, n, ks * 4 + yl), bd[nt]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          bf16x8 qa[2];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) qa[ks] = rows7(lqu, qt, ks, xl, yl);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            s4[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s4[kt] = mfma16(qa[ks], rows128(lk, 32 * w + 16 * kt + xl, ks * 4 + yl), s4[kt]);
          }
        }
        // rel_shift (espnet_multihead_attention.py:292-311) without leaving the registers: element (q_local = 4y + r, key x of key
        // tile kt) is band row 16 kt + 15 - q_local + x, held by lane (x + 15 - q_local) & 15 of the SAME 16-lane row (same y, same
        // register r) in band tile kt or kt + 1: a rotation of the row to the right by q_local + 1 lanes — ds_bpermute_b32 through
        // the LDS crossbar, twelve per query tile, one wait — and a select between the two tiles.  (Before: twelve LDS writes,
        // eight reads and four waits per query tile through a wave-private scratch of 18 KiB in all; as DPP row rotates, whose
        // amount cannot depend on the row: three per register, 72 vector instructions per tile in a phase bound by vector issue.)
        if constexpr (REL)
        {
          float rt[3][4];
#if S2T_RPB_SHIFT == 1
#define S2T_RPB_ROT(r)                                                                                \
  _Pragma("unroll") for (int nt = 0; nt < 3; ++nt) {                                                  \
    const float bv_ = bd[nt][r]; /* (a copy: bit_cast of a vector ELEMENT lvalue reads element 0) */ \
    int t = __float_as_int(bv_);                                                       \
    t = __builtin_amdgcn_update_dpp(t, t, 0x121 + r, 0xf, 0xf, false); /* row_ror:(r + 1) */          \
    t = __builtin_amdgcn_update_dpp(t, t, 0x124, 0xa, 0xf, false);     /* row_ror:4 on rows 1, 3 */   \
    t = __builtin_amdgcn_update_dpp(t, t, 0x128, 0xc, 0xf, false);     /* row_ror:8 on rows 2, 3 */   \
    rt[nt][r] = __builtin_bit_cast(float, t);                                                         \
  }
          S2T_RPB_ROT(0) S2T_RPB_ROT(1) S2T_RPB_ROT(2) S2T_RPB_ROT(3)
#undef S2T_RPB_ROT
#else
#if S2T_RPB_DBG & 16
          asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int src = ((lane & 48) | ((xl - (4 * yl + r + 1)) & 15)) << 2;  // byte address of the source lane
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
            {
              const float bv_ = bd[nt][r];  // (a copy: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0 — hipcc 7.2)
              rt[nt][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(bv_)));
            }
          }
#endif
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool first = xl <= 4 * yl + r;  // band row 15 - q_local + x lies in the lower of the two tiles
            s4[0][r] += first ? rt[0][r] : rt[1][r];
            s4[1][r] += first ? rt[1][r] : rt[2][r];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          bf16x8 dof[2];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) dof[ks] = rows7(ldo, qt, ks, xl, yl);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            dp4[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dp4[kt] = mfma16(dof[ks], vf[kt][ks], dp4[kt]);
          }
        }
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * qt + 4 * yl);
        const f32x4 del4 = *reinterpret_cast<const f32x4*>(del_s + 16 * qt + 4 * yl);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int jkey = 32 * w + 16 * kt + xl;
          const int jck = min(jkey, nk - 1);
          const bool key_okk = jkey < klen;
          // dropout bits as attn_bwd_dkv_kernel draws them: one 32-bit hash serves an even / odd key pair of a query row; lanes
          // xl and xl ^ 1 hash one query each of a pair of queries and swap
          uint32_t rbits[4] = {65535u, 65535u, 65535u, 65535u};
          if constexpr (DROP != 0 && !(S2T_RPB_DBG & 8)) {
            if constexpr (DROP == 1) {
#pragma unroll
              for (int rb = 0; rb < 4; rb += 2) {
                const int im = min(q0 + 16 * qt + 4 * yl + rb + (xl & 1), nq - 1);
                const uint32_t pair = (uint32_t)z * (uint32_t)T * (uint32_t)(Tk >> 1) + __umul24((uint32_t)im, (uint32_t)(Tk >> 1)) +
                                      (uint32_t)(jck >> 1);
                const uint32_t hm = s2t_mix32(pair ^ (uint32_t)dkey) ^ (uint32_t)(dkey >> 32);
                const uint32_t ho = (uint32_t)__builtin_amdgcn_mov_dpp((int)hm, 0xB1, 0xf, 0xf, true);  // lane xl ^ 1
                const uint32_t h0 = (xl & 1) ? ho : hm, h1 = (xl & 1) ? hm : ho;
                const uint32_t hsh = (uint32_t)(jck & 1) << 4;  // the odd key of a pair takes the high half: one bit-field extract
                rbits[rb] = __builtin_amdgcn_ubfe(h0, hsh, 16);
                rbits[rb + 1] = __builtin_amdgcn_ubfe(h1, hsh, 16);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = min(q0 + 16 * qt + 4 * yl + r, nq - 1);
                rbits[r] = s2t_rand_u32(dkey, ((uint64_t)z * T + (uint64_t)i) * (uint64_t)Tk + jck);
              }
            }
          }
          float pd4[4], ds4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = q0 + 16 * qt + 4 * yl + r;
            const bool ok = key_okk & (i < nq) & !(!REL && a.causal != 0 && jkey > i);
            const float p = __expf(ok ? s4[kt][r] * a.scale - lse4[r] : -INFINITY);
            float dpv = dp4[kt][r], pdrop = p;
            if constexpr (DROP != 0 && !(S2T_RPB_DBG & 8)) {
              const float m = rbits[r] >= dth ? dinv : 0.f;  // (one select on the multiplier instead of one per product)
              dpv *= m;
              pdrop = p * m;
            }
            pd4[r] = pdrop;
            ds4[r] = (p * a.scale) * (dpv - del4[r]);
          }
          pw[kt][0] = bf16pack(pd4[0], pd4[1]);
          pw[kt][1] = bf16pack(pd4[2], pd4[3]);
          dw[kt][0] = bf16pack(ds4[0], ds4[1]);
          dw[kt][1] = bf16pack(ds4[2], ds4[3]);
        }
        // dV^T[c][key] += dO^T[c][q] Pd[q][key];  dK^T[c][key] += (Q+u)^T[c][q] dS[q][key] over THIS query tile's 16 queries
        // (16x16x16 products: the B fragments are the lane's four values of the tile as they are — with the 16x16x32 form both
        // query tiles' values stayed live to the end of the phase, sixteen registers the kernel does not have)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const s16x4v ado = cols7_k16(ldo, qt, ct, xl, yl), aqu = cols7_k16(lqu, qt, ct, xl, yl);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            dv[kt][ct] = mfma16k16(ado, __builtin_bit_cast(s16x4v, make_uint2(pw[kt][0], pw[kt][1])), dv[kt][ct]);
            dk[kt][ct] = mfma16k16(aqu, __builtin_bit_cast(s16x4v, make_uint2(dw[kt][0], dw[kt][1])), dk[kt][ct]);
          }
        }
        // dS -> the skewed image (row q, column n = T-1-(q0+q) + key) and the copy by key: addresses linear in (q, key), one base
        // each per lane (query 4y, key 32w + x) and immediate offsets.  No branch per row: a row beyond the utterance holds zeros
        // (its probabilities are exp(-inf)); beyond the PADDED length its columns run below zero, into the pad and the highest
        // columns of the row in front of it, which no product of the tile reads (they lie beyond the band of its last row).
        {
          char* wi = REL ? ld + (4 * yl) * (DROW - 2) + (T - 1 - q0 + 32 * w + xl) * 2 : nullptr;
          char* wa = lda + (4 * yl) * AROW + (32 * w + xl) * 2;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              const uint32_t wd = dw[kt][r >> 1];
              const bf16_t val = (bf16_t)((r & 1) ? (wd >> 16) : (wd & 0xffffu));
              if constexpr (REL) *reinterpret_cast<bf16_t*>(wi + (16 * qt + r) * (DROW - 2) + kt * 32) = val;
              *reinterpret_cast<bf16_t*>(wa + (16 * qt + r) * AROW + kt * 32) = val;
            }
        }
      }
    }
#if S2T_RPB_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    RSTAMP();
    __syncthreads();
    RSTAMP();
    tile_load(q0 + TQ);  // (here, not a phase earlier: its registers stay out of phase B; unconditional — rows are clamped — so that
                         //  the old values are dead: under a condition they stay live, spilled, across the next phase B)
    // ---- phase C
    int xc = x, yc = y;
    asm volatile("" : "+v"(xc), "+v"(yc));
    const int q = 16 * qh + xc;       // this lane's query of the tile in products (1) and (3)
    const int i = q0 + q;
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
    // (3) dQ(ac)^T[c][q] = sum_key K^T[c][key] dS^T[key][q]: B = aligned 16-byte pieces of row q of the copy by key (every wave
    // wrote all 32 of its keys of every row, zeros beyond the utterance: no stale value is read)
    if (!(S2T_RPB_DBG & 4)) {
      f32x4 acc3b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int s = 0; s < nkw; s += 2) {  // (pairs: K rows beyond the utterance are zero and so are the dS of their waves)
        const bf16x8 k0 = cols128(lk, 32 * s, ct1, xc, yc), k1 = cols128(lk, 32 * s + 32, ct1, xc, yc);
        const bf16x8 b0 = as_frag(*reinterpret_cast<const uint4*>(lda + q * AROW + (4 * s + yc) * 16));
        const bf16x8 b1 = as_frag(*reinterpret_cast<const uint4*>(lda + q * AROW + (4 * s + 4 + yc) * 16));
        acc3 = mfma16(k0, b0, acc3);
        acc3b = mfma16(k1, b1, acc3b);
      }
      acc3 += acc3b;
    }
    RSTAMP();
    // (1) dQ(bd)^T[c][q] = sum_n P^T[c][n] dbd^T[n][q] over the touched columns (k-steps of 32)
    if constexpr (REL && !(S2T_RPB_DBG & 1)) {
      f32x4 accb = {0.f, 0.f, 0.f, 0.f};
      const int ks_lo = nlo >> 5, ks_hi = nhi >> 5;
      int ks = ks_lo;
      for (; ks + 1 <= ks_hi; ks += 2) {
        const bf16x8 pa0 = cols128(lp, 32 * ks, ct1, xc, yc), pa1 = cols128(lp, 32 * ks + 32, ct1, xc, yc);
        const bf16x8 db0 = as_frag(*reinterpret_cast<const uint4*>(ld + q * DROW + (4 * ks + yc) * 16));
        const bf16x8 db1 = as_frag(*reinterpret_cast<const uint4*>(ld + q * DROW + (4 * ks + 4 + yc) * 16));
        acc1 = mfma16(pa0, db0, acc1);
        accb = mfma16(pa1, db1, accb);
      }
      if (ks <= ks_hi) {
        const bf16x8 pa0 = cols128(lp, 32 * ks, ct1, xc, yc);
        const bf16x8 db0 = as_frag(*reinterpret_cast<const uint4*>(ld + q * DROW + (4 * ks + yc) * 16));
        acc1 = mfma16(pa0, db0, acc1);
      }
      acc1 += accb;
    }
    RSTAMP();
    {
      // running column sums of the two dQ branches for the position-bias gradients (per lane; folded after the last tile)
      const bool live = i < nq;
      float n4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        n4[r] = acc3[r] + acc1[r];
        if constexpr (REL) {
          su[r] += live ? acc3[r] : 0.f;
          sv[r] += live ? acc1[r] : 0.f;
        }
      }
      if (live) st4_from_f32<bf16_t>(a.dq + (int64_t)b * a.q_sb + (int64_t)i * a.q_sr + h * DK + 16 * ct1 + 4 * yc, n4);
    }
    // (2) dp^T[c][n] += sum_q (Q+v)^T[c][q] dbd[q][n]: this wave's position tiles w + 8 nt that meet the touched columns
    if constexpr (REL && !(S2T_RPB_DBG & 2)) {
      const int t_lo = nlo >> 4, t_hi = nhi >> 4;
      bf16x8 qa[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) qa[ct] = cols7_nat(lqv, ct, xc, yc);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int tg = w + 8 * nt;
        if (tg >= t_lo && tg <= t_hi) {  // (wave-uniform)
          const bf16x8 dbn = colsD(ld, tg, xc, yc);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) dp[nt][ct] = mfma16(qa[ct], dbn, dp[nt][ct]);
        }
      }
    }
    RSTAMP();
  }
#if S2T_RPB_DBG & 64
  stamp_on = 1;
  RSTAMP();
  if (REL && lane == 0 && blockIdx.x == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.dp_part + (int64_t)a.B * npos * d) + w * 16;
    for (int t = 0; t < 16; ++t) dbg[t] = t < nstamp ? stamp[t] : 0ull;
  }
#endif
  // ---- dK, dV of this wave's keys
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int j = 32 * w + 16 * kt + x;
    if (j < nk) {
      bf16_t* kp = a.dk + (int64_t)b * a.k_sb + (int64_t)j * a.k_sr + h * DK;
      bf16_t* vp = a.dv + (int64_t)b * a.v_sb + (int64_t)j * a.v_sr + h * DK;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float k4[4] = {dk[kt][ct][0], dk[kt][ct][1], dk[kt][ct][2], dk[kt][ct][3]};
        const float v4[4] = {dv[kt][ct][0], dv[kt][ct][1], dv[kt][ct][2], dv[kt][ct][3]};
        st4_from_f32<bf16_t>(kp + 16 * ct + 4 * y, k4);
        st4_from_f32<bf16_t>(vp + 16 * ct + 4 * y, v4);
      }
    }
  }
  if constexpr (REL) {
  // ---- dp^T -> [n][64 channels] bf16 rows in LDS (the position image's place), then whole 128-byte rows to the partial table
  __syncthreads();  // every product is done with the position image
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = 16 * (w + 8 * nt) + x;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const uint2 v = make_uint2(bf16pack(dp[nt][ct][0], dp[nt][ct][1]), bf16pack(dp[nt][ct][2], dp[nt][ct][3]));
      *reinterpret_cast<uint2*>(lp + n * 128 + (((2 * ct + (y >> 1)) ^ (n & 7)) << 4) + (y & 1) * 8) = v;
    }
  }
  // column sums of the two dQ branches: 16 query lanes by shuffles, the two waves of a channel tile through LDS (the copy by key
  // is no longer read: the barrier above), one atomic per channel and branch into a replica of the workspace
  float* red = reinterpret_cast<float*>(lds + L_DA);  // [2][8 waves][64]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    su[r] = s2t_sum16_up(su[r]);
    sv[r] = s2t_sum16_up(sv[r]);
  }
  if (x == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[(0 * 8 + w) * 64 + 4 * y + r] = su[r];
      red[(1 * 8 + w) * 64 + 4 * y + r] = sv[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = tid + 512 * u;
    const int n = c >> 3, ch = c & 7;
    if (n < npos) *reinterpret_cast<uint4*>(out + (int64_t)n * d + ch * 8) = *reinterpret_cast<const uint4*>(lp + n * 128 + ((ch ^ (n & 7)) << 4));
  }
  if (tid < 128) {
    const int br = tid >> 6, c = tid & 63;
    const int ct = c >> 4, sl = c & 15;
    const float sum = red[(br * 8 + ct) * 64 + sl] + red[(br * 8 + ct + 4) * 64 + sl];
    const int64_t ro = (int64_t)(z % a.replicas) * a.replica_stride;
    atomicAdd((br ? a.dv_ : a.du) + ro + h * DK + c, sum);
  }
  }
}

}  // namespace

extern "C" int s2t_relpos_attn_bwd(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr,
                                   const void* v, int64_t v_sb, int64_t v_sr, const void* o, const void* dO, int64_t o_sb,
                                   int64_t o_sr, const float* lse, void* dq, void* dk, void* dv, const void* pos_p, int64_t p_sr,
                                   const float* pos_u, const float* pos_v, float* dpos_u, float* dpos_v, int replicas,
                                   int64_t replica_stride, void* dp_part, int B, int H, int T, int dk_dim, const int32_t* key_lens,
                                   float scale, float drop_p, const uint64_t* drop_seed, uint32_t drop_site, const int32_t* cu,
                                   void* stream) {
  if (!q || !k || !v || !o || !dO || !lse || !dq || !dk || !dv || !pos_p || !pos_u || !pos_v || !dpos_u || !dpos_v || !dp_part ||
      B <= 0 || H <= 0 || T <= 0 || replicas < 1)
    return S2T_ERR_ARG;
  if (dk_dim != DK || T > KMAX) return S2T_ERR_UNSUPPORTED;
  if (q_sr % 8 || k_sr % 8 || v_sr % 8 || o_sr % 8 || p_sr % 8 || q_sb % 8 || k_sb % 8 || v_sb % 8 || o_sb % 8) return S2T_ERR_ARG;
  if (((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)v % 16) || ((uintptr_t)o % 16) || ((uintptr_t)dO % 16) ||
      ((uintptr_t)dq % 16) || ((uintptr_t)dk % 16) || ((uintptr_t)dv % 16) || ((uintptr_t)pos_p % 16) || ((uintptr_t)dp_part % 16))
    return S2T_ERR_ALIGN;
  if (drop_p < 0.f || drop_p >= 1.f) return S2T_ERR_ARG;
  RpbArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr;
  a.o = (const bf16_t*)o; a.dO = (const bf16_t*)dO; a.o_sb = o_sb; a.o_sr = o_sr; a.lse = lse;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.pos_p = (const bf16_t*)pos_p; a.p_sr = p_sr;
  a.pos_u = pos_u; a.pos_v = pos_v; a.du = dpos_u; a.dv_ = dpos_v; a.replicas = replicas; a.replica_stride = replica_stride;
  a.dp_part = (bf16_t*)dp_part; a.B = B; a.H = H; a.T = T; a.Tk = T; a.key_lens = key_lens; a.scale = scale; a.drop_p = drop_p;
  a.drop_seed = drop_seed; a.drop_site = drop_site; a.cu = cu;
  const bool fast = (T & 1) == 0 && (uint64_t)B * H * (uint64_t)T * (uint64_t)(T >> 1) < (1ull << 32);
  hipStream_t st = (hipStream_t)stream;
  if (drop_p <= 0.f) hipLaunchKernelGGL((relpos_attn_bwd_kernel<0, true>), dim3(B * H), dim3(512), 0, st, a);
  else if (fast) hipLaunchKernelGGL((relpos_attn_bwd_kernel<1, true>), dim3(B * H), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((relpos_attn_bwd_kernel<2, true>), dim3(B * H), dim3(512), 0, st, a);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_attn_bwd_one_pass(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr, const void* v,
                                     int64_t v_sb, int64_t v_sr, const void* o, const void* dO, int64_t o_sb, int64_t o_sr,
                                     const float* lse, void* dq, void* dk, void* dv, int B, int H, int Tq, int Tk, int dk_dim,
                                     const int32_t* key_lens, int causal, float scale, float drop_p, const uint64_t* drop_seed,
                                     uint32_t drop_site, const int32_t* cu_q, const int32_t* cu_k, const void* o_lo, void* stream) {
  if (!q || !k || !v || !o || !dO || !lse || !dq || !dk || !dv || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return S2T_ERR_ARG;
  if (dk_dim != DK || Tk > KMAX) return S2T_ERR_UNSUPPORTED;
  if (q_sr % 8 || k_sr % 8 || v_sr % 8 || o_sr % 8 || q_sb % 8 || k_sb % 8 || v_sb % 8 || o_sb % 8) return S2T_ERR_ARG;
  if (((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)v % 16) || ((uintptr_t)o % 16) || ((uintptr_t)dO % 16) ||
      ((uintptr_t)dq % 16) || ((uintptr_t)dk % 16) || ((uintptr_t)dv % 16) || (o_lo && ((uintptr_t)o_lo % 16)))
    return S2T_ERR_ALIGN;
  if (drop_p < 0.f || drop_p >= 1.f) return S2T_ERR_ARG;
  RpbArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr;
  a.o = (const bf16_t*)o; a.dO = (const bf16_t*)dO; a.o_sb = o_sb; a.o_sr = o_sr; a.lse = lse;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.replicas = 1;
  a.B = B; a.H = H; a.T = Tq; a.Tk = Tk; a.causal = causal; a.o_lo = (const bf16_t*)o_lo; a.key_lens = key_lens; a.scale = scale;
  a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_site = drop_site; a.cu = cu_q; a.cu_k = cu_k;
  const bool fast = (Tk & 1) == 0 && (uint64_t)B * H * (uint64_t)Tq * (uint64_t)(Tk >> 1) < (1ull << 32);
  hipStream_t st = (hipStream_t)stream;
  if (drop_p <= 0.f) hipLaunchKernelGGL((relpos_attn_bwd_kernel<0, false>), dim3(B * H), dim3(512), 0, st, a);
  else if (fast) hipLaunchKernelGGL((relpos_attn_bwd_kernel<1, false>), dim3(B * H), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((relpos_attn_bwd_kernel<2, false>), dim3(B * H), dim3(512), 0, st, a);
  return S2T_LAUNCH_CHECK();
}
