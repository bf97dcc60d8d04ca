// Fused (flash-style) multi-head attention for gfx950, bf16 storage / fp32 softmax, head dim 64.
//
// Replaces, for the bf16 path, the GEMM-composed attention (batched QK^T GEMM -> softmax kernel -> batched PV GEMM,
// functional.AttentionFn) whose fp32 score tensors [B*h, T, T] dominated its cost: scores never leave the chip.
//   modules/multihead_attention.py:161-431          kind "abs": scores = (q*dk^-.5) k^T, key-pad / causal -> -inf
//   modules/espnet_multihead_attention.py:313-356   kind "rel": scores = ((q+u) k^T + (q+v) p[T-1-i+j]^T) / sqrt(dk),
//                                                   key-pad -> -inf, clamp(+-1e8), fp32 softmax
//
// Work decomposition: workgroup = 4 waves = 64 query rows of one (utterance, head); wave = 16 query rows; the key axis
// is walked in blocks of 64 keys staged in LDS (K and V images share one layout: 128-byte rows, 16-byte chunk c of row r
// at r*128 + ((c ^ (r&7))<<4); K is read row-wise with ds_read_b128, V column-wise with ds_read_b64_tr_b16 — both
// conflict-free).  All products are "swapped" MFMAs (16x16x32 bf16) producing S^T / O^T tiles, so that a lane owns ONE
// query row (lane&15) and the softmax statistics, the dropout mask, the O rescale and the P -> PV operand hand-over are
// lane-local (P's accumulator registers ARE the B-operand fragments of the PV product, k-order permuted consistently
// with the V fragments).
//
// Relative positions: bd[i][j] = (q_i+v) . p[T-1-i+j].  Per (16 queries x 64 keys) the 79 needed rows of p form a
// contiguous band; BD^T[n][q] for the band is produced by 10 MFMAs and re-indexed (the reference's rel_shift) through a
// small wave-private LDS scratch: element (q, key) reads band row 15 - q_local + key_local.
#include "common.h"

#ifndef S2T_ATT_DBG
#define S2T_ATT_DBG 0  // kernel-experiment switches (tools/dbg_variant.sh): 1 no dBD zero-fill, 2 no dBD stores, 4 no
                       // position band in the backward scores, 8 no dropout arithmetic in the backward
#endif

namespace {

constexpr int DK = 64;      // head dimension
constexpr int KB = 64;      // keys per block
constexpr int SC = 20;      // scratch row stride (floats) for the rel-shift band
constexpr int BAND = 80;    // band rows per (16 q x 64 keys)

typedef short s16x4v __attribute__((ext_vector_type(4)));

struct FusedArgs {
  const bf16_t *q, *k, *v;
  int64_t q_sb, q_sr, k_sb, k_sr, v_sb, v_sr;
  bf16_t* o;
  int64_t o_sb, o_sr;
  bf16_t* o_lo;  // optional, layout of o: the bf16-rounded REMAINDER of the fp32 output (o + o_lo carries 16 mantissa bits) —
                 // written by the forward, read by the backward's delta = rowsum(dO * O) (s2t_attn_fused_fwd in the header)
  float* lse;  // [Z][Tq]
  int B, H, Tq, Tk;
  const int32_t* key_lens;
  int causal;
  float scale;
  int rel;
  const bf16_t* pos_p;  // [2Tq-1][p_sr], head offset h*DK
  int64_t p_sr;
  const float *pos_u, *pos_v;  // [H*DK]
  float drop_p;
  const uint64_t* drop_seed;
  uint32_t drop_site;
  // backward
  const bf16_t* dO;  // layout of o
  const float* delta;  // [Z][Tq] rowsum(dO * O)
  bf16_t *dq, *dk, *dv;  // layouts of q, k, v
  bf16_t* dbd;           // [H][B][Tq][ldb] (rel): skewed dS for the position projections, may be null
  int64_t ldb;
  int dbd_band_only;     // the out-of-band part of dbd is already zero
  const bf16_t* pos_pt;  // (rel, optional) TRANSPOSED position projections: element (c, n) at pos_pt[(h*DK + c)*pt_ld + n],
  int64_t pt_ld;         // readable (zeros) for n in [-16, 2Tq-2 + 96]; with it dq receives the (Q+v) branch too and
  float *dpos_u, *dpos_v;  // the column sums of the two branches are added here ([H*DK] fp32 each)
  bf16_t* qv_out;        // (rel, optional) [B*Tq][H*DK]: Q + pos_bias_v as the kernels round it, for the position-table gradient
  // Packed batch (include/s2t_hip.h): rows of utterance b = cu[b] .. cu[b + 1] of the query / key side.  nq / nk: the rows a
  // kernel walks and stores (the host sets Tq / Tk; bind_utt() replaces them by the utterance's own row counts).  Tq / Tk
  // stay the padded lengths: the centre of the position table, the strides of lse / delta / dbd / qv_out and the index space
  // of the dropout mask do not move with the fill of a batch.
  const int32_t *cu_q, *cu_k;
  int nq, nk;
};

// this workgroup's utterance: row counts and base pointers of a packed batch (pointers are moved so that the uniform-layout
// expression  base + b * X_sb + row * X_sr  lands on row cu[b] + row)
__device__ __forceinline__ void bind_utt(FusedArgs& a, int b) {
  if (a.cu_q) {
    const int r0 = a.cu_q[b];
    a.nq = a.cu_q[b + 1] - r0;
    const int64_t oq = (int64_t)r0 * a.q_sr - (int64_t)b * a.q_sb, oo = (int64_t)r0 * a.o_sr - (int64_t)b * a.o_sb;
    a.q += oq;
    if (a.dq) a.dq += oq;
    a.o += oo;
    if (a.o_lo) a.o_lo += oo;
    if (a.dO) a.dO += oo;
  }
  if (a.cu_k) {
    const int r0 = a.cu_k[b];
    a.nk = a.cu_k[b + 1] - r0;
    const int64_t ok = (int64_t)r0 * a.k_sr - (int64_t)b * a.k_sb, ov = (int64_t)r0 * a.v_sr - (int64_t)b * a.v_sb;
    a.k += ok;
    a.v += ov;
    if (a.dk) a.dk += ok;
    if (a.dv) a.dv += ov;
  }
}

__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// 16 random bits of each of the four (query, key .. key + 3) elements starting at element index `base` (= row base + a
// multiple of 4).  `fast`: even Tk and an index space below 2^32 (wave-uniform): the row base is then even and the 32-bit
// form needs neither 64-bit arithmetic nor an odd-start variant — the same bits either way.
__device__ __forceinline__ void attn_rand4(uint64_t key, uint64_t base, bool fast, uint32_t (&r16)[4]) {
  if (fast) s2t_rand_run_even32<4>(key, (uint32_t)base, r16);
  else s2t_rand_run<4>(key, base, r16);
}
__device__ __forceinline__ bool attn_fast_mask(const FusedArgs& a) {
  return (a.Tk & 1) == 0 && (uint64_t)a.B * a.H * (uint64_t)a.Tq * (uint64_t)a.Tk < (1ull << 32);
}

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// add a per-column fp32 bias to 8 bf16 values (q + pos_bias_u / q + pos_bias_v), round to bf16
__device__ __forceinline__ uint4 add_bias8(uint4 qv, const float* __restrict__ bias) {
  const uint32_t w[4] = {qv.x, qv.y, qv.z, qv.w};
  uint32_t o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float a = __uint_as_float(w[t] << 16) + bias[2 * t];
    const float b = __uint_as_float(w[t] & 0xffff0000u) + bias[2 * t + 1];
    o[t] = bf16pack(a, b);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// A 64-row x 64-col bf16 tile travels global -> registers -> LDS in two steps so that the loads of the NEXT tile are in
// flight while the current one is being multiplied (sequences are short: 4 key blocks at T' = 250, so an unhidden load
// round trip per block is a large part of the kernel).  Loads are unconditional (rows clamped); rows beyond nrows are
// zeroed, and an optional per-column fp32 bias (q + pos_bias_u / q + pos_bias_v) is added, on the way into the LDS.
struct TileRegs {
  uint4 v[2];
};
__device__ __forceinline__ void tile_load(TileRegs& t, const bf16_t* __restrict__ base, int64_t sr, int row0, int nrows,
                                          int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = tid + 256 * u;
    // (an utterance of a packed batch may hold no row at all: the clamp then lands on row 0, never on -1 — which the 24-bit
    // multiply below would turn into 0xFFFFFF rows)
    const int r = max(min(row0 + (c >> 3), nrows - 1), 0), ch = c & 7;
    // (row < 2^16, row stride < 2^16 elements — the entry points check — so the offset is ONE full-rate 24-bit multiply; the
    // 64-bit form was two quarter-rate 32-bit multiplies and a 64-bit multiply-add per load, 24 such per key block in the backward
    // kernels, which are bound by vector issue)
    t.v[u] = ldg16(base + (__umul24((uint32_t)r, (uint32_t)sr) + (uint32_t)(ch * 8)));
  }
}
__device__ __forceinline__ void tile_store(char* lds, const TileRegs& t, int row0, int nrows, const float* __restrict__ bias,
                                           int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = tid + 256 * u;
    const int r = c >> 3, ch = c & 7;
    uint4 v = t.v[u];
    if (row0 + r >= nrows) v = make_uint4(0, 0, 0, 0);
    else if (bias) v = add_bias8(v, bias + ch * 8);
    *reinterpret_cast<uint4*>(lds + r * 128 + ((ch ^ (r & 7)) << 4)) = v;
  }
}

// row-wise fragment (A operand, rows = tile rows): 8 consecutive k of row (blk*16 + x), k-step ks
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int blk, int ks, int x, int y) {
  const int r = blk * 16 + x;
  const int c = ks * 4 + y;
  return as_frag(*reinterpret_cast<const uint4*>(lds + r * 128 + ((c ^ (r & 7)) << 4)));
}

// column-wise fragment (A operand = tile^T): rows of the product = tile columns (cblk*16 + x), k = tile rows in the
// permuted order kappa(y, j) = 32*s + 16*(j>>2) + 4*y + (j&3)  (matches P's accumulator registers)
__device__ __forceinline__ bf16x8 frag_cols_perm(const char* lds, int cblk, int s, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = 32 * s + 16 * h + 4 * y + qq;
    const int chunk = 2 * cblk + (p >> 1);
    const char* a = lds + R * 128 + ((chunk ^ (R & 7)) << 4) + (p & 1) * 8;
    s16x4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a));
    const uint2 tt = __builtin_bit_cast(uint2, t);
    w[2 * h] = tt.x;
    w[2 * h + 1] = tt.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}

// column-wise fragment in NATURAL k order: k = 32*s + 8*y + j (tile rows), product rows = tile columns
__device__ __forceinline__ bf16x8 frag_cols_nat(const char* lds, int cblk, int s, int x, int y) {
  const int qq = x >> 2, p = x & 3;
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = 32 * s + 8 * y + 4 * h + qq;
    const int chunk = 2 * cblk + (p >> 1);
    const char* a = lds + R * 128 + ((chunk ^ (R & 7)) << 4) + (p & 1) * 8;
    s16x4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a));
    const uint2 tt = __builtin_bit_cast(uint2, t);
    w[2 * h] = tt.x;
    w[2 * h + 1] = tt.y;
  }
  return as_frag(make_uint4(w[0], w[1], w[2], w[3]));
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// pack 8 fp32 -> bf16x8 fragment
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  uint32_t o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = bf16pack(v[2 * t], v[2 * t + 1]);
  return as_frag(make_uint4(o[0], o[1], o[2], o[3]));
}

// ---- S^T tile block: st[kt][r] = score(q = q0w + x, key = k0 + 16*kt + 4*y + r), scaled and masked -------------------
struct QFrags {
  bf16x8 qa[2];   // q (abs) or q+u (rel), k-steps 0/1
  bf16x8 qv[2];   // q+v (rel only)
};

// PRE: where the position rows come from — 0 global memory on the spot (experiment switch), 2 the workgroup's LDS image of
// the 128 rows its four waves need (ptile_*; wave w's band tile nt is image tile pblk0 + nt)
// The 128 position rows the four waves of a (64-query x 64-key) block need — nb3 + (0..127) with nb3 = Tq-1-(q0+63)+k0,
// wave w's 80 rows start 48 - 16 w further — as ONE workgroup tile: whole 128-byte head slices by coalesced loads (a
// fragment load straight from global memory is sixteen 64-byte strided requests per instruction, and ten of those per wave
// and key block were a large part of these kernels), global -> registers -> LDS like the K / V tiles, same LDS layout.
struct PTile {
  uint4 v[4];
};
__device__ __forceinline__ void ptile_load(const FusedArgs& a, PTile& t, int h, int q0, int k0, int tid) {
  const int nb3 = a.Tq - 1 - (q0 + 63) + k0;
  const int nmax = 2 * a.Tq - 2;
  const bf16_t* pp = a.pos_p + h * DK;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = tid + 256 * u;
    int n = nb3 + (c >> 3);
    n = n < 0 ? 0 : (n > nmax ? nmax : n);
    t.v[u] = ldg16(pp + (__umul24((uint32_t)n, (uint32_t)a.p_sr) + (uint32_t)((c & 7) * 8)));
  }
}
__device__ __forceinline__ void ptile_store(char* lp, const PTile& t, int tid) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = tid + 256 * u;
    const int r = c >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(lp + r * 128 + ((ch ^ (r & 7)) << 4)) = t.v[u];
  }
}

// LOG2: the scores leave multiplied by scale * log2(e) (the forward kernels' softmax runs on v_exp_f32 as it is; both forward
// kernels — this block form and the resident form — do the same arithmetic, so which of them runs changes no bit)
template <bool REL, int PRE = 0, bool LOG2 = false>
__device__ __forceinline__ void scores_block(const FusedArgs& a, const QFrags& qf, const char* lk, float* scratch,
                                             int h, int q0w, int k0, int klen, int x, int y, f32x4 (&st)[4],
                                             const char* lp = nullptr, int pblk0 = 0) {
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = mfma16(frag_rows(lk, kt, ks, x, y), qf.qa[ks], acc);
    st[kt] = acc;
  }
  if constexpr (REL) {
    // band of position rows n = nbase + (0..79), nbase = Tq-1-(q0w+15)+k0
    const int nbase = a.Tq - 1 - (q0w + 15) + k0;
    const int nmax = 2 * a.Tq - 2;
    const bf16_t* pp = a.pos_p + h * DK;
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      int n = nbase + 16 * nt + x;
      n = n < 0 ? 0 : (n > nmax ? nmax : n);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 pf;
        if constexpr (PRE == 2) pf = frag_rows(lp, pblk0 + nt, ks, x, y);  // compile-time choice: a load under a run-time
        else pf = as_frag(ldg16(pp + (int64_t)n * a.p_sr + (ks * 4 + y) * 8));  // test is waited for on the spot
        acc = mfma16(pf, qf.qv[ks], acc);
      }
      // lane (x = q, y) holds band rows 16nt + 4y + r
#pragma unroll
      for (int r = 0; r < 4; ++r) scratch[(16 * nt + 4 * y + r) * SC + x] = acc[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[kt][r] += scratch[(15 - x + 16 * kt + 4 * y + r) * SC + x];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // (bitwise predicates and a select: with short-circuit conditions the compiler builds an exec-mask branch per element,
  // 16 per block and wave)
  const int i = q0w + x;
  const float sc = LOG2 ? a.scale * 1.44269504088896f : a.scale;
  if (!a.causal && k0 + KB <= klen) {  // the block lies wholly inside the utterance: no per-element mask (wave-uniform branch)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[kt][r] *= sc;
    return;
  }
  const int jlim = a.causal ? min(klen, i + 1) : klen;  // keys j >= jlim are masked for this lane's query
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = k0 + 16 * kt + 4 * y + r;
      const float s = st[kt][r] * sc;
      st[kt][r] = j >= jlim ? -INFINITY : s;
    }
}

__device__ __forceinline__ void load_qfrags(const FusedArgs& a, QFrags& qf, int b, int h, int i, int y, bool rel) {
  const int ic = i < a.nq ? i : a.nq - 1;
  const bf16_t* qp = a.q + (int64_t)b * a.q_sb + (int64_t)ic * a.q_sr + h * DK;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint4 raw = ldg16(qp + (ks * 4 + y) * 8);
    if (rel) {
      qf.qa[ks] = as_frag(add_bias8(raw, a.pos_u + h * DK + (ks * 4 + y) * 8));
      qf.qv[ks] = as_frag(add_bias8(raw, a.pos_v + h * DK + (ks * 4 + y) * 8));
    } else {
      qf.qa[ks] = as_frag(raw);
      qf.qv[ks] = as_frag(raw);
    }
  }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <bool REL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const FusedArgs a_in) {
  FusedArgs a = a_in;
  bind_utt(a, blockIdx.x / a_in.H);
  if ((int)blockIdx.y * 64 >= a.nq) return;  // packed batch: no row of this utterance in the query block
  __shared__ __attribute__((aligned(16))) char lds[16384 + 4 * BAND * SC * 4 + (REL ? 16384 : 0)];
  char* lk = lds;
  char* lv = lds + 8192;
  char* lp = lds + 16384 + 4 * BAND * SC * 4;  // (REL) the block's 128 position rows
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int x = lane & 15, y = lane >> 4;
  float* scratch = reinterpret_cast<float*>(lds + 16384) + w * BAND * SC;
  const int z = blockIdx.x;  // (utterance, head) fastest: the row blocks of one (b, h) share an XCD's L2 (K/V re-reads)
  const int b = z / a.H, h = z % a.H;
  const int q0 = blockIdx.y * 64;
  const int q0w = q0 + 16 * w;
  const int i = q0w + x;
  const int klen = a.key_lens ? min(a.key_lens[b], a.nk) : a.nk;

  QFrags qf;
  load_qfrags(a, qf, b, h, i, y, REL);

  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;

  const bf16_t* kb = a.k + (int64_t)b * a.k_sb + h * DK;
  const bf16_t* vb = a.v + (int64_t)b * a.v_sb + h * DK;
  // key blocks that hold only masked keys contribute exp(-inf) = 0 to every row: not walked (the same bits)
  int kend = min(a.nk, (klen + KB - 1) / KB * KB);
  if (a.causal) kend = min(kend, q0 + 64);  // keys beyond the last query of the workgroup are masked for all its rows
  const uint64_t dkey = a.drop_p > 0.f ? s2t_drop_key(a.drop_seed, a.drop_site) : 0ull;
  const uint32_t dth = s2t_drop_thresh(a.drop_p);
  const float dinv = s2t_drop_scale(a.drop_p);

  TileRegs tk, tv;
  tile_load(tk, kb, a.k_sr, 0, a.nk, tid);
  tile_load(tv, vb, a.v_sr, 0, a.nk, tid);
  PTile tp;  // position rows of the next block (travels like the K / V tiles)
  if constexpr (REL) ptile_load(a, tp, h, q0, 0, tid);
  for (int k0 = 0; k0 < kend; k0 += KB) {
    __syncthreads();
    tile_store(lk, tk, k0, a.nk, nullptr, tid);
    tile_store(lv, tv, k0, a.nk, nullptr, tid);
    if constexpr (REL) ptile_store(lp, tp, tid);
    __syncthreads();
    if (k0 + KB < kend) {  // next block's K/V (and position rows) in flight during this block's MFMAs
      tile_load(tk, kb, a.k_sr, k0 + KB, a.nk, tid);
      tile_load(tv, vb, a.v_sr, k0 + KB, a.nk, tid);
      if constexpr (REL) ptile_load(a, tp, h, q0, k0 + KB, tid);
    }
    f32x4 st[4];
    scores_block<REL, REL ? 2 : 0, true>(a, qf, lk, scratch, h, q0w, k0, klen, x, y, st, lp, 3 - w);  // log2 domain
    // ---- online softmax (row = lane's query; its 16 keys in registers, the other 48 in the 3 other y-groups)
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = s2t_xmax<16>(mx);
    mx = s2t_xmax<32>(mx);
    const float mn = fmaxf(m, mx);
    // branch-free: exp(-inf) = 0 covers masked keys and the first block (m = -inf: alpha = 0 scales l = 0 and O = 0); a row
    // that has seen no valid key yet (mn = -inf) subtracts 0 instead
    const float mref = (mn == -INFINITY) ? 0.f : mn;
    const float alpha = __builtin_amdgcn_exp2f(m - mref);
    float rs = 0.f;
    float pr[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[kt][r] - mref);
        rs += p;
        pr[kt][r] = p;
      }
    rs = s2t_xadd<16>(rs);
    rs = s2t_xadd<32>(rs);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
    if (a.drop_p > 0.f) {
      const uint64_t rowbase = ((uint64_t)z * a.Tq + (uint64_t)(i < a.nq ? i : 0)) * (uint64_t)a.Tk;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        uint32_t r16[4];
        attn_rand4(dkey, rowbase + (uint64_t)(k0 + 16 * kt + 4 * y), attn_fast_mask(a), r16);
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[kt][r] = r16[r] >= dth ? pr[kt][r] * dinv : 0.f;
      }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float pv8[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv8[r] = pr[2 * s][r];
        pv8[4 + r] = pr[2 * s + 1][r];
      }
      const bf16x8 pf = pack8(pv8);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(frag_cols_perm(lv, dt, s, x, y), pf, o[dt]);
    }
  }
  // ---- epilogue: O = O^T / l ; lane (x = query, y) holds d = 16dt + 4y + r
  if (i < a.nq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* op = a.o + (int64_t)b * a.o_sb + (int64_t)i * a.o_sr + h * DK;
    bf16_t* lo = a.o_lo ? a.o_lo + (int64_t)b * a.o_sb + (int64_t)i * a.o_sr + h * DK : nullptr;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float v4[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
      st4_from_f32<bf16_t>(op + 16 * dt + 4 * y, v4);
      if (lo) {  // what the bf16 rounding of O dropped, itself rounded to bf16
        float r4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) r4[r] = v4[r] - bf2f(f2bf(v4[r]));
        st4_from_f32<bf16_t>(lo + 16 * dt + 4 * y, r4);
      }
    }
    if (y == 0 && a.lse) a.lse[(int64_t)z * a.Tq + i] = (l > 0.f) ? (m + __log2f(l)) * 0.693147180559945f : -INFINITY;
  }
}

// =====================================================================================================================
// forward, resident form (round 3): ONE workgroup per (utterance, head) for Tq = Tk <= 256 with relative positions — the
// encoder self-attention of the recipes (T' = 250).  K, V and ALL 2T-1 projected position rows of the head are staged in LDS
// once (128 KiB) instead of once per 64-query block (four times at T' = 250, each behind two workgroup barriers and a
// global round trip), the eight waves then walk their query tiles (wave w: tiles w and w + 8) over the resident images with
// no barrier and no global load in the loop.  Arithmetic, masks and dropout bits are those of attn_fwd_kernel.
// The rel-shift scratch is 48 band rows per wave (3.75 KiB: eight waves beside the 128 KiB of images): the 80-row band of
// a (16 query x 64 key) block goes through it in two overlapping halves (rows 0..47 for keys 0..31, rows 32..79 for keys
// 32..63; the shared tile of 16 rows stays in registers between the halves).
// =====================================================================================================================
constexpr int BH_MAXT = 256;
constexpr int BH_BAND = 48;
constexpr int BH_IMG = 4 * BH_MAXT * 128;              // K | V | 2 * BH_MAXT position rows
constexpr int BH_LDS = BH_IMG + 8 * BH_BAND * SC * 4;  // 161 792 bytes

// S^T tiles of (16 queries of this wave) x (64 keys from k0) from the resident images, scaled and masked (scores_block's
// arithmetic)
__device__ __forceinline__ void scores_block_res(const FusedArgs& a, const QFrags& qf, const char* lk, const char* lp,
                                                 float* scratch, int q0w, int k0, int klen, int x, int y, f32x4 (&st)[4]) {
  const char* lkb = lk + k0 * 128;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = mfma16(frag_rows(lkb, kt, ks, x, y), qf.qa[ks], acc);
    st[kt] = acc;
  }
  {
    const int nbase = a.Tq - 1 - (q0w + 15) + k0;
    const int nmax = 2 * a.Tq - 2;
    f32x4 band[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      int n = nbase + 16 * nt + x;
      n = n < 0 ? 0 : (n > nmax ? nmax : n);
      const char* row = lp + n * 128;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        acc = mfma16(as_frag(*reinterpret_cast<const uint4*>(row + (((ks * 4 + y) ^ (n & 7)) << 4))), qf.qv[ks], acc);
      band[nt] = acc;
    }
    // lane (x = q, y) holds band rows 16 nt + 4 y + r; element (q, key) reads band row 15 - q + key
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[(16 * t + 4 * y + r) * SC + x] = band[2 * half + t][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[2 * half + kk][r] += scratch[(15 - x + 16 * kk + 4 * y + r) * SC + x];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  // scores leave in the LOG2 domain (scale * log2(e) folded into one multiply: the softmax then takes v_exp_f32 as it is);
  // a block that lies wholly inside the utterance skips the per-element mask (wave-uniform branch)
  const float c2 = a.scale * 1.44269504088896f;
  if (k0 + KB <= klen) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[kt][r] *= c2;
  } else {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = k0 + 16 * kt + 4 * y + r;
        const float s = st[kt][r] * c2;
        st[kt][r] = j >= klen ? -INFINITY : s;
      }
  }
}

// K, V (rows >= Tk zero) and the position rows (n clamped to 2Tq-2) of (b, h) into the resident images: 128-byte rows,
// 16-byte chunk c of row r at r*128 + ((c ^ (r & 7)) << 4)
__device__ __forceinline__ void bh_stage_images(const FusedArgs& a, char* lds, int b, int h, int tid) {
  const bf16_t* kb = a.k + (int64_t)b * a.k_sb + h * DK;
  const bf16_t* vb = a.v + (int64_t)b * a.v_sb + h * DK;
  const bf16_t* pp = a.pos_p + h * DK;
  const int nmax = 2 * a.Tq - 2;
  uint4 t[16];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = tid + 512 * u;
    const int r = min(c >> 3, a.nk - 1), ch = c & 7;
    t[u] = ldg16(kb + (int64_t)r * a.k_sr + ch * 8);
    t[4 + u] = ldg16(vb + (int64_t)r * a.v_sr + ch * 8);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = tid + 512 * u;
    const int n = min(c >> 3, nmax), ch = c & 7;
    t[8 + u] = ldg16(pp + (int64_t)n * a.p_sr + ch * 8);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = tid + 512 * u;
    const int r = c >> 3, ch = c & 7;
    const bool ok = r < a.nk;
    *reinterpret_cast<uint4*>(lds + r * 128 + ((ch ^ (r & 7)) << 4)) = ok ? t[u] : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(lds + BH_MAXT * 128 + r * 128 + ((ch ^ (r & 7)) << 4)) = ok ? t[4 + u] : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = tid + 512 * u;
    const int r = c >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(lds + 2 * BH_MAXT * 128 + r * 128 + ((ch ^ (r & 7)) << 4)) = t[8 + u];
  }
}

__global__ __launch_bounds__(512, 2) void attn_bh_fwd_kernel(const FusedArgs a_in) {
  FusedArgs a = a_in;
  bind_utt(a, blockIdx.x / a_in.H);
  if (a.nq <= 0 || a.nk <= 0) return;  // packed batch: an utterance without rows (nothing to read, nothing to store)
  __shared__ __attribute__((aligned(16))) char lds[BH_LDS];
  char* lk = lds;
  char* lv = lds + BH_MAXT * 128;
  char* lp = lds + 2 * BH_MAXT * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int x = lane & 15, y = lane >> 4;
  float* scratch = reinterpret_cast<float*>(lds + BH_IMG) + w * BH_BAND * SC;
  const int z = blockIdx.x;
  const int b = z / a.H, h = z % a.H;
  const int klen = a.key_lens ? min(a.key_lens[b], a.nk) : a.nk;
  // key blocks that hold only masked keys contribute exp(-inf) = 0 to every row: not walked (the same bits)
  const int kend = min(a.nk, (klen + KB - 1) / KB * KB);
  bh_stage_images(a, lds, b, h, tid);
  const uint64_t dkey = a.drop_p > 0.f ? s2t_drop_key(a.drop_seed, a.drop_site) : 0ull;
  const uint32_t dth = s2t_drop_thresh(a.drop_p);
  const float dinv = s2t_drop_scale(a.drop_p);
  __syncthreads();
  for (int q0w = 16 * w; q0w < a.nq; q0w += 128) {
    const int i = q0w + x;
    QFrags qf;
    load_qfrags(a, qf, b, h, i, y, true);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < kend; k0 += KB) {
      f32x4 st[4];
      scores_block_res(a, qf, lk, lp, scratch, q0w, k0, klen, x, y, st);
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
      mx = s2t_xmax<16>(mx);
      mx = s2t_xmax<32>(mx);
      const float mn = fmaxf(m, mx);
      const float mref = (mn == -INFINITY) ? 0.f : mn;
      const float alpha = __builtin_amdgcn_exp2f(m - mref);
      float rs = 0.f;
      float pr[4][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[kt][r] - mref);
          rs += p;
          pr[kt][r] = p;
        }
      rs = s2t_xadd<16>(rs);
      rs = s2t_xadd<32>(rs);
      l = l * alpha + rs;
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
      if (a.drop_p > 0.f) {
        const uint64_t rowbase = ((uint64_t)z * a.Tq + (uint64_t)(i < a.nq ? i : 0)) * (uint64_t)a.Tk;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          uint32_t r16[4];
          attn_rand4(dkey, rowbase + (uint64_t)(k0 + 16 * kt + 4 * y), attn_fast_mask(a), r16);
#pragma unroll
          for (int r = 0; r < 4; ++r) pr[kt][r] = r16[r] >= dth ? pr[kt][r] * dinv : 0.f;
        }
      }
      const char* lvb = lv + k0 * 128;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float pv8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pv8[r] = pr[2 * s][r];
          pv8[4 + r] = pr[2 * s + 1][r];
        }
        const bf16x8 pf = pack8(pv8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(frag_cols_perm(lvb, dt, s, x, y), pf, o[dt]);
      }
    }
    if (i < a.nq) {
      const float inv = l > 0.f ? 1.f / l : 0.f;
      bf16_t* op = a.o + (int64_t)b * a.o_sb + (int64_t)i * a.o_sr + h * DK;
      bf16_t* lo = a.o_lo ? a.o_lo + (int64_t)b * a.o_sb + (int64_t)i * a.o_sr + h * DK : nullptr;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float v4[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
        st4_from_f32<bf16_t>(op + 16 * dt + 4 * y, v4);
        if (lo) {
          float r4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) r4[r] = v4[r] - bf2f(f2bf(v4[r]));
          st4_from_f32<bf16_t>(lo + 16 * dt + 4 * y, r4);
        }
      }
      // (m and the scores are in the log2 domain here; the backward kernels take the natural-log row statistic)
      if (y == 0 && a.lse) a.lse[(int64_t)z * a.Tq + i] = (l > 0.f) ? (m + __log2f(l)) * 0.693147180559945f : -INFINITY;
    }
  }
}

// =====================================================================================================================
// backward
// =====================================================================================================================
// ---- dQ: workgroup = 64 queries of one (b,h), wave = 16 queries; walks the key blocks ------------------------------
template <bool REL, bool FUSEV = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const FusedArgs a_in) {
  FusedArgs a = a_in;
  bind_utt(a, blockIdx.x / a_in.H);
  if ((int)blockIdx.y * 64 >= a.nq) return;  // packed batch: no row of this utterance in the query block
  __shared__ __attribute__((aligned(16))) char lds[16384 + 4 * BAND * SC * 4 + (REL ? 16384 : 0)];
  char* lp = lds + 16384 + 4 * BAND * SC * 4;  // (REL) the block's 128 position rows (ptile_*)
  char* lk = lds;
  char* lv = lds + 8192;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int x = lane & 15, y = lane >> 4;
  float* scratch = reinterpret_cast<float*>(lds + 16384) + w * BAND * SC;
  const int z = blockIdx.x;  // (utterance, head) fastest: the row blocks of one (b, h) share an XCD's L2 (K/V re-reads)
  const int b = z / a.H, h = z % a.H;
  const int q0 = blockIdx.y * 64;
  const int q0w = q0 + 16 * w;
  const int i = q0w + x;
  const int ic = i < a.nq ? i : a.nq - 1;
  const int klen = a.key_lens ? min(a.key_lens[b], a.nk) : a.nk;

#if S2T_ATT_DBG & 64
  unsigned long long stamp[40];
  int nstamp = 0;
#define ASTAMP()                                         \
  do {                                                   \
    __builtin_amdgcn_sched_barrier(0);                   \
    if (nstamp < 40) stamp[nstamp++] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#else
#define ASTAMP()
#endif
  ASTAMP();
  QFrags qf;
  load_qfrags(a, qf, b, h, i, y, REL);
  if (REL && a.qv_out && i < a.nq) {  // this lane's two 16-byte pieces of (q + v) of its query row
    bf16_t* qo = a.qv_out + ((int64_t)b * a.Tq + i) * (a.H * DK) + h * DK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) *reinterpret_cast<uint4*>(qo + (ks * 4 + y) * 8) = __builtin_bit_cast(uint4, qf.qv[ks]);
  }
  bf16x8 dof[2];
  {
    const bf16_t* dp = a.dO + (int64_t)b * a.o_sb + (int64_t)ic * a.o_sr + h * DK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dof[ks] = as_frag(ldg16(dp + (ks * 4 + y) * 8));
  }
  const float lse_i = a.lse[(int64_t)z * a.Tq + ic];
  // delta_i = sum_c dO[i][c] * O[i][c]: the four lanes of a query row hold a quarter of it each (the dO fragments);
  // written out for the dK / dV kernel that follows.  dS = P (dP - delta) CANCELS where the probabilities are nearly uniform
  // (the decoder's encoder-decoder attention over 250 keys): delta from the bf16-ROUNDED O then carries an error of the size of
  // the difference (the composed path sums P dP in fp32).  With o_lo — what the rounding of O dropped — the sum is taken on
  // O + o_lo, 16 mantissa bits.
  float del_i;
  {
    const bf16_t* op = a.o + (int64_t)b * a.o_sb + (int64_t)ic * a.o_sr + h * DK;
    const bf16_t* lp_ = a.o_lo ? a.o_lo + (int64_t)b * a.o_sb + (int64_t)ic * a.o_sr + h * DK : nullptr;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 ov = ldg16(op + (ks * 4 + y) * 8);
      const uint4 dv = __builtin_bit_cast(uint4, dof[ks]);
      const uint32_t ow[4] = {ov.x, ov.y, ov.z, ov.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
      if (lp_) {
        const uint4 lv_ = ldg16(lp_ + (ks * 4 + y) * 8);
        const uint32_t lw[4] = {lv_.x, lv_.y, lv_.z, lv_.w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          part += (__uint_as_float(ow[t] << 16) + __uint_as_float(lw[t] << 16)) * __uint_as_float(dw[t] << 16) +
                  (__uint_as_float(ow[t] & 0xffff0000u) + __uint_as_float(lw[t] & 0xffff0000u)) * __uint_as_float(dw[t] & 0xffff0000u);
        continue;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        part += __uint_as_float(ow[t] << 16) * __uint_as_float(dw[t] << 16) +
                __uint_as_float(ow[t] & 0xffff0000u) * __uint_as_float(dw[t] & 0xffff0000u);
    }
    part = s2t_xadd<16>(part);
    part = s2t_xadd<32>(part);
    del_i = part;
    if (y == 0 && i < a.nq) const_cast<float*>(a.delta)[(int64_t)z * a.Tq + i] = part;
  }

  f32x4 dq[4], dqv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = dqv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr bool fuse_v = REL && FUSEV;  // a separate instantiation: its code costs the plain form 25 us when merely present

  const bf16_t* kb = a.k + (int64_t)b * a.k_sb + h * DK;
  const bf16_t* vb = a.v + (int64_t)b * a.v_sb + h * DK;
  int kend = a.nk;
  if (a.causal) kend = min(a.nk, q0 + 64);
  const uint64_t dkey = a.drop_p > 0.f ? s2t_drop_key(a.drop_seed, a.drop_site) : 0ull;
  const uint32_t dth = s2t_drop_thresh(a.drop_p);
  const float dinv = s2t_drop_scale(a.drop_p);
  if (REL && a.dbd && !a.dbd_band_only && !(S2T_ATT_DBG & 1)) {
    // zero the part of each of this wave's rows that lies outside the band n in [Tq-1-i, Tq-1-i+Tk)
    for (int rr = 0; rr < 16; ++rr) {
      const int ii = q0w + rr;
      if (ii >= a.nq) break;
      bf16_t* row = a.dbd + (((int64_t)h * a.B + b) * a.Tq + ii) * a.ldb;
      const int lo = a.Tq - 1 - ii, hi = lo + a.nk;
      for (int n = lane; n < a.ldb; n += 64)
        if (n < lo || n >= hi) row[n] = 0;
    }
  }

  TileRegs tk, tv;
  tile_load(tk, kb, a.k_sr, 0, a.nk, tid);
  tile_load(tv, vb, a.v_sr, 0, a.nk, tid);
  constexpr bool BAND_ON = REL && !(S2T_ATT_DBG & 4);
  PTile tp;  // position rows of the next block (travels like the K / V tiles)
  constexpr bool PRE_ON = BAND_ON && !(S2T_ATT_DBG & 128);
  if constexpr (PRE_ON) ptile_load(a, tp, h, q0, 0, tid);
  ASTAMP();
  for (int k0 = 0; k0 < kend; k0 += KB) {
    __syncthreads();
    ASTAMP();
    tile_store(lk, tk, k0, a.nk, nullptr, tid);
    tile_store(lv, tv, k0, a.nk, nullptr, tid);
    if constexpr (PRE_ON) ptile_store(lp, tp, tid);
    __syncthreads();
    ASTAMP();
    if (k0 + KB < kend) {  // next block's K/V (and position rows) in flight during this block's MFMAs
      tile_load(tk, kb, a.k_sr, k0 + KB, a.nk, tid);
      tile_load(tv, vb, a.v_sr, k0 + KB, a.nk, tid);
      if constexpr (PRE_ON) ptile_load(a, tp, h, q0, k0 + KB, tid);
    }
    f32x4 st[4];
    scores_block<BAND_ON, PRE_ON ? 2 : 0>(a, qf, lk, scratch, h, q0w, k0, klen, x, y, st, lp, 3 - w);
    ASTAMP();
    // dP^T[key][q] = V[key] . dO[q]
    f32x4 dpt[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc = mfma16(frag_rows(lv, kt, ks, x, y), dof[ks], acc);
      dpt[kt] = acc;
    }
    const uint64_t rowbase = ((uint64_t)z * a.Tq + (uint64_t)ic) * (uint64_t)a.Tk;
    const float lse_ref = (lse_i == -INFINITY) ? 0.f : lse_i;  // a row without a valid key: every p is exp(-inf - 0) = 0
    float ds[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      uint32_t r16[4] = {65535u, 65535u, 65535u, 65535u};
      if (a.drop_p > 0.f) attn_rand4(dkey, rowbase + (uint64_t)(k0 + 16 * kt + 4 * y), attn_fast_mask(a), r16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[kt][r] - lse_ref);  // masked keys: exp(-inf) = 0, no branch per element
        float dp = dpt[kt][r];
        if (a.drop_p > 0.f) dp = r16[r] >= dth ? dp * dinv : 0.f;
        ds[kt][r] = p * (dp - del_i) * a.scale;
      }
    }
    ASTAMP();
    constexpr int DSS = 68;
    if (REL && (a.dbd || fuse_v)) {
      // the wave's 16 x 64 dS tile in its LDS scratch, [query][key], stride DSS floats
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        *reinterpret_cast<f32x4*>(scratch + x * DSS + 16 * kt + 4 * y) = (f32x4){ds[kt][0], ds[kt][1], ds[kt][2], ds[kt][3]};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (fuse_v) {
      // (Q+v) branch: dQv^T[c][q] += sum_n P^T[c][n] band[n][q], band[n][q] = dS[q][key] at n = Tq-1-i+key — over the
      // 96-wide window of positions starting at n0a (the block's lowest position, rounded down to a 16-byte boundary of the
      // transposed table): lane (q = x, y) assembles its B fragments from its dS row, shifted by sh = (15 - x) + (nbase - n0a)
      const int nbase = a.Tq - 1 - (q0w + 15) + k0;
      const int n0a = nbase & ~7;
      const int sh = 15 - x + (nbase - n0a);
      const bf16_t* pt = a.pos_pt + (int64_t)(h * DK + x) * a.pt_ld + n0a + 8 * y;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        float b8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int kl = 32 * ks + 8 * y + t - sh;
#if S2T_ATT_DBG & 32
          const float v = ds[t & 3][t >> 2];
#else
          const float v = scratch[x * DSS + (kl < 0 ? 0 : (kl > 63 ? 63 : kl))];
#endif
          b8[t] = (kl >= 0 && kl < 64) ? v : 0.f;
        }
        const bf16x8 bfr = pack8(b8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#if S2T_ATT_DBG & 16
          dqv[dt] = mfma16(bfr, bfr, dqv[dt]);
#else
          dqv[dt] = mfma16(as_frag(ldg16(pt + (int64_t)(16 * dt) * a.pt_ld + 32 * ks)), bfr, dqv[dt]);
#endif
      }
    }
    ASTAMP();
    // dQ^T[c][q] += K^T[c][key] dS^T[key][q]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float d8[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d8[r] = ds[2 * s][r];
        d8[4 + r] = ds[2 * s + 1][r];
      }
      const bf16x8 df = pack8(d8);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(frag_cols_perm(lk, dt, s, x, y), df, dq[dt]);
    }
    if (REL && a.dbd && !(S2T_ATT_DBG & 2)) {
      // dBD row of query i, entries Tq-1-i + (k0 .. k0+63), from the scratch tile: ONE store instruction writes 64
      // consecutive entries of one row.  All 16 row reads go out first (the dS registers are dead by now), then the stores:
      // one LDS round trip per block instead of one per row.
      const int jj = k0 + lane;
      float rowv[16];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) rowv[rr] = scratch[rr * DSS + lane];
      // buffer stores into the (h, b) slab: a lane without an element gets an out-of-range offset, which the hardware
      // drops — no exec-mask branch per row (16 per block and wave before)
      const __amdgpu_buffer_rsrc_t dsrd = __builtin_amdgcn_make_buffer_rsrc(
          a.dbd + ((int64_t)h * a.B + b) * a.Tq * a.ldb, 0, (int)((int64_t)a.Tq * a.ldb * 2), 0x00020000);
      const bool jok = jj < a.nk;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int ii = q0w + rr;
        const uint32_t off = (jok && ii < a.nq) ? (uint32_t)(((int64_t)ii * a.ldb + (a.Tq - 1 - ii + jj)) * 2) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(rowv[rr]), dsrd, off, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  ASTAMP();
#if S2T_ATT_DBG & 64
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 200) && blockIdx.y == 1) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.delta) + (int64_t)a.B * a.H * a.Tq) +
                              ((blockIdx.x ? 4 : 0) + w) * 40;
    for (int t = 0; t < 40; ++t) dbg[t] = t < nstamp ? stamp[t] : 0ull;
  }
#endif
  if (i < a.nq) {
    bf16_t* op = a.dq + (int64_t)b * a.q_sb + (int64_t)i * a.q_sr + h * DK;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float v4[4] = {dq[dt][0] + dqv[dt][0], dq[dt][1] + dqv[dt][1], dq[dt][2] + dqv[dt][2], dq[dt][3] + dqv[dt][3]};
      st4_from_f32<bf16_t>(op + 16 * dt + 4 * y, v4);
    }
  }
  if constexpr (fuse_v) {
    // pos_bias_u / pos_bias_v gradients: column sums of the two branches over the workgroup's 64 queries (16 lanes of a
    // wave by shuffles, the four waves through LDS), one atomic per column, branch and workgroup
    __syncthreads();  // the K / V tiles are no longer read
    float* red = reinterpret_cast<float*>(lds);  // [4 waves][2][64]
    const bool live = i < a.nq;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float su = live ? dq[dt][r] : 0.f, sv = live ? dqv[dt][r] : 0.f;
        su = s2t_sum16_up(su);
        sv = s2t_sum16_up(sv);
        if (x == 0) {
          red[(w * 2 + 0) * 64 + 16 * dt + 4 * y + r] = su;
          red[(w * 2 + 1) * 64 + 16 * dt + 4 * y + r] = sv;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const int br = tid >> 6, c = tid & 63;
      const float sum = red[(0 * 2 + br) * 64 + c] + red[(1 * 2 + br) * 64 + c] + red[(2 * 2 + br) * 64 + c] + red[(3 * 2 + br) * 64 + c];
      atomicAdd((br ? a.dpos_v : a.dpos_u) + h * DK + c, sum);
    }
  }
}

// ---- dK, dV: workgroup = 64 keys of one (b,h), wave = 16 keys; walks the query blocks ------------------------------
constexpr int SC2 = 36;  // scratch row stride (floats) of the [16 q][32 n] band in the dK/dV kernel

template <bool REL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const FusedArgs a_in) {
  FusedArgs a = a_in;
  bind_utt(a, blockIdx.x / a_in.H);
  if ((int)blockIdx.y * 64 >= a.nk) return;  // packed batch: no row of this utterance in the key block
  __shared__ __attribute__((aligned(16))) char lds[3 * 8192 + 4 * 16 * SC2 * 4 + 512 + (REL ? 16384 : 0)];
  char* lp = lds + 3 * 8192 + 4 * 16 * SC2 * 4 + 512;  // (REL) the block's 128 position rows (ptile_*)
  char* lqa = lds;            // q (abs) / q+u (rel)
  char* ldo = lds + 8192;     // dO
  char* lqv = lds + 16384;    // q+v (rel)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int x = lane & 15, y = lane >> 4;
  float* scratch = reinterpret_cast<float*>(lds + 3 * 8192) + w * 16 * SC2;
  float* lse_s = reinterpret_cast<float*>(lds + 3 * 8192 + 4 * 16 * SC2 * 4);
  float* del_s = lse_s + 64;
  const int z = blockIdx.x;  // (utterance, head) fastest: the row blocks of one (b, h) share an XCD's L2 (K/V re-reads)
  const int b = z / a.H, h = z % a.H;
  const int k0 = blockIdx.y * 64;
  const int k0w = k0 + 16 * w;
  const int j = k0w + x;                       // this lane's key
  const int jc = j < a.nk ? j : a.nk - 1;
  const int klen = a.key_lens ? min(a.key_lens[b], a.nk) : a.nk;
  const bool key_ok = j < klen;

  // K and V fragments of this lane's key (B operands: k = 32ks + 8y + jj)
  bf16x8 kf[2], vf[2];
  {
    const bf16_t* kp = a.k + (int64_t)b * a.k_sb + (int64_t)jc * a.k_sr + h * DK;
    const bf16_t* vp = a.v + (int64_t)b * a.v_sb + (int64_t)jc * a.v_sr + h * DK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = as_frag(ldg16(kp + (ks * 4 + y) * 8));
      vf[ks] = as_frag(ldg16(vp + (ks * 4 + y) * 8));
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    dk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bf16_t* qb = a.q + (int64_t)b * a.q_sb + h * DK;
  const bf16_t* dob = a.dO + (int64_t)b * a.o_sb + h * DK;
  const uint64_t dkey = a.drop_p > 0.f ? s2t_drop_key(a.drop_seed, a.drop_site) : 0ull;
  const uint32_t dth = s2t_drop_thresh(a.drop_p);
  const float dinv = s2t_drop_scale(a.drop_p);

  const bool fast_mask = (a.Tk & 1) == 0 && (uint64_t)a.B * a.H * (uint64_t)a.Tq * (uint64_t)(a.Tk >> 1) < (1ull << 32);
  const int qstart = a.causal ? (k0 / 64) * 64 : 0;  // queries before the key block see none of its keys
  TileRegs tq, tdo;
  float nlse = 0.f, ndel = 0.f;
  auto prefetch = [&](int q0) __attribute__((always_inline)) {
    tile_load(tq, qb, a.q_sr, q0, a.nq, tid);
    tile_load(tdo, dob, a.o_sr, q0, a.nq, tid);
    if (tid < 64) {
      const int qi = min(q0 + tid, a.nq - 1);
      nlse = a.lse[(int64_t)z * a.Tq + qi];
      ndel = a.delta[(int64_t)z * a.Tq + qi];
    }
  };
  if (qstart < a.nq) prefetch(qstart);
  // Position rows of a whole 64-query block against this wave's 16 keys: n = nb64 + (0..79), nb64 = Tq-1-(q0+63)+k0w —
  // the band of query tile qt, position tile nt is tile 3 - qt + nt of these five.  The four waves' bands (16 rows apart)
  // are one 128-row workgroup tile starting at Tq-1-(q0+63)+k0, staged through LDS like the Q / dO tiles one block ahead
  // (ptile_*): wave w's tile t is image tile w + t.
  PTile tp;
  if constexpr (REL) {
    if (qstart < a.nq) ptile_load(a, tp, h, qstart, k0, tid);
  }
  for (int q0 = qstart; q0 < a.nq; q0 += 64) {
    __syncthreads();
    tile_store(lqa, tq, q0, a.nq, REL ? a.pos_u + h * DK : nullptr, tid);
    tile_store(ldo, tdo, q0, a.nq, nullptr, tid);
    if (REL) tile_store(lqv, tq, q0, a.nq, a.pos_v + h * DK, tid);
    if constexpr (REL) ptile_store(lp, tp, tid);
    if (tid < 64) {
      lse_s[tid] = nlse;
      del_s[tid] = ndel;
    }
    __syncthreads();
    if (q0 + 64 < a.nq) {  // next query block (and its position rows) in flight during this block's MFMAs
      prefetch(q0 + 64);
      if constexpr (REL) ptile_load(a, tp, h, q0 + 64, k0, tid);
    }
    float pd[4][4], ds[4][4];  // [q tile][r]: q = q0 + 16qt + 4y + r, key = this lane's
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      // S[q][key] = qa[q] . K[key]
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) s4 = mfma16(frag_rows(lqa, qt, ks, x, y), kf[ks], s4);
      if constexpr (REL) {
        // band for (16 q x 16 keys): n = nb + (0..30), nb = Tq-1-(q0+16qt+15)+k0w ; element (ql, key x): 15 - ql + x
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            acc = mfma16(frag_rows(lqv, qt, ks, x, y), frag_rows(lp, w + 3 - qt + nt, ks, x, y), acc);
          // lane (x = n index, y) holds q_local = 4y + r
#pragma unroll
          for (int r = 0; r < 4; ++r) scratch[(4 * y + r) * SC2 + 16 * nt + x] = acc[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) s4[r] += scratch[(4 * y + r) * SC2 + 15 - (4 * y + r) + x];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      // dP[q][key] = dO[q] . V[key]
      f32x4 dp4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dp4 = mfma16(frag_rows(ldo, qt, ks, x, y), vf[ks], dp4);
      // the row statistics of this lane's four queries: one 16-byte LDS read each, BEFORE the per-element code (left to the
      // compiler they end up as a 4-byte read + wait inside an exec-mask branch per element)
      const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * qt + 4 * y);
      const f32x4 del4 = *reinterpret_cast<const f32x4*>(del_s + 16 * qt + 4 * y);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // dropout bits of this lane's four (query, key) elements.  One 32-bit hash serves the two elements of an even/odd key
      // pair of a query row; this lane holds ONE key and four queries, so lanes x and x ^ 1 (the two keys of a pair) each
      // hash one query of a pair of queries and swap: two hashes per lane and query tile instead of four, in 32-bit index
      // arithmetic (the general form, one 64-bit-indexed hash per element, serves odd Tk and index spaces beyond 2^32)
      uint32_t rbits[4] = {65535u, 65535u, 65535u, 65535u};
      if (a.drop_p > 0.f) {
        if (fast_mask) {
#pragma unroll
          for (int rb = 0; rb < 4; rb += 2) {
            const int im = min(q0 + 16 * qt + 4 * y + rb + (x & 1), a.nq - 1);
            // ((z Tq + im) (Tk / 2) in 32-bit wrap-around arithmetic, the per-lane part as a full-rate 24-bit multiply: im, Tk < 2^16)
            const uint32_t pair = (uint32_t)z * (uint32_t)a.Tq * (uint32_t)(a.Tk >> 1) + __umul24((uint32_t)im, (uint32_t)(a.Tk >> 1)) + (uint32_t)(jc >> 1);
            const uint32_t hm = s2t_mix32(pair ^ (uint32_t)dkey) ^ (uint32_t)(dkey >> 32);
            const uint32_t ho = (uint32_t)__builtin_amdgcn_mov_dpp((int)hm, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]: lane x ^ 1
            const uint32_t h0 = (x & 1) ? ho : hm, h1 = (x & 1) ? hm : ho;  // hashes of queries rb, rb + 1
            rbits[rb] = (jc & 1) ? (h0 >> 16) : (h0 & 0xffffu);
            rbits[rb + 1] = (jc & 1) ? (h1 >> 16) : (h1 & 0xffffu);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = q0 + 16 * qt + 4 * y + r;
            rbits[r] = s2t_rand_u32(dkey, ((uint64_t)z * a.Tq + (uint64_t)min(i, a.nq - 1)) * (uint64_t)a.Tk + jc);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * qt + 4 * y + r;
        const int i = q0 + ql;
        const bool ok = key_ok & (i < a.nq) & !((a.causal != 0) & (j > i));
        const float p = __expf(ok ? s4[r] * a.scale - lse4[r] : -INFINITY);  // select on the argument, no branch
        float dp = dp4[r];
        float pdrop = p;
        if (a.drop_p > 0.f) {
          const bool keep = rbits[r] >= dth;
          dp = keep ? dp * dinv : 0.f;
          pdrop = keep ? p * dinv : 0.f;
        }
        pd[qt][r] = pdrop;
        ds[qt][r] = p * (dp - del4[r]) * a.scale;
      }
    }
    // dV^T[c][key] += dO^T[c][q] Pd[q][key] ; dK^T[c][key] += qa^T[c][q] dS[q][key]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float p8[8], d8[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p8[r] = pd[2 * s][r];
        p8[4 + r] = pd[2 * s + 1][r];
        d8[r] = ds[2 * s][r];
        d8[4 + r] = ds[2 * s + 1][r];
      }
      const bf16x8 pf = pack8(p8), df = pack8(d8);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        dv[ct] = mfma16(frag_cols_perm(ldo, ct, s, x, y), pf, dv[ct]);
        dk[ct] = mfma16(frag_cols_perm(lqa, ct, s, x, y), df, dk[ct]);
      }
    }
  }
  if (j < a.nk) {
    bf16_t* kp = a.dk + (int64_t)b * a.k_sb + (int64_t)j * a.k_sr + h * DK;
    bf16_t* vp = a.dv + (int64_t)b * a.v_sb + (int64_t)j * a.v_sr + h * DK;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float k4[4] = {dk[ct][0], dk[ct][1], dk[ct][2], dk[ct][3]};
      float v4[4] = {dv[ct][0], dv[ct][1], dv[ct][2], dv[ct][3]};
      st4_from_f32<bf16_t>(kp + 16 * ct + 4 * y, k4);
      st4_from_f32<bf16_t>(vp + 16 * ct + 4 * y, v4);
    }
  }
}

// =====================================================================================================================
// d(Q + pos_bias_v) of the relative form after the fused backward (espnet_multihead_attention.py:331-337 backward):
//   dqv[i][c] = sum_n dbd[i][n] * p[n][c]   over the band n in [Tq-1-i, 2Tq-2-i] that dbd holds for query i,
//   dq += dqv (in place),  dpos_u[h*64+c] += sum_i dq_old[i][c],  dpos_v[h*64+c] += sum_i dqv[i][c].
// One launch in place of a batched GEMM (K = 2Tq-1 with half of every row zero, N = 64 in 128-wide tiles) and the add +
// two column sums behind it.  Both operands come straight from global memory as 16-byte fragments — dbd rows (B operand,
// n = query) and the TRANSPOSED projected positions (A operand, m = channel; project_positions keeps them, zero padded).
// The kernel waits on memory, not on arithmetic: a workgroup is ONE wave that owns 64 queries (four 16-query tiles) of one
// (utterance, head) and walks the union of their bands in K-steps of 32 with four steps in flight; the four position
// fragments of a step serve all four query tiles.  The swapped product leaves lane (x = query, g) with channels
// 16 nt + 4 g .. +4 of its query: 8-byte read-modify-writes of dq.
__global__ __launch_bounds__(64) void relpos_dqv_kernel(const bf16_t* __restrict__ dbd, int64_t ldb,
                                                        const bf16_t* __restrict__ pos_pt, int64_t pt_ld,
                                                        bf16_t* __restrict__ dq, int64_t dq_sb, int64_t dq_sr,
                                                        float* __restrict__ du, float* __restrict__ dv, int replicas,
                                                        int64_t replica_stride, int B, int H, int Tq) {
  const int lane = threadIdx.x;
  const int x = lane & 15, g = lane >> 4;
  const int z = blockIdx.x;
  const int b = z / H, h = z % H;
  const int i0 = blockIdx.y * 64;
  f32x4 acc[4][4];  // [query tile][channel tile]
#pragma unroll
  for (int qt = 0; qt < 4; ++qt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[qt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int n_lo = max(Tq - 1 - (i0 + 63), 0) & ~7;
  const int n_hi = 2 * Tq - 2 - i0;  // last band column of the first query
  const int steps = (n_hi + 1 - n_lo + 31) / 32;
  const bf16_t* abase = dbd + (((int64_t)h * B + b) * Tq) * ldb + 8 * g;
  const bf16_t* prow = pos_pt + ((int64_t)h * DK + x) * pt_ld + 8 * g;
  // the dq values this lane will update travel during the products
  uint2 old[4][4];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int i = min(i0 + 16 * qt + x, Tq - 1);
    const bf16_t* qrow = dq + (int64_t)b * dq_sb + (int64_t)i * dq_sr + h * DK + 4 * g;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) old[qt][nt] = *reinterpret_cast<const uint2*>(qrow + 16 * nt);
  }
  struct Step {
    uint4 a[4], p[4];
  };
  auto load = [&](int st, Step& t) __attribute__((always_inline)) {
    const int n = n_lo + 32 * st;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) t.p[nt] = ldg16(prow + (int64_t)(16 * nt) * pt_ld + n);
    // (columns outside the band of tile qt, still inside the union, hold zeros in dbd; rows / columns past the end: the load
    // goes to a clamped address and the value is replaced — a load under a per-lane condition becomes an exec-mask branch)
    const int nc = min(n, (int)ldb - 8 - 8 * g);
    const bool col_ok = n + 8 * g < ldb;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int i = i0 + 16 * qt + x;
      const uint4 v = ldg16(abase + (int64_t)min(i, Tq - 1) * ldb + nc);
      t.a[qt] = (col_ok && i < Tq) ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto mma = [&](const Step& t) __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[qt][nt] = mfma16(as_frag(t.p[nt]), as_frag(t.a[qt]), acc[qt][nt]);
  };
  {
    Step s0, s1, s2, s3;
    load(0, s0);
    if (1 < steps) load(1, s1);
    if (2 < steps) load(2, s2);
    for (int st = 0; st < steps; st += 4) {
      if (st + 3 < steps) load(st + 3, s3);
      mma(s0);
      if (st + 4 < steps) load(st + 4, s0);
      if (st + 1 < steps) mma(s1);
      if (st + 5 < steps) load(st + 5, s1);
      if (st + 2 < steps) mma(s2);
      if (st + 6 < steps) load(st + 6, s2);
      if (st + 3 < steps) mma(s3);
    }
  }
  float su[4][4], sv[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) su[nt][r] = sv[nt][r] = 0.f;
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int i = i0 + 16 * qt + x;
    if (i < Tq) {
      bf16_t* qrow = dq + (int64_t)b * dq_sb + (int64_t)i * dq_sr + h * DK + 4 * g;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const uint2 o = old[qt][nt];
        const float o4[4] = {__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u),
                             __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
        float n4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          su[nt][r] += o4[r];
          sv[nt][r] += acc[qt][nt][r];
          n4[r] = o4[r] + acc[qt][nt][r];
        }
        st4_from_f32<bf16_t>(qrow + 16 * nt, n4);
      }
    }
  }
  // column sums over the wave's queries by lane exchanges, then one atomic per channel
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      su[nt][r] = s2t_sum16_up(su[nt][r]);
      sv[nt][r] = s2t_sum16_up(sv[nt][r]);
    }
  // (a thousand workgroups adding to the same 2 x 64 floats of a head serialise in the L2 atomic units — measured: half of
  // the kernel's time — so the sums are spread over `replicas` copies that a later fold adds up: the LayerNorm
  // parameter-gradient workspace and its fold kernel serve)
  // (two atomic instructions with one channel per lane instead of 32 with four live lanes each: the sums change hands in LDS)
  __shared__ float red[2][DK];
  if (x == 0) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[0][16 * nt + 4 * g + r] = su[nt][r];
        red[1][16 * nt + 4 * g + r] = sv[nt][r];
      }
  }
  __syncthreads();
  const int64_t ro = (int64_t)((blockIdx.x / H + blockIdx.y * 7) % replicas) * replica_stride;
  atomicAdd(du + ro + h * DK + lane, red[0][lane]);
  atomicAdd(dv + ro + h * DK + lane, red[1][lane]);
}

}  // namespace

extern "C" int s2t_attn_fused_fwd(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr,
                                  const void* v, int64_t v_sb, int64_t v_sr, void* o, int64_t o_sb, int64_t o_sr,
                                  float* lse, int B, int H, int Tq, int Tk, int dk, const int32_t* key_lens, int causal,
                                  float scale, const void* pos_p, int64_t p_sr, const float* pos_u, const float* pos_v,
                                  float drop_p, const uint64_t* drop_seed, uint32_t drop_site, const int32_t* cu_q,
                                  const int32_t* cu_k, void* o_lo, void* stream) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return S2T_ERR_ARG;
  if (o_lo && ((uintptr_t)o_lo % 8)) return S2T_ERR_ALIGN;
  if (dk != DK) return S2T_ERR_UNSUPPORTED;
  if (q_sr >= 65536 || k_sr >= 65536 || v_sr >= 65536 || o_sr >= 65536 || p_sr >= 65536 || Tq >= 65536 || Tk >= 65536 ||
      q_sr < 0 || k_sr < 0 || v_sr < 0 || o_sr < 0 || p_sr < 0)
    return S2T_ERR_UNSUPPORTED;  // (24-bit row x stride products in the tile loads)
  if (pos_p && (uint64_t)(2 * Tq - 1) * (uint64_t)p_sr >= (1ull << 32))
    return S2T_ERR_UNSUPPORTED;  // (position rows run to 2 Tq - 2: their 24-bit product must stay inside 32 bits)
  if (drop_p < 0.f || drop_p >= 1.f) return S2T_ERR_ARG;
  if (pos_p && (!pos_u || !pos_v || Tq != Tk)) return S2T_ERR_ARG;
  if ((q_sr % 8) || (k_sr % 8) || (v_sr % 8) || (o_sr % 4) || ((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)v % 16))
    return S2T_ERR_ALIGN;
  FusedArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr;
  a.o = (bf16_t*)o; a.o_sb = o_sb; a.o_sr = o_sr; a.lse = lse; a.o_lo = (bf16_t*)o_lo;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.key_lens = key_lens; a.causal = causal; a.scale = scale;
  a.rel = pos_p != nullptr; a.pos_p = (const bf16_t*)pos_p; a.p_sr = p_sr; a.pos_u = pos_u; a.pos_v = pos_v;
  a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_site = drop_site;
  a.cu_q = cu_q; a.cu_k = cu_k; a.nq = Tq; a.nk = Tk;
  dim3 grid(B * H, (Tq + 63) / 64), block(256);
  // S2T_ATTN_BH=0 keeps the block-per-64-queries kernels for the short relative-position case too (A/B switch)
  static const bool bh_on = [] { const char* e = getenv("S2T_ATTN_BH"); return !(e && e[0] == '0'); }();
  if (a.rel && bh_on && Tq == Tk && Tk <= BH_MAXT && !causal)
    hipLaunchKernelGGL(attn_bh_fwd_kernel, dim3(B * H), dim3(512), 0, (hipStream_t)stream, a);
  else if (a.rel) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_relpos_dqv(const void* dbd, int64_t ldb, const void* pos_pt, int64_t pt_ld, void* dq, int64_t dq_sb,
                              int64_t dq_sr, float* dpos_u, float* dpos_v, int replicas, int64_t replica_stride, int B,
                              int H, int Tq, int dk, void* stream) {
  if (!dbd || !pos_pt || !dq || !dpos_u || !dpos_v || B <= 0 || H <= 0 || Tq <= 0 || replicas < 1) return S2T_ERR_ARG;
  if (dk != DK) return S2T_ERR_UNSUPPORTED;
  if (ldb < 2 * Tq - 1 || ldb % 8 || pt_ld % 8 || dq_sr % 4 || dq_sb % 4) return S2T_ERR_ARG;
  if (((uintptr_t)dbd % 16) || ((uintptr_t)pos_pt % 16) || ((uintptr_t)dq % 8)) return S2T_ERR_ALIGN;
  hipLaunchKernelGGL(relpos_dqv_kernel, dim3(B * H, (Tq + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)dbd,
                     ldb, (const bf16_t*)pos_pt, pt_ld, (bf16_t*)dq, dq_sb, dq_sr, dpos_u, dpos_v, replicas, replica_stride, B, H, Tq);
  return S2T_LAUNCH_CHECK();
}

extern "C" int s2t_attn_fused_bwd(const void* q, int64_t q_sb, int64_t q_sr, const void* k, int64_t k_sb, int64_t k_sr,
                                  const void* v, int64_t v_sb, int64_t v_sr, const void* o, const void* dO, int64_t o_sb,
                                  int64_t o_sr, const float* lse, float* delta, void* dq, void* dk_, void* dv, void* dbd,
                                  int64_t ldb, int B, int H, int Tq, int Tk, int dk, const int32_t* key_lens, int causal,
                                  float scale, const void* pos_p, int64_t p_sr, const float* pos_u, const float* pos_v,
                                  float drop_p, const uint64_t* drop_seed, uint32_t drop_site, int dbd_band_only,
                                  const void* pos_pt, int64_t pt_ld, float* dpos_u, float* dpos_v, void* qv_out,
                                  const int32_t* cu_q, const int32_t* cu_k, const void* o_lo, void* stream) {
  if (!q || !k || !v || !o || !dO || !lse || !delta || !dq || !dk_ || !dv || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0)
    return S2T_ERR_ARG;
  if (o_lo && ((uintptr_t)o_lo % 16)) return S2T_ERR_ALIGN;
  if (dk != DK) return S2T_ERR_UNSUPPORTED;
  if (q_sr >= 65536 || k_sr >= 65536 || v_sr >= 65536 || o_sr >= 65536 || p_sr >= 65536 || Tq >= 65536 || Tk >= 65536 ||
      q_sr < 0 || k_sr < 0 || v_sr < 0 || o_sr < 0 || p_sr < 0)
    return S2T_ERR_UNSUPPORTED;  // (24-bit row x stride products in the tile loads)
  if (pos_p && (uint64_t)(2 * Tq - 1) * (uint64_t)p_sr >= (1ull << 32))
    return S2T_ERR_UNSUPPORTED;  // (position rows run to 2 Tq - 2: their 24-bit product must stay inside 32 bits)
  if (drop_p < 0.f || drop_p >= 1.f) return S2T_ERR_ARG;
  if (pos_p && (!pos_u || !pos_v || Tq != Tk)) return S2T_ERR_ARG;
  if (dbd && ldb < 2 * Tq - 1) return S2T_ERR_ARG;
  if (pos_pt && (!pos_p || !dpos_u || !dpos_v || pt_ld % 8 || ((uintptr_t)pos_pt % 16))) return S2T_ERR_ARG;
  FusedArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr;
  a.o = (bf16_t*)const_cast<void*>(o); a.o_sb = o_sb; a.o_sr = o_sr; a.lse = const_cast<float*>(lse);
  a.o_lo = (bf16_t*)const_cast<void*>(o_lo);
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.key_lens = key_lens; a.causal = causal; a.scale = scale;
  a.rel = pos_p != nullptr; a.pos_p = (const bf16_t*)pos_p; a.p_sr = p_sr; a.pos_u = pos_u; a.pos_v = pos_v;
  a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_site = drop_site;
  a.dO = (const bf16_t*)dO; a.delta = delta; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk_; a.dv = (bf16_t*)dv;
  a.dbd = (bf16_t*)dbd; a.ldb = ldb; a.dbd_band_only = dbd_band_only;
  a.pos_pt = (const bf16_t*)pos_pt; a.pt_ld = pt_ld; a.dpos_u = dpos_u; a.dpos_v = dpos_v;
  a.qv_out = (bf16_t*)qv_out;
  a.cu_q = cu_q; a.cu_k = cu_k; a.nq = Tq; a.nk = Tk;
  if (qv_out && (!pos_p || ((uintptr_t)qv_out % 16))) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  // (delta = rowsum(dO * O) is produced by the dQ kernel, which runs first, and read by the dK / dV kernel)
  dim3 gq(B * H, (Tq + 63) / 64), gk(B * H, (Tk + 63) / 64), block(256);
  if (a.rel) {
    if (a.pos_pt) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, true>), gq, block, 0, s, a);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false>), gq, block, 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, gk, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false>), gq, block, 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, gk, block, 0, s, a);
  }
  return S2T_LAUNCH_CHECK();
}
