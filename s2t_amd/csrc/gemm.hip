// MFMA GEMM with fused epilogue for gfx950 (see include/s2t_hip.h: s2t_gemm).
//
// Tile 128(M) x 128(N) per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles),
// K-step = 128 bytes of K per row (64 bf16 / 32 f32), operands register-staged global -> LDS with a
// double-buffered LDS image (one barrier per K-step; the next tile's global loads are issued before the
// MFMAs of the current one and written to LDS after them).
//
//   bf16: v_mfma_f32_16x16x32_bf16, fp32 accumulate.      f32: v_mfma_f32_16x16x4_f32 (exact f32 chain).
//
// The MFMA is issued "swapped" (first operand = B-tile rows, second = A-tile rows) so that a lane ends up
// holding 4 CONSECUTIVE output columns of one output row (D row index = n, D column = m): the epilogue
// then works on 8/16-byte vectors (bias, residual, GLU pairs, stores) instead of 2/4-byte scalars.
//
// LDS images (16 KiB per operand per buffer):
//   row-major operand  ([row][k], 128 B per row, 16-B chunks c=0..7):  chunk c of row r at r*128 + ((c^(r&7))<<4)
//                      -> ds_read_b128 of 16 rows x same chunk is conflict-free.
//   k-major operand    bf16: [k][128 cols] 256 B per k-row, chunk c=0..15 at k*256 + ((c ^ swz(k))<<4),
//                      swz(k) = 2*((k&3) | ((k>>3)&1)<<2): the 8 k-rows one half-wave touches in a
//                      ds_read_b64_tr_b16 land on 8 distinct 32-B column pairs = all 64 banks once.
//                      f32: [k][128 cols] 512 B per k-row, plain; read with ds_read_b32.
#include "common.h"

#define BM 128
#define BN 128

namespace {

template <typename T>
struct TileTraits;
template <>
struct TileTraits<float> {
  static constexpr int EPB = 4;    // elements per 16-byte chunk
  static constexpr int BKE = 32;   // K elements per tile
};
template <>
struct TileTraits<bf16_t> {
  static constexpr int EPB = 8;
  static constexpr int BKE = 64;
};

template <typename T>
__device__ __forceinline__ void mask_tail(uint4& v, int nvalid) {
  // keep the first nvalid elements of the 16-byte chunk, zero the rest
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i >= nvalid) w[i] = 0;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (2 * i >= nvalid) w[i] = 0;
      else if (2 * i + 1 >= nvalid) w[i] &= 0xffffu;
    }
  }
  v = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ int kswz(int k) { return 2 * ((k & 3) | (((k >> 3) & 1) << 2)); }

// ---- global -> registers ---------------------------------------------------------------------
// row-major operand: element (row, k) at base[row*ld + k]
template <typename T, bool GLU_B>
__device__ __forceinline__ void load_rowmajor(uint4 (&reg)[4], const T* __restrict__ base, int64_t ld, int row0,
                                              int nrows, int k0, int K, bool ktail, int tid, int glu_half_rows) {
  constexpr int EPB = TileTraits<T>::EPB;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cid = tid + 256 * u;
    const int r = cid >> 3, ch = cid & 7;
    int grow;
    bool rv;
    if constexpr (GLU_B) {
      // 16-row blocks alternate value / gate rows of the weight: block s -> half = s&1
      const int s = r >> 4;
      const int nloc = (s >> 1) * 16 + (r & 15);
      const int o = row0 + nloc;  // output column; row0 = tn*64
      rv = o < glu_half_rows;
      grow = (s & 1) * glu_half_rows + o;
    } else {
      grow = row0 + r;
      rv = grow < nrows;
    }
    const int k = k0 + ch * EPB;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rv && k < K) {
      v = *reinterpret_cast<const uint4*>(base + (int64_t)grow * ld + k);
      if (ktail && k + EPB > K) mask_tail<T>(v, K - k);
    }
    reg[u] = v;
  }
}

// k-major operand: element (k, col) at base[k*ld + col]
template <typename T>
__device__ __forceinline__ void load_kmajor(uint4 (&reg)[4], const T* __restrict__ base, int64_t ld, int col0,
                                            int ncols, int k0, int K, bool ctail, int tid) {
  constexpr int EPB = TileTraits<T>::EPB;
  constexpr int CPR = 128 / EPB;  // chunks per k-row: 16 (bf16) / 32 (f32)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cid = tid + 256 * u;
    const int kr = cid / CPR, ch = cid % CPR;
    const int gk = k0 + kr;
    const int gc = col0 + ch * EPB;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gk < K && gc < ncols) {
      v = *reinterpret_cast<const uint4*>(base + (int64_t)gk * ld + gc);
      if (ctail && gc + EPB > ncols) mask_tail<T>(v, ncols - gc);
    }
    reg[u] = v;
  }
}

// ---- registers -> LDS ------------------------------------------------------------------------
__device__ __forceinline__ void store_rowmajor(char* lds, const uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cid = tid + 256 * u;
    const int r = cid >> 3, ch = cid & 7;
    *reinterpret_cast<uint4*>(lds + r * 128 + ((ch ^ (r & 7)) << 4)) = reg[u];
  }
}
template <typename T>
__device__ __forceinline__ void store_kmajor(char* lds, const uint4 (&reg)[4], int tid) {
  constexpr int EPB = TileTraits<T>::EPB;
  constexpr int CPR = 128 / EPB;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cid = tid + 256 * u;
    const int kr = cid / CPR, ch = cid % CPR;
    if constexpr (sizeof(T) == 2)
      *reinterpret_cast<uint4*>(lds + kr * 256 + ((ch ^ kswz(kr)) << 4)) = reg[u];
    else
      *reinterpret_cast<uint4*>(lds + kr * 512 + (ch << 4)) = reg[u];
  }
}

// ---- LDS -> MFMA fragments -------------------------------------------------------------------
// Fragment of a 16-row (row-major operand) / 16-column (k-major operand) block starting at blk0 of the tile,
// for K sub-step ks (bf16: 32 k per sub-step, f32: 16 k per sub-step = 4 MFMAs of k=4).
struct Frag {
  uint4 v;  // bf16: 8 elements k = ks*32 + 8y + j; f32: 4 elements k = ks*16 + 4y + jj
};

template <typename T, bool KM>
__device__ __forceinline__ Frag read_frag(const char* lds, int blk0, int ks, int x, int y) {
  Frag f;
  if constexpr (!KM) {
    const int r = blk0 + x;
    const int c = ks * 4 + y;
    f.v = *reinterpret_cast<const uint4*>(lds + r * 128 + ((c ^ (r & 7)) << 4));
  } else if constexpr (sizeof(T) == 2) {
    const int q = x >> 2, p = x & 3;
    const int col = blk0 + 4 * p;
    const int chunk = col >> 3;
    const int within = (p & 1) * 8;
    uint32_t w[4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int R = ks * 32 + 8 * y + 4 * half + q;
      const char* a = lds + R * 256 + ((chunk ^ kswz(R)) << 4) + within;
      s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(a));
      uint2 tt = __builtin_bit_cast(uint2, t);
      w[2 * half] = tt.x;
      w[2 * half + 1] = tt.y;
    }
    f.v = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int R = ks * 16 + 4 * y + jj;
      w[jj] = *reinterpret_cast<const uint32_t*>(lds + R * 512 + (blk0 + x) * 4);
    }
    f.v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  return f;
}

template <typename T>
__device__ __forceinline__ void mma(f32x4& acc, const Frag& first, const Frag& second) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, first.v),
                                                   __builtin_bit_cast(bf16x8, second.v), acc, 0, 0, 0);
  } else {
    const uint32_t a[4] = {first.v.x, first.v.y, first.v.z, first.v.w};
    const uint32_t b[4] = {second.v.x, second.v.y, second.v.z, second.v.w};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[jj]), __uint_as_float(b[jj]), acc, 0, 0, 0);
  }
}

// ---- epilogue: 8 consecutive output columns of one row per thread (after the LDS transpose) ------
template <typename X>
__device__ __forceinline__ void ld8(const X* ptr, bool vec, int nv, float (&o)[8]) {
  if (vec && nv == 8) {
    if constexpr (sizeof(X) == 2) {
      const uint4 t = *reinterpret_cast<const uint4*>(ptr);
      const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[2 * q] = __uint_as_float(w[q] << 16);
        o[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
      }
    } else {
      const float4 a = *reinterpret_cast<const float4*>(ptr);
      const float4 b = *reinterpret_cast<const float4*>(ptr + 4);
      o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = r < nv ? ld_as_f32<X>(ptr + r) : 0.f;
  }
}
template <typename X>
__device__ __forceinline__ void st8(X* ptr, bool vec, int nv, const float (&o)[8]) {
  if (vec && nv == 8) {
    if constexpr (sizeof(X) == 2) {
      uint4 t;
      t.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
      t.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
      t.z = (uint32_t)f2bf(o[4]) | ((uint32_t)f2bf(o[5]) << 16);
      t.w = (uint32_t)f2bf(o[6]) | ((uint32_t)f2bf(o[7]) << 16);
      *reinterpret_cast<uint4*>(ptr) = t;
    } else {
      *reinterpret_cast<float4*>(ptr) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(ptr + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < nv) st_from_f32<X>(ptr + r, o[r]);
  }
}

template <typename TC>
struct Epi {
  const s2t_gemm_args& p;
  TC* C;
  const TC* R;
  TC* P;
  const TC* Z;
  int nout;        // output columns (N, or N/2 under GLU)
  bool vec_c, vec_r, vec_p, vec_z;

  __device__ __forceinline__ void bias8(int n0, int nv, float (&b)[8]) const {
    if (!p.bias) {
#pragma unroll
      for (int r = 0; r < 8; ++r) b[r] = 0.f;
      return;
    }
    if (p.bias_dtype == S2T_F32) {
      const float* bp = reinterpret_cast<const float*>(p.bias) + n0;
      ld8<float>(bp, ((uintptr_t)bp % 16) == 0, nv, b);
    } else {
      const bf16_t* bp = reinterpret_cast<const bf16_t*>(p.bias) + n0;
      ld8<bf16_t>(bp, ((uintptr_t)bp % 16) == 0, nv, b);
    }
  }
  __device__ __forceinline__ bool row_masked(int64_t grow) const {
    if (!p.row_lens) return false;
    const int b = (int)(grow / p.row_T), t = (int)(grow % p.row_T);
    return t >= p.row_lens[b];
  }
  // v: post-bias (post-GLU) values for output columns n0..n0+7 of row m
  __device__ __forceinline__ void finish(int m, int n0, int64_t grow, float (&v)[8]) const {
    const int nv = min(8, nout - n0);
    if (p.act == S2T_ACT_RELU || p.act == S2T_ACT_SWISH) {
      if (P) st8<TC>(P + (int64_t)m * p.ldp + n0, vec_p, nv, v);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = act_apply(p.act, v[r]);
    }
    if (Z) {
      float z[8];
      ld8<TC>(Z + (int64_t)m * p.ldz + n0, vec_z, nv, z);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] *= act_grad(p.dact, z[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= p.alpha;
    if (row_masked(grow)) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = 0.f;
    }
    if (R) {
      float q[8];
      ld8<TC>(R + (int64_t)m * p.ldr + n0, vec_r, nv, q);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += q[r];
    }
    st8<TC>(C + (int64_t)m * p.ldc + n0, vec_c, nv, v);
  }
};

// fp32 C tile in LDS: [128 rows][128 cols], 16-byte chunk c of row r at r*512 + ((c ^ (r&7))<<4)
__device__ __forceinline__ float4 ctile_ld4(const char* smem, int row, int chunk) {
  return *reinterpret_cast<const float4*>(smem + row * 512 + ((chunk ^ (row & 7)) << 4));
}
__device__ __forceinline__ void ctile_ld8(const char* smem, int row, int col0, float (&v)[8]) {
  const float4 a = ctile_ld4(smem, row, col0 >> 2);
  const float4 b = ctile_ld4(smem, row, (col0 >> 2) + 1);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <typename T, bool AKM, bool BKM, typename TC, bool GLU>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const s2t_gemm_args p) {
  constexpr int BKE = TileTraits<T>::BKE;
  constexpr int EPB = TileTraits<T>::EPB;
  __shared__ __attribute__((aligned(16))) char smem[65536];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int x = lane & 15, y = lane >> 4;

  const int nout = GLU ? p.N / 2 : p.N;
  const int bn_out = GLU ? 64 : 128;  // output columns per tile
  const int tiles_n = (nout + bn_out - 1) / bn_out;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;

  const int z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z % p.zdiv;
  const T* A = reinterpret_cast<const T*>(p.A) + z0 * p.a_s0 + z1 * p.a_s1;
  const T* B = reinterpret_cast<const T*>(p.B) + z0 * p.b_s0 + z1 * p.b_s1;
  const int64_t coff = z0 * p.c_s0 + z1 * p.c_s1;

  const int ktiles = (p.K + BKE - 1) / BKE;
  const int per = (ktiles + p.split_k - 1) / p.split_k;
  const int kt0 = blockIdx.y * per;
  const int kt1 = min(ktiles, kt0 + per);

  const bool a_tail = AKM ? (p.M % EPB) != 0 : (p.K % EPB) != 0;
  const bool b_tail = BKM ? (p.N % EPB) != 0 : (p.K % EPB) != 0;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kt * BKE;
    if constexpr (AKM) load_kmajor<T>(ra, A, p.lda, tm * BM, p.M, k0, p.K, a_tail, tid);
    else load_rowmajor<T, false>(ra, A, p.lda, tm * BM, p.M, k0, p.K, a_tail, tid, 0);
    if constexpr (BKM) load_kmajor<T>(rb, B, p.ldb, tn * BN, p.N, k0, p.K, b_tail, tid);
    else load_rowmajor<T, GLU>(rb, B, p.ldb, tn * bn_out, p.N, k0, p.K, b_tail, tid, nout);
  };
  auto lstore = [&](int buf) {
    char* la = smem + buf * 32768;
    char* lb = la + 16384;
    if constexpr (AKM) store_kmajor<T>(la, ra, tid); else store_rowmajor(la, ra, tid);
    if constexpr (BKM) store_kmajor<T>(lb, rb, tid); else store_rowmajor(lb, rb, tid);
  };

  if (kt0 < kt1) {
    gload(kt0);
    lstore(0);
  }
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) gload(kt + 1);
    const char* la = smem + buf * 32768;
    const char* lb = la + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<T, AKM>(la, wm * 64 + i * 16, ks, x, y);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<T, BKM>(lb, wn * 64 + j * 16, ks, x, y);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma<T>(acc[i][j], fb[j], fa[i]);
    }
    if (kt + 1 < kt1) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // Transpose the accumulators through LDS (the operand buffers are free after the loop's last barrier) so that
  // every global access of the epilogue is a 16-byte vector on a full 128-byte row segment.
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wm * 64 + i * 16 + x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int chunk = wn * 16 + j * 4 + y;
      *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 7)) << 4)) = acc[i][j];
    }
  }
  __syncthreads();

  if (p.split_k > 1 || p.c_atomic) {
    // one float per lane, 64 consecutive columns per wave-instruction = 256 contiguous bytes per atomic instruction
    float* C = reinterpret_cast<float*>(p.C) + coff;
    const int col = tid & 127;
    const int n = tn * BN + col;
    if (n < p.N) {
#pragma unroll 4
      for (int pass = 0; pass < 64; ++pass) {
        const int row = pass * 2 + (tid >> 7);
        const int m = tm * BM + row;
        if (m < p.M) {
          const float v = *reinterpret_cast<const float*>(smem + row * 512 + (((col >> 2) ^ (row & 7)) << 4) + ((col & 3) << 2));
          atomicAdd(C + (int64_t)m * p.ldc + n, p.alpha * v);
        }
      }
    }
    return;
  }

  Epi<TC> e{p,
            reinterpret_cast<TC*>(p.C) + coff,
            p.residual ? reinterpret_cast<const TC*>(p.residual) + coff : nullptr,
            p.preact ? reinterpret_cast<TC*>(p.preact) + (z0 * p.p_s0 + z1 * p.p_s1) : nullptr,
            p.dact_z ? reinterpret_cast<const TC*>(p.dact_z) + coff : nullptr,
            nout,
            false, false, false, false};
  auto vec_ok = [](const void* ptr, int64_t ld) { return ((ld * (int64_t)sizeof(TC)) % 16 == 0) && (((uintptr_t)ptr) % 16 == 0); };
  e.vec_c = vec_ok(e.C, p.ldc);
  e.vec_r = e.R && vec_ok(e.R, p.ldr);
  e.vec_p = e.P && vec_ok(e.P, p.ldp);
  e.vec_z = e.Z && vec_ok(e.Z, p.ldz);

  const int c8 = tid & 7;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 32 + (tid >> 3);
    const int m = tm * BM + row;
    if (m >= p.M) continue;
    const int64_t grow = (int64_t)z * p.M + m;
    if constexpr (GLU) {
      // LDS columns per 32-column group q: [16 value | 16 gate]; output column o = q*16 + (0..15)
      const int o0 = c8 * 8;
      const int n0 = tn * 64 + o0;
      if (n0 >= nout) continue;
      const int lcol = (o0 >> 4) * 32 + (o0 & 15);
      const int nv = min(8, nout - n0);
      float a[8], g[8], ba[8], bg[8], v[8];
      ctile_ld8(smem, row, lcol, a);
      ctile_ld8(smem, row, lcol + 16, g);
      e.bias8(n0, nv, ba);
      e.bias8(nout + n0, nv, bg);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        a[r] += ba[r];
        g[r] += bg[r];
        v[r] = a[r] * sigmoidf_(g[r]);
      }
      if (e.P) {
        st8<TC>(e.P + (int64_t)m * p.ldp + n0, e.vec_p, nv, a);
        st8<TC>(e.P + (int64_t)m * p.ldp + nout + n0, e.vec_p && ((nout * (int)sizeof(TC)) % 16 == 0), nv, g);
      }
      e.finish(m, n0, grow, v);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col0 = h * 64 + c8 * 8;
        const int n0 = tn * BN + col0;
        if (n0 >= nout) continue;
        float v[8], b[8];
        ctile_ld8(smem, row, col0, v);
        e.bias8(n0, min(8, nout - n0), b);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += b[r];
        e.finish(m, n0, grow, v);
      }
    }
  }
}

template <typename T, typename TC>
int launch(const s2t_gemm_args& p, hipStream_t s) {
  const bool glu = p.act == S2T_ACT_GLU;
  const int nout = glu ? p.N / 2 : p.N;
  const int bn_out = glu ? 64 : 128;
  const int tiles = ((p.M + BM - 1) / BM) * ((nout + bn_out - 1) / bn_out);
  dim3 grid(tiles, p.split_k, p.batch), block(256);
#define GO(AK, BK, G) hipLaunchKernelGGL((gemm_kernel<T, AK, BK, TC, G>), grid, block, 0, s, p)
  if (glu) {
    if (p.a_kmajor || p.b_kmajor) return S2T_ERR_UNSUPPORTED;
    GO(false, false, true);
  } else if (!p.a_kmajor && !p.b_kmajor) GO(false, false, false);
  else if (!p.a_kmajor && p.b_kmajor) GO(false, true, false);
  else if (p.a_kmajor && !p.b_kmajor) GO(true, false, false);
  else GO(true, true, false);
#undef GO
  return S2T_LAUNCH_CHECK();
}

}  // namespace

extern "C" int s2t_gemm(const s2t_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) return S2T_ERR_ARG;
  s2t_gemm_args p = *a;
  if (p.M <= 0 || p.N <= 0 || p.K < 0) return S2T_ERR_ARG;
  if (p.batch <= 0) p.batch = 1;
  if (p.zdiv <= 0) p.zdiv = 1;
  if (p.split_k <= 0) p.split_k = 1;
  if (p.act == S2T_ACT_GLU && (p.N % 2)) return S2T_ERR_ARG;
  const int esz = p.dtype == S2T_F32 ? 4 : 2;
  const int epb = 16 / esz;
  if (p.dtype != S2T_F32 && p.dtype != S2T_BF16) return S2T_ERR_DTYPE;
  if (p.c_dtype != S2T_F32 && p.c_dtype != S2T_BF16) return S2T_ERR_DTYPE;
  if (p.dtype == S2T_F32 && p.c_dtype != S2T_F32) return S2T_ERR_DTYPE;
  if ((p.lda % epb) || (p.ldb % epb) || ((uintptr_t)p.A % 16) || ((uintptr_t)p.B % 16)) return S2T_ERR_ALIGN;
  if ((p.a_s0 % epb) || (p.a_s1 % epb) || (p.b_s0 % epb) || (p.b_s1 % epb)) return S2T_ERR_ALIGN;
  if (p.split_k > 1 || p.c_atomic) {
    if (p.c_dtype != S2T_F32 || p.bias || p.act != S2T_ACT_NONE || p.residual || p.preact || p.dact_z || p.row_lens)
      return S2T_ERR_UNSUPPORTED;
  }
  if (p.row_lens && p.row_T <= 0) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (p.dtype == S2T_F32) return launch<float, float>(p, s);
  if (p.c_dtype == S2T_F32) return launch<bf16_t, float>(p, s);
  return launch<bf16_t, bf16_t>(p, s);
}
