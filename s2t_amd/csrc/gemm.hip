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

// ---- epilogue on one lane's 4 consecutive output columns --------------------------------------
template <typename TC>
struct Epi {
  const s2t_gemm_args& p;
  TC* C;
  const TC* R;
  TC* P;
  const TC* Z;
  int nout;        // output columns (N, or N/2 under GLU)
  bool vec_c, vec_r, vec_p, vec_z;

  __device__ __forceinline__ float bias_at(int n) const {
    if (!p.bias) return 0.f;
    return p.bias_dtype == S2T_F32 ? reinterpret_cast<const float*>(p.bias)[n]
                                   : bf2f(reinterpret_cast<const bf16_t*>(p.bias)[n]);
  }
  __device__ __forceinline__ bool row_masked(int64_t grow) const {
    if (!p.row_lens) return false;
    const int b = (int)(grow / p.row_T), t = (int)(grow % p.row_T);
    return t >= p.row_lens[b];
  }
  template <typename X>
  __device__ __forceinline__ void ld(const X* ptr, bool vec, int nv, float (&o)[4]) const {
    if (vec && nv == 4) {
      ld4_as_f32<X>(ptr, o);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = r < nv ? ld_as_f32<X>(ptr + r) : 0.f;
    }
  }
  template <typename X>
  __device__ __forceinline__ void st(X* ptr, bool vec, int nv, const float (&o)[4]) const {
    if (vec && nv == 4) {
      st4_from_f32<X>(ptr, o);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < nv) st_from_f32<X>(ptr + r, o[r]);
    }
  }
  // v: post-bias (post-GLU) values for output columns n0..n0+3 of row m
  __device__ __forceinline__ void finish(int m, int n0, int64_t grow, float (&v)[4]) const {
    const int nv = min(4, nout - n0);
    if (p.act == S2T_ACT_RELU || p.act == S2T_ACT_SWISH) {
      if (P) st<TC>(P + (int64_t)m * p.ldp + n0, vec_p, nv, v);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = act_apply(p.act, v[r]);
    }
    if (Z) {
      float z[4];
      ld<TC>(Z + (int64_t)m * p.ldz + n0, vec_z, nv, z);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= act_grad(p.dact, z[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
    if (row_masked(grow)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = 0.f;
    }
    if (R) {
      float q[4];
      ld<TC>(R + (int64_t)m * p.ldr + n0, vec_r, nv, q);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += q[r];
    }
    st<TC>(C + (int64_t)m * p.ldc + n0, vec_c, nv, v);
  }
};

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
  if (p.split_k > 1 || p.c_atomic) {
    float* C = reinterpret_cast<float*>(p.C) + coff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = tm * BM + wm * 64 + i * 16 + x;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n0 = tn * BN + wn * 64 + j * 16 + 4 * y;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < p.N) atomicAdd(C + (int64_t)m * p.ldc + n0 + r, p.alpha * acc[i][j][r]);
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
  constexpr int VB = 4 * (int)sizeof(TC);  // vector bytes
  e.vec_c = (p.ldc % 4 == 0) && (((uintptr_t)e.C) % VB == 0);
  e.vec_r = e.R && (p.ldr % 4 == 0) && (((uintptr_t)e.R) % VB == 0);
  e.vec_p = e.P && (p.ldp % 4 == 0) && (((uintptr_t)e.P) % VB == 0);
  e.vec_z = e.Z && (p.ldz % 4 == 0) && (((uintptr_t)e.Z) % VB == 0);

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = tm * BM + wm * 64 + i * 16 + x;
    if (m >= p.M) continue;
    const int64_t grow = (int64_t)z * p.M + m;
    if constexpr (GLU) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int n0 = tn * 64 + wn * 32 + jp * 16 + 4 * y;
        if (n0 >= nout) continue;
        float a[4], g[4], v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = min(n0 + r, nout - 1);
          a[r] = acc[i][2 * jp][r] + e.bias_at(n);
          g[r] = acc[i][2 * jp + 1][r] + e.bias_at(nout + n);
          v[r] = a[r] * sigmoidf_(g[r]);
        }
        if (e.P) {
          const int nv = min(4, nout - n0);
          e.template st<TC>(e.P + (int64_t)m * p.ldp + n0, e.vec_p, nv, a);
          e.template st<TC>(e.P + (int64_t)m * p.ldp + nout + n0, e.vec_p && (nout % 4 == 0), nv, g);
        }
        e.finish(m, n0, grow, v);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n0 = tn * BN + wn * 64 + j * 16 + 4 * y;
        if (n0 >= nout) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + e.bias_at(min(n0 + r, nout - 1));
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
