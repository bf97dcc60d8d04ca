// Shared device code of the GEMM kernels (gemm.hip: generic register-staged; gemm_grouped.hip: grouped weight
// gradients): LDS images, MFMA fragment maps and the fused epilogue.
#pragma once
#include "common.h"

#ifndef S2T_DBG_EPI
#define S2T_DBG_EPI 0  // kernel-experiment switch (tools/dbg_build.sh); 0 in every shipped build
#endif

#define BM 128
#define BN 128


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

// 16-byte global load as a native vector (a HIP_vector_type struct copy lowers to llvm.memcpy into the private
// array, which kept the staging registers in scratch memory)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldg_u4(const void* ptr) {
  const u32x4_t t = *reinterpret_cast<const u32x4_t*>(ptr);
  return make_uint4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ int kswz(int k) { return 2 * ((k & 3) | (((k >> 3) & 1) << 2)); }

// ---- global -> registers ---------------------------------------------------------------------
// The loads are UNCONDITIONAL and branch-free: a bounds branch or a tail mask next to a load makes the compiler wait
// for every load individually (s_waitcnt vmcnt(0) after each), which serialises the 8 loads of a K-step, and the
// zero-fill code of ragged tiles bloats a kernel whose per-tile fixed cost is instruction fetch.
//   * rows / columns beyond M or N are CLAMPED to the last valid one: they only feed accumulator rows / columns that
//     the epilogue never stores (an MFMA output element depends on its own operand row and column only);
//   * K beyond the problem's K must be ZERO.  Only the KT (K % BKE != 0) instantiations carry that code: chunks past
//     K read offset 0 of the row and are zeroed / tail-masked by fix_*() right before the LDS store.
// row-major operand: element (row, k) at base[row*ld + k]
template <typename T, bool GLU_B>
__device__ __forceinline__ int rowmajor_row(int r, int row0, int nrows, int glu_half_rows) {
  if constexpr (GLU_B) {
    // 16-row blocks alternate value / gate rows of the weight: block s -> half = s&1
    const int s = r >> 4;
    const int o = min(row0 + (s >> 1) * 16 + (r & 15), glu_half_rows - 1);  // output column; row0 = tn*64
    return (s & 1) * glu_half_rows + o;
  } else {
    return min(row0 + r, nrows - 1);
  }
}

// Per-thread load plan of one operand tile: 32-bit BYTE offsets (from the operand base) of the thread's four 16-byte
// chunks at K-step 0.  A K-step's loads are then "uniform 64-bit base + 32-bit VGPR offset" (the global_load saddr
// form): no per-load 64-bit address arithmetic in the main loop, whose cost on a wave64 is VALU issue slots.
// The host guarantees every operand spans less than 4 GiB.
struct LoadPlan {
  uint32_t off[4];
};

template <typename T, bool GLU_B>
__device__ __forceinline__ void plan_rowmajor(LoadPlan& pl, int64_t ld, int row0, int nrows, int tid, int glu_half_rows) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cid = tid + 256 * u;
    const int r = cid >> 3, ch = cid & 7;
    const int grow = rowmajor_row<T, GLU_B>(r, row0, nrows, glu_half_rows);
    pl.off[u] = (uint32_t)((int64_t)grow * ld * (int64_t)sizeof(T)) + (uint32_t)(ch * 16);
  }
}
template <typename T, bool KT>
__device__ __forceinline__ void load_rowmajor(uint4 (&reg)[4], const char* __restrict__ base, const LoadPlan& pl, int k0,
                                              int K, int tid) {
  constexpr int EPB = TileTraits<T>::EPB;
  if constexpr (!KT) {
    const char* bk = base + (int64_t)k0 * (int64_t)sizeof(T);  // workgroup-uniform
#pragma unroll
    for (int u = 0; u < 4; ++u) reg[u] = ldg_u4(bk + pl.off[u]);
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = (tid + 256 * u) & 7;
      // chunks beyond K read the start of their row (zeroed by fix_rowmajor)
      const uint32_t kterm = (k0 + ch * EPB < K) ? (uint32_t)(k0 * (int)sizeof(T)) : (uint32_t)(-(ch * 16));
      reg[u] = ldg_u4(base + (uint32_t)(pl.off[u] + kterm));
    }
  }
}
template <typename T>
__device__ __forceinline__ void fix_rowmajor(uint4 (&reg)[4], int k0, int K, int tid) {
  constexpr int EPB = TileTraits<T>::EPB;
  const bool ktail = (K % EPB) != 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ch = (tid + 256 * u) & 7;
    const int k = k0 + ch * EPB;
    if (k >= K) reg[u] = make_uint4(0, 0, 0, 0);
    else if (ktail && k + EPB > K) mask_tail<T>(reg[u], K - k);
  }
}

// k-major operand: element (k, col) at base[k*ld + col]
template <typename T>
__device__ __forceinline__ void plan_kmajor(LoadPlan& pl, int64_t ld, int col0, int ncols, int tid) {
  constexpr int EPB = TileTraits<T>::EPB;
  constexpr int CPR = 128 / EPB;  // chunks per k-row: 16 (bf16) / 32 (f32)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cid = tid + 256 * u;
    const int kr = cid / CPR, ch = cid % CPR;
    int gc = col0 + ch * EPB;
    gc = gc < ncols ? gc : 0;
    pl.off[u] = (uint32_t)((int64_t)kr * ld * (int64_t)sizeof(T)) + (uint32_t)(gc * (int)sizeof(T));
  }
}
template <typename T, bool KT>
__device__ __forceinline__ void load_kmajor(uint4 (&reg)[4], const char* __restrict__ base, const LoadPlan& pl, int64_t ld,
                                            int k0, int K, int tid) {
  constexpr int CPR = 128 / TileTraits<T>::EPB;
  if constexpr (!KT) {
    const char* bk = base + (int64_t)k0 * ld * (int64_t)sizeof(T);  // workgroup-uniform
#pragma unroll
    for (int u = 0; u < 4; ++u) reg[u] = ldg_u4(bk + pl.off[u]);
  } else {
    const uint32_t ldb = (uint32_t)(ld * (int64_t)sizeof(T));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kr = (tid + 256 * u) / CPR;
      // k-rows beyond K read k-row 0 (zeroed by fix_kmajor)
      const uint32_t kterm = (k0 + kr < K) ? (uint32_t)k0 * ldb : (uint32_t)(-kr) * ldb;
      reg[u] = ldg_u4(base + (uint32_t)(pl.off[u] + kterm));
    }
  }
}
template <typename T>
__device__ __forceinline__ void fix_kmajor(uint4 (&reg)[4], int k0, int K, int tid) {
  constexpr int CPR = 128 / TileTraits<T>::EPB;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kr = (tid + 256 * u) / CPR;
    if (k0 + kr >= K) reg[u] = make_uint4(0, 0, 0, 0);
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
      t.x = bf16pack(o[0], o[1]);
      t.y = bf16pack(o[2], o[3]);
      t.z = bf16pack(o[4], o[5]);
      t.w = bf16pack(o[6], o[7]);
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

// VEC instantiations assume 16-byte aligned rows / pointers and N % 8 == 0 for every tensor the epilogue touches
// (checked on the host): they carry no scalar fallbacks, which keeps the once-per-tile epilogue code short.
template <typename TC, bool VEC = false>
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
      ld8<float>(bp, VEC || ((uintptr_t)bp % 16) == 0, VEC ? 8 : nv, b);
    } else {
      const bf16_t* bp = reinterpret_cast<const bf16_t*>(p.bias) + n0;
      ld8<bf16_t>(bp, VEC || ((uintptr_t)bp % 16) == 0, VEC ? 8 : nv, b);
    }
  }
  __device__ __forceinline__ bool row_masked(int64_t grow) const {
    if (!p.row_lens) return false;
    // 32-bit division (the host rejects batch * M >= 2^31 rows): the 64-bit one is a long software sequence
    return s2t_row_masked32(p.row_lens, p.row_T, (uint32_t)grow);
  }
  // v: post-bias (post-GLU) values for output columns n0..n0+7 of row m
  __device__ __forceinline__ void finish(int m, int n0, int64_t grow, float (&v)[8]) const {
    const int nv = VEC ? 8 : min(8, nout - n0);
    if (p.act == S2T_ACT_RELU || p.act == S2T_ACT_SWISH) {
      if (P) st8<TC>(P + (int64_t)m * p.ldp + n0, VEC || vec_p, nv, v);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = act_apply(p.act, v[r]);
    }
    if (Z) {
      float z[8];
      ld8<TC>(Z + (int64_t)m * p.ldz + n0, VEC || vec_z, nv, z);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] *= act_grad(p.dact, z[r]);
    }
    if (p.drop_p > 0.f) {
      const uint64_t key = s2t_drop_key(p.drop_seed, p.drop_site);
      const uint32_t th = s2t_drop_thresh(p.drop_p);
      const float inv = s2t_drop_scale(p.drop_p);
      const uint64_t base = (uint64_t)grow * (uint64_t)nout + (uint64_t)n0;
      uint32_t r16[8];
      s2t_rand_run<8>(key, base, r16);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = r16[r] >= th ? v[r] * inv : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= p.alpha;
    if (row_masked(grow)) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = 0.f;
    }
    if (R) {
      float q[8];
      ld8<TC>(R + (int64_t)m * p.ldr + n0, VEC || vec_r, nv, q);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += q[r];
    }
#if S2T_DBG_EPI == 1
    if (nout < 0)
#endif
    st8<TC>(C + (int64_t)m * p.ldc + n0, VEC || vec_c, nv, v);
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

