// Shared device/host helpers for libs2t_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s2t_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits in HBM

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define S2T_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// two fp32 -> one dword of two bf16 (a in the low half): ONE v_cvt_pk_bf16_f32.  (The scalar form
// (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16) compiles to two conversions, a shift and an SDWA or: four VALU issues per pair
// in epilogues that are VALU-issue bound.)
typedef float s2t_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 s2t_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16pack(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((s2t_f32x2){a, b}, s2t_bf16x2));
}

template <typename T>
__device__ __forceinline__ float ld_as_f32(const T* p);
template <>
__device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T>
__device__ __forceinline__ void st_from_f32(T* p, float v);
template <>
__device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st_from_f32<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 4 consecutive elements
template <typename T>
__device__ __forceinline__ void ld4_as_f32(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void ld4_as_f32<float>(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void ld4_as_f32<bf16_t>(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void st4_from_f32(T* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void st4_from_f32<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void st4_from_f32<bf16_t>(bf16_t* p, const float (&v)[4]) {
  uint2 t;
  t.x = bf16pack(v[0], v[1]);
  t.y = bf16pack(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}

// v_rcp_f32 (1 ulp) instead of an IEEE division: the fused GEMM epilogues are VALU-issue bound
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float act_apply(int act, float x) {
  if (act == S2T_ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == S2T_ACT_SWISH) return x * sigmoidf_(x);
  return x;
}
// derivative of the activation w.r.t. its pre-activation input z
__device__ __forceinline__ float act_grad(int act, float z) {
  if (act == S2T_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == S2T_ACT_SWISH) {
    float s = sigmoidf_(z);
    return s * (1.f + z * (1.f - s));
  }
  return 1.f;
}

// ---- cross-lane steps without the LDS crossbar -----------------------------------------------------------------------------
// `__shfl_xor` compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): an LDS round trip (100+ cycles, behind whatever the LDS
// queue holds) per butterfly step, ten of them one after the other in a LayerNorm prologue.  The same exchanges as vector
// instructions: DPP inside a 16-lane row (quad_perm for xor 1 / 2, two bank-masked row shifts for xor 4, a row rotation for
// xor 8), v_permlane16_swap / v_permlane32_swap (gfx950) across rows.  s2t_xadd<O>(v) == v + __shfl_xor(v, O) bit for bit
// (the add is commutative), likewise s2t_xmax.  tools/ubench/red_dpp.hip checks every step against __shfl_xor on the GPU.
template <int O>
__device__ __forceinline__ float s2t_lane_xor_dpp(float v) {   // value of lane ^ O, O in {1, 2, 4, 8}
  const int iv = __builtin_bit_cast(int, v);
  if constexpr (O == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
  else if constexpr (O == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
  else if constexpr (O == 4) {
    int t = __builtin_amdgcn_update_dpp(iv, iv, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0, 2: lane i <- lane i + 4
    t = __builtin_amdgcn_update_dpp(t, iv, 0x114, 0xf, 0xA, false);        // row_shr:4 into banks 1, 3: lane i <- lane i - 4
    return __builtin_bit_cast(float, t);
  } else {
    static_assert(O == 8, "DPP covers xor 1, 2, 4, 8");
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x128, 0xf, 0xf, false));  // row_ror:8
  }
}
// v_permlane16_swap a, b: rows 1 and 3 of a change places with rows 0 and 2 of b.  With a = b = v: a' = [r0 r0 r2 r2],
// b' = [r1 r1 r3 r3] — every lane holds its own value and its xor-16 partner's (in one order or the other).
// (written as inline assembly: hipcc 7.2 hands back the FIRST result for both elements of __builtin_amdgcn_permlane16_swap's
// pair — `v_add_f32 v1, v1, v1` behind the swap.  The s_nop covers the two wait states a VALU write needs before a permlane
// reads the register; the compiler cannot see into the asm.)
__device__ __forceinline__ void s2t_pair16(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void s2t_pair32(float v, float& a, float& b) {   // the same across the wave's halves (xor 32)
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int O>
__device__ __forceinline__ float s2t_xadd(float v) {
  if constexpr (O == 16) { float a, b; s2t_pair16(v, a, b); return a + b; }
  else if constexpr (O == 32) { float a, b; s2t_pair32(v, a, b); return a + b; }
  else return v + s2t_lane_xor_dpp<O>(v);
}
template <int O>
__device__ __forceinline__ float s2t_xmax(float v) {
  if constexpr (O == 16) { float a, b; s2t_pair16(v, a, b); return fmaxf(a, b); }
  else if constexpr (O == 32) { float a, b; s2t_pair32(v, a, b); return fmaxf(a, b); }
  else return fmaxf(v, s2t_lane_xor_dpp<O>(v));
}
// butterflies in the order of the loops they replace (widest step first): results bit-equal to those loops
__device__ __forceinline__ float s2t_sum16(float v) { return s2t_xadd<1>(s2t_xadd<2>(s2t_xadd<4>(s2t_xadd<8>(v)))); }
__device__ __forceinline__ float s2t_sum16_up(float v) { return s2t_xadd<8>(s2t_xadd<4>(s2t_xadd<2>(s2t_xadd<1>(v)))); }  // narrowest step first
__device__ __forceinline__ float s2t_sum32(float v) { return s2t_sum16(s2t_xadd<16>(v)); }
__device__ __forceinline__ float s2t_sum64(float v) { return s2t_sum32(s2t_xadd<32>(v)); }
__device__ __forceinline__ float s2t_max16(float v) { return s2t_xmax<1>(s2t_xmax<2>(s2t_xmax<4>(s2t_xmax<8>(v)))); }
__device__ __forceinline__ float s2t_max32(float v) { return s2t_max16(s2t_xmax<16>(v)); }
__device__ __forceinline__ float s2t_max64(float v) { return s2t_max32(s2t_xmax<32>(v)); }
__device__ __forceinline__ float wave_sum(float v) { return s2t_sum64(v); }
__device__ __forceinline__ float wave_max(float v) { return s2t_max64(v); }

// Counter-based dropout RNG: keep(seed, site, element index) is a pure function, so the backward pass regenerates
// the mask instead of storing it.  One 32-bit hash serves TWO consecutive elements (16 random bits each; the keep
// probability is quantised to 1/65536 and the survivors are scaled by the quantised value's inverse): the fused GEMM
// and attention epilogues draw a mask per stored element and are VALU-issue bound, and 32-bit integer multiplies and
// 64-bit arithmetic (the first version was splitmix64 per element) are the expensive part.
// hash = ONE round of the "lowbias32" multiply/xor-shift finaliser over (pair index ^ low key half), the high key half
// (and, for tensors beyond 2^33 elements, the pair index's high word) added behind it with full-rate ops: two quarter-rate
// 32-bit multiplies per 32 random bits (the first form took five).
__device__ __forceinline__ uint32_t s2t_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t s2t_pair_hash(uint64_t key, uint64_t pair) {
  const uint32_t hi = (uint32_t)(pair >> 32);
  return s2t_mix32((uint32_t)pair ^ (uint32_t)key) ^ ((uint32_t)(key >> 32) + ((hi << 16) | (hi >> 16)));
}
// 16 uniform random bits of element idx
__device__ __forceinline__ uint32_t s2t_rand_u32(uint64_t key, uint64_t idx) {
  const uint32_t h = s2t_pair_hash(key, idx >> 1);
  return (idx & 1) ? (h >> 16) : (h & 0xffffu);
}
// the same for N consecutive elements starting at base: N/2 hashes when base is even (+1 when odd)
template <int N>
__device__ __forceinline__ void s2t_rand_run(uint64_t key, uint64_t base, uint32_t (&r16)[N]) {
  static_assert(N % 2 == 0, "even run length");
  const uint64_t p0 = base >> 1;
  uint32_t h[N / 2];
#pragma unroll
  for (int q = 0; q < N / 2; ++q) h[q] = s2t_pair_hash(key, p0 + q);
  if ((base & 1) == 0) {
#pragma unroll
    for (int r = 0; r < N; ++r) r16[r] = (r & 1) ? (h[r >> 1] >> 16) : (h[r >> 1] & 0xffffu);
  } else {
    const uint32_t hl = s2t_pair_hash(key, p0 + N / 2);
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const uint32_t hh = ((r + 1) >> 1) < N / 2 ? h[((r + 1) >> 1) < N / 2 ? ((r + 1) >> 1) : 0] : hl;
      r16[r] = ((r + 1) & 1) ? (hh >> 16) : (hh & 0xffffu);
    }
  }
}
// The same bits for N consecutive elements starting at an EVEN index below 2^32 (32-bit arithmetic throughout, no odd-start
// variant to select from): what the row-block kernels use, whose element indices are row * F + multiple of 4.
template <int N>
__device__ __forceinline__ void s2t_rand_run_even32(uint64_t key, uint32_t base, uint32_t (&r16)[N]) {
  static_assert(N % 2 == 0, "even run length");
  const uint32_t p0 = base >> 1, klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
#pragma unroll
  for (int q = 0; q < N / 2; ++q) {
    const uint32_t h = s2t_mix32((p0 + (uint32_t)q) ^ klo) ^ khi;
    r16[2 * q] = h & 0xffffu;
    r16[2 * q + 1] = h >> 16;
  }
}
__device__ __forceinline__ uint64_t s2t_drop_key_of(uint64_t seed, uint32_t site) {
  return (seed * 0xD1342543DE82EF95ull) ^ ((uint64_t)site * 0xA24BAED4963EE407ull);
}
__device__ __forceinline__ uint64_t s2t_drop_key(const uint64_t* seed_ptr, uint32_t site) {
  return s2t_drop_key_of(seed_ptr ? *seed_ptr : 0ull, site);
}
// element kept iff its 16 random bits >= thresh; survivors scaled by s2t_drop_scale
__device__ __forceinline__ uint32_t s2t_drop_thresh(float p) { return (uint32_t)fminf(p * 65536.0f + 0.5f, 65535.0f); }
__device__ __forceinline__ float s2t_drop_scale(float p) { return 65536.0f / (65536.0f - (float)s2t_drop_thresh(p)); }

// ---- row geometry of an activation matrix (include/s2t_hip.h, "Packed rows") ------------------------------------------
// (lens, T) with T > 0: uniform layout, row m = b * T + t, frame valid iff t < lens[b].
// T == S2T_ROWS_PACKED / S2T_ROWS_BOUND: `lens` is a ROW MAP of a packed batch — lens[m] >= 0 on rows that hold a frame
// ((utterance << 16) | frame), < 0 on rows that hold none (the halo rows behind an utterance: masked like padded frames);
// lens[-1] = live rows: rows at and beyond it are neither computed nor stored.  S2T_ROWS_BOUND applies the bound only.
__device__ __forceinline__ bool s2t_row_masked(const int32_t* __restrict__ lens, int T, int64_t row) {
  if (T > 0) return (int)(row % T) >= lens[row / T];
  return T == S2T_ROWS_PACKED && lens[row] < 0;
}
// 32-bit form for epilogues (row < 2^31): the 64-bit division is a long software sequence
// The test in two halves: s2t_row_mask_entry REQUESTS what the test of a row needs (one load, or a constant that never masks
// when there is no mask), s2t_row_mask_test evaluates it.  A prologue that masks several rows requests all entries first and
// tests them where it uses them: with request and test in one expression (a bool per row) the compiler waits for each entry
// where it is requested — one memory round trip per row, one after the other.
__device__ __forceinline__ int s2t_row_mask_entry(const int32_t* __restrict__ lens, int T, uint32_t row) {
  if (!lens || (T <= 0 && T != S2T_ROWS_PACKED)) return T > 0 ? 0x7fffffff : 0;
  return lens[T > 0 ? row / (uint32_t)T : row];
}
__device__ __forceinline__ bool s2t_row_mask_test(int T, uint32_t row, int entry) {
  return T > 0 ? (int)(row % (uint32_t)T) >= entry : entry < 0;
}
__device__ __forceinline__ bool s2t_row_masked32(const int32_t* __restrict__ lens, int T, uint32_t row) {
  return s2t_row_mask_test(T, row, s2t_row_mask_entry(lens, T, row));
}
// rows of a launch: the host's bound, or the live row count of a packed batch when that is smaller
__device__ __forceinline__ int64_t s2t_live_rows(const int32_t* __restrict__ lens, int T, int64_t rows) {
  if (lens && T < 0) {
    const int64_t n = lens[-1];
    return n < rows ? n : rows;
  }
  return rows;
}
// first row and row capacity of utterance b: cu[b] .. cu[b + 1] of a packed batch, b * T .. + T otherwise
__device__ __forceinline__ int s2t_utt_row0(const int32_t* __restrict__ cu, int b, int T) { return cu ? cu[b] : b * T; }
__device__ __forceinline__ int s2t_utt_rows(const int32_t* __restrict__ cu, int b, int T) { return cu ? cu[b + 1] - cu[b] : T; }

// host-side check of a (lens, T) argument pair: T > 0, or one of the row-map forms
static inline bool s2t_rows_arg_bad(const int32_t* lens, int T) {
  return lens && T <= 0 && T != S2T_ROWS_PACKED && T != S2T_ROWS_BOUND;
}

static inline int s2t_hip_status(hipError_t e) { return e == hipSuccess ? S2T_OK : (int)e; }
#define S2T_LAUNCH_CHECK() s2t_hip_status(hipGetLastError())
