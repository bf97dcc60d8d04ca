// Persistent bf16 MFMA GEMM with an LDS-DMA ring (fast path of s2t_gemm: bf16 operands, K % 64 == 0).
//
// Why a second kernel: the generic kernel (gemm.hip) keeps ONE K-step of global loads in flight per workgroup and is
// latency-bound on this path's shapes (K = 256 ... 2048, 250 - 2000 output tiles): measured ~1.9 us per K-step against
// 0.21 us of MFMA work.  Here
//   * one 256-thread workgroup per CU owns all 160 KiB of LDS as a ring of NS = 5 stages (A 16 KiB + B 16 KiB each);
//   * operands go global -> LDS by `global_load_lds_dwordx4` (no VGPR staging), PF = 3 stages ahead, ACROSS output
//     tiles (the workgroup is persistent and walks its list of (tile, K-step) items), so the prologue latency and
//     the epilogue of one tile overlap the loads of the next;
//   * one raw `s_barrier` per K-step; a counted `s_waitcnt vmcnt(8*(stages still in flight))` makes only the stage
//     about to be consumed land (never vmcnt(0) in steady state); the stage that becomes free is refilled right
//     after the barrier (it was last read two steps ago);
//   * LDS images, XOR swizzles (applied to the per-lane SOURCE address, the LDS destination of an LDS-DMA is
//     lane-linear), MFMA fragment maps and the LDS-transposed epilogue are those of gemm_common.h; the C tile is
//     transposed through the two ring stages that are free at the end of a tile;
//   * items are dealt so that each XCD (blockIdx % 8 shares an L2) walks a contiguous range of tiles.
// Rows / columns beyond the matrix edge are CLAMPED to valid addresses (their products only reach outputs that are
// never stored); K never has a tail here (K % 64 == 0), which is what makes the clamp safe.
#include "gemm_common.h"

namespace {

constexpr int NS = 5;   // ring stages
constexpr int PF = 3;   // stages in flight ahead of the one being consumed
constexpr int STAGE_BYTES = 32768;
constexpr int LOADS_PER_STAGE = 8;  // global_load_lds instructions per wave per stage (4 for A + 4 for B)

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

struct Item {
  int tm, tn, z, kt0, kt1;
  bool valid;
};

struct Geom {
  int tiles_n, tiles, batch, split, ktiles, per, total, nout, bn_out;
};

__device__ __forceinline__ Item decode(const Geom& g, int vid) {
  Item it;
  it.valid = vid < g.total;
  if (!it.valid) {
    it.tm = it.tn = it.z = it.kt0 = it.kt1 = 0;
    return it;
  }
  // XCD-contiguous remap (bijective for any total): virtual id v = pass*G + block runs on XCD (block % 8)
  const int q = g.total / 8, r = g.total % 8;
  const int xcd = vid % 8, idx = vid / 8;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tile = id % g.tiles;
  const int rest = id / g.tiles;
  it.z = rest % g.batch;
  const int ks = rest / g.batch;
  it.tm = tile / g.tiles_n;
  it.tn = tile % g.tiles_n;
  it.kt0 = ks * g.per;
  it.kt1 = min(g.ktiles, it.kt0 + g.per);
  if (it.kt0 >= it.kt1) it.kt1 = it.kt0;  // empty split slice: zero K-steps (still "valid": nothing to do)
  return it;
}

// with a remap, ids of one XCD are only contiguous if every XCD has the same number of virtual ids per pass;
// virtual ids are vid = pass * gridDim.x + blockIdx.x with gridDim.x a multiple of 8 (launcher guarantees it).

template <bool KM, bool GLU_B>
__device__ __forceinline__ void issue_operand(char* lds_tile, const bf16_t* __restrict__ base, int64_t ld, int first,
                                              int extent, int k0, int tid, int glu_half) {
  // row-major: first/extent = first row, number of rows;  k-major: first/extent = first column, number of columns
  const int wave = tid >> 6;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void*)lds_tile;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cid = tid + 256 * u;
    const bf16_t* src;
    if constexpr (!KM) {
      const int r = cid >> 3, pc = cid & 7;
      const int c = pc ^ (r & 7);
      int grow;
      if constexpr (GLU_B) {
        const int s = r >> 4;
        const int o = min(first + (s >> 1) * 16 + (r & 15), glu_half - 1);
        grow = (s & 1) * glu_half + o;
      } else {
        grow = min(first + r, extent - 1);
      }
      src = base + (int64_t)grow * ld + k0 + c * 8;
    } else {
      const int kr = cid >> 4, pc = cid & 15;
      const int ch = pc ^ kswz(kr);
      const int last = ((extent + 7) & ~7) - 8;  // last chunk start that stays inside the padded row
      const int gc = min(first + ch * 8, last);
      src = base + (int64_t)(k0 + kr) * ld + gc;
    }
    // LDS-DMA issued through inline asm on purpose: with the builtin, hipcc (ROCm 7.2) sees an LDS write it cannot
    // disambiguate from the fragment ds_reads and drains the whole ring (s_waitcnt vmcnt(0)) before every K-step;
    // here the counted s_waitcnt in the main loop is the only wait (cdna_hip_programming.md §5.7).
    // M0 = wave-uniform LDS byte address; the DMA writes M0 + lane*16.
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((u * 256 + wave * 64) * 16));
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst)
        : "memory");
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ char* crow(char* c0, char* c1, int row) { return (row < 64 ? c0 : c1) + (row & 63) * 512; }

template <bool AKM, bool BKM, typename TC, bool GLU>
__global__ __launch_bounds__(256, 1) void gemm_ring_kernel(const s2t_gemm_args p, const Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = bf16_t;
  constexpr int BKE = 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int x = lane & 15, y = lane >> 4;
  const int G = gridDim.x;

  int ld_vid = blockIdx.x, cp_vid = blockIdx.x;
  Item ld = decode(g, ld_vid), cp = decode(g, cp_vid);
  int ld_kt = ld.kt0, cp_kt = cp.kt0;
  // skip empty items
  auto skip_empty = [&](Item& it, int& vid, int& kt) {
    while (it.valid && it.kt0 >= it.kt1) {
      vid += G;
      it = decode(g, vid);
      kt = it.kt0;
    }
  };
  skip_empty(ld, ld_vid, ld_kt);
  skip_empty(cp, cp_vid, cp_kt);

  auto issue = [&](int slot) {
    char* la = smem + slot * STAGE_BYTES;
    char* lb = la + 16384;
    const int z0 = ld.z / p.zdiv, z1 = ld.z % p.zdiv;
    const T* A = reinterpret_cast<const T*>(p.A) + z0 * p.a_s0 + z1 * p.a_s1;
    const T* B = reinterpret_cast<const T*>(p.B) + z0 * p.b_s0 + z1 * p.b_s1;
    const int k0 = ld_kt * BKE;
    issue_operand<AKM, false>(la, A, p.lda, ld.tm * BM, p.M, k0, tid, 0);
    issue_operand<BKM, GLU>(lb, B, p.ldb, ld.tn * g.bn_out, p.N, k0, tid, g.nout);
    ++ld_kt;
    if (ld_kt >= ld.kt1) {
      ld_vid += G;
      ld = decode(g, ld_vid);
      ld_kt = ld.kt0;
      skip_empty(ld, ld_vid, ld_kt);
    }
  };

  int issued = 0, consumed = 0;
#pragma unroll 1
  for (int i = 0; i < PF; ++i) {
    if (ld.valid) {
      issue(issued % NS);
      ++issued;
    }
  }

  f32x4 acc[4][4];

#pragma unroll 1
  while (cp.valid) {
    // ---- make the stage about to be consumed land (only it), then rendezvous
    const int younger = issued - consumed - 1;
    if (younger >= 2) wait_vmcnt<2 * LOADS_PER_STAGE>();
    else if (younger == 1) wait_vmcnt<LOADS_PER_STAGE>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- refill the stage that was consumed two steps ago
    if (ld.valid) {
      issue(issued % NS);
      ++issued;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA on the landed stage
    if (cp_kt == cp.kt0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int slot = consumed % NS;
    const char* la = smem + slot * STAGE_BYTES;
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
    ++consumed;
    ++cp_kt;
    if (cp_kt < cp.kt1) continue;

    // ================= epilogue of the finished tile =================
    // free stages: the one just consumed and the one before it (the refill of the latter happens after the NEXT barrier)
    char* c0 = smem + ((consumed - 1) % NS) * STAGE_BYTES;
    char* c1 = smem + ((consumed + NS - 2) % NS) * STAGE_BYTES;
    __builtin_amdgcn_s_barrier();  // every wave is done reading the just-consumed stage
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int chunk = wn * 16 + j * 4 + y;
        *reinterpret_cast<f32x4*>(crow(c0, c1, row) + ((chunk ^ (row & 7)) << 4)) = acc[i][j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int tm = cp.tm, tn = cp.tn, z = cp.z;
    const int z0 = z / p.zdiv, z1 = z % p.zdiv;
    const int64_t coff = z0 * p.c_s0 + z1 * p.c_s1;
    if (p.split_k > 1 || p.c_atomic) {
      float* C = reinterpret_cast<float*>(p.C) + coff;
      const int col = tid & 127;
      const int n = tn * BN + col;
      if (n < p.N) {
#pragma unroll 4
        for (int pass = 0; pass < 64; ++pass) {
          const int row = pass * 2 + (tid >> 7);
          const int m = tm * BM + row;
          if (m < p.M) {
            const float v = *reinterpret_cast<const float*>(crow(c0, c1, row) + (((col >> 2) ^ (row & 7)) << 4) + ((col & 3) << 2));
            atomicAdd(C + (int64_t)m * p.ldc + n, p.alpha * v);
          }
        }
      }
    } else {
      Epi<TC> e{p,
                reinterpret_cast<TC*>(p.C) + coff,
                p.residual ? reinterpret_cast<const TC*>(p.residual) + coff : nullptr,
                p.preact ? reinterpret_cast<TC*>(p.preact) + (z0 * p.p_s0 + z1 * p.p_s1) : nullptr,
                p.dact_z ? reinterpret_cast<const TC*>(p.dact_z) + coff : nullptr,
                g.nout,
                false, false, false, false};
      auto vec_ok = [](const void* ptr, int64_t ldx) { return ((ldx * (int64_t)sizeof(TC)) % 16 == 0) && (((uintptr_t)ptr) % 16 == 0); };
      e.vec_c = vec_ok(e.C, p.ldc);
      e.vec_r = e.R && vec_ok(e.R, p.ldr);
      e.vec_p = e.P && vec_ok(e.P, p.ldp);
      e.vec_z = e.Z && vec_ok(e.Z, p.ldz);
      const int c8 = tid & 7;
      auto ld8c = [&](int row, int col0, float (&v)[8]) {
        const char* rb = crow(c0, c1, row);
        const float4 a = *reinterpret_cast<const float4*>(rb + ((((col0 >> 2)) ^ (row & 7)) << 4));
        const float4 b = *reinterpret_cast<const float4*>(rb + ((((col0 >> 2) + 1) ^ (row & 7)) << 4));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      };
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 32 + (tid >> 3);
        const int m = tm * BM + row;
        if (m >= p.M) continue;
        const int64_t grow = (int64_t)z * p.M + m;
        if constexpr (GLU) {
          const int o0 = c8 * 8;
          const int n0 = tn * 64 + o0;
          if (n0 >= g.nout) continue;
          const int lcol = (o0 >> 4) * 32 + (o0 & 15);
          const int nv = min(8, g.nout - n0);
          float a[8], gt[8], ba[8], bg[8], v[8];
          ld8c(row, lcol, a);
          ld8c(row, lcol + 16, gt);
          e.bias8(n0, nv, ba);
          e.bias8(g.nout + n0, nv, bg);
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            a[r] += ba[r];
            gt[r] += bg[r];
            v[r] = a[r] * sigmoidf_(gt[r]);
          }
          if (e.P) {
            st8<TC>(e.P + (int64_t)m * p.ldp + n0, e.vec_p, nv, a);
            st8<TC>(e.P + (int64_t)m * p.ldp + g.nout + n0, e.vec_p && ((g.nout * (int)sizeof(TC)) % 16 == 0), nv, gt);
          }
          e.finish(m, n0, grow, v);
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int col0 = h * 64 + c8 * 8;
            const int n0 = tn * BN + col0;
            if (n0 >= g.nout) continue;
            float v[8], b[8];
            ld8c(row, col0, v);
            e.bias8(n0, min(8, g.nout - n0), b);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += b[r];
            e.finish(m, n0, grow, v);
          }
        }
      }
    }
    // next item
    cp_vid += G;
    cp = decode(g, cp_vid);
    cp_kt = cp.kt0;
    skip_empty(cp, cp_vid, cp_kt);
  }
}

int g_num_cu = 0;
bool g_optin[8] = {false, false, false, false, false, false, false, false};

template <bool AK, bool BK, typename TC, bool GLU>
int go(const s2t_gemm_args& p, const Geom& g, int slot, hipStream_t s) {
  auto kern = gemm_ring_kernel<AK, BK, TC, GLU>;
  if (!g_optin[slot]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE_BYTES);
    if (e != hipSuccess) return (int)e;
    g_optin[slot] = true;
  }
  int grid = g.total < g_num_cu ? g.total : g_num_cu;
  grid = (grid + 7) / 8 * 8;  // whole rounds over the 8 XCDs (surplus workgroups find no item and exit)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), NS * STAGE_BYTES, s, p, g);
  return S2T_LAUNCH_CHECK();
}

}  // namespace

// called by s2t_gemm (gemm.hip) for bf16 operands with K % 64 == 0; returns S2T_ERR_UNSUPPORTED to fall back
int s2t_gemm_ring_launch(const s2t_gemm_args& p, void* stream) {
  if (p.dtype != S2T_BF16 || p.K % 64 != 0 || p.K == 0 || p.colsum_a) return S2T_ERR_UNSUPPORTED;
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return S2T_ERR_UNSUPPORTED;
    g_num_cu = prop.multiProcessorCount;
  }
  const bool glu = p.act == S2T_ACT_GLU;
  if (glu && (p.a_kmajor || p.b_kmajor)) return S2T_ERR_UNSUPPORTED;
  Geom g;
  g.nout = glu ? p.N / 2 : p.N;
  g.bn_out = glu ? 64 : 128;
  g.tiles_n = (g.nout + g.bn_out - 1) / g.bn_out;
  g.tiles = ((p.M + BM - 1) / BM) * g.tiles_n;
  g.batch = p.batch;
  g.split = p.split_k;
  g.ktiles = p.K / 64;
  g.per = (g.ktiles + g.split - 1) / g.split;
  const int64_t total = (int64_t)g.tiles * g.batch * g.split;
  if (total > (1 << 30)) return S2T_ERR_UNSUPPORTED;
  g.total = (int)total;
  hipStream_t s = (hipStream_t)stream;
  const bool cf32 = p.c_dtype == S2T_F32;
  if (glu) return cf32 ? go<false, false, float, true>(p, g, 0, s) : go<false, false, bf16_t, true>(p, g, 1, s);
  if (!p.a_kmajor && !p.b_kmajor) return cf32 ? go<false, false, float, false>(p, g, 2, s) : go<false, false, bf16_t, false>(p, g, 3, s);
  if (!p.a_kmajor && p.b_kmajor) return cf32 ? go<false, true, float, false>(p, g, 4, s) : go<false, true, bf16_t, false>(p, g, 5, s);
  if (p.a_kmajor && !p.b_kmajor) return cf32 ? go<true, false, float, false>(p, g, 6, s) : go<true, false, bf16_t, false>(p, g, 7, s);
  // both k-major (wgrad): separate opt-in slots would exceed the table; reuse by symbol-specific statics
  static bool optin_f = false, optin_b = false;
  if (cf32) {
    auto kern = gemm_ring_kernel<true, true, float, false>;
    if (!optin_f) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE_BYTES);
      if (e != hipSuccess) return (int)e;
      optin_f = true;
    }
    int grid = g.total < g_num_cu ? g.total : g_num_cu;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), NS * STAGE_BYTES, s, p, g);
  } else {
    auto kern = gemm_ring_kernel<true, true, bf16_t, false>;
    if (!optin_b) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE_BYTES);
      if (e != hipSuccess) return (int)e;
      optin_b = true;
    }
    int grid = g.total < g_num_cu ? g.total : g_num_cu;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), NS * STAGE_BYTES, s, p, g);
  }
  return S2T_LAUNCH_CHECK();
}
