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
#include <stdlib.h>

#include "gemm_common.h"

namespace {

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

  // optional fused bias gradient: column sums of the (k-major) A operand, taken from the staged registers by the
  // workgroups of the first tile column (every A tile is staged exactly once per such workgroup)
  constexpr int CPR_A = 128 / EPB;
  const bool do_colsum = AKM && p.colsum_a != nullptr && tn == 0;
  float csum[EPB];
#pragma unroll
  for (int e = 0; e < EPB; ++e) csum[e] = 0.f;
  auto colsum_acc = [&]() {
    if constexpr (AKM) {
      if (do_colsum) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t w[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w};
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              csum[2 * q] += __uint_as_float(w[q] << 16);
              csum[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) csum[q] += __uint_as_float(w[q]);
          }
        }
      }
    }
  };

  if (kt0 < kt1) {
    gload(kt0);
    colsum_acc();
    lstore(0);
  }
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) {
      gload(kt + 1);
      colsum_acc();
    }
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

  if constexpr (AKM) {
    if (do_colsum) {  // workgroup-uniform
      float* lc = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int e = 0; e < EPB; ++e) lc[tid * EPB + e] = csum[e];
      __syncthreads();
      if (tid < 128) {
        const int ch = tid / EPB, e = tid % EPB;
        float sum = 0.f;
        for (int j = 0; j < 256 / CPR_A; ++j) sum += lc[(ch + CPR_A * j) * EPB + e];
        const int m = tm * BM + tid;
        if (m < p.M) atomicAdd(p.colsum_a + m, p.alpha * sum);
      }
      __syncthreads();
    }
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

int s2t_gemm_ring_launch(const s2t_gemm_args& p, void* stream);  // gemm_ring.hip

// The persistent LDS-DMA ring kernel (gemm_ring.hip) is opt-in (S2T_GEMM_RING=1): measured on MI355X it ties the
// generic kernel at K = 2048 and loses at K = 256 and on 8192^3 (512 vs 810 TFLOP/s) because one 4-wave workgroup per
// CU leaves nothing to overlap the DMA issue / epilogue with (profiles/r01_gemm_microbench.txt).
static bool use_ring() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("S2T_GEMM_RING");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

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
    if (p.c_dtype != S2T_F32 || p.bias || p.act != S2T_ACT_NONE || p.residual || p.preact || p.dact_z || p.row_lens || p.drop_p > 0.f)
      return S2T_ERR_UNSUPPORTED;
  }
  if (p.row_lens && p.row_T <= 0) return S2T_ERR_ARG;
  if (p.drop_p < 0.f || p.drop_p >= 1.f) return S2T_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (p.dtype == S2T_BF16 && p.K % 64 == 0 && p.K > 0 && use_ring()) {
    const int rc = s2t_gemm_ring_launch(p, stream);
    if (rc != S2T_ERR_UNSUPPORTED) return rc;
  }
  if (p.dtype == S2T_F32) return launch<float, float>(p, s);
  if (p.c_dtype == S2T_F32) return launch<bf16_t, float>(p, s);
  return launch<bf16_t, bf16_t>(p, s);
}
