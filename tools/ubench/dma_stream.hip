// How fast does the L2 -> LDS stream (buffer_load ... lds, 1 KiB per wave instruction) run on gfx950, per CU, as a function of
// how many CUs stream at once, how many pieces a workgroup keeps in flight, and where the bytes come from?  The K loops of
// gemm256.hip and ffn_pc.hip both settle at ~64 KiB per ~3.8 k cycles (17 B/clk/CU) with every CU streaming; this measures
// whether that is a per-CU limit (in-flight pieces x latency) or a chip-wide one.
//   workgroup = 512 threads, 128 KiB of LDS; a "step" = 64 pieces of 1 KiB (8 per wave) from a region of `span` bytes that is
//   either private to the workgroup (span * blocks bytes in total: past L2 when large) or shared by all of them (L2 resident);
//   DEPTH steps are kept in flight (1: issue, wait, barrier; 2: the next step is issued before the wait for this one).
// hipcc --offload-arch=gfx950 -O3 dma_stream.hip -o dma_stream && ./dma_stream
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(uint32_t lds_base, uint32_t voff, i32x4 srd, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
// WORK bit 2: ONE wave (wave 0) issues all 8 * PIECES pieces of a step, the others none;  bit 3: two waves (0 and 1) half each
// WORK bit 0: 24 ds_read_b128 per wave and step from the stage that is NOT being written (the fragment reads of a 256 x 256 x 64
// step), bit 1: 64 v_mfma_f32_16x16x32_bf16 on them
template <int DEPTH, int PIECES, int WORK = 0>
__global__ __launch_bounds__(512) void k(const char* src, uint32_t span, int shared_src, int steps, unsigned long long* cyc, float* sink = nullptr) {
  __shared__ __attribute__((aligned(16))) char smem[131072];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (shared_src ? 0 : (size_t)blockIdx.x * span);
  i32x4 srd;
  srd.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)base);
  srd.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)base >> 32));
  srd.z = (int)span;
  srd.w = 0x00020000;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t step_bytes = 8u * PIECES * 1024u;
  auto issue = [&](int s) __attribute__((always_inline)) {
    const uint32_t so = (uint32_t)(((uint64_t)s * step_bytes) % (span - step_bytes + 1)) & ~1023u;
    if constexpr (WORK & 4) {
      if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 8 * PIECES; ++q)
          dma16(lds0 + (uint32_t)((s & 1) * 65536 + (q * 1024) % 65536), (uint32_t)(lane * 16), srd, so + (uint32_t)(q * 1024));
      }
    } else if constexpr (WORK & 16) {  // four loader waves (0-3: one per SIMD), 2 * PIECES pieces each
      if (wave < 4) {
#pragma unroll
        for (int q = 0; q < 2 * PIECES; ++q)
          dma16(lds0 + (uint32_t)((s & 1) * 65536 + ((wave * 2 * PIECES + q) * 1024) % 65536), (uint32_t)(lane * 16), srd,
                so + (uint32_t)((wave * 2 * PIECES + q) * 1024));
      }
    } else if constexpr (WORK & 8) {
      if (wave < 2) {
#pragma unroll
        for (int q = 0; q < 4 * PIECES; ++q)
          dma16(lds0 + (uint32_t)((s & 1) * 65536 + ((wave * 4 * PIECES + q) * 1024) % 65536), (uint32_t)(lane * 16), srd,
                so + (uint32_t)((wave * 4 * PIECES + q) * 1024));
      }
    } else {
#pragma unroll
      for (int q = 0; q < PIECES; ++q)
        dma16(lds0 + (uint32_t)((s & 1) * 65536 + ((wave * PIECES + q) * 1024) % 65536), (uint32_t)((wave * PIECES + q) * 1024 + lane * 16), srd, so);
    }
  };
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int x = lane & 15, y = lane >> 4;
  const uint32_t lo0 = (uint32_t)(x * 128 + 16 * (y ^ (x >> 1)));
  auto work = [&](int s) __attribute__((always_inline)) {
    if constexpr (WORK & 1) {
      const char* st = smem + ((s & 1) ^ 1) * 65536;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(st + 32768 + (wave >> 1) * 8192 + j * 2048 + (lo0 ^ (ks * 64)));
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const uint4*>(st + (wave & 1) * 16384 + i * 2048 + (lo0 ^ (ks * 64)));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (WORK & 2)
              acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i * 4 + j], 0, 0, 0);
            else
              asm volatile("" ::"v"(fb[j].x), "v"(fa[i].x));
          }
      }
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (DEPTH == 2) issue(0);
  for (int s = 0; s < steps; ++s) {
    if (DEPTH == 2) {
      if (s + 1 < steps) issue(s + 1);
      work(s);
      if constexpr (WORK & 28) {
        // (the issuing waves wait for their step-s pieces: everything but the pieces of step s + 1 just issued)
        constexpr int NP = (WORK & 4) ? 8 * PIECES : (WORK & 8) ? 4 * PIECES : 2 * PIECES;
        if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP > 63 ? 63 : NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      issue(s);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (WORK & 2) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sink && t == 12345.f) sink[tid] = t;
  }
}
template <int DEPTH, int PIECES, int WORK = 0>
static void run(const char* src, size_t bytes, int blocks, uint32_t span, int shared_src, int steps, unsigned long long* dcyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<DEPTH, PIECES, WORK>), dim3(blocks), dim3(512), 0, 0, src, span, shared_src, steps, dcyc, (float*)nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<DEPTH, PIECES, WORK>), dim3(blocks), dim3(512), 0, 0, src, span, shared_src, steps, dcyc, (float*)nullptr);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double cy = (double)h[blocks / 2] / steps;
  const double bytes_step = 8.0 * PIECES * 1024;
  printf("work %d blocks %3d depth %d pieces/wave %d span %8u %s : %7.0f cycles/step  %5.1f B/clk/CU  wall %.1f us  %.2f TB/s  clock %.2f GHz\n", WORK, blocks, DEPTH,
         PIECES, span, shared_src ? "shared " : "private", cy, bytes_step / cy, ms * 1e3, blocks * bytes_step * steps / (ms * 1e-3) / 1e12,
         (double)h[blocks / 2] / (ms * 1e-3) / 1e9);
  fflush(stdout);
}
int main() {
  const size_t bytes = (size_t)256 * (8u << 20);
  char* src;
  hipMalloc(&src, bytes);
  hipMemset(src, 1, bytes);
  unsigned long long* dcyc;
  hipMalloc(&dcyc, 4096);
  const int steps = 256;
  for (int blocks : {1, 8, 32, 128, 256}) {
    for (int shared_src : {1, 0}) {
      const uint32_t span = shared_src ? (2u << 20) : (8u << 20);
      run<1, 8>(src, bytes, blocks, span, shared_src, steps, dcyc);
      run<2, 8>(src, bytes, blocks, span, shared_src, steps, dcyc);
      run<2, 4>(src, bytes, blocks, span, shared_src, steps, dcyc);
    }
  }
  // the same stream beside the fragment reads (and the MFMAs) of a 256 x 256 x 64 step, L2-resident source
  for (int blocks : {1, 256}) {
    run<2, 8, 1>(src, bytes, blocks, 2u << 20, 1, steps, dcyc);
    run<2, 8, 3>(src, bytes, blocks, 2u << 20, 1, steps, dcyc);
  }
  // one / two loader waves
  for (int blocks : {1, 256}) {
    run<2, 8, 4>(src, bytes, blocks, 2u << 20, 1, steps, dcyc);
    run<2, 8, 7>(src, bytes, blocks, 2u << 20, 1, steps, dcyc);
    run<2, 8, 11>(src, bytes, blocks, 2u << 20, 1, steps, dcyc);
    run<2, 8, 19>(src, bytes, blocks, 2u << 20, 1, steps, dcyc);
    run<2, 8, 16>(src, bytes, blocks, 2u << 20, 1, steps, dcyc);
  }
  // ... and with a quarter of the bytes from beyond L2 (A slabs of a GEMM: private), three quarters shared
  // private 256 KiB regions: L2 resident per workgroup once warmed (the walk wraps 4 x per launch)
  for (int blocks : {32, 256}) run<2, 8>(src, bytes, blocks, 256u << 10, 0, steps, dcyc);
  return 0;
}
