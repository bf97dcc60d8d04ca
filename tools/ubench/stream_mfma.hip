// What bounds a chunk loop that streams weight panels L2 -> LDS (buffer_load ... lds) beside the MFMAs that consume them?  §4a of
// three differently cut production loops settle at 18 - 20 B/clk/CU where the bare stream reaches 58.  This strips the
// loop to its skeleton at the fused FFN's ratio — per step 64 KiB of DMA (64 pieces of 1 KiB), per wave 24 ds_read_b128 of the
// OTHER stage and 64 v_mfma_f32_16x16x32_bf16 (1024 clocks: 2048 per SIMD with two waves) — and varies only the SCHEDULE:
//   MODE 0  burst    every wave issues its 8 pieces, then reads + multiplies, counted wait, barrier       (dma_stream.hip "work 3")
//   MODE 1  spread   one piece behind every 8 MFMAs
//   MODE 2  opposed  waves 0-3: pieces first, then MFMAs; waves 4-7 (the SIMD partners): MFMAs first, then pieces
//   MODE 3  halves   the step is cut into two half-steps of 32 KiB with a barrier each (four 32 KiB stages, DMA two half-steps ahead)
//   MODE 4  opposed + spread: waves 0-3 issue in the first half of their MFMAs, waves 4-7 in the second half
//   hipcc --offload-arch=gfx950 -O3 stream_mfma.hip -o stream_mfma && ./stream_mfma
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
__device__ __forceinline__ void dma16(uint32_t lds_base, uint32_t voff, i32x4 srd, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const char* src, uint32_t span, int steps, unsigned long long* cyc, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[131072];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  i32x4 srd;
  srd.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)src);
  srd.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)src >> 32));
  srd.z = (int)span;
  srd.w = 0x00020000;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int x = lane & 15, y = lane >> 4;
  const uint32_t lo0 = (uint32_t)(x * 128 + 16 * (y ^ (x >> 1)));
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // piece q (0..7) of this wave for step s: 1 KiB at stage (s & 1), offset (8 wave + q) KiB
  auto piece = [&](int s, int q) __attribute__((always_inline)) {
    const uint32_t so = (uint32_t)(((uint64_t)s * 65536u) % (span - 65536u + 1)) & ~1023u;
    dma16(lds0 + (uint32_t)((s & 1) * 65536 + (wave * 8 + q) * 1024), (uint32_t)((wave * 8 + q) * 1024 + lane * 16), srd, so);
  };
  // half ks (0, 1) of the step's reads + MFMAs from stage st: 12 ds_read_b128, 32 MFMAs; `after8(g)` runs behind every 8 MFMAs
  auto half = [&](const char* st, int ks, auto after8) __attribute__((always_inline)) {
    uint4 fa[8], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(st + 32768 + (wave >> 1) * 8192 + j * 2048 + (lo0 ^ (ks * 64)));
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const uint4*>(st + (wave & 1) * 16384 + i * 2048 + (lo0 ^ (ks * 64)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i * 4 + j], 0, 0, 0);
      if ((i & 1) == 1) after8(ks * 4 + (i >> 1));
    }
  };
  auto none = [](int) __attribute__((always_inline)) {};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (MODE == 6 || MODE == 7) {
    // register-staged stream: 8 buffer_load_dwordx4 per wave and step into 32 registers at the head of the step (issue only), the
    // ds_write_b128 into the other stage behind the MFMAs (MODE 6: all eight at the end; MODE 7: four behind each half)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)span, 0x00020000);
    u32x4 stg[8];
    auto loads = [&](int s) __attribute__((always_inline)) {
      const uint32_t so = (uint32_t)(((uint64_t)s * 65536u) % (span - 65536u + 1)) & ~1023u;
#pragma unroll
      for (int q = 0; q < 8; ++q) stg[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)((wave * 8 + q) * 1024 + lane * 16), so, 0);
    };
    auto writes = [&](int s, int q0, int q1) __attribute__((always_inline)) {
#pragma unroll
      for (int q = q0; q < q1; ++q) *reinterpret_cast<u32x4*>(smem + (s & 1) * 65536 + (wave * 8 + q) * 1024 + lane * 16) = stg[q];
    };
    loads(0);
    writes(0, 0, 8);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      const char* st = smem + (s & 1) * 65536;
      const bool more = s + 1 < steps;
      if (more) loads(s + 1);
      half(st, 0, none);
      if (MODE == 7 && more) writes(s + 1, 0, 4);
      half(st, 1, none);
      if (more) writes(s + 1, MODE == 7 ? 4 : 0, 8);
      __syncthreads();
    }
  } else if constexpr (MODE == 3) {
    // half-steps h = 2 s + ks: 32 KiB each (4 pieces per wave) into stage (h & 3) of four 32 KiB stages; the work of half-step h reads
    // the image of half-step h (same bytes as the whole-step layout: stage pair (s & 1), half ks)
    auto hpiece = [&](int h, int q) __attribute__((always_inline)) {
      const int s = h >> 1, ks = h & 1;
      const uint32_t so = (uint32_t)(((uint64_t)s * 65536u) % (span - 65536u + 1)) & ~1023u;
      const int pq = (wave * 8 + ks * 4 + q);
      dma16(lds0 + (uint32_t)((s & 1) * 65536 + pq * 1024), (uint32_t)(pq * 1024 + lane * 16), srd, so);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) hpiece(0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) hpiece(1, q);
    const int H = 2 * steps;
    for (int h = 0; h < H; ++h) {
      // stage of half-step h + 2 = stage of half-step h - 2: read two half-steps ago, free since the last barrier
      if (h + 2 < H) {
#pragma unroll
        for (int q = 0; q < 4; ++q) hpiece(h + 2, q);
      }
      // (reads the stage that is NOT being written by the pieces just issued: half-step h's own image landed a half-step ago)
      half(smem + (((h >> 1) & 1) ^ 1) * 65536, h & 1, none);
      if (h + 2 < H) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) piece(0, q);
    for (int s = 0; s < steps; ++s) {
      const char* st = smem + ((s & 1) ^ 1) * 65536;
      const bool more = s + 1 < steps;
      if constexpr (MODE == 0) {
        if (more) {
#pragma unroll
          for (int q = 0; q < 8; ++q) piece(s + 1, q);
        }
        half(st, 0, none);
        half(st, 1, none);
      } else if constexpr (MODE == 1) {
        auto one = [&](int g) __attribute__((always_inline)) { if (more) piece(s + 1, g); };
        half(st, 0, one);
        half(st, 1, one);
      } else if constexpr (MODE == 2) {
        // ONE code path (a scalar predicate per burst: duplicated bodies made hipcc spill the 128 accumulator registers)
        if (more && wave < 4) {
#pragma unroll
          for (int q = 0; q < 8; ++q) piece(s + 1, q);
        }
        half(st, 0, none);
        if (more && wave >= 4) {
#pragma unroll
          for (int q = 0; q < 8; ++q) piece(s + 1, q);
        }
        half(st, 1, none);
      } else if constexpr (MODE == 4) {
        auto first = [&](int g) __attribute__((always_inline)) {
          if (more && wave < 4) {
            piece(s + 1, 2 * (g & 3));
            piece(s + 1, 2 * (g & 3) + 1);
          }
        };
        auto second = [&](int g) __attribute__((always_inline)) {
          if (more && wave >= 4) {
            piece(s + 1, 2 * (g & 3));
            piece(s + 1, 2 * (g & 3) + 1);
          }
        };
        half(st, 0, first);
        half(st, 1, second);
      } else if constexpr (MODE == 8) {   // no stream at all: fragment reads + MFMAs + barrier (the floor of this wave structure)
        half(st, 0, none);
        half(st, 1, none);
      } else {  // MODE 5: every wave issues its pieces in the MIDDLE of the step (between the two halves of its MFMAs)
        half(st, 0, none);
        if (more) {
#pragma unroll
          for (int q = 0; q < 8; ++q) piece(s + 1, q);
        }
        half(st, 1, none);
      }
      if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (the pieces of step s + 1 stay in flight; those of step s landed)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sink && t == 12345.f) sink[tid] = t;
}
template <int MODE>
static void run(const char* name, const char* src, int blocks, uint32_t span, int steps, unsigned long long* dcyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, src, span, steps, dcyc, (float*)nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, src, span, steps, dcyc, (float*)nullptr);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double cy = (double)h[blocks / 2] / steps;
  printf("%-16s blocks %3d span %7u: %6.0f clocks per 64 KiB step  %5.1f B/clk/CU  MFMA duty %4.1f %%  wall %.1f us  clock %.2f GHz\n", name, blocks, span, cy,
         65536.0 / cy, 100.0 * 2048.0 / cy, ms * 1e3, (double)h[blocks / 2] / (ms * 1e-3) / 1e9);
  fflush(stdout);
}
int main() {
  const size_t bytes = (size_t)8u << 20;
  char* src;
  hipMalloc(&src, bytes);
  hipMemset(src, 1, bytes);
  unsigned long long* dcyc;
  hipMalloc(&dcyc, 4096);
  const int steps = 128;
  for (int blocks : {1, 256}) {
    for (uint32_t span : {2u << 20}) {
      run<0>("burst", src, blocks, span, steps, dcyc);
      run<1>("spread", src, blocks, span, steps, dcyc);
      run<2>("opposed", src, blocks, span, steps, dcyc);
      run<3>("half-steps", src, blocks, span, steps, dcyc);
      run<4>("opposed+spread", src, blocks, span, steps, dcyc);
      run<5>("burst mid-step", src, blocks, span, steps, dcyc);
      run<8>("no stream", src, blocks, span, steps, dcyc);
      run<6>("reg-staged end", src, blocks, span, steps, dcyc);
      run<7>("reg-staged split", src, blocks, span, steps, dcyc);
    }
  }
  return 0;
}
