// Do LDS fragment reads and MFMAs of the SAME kernel overlap on gfx950?  One workgroup of 8 waves per CU (two per SIMD), each
// wave per iteration: 32 independent v_mfma_f32_16x16x32_bf16 (the work of one K-step of a 256 x 256 tile) and / or the 12
// operand fragments of such a step from LDS (24 ds_read_b64_tr_b16, or 12 ds_read_b128), interleaved as four groups of
// (6 reads, 8 MFMAs) whose MFMAs never use the registers the reads of the same iteration write.
// hipcc --offload-arch=gfx950 -O3 lds_mfma_overlap.hip -o lds_mfma_overlap && ./lds_mfma_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr64(const char* a) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a)));
}
// MODE bit 0: MFMAs, bit 1: transposed reads, bit 2: b128 reads instead; BAR: one workgroup barrier per iteration
template <int MODE, bool BAR>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + i;
  __syncthreads();
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 fr[2][12];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 12; ++i) fr[s][i] = make_uint4(0x3c003c00u, lane, i, s);
  // conflict-free addresses: 512-byte rows, 16-byte chunk ^ row key (the wgrad256 image)
  const int x = lane & 15, y = lane >> 4, q = x >> 2, pb = x & 3;
  const int key = 2 * (q | ((y & 1) << 2));
  const uint32_t base = (uint32_t)((8 * y + q) * 512 + 16 * (pb >> 1) + 8 * (pb & 1));
  uint32_t off[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) off[i] = base + 16u * (uint32_t)((2 * i) ^ key);
  const uint32_t roff = (uint32_t)(x * 512 + 16 * (y ^ x));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const char* st = smem + (it & 1) * 32768;
    if (BAR) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // (two steps per iteration so that the register roles are static)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (MODE & 2) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const uint2 lo = tr64(st + off[3 * g + r]), hi = tr64(st + off[3 * g + r] + 2048);
            fr[half ^ 1][3 * g + r] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
        if (MODE & 4) {
#pragma unroll
          for (int r = 0; r < 3; ++r) fr[half ^ 1][3 * g + r] = *reinterpret_cast<const uint4*>(st + roff + 16 * ((3 * g + r) * 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 1) {
#pragma unroll
          for (int m = 0; m < 8; ++m)
            acc[8 * g + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[half][m & 3]),
                                                                     __builtin_bit_cast(bf16x8, fr[half][4 + (m >> 2) + 2 * (g & 1)]),
                                                                     acc[8 * g + m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(MODE & 1)) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("" ::"v"(fr[half ^ 1][i].x), "v"(fr[half ^ 1][i].w));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 12; ++i) s += (float)(fr[0][i].x ^ fr[1][i].y);
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, bool BAR>
static void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, BAR>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime counts at 100 MHz on this part: report time per step instead of cycles
  printf("%-58s %s  %.1f ns per step (2 steps per iteration)\n", name, BAR ? "barrier/iter" : "no barrier  ", (double)c * 10.0 / (iters * 2.0));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<1, false>("32 MFMAs per wave and step", out, cyc);
  run<2, false>("24 ds_read_b64_tr_b16 per wave and step", out, cyc);
  run<3, false>("both, interleaved 4 x (6 reads, 8 MFMAs)", out, cyc);
  run<4, false>("12 ds_read_b128 per wave and step", out, cyc);
  run<5, false>("both with ds_read_b128, 4 x (3 reads, 8 MFMAs)", out, cyc);
  run<3, true>("both (transposed reads)", out, cyc);
  run<1, true>("32 MFMAs", out, cyc);
  run<2, true>("24 transposed reads", out, cyc);
  return 0;
}
