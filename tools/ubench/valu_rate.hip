// VALU issue cost of integer multiply flavours on gfx950 (cycles per wave-instruction, one wave per SIMD): the inputs of the
// dropout hash (common.h).  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int OP>
__global__ void k(uint32_t* out, uint32_t seed, int iters, unsigned long long* cyc) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 12345u, d = b + 777u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) { a *= 0x7feb352dU; b *= 0x846ca68bU; c *= 0x7feb352dU; d *= 0x846ca68bU; }
      if (OP == 1) {
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(0x6ca68bu));
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b) : "v"(0x6ca68bu));
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(0x6ca68bu));
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(d) : "v"(0x6ca68bu));
      }
      if (OP == 2) {
        unsigned long long p = (unsigned long long)a * 0x846ca68bU; a = (uint32_t)p ^ (uint32_t)(p >> 32);
        p = (unsigned long long)b * 0x846ca68bU; b = (uint32_t)p ^ (uint32_t)(p >> 32);
        p = (unsigned long long)c * 0x846ca68bU; c = (uint32_t)p ^ (uint32_t)(p >> 32);
        p = (unsigned long long)d * 0x846ca68bU; d = (uint32_t)p ^ (uint32_t)(p >> 32);
      }
      if (OP == 3) { a ^= a >> 15; b ^= b >> 15; c ^= c >> 15; d ^= d >> 15; }
      if (OP == 4) {
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(0x846ca68bu));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(b) : "v"(0x846ca68bu));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(0x846ca68bu));
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d) : "v"(0x846ca68bu));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  uint32_t* out; unsigned long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const char* names[] = {"v_mul_lo_u32", "v_mul_u32_u24", "64-bit product (mul_lo + mul_hi) + xor", "xorshift (2 ops)", "v_mul_hi_u32"};
  const int per[] = {32, 32, 32, 32, 32};
  for (int w = 1; w <= 2; ++w)
    for (int op = 0; op < 5; ++op) {
      const int iters = 1000;
      for (int rep = 0; rep < 2; ++rep) {
        if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * w), 0, 0, out, 1u, iters, cyc);
        if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * w), 0, 0, out, 1u, iters, cyc);
        if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * w), 0, 0, out, 1u, iters, cyc);
        if (op == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256 * w), 0, 0, out, 1u, iters, cyc);
        if (op == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256 * w), 0, 0, out, 1u, iters, cyc);
      }
      hipDeviceSynchronize();
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%d wave(s)/SIMD  %-42s %.1f cycles per group-of-4 element-steps (%d per iteration)\n", w, names[op], (double)c / (iters * 8.0), per[op]);
    }
  return 0;
}
