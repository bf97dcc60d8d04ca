// Cross-lane steps of csrc/common.h (DPP, v_permlane16_swap, v_permlane32_swap) against __shfl_xor on the GPU: every lane,
// every step, bit for bit, and the 16 / 32 / 64-lane butterflies against the loops they replace.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude tools/ubench/red_dpp.hip -o tools/ubench/red_dpp && tools/ubench/red_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../s2t_amd/csrc/common.h"
__global__ void k(const float* in, float* o) {
  const float v = in[threadIdx.x];
  const int t = threadIdx.x;
  o[0 * 64 + t] = s2t_xadd<1>(v);   o[1 * 64 + t] = v + __shfl_xor(v, 1, 64);
  o[2 * 64 + t] = s2t_xadd<2>(v);   o[3 * 64 + t] = v + __shfl_xor(v, 2, 64);
  o[4 * 64 + t] = s2t_xadd<4>(v);   o[5 * 64 + t] = v + __shfl_xor(v, 4, 64);
  o[6 * 64 + t] = s2t_xadd<8>(v);   o[7 * 64 + t] = v + __shfl_xor(v, 8, 64);
  o[8 * 64 + t] = s2t_xadd<16>(v);  o[9 * 64 + t] = v + __shfl_xor(v, 16, 64);
  o[10 * 64 + t] = s2t_xadd<32>(v); o[11 * 64 + t] = v + __shfl_xor(v, 32, 64);
  o[12 * 64 + t] = s2t_xmax<4>(v);  o[13 * 64 + t] = fmaxf(v, __shfl_xor(v, 4, 64));
  o[14 * 64 + t] = s2t_xmax<16>(v); o[15 * 64 + t] = fmaxf(v, __shfl_xor(v, 16, 64));
  float a = v, b = v, c = v, m = v;
  for (int s = 8; s > 0; s >>= 1) a += __shfl_xor(a, s, 64);
  for (int s = 16; s > 0; s >>= 1) b += __shfl_xor(b, s, 64);
  for (int s = 32; s > 0; s >>= 1) c += __shfl_xor(c, s, 64);
  for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
  o[16 * 64 + t] = s2t_sum16(v); o[17 * 64 + t] = a;
  o[18 * 64 + t] = s2t_sum32(v); o[19 * 64 + t] = b;
  o[20 * 64 + t] = s2t_sum64(v); o[21 * 64 + t] = c;
  o[22 * 64 + t] = s2t_max64(v); o[23 * 64 + t] = m;
}
int main() {
  float h[64], *d, *o;
  unsigned x = 12345u;
  for (int i = 0; i < 64; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(int)(x >> 8) * 1.1920929e-7f * 3.7f - 17.3f; }
  hipMalloc(&d, 256); hipMalloc(&o, 24 * 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  static float r[24 * 64];
  if (hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED (hip)\n"); return 2; }
  int bad = 0;
  for (int p = 0; p < 12; ++p)
    for (int i = 0; i < 64; ++i)
      if (memcmp(&r[(2 * p) * 64 + i], &r[(2 * p + 1) * 64 + i], 4)) {
        if (bad < 20) printf("pair %d lane %d: %.9g vs %.9g\n", p, i, r[(2 * p) * 64 + i], r[(2 * p + 1) * 64 + i]);
        ++bad;
      }
  printf(bad ? "FAILED (%d)\n" : "ok\n", bad);
  return bad != 0;
}
