#include "common.h"

extern "C" int s2t_version(void) { return 1; }

extern "C" int s2t_device_cu_count(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  return prop.multiProcessorCount;
}

// ---- s2t_occupy_cus: hold `n` compute units for a bounded time (a rehearsal of kernels that share the GPU with this path's: the
// RCCL all-reduce beside backward, legacy_distributed_data_parallel.py:76-160).  Each workgroup pins 96 KiB of LDS — none of
// the row-block / fused feed-forward / grouped weight-gradient workgroups (128 - 160 KiB) fits beside it, so the CU is lost to
// them — and sleeps until *stop becomes non-zero or `ms` milliseconds of the constant 100 MHz clock have passed (an exit every
// wave reaches: the grid always drains).
namespace {
__global__ __launch_bounds__(256) void occupy_kernel(const uint32_t* stop, unsigned long long ticks, uint32_t* arrived) {
  __shared__ char pin[96 * 1024];
  if (threadIdx.x == 0) {
    pin[0] = 1;
    if (arrived) atomicAdd(arrived, 1u);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  typedef __attribute__((address_space(1))) uint32_t gu32;
  while (true) {
    if (stop && __hip_atomic_load((gu32*)stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(64);
  }
  if (threadIdx.x == 300) pin[1] = pin[0];  // (keeps the array)
}
}  // namespace

extern "C" int s2t_occupy_cus(int n, int ms, const uint32_t* stop, uint32_t* arrived, void* stream) {
  if (n <= 0 || n > 1024 || ms <= 0 || ms > 2000) return S2T_ERR_ARG;
  hipLaunchKernelGGL(occupy_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, stop, (unsigned long long)ms * 100000ull, arrived);
  return S2T_LAUNCH_CHECK();
}
