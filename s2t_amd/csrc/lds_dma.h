// LDS-DMA helpers (gfx950): buffer_load ... lds moves 64 lanes x 16 bytes from global memory straight into LDS — no
// staging registers, no ds_write.  The compiler neither counts nor waits for these loads: every wait is a hand-placed
// counted s_waitcnt vmcnt(N) (loads, stores and DMA count together, in issue order).
#pragma once
#include "common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// global (srd base + voff + soff) -> LDS (lds_base + lane*16)
__device__ __forceinline__ void dma16(uint32_t lds_base, uint32_t voff, i32x4 srd, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// the same with the non-temporal policy: for bytes this launch reads ONCE (they then leave the lines of re-read operands in L2)
__device__ __forceinline__ void dma16_nt(uint32_t lds_base, uint32_t voff, i32x4 srd, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
               :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// the same with an instruction offset (0..4095) that moves BOTH the global source and the LDS destination
template <int OFF>
__device__ __forceinline__ void dma16_off(uint32_t lds_base, uint32_t voff, i32x4 srd, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:%4 lds"
               :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff), "i"(OFF) : "memory");
}

// raw buffer descriptor: offsets >= bytes read as zero
__device__ __forceinline__ i32x4 make_srd(const void* base, uint32_t bytes) {
  const uint64_t b = (uint64_t)base;
  i32x4 s;
  s.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  s.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
  s.z = __builtin_amdgcn_readfirstlane((int)bytes);
  s.w = __builtin_amdgcn_readfirstlane(0x00020000);
  return s;
}

}  // namespace
