// The 256 x 256 LDS-DMA path of s2t_gemm (gemm256.hip), as gemm.hip's dispatcher sees it.
#pragma once
#include "common.h"

// true when `p` (already normalised by s2t_gemm: batch / split_k >= 1, p.ws cleared when unused) runs on the large-tile kernel
bool s2t_gemm256_eligible(const s2t_gemm_args& p);
// rows per tile the large-tile path would use (256 / 128), 0 when the arguments stay on the 128 x 128 kernel
int s2t_gemm256_tile_rows(const s2t_gemm_args& p);
// vec: every tensor of the epilogue is 16-byte aligned with N % 8 == 0 (gemm.hip, epilogue_vectorisable)
int s2t_gemm256_launch(const s2t_gemm_args& p, bool vec, hipStream_t s);
// the kernel symbol s2t_gemm256_launch starts for these arguments, as a profiler prints it
int s2t_gemm256_describe(const s2t_gemm_args& p, bool vec, char* buf, int buflen);
