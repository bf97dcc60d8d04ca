// Kernel-side argument block of the fused feed-forward kernels (rowblock.hip: 64-row blocks; ffn_pc.hip: 128-row
// producer / consumer blocks): the public s2t_ffn_args + what only the backward flavour uses + the pair exchange.
#pragma once
#include "common.h"

// Word of the flag area that counts exchange time-outs (the flags of every split / row count stay below word 2 560): a FIXED
// place for every shape, read by the host (s2t_amd/kernels.py: ffn_exchange_check / ffn_exchange_poll).
#define S2T_PC_ERR_WORD 4095

namespace {

// kernel arguments: the forward's public struct + what only the backward flavour uses (LayerNorm backward in the epilogue)
struct FfnK : s2t_ffn_args {
  const void* lb_x;        // [M][256] bf16 input of the leading LayerNorm (NULL: the epilogue stores dxn as it is)
  const float* lb_gamma;
  const float* lb_mean;
  const float* lb_rstd;
  const void* lb_dres;     // gradient arriving on the residual branch, added to dx (may be NULL)
  float* lb_ws;            // [replicas][2][256] fp32 partial sums of dgamma | dbeta (atomics)
  int lb_replicas;
  void* lb_dx;             // [M][256] bf16
  void* lb_dx_drop;        // optional dropout(dx) under the mask (lb_drop_p, lb_drop_site)
  float lb_drop_p;
  uint32_t lb_drop_site;
  // backward of the LayerNorm BEHIND the block (final_norm) in the prologue: x is then the gradient w.r.t. that LayerNorm's
  // output, the kernel derives dres (gradient w.r.t. the block output y) and its dropped image (the products' input)
  const void* pl_y;        // [M][256] bf16 block output the LayerNorm normalised (NULL: x is used as it is)
  const float* pl_gamma;
  const float* pl_mean;
  const float* pl_rstd;
  const int32_t* pl_lens;  // padded-frame mask of that LayerNorm's output (rows t >= lens[b] carry no gradient)
  int pl_T;
  float* pl_ws;            // [replicas][2][256] partial sums of its dgamma | dbeta
  int pl_replicas;
  void* pl_dres;           // [M][256] bf16 out
  void* pl_dy;             // [M][256] bf16 out: dropout(dres) under (drop_o_p, drop_o_site); NULL without output dropout
  // ffn_pc.hip with the hidden dimension split over a PAIR of workgroups: fp32 partial rows and one flag per workgroup
  float* xws;              // [pairs][2][64][256] fp32
  uint32_t* xflags;        // [pairs][2] (or [blocks][split][split]), zero between launches; word PC_ERR_WORD: exchange time-outs
  int xfault;              // test hook (s2t_ffn_debug_fault): part 1 of every block never raises its flag, short spin limit
  int z_tiled;             // z (training forward: written, backward: read) is in the tiled layout of include/s2t_hip.h
};

}  // namespace
