// attn_common.h — shared pieces of the attention kernels (forward, dQ, dK/dV).
//
// MFMA shape: v_mfma_f32_32x32x16_bf16.  Fragment maps (lane l, h = l>>5):
//   A[i][k] : i = l&31, k = 8h + j (j = 0..7)        B[k][n] : n = l&31, k = 8h + j
//   C[r][n] : n = l&31, r = 8a + 4h + b  for register index 4a + b (a,b = 0..3)
//
// "Row permutation" trick: when a 32-row operand tile is the MFMA A operand and its rows become the
// contraction index of the NEXT MFMA, rows are fetched in the order  pi(i) = i with bits 2 and 3 swapped.
// Then a lane's accumulator registers 8m..8m+7 (m = 0,1) hold rows 16m + 8h + {0..7} — exactly the 8
// consecutive k-slots the next MFMA's B operand wants from that lane: no cross-lane traffic, no LDS bounce.
#pragma once
#include "common.h"

#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int perm23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// actual row (0..31) held by accumulator register reg of a lane with half h, after the perm23 fetch order
__device__ __forceinline__ int acc_row(int reg, int h) { return 16 * (reg >> 3) + 8 * h + (reg & 7); }

// swizzled byte offset inside a row-major LDS tile whose rows are ROWB bytes (128, 192 or 256)
template <int ROWB>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  if (ROWB == 256) return row * 256 + ((chunk ^ (row & 15)) << 4);
  if (ROWB == 192)               // head_dim 96: 12 chunks per row; rows r, r+4, r+8, r+12 share a bank base at a 192-byte pitch -> XOR the low two chunk
    return row * 192 + ((chunk ^ ((row >> 2) & 3)) << 4);   // bits with (row>>2)&3 (stays inside the aligned group of 4 chunks: a bijection on 0..11)
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = f2bf(v[j]);
  return o;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// exchange between the two 32-lane halves of a wave (v_permlane32_swap: VALU, no LDS round trip): every lane gets max / sum of lanes l and l ^ 32
__device__ __forceinline__ float xhalf_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// XCD-aware workgroup -> (tile, head, batch) mapping for the attention kernels.  Grids stay (tiles, H, B); the hardware hands consecutive linear
// workgroup ids (x fastest) to the 8 XCDs round-robin, so with the identity mapping the tiles of ONE head are spread over all 8 XCDs and every
// XCD's private L2 streams the K/V (or Q/dO) tiles of every head — 8 L2 fills per tile, and each XCD's L2 holds 14 heads' working windows at once.
// Here linear id L goes to head*batch index (L % 8) + 8 * ((L / 8) / tiles): all tiles of a head run on one XCD, each K/V tile is filled into one L2
// once and then hit by the other workgroups of that head.  Needs H*B % 8 == 0 (Flux 24 x B, SD3 24 x B, PixArt 16 x B, SDXL 10/20 x even B);
// otherwise the identity mapping is kept.  ST355_ATTN_NO_XCD (lab builds) compiles the identity mapping for A/B runs.
struct WgMap { int tile, head, b; };
__device__ __forceinline__ WgMap attn_wg_map() {
  const int gx = gridDim.x, H = gridDim.y, HB = H * gridDim.z;
  int tile = blockIdx.x, hb = blockIdx.y + H * blockIdx.z;
#ifndef ST355_ATTN_NO_XCD
  if ((HB & 7) == 0) {
    const int L = tile + gx * hb, slot = L >> 3, grp = slot / gx;
    hb = (L & 7) + 8 * grp;
    tile = slot - grp * gx;
  }
#endif
  WgMap m; m.tile = tile; m.b = hb / H; m.head = hb - m.b * H;
  return m;
}
