// gemm.hip — bf16 MFMA GEMM for every dense contraction of the DiT step (K4, K8, K9, K11, K12):
//     C[M,N] = A[M,K] · B[N,K]^T  (+ A2[M,K2] · B2[N,K2]^T)   with fused epilogues.
//
// Common gfx950 design (cdna_hip_programming.md §5):
//   * v_mfma_f32_32x32x16_bf16; each wave owns a 64x64 output sub-tile (2x2 MFMA tiles, 64 fp32 accumulators / lane).
//   * operands are SWAPPED inside the MFMA (weights = MFMA "A", activations = MFMA "B") so that a lane's 4
//     consecutive accumulator registers are 4 consecutive FEATURES of one token: row-major C gets 8-byte
//     packed stores and the epilogue reads bias / gate / residual with the same contiguity.
//   * global -> LDS by direct LDS-DMA (global_load_lds_dwordx4, 16 B / lane); the LDS image stays
//     lane-linear and the bank-conflict swizzle  chunk' = chunk ^ ((row>>1)&7)  is applied on the SOURCE
//     address and again on the ds_read_b128 address (rule 21: both-sides-or-neither).
//   * the LoRA low-rank term is a K-extension of the same accumulators (K2 extra columns), i.e. the fused
//     base+low-rank GEMM of the north star: y = [x | xA^T] · [W | sB]^T.
//   * XCD-aware bijective block remap + grouped tile order for L2 locality.
//
// Two schedules:
//   k_gemm_p3   256(tokens) x 128(features) x 64 tile, 8 waves (2 per SIMD), THREE-stage LDS ring (3 x 48 KiB), LDS-DMA
//               issued two K-tiles ahead and retired with a COUNTED s_waitcnt vmcnt(6) + raw s_barrier, so one tile's
//               loads always stay in flight across the barrier (T3/T4: "never drain vmcnt to 0 in the main loop").
//               Takes up to 2 independent problems per launch (img + txt stream of an MMDiT block share one grid, so the
//               512-row txt GEMM fills the tail of the 4096-row img GEMM instead of running at 19% occupancy).
//   k_gemm_s2   128 x 128 x 64 tile, 4 waves, double-buffered, one drained barrier per K-tile; used for small problems.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

#define BK 64
#define EPI_SPLITK 100        // internal: k_gemm_s2 writes fp32 K-slice slabs instead of C

// Optional in-kernel timeline (tools/gemm_lab.hip builds this file with -DST355_TRACE): waves 0 and 4 of one workgroup stamp
// s_memtime at the section boundaries of two k-steps; the stamps stay in SGPRs until the kernel ends.
#ifdef ST355_TRACE
#define TR_T0 40
#define TR_MARKS 12
__device__ uint64_t st355_trace_buf[2][TR_MARKS];
#define TR_DECL uint64_t tr_[TR_MARKS] = {}
// the stamp stays an un-consumed SMEM result (no s_waitcnt is forced at the mark itself)
#define TR(t, k)                                                  \
  do {                                                            \
    if ((t) == TR_T0) tr_[k] = __builtin_amdgcn_s_memtime();       \
  } while (0)
#define TR_FLUSH(wv, lane)                                                              \
  do {                                                                                  \
    if (blockIdx.x == 64 && (lane) == 0 && ((wv) & 3) == 0)                              \
      for (int k_ = 0; k_ < TR_MARKS; k_++) st355_trace_buf[(wv) >> 2][k_] = tr_[k_];    \
  } while (0)
#else
#define TR_DECL
#define TR(t, k)
#define TR_FLUSH(wv, lane)
#endif

struct GemmP {
  const bf16* A; int64_t lda;
  const bf16* B; int64_t ldb;
  const bf16* A2; int64_t lda2;
  const bf16* B2; int64_t ldb2;
  bf16* C; int64_t ldc;
  int M, N, K, K2;
  const bf16* bias;
  bf16* aux_out; int64_t ld_aux_out;
  const bf16* aux_in; int64_t ld_aux_in;
  const bf16* gate; int64_t gate_stride; int64_t rows_per_batch;
  float* partial; int ksplit;          // split-K: fp32 slabs [ksplit][M][part_ld]
  int64_t part_ld;                     // slab row stride (N, or 9*N when the nine conv-wgrad taps share one launch)
  const float* scale_a; const float* scale_b;   // fp8 Linear: out = acc * scale_a[0] * scale_b[n] (+ bias); NULL otherwise
  // convolution-as-GEMM over a zero-bordered NHWC grid (conv.hip / st355_conv_bf16): rows = grid positions, K = taps * Cin.
  // K-tile u reads the A rows shifted by (ty*Wp + tx) positions, tap = u / conv_tpt = 3*ty + tx (taps == 9; no shift when taps == 1);
  // the epilogue adds img_add[image, n] and writes ZERO on border positions.  conv_taps == 0: plain GEMM.
  int conv_taps, conv_tpt, conv_wp, conv_hp;
  int skip_dead = 1;            // conv mode: waves whose 64 output columns lie beyond N issue no MFMAs (ST355_CONV_SKIP_DEAD=0: A/B)
  int64_t conv_row0;                            // grid position of GEMM row 0 (border test / image index)
  const bf16* img_add; int64_t img_add_stride;
  // segmented rows (st355_gemm_args.seg_rows): logical row m of the problem lives at physical row (m / seg_rows) * stride_X + m % seg_rows of operand X.
  // seg_xX = (stride_X - seg_rows) * ld_X, the extra ELEMENTS to skip per segment: a tile (whose rows never straddle a segment: seg_rows is a multiple
  // of 256) just shifts its operand base pointers by (m0 / seg_rows) * seg_xX (seg_view below) — the K loop and the epilogues are untouched.
  int seg_rows = 0;
  int64_t seg_xa = 0, seg_xa2 = 0, seg_xc = 0, seg_xin = 0, seg_xout = 0;
  // TN form with a SEGMENTED contraction axis (st355_gemm_tn_seg_bf16): K-tile u of operand X starts tn_skip_X extra BYTES further for every whole segment of
  // tn_tps K-tiles before it (tn_magic = ceil(2^32 / tn_tps): u / tn_tps = umulhi(u, tn_magic), exact for u < 2^16).  0 / 0 / 0: plain.
  uint32_t tn_magic = 0; int tn_skip_a = 0, tn_skip_b = 0;
  // ST355_EPI_QK_NORM_ROPE (st355_gemm_args.rope): the fused QKV projection of an MMDiT attention (flux/transformer.py:140-207).  Output columns
  // [0, D) are q heads, [D, 2D) k heads, [2D, 3D) v heads (D = rH * 128).  q / k tiles: per-head RMSNorm (weights rwq / rwk, NULL = none) and RoPE from
  // the fp32 accumulators, written head-major to rq / rk [B, rH, rS, 128] at joint position rpos0 + m % rows_per_batch of sample m / rows_per_batch,
  // plus 1/rms to rrms [B*rS, 2*rH] for the backward; v tiles: plain rows of C (column n - 2D), through the segment view.
  bf16* rq = nullptr; bf16* rk = nullptr; float* rrms = nullptr; bf16* rvt = nullptr; int rSp = 0;     // rvt: optional head-major V^T [B, rH, 128, rSp]
  const bf16* rwq = nullptr; const bf16* rwk = nullptr;
  const float* rcos = nullptr; const float* rsin = nullptr;
  int rH = 0, rS = 0, rpos0 = 0;
  float reps = 0.f;
  // ST355_EPI_HEADS (st355_heads; reuses rq / rk / rvt / rH / rS / rpos0 / rSp and rows_per_batch): output columns [0, hq) are q heads, [hq, hq + hk) k heads, the
  // rest v heads, 64 columns per head
  int hq = 0, hk = 0;
};

// the tile at row m0 of a segmented problem sees plain operands whose base pointers are shifted to its segment
__device__ __forceinline__ GemmP seg_view(const GemmP& p0, int m0) {
  GemmP p = p0;
  if (p0.seg_rows) {
    const int64_t s = __builtin_amdgcn_readfirstlane(m0 / p0.seg_rows);    // (the division runs on the VALU: keep the uniform quotient, and the pointers, scalar)
    p.A = p0.A + s * p0.seg_xa;
    p.A2 = p0.A2 ? p0.A2 + s * p0.seg_xa2 : nullptr;
    p.C = p0.C + s * p0.seg_xc;
    p.aux_in = p0.aux_in ? p0.aux_in + s * p0.seg_xin : nullptr;
    p.aux_out = p0.aux_out ? p0.aux_out + s * p0.seg_xout : nullptr;
  }
  return p;
}

// grid position -> (is border, image index)
__device__ __forceinline__ bool conv_border(const GemmP& p, int m, int& img) {
  const int64_t pos = p.conv_row0 + m;
  const int per = p.conv_hp * p.conv_wp;
  img = (int)(pos / per);
  const int q = (int)(pos - (int64_t)img * per);
  const int y = q / p.conv_wp, x = q - y * p.conv_wp;
  return y == 0 || y == p.conv_hp - 1 || x == 0 || x == p.conv_wp - 1;
}
// element offset (A operand) of K-tile u in conv mode
__device__ __forceinline__ int64_t conv_koff(const GemmP& p, int u, int64_t lda, int bk) {
  const int tap = u / p.conv_tpt, c = u - tap * p.conv_tpt;
  const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;               // tap / 3 for tap < 9
  const int shift = p.conv_taps == 9 ? ty * p.conv_wp + tx : 0;
  return (int64_t)shift * lda + (int64_t)c * bk;
}

struct GemmGroup {
  GemmP p[2];
  int tiles0;  // tiles of problem 0 (blocks with id >= tiles0 work on problem 1)
  // stream-K tail (k_gemm_pq<..., SK = true>; run_one): blocks [0, sk_first) run whole tiles, the tiles [sk_first, tiles0) of the last, mostly empty round are
  // cut into sk_s K-slices each (one workgroup per slice)
  int sk_first = 0x7fffffff, sk_s = 1;
  float* sk_ws = nullptr;     // fp32 slabs of the slices 0 .. sk_s - 2 of every cut tile: [tile][slice][256 x 256]
  int* sk_flags = nullptr;    // one arrival counter per cut tile; zero before and after the launch
};

__device__ __forceinline__ void glds16(const bf16* gsrc, char* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

// grouped tile order: GROUP consecutive m-tiles share one W panel
#ifndef ST355_TILE_GROUP          // tools/gemm_lab builds: the tile-order experiment (tools/gpu_lease.sh gemm_power); the library always builds with 8
#define ST355_TILE_GROUP 8
#endif
// measured r5 (profiles/r05_gemm_power.md): 8 sits at the fetch minimum for K <= 3072 (8 x 4 concurrent tiles per XCD: 12 operand panels per 32 tiles); with
// K >= 8192 panels (4-6 MB, above the XCD's L2) 4 is 2.8 % faster (1325 vs 1290 TFLOP/s at 36864 x 3072 x 12288) at equal fetch
__device__ __forceinline__ int tile_group_for(int K) { return (ST355_TILE_GROUP == 8 && K >= 8192) ? 4 : ST355_TILE_GROUP; }
__device__ __forceinline__ void tile_coords(int id, int nbm, int nbn, int& pm, int& pn, const int GROUP = ST355_TILE_GROUP) {
  const int width = GROUP * nbn;
  const int group_id = id / width;
  const int first_m = group_id * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  pm = first_m + (id % width) % gsz;
  pn = (id % width) / gsz;
}

// ---- shared epilogue: acc[i][j] is D[n_local][m_local] of the wave's 64x64 sub-tile at (mw0, nw0) ----
// lane: token m = mw0 + j*32 + (lane&31); features nw0 + i*32 + 8a + 4*(lane>>5) + b for register 4a+b
template <int EPI, int NI = 2, int NJ = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[NI][NJ], int mw0, int nw0, int lane) {
  const int khalf = lane >> 5;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int m = mw0 + j * 32 + (lane & 31);
    if (m >= p.M) continue;
    const int64_t bidx = (EPI == ST355_EPI_GATE_RESIDUAL) ? (int64_t)(m / p.rows_per_batch) : 0;
    int img = 0;
    const bool border = p.conv_taps ? conv_border(p, m, img) : false;
#pragma unroll
    for (int i = 0; i < NI; i++) {
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int n = nw0 + i * 32 + 8 * a + 4 * khalf;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int b = 0; b < 4; b++) v[b] = acc[i][j][4 * a + b];
        if (p.bias) {
          bf16x4 bv = *(const bf16x4*)(p.bias + n);
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] += bf2f(bv[b]);
        }
        if (p.img_add) {
          bf16x4 tv = *(const bf16x4*)(p.img_add + (int64_t)img * p.img_add_stride + n);
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] += bf2f(tv[b]);
        }
        if (EPI == ST355_EPI_GELU) {
          if (p.aux_out) {
            bf16x4 pre;
#pragma unroll
            for (int b = 0; b < 4; b++) pre[b] = f2bf(v[b]);
            *(bf16x4*)(p.aux_out + (int64_t)m * p.ld_aux_out + n) = pre;
#pragma unroll
            for (int b = 0; b < 4; b++) v[b] = bf2f(pre[b]);  // activation of the stored (rounded) pre-activation
          }
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = gelu_tanh(v[b]);
        } else if (EPI == ST355_EPI_GATE_RESIDUAL) {
          if (p.aux_out) {                                   // full fine-tune: keep the un-gated branch output y (d gate = sum dOut * y)
            bf16x4 yv;
#pragma unroll
            for (int b = 0; b < 4; b++) yv[b] = f2bf(v[b]);
            *(bf16x4*)(p.aux_out + (int64_t)m * p.ld_aux_out + n) = yv;
          }
          bf16x4 gv = *(const bf16x4*)(p.gate + bidx * p.gate_stride + n);
          bf16x4 rv = *(const bf16x4*)(p.aux_in + (int64_t)m * p.ld_aux_in + n);
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = bf2f(rv[b]) + bf2f(gv[b]) * v[b];
        } else if (EPI == ST355_EPI_ADD) {
          bf16x4 rv = *(const bf16x4*)(p.aux_in + (int64_t)m * p.ld_aux_in + n);
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] += bf2f(rv[b]);
        } else if (EPI == ST355_EPI_MUL_GELU_GRAD) {
          bf16x4 hv = *(const bf16x4*)(p.aux_in + (int64_t)m * p.ld_aux_in + n);
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] *= gelu_tanh_grad(bf2f(hv[b]));
        }
        bf16x4 o;
#pragma unroll
        for (int b = 0; b < 4; b++) o[b] = f2bf(border ? 0.f : v[b]);
        *(bf16x4*)(p.C + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
}

// ---- coalesced epilogue for the 256x256 schedules (wave tile 128 tokens x 64 features) ----
// The accumulator layout gives a lane 4 consecutive features of ONE token: direct stores touch 32 rows x 16 B per instruction and
// the L2 has to merge eight instructions into each 128-byte line (rocprofv3 WRITE_SIZE showed 2-2.4x the algorithmic bytes on the
// two-output GELU epilogue).  Here the wave transposes through its own slice of the (now idle) LDS ring in two 64-token passes,
// fp32, so that 8 consecutive lanes own one token's 64 features: bias / residual / pre-activation reads and all stores are whole
// 128-byte lines, and the epilogue arithmetic is still done once in fp32 before the single rounding to bf16.
#define EPL_PITCH 272                         // bytes per staged token row: 64 fp32 + 16 B pad (bank spread for the b128 writes)
#define EPL_WAVE (64 * EPL_PITCH)             // 17 KiB per wave
// CONV / F8: compile-time gates of the convolution (border zeroing, per-image add) and fp8 (row scales) code, so the plain bf16 GEMM carries none of it
template <int EPI, bool CONV = true, bool F8 = true>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmP& p, f32x16 (&acc)[2][4], int mw0, int nw0, int lane, char* stage) {
  const int khalf = lane >> 5, l31 = lane & 31;
  const int rrow = lane >> 3, rc = lane & 7;          // read side: 8 lanes per token row, 8 features each
  const int n = nw0 + rc * 8;
  const bool n_ok = n < p.N;
  float bias8[8];
#pragma unroll
  for (int b = 0; b < 8; b++) bias8[b] = 0.f;
  if (p.bias && n_ok) {
    const bf16x8 bv = *(const bf16x8*)(p.bias + n);
#pragma unroll
    for (int b = 0; b < 8; b++) bias8[b] = bf2f(bv[b]);
  }
  // row pointers advance by 8 rows per read-back step: ONE 64-bit multiply per operand up front, then adds of wave-uniform strides (the per-row
  // `m * ld` form cost ~6 quarter-rate integer instructions per operand per row — as many issue cycles as the GELU arithmetic itself)
  const int m_first = mw0 + rrow;
  bf16* c_row = p.C + (int64_t)m_first * p.ldc + n;
  const bf16* in_row = p.aux_in ? p.aux_in + (int64_t)m_first * p.ld_aux_in + n : nullptr;
  bf16* out_row = p.aux_out ? p.aux_out + (int64_t)m_first * p.ld_aux_out + n : nullptr;
  float* part_row = (EPI == EPI_SPLITK) ? p.partial + (int64_t)m_first * p.part_ld + n : nullptr;
  const int64_t c_step = 8 * p.ldc, in_step = 8 * p.ld_aux_in, out_step = 8 * p.ld_aux_out, part_step = 8 * p.part_ld;
#pragma unroll
  for (int ps = 0; ps < 2; ps++) {
    // write: token row jj*32 + l31, features i*32 + 8a + 4khalf .. +4
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][2 * ps + jj][4 * a + b];
          *(f32x4*)(stage + (jj * 32 + l31) * EPL_PITCH + (i * 32 + 8 * a + 4 * khalf) * 4) = v;
        }
    // read back token-major (same wave: LDS operations of one wave complete in order)
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + rrow;
      const int m = mw0 + ps * 64 + row;
      const f32x4 lo = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32);
      const f32x4 hi = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32 + 16);
      bf16* const c_ptr = c_row;
      const bf16* const in_ptr = in_row;
      bf16* const out_ptr = out_row;
      float* const part_ptr = part_row;
      c_row += c_step; in_row += in_step; out_row += out_step; part_row += part_step;
      if (m >= p.M || !n_ok) continue;
      if (EPI == EPI_SPLITK) {                        // fp32 K-slice slab (p.partial already points at this slice)
        *(f32x4*)part_ptr = lo;
        *(f32x4*)(part_ptr + 4) = hi;
        continue;
      }
      int img = 0;
      const bool border = (CONV && p.conv_taps) ? conv_border(p, m, img) : false;
      float v[8];
      if (F8 && p.scale_b) {                            // fp8 Linear: row-wise scaling of torch._scaled_mm (fp8_native.py:64-75)
        const float sa = p.scale_a[0];
        const f32x4 s0 = *(const f32x4*)(p.scale_b + n), s1 = *(const f32x4*)(p.scale_b + n + 4);
#pragma unroll
        for (int b = 0; b < 4; b++) { v[b] = lo[b] * (sa * s0[b]) + bias8[b]; v[4 + b] = hi[b] * (sa * s1[b]) + bias8[4 + b]; }
      } else {
#pragma unroll
        for (int b = 0; b < 4; b++) { v[b] = lo[b] + bias8[b]; v[4 + b] = hi[b] + bias8[4 + b]; }
      }
      if (CONV && p.img_add) {
        const bf16x8 tv = *(const bf16x8*)(p.img_add + (int64_t)img * p.img_add_stride + n);
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] += bf2f(tv[b]);
      }
      if (EPI == ST355_EPI_GELU) {
        if (p.aux_out) {
          bf16x8 pre;
#pragma unroll
          for (int b = 0; b < 8; b++) pre[b] = f2bf(v[b]);
          *(bf16x8*)out_ptr = pre;
#pragma unroll
          for (int b = 0; b < 8; b++) v[b] = bf2f(pre[b]);   // activation of the stored (rounded) pre-activation
        }
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = gelu_tanh(v[b]);
      } else if (EPI == ST355_EPI_GATE_RESIDUAL) {
        if (p.aux_out) {                                     // full fine-tune: keep the un-gated branch output y (d gate = sum dOut * y)
          bf16x8 yv;
#pragma unroll
          for (int b = 0; b < 8; b++) yv[b] = f2bf(v[b]);
          *(bf16x8*)out_ptr = yv;
        }
        const int64_t bidx = (int64_t)(m / p.rows_per_batch);
        const bf16x8 gv = *(const bf16x8*)(p.gate + bidx * p.gate_stride + n);
        const bf16x8 rv = *(const bf16x8*)in_ptr;
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = bf2f(rv[b]) + bf2f(gv[b]) * v[b];
      } else if (EPI == ST355_EPI_ADD) {
        const bf16x8 rv = *(const bf16x8*)in_ptr;
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] += bf2f(rv[b]);
      } else if (EPI == ST355_EPI_MUL_GELU_GRAD) {
        const bf16x8 hv = *(const bf16x8*)in_ptr;
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] *= gelu_tanh_grad(bf2f(hv[b]));
      }
      bf16x8 o;
#pragma unroll
      for (int b = 0; b < 8; b++) o[b] = f2bf(border ? 0.f : v[b]);
      *(bf16x8*)c_ptr = o;
    }
  }
}
// ---- ST355_EPI_QK_NORM_ROPE: q / k tiles of the fused QKV projection (256x256 schedule only) ----
// Same wave-private fp32 transpose as above (two 64-token passes).  A 128-channel head spans the two waves wn, wn^1 of a 128-token group: each wave
// reduces sum(x^2) over its 64 channels (8 lanes per token row, xor-shuffles), the halves meet through a 4 KiB exchange area behind the transpose
// slices (one workgroup barrier per pass; the V tiles of the same launch are other workgroups and take the plain epilogue).  Normalisation, the
// rotation and the single rounding to bf16 all happen on the fp32 accumulator values — the unfused form (st355_qk_norm_rope_fwd) rounded the
// projection to bf16 first and spent one full HBM read + write pass of the [tokens, 3D] tensor on it.
#define QKR_XCHG (2 * 8 * 64 * 4)               // [pass][wave][64 rows] fp32
__device__ __forceinline__ void gemm_epilogue_qkrope(const GemmP& p, f32x16 (&acc)[2][4], int m0, int n0, int wm, int wn, int wv, int lane, char* smem) {
  char* stage = smem + wv * EPL_WAVE;
  float* xchg = (float*)(smem + 8 * EPL_WAVE);
  const int khalf = lane >> 5, l31 = lane & 31;
  const int rrow = lane >> 3, rc = lane & 7;
  const int Dm = p.rH * 128;
  const int part = n0 >= Dm ? 1 : 0;                                    // 0: q, 1: k  (workgroup-uniform)
  const int ncol = n0 - part * Dm + wn * 64;                            // first column of this wave inside the part
  const int head = ncol >> 7, half = (ncol >> 6) & 1;
  const int cch = half * 64 + rc * 8;                                   // head channel of this lane's first feature
  const bf16* w = part ? p.rwk : p.rwq;
  bf16* dst = part ? p.rk : p.rq;
  float w8[8], bias8[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { w8[j] = w ? bf2f(w[cch + j]) : 1.f; bias8[j] = p.bias ? bf2f(p.bias[n0 + wn * 64 + rc * 8 + j]) : 0.f; }
  const int rpb = (int)p.rows_per_batch;
  const int bsm = __builtin_amdgcn_readfirstlane(m0 / rpb);             // sample of this tile (rows_per_batch is a multiple of 256)
  const int mw0 = m0 + wm * 128;
  const int pos_w = p.rpos0 + (mw0 - bsm * rpb);                        // joint sequence position of the wave's first token
  bf16* dst_h = dst + ((int64_t)bsm * p.rH + head) * (int64_t)p.rS * 128 + cch;
  float* rr_out = p.rrms + (int64_t)bsm * p.rS * (2 * p.rH) + part * p.rH + head;
#pragma unroll
  for (int ps = 0; ps < 2; ps++) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][2 * ps + jj][4 * a + b];
          *(f32x4*)(stage + (jj * 32 + l31) * EPL_PITCH + (i * 32 + 8 * a + 4 * khalf) * 4) = v;
        }
    float* xw = xchg + (ps * 8 + wv) * 64;
    const float* xp = xchg + (ps * 8 + (wv ^ 1)) * 64;
    // pass 1: this wave's half of sum((x + bias)^2) per token row
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + rrow;
      const f32x4 lo = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32);
      const f32x4 hi = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32 + 16);
      float sq = 0.f;
#pragma unroll
      for (int b = 0; b < 4; b++) { const float x0 = lo[b] + bias8[b], x1 = hi[b] + bias8[4 + b]; sq += x0 * x0 + x1 * x1; }
      sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
      if (rc == 0) xw[row] = sq;
    }
    // one (cos, sin) per rotation pair: [S, 64] tables (the full-width tables of flux/transformer.py:73-98 repeat every entry twice; read from them the
    // epilogue pulled 4x its own output bytes through the CU's L2 port — the two heads of a tile and the two entries of a pair — and cost more than the
    // pass it replaced).  Issued BEFORE the barrier: their L2 latency hides behind the wait for the partner wave's half sums.
    f32x4 csv[8], snv[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int pos = min(pos_w + ps * 64 + it * 8 + rrow, p.rS - 1);
      csv[it] = *(const f32x4*)(p.rcos + (int64_t)pos * 64 + (cch >> 1));
      snv[it] = *(const f32x4*)(p.rsin + (int64_t)pos * 64 + (cch >> 1));
    }
    __syncthreads();
    // pass 2: normalise, rotate, store head-major
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + rrow;
      const int m = mw0 + ps * 64 + row;
      const f32x4 lo = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32);
      const f32x4 hi = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32 + 16);
      const float rr = w ? rsqrtf((xw[row] + xp[row]) * (1.f / 128.f) + p.reps) : 1.f;
      const int pos = pos_w + ps * 64 + row;
      float y[8];
      if (m < p.M) {
        const f32x4 cs = csv[it], sn = snv[it];
#pragma unroll
        for (int b = 0; b < 4; b++) { y[b] = (lo[b] + bias8[b]) * rr * w8[b]; y[4 + b] = (hi[b] + bias8[4 + b]) * rr * w8[4 + b]; }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {      // out = x*cos + rot(x)*sin on interleaved pairs (flux/transformer.py:91-98), as k_qk_norm_rope_fwd
          o[j] = f2bf(y[j] * cs[j >> 1] - y[j + 1] * sn[j >> 1]);
          o[j + 1] = f2bf(y[j + 1] * cs[j >> 1] + y[j] * sn[j >> 1]);
        }
        *(bf16x8*)(dst_h + (int64_t)pos * 128) = o;
        if (rc == 0 && half == 0) rr_out[(int64_t)pos * (2 * p.rH)] = rr;
      }
    }
  }
}

// v tiles of the same launch: the row-major V rows (what the backward reads) as in gemm_epilogue_lds, plus — when rvt is given — the head-major V^T
// [B, H, 128, Sp] the forward attention kernel streams, read back TRANSPOSED from the same fp32 staging slice (8 lanes = 64 consecutive tokens of one
// channel = one 128-byte line): the transposed copy costs one extra write of V and no read pass.
__device__ __forceinline__ void gemm_epilogue_vdual(const GemmP& p, bf16* C, f32x16 (&acc)[2][4], int m0, int n0, int wm, int wn, int wv, int lane, char* smem) {
  char* stage = smem + wv * EPL_WAVE;
  const int khalf = lane >> 5, l31 = lane & 31;
  const int rrow = lane >> 3, rc = lane & 7;
  const int Dm = p.rH * 128;
  const int ncol = n0 - 2 * Dm + wn * 64;                               // first column of this wave inside the v part
  const int head = ncol >> 7, half = (ncol >> 6) & 1;
  const int n = n0 + wn * 64 + rc * 8;
  float bias8[8];
#pragma unroll
  for (int j = 0; j < 8; j++) bias8[j] = p.bias ? bf2f(p.bias[n + j]) : 0.f;
  const int rpb = (int)p.rows_per_batch;
  const int bsm = __builtin_amdgcn_readfirstlane(m0 / rpb);
  const int mw0 = m0 + wm * 128;
  const int pos_w = p.rpos0 + (mw0 - bsm * rpb);
  bf16* c_row = C + (int64_t)(mw0 + rrow) * p.ldc + (n - 2 * Dm);
  const int64_t c_step = 8 * p.ldc;
  // transposed side: lane = (channel fsub of 8, token group tg of 8 tokens)
  const int fsub = lane >> 3, tg = lane & 7;
  bf16* vt_w = p.rvt ? p.rvt + (((int64_t)bsm * p.rH + head) * 128 + half * 64) * (int64_t)p.rSp + pos_w + tg * 8 : nullptr;
#pragma unroll
  for (int ps = 0; ps < 2; ps++) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][2 * ps + jj][4 * a + b];
          *(f32x4*)(stage + (jj * 32 + l31) * EPL_PITCH + (i * 32 + 8 * a + 4 * khalf) * 4) = v;
        }
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + rrow;
      const int m = mw0 + ps * 64 + row;
      const f32x4 lo = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32);
      const f32x4 hi = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32 + 16);
      bf16* const c_ptr = c_row;
      c_row += c_step;
      if (m >= p.M) continue;
      bf16x8 o;
#pragma unroll
      for (int b = 0; b < 4; b++) { o[b] = f2bf(lo[b] + bias8[b]); o[4 + b] = f2bf(hi[b] + bias8[4 + b]); }
      *(bf16x8*)c_ptr = o;
    }
    if (vt_w) {
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int f = it * 8 + fsub;                                      // channel inside the wave's 64
        const float bf_ = p.bias ? bf2f(p.bias[n0 + wn * 64 + f]) : 0.f;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = f2bf(*(const float*)(stage + (tg * 8 + e) * EPL_PITCH + f * 4) + bf_);
        if (mw0 + ps * 64 + tg * 8 < p.M) *(bf16x8*)(vt_w + (int64_t)f * p.rSp + ps * 64) = o;
      }
    }
  }
}

// ---- ST355_EPI_GEGLU / ST355_EPI_GEGLU_GRAD (256x256 schedule): the UNet feed-forward's value * gelu(gate) inside the two GEMMs around it (st355.h) ----
// Same wave-private fp32 transpose as gemm_epilogue_lds (two 64-token passes; read side: 8 lanes per token row, 8 features each).  With the interleaved weight
// rows a wave's 64 columns are [32 values | the 32 gates of the same features]: lanes rc < 4 read their 8 values AND the 8 gates 32 columns further from the
// staging row, no cross-wave exchange.  Forward: every lane stores its 8 pre-activations (the backward's operand, interleaved order), lanes rc < 4 the 8 outputs.
__device__ __forceinline__ void gemm_epilogue_geglu(const GemmP& p, f32x16 (&acc)[2][4], int mw0, int nw0, int lane, char* stage) {
  const int khalf = lane >> 5, l31 = lane & 31;
  const int rrow = lane >> 3, rc = lane & 7;
  const int n = nw0 + rc * 8;                                  // interleaved column of the 8 pre-activations this lane stores
  const bool n_ok = n < p.N;
  // the activation uses ALL lanes: lane (row, rc) produces outputs 4 rc .. 4 rc + 3 of the wave's 32 features (value at staging column 4 rc, gate at 32 + 4 rc)
  float bias8[8], biasv[4], biasg[4];
#pragma unroll
  for (int b = 0; b < 8; b++) bias8[b] = 0.f;
#pragma unroll
  for (int b = 0; b < 4; b++) { biasv[b] = 0.f; biasg[b] = 0.f; }
  if (p.bias && n_ok) {
    const bf16x8 bv = *(const bf16x8*)(p.bias + n);
#pragma unroll
    for (int b = 0; b < 8; b++) bias8[b] = bf2f(bv[b]);
    const bf16x4 vv = *(const bf16x4*)(p.bias + nw0 + 4 * rc), gv = *(const bf16x4*)(p.bias + nw0 + 32 + 4 * rc);
#pragma unroll
    for (int b = 0; b < 4; b++) { biasv[b] = bf2f(vv[b]); biasg[b] = bf2f(gv[b]); }
  }
  const int jf = (nw0 >> 1) + 4 * rc;                          // first output feature of this lane: 32 * (nw0 / 64) + 4 rc
  const int m_first = mw0 + rrow;
  bf16* c_row = p.C + (int64_t)m_first * p.ldc + jf;
  bf16* out_row = p.aux_out + (int64_t)m_first * p.ld_aux_out + n;
  const int64_t c_step = 8 * p.ldc, out_step = 8 * p.ld_aux_out;
#pragma unroll
  for (int ps = 0; ps < 2; ps++) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][2 * ps + jj][4 * a + b];
          *(f32x4*)(stage + (jj * 32 + l31) * EPL_PITCH + (i * 32 + 8 * a + 4 * khalf) * 4) = v;
        }
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + rrow;
      const int m = mw0 + ps * 64 + row;
      const f32x4 lo = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32);
      const f32x4 hi = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32 + 16);
      const f32x4 av = *(const f32x4*)(stage + row * EPL_PITCH + rc * 16);               // values 4 rc .. 4 rc + 3
      const f32x4 ag = *(const f32x4*)(stage + row * EPL_PITCH + 128 + rc * 16);         // their gates (staging column 32 + 4 rc)
      bf16* const c_ptr = c_row;
      bf16* const out_ptr = out_row;
      c_row += c_step; out_row += out_step;
      if (m >= p.M || !n_ok) continue;
      bf16x8 pre;
#pragma unroll
      for (int b = 0; b < 4; b++) { pre[b] = f2bf(lo[b] + bias8[b]); pre[4 + b] = f2bf(hi[b] + bias8[4 + b]); }
      *(bf16x8*)out_ptr = pre;
      bf16x4 o;
#pragma unroll
      for (int b = 0; b < 4; b++)                                                       // both halves rounded to bf16 first, as st355_geglu_fwd reads them
        o[b] = f2bf(bf2f(f2bf(av[b] + biasv[b])) * gelu_erf(bf2f(f2bf(ag[b] + biasg[b]))));
      *(bf16x4*)c_ptr = o;
    }
  }
}
// backward: acc = d out for the wave's 64 features j (two interleave groups of 32): lane (row, rc) owns features 8 rc .. 8 rc + 7, i.e. group rc >> 2, offset
// 8 (rc & 3): it reads value / gate of those features from the kept pre-activation (columns 64 G + off and + 32) and writes d value / d gate to the same columns of C
__device__ __forceinline__ void gemm_epilogue_geglu_grad(const GemmP& p, f32x16 (&acc)[2][4], int mw0, int nw0, int lane, char* stage) {
  const int khalf = lane >> 5, l31 = lane & 31;
  const int rrow = lane >> 3, rc = lane & 7;
  const bool n_ok = nw0 + rc * 8 < p.N;
  const int col = 2 * nw0 + 64 * (rc >> 2) + (rc & 3) * 8;     // 64 * (nw0 / 32 + (rc >> 2)) + 8 (rc & 3)
  const int m_first = mw0 + rrow;
  bf16* c_row = p.C + (int64_t)m_first * p.ldc + col;
  const bf16* in_row = p.aux_in + (int64_t)m_first * p.ld_aux_in + col;
  const int64_t c_step = 8 * p.ldc, in_step = 8 * p.ld_aux_in;
#pragma unroll
  for (int ps = 0; ps < 2; ps++) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][2 * ps + jj][4 * a + b];
          *(f32x4*)(stage + (jj * 32 + l31) * EPL_PITCH + (i * 32 + 8 * a + 4 * khalf) * 4) = v;
        }
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + rrow;
      const int m = mw0 + ps * 64 + row;
      const f32x4 lo = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32);
      const f32x4 hi = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32 + 16);
      bf16* const c_ptr = c_row;
      const bf16* const in_ptr = in_row;
      c_row += c_step; in_row += in_step;
      if (m >= p.M || !n_ok) continue;
      const bf16x8 vv = *(const bf16x8*)in_ptr, gv = *(const bf16x8*)(in_ptr + 32);
      bf16x8 dv, dg;
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const float d = bf2f(f2bf(b < 4 ? lo[b] : hi[b - 4]));                           // d out rounded to bf16, as the unfused pair (GEMM store, st355_geglu_bwd read)
        const float g = bf2f(gv[b]);
        dv[b] = f2bf(d * gelu_erf(g));
        dg[b] = f2bf(d * bf2f(vv[b]) * gelu_erf_grad(g));
      }
      *(bf16x8*)c_ptr = dv;
      *(bf16x8*)(c_ptr + 32) = dg;
    }
  }
}

// ---- ST355_EPI_HEADS (256x256 schedule): 64-wide heads written head-major straight from the accumulators (st355.h) ----
// A wave's 64 columns are exactly one head.  Same wave-private fp32 transpose as gemm_epilogue_lds; the read side (8 lanes per token row, 8 features each) stores a
// token's 64 head channels as one 128-byte line of Q / K [B, H, S, 64]; v heads keep their row-major C rows and — as gemm_epilogue_vdual — leave through a
// transposed read of the staging slice as well (8 lanes = 64 consecutive tokens of one channel) when the V^T destination is given.  The sample / position of a row
// come from one integer division per row (rows_per_batch is arbitrary: SD3's 231 text rows, a UNet level's 1024).
__device__ __forceinline__ void gemm_epilogue_heads(const GemmP& p, bf16* C, f32x16 (&acc)[2][4], int mw0, int nw0, int lane, char* stage) {
  const int khalf = lane >> 5, l31 = lane & 31;
  const int rrow = lane >> 3, rc = lane & 7;
  const int part = nw0 < p.hq ? 0 : (nw0 < p.hq + p.hk ? 1 : 2);                   // wave-uniform: q, k or v head
  const int ncol = nw0 - (part == 0 ? 0 : part == 1 ? p.hq : p.hq + p.hk);        // first column of this wave inside its part
  const int head = ncol >> 6;
  const bool n_ok = nw0 < p.N;
  float bias8[8];
#pragma unroll
  for (int j = 0; j < 8; j++) bias8[j] = (p.bias && n_ok) ? bf2f(p.bias[nw0 + rc * 8 + j]) : 0.f;
  const int rpb = (int)p.rows_per_batch;
  bf16* hm = part == 0 ? p.rq : p.rk;
  const int fsub = lane >> 3, tg = lane & 7;                                        // transposed side (v heads): channel fsub of 8, token group tg of 8 tokens
#pragma unroll
  for (int ps = 0; ps < 2; ps++) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][2 * ps + jj][4 * a + b];
          *(f32x4*)(stage + (jj * 32 + l31) * EPL_PITCH + (i * 32 + 8 * a + 4 * khalf) * 4) = v;
        }
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + rrow;
      const int m = mw0 + ps * 64 + row;
      const f32x4 lo = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32);
      const f32x4 hi = *(const f32x4*)(stage + row * EPL_PITCH + rc * 32 + 16);
      if (m >= p.M || !n_ok) continue;
      bf16x8 o;
#pragma unroll
      for (int b = 0; b < 4; b++) { o[b] = f2bf(lo[b] + bias8[b]); o[4 + b] = f2bf(hi[b] + bias8[4 + b]); }
      if (part < 2) {
        const int bsm = m / rpb;
        const int pos = p.rpos0 + (m - bsm * rpb);
        *(bf16x8*)(hm + (((int64_t)bsm * p.rH + head) * p.rS + pos) * 64 + rc * 8) = o;
      } else {
        *(bf16x8*)(C + (int64_t)m * p.ldc + ncol + rc * 8) = o;
      }
    }
    if (part == 2 && p.rvt && n_ok) {
      const int m8 = mw0 + ps * 64 + tg * 8;                                        // first of this lane's 8 consecutive tokens (same sample: rows_per_batch % 8 == 0)
      if (m8 < p.M) {
        const int bsm = m8 / rpb;
        const int pos = p.rpos0 + (m8 - bsm * rpb);
        bf16* vt = p.rvt + (((int64_t)bsm * p.rH + head) * 64) * (int64_t)p.rSp + pos;
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const int f = it * 8 + fsub;
          const float bf_ = p.bias ? bf2f(p.bias[nw0 + f]) : 0.f;
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e++) o[e] = f2bf(*(const float*)(stage + (tg * 8 + e) * EPL_PITCH + f * 4) + bf_);
          *(bf16x8*)(vt + (int64_t)f * p.rSp) = o;
        }
      }
    }
  }
}

// the coalesced path needs 16-byte alignment of every row it touches with bf16x8 accesses
__device__ __forceinline__ bool epl_aligned(const GemmP& p) {
  bool ok = (p.N % 8 == 0) && (p.ldc % 8 == 0) && (((uintptr_t)p.C & 15) == 0);
  if (p.bias) ok = ok && (((uintptr_t)p.bias & 15) == 0);
  if (p.aux_out) ok = ok && (p.ld_aux_out % 8 == 0) && (((uintptr_t)p.aux_out & 15) == 0);
  if (p.aux_in) ok = ok && (p.ld_aux_in % 8 == 0) && (((uintptr_t)p.aux_in & 15) == 0);
  if (p.gate) ok = ok && (p.gate_stride % 8 == 0) && (((uintptr_t)p.gate & 15) == 0);
  if (p.img_add) ok = ok && (p.img_add_stride % 8 == 0) && (((uintptr_t)p.img_add & 15) == 0);
  return ok;
}

// one K=64 tile of MFMA work for a wave: 4 k-steps x (2 W frags + 2 X frags -> 4 MFMAs)
__device__ __forceinline__ void mma_tile(const char* xs, const char* ws, const int (&x_off)[2], const int (&x_sw)[2],
                                         const int (&w_off)[2], const int (&w_sw)[2], int khalf, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    const int c = 2 * ks + khalf;
    bf16x8 wf[2], xf[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      wf[i] = *(const bf16x8*)(ws + w_off[i] + ((c ^ w_sw[i]) << 4));
      xf[i] = *(const bf16x8*)(xs + x_off[i] + ((c ^ x_sw[i]) << 4));
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
  }
}

// =================================================================================================
// k_gemm_p3: 256x128x64, 8 waves, 3-stage LDS-DMA ring, counted vmcnt
// =================================================================================================
#define P3_BM 256
#define P3_BN 128
#define P3_THREADS 512
#define P3_XBYTES (P3_BM * BK * 2)          // 32 KiB
#define P3_WBYTES (P3_BN * BK * 2)          // 16 KiB
#define P3_STAGE (P3_XBYTES + P3_WBYTES)    // 48 KiB
#define P3_LDS (3 * P3_STAGE)               // 144 KiB

template <int EPI>
__global__ void __launch_bounds__(P3_THREADS, 2) k_gemm_p3(GemmGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;     // 4 (tokens) x 2 (features) waves

  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int pi = id >= g.tiles0 ? 1 : 0;
  if (pi) id -= g.tiles0;
  const GemmP& p0 = g.p[pi];
  const int nbm = (p0.M + P3_BM - 1) / P3_BM, nbn = (p0.N + P3_BN - 1) / P3_BN;
  int pm, pn;
  tile_coords(id, nbm, nbn, pm, pn);
  const int m0 = pm * P3_BM, n0 = pn * P3_BN;
  const GemmP p = seg_view(p0, m0);
  const int nt1 = p.K / BK;
  const int nt = nt1 + p.K2 / BK;

  // hoist everything the K loop needs out of the (dynamically indexed) kernarg struct: scalars live in SGPRs, the per-lane
  // source offsets of the main segment are computed once (the loop only adds k0)
  const bf16* A1 = p.A; const bf16* B1 = p.B; const bf16* A2 = p.A2; const bf16* B2 = p.B2;
  const int64_t la2 = p.lda2, lb2 = p.ldb2;
  const int M = p.M, N = p.N;
  const int st_row = lane >> 3, st_cp = lane & 7;
  int64_t xo[4], wo[2];
  int xrow[4], wrow[2], xc[4], wc[2];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int row = (wv * 4 + j) * 8 + st_row;
    xc[j] = (st_cp ^ ((row >> 1) & 7)) * 8;
    xrow[j] = min(m0 + row, M - 1);
    xo[j] = (int64_t)xrow[j] * p.lda + xc[j];
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int row = (wv * 2 + j) * 8 + st_row;
    wc[j] = (st_cp ^ ((row >> 1) & 7)) * 8;
    wrow[j] = min(n0 + row, N - 1);
    wo[j] = (int64_t)wrow[j] * p.ldb + wc[j];
  }
  auto stage = [&](int t, int buf) {
    char* xs = smem + buf * P3_STAGE;
    char* ws = xs + P3_XBYTES;
    if (t < nt1) {
      const int k0 = t * BK;
#pragma unroll
      for (int j = 0; j < 4; j++) glds16(A1 + xo[j] + k0, xs + (wv * 4 + j) * 1024);
#pragma unroll
      for (int j = 0; j < 2; j++) glds16(B1 + wo[j] + k0, ws + (wv * 2 + j) * 1024);
    } else {                                  // low-rank K-extension segment (1-2 tiles)
      const int k0 = (t - nt1) * BK;
#pragma unroll
      for (int j = 0; j < 4; j++) glds16(A2 + (int64_t)xrow[j] * la2 + xc[j] + k0, xs + (wv * 4 + j) * 1024);
#pragma unroll
      for (int j = 0; j < 2; j++) glds16(B2 + (int64_t)wrow[j] * lb2 + wc[j] + k0, ws + (wv * 2 + j) * 1024);
    }
  };

  int w_off[2], w_sw[2], x_off[2], x_sw[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int wr = wn * 64 + i * 32 + (lane & 31);
    w_off[i] = wr * 128; w_sw[i] = (wr >> 1) & 7;
    const int xr = wm * 64 + i * 32 + (lane & 31);
    x_off[i] = xr * 128; x_sw[i] = (xr >> 1) & 7;
  }
  const int khalf = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // ---- prologue: two tiles in flight; tile 0 retired by a counted wait ----
  stage(0, 0);
  if (nt > 1) {
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int t = 0; t < nt; t++) {
    // tile t+2 goes into the buffer read during iteration t-1 (every wave finished those reads before the last barrier)
    if (t + 2 < nt) stage(t + 2, cur == 0 ? 2 : cur - 1);
    const char* xs = smem + cur * P3_STAGE;
    mma_tile(xs, xs + P3_XBYTES, x_off, x_sw, w_off, w_sw, khalf, acc);
    if (t + 1 < nt) {
      // retire tile t+1 (this wave's 6 oldest LDS-DMAs), leave tile t+2 in flight across the barrier
      if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my ds_reads of `cur` are done before anyone may overwrite it
      __builtin_amdgcn_s_barrier();
    }
    cur = (cur == 2) ? 0 : cur + 1;
  }
  gemm_epilogue<EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// raw workgroup barrier (no vmcnt / lgkmcnt drain: the schedule keeps LDS-DMA in flight across it and waits with counted s_waitcnt)
#define PP_BARRIER()                                   \
  do {                                                 \
    asm volatile("" ::: "memory");                     \
    __builtin_amdgcn_s_barrier();                      \
    asm volatile("" ::: "memory");                     \
  } while (0)

// =================================================================================================
// k_gemm_pq: 256x256 tile, BK=64 (128-byte rows = whole cache lines: the LDS-DMA path moves 43.7 B/clk/CU with 128-B rows but
// only 29 B/clk/CU with 64-B rows — tools/probes/glds_probe.hip), ping-pong schedule (T3+T4+T5): two groups of 4 waves, one wave of each per SIMD,
// group 1 one barrier interval behind, so one wave of a SIMD issues MFMAs at raised priority while its partner issues ds_reads + one LDS-DMA refill.
// A K-tile (64 KiB) is four 16-KiB REGIONS, grouped by the phase that reads them rather than by tile rows:
//   XA = the first 64 tokens of each group's 128 (tile rows 0-63, 128-191)   XB = the last 64 (64-127, 192-255)
//   WA = the first 32 features of each wave's 64                            WB = the last 32
// Phases of K-tile t (buffer t&1):  P0: WA x XA (reads XA, WA)   P1: WB x XA (reads WB)   P2: WB x XB (reads XB)   P3: WA x XB (-)
// Each phase = 8 MFMAs (2 accumulators x 4 k-steps) and ONE region refill (2 LDS-DMA pieces per wave), issued in the order
//   ... XA(t+2)@P2  WA(t+2)@P3  WB(t+2)@P0'  XB(t+2)@P1' ...   — every region is refilled >= 2 phases after its last read
// (WAR-safe across the one-barrier stagger of the two wave groups) and retired by the uniform  s_waitcnt vmcnt(8)  at the end of
// each phase's load section, 4 phases after its issue (RAW-safe: it is first read one phase after that wait).
// =================================================================================================
#define PQ_BM 256
#define PQ_BN 256
#define PQ_BK 64
#define PQ_THREADS 512
#define PQ_REGION 16384
#define PQ_BUF (4 * PQ_REGION)              // 64 KiB: [XA][XB][WA][WB]
#define PQ_LDS (8 * 64 * 272)              // 136 KiB: two 64-KiB K-tile buffers; the epilogue transpose uses 8 x 17 KiB
#ifndef PQ_BUFLD
#define PQ_BUFLD 1                          // 1: LDS-DMA by buffer_load ... lds (SRSRC + 32-bit voffset + scalar K offset) instead of global_load_lds
#endif
#ifndef PQ_VM
#define PQ_VM 8                             // LDS-DMA pieces that may stay in flight at a phase's counted wait (lab knob; 8 = four regions)
#endif
#ifndef PQ_ABL
#define PQ_ABL 0                            // lab-only ablations (wrong results!): bit 0 = no fragment reads in the loop, bit 1 = no LDS-DMA refills
#endif
#ifndef PQ_PRIO
#ifndef PQ_F8_MX
#define PQ_F8_MX 1                          // fp8-native Linear on the K = 64 f8f6f4 MFMA (2x the bf16 issue rate) with unit block scales; 0 = the K = 16 fp8_bf8 form
#endif
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8 frag32(const long* p) {       // four consecutive 8-byte fragment pieces -> the 32-byte operand of the K = 64 MFMA
  i32x8 v;
#pragma unroll
  for (int q = 0; q < 4; q++) { v[2 * q] = (int)(p[q] & 0xffffffffl); v[2 * q + 1] = (int)(p[q] >> 32); }
  return v;
}
#define PQ_PRIO 1                           // raise the wave priority around the MFMA clusters (T5)
#endif
#ifndef PQ_GL
#define PQ_GL 1                             // where a phase issues its LDS-DMA pieces: 0 before the ds_reads, 1 after them, 2 at the end of the previous MFMA section
#endif

__device__ __forceinline__ void wait_vm_rt(int n) {    // n = LDS-DMA pieces that may stay in flight (even, 0..8)
  if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// TN = true: the "weight-gradient" form  C[P,Q] = sum_m L[m,P] R[m,Q]  (GemmP: A = L, B = R, M = P, N = Q, K = contraction length):
// both operands have the contraction index as their SLOW axis.  The ring, the phases and the epilogue are unchanged; a region is
// then [64 contraction rows][128 output columns] (256-byte rows, 16-byte chunks XOR-ed with (row&3)<<2 on the DMA source side) and
// the MFMA fragments are gathered by the transposing LDS read (two ds_read_b64_tr_b16 per k-fragment).
// F8 = true: the fp8-native Linear (fp8.hip): A = e5m2 activations, B = e4m3 weights, one byte per element.  The byte geometry of the
// ring is unchanged (128-byte rows = 128 K-elements per K-tile, same swizzle); a phase then runs 16 MFMAs (8 k-steps of
// v_mfma_f32_32x32x16_fp8_bf8) on fragments fetched with 8-byte reads, i.e. twice the MFMA work per LDS-DMA byte.
// SK = 2 / 3 / 4 (slices per cut tile): the stream-K tail.  A problem whose 256x256 tiles leave the chip's last round mostly empty (320 tiles on 256 CUs: the N = 1280 projections of the SDXL
// 32^2 level at batch 16 ran two rounds for 1.25 rounds of work — 733 TFLOP/s against 1137 for the N = 3840 projection of the same rows, rocprofv3 r06) has the
// tiles of that round cut along K: block sk_first + l runs K-slice (l >> 3) % sk_s of tile sk_first + 8 * ((l >> 3) / sk_s) + (l & 7) (the slices of a tile share
// an XCD).  Slices 0 .. sk_s - 2 leave their fp32 accumulators in a slab (lane-linear: one 16-byte store per lane and register quad), wait for the L2's write
// acknowledgements (same XCD = same L2: no cache-wide write-back / invalidate) and count themselves in; the LAST slice — which also owns the low-rank K-extension — waits for the count, adds the slabs in slice order (fixed order:
// deterministic) and runs the tile's ordinary epilogue, then zeroes the counter for the next launch.  At most 128 waiting workgroups exist and the slices they wait
// for never wait themselves, so every slice is scheduled whatever else the chip is running.
template <int EPI, bool TN, bool F8 = false, bool CONV = false, int SK = 0>     // SK: slices per cut tile (0 = no stream-K tail); CONV: the A operand's K-tiles are row-shifted views (st355_conv_bf16)
__global__ void __launch_bounds__(PQ_THREADS, 2) k_gemm_pq(GemmGroup g) {
  static_assert(!SK || (!TN && !F8 && !CONV && EPI != EPI_SPLITK), "the stream-K tail is built for the plain NT bf16 problems");
  constexpr int ES = F8 ? 1 : 2;               // bytes per operand element
  constexpr int KSN = F8 ? 8 : 4;              // k-steps (MFMAs per accumulator) per K-tile
  static_assert(!(F8 && TN), "fp8 weight gradients are not built");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;     // wm: group (128-token half); wn: 64-feature column of the tile

  // split-K (EPI_SPLITK): blockIdx = slice * tiles0 + tile; slice s owns K-tiles [s*per, min(all, (s+1)*per)) and writes an fp32 slab
  int slice = (EPI == EPI_SPLITK) ? blockIdx.x / g.tiles0 : 0;
  int id = (EPI == EPI_SPLITK) ? xcd_remap(blockIdx.x % g.tiles0, g.tiles0) : xcd_remap(blockIdx.x, SK ? min((int)gridDim.x, g.sk_first) : (int)gridDim.x);
  int sk_n = 1, sk_tl = 0;                 // SK: slices of this block's tile, the tile's index among the cut tiles
  if (SK && (int)blockIdx.x >= g.sk_first) {
    const int l = blockIdx.x - g.sk_first;
    sk_n = SK;
    const int lq = __builtin_amdgcn_readfirstlane((l >> 3) / sk_n);       // (readfirstlane: the division is expanded on the VALU — see segi below)
    slice = (l >> 3) - lq * sk_n;
    sk_tl = 8 * lq + (l & 7);
    if (sk_tl >= g.tiles0 - g.sk_first) return;         // (the cut-tile count rounded up to the 8 XCDs)
    id = g.sk_first + sk_tl;
  }
  const int pi = id >= g.tiles0 ? 1 : 0;
  if (pi) id -= g.tiles0;
  const GemmP& p = g.p[pi];
  const int nbm = (p.M + PQ_BM - 1) / PQ_BM, nbn = (p.N + PQ_BN - 1) / PQ_BN;
  // TN with conv_taps == 9 (weight gradient of a 3x3 convolution, st355_conv_wgrad_bf16): the nine taps are nine column blocks of the output
  // ([P, 9*N], block tap at columns tap*N) whose R operand is the SAME matrix shifted by (ty*Wp + tx) contraction rows: one launch for all taps
  const int wtaps = (TN && p.conv_taps == 9) ? 9 : 1;
  int pm, pn;
  tile_coords(id, nbm, nbn * wtaps, pm, pn, TN ? ST355_TILE_GROUP : tile_group_for(p.K + p.K2));
  const int wtap = pn / nbn;
  pn -= wtap * nbn;
  const int m0 = pm * PQ_BM, n0 = pn * PQ_BN;
  // segmented rows: this tile's segment index.  Only the two staging bases (below) and the epilogue's private GemmP copy (at the end) see it — the
  // K loop's registers are exactly those of the plain kernel (a by-value seg_view() copy up here cost the aux_in epilogues 15 VGPRs and 15 % speed).
  // seg_x* are 0 for absent operands and when segments are off (to_p), so the pointer arithmetic is unconditional.
  // (readfirstlane: the integer division is expanded on the VALU, which would leave a wave-uniform value — and every pointer derived from it — in VGPRs)
  const int64_t segi = (!TN && !F8 && !CONV && p.seg_rows) ? __builtin_amdgcn_readfirstlane(m0 / p.seg_rows) : 0;
  const int nt_all = p.K / (PQ_BK * 2 / ES);
  const int per = (EPI == EPI_SPLITK) ? (nt_all + p.ksplit - 1) / p.ksplit : SK ? __builtin_amdgcn_readfirstlane((nt_all + sk_n - 1) / sk_n) : nt_all;
  const int t_first = slice * per;
  const int nt1 = (EPI == EPI_SPLITK || SK) ? max(0, min(nt_all, t_first + per) - t_first) : nt_all;
  const int nt = nt1 + ((EPI == EPI_SPLITK || (SK && slice != sk_n - 1)) ? 0 : p.K2 / PQ_BK);

  const bf16* A1 = p.A + segi * p.seg_xa; const bf16* B1 = p.B; const bf16* A2 = p.A2 + segi * p.seg_xa2; const bf16* B2 = p.B2;
  const int64_t la2 = p.lda2, lb2 = p.ldb2;
  const int M = p.M, N = p.N;
  // staging: a piece = 8 region rows x 128 B; this wave issues pieces 2wv, 2wv+1 of whichever region is being refilled.
  // region row lr -> tile row:  X: (lr>>6)*128 + (lr&63) [+64 for XB]     W: (lr>>5)*64 + (lr&31) [+32 for WB]
  // Addresses are (uniform 64-bit base of the tile's first row + k) + a 32-bit per-lane byte offset: the LDS-DMA takes the
  // SGPR-base + VGPR-offset form, so the K loop spends no VALU (and only 8 VGPRs) on addressing.
  const char* xbase = TN ? (const char*)A1 + (int64_t)m0 * 2 : (const char*)A1 + (int64_t)m0 * p.lda * ES;
  const char* wbase = TN ? (const char*)B1 + (int64_t)n0 * 2 + (int64_t)((wtap / 3) * p.conv_wp + (wtap % 3)) * (wtaps > 1 ? p.ldb * 2 : 0)
                         : (const char*)B1 + (int64_t)n0 * p.ldb * ES;
  const uint32_t lda_b = (uint32_t)p.lda * ES, ldb_b = (uint32_t)p.ldb * ES;
  const int64_t xk_step = TN ? (int64_t)PQ_BK * lda_b : PQ_BK * 2;     // bytes per K-tile along the contraction
  const int64_t wk_step = TN ? (int64_t)PQ_BK * ldb_b : PQ_BK * 2;
  constexpr bool conv = CONV;                  // compile-time: the tap arithmetic costs the plain GEMM's K loop nothing
  static_assert(!(CONV && (TN || F8)), "conv mode is the NT bf16 form");
  if (!conv) xbase += t_first * xk_step;
  wbase += t_first * wk_step;

  auto x_rel = [&](int lr, int r) { return min(m0 + (lr >> 6) * 128 + (lr & 63) + r * 64, M - 1) - m0; };
  auto w_rel = [&](int lr, int r) { return min(n0 + (lr >> 5) * 64 + (lr & 31) + r * 32, N - 1) - n0; };
  uint32_t xo[2][2], wo[2][2];               // [region A/B][piece] byte offsets from xbase / wbase
#pragma unroll
  for (int j = 0; j < 2; j++) {
    if (TN) {
      // piece = 4 contraction rows x 256 B; LDS chunk position c <- source chunk c ^ ((row&3)<<2)
      const int r4 = (wv * 2 + j) * 4 + (lane >> 4);
      const int cs = (lane & 15) ^ ((r4 & 3) << 2);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int xc = min(m0 + (cs >> 3) * 128 + (cs & 7) * 8 + r * 64, M - 8) - m0;     // output-row (p) columns of L
        const int wc = min(n0 + (cs >> 2) * 64 + (cs & 3) * 8 + r * 32, N - 8) - n0;      // output-col (q) columns of R
        xo[r][j] = (uint32_t)r4 * lda_b + (uint32_t)(xc * 2);
        wo[r][j] = (uint32_t)r4 * ldb_b + (uint32_t)(wc * 2);
      }
    } else {
      const int lr = (wv * 2 + j) * 8 + (lane >> 3);
      const uint32_t scb = (uint32_t)(((lane & 7) ^ ((lr >> 1) & 7)) * 16);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        xo[r][j] = (uint32_t)x_rel(lr, r) * lda_b + scb;
        wo[r][j] = (uint32_t)w_rel(lr, r) * ldb_b + scb;
      }
    }
  }
  auto stage_ext = [&](const bf16* base, int64_t ld, int row0, bool is_x, int r, int k0, char* dst) {   // low-rank K-extension (rare)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int lr = (wv * 2 + j) * 8 + (lane >> 3);
      const int rel = is_x ? x_rel(lr, r) : w_rel(lr, r);
      glds16(base + (int64_t)(row0 + rel) * ld + ((lane & 7) ^ ((lr >> 1) & 7)) * 8 + k0, dst + j * 1024);
    }
  };
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, 0x7FFFFFFF, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7FFFFFFF, 0x00020000);
  auto stage_x = [&](int u, int r) {        // r = 0: XA, 1: XB
    char* dst = smem + (u & 1) * PQ_BUF + r * PQ_REGION + wv * 2048;
    if (u < nt1) {
      const int x_ko = conv ? (int)(conv_koff(p, u + t_first, p.lda, PQ_BK) * 2)
                            : (int)(u * xk_step) + (TN ? (int)__umulhi((uint32_t)(u + t_first), p.tn_magic) * p.tn_skip_a : 0);
      if (PQ_BUFLD) {       // measured +6..7 % over global_load_lds with 64-bit per-lane addresses (8192^3: 1317 -> 1414 TFLOP/s)
#pragma unroll
        for (int j = 0; j < 2; j++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, xo[r][j], x_ko, 0, 0);
        return;
      }
      const char* kb = xbase + x_ko;
#pragma unroll
      for (int j = 0; j < 2; j++) glds16((const bf16*)(kb + xo[r][j]), dst + j * 1024);
    } else {
      stage_ext(A2, la2, m0, true, r, (u - nt1) * PQ_BK, dst);
    }
  };
  auto stage_w = [&](int u, int r) {        // r = 0: WA, 1: WB
    char* dst = smem + (u & 1) * PQ_BUF + (2 + r) * PQ_REGION + wv * 2048;
    if (u < nt1) {
      const int w_ko = (int)(u * wk_step) + (TN ? (int)__umulhi((uint32_t)(u + t_first), p.tn_magic) * p.tn_skip_b : 0);
      if (PQ_BUFLD) {
#pragma unroll
        for (int j = 0; j < 2; j++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, wo[r][j], w_ko, 0, 0);
        return;
      }
      const char* kb = wbase + w_ko;
#pragma unroll
      for (int j = 0; j < 2; j++) glds16((const bf16*)(kb + wo[r][j]), dst + j * 1024);
    } else {
      stage_ext(B2, lb2, n0, false, r, (u - nt1) * PQ_BK, dst);
    }
  };
  // the refill of phase q = 4t+ph:  P0: WB(t+1)  P1: XB(t+1)  P2: XA(t+2)  P3: WA(t+2)
  auto refill = [&](int t, int ph, bool check) {      // check = false in the steady-state loop (every refill exists there)
    if (PQ_ABL & 2) return;
    if (ph == 0) { if (!check || t + 1 < nt) stage_w(t + 1, 1); }
    else if (ph == 1) { if (!check || t + 1 < nt) stage_x(t + 1, 1); }
    else if (ph == 2) { if (!check || t + 2 < nt) stage_x(t + 2, 0); }
    else { if (!check || t + 2 < nt) stage_w(t + 2, 0); }
  };

  // fragment addresses: region row (lane&31) + block base, 16-byte chunk (2ks+khalf) ^ ((row>>1)&7)
  const int khalf = lane >> 5;
  const int l31 = lane & 31;
  int xk[4], wk[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    const int ch = ((2 * ks + khalf) ^ ((l31 >> 1) & 7)) << 4;
    xk[ks] = (wm * 64 + l31) * 128 + ch;                       // + j*4096 (block), + region, + buffer
    wk[ks] = 2 * PQ_REGION + (wn * 32 + l31) * 128 + ch;
  }
  // TN: transposed-read bases.  group g = lane>>4, ti = (lane>>2)&3 (row inside a 4-row block), ts = lane&3 (4-column segment)
  const int tg = lane >> 4, tti = (lane >> 2) & 3, tts = lane & 3;
  const int t_row = (8 * (tg >> 1) + tti) * 256 + (tts & 1) * 8;
  const int xt = t_row + (((wm * 8 + 2 * (tg & 1) + (tts >> 1)) ^ (tti << 2)) << 4);                  // ^ (j<<6), + ks*4096 + rd*1024
  const int wt = 2 * PQ_REGION + t_row + (((wn * 4 + 2 * (tg & 1) + (tts >> 1)) ^ (tti << 2)) << 4);
  // fp8 fragments, eight 8-byte pieces per row and K-tile (128 K-elements):
  //   PQ_F8_MX = 0  v_mfma_f32_32x32x16_fp8_bf8 (issues at the bf16 rate): piece ks = the k-slice 16 ks + 8 khalf .. +8: chunk ks (swizzled), half khalf
  //   PQ_F8_MX = 1  v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 127) — the K = 64 form that issues at TWICE the bf16 rate; the row scales of
  //                 the reference's fp8-native Linear (fp8_native.py:64-75) stay in the epilogue, so MX scaling itself is not used.  A lane supplies 32 bytes
  //                 of its row per MFMA: pieces 4 s .. 4 s + 3 = bytes 64 s + 32 khalf .. +32 (two swizzled 16-byte chunks).  Both operands use the same
  //                 lane -> K mapping, and the MFMA pairs equal byte positions of the A and the B lane, so any mapping that covers the 64 K-elements of a
  //                 step once is correct: no knowledge of the instruction's internal K order is needed.
  int xk8[8], wk8[8];
#pragma unroll
  for (int ks = 0; ks < 8; ks++) {
    const int chunk = PQ_F8_MX ? 4 * (ks >> 2) + 2 * khalf + ((ks & 3) >> 1) : ks;
    const int half8 = PQ_F8_MX ? (ks & 1) : khalf;
    const int ch = ((chunk ^ ((l31 >> 1) & 7)) << 4) + half8 * 8;
    xk8[ks] = (wm * 64 + l31) * 128 + ch;
    wk8[ks] = 2 * PQ_REGION + (wn * 32 + l31) * 128 + ch;
  }
  auto ld_x8 = [&](const char* base, int j, int ks) -> long { return *(const long*)(base + xk8[ks] + j * 4096); };
  auto ld_w8 = [&](const char* base, int ks) -> long { return *(const long*)(base + wk8[ks]); };
  auto ld_x = [&](const char* base, int j, int ks) -> bf16x8 {
    if (PQ_ABL & 1) { bf16x8 z; for (int q_ = 0; q_ < 8; q_++) z[q_] = (bf16)(float)(lane + j + ks); asm volatile("" : "+v"(z)); return z; }
    if (TN) { const char* q = base + (xt ^ (j << 6)) + ks * 4096; return lds_tr16x2(q, q + 1024); }
    return *(const bf16x8*)(base + xk[ks] + j * 4096);
  };
  auto ld_w = [&](const char* base, int ks) -> bf16x8 {
    if (PQ_ABL & 1) { bf16x8 z; for (int q_ = 0; q_ < 8; q_++) z[q_] = (bf16)(float)(lane + ks); asm volatile("" : "+v"(z)); return z; }
    if (TN) { const char* q = base + wt + ks * 4096; return lds_tr16x2(q, q + 1024); }
    return *(const bf16x8*)(base + wk[ks]);
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // ---- prologue: XA(0) WA(0) WB(0) XB(0) XA(1) WA(1) in ring order; XA(0), WA(0) retired + published ----
  stage_x(0, 0); stage_w(0, 0); stage_w(0, 1); stage_x(0, 1);
  if (nt > 1) { stage_x(1, 0); stage_w(1, 0); }
  if (PQ_ABL & 2) wait_vm_rt(0); else
  wait_vm_rt(nt > 1 ? 8 : 4);
  PP_BARRIER();
  if (wm == 1) PP_BARRIER();                           // group 1 runs one barrier interval behind group 0

  bf16x8 xf[2][4], w0f[4], w1f[4];
  long xf8[2][8], w0f8[8], w1f8[8];              // fp8 fragments (the unused set is dead code for the other instantiation)
  // conv mode (the UNet's N = 320 / 640 channel counts: the last column tile is 64 / 128 of 256 wide): a wave whose 64 columns lie beyond N keeps staging and
  // meeting the barriers but issues no MFMAs — its accumulators are never stored, and under the package power cap the matrix pipe's idle share is clock
  const bool mma_live = !CONV || !p.skip_dead || (n0 + wn * 64 < N);
  TR_DECL;
#define PQ_MMA(WF, I, J0, T, PH)                                                                              \
  do {                                                                                                        \
    if (PQ_PRIO) __builtin_amdgcn_s_setprio(1);                                                               \
    if (F8 && PQ_F8_MX) {                                                                                     \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; s2++)                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                                                         \
          acc[I][J0 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag32(&WF##8[4 * s2]), frag32(&xf8[j][4 * s2]), acc[I][J0 + j], \
                                                                            0 /* A = e4m3 weights */, 1 /* B = e5m2 activations */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
    } else if (F8) {                                                                                          \
      _Pragma("unroll") for (int ks = 0; ks < 8; ks++)                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                                                         \
          acc[I][J0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_bf8(WF##8[ks], xf8[j][ks], acc[I][J0 + j], 0, 0, 0); \
    } else if (!CONV || mma_live)                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 4; ks++) {                                                        \
      _Pragma("unroll") for (int j = 0; j < 2; j++)                                                           \
        acc[I][J0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ks], xf[j][ks], acc[I][J0 + j], 0, 0, 0); \
      if (PQ_GL == 3 && ks == 1) { __builtin_amdgcn_sched_barrier(0); refill(T, PH, TAIL); __builtin_amdgcn_sched_barrier(0); } \
    }                                                                                                         \
    if (PQ_PRIO) __builtin_amdgcn_s_setprio(0);                                                               \
  } while (0)

  auto body = [&](int t, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    const char* bf = smem + (t & 1) * PQ_BUF;
    // allowed in-flight pieces at the end of each phase's load section (8 in steady state)
    const int rem = nt - 1 - t;                        // K-tiles after this one
    // ---------------- P0: WA x XA ----------------
    TR(t, 0);
    if (PQ_GL == 0) refill(t, 0, TAIL);
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int ks = 0; ks < KSN; ks++) { if (F8) xf8[j][ks] = ld_x8(bf, j, ks); else xf[j][ks & 3] = ld_x(bf, j, ks & 3); }
#pragma unroll
    for (int ks = 0; ks < KSN; ks++) { if (F8) w0f8[ks] = ld_w8(bf, ks); else w0f[ks & 3] = ld_w(bf, ks & 3); }
    if (PQ_GL == 1) refill(t, 0, TAIL);
    TR(t, 1);
    if (!TAIL) wait_vm_rt(PQ_VM); else wait_vm_rt(rem >= 2 ? 8 : (rem == 1 ? 8 : 2));
    PP_BARRIER();
    TR(t, 2);
    PQ_MMA(w0f, 0, 0, t, 1);
    if (PQ_GL == 2) refill(t, 1, TAIL);
    TR(t, 3);
    PP_BARRIER();
    // ---------------- P1: WB x XA ----------------
    if (PQ_GL == 0) refill(t, 1, TAIL);
#pragma unroll
    for (int ks = 0; ks < KSN; ks++) { if (F8) w1f8[ks] = ld_w8(bf + PQ_REGION, ks); else w1f[ks & 3] = ld_w(bf + PQ_REGION, ks & 3); }
    if (PQ_GL == 1) refill(t, 1, TAIL);
    TR(t, 4);
    if (!TAIL) wait_vm_rt(PQ_VM); else wait_vm_rt(rem >= 2 ? 8 : (rem == 1 ? 8 : 0));
    PP_BARRIER();
    TR(t, 5);
    PQ_MMA(w1f, 1, 0, t, 2);
    if (PQ_GL == 2) refill(t, 2, TAIL);
    TR(t, 6);
    PP_BARRIER();
    // ---------------- P2: WB x XB ----------------
    if (PQ_GL == 0) refill(t, 2, TAIL);
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int ks = 0; ks < KSN; ks++) { if (F8) xf8[j][ks] = ld_x8(bf + PQ_REGION, j, ks); else xf[j][ks & 3] = ld_x(bf + PQ_REGION, j, ks & 3); }
    if (PQ_GL == 1) refill(t, 2, TAIL);
    TR(t, 7);
    if (!TAIL) wait_vm_rt(PQ_VM); else wait_vm_rt(rem >= 2 ? 8 : (rem == 1 ? 6 : 0));
    PP_BARRIER();
    TR(t, 8);
    PQ_MMA(w1f, 1, 2, t, 3);
    if (PQ_GL == 2) refill(t, 3, TAIL);
    TR(t, 9);
    PP_BARRIER();
    // ---------------- P3: WA x XB ----------------
    if (PQ_GL == 0) refill(t, 3, TAIL);
    if (PQ_GL == 1) refill(t, 3, TAIL);
    if (!TAIL) wait_vm_rt(PQ_VM); else wait_vm_rt(rem >= 2 ? 8 : (rem == 1 ? 4 : 0));
    PP_BARRIER();
    TR(t, 10);
    PQ_MMA(w0f, 0, 2, t + 1, 0);
    if (PQ_GL == 2) refill(t + 1, 0, TAIL);
    TR(t, 11);
    PP_BARRIER();
  };
  if (PQ_GL == 2 || PQ_GL == 3) refill(0, 0, true);
  int t = 0;
  for (; t < nt - 2; t++) body(t, std::false_type{});
  for (; t < nt; t++) body(t, std::true_type{});
#undef PQ_MMA
  TR_FLUSH(wv, lane);
  if (wm == 0) PP_BARRIER();                           // pairs with group 1's extra barrier
  // every wave is past its last ring read and no LDS-DMA is in flight: the ring is free for the epilogue transpose
  if (SK && sk_n > 1) {
    float* slab = g.sk_ws + ((int64_t)sk_tl * (sk_n - 1)) * (PQ_BM * PQ_BN) + tid * 4;
    int* flag = g.sk_flags + sk_tl;
    if (slice < sk_n - 1) {
      slab += (int64_t)slice * (PQ_BM * PQ_BN);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int a = 0; a < 4; a++) {
            f32x4 v;
#pragma unroll
            for (int b = 0; b < 4; b++) v[b] = acc[i][j][4 * a + b];
            *(f32x4*)(slab + ((i * 4 + j) * 4 + a) * (PQ_THREADS * 4)) = v;
          }
      // The slices of a tile run on ONE XCD (block ids congruent mod 8), so writer and reader meet in the same L2: the slab stores only have to be acknowledged
      // by it (vmcnt(0): the per-CU cache is write-through) before the count, and the reader's first touch of a slab line misses its CU's cache by construction.
      // No cache-wide write-back / invalidate: the device-scope fences of the first form (buffer_wbl2 / buffer_inv sc1 from every wave of 256 workgroups) made
      // the cut round 100 us slower than the uncut one (r06, SDXL-LoRA batch 16: 173 vs 77 us per 16384 x 1280 x 1280 launch).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0)
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sk_n - 1) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    asm volatile("" ::: "memory");                       // the slab loads below are issued after the count was seen complete
    // 8 dependent round trips, each with the (i, j) quads of EVERY slab in flight (4 (SK - 1) quads per lane).  Measured r06 (SDXL-LoRA batch 16, per launch of the
    // cut shapes): one slab at a time (24 round trips at SK = 4) and an LDS-DMA landing zone (6 round trips of 16 KB per wave, nothing overlapped) were both slower.
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        f32x4 v[SK > 1 ? SK - 1 : 1][4];
#pragma unroll
        for (int k = 0; k < SK - 1; k++)
#pragma unroll
          for (int a = 0; a < 4; a++) v[k][a] = *(const f32x4*)(slab + (int64_t)k * (PQ_BM * PQ_BN) + ((i * 4 + j) * 4 + a) * (PQ_THREADS * 4));
#pragma unroll
        for (int k = 0; k < SK - 1; k++)                 // slice order
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[i][j][4 * a + b] += v[k][a][b];
        asm volatile("" ::: "memory");
      }
    if (tid == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (EPI == EPI_SPLITK) {
    GemmP ps = p;
    ps.partial = p.partial + (int64_t)slice * p.M * p.part_ld + (int64_t)wtap * p.N;
    ps.bias = nullptr;
    ps.conv_taps = 0;
    gemm_epilogue_lds<EPI, false, false>(ps, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wv * EPL_WAVE);
  } else if (wtaps > 1) {
    GemmP ps = p;
    ps.C = p.C + (int64_t)wtap * p.N;
    if (ps.aux_in) ps.aux_in = p.aux_in + (int64_t)wtap * p.N;
    ps.conv_taps = 0;
    gemm_epilogue_lds<EPI, false, false>(ps, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wv * EPL_WAVE);
  } else if (EPI == ST355_EPI_HEADS) {
    gemm_epilogue_heads(p, p.C + segi * p.seg_xc, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wv * EPL_WAVE);
  } else if (EPI == ST355_EPI_GEGLU) {
    gemm_epilogue_geglu(p, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wv * EPL_WAVE);
  } else if (EPI == ST355_EPI_GEGLU_GRAD) {
    gemm_epilogue_geglu_grad(p, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wv * EPL_WAVE);
  } else if (EPI == ST355_EPI_QK_NORM_ROPE) {
    if (n0 >= 2 * p.rH * 128) {            // v heads: rows of the V buffer (column n - 2D) in this tile's segment (+ the head-major V^T)
      gemm_epilogue_vdual(p, p.C + segi * p.seg_xc, acc, m0, n0, wm, wn, wv, lane, smem);
    } else {
      gemm_epilogue_qkrope(p, acc, m0, n0, wm, wn, wv, lane, smem);
    }
  } else if (!TN && !F8 && !CONV) {      // plain NT bf16 GEMM: the epilogue sees this tile's segment of C / aux rows (segi = 0, extras = 0: unchanged)
    GemmP ps = p;
    ps.C = p.C + segi * p.seg_xc;
    ps.aux_in = p.aux_in + segi * p.seg_xin;
    ps.aux_out = p.aux_out + segi * p.seg_xout;
    if (epl_aligned(ps)) gemm_epilogue_lds<EPI, false, false>(ps, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wv * EPL_WAVE);
    else gemm_epilogue<EPI, 2, 4>(ps, acc, m0 + wm * 128, n0 + wn * 64, lane);
  } else if (epl_aligned(p)) gemm_epilogue_lds<EPI, CONV, F8>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wv * EPL_WAVE);
  else gemm_epilogue<EPI, 2, 4>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
}

// =================================================================================================
// k_gemm_pz: the PERSISTENT form of k_gemm_pq (round 3).  One workgroup per CU walks tiles  q, q + G, q + 2G, ...  (q = XCD-remapped block id,
// G = grid size <= number of CUs), and the LDS ring simply keeps running across the tile seam: in the last two K-tiles of a tile the region
// refills address the NEXT tile's first K-tiles (same ring slots, same counted waits), so the next tile's prologue — a cold HBM round trip per
// tile in the one-tile-per-workgroup kernel — is hidden behind the current tile's tail and epilogue, and the epilogue's stores drain under the
// next tile's first K-tiles instead of under an idle CU (the workgroup does not end).  Measured in the round-2 profile: the K = 3072 shapes of
// the Flux step (48 K-tiles per tile) ran at ~1000-1060 TFLOP/s against ~1400 for K = 12288 — about 12 K-tile times of prologue + epilogue per tile.
//   * NT bf16 only, full tiles only (M % 256 == 0, N % 256 == 0), K-tiles >= 4, one problem per launch; LoRA K-extension and segmented rows supported;
//     everything else stays on k_gemm_pq (the dispatcher checks).
//   * LDS = 160 KiB: the 128-KiB ring + 32 KiB.  The epilogue transposes through 8 KiB per wave (32 tokens x 64 fp32 features, XOR-swizzled instead
//     of padded), placed where the ring is idle at the seam: the XB and WB regions of the buffer of the LAST K-tile (their next refills come after
//     the epilogue) + the spare 32 KiB.  The XA / WA regions of that buffer and the whole other buffer are already receiving the next tile.
//   * vmcnt with stores in the queue: a counted  s_waitcnt vmcnt(N)  stays CORRECT when older stores are outstanding (N counts the loads younger than
//     the one to retire; loads return in order; a store can only make the wait stricter) but it would wait for the epilogue's stores to be
//     acknowledged.  So the two regions still in flight at the seam are retired BEFORE the stores are issued (vmcnt(4)), and K-tile 0 of a
//     non-first tile skips the waits of its first two phases (nothing to retire there); the first counted wait after the stores is in phase 2.
//   * results are bit-identical to k_gemm_pq: same accumulation order, same fp32 epilogue arithmetic (tests/test_kernels_gpu.py).
// =================================================================================================
#define PZ_LDS (2 * PQ_BUF + 32768)         // 160 KiB
#define PZ_STAGE 8192                       // epilogue staging bytes per wave

// 32 tokens x 64 features (fp32) per sub-pass through an XOR-swizzled 8-KiB slice: 16-byte chunk c of token row r lives at chunk c ^ (r & 15).
// Writes (a lane owns one token of the 32 and 4 consecutive features per register quad) and the token-major read-back (8 lanes per token row,
// 8 features each) are both bank-conflict free: 16 consecutive lanes always touch 16 distinct chunks.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_lds8(const GemmP& p, f32x16 (&acc)[2][4], int mw0, int nw0, int lane, char* stage) {
  const int khalf = lane >> 5, l31 = lane & 31;
  const int rrow = lane >> 3, rc = lane & 7;
  const int n = nw0 + rc * 8;
  float bias8[8];
#pragma unroll
  for (int b = 0; b < 8; b++) bias8[b] = 0.f;
  if (p.bias) {
    const bf16x8 bv = *(const bf16x8*)(p.bias + n);
#pragma unroll
    for (int b = 0; b < 8; b++) bias8[b] = bf2f(bv[b]);
  }
  const int m_first = mw0 + rrow;
  // ST355_EPI_GEGLU_GRAD (r6, as gemm_epilogue_geglu_grad of the one-tile-per-workgroup kernel): the lane's 8 features 8 rc .. 8 rc + 7 of the wave's 64 are group
  // rc >> 2, offset 8 (rc & 3) of the interleaved pre-activation — value columns nc, gate columns nc + 32 of both the kept pre-activation and C (d value | d gate)
  constexpr bool GG = EPI == ST355_EPI_GEGLU_GRAD;
  const int nc = GG ? 2 * nw0 + 64 * (rc >> 2) + (rc & 3) * 8 : n;
  bf16* c_row = p.C + (int64_t)m_first * p.ldc + nc;
  const bf16* in_row = p.aux_in ? p.aux_in + (int64_t)m_first * p.ld_aux_in + nc : nullptr;
  bf16* out_row = p.aux_out ? p.aux_out + (int64_t)m_first * p.ld_aux_out + n : nullptr;
  const int64_t c_step = 8 * p.ldc, in_step = 8 * p.ld_aux_in, out_step = 8 * p.ld_aux_out;
  // write side: chunk (i*8 + 2a + khalf) ^ (l31 & 15) of row l31 = row base ^ ((khalf ^ (l31 & 15)) << 4) ^ ((i*8 + 2a) << 4): ONE lane-dependent base, the
  // eight positions are XORs with constants — recomputed per sub-pass (the base is made opaque there) instead of living in eight registers.
  // read side: row = it*8 + rrow, so row & 15 = rrow | 8*(it & 1): two bases (even / odd `it`) per 16-byte half, the row advance is an immediate offset.
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
  const uint32_t stage_u = (uint32_t)(uintptr_t)(lds_char*)stage;             // LDS byte address of this wave's staging slice
  const uint32_t wbase0 = stage_u + (uint32_t)(l31 * 256 + ((khalf ^ (l31 & 15)) << 4));
  const uint32_t rb = stage_u + (uint32_t)(rrow * 256);
  const uint32_t r_lo_e = rb + (uint32_t)((((2 * rc) ^ rrow)) << 4), r_hi_e = rb + (uint32_t)((((2 * rc + 1) ^ rrow)) << 4);
  const uint32_t r_lo_o = rb + (uint32_t)((((2 * rc) ^ (rrow | 8))) << 4), r_hi_o = rb + (uint32_t)((((2 * rc + 1) ^ (rrow | 8))) << 4);
  // the residual / pre-activation rows an epilogue READS are fetched ONE SUB-PASS AHEAD (32 rows = 4 x 16 B per lane, double-buffered): issued in the
  // loop below they were 16 dependent HBM round trips per tile (a C store may alias them, so the compiler cannot hoist them itself) — measured r03:
  // the x GELU' epilogue at 1183 TFLOP/s against 1355 for the plain one on the same shape.  Row r of the tile is read before row r is written
  // (in-place residual adds stay correct): a prefetched row is never one an earlier sub-pass stores.
  constexpr bool AUXIN = (EPI == ST355_EPI_GATE_RESIDUAL || EPI == ST355_EPI_ADD || EPI == ST355_EPI_MUL_GELU_GRAD || GG);
  bf16x8 auxv[2][4], gatev[2][4];
  const bf16* in_pre = in_row;
  int m_pre = m_first;
  auto preload = [&](bf16x8 (&av)[4], bf16x8 (&gv)[4]) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
      if (AUXIN) av[it] = *(const bf16x8*)in_pre;
      if (EPI == ST355_EPI_GATE_RESIDUAL) gv[it] = *(const bf16x8*)(p.gate + (int64_t)(m_pre / p.rows_per_batch) * p.gate_stride + n);
      if (GG) gv[it] = *(const bf16x8*)(in_pre + 32);
      in_pre += in_step; m_pre += 8;
    }
  };
  if (AUXIN) preload(auxv[0], gatev[0]);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t wb = wbase0;
    asm volatile("" : "+v"(wb));
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int a = 0; a < 4; a++) {
        f32x4 v;
#pragma unroll
        for (int b = 0; b < 4; b++) v[b] = acc[i][j][4 * a + b];
        *(lds_f32x4*)(wb ^ (uint32_t)((i * 8 + 2 * a) << 4)) = v;
      }
    if (AUXIN && j < 3) preload(auxv[(j + 1) & 1], gatev[(j + 1) & 1]);
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const f32x4 lo = *(const lds_f32x4*)(((it & 1) ? r_lo_o : r_lo_e) + (uint32_t)(it * 2048));
      const f32x4 hi = *(const lds_f32x4*)(((it & 1) ? r_hi_o : r_hi_e) + (uint32_t)(it * 2048));
      bf16* const c_ptr = c_row;
      bf16* const out_ptr = out_row;
      c_row += c_step; out_row += out_step;
      float v[8];
#pragma unroll
      for (int b = 0; b < 4; b++) { v[b] = lo[b] + bias8[b]; v[4 + b] = hi[b] + bias8[4 + b]; }
      if (EPI == ST355_EPI_GELU) {
        if (p.aux_out) {
          bf16x8 pre;
#pragma unroll
          for (int b = 0; b < 8; b++) pre[b] = f2bf(v[b]);
          *(bf16x8*)out_ptr = pre;
#pragma unroll
          for (int b = 0; b < 8; b++) v[b] = bf2f(pre[b]);
        }
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = gelu_tanh(v[b]);
      } else if (EPI == ST355_EPI_GATE_RESIDUAL) {
        if (p.aux_out) {
          bf16x8 yv;
#pragma unroll
          for (int b = 0; b < 8; b++) yv[b] = f2bf(v[b]);
          *(bf16x8*)out_ptr = yv;
        }
        const bf16x8 gv = gatev[j & 1][it];
        const bf16x8 rv = auxv[j & 1][it];
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = bf2f(rv[b]) + bf2f(gv[b]) * v[b];
      } else if (EPI == ST355_EPI_ADD) {
        const bf16x8 rv = auxv[j & 1][it];
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] += bf2f(rv[b]);
      } else if (EPI == ST355_EPI_MUL_GELU_GRAD) {
        const bf16x8 hv = auxv[j & 1][it];
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] *= gelu_tanh_grad(bf2f(hv[b]));
      }
      if (GG) {
        const bf16x8 vv = auxv[j & 1][it], gv = gatev[j & 1][it];
        bf16x8 dv, dg;
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const float d = bf2f(f2bf(v[b]));                 // d out rounded to bf16, as the unfused pair (GEMM store, st355_geglu_bwd read)
          const float g = bf2f(gv[b]);
          dv[b] = f2bf(d * gelu_erf(g));
          dg[b] = f2bf(d * bf2f(vv[b]) * gelu_erf_grad(g));
        }
        *(bf16x8*)c_ptr = dv;
        *(bf16x8*)(c_ptr + 32) = dg;
        continue;
      }
      bf16x8 o;
#pragma unroll
      for (int b = 0; b < 8; b++) o[b] = f2bf(v[b]);
      *(bf16x8*)c_ptr = o;
    }
  }
}

template <int EPI>
__global__ void __launch_bounds__(PQ_THREADS, 2) k_gemm_pz(GemmP p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int nbm = p.M / PQ_BM, nbn = p.N / PQ_BN;
  const int nt1 = p.K / PQ_BK;
  const int nt = nt1 + p.K2 / PQ_BK;
  const int G = gridDim.x;
  const uint32_t lda_b = (uint32_t)p.lda * 2, ldb_b = (uint32_t)p.ldb * 2;
  const int64_t la2 = p.lda2, lb2 = p.ldb2;

  // tile -> scalar bases
  struct Tile { int m0, n0; const char* xb; const char* wb; const bf16* a2; const bf16* b2; int64_t segi; };
  auto tile_at = [&](int id) {
    Tile t;
    int pm, pn;
    tile_coords(id, nbm, nbn, pm, pn);
    t.m0 = __builtin_amdgcn_readfirstlane(pm * PQ_BM);
    t.n0 = __builtin_amdgcn_readfirstlane(pn * PQ_BN);
    t.segi = p.seg_rows ? (int64_t)__builtin_amdgcn_readfirstlane(t.m0 / p.seg_rows) : 0;
    t.xb = (const char*)(p.A + t.segi * p.seg_xa) + (int64_t)t.m0 * lda_b;
    t.wb = (const char*)p.B + (int64_t)t.n0 * ldb_b;
    t.a2 = p.A2 + t.segi * p.seg_xa2 + (int64_t)t.m0 * la2;
    t.b2 = p.B2 + (int64_t)t.n0 * lb2;
    return t;
  };

  int tile_id = xcd_remap(blockIdx.x, G);
  Tile cur = tile_at(tile_id);
  int par = 0;                                   // ring parity of this tile's K-tile 0
  bool first = true;

  while (true) {
    // per-lane addressing state is REBUILT per tile from an opaque copy of the lane id: nothing of it stays live across the epilogue (kept live, it and the
    // epilogue's hoisted address arithmetic pushed the accumulators into scratch — and every scratch reload waits vmcnt(0), i.e. for the epilogue's stores)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    uint32_t xo[2][2], wo[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int lr = (wv * 2 + j) * 8 + (ln >> 3);
      const uint32_t scb = (uint32_t)(((ln & 7) ^ ((lr >> 1) & 7)) * 16);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        xo[r][j] = (uint32_t)((lr >> 6) * 128 + (lr & 63) + r * 64) * lda_b + scb;
        wo[r][j] = (uint32_t)((lr >> 5) * 64 + (lr & 31) + r * 32) * ldb_b + scb;
      }
    }
    const int khalf = ln >> 5;
    const int l31 = ln & 31;
    int xk[4], wk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int ch = ((2 * ks + khalf) ^ ((l31 >> 1) & 7)) << 4;
      xk[ks] = (wm * 64 + l31) * 128 + ch;
      wk[ks] = 2 * PQ_REGION + (wn * 32 + l31) * 128 + ch;
    }
    auto ld_x = [&](const char* base, int j, int ks) -> bf16x8 { return *(const bf16x8*)(base + xk[ks] + j * 4096); };
    auto ld_w = [&](const char* base, int ks) -> bf16x8 { return *(const bf16x8*)(base + wk[ks]); };
    const bool has_next = tile_id + G < ntiles;
    const Tile nxt = has_next ? tile_at(tile_id + G) : cur;
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)cur.xb, 0, 0x7FFFFFFF, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)cur.wb, 0, 0x7FFFFFFF, 0x00020000);
    const auto xn_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)nxt.xb, 0, 0x7FFFFFFF, 0x00020000);
    const auto wn_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)nxt.wb, 0, 0x7FFFFFFF, 0x00020000);

    // region refill of ring K-tile u (u counts from this tile's K-tile 0; u >= nt: K-tile u - nt of the next tile)
    auto stage_x = [&](int u, int r, int mode) {
      char* dst = smem + (((u & 1) ^ par) * PQ_BUF) + r * PQ_REGION + wv * 2048;
      const bool nx = mode == 1 && u >= nt;
      const int uu = nx ? u - nt : u;
      if (uu < nt1) {
#pragma unroll
        for (int j = 0; j < 2; j++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(nx ? xn_rsrc : x_rsrc, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, xo[r][j], uu * (PQ_BK * 2), 0, 0);
      } else {
        const bf16* base = nx ? nxt.a2 : cur.a2;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int lr = (wv * 2 + j) * 8 + (ln >> 3);
          glds16(base + (int64_t)((lr >> 6) * 128 + (lr & 63) + r * 64) * la2 + ((ln & 7) ^ ((lr >> 1) & 7)) * 8 + (uu - nt1) * PQ_BK, dst + j * 1024);
        }
      }
    };
    auto stage_w = [&](int u, int r, int mode) {
      char* dst = smem + (((u & 1) ^ par) * PQ_BUF) + (2 + r) * PQ_REGION + wv * 2048;
      const bool nx = mode == 1 && u >= nt;
      const int uu = nx ? u - nt : u;
      if (uu < nt1) {
#pragma unroll
        for (int j = 0; j < 2; j++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(nx ? wn_rsrc : w_rsrc, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, wo[r][j], uu * (PQ_BK * 2), 0, 0);
      } else {
        const bf16* base = nx ? nxt.b2 : cur.b2;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int lr = (wv * 2 + j) * 8 + (ln >> 3);
          glds16(base + (int64_t)((lr >> 5) * 64 + (lr & 31) + r * 32) * lb2 + ((ln & 7) ^ ((lr >> 1) & 7)) * 8 + (uu - nt1) * PQ_BK, dst + j * 1024);
        }
      }
    };
    // MODE 0: steady state (every refill exists, in this tile).  MODE 1: the last two K-tiles of a tile that HAS a successor (refills beyond nt address
    // the next tile; waits as in steady state).  MODE 2: the last two K-tiles of the workgroup's last tile (refills beyond nt skipped, waits shortened).
    auto refill = [&](int t, int ph, int mode) {
      const int u = (ph < 2) ? t + 1 : t + 2;
      if (mode == 2 && u >= nt) return;
      if (ph == 0) stage_w(u, 1, mode);
      else if (ph == 1) stage_x(u, 1, mode);
      else if (ph == 2) stage_x(u, 0, mode);
      else stage_w(u, 0, mode);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    if (first) {
      // prologue of the workgroup's first tile (as k_gemm_pq): XA(0) WA(0) WB(0) XB(0) XA(1) WA(1) in ring order; XA(0), WA(0) retired + published
      stage_x(0, 0, 0); stage_w(0, 0, 0); stage_w(0, 1, 0); stage_x(0, 1, 0);
      stage_x(1, 0, 0); stage_w(1, 0, 0);
      wait_vm_rt(8);
      PP_BARRIER();
    }
    if (wm == 1) PP_BARRIER();                         // group 1 runs one barrier interval behind group 0

    bf16x8 xf[2][4], w0f[4], w1f[4];
#define PZ_MMA(WF, I, J0)                                                                                     \
  do {                                                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ks++)                                                          \
      _Pragma("unroll") for (int j = 0; j < 2; j++)                                                           \
        acc[I][J0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ks], xf[j][ks], acc[I][J0 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                            \
  } while (0)

    // FIRSTK: K-tile 0 of a non-first tile — WB(0) / XB(0) were retired at the seam, before the epilogue's stores entered the queue: no wait in P0 / P1
    auto body = [&](int t, auto mode_c, auto firstk_c) {
      constexpr int MODE = decltype(mode_c)::value;
      constexpr bool FIRSTK = decltype(firstk_c)::value;
      const char* bf = smem + (((t & 1) ^ par) * PQ_BUF);
      const int rem = nt - 1 - t;
      // ---------------- P0: WA x XA ----------------
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) xf[j][ks] = ld_x(bf, j, ks);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) w0f[ks] = ld_w(bf, ks);
      refill(t, 0, MODE);
      if (MODE != 2) { if (!FIRSTK) wait_vm_rt(8); } else wait_vm_rt(rem >= 1 ? 8 : 2);
      PP_BARRIER();
      PZ_MMA(w0f, 0, 0);
      PP_BARRIER();
      // ---------------- P1: WB x XA ----------------
#pragma unroll
      for (int ks = 0; ks < 4; ks++) w1f[ks] = ld_w(bf + PQ_REGION, ks);
      refill(t, 1, MODE);
      if (MODE != 2) { if (!FIRSTK) wait_vm_rt(8); } else wait_vm_rt(rem >= 1 ? 8 : 0);
      PP_BARRIER();
      PZ_MMA(w1f, 1, 0);
      PP_BARRIER();
      // ---------------- P2: WB x XB ----------------
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) xf[j][ks] = ld_x(bf + PQ_REGION, j, ks);
      refill(t, 2, MODE);
      if (MODE != 2) wait_vm_rt(8); else wait_vm_rt(rem >= 1 ? 6 : 0);
      PP_BARRIER();
      PZ_MMA(w1f, 1, 2);
      PP_BARRIER();
      // ---------------- P3: WA x XB ----------------
      refill(t, 3, MODE);
      if (MODE != 2) wait_vm_rt(8); else wait_vm_rt(rem >= 1 ? 4 : 0);
      PP_BARRIER();
      PZ_MMA(w0f, 0, 2);
      PP_BARRIER();
    };
    int t = 0;
    if (!first) { body(0, std::integral_constant<int, 0>{}, std::true_type{}); t = 1; }
    for (; t < nt - 2; t++) body(t, std::integral_constant<int, 0>{}, std::false_type{});
    // the workgroup's LAST tile prefetches "its successor" too (nxt = cur: two K-tiles of valid, unused operand rows): one tail for every tile keeps
    // the control flow — and the register allocation at the epilogue — single-path (a separate no-prefetch tail cost ~90 spilled accumulators)
    for (; t < nt; t++) body(t, std::integral_constant<int, 1>{}, std::false_type{});
#undef PZ_MMA
    if (wm == 0) PP_BARRIER();                         // pairs with group 1's extra barrier: every wave is past its last ring read of this tile
    // in flight now (has_next): WB(n0) XB(n0) [issued in the last K-tile's P0 / P1] and XA(n1) WA(n1) [P2 / P3]; retire the first two BEFORE any store
    wait_vm_rt(4);
    {
      GemmP ps = p;
      ps.C = p.C + cur.segi * p.seg_xc;
      ps.aux_in = p.aux_in ? p.aux_in + cur.segi * p.seg_xin : nullptr;
      ps.aux_out = p.aux_out ? p.aux_out + cur.segi * p.seg_xout : nullptr;
      // staging: the XB / WB regions of the LAST K-tile's buffer (refilled only after the epilogue) and the spare 32 KiB above the ring
      char* const lastbuf = smem + ((((nt - 1) & 1) ^ par) * PQ_BUF);
      char* const stg = wv < 2 ? lastbuf + PQ_REGION + wv * PZ_STAGE : (wv < 4 ? lastbuf + 3 * PQ_REGION + (wv - 2) * PZ_STAGE : smem + 2 * PQ_BUF + (wv - 4) * PZ_STAGE);
      int le = lane;
      asm volatile("" : "+v"(le));           // the epilogue's address arithmetic starts here, not above the K loop
      gemm_epilogue_lds8<EPI>(ps, acc, cur.m0 + wm * 128, cur.n0 + wn * 64, le, stg);
    }
    if (!has_next) { wait_vm_rt(0); break; }           // (the dummy prefetch of the last tile lands before the wave ends)
    PP_BARRIER();                                      // every wave has read its staging slice back: the ring slots may be refilled
    par ^= (nt & 1);
    tile_id += G;
    cur = nxt;
    first = false;
  }
}

// =================================================================================================
// k_gemm_s2: 128x128x64, 4 waves, double buffer (small problems)
// =================================================================================================
#define S2_BM 128
#define S2_BN 128
#define S2_THREADS 256
#define S2_TILE (128 * BK * 2)
#define S2_STAGE (2 * S2_TILE)
#define S2_LDS (2 * S2_STAGE)

template <int EPI>
__global__ void __launch_bounds__(S2_THREADS, 2) k_gemm_s2(GemmP p_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int nbm = (p_in.M + S2_BM - 1) / S2_BM, nbn = (p_in.N + S2_BN - 1) / S2_BN;
  int pm, pn;
  // split-K: blockIdx.x = slice * tiles + tile; slice s owns K-tiles [s*per, min(nt1, (s+1)*per))  (no K2 segment when splitting)
  const int tiles = nbm * nbn;
  const int slice = (EPI == EPI_SPLITK) ? blockIdx.x / tiles : 0;
  tile_coords(xcd_remap((EPI == EPI_SPLITK) ? blockIdx.x % tiles : blockIdx.x, tiles), nbm, nbn, pm, pn);
  const int m0 = pm * S2_BM, n0 = pn * S2_BN;
  const GemmP p = seg_view(p_in, m0);
  const int nt1 = p.K / BK;
  const int per = (EPI == EPI_SPLITK) ? (nt1 + p.ksplit - 1) / p.ksplit : 0;
  const int t_first = (EPI == EPI_SPLITK) ? slice * per : 0;
  const int nt = (EPI == EPI_SPLITK) ? min(nt1, t_first + per) : nt1 + p.K2 / BK;
  const int st_row = lane >> 3, st_cp = lane & 7;
  auto stage = [&](int t, int buf) {
    const bf16* Ap; const bf16* Bp; int64_t la, lb; int k0;
    if (t < nt1) { Ap = p.A; la = p.lda; Bp = p.B; lb = p.ldb; k0 = t * BK; }
    else { Ap = p.A2; la = p.lda2; Bp = p.B2; lb = p.ldb2; k0 = (t - nt1) * BK; }
    const int64_t ka = (p.conv_taps && t < nt1) ? conv_koff(p, t, la, BK) : (int64_t)k0;
    char* xs = smem + buf * S2_STAGE;
    char* ws = xs + S2_TILE;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int r0 = (wv * 4 + j) * 8;
      const int row = r0 + st_row;
      const int c = st_cp ^ ((row >> 1) & 7);
      glds16(Ap + (int64_t)min(m0 + row, p.M - 1) * la + ka + c * 8, xs + r0 * 128);
      glds16(Bp + (int64_t)min(n0 + row, p.N - 1) * lb + k0 + c * 8, ws + r0 * 128);
    }
  };
  int w_off[2], w_sw[2], x_off[2], x_sw[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int wr = wn * 64 + i * 32 + (lane & 31);
    w_off[i] = wr * 128; w_sw[i] = (wr >> 1) & 7;
    const int xr = wm * 64 + i * 32 + (lane & 31);
    x_off[i] = xr * 128; x_sw[i] = (xr >> 1) & 7;
  }
  const int khalf = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  if (t_first < nt) stage(t_first, 0);
  __syncthreads();
  for (int t = t_first; t < nt; t++) {
    const int buf = (t - t_first) & 1;
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    const char* xs = smem + buf * S2_STAGE;
    mma_tile(xs, xs + S2_TILE, x_off, x_sw, w_off, w_sw, khalf, acc);
    __syncthreads();
  }
  if (EPI == EPI_SPLITK) {
    // fp32 slab of this K slice: same lane -> (token, 4 consecutive features) map as gemm_epilogue, 16-byte stores
    float* slab = p.partial + (int64_t)slice * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int m = m0 + wm * 64 + j * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          const int n = n0 + wn * 64 + i * 32 + 8 * a + 4 * khalf;
          if (n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][j][4 * a + b];
          *(f32x4*)(slab + (int64_t)m * p.N + n) = v;
        }
    }
  } else {
    gemm_epilogue<(EPI == EPI_SPLITK) ? ST355_EPI_NONE : EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
  }
}

// fixed-order sum of the K-slice slabs (+ bias) -> bf16 C
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ slabs, int ksplit, const bf16* __restrict__ bias,
                                                      bf16* __restrict__ C, int64_t ldc, int M, int N, int accumulate) {
  const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread = 4 consecutive features
  const int n4 = N / 4;
  if (i4 >= (int64_t)M * n4) return;
  const int m = (int)(i4 / n4), n = (int)(i4 % n4) * 4;
  f32x4 s = *(const f32x4*)(slabs + (int64_t)m * N + n);
  for (int k = 1; k < ksplit; k++) {
    const f32x4 v = *(const f32x4*)(slabs + ((int64_t)k * M + m) * N + n);
    s += v;
  }
  bf16x4 o;
#pragma unroll
  for (int b = 0; b < 4; b++) o[b] = f2bf(s[b] + (bias ? bf2f(bias[n + b]) : 0.f));
  if (accumulate) {
    const bf16x4 c0 = *(const bf16x4*)(C + (int64_t)m * ldc + n);
#pragma unroll
    for (int b = 0; b < 4; b++) o[b] = f2bf(s[b] + bf2f(c0[b]));
  }
  *(bf16x4*)(C + (int64_t)m * ldc + n) = o;
}

// =================================================================================================
// host side
// =================================================================================================
static int validate(const st355_gemm_args* a) {
  ST_REQUIRE(a && a->A && a->B && a->C, "gemm: null pointer");
  ST_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "gemm: empty shape M=%d N=%d K=%d", a->M, a->N, a->K);
  ST_REQUIRE(a->K % BK == 0 && a->K2 % BK == 0, "gemm: K (%d) and K2 (%d) must be multiples of 64", a->K, a->K2);  // (the BK=32 schedule needs only 32)
  ST_REQUIRE(a->N % 4 == 0 && a->ldc % 4 == 0, "gemm: N (%d) and ldc must be multiples of 4", a->N);
  ST_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 (16-byte rows)");
  ST_REQUIRE(((uintptr_t)a->A % 16 == 0) && ((uintptr_t)a->B % 16 == 0) && ((uintptr_t)a->C % 8 == 0), "gemm: misaligned pointer");
  if (a->K2 > 0) {
    ST_REQUIRE(a->A2 && a->B2 && a->lda2 % 8 == 0 && a->ldb2 % 8 == 0, "gemm: low-rank extension operands missing/misaligned");
    ST_REQUIRE(((uintptr_t)a->A2 % 16 == 0) && ((uintptr_t)a->B2 % 16 == 0), "gemm: misaligned A2/B2");
  }
  if (a->epilogue == ST355_EPI_GATE_RESIDUAL)
    ST_REQUIRE(a->gate && a->aux_in && a->rows_per_batch > 0 && a->gate_stride % 4 == 0 && a->ld_aux_in % 4 == 0,
               "gemm: gate/residual epilogue operands missing");
  if (a->epilogue == ST355_EPI_MUL_GELU_GRAD || a->epilogue == ST355_EPI_ADD)
    ST_REQUIRE(a->aux_in && a->ld_aux_in % 4 == 0, "gemm: gelu-grad/add epilogue needs aux_in");
  if ((a->epilogue == ST355_EPI_GELU || a->epilogue == ST355_EPI_GATE_RESIDUAL) && a->aux_out)
    ST_REQUIRE(a->ld_aux_out % 4 == 0, "gemm: ld_aux_out must be a multiple of 4");
  ST_REQUIRE(a->epilogue >= 0 && a->epilogue <= ST355_EPI_HEADS, "gemm: unknown epilogue %d", a->epilogue);
  if (a->epilogue == ST355_EPI_HEADS) {
    const st355_heads* h = a->heads;
    ST_REQUIRE(h && h->H > 0 && h->S > 0 && h->n_q >= 0 && h->n_k >= 0 && h->n_q % 64 == 0 && h->n_k % 64 == 0 && a->N % 64 == 0 && h->n_q + h->n_k <= a->N,
               "gemm: EPI_HEADS needs args->heads with 64-column head blocks (N=%d)", a->N);
    const int n_v = a->N - h->n_q - h->n_k;
    ST_REQUIRE((h->n_q == 0 || (h->Q && h->n_q == h->H * 64)) && (h->n_k == 0 || (h->K && h->n_k == h->H * 64)) && (n_v == 0 || n_v == h->H * 64),
               "gemm: EPI_HEADS: every present part (q / k / v) is H heads of 64 columns");
    ST_REQUIRE(a->rows_per_batch > 0 && a->M % a->rows_per_batch == 0 && h->pos0 >= 0 && h->pos0 + a->rows_per_batch <= h->S, "gemm: EPI_HEADS rows_per_batch / pos0 / S");
    ST_REQUIRE(!a->seg_rows && a->K2 >= 0 && a->ldc % 8 == 0 && ((uintptr_t)a->C % 16 == 0) && (!h->Q || (uintptr_t)h->Q % 16 == 0) && (!h->K || (uintptr_t)h->K % 16 == 0) &&
               (!a->bias || (uintptr_t)a->bias % 16 == 0), "gemm: EPI_HEADS operands must be 16-byte aligned, no segmented rows");
    ST_REQUIRE(!h->Vt || (n_v > 0 && a->rows_per_batch % 8 == 0 && h->pos0 % 8 == 0 && h->Sp % 8 == 0 && h->Sp >= h->S && ((uintptr_t)h->Vt % 16 == 0)),
               "gemm: EPI_HEADS V^T needs v heads, rows_per_batch / pos0 / Sp multiples of 8, Sp >= S");
  }
  if (a->epilogue == ST355_EPI_GEGLU || a->epilogue == ST355_EPI_GEGLU_GRAD) {
    const bool fwd = a->epilogue == ST355_EPI_GEGLU;
    ST_REQUIRE(a->N % 64 == 0 && a->K2 == 0 && !a->seg_rows && !a->gate, "gemm: the GEGLU epilogues take a plain problem with N %% 64 == 0 (N=%d)", a->N);
    ST_REQUIRE(a->ldc % 8 == 0 && ((uintptr_t)a->C % 16 == 0) && a->ldc >= (fwd ? a->N / 2 : 2 * a->N), "gemm: GEGLU C rows: 16-byte aligned, ldc >= %s", fwd ? "N / 2" : "2 N");
    if (fwd) ST_REQUIRE(a->aux_out && a->ld_aux_out % 8 == 0 && a->ld_aux_out >= a->N && ((uintptr_t)a->aux_out % 16 == 0) && (!a->bias || (uintptr_t)a->bias % 16 == 0),
                        "gemm: EPI_GEGLU keeps the interleaved pre-activation in aux_out [M, N] (16-byte aligned rows)");
    else ST_REQUIRE(a->aux_in && a->ld_aux_in % 8 == 0 && a->ld_aux_in >= 2 * a->N && ((uintptr_t)a->aux_in % 16 == 0) && !a->bias,
                    "gemm: EPI_GEGLU_GRAD reads the interleaved pre-activation from aux_in [M, 2 N] (16-byte aligned rows); no bias");
  }
  if (a->epilogue == ST355_EPI_QK_NORM_ROPE) {
    const st355_qk_rope* r = a->rope;
    ST_REQUIRE(r && r->Q && r->K && r->rrms && r->cos && r->sin, "gemm: EPI_QK_NORM_ROPE needs args->rope with Q, K, rrms, cos, sin");
    ST_REQUIRE(r->H > 0 && r->H % 2 == 0 && a->N == 3 * r->H * 128, "gemm: EPI_QK_NORM_ROPE is built for head_dim 128, even H, N = 3*H*128 (N=%d H=%d)", a->N, r->H);
    ST_REQUIRE(a->rows_per_batch > 0 && a->rows_per_batch % 256 == 0 && a->M % a->rows_per_batch == 0 && r->pos0 >= 0 && r->pos0 + a->rows_per_batch <= r->S,
               "gemm: EPI_QK_NORM_ROPE rows_per_batch (%lld) must be a multiple of 256 dividing M, pos0 + rows_per_batch <= S", (long long)a->rows_per_batch);
    ST_REQUIRE(a->ldc % 8 == 0 && ((uintptr_t)a->C % 16 == 0) && ((uintptr_t)r->Q % 16 == 0) && ((uintptr_t)r->K % 16 == 0) && ((uintptr_t)r->cos % 16 == 0) &&
               ((uintptr_t)r->sin % 16 == 0) && (!a->bias || (uintptr_t)a->bias % 16 == 0), "gemm: EPI_QK_NORM_ROPE operands must be 16-byte aligned");
    ST_REQUIRE(!a->seg_rows || a->seg_rows == a->rows_per_batch, "gemm: EPI_QK_NORM_ROPE segments are the per-sample row blocks (seg_rows == rows_per_batch)");
    ST_REQUIRE(!r->Vt || (r->Sp >= r->S && r->Sp % 8 == 0 && r->pos0 % 8 == 0 && ((uintptr_t)r->Vt % 16 == 0)), "gemm: EPI_QK_NORM_ROPE V^T needs Sp >= S, Sp and pos0 multiples of 8");
  }
  if (a->seg_rows) {
    ST_REQUIRE(a->seg_rows > 0 && a->seg_rows % 256 == 0 && a->M % a->seg_rows == 0, "gemm: seg_rows (%lld) must be a multiple of 256 that divides M (%d)",
               (long long)a->seg_rows, a->M);
    ST_REQUIRE(a->seg_a >= 0 && a->seg_a2 >= 0 && a->seg_c >= 0 && a->seg_in >= 0 && a->seg_out >= 0, "gemm: negative segment stride");
    ST_REQUIRE((a->seg_a == 0 || a->seg_a >= a->seg_rows) && (a->seg_c == 0 || a->seg_c >= a->seg_rows), "gemm: segment stride smaller than seg_rows");
    ST_REQUIRE(a->K2 <= 0 || a->seg_a2 == 0 || a->seg_a2 >= a->seg_rows, "gemm: A2 segment stride smaller than seg_rows");
    ST_REQUIRE(!a->aux_in || a->seg_in == 0 || a->seg_in >= a->seg_rows, "gemm: aux_in segment stride smaller than seg_rows");
    ST_REQUIRE(!a->aux_out || a->seg_out == 0 || a->seg_out >= a->seg_rows, "gemm: aux_out segment stride smaller than seg_rows");
  }
  ST_REQUIRE(256 * (a->lda > a->ldb ? a->lda : a->ldb) * 2 + (int64_t)a->K * 2 < ((int64_t)1 << 31), "gemm: a 256-row tile must fit 32-bit buffer offsets");
  return ST355_OK;
}

static GemmP to_p(const st355_gemm_args* a) {
  GemmP p;
  p.A = (const bf16*)a->A; p.lda = a->lda; p.B = (const bf16*)a->B; p.ldb = a->ldb;
  p.A2 = (const bf16*)a->A2; p.lda2 = a->lda2; p.B2 = (const bf16*)a->B2; p.ldb2 = a->ldb2;
  p.C = (bf16*)a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.K2 = a->K2;
  p.bias = (const bf16*)a->bias;
  p.aux_out = (bf16*)a->aux_out; p.ld_aux_out = a->ld_aux_out;
  p.aux_in = (const bf16*)a->aux_in; p.ld_aux_in = a->ld_aux_in;
  p.gate = (const bf16*)a->gate; p.gate_stride = a->gate_stride; p.rows_per_batch = a->rows_per_batch;
  p.partial = nullptr; p.ksplit = 1; p.part_ld = a->N;
  p.scale_a = nullptr; p.scale_b = nullptr;
  p.conv_taps = 0; p.conv_tpt = 1; p.conv_wp = 0; p.conv_hp = 0; p.conv_row0 = 0; p.img_add = nullptr; p.img_add_stride = 0;
  if (a->seg_rows) {
    auto extra = [&](int64_t stride, int64_t ld) { return stride ? (stride - a->seg_rows) * ld : (int64_t)0; };
    p.seg_rows = (int)a->seg_rows;
    p.seg_xa = extra(a->seg_a, a->lda); p.seg_xa2 = (a->A2 && a->K2) ? extra(a->seg_a2, a->lda2) : 0; p.seg_xc = extra(a->seg_c, a->ldc);
    p.seg_xin = a->aux_in ? extra(a->seg_in, a->ld_aux_in) : 0; p.seg_xout = a->aux_out ? extra(a->seg_out, a->ld_aux_out) : 0;
  }
  if (a->epilogue == ST355_EPI_HEADS && a->heads) {
    const st355_heads* h = a->heads;
    p.rq = (bf16*)h->Q; p.rk = (bf16*)h->K; p.rvt = (bf16*)h->Vt; p.rH = h->H; p.rS = h->S; p.rpos0 = h->pos0; p.rSp = h->Sp; p.hq = h->n_q; p.hk = h->n_k;
    p.rows_per_batch = a->rows_per_batch;
  }
  if (a->epilogue == ST355_EPI_QK_NORM_ROPE && a->rope) {
    const st355_qk_rope* r = a->rope;
    p.rq = (bf16*)r->Q; p.rk = (bf16*)r->K; p.rrms = r->rrms; p.rwq = (const bf16*)r->wq; p.rwk = (const bf16*)r->wk;
    p.rcos = r->cos; p.rsin = r->sin; p.rH = r->H; p.rS = r->S; p.rpos0 = r->pos0; p.reps = r->eps; p.rvt = (bf16*)r->Vt; p.rSp = r->Sp;
    p.rows_per_batch = a->rows_per_batch;
  }
  return p;
}

// algorithmic work: the zero-padded columns of the LoRA K-extension (rank 32 -> 64-column granule) are executed but are not work
static double k2_alg(const st355_gemm_args* a) { return (a->K2_real > 0 && a->K2_real < a->K2) ? a->K2_real : a->K2; }
static double gemm_flops(const st355_gemm_args* a) { return 2.0 * a->M * a->N * ((double)a->K + k2_alg(a)); }
static double gemm_bytes(const st355_gemm_args* a) {
  return 2.0 * ((double)a->M * (a->K + k2_alg(a)) + (double)a->N * (a->K + k2_alg(a)) + (double)a->M * a->N);
}

template <int EPI>
static int launch_s2(void* stream, const GemmP& p) {
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_s2<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS); }
  const int nbm = (p.M + S2_BM - 1) / S2_BM, nbn = (p.N + S2_BN - 1) / S2_BN;
  hipLaunchKernelGGL(k_gemm_s2<EPI>, dim3(nbm * nbn), dim3(S2_THREADS), S2_LDS, (hipStream_t)stream, p);
  return st355_check_launch("gemm_s2");
}

template <int EPI>
static int launch_p3(void* stream, const GemmGroup& g, int tiles) {
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_p3<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, P3_LDS); }
  hipLaunchKernelGGL(k_gemm_p3<EPI>, dim3(tiles), dim3(P3_THREADS), P3_LDS, (hipStream_t)stream, g);
  return st355_check_launch("gemm_p3");
}

template <int EPI>
static int launch_pq(void* stream, const GemmGroup& g, int tiles) {
  constexpr int lds = PQ_LDS + (EPI == ST355_EPI_QK_NORM_ROPE ? QKR_XCHG : 0);        // + the half-head sum exchange of the fused q/k epilogue
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_pq<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
  hipLaunchKernelGGL((k_gemm_pq<EPI, false>), dim3(tiles), dim3(PQ_THREADS), lds, (hipStream_t)stream, g);
  return st355_check_launch("gemm_pq");
}
template <int EPI, int S>
static int launch_pq_sk_s(void* stream, const GemmGroup& g, int blocks) {
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_pq<EPI, false, false, false, S>, hipFuncAttributeMaxDynamicSharedMemorySize, PQ_LDS); }
  hipLaunchKernelGGL((k_gemm_pq<EPI, false, false, false, S>), dim3(blocks), dim3(PQ_THREADS), PQ_LDS, (hipStream_t)stream, g);
  return st355_check_launch("gemm_pq_sk");
}
template <int EPI>
static int launch_pq_sk(void* stream, const GemmGroup& g, int blocks) {
  return g.sk_s == 2 ? launch_pq_sk_s<EPI, 2>(stream, g, blocks) : g.sk_s == 3 ? launch_pq_sk_s<EPI, 3>(stream, g, blocks) : launch_pq_sk_s<EPI, 4>(stream, g, blocks);
}
template <int EPI>
static int launch_pq_conv(void* stream, const GemmGroup& g, int tiles) {
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_pq<EPI, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PQ_LDS); }
  hipLaunchKernelGGL((k_gemm_pq<EPI, false, false, true>), dim3(tiles), dim3(PQ_THREADS), PQ_LDS, (hipStream_t)stream, g);
  return st355_check_launch("gemm_pq_conv");
}
static int p4_tiles(const GemmP& p) { return ((p.M + PQ_BM - 1) / PQ_BM) * ((p.N + PQ_BN - 1) / PQ_BN); }   // 256x256 tiles

static int p3_tiles(const GemmP& p) { return ((p.M + P3_BM - 1) / P3_BM) * ((p.N + P3_BN - 1) / P3_BN); }

static int gemm_impl_choice() {
  static int c = -1;
  if (c < 0) {
    // A/B testing: ST355_GEMM_IMPL = s2 (128x128 only) | p3 (+256x128 ring) | pq (default: + 256x256 BK=64 ping-pong).  The two earlier 256x256
    // schedules (p4 lock-step ring, pp BK=32 ping-pong) were retired in round 2: pq superseded both on every shape.
    const char* e = getenv("ST355_GEMM_IMPL");
    c = 4;
    if (e && e[0] == 's') c = 0;
    else if (e && e[0] == 'p' && e[1] == '3') c = 1;
  }
  return c;
}

// the 256x256 schedules run one workgroup per CU: only worth it when the tiles (nearly) fill the 256 CUs
static int min_tiles_256() {
  static int v = -1;
  // measured r3 (same box, SDXL-LoRA batch 4 graph / SD3 full fine-tune): 200 -> 128 tiles: 162.9 -> 160.0 ms and 337.0 -> 336.0 ms; 64: 164.6 / 336.2
  if (v < 0) { const char* e = getenv("ST355_GEMM_MIN_TILES"); v = e ? atoi(e) : 128; }
  return v;
}

static int g_persist_override = -1;          // lab / test hook (tools/gemm_lab.hip compiles this file in): 0 / 1 force the choice, -1 = environment
static int persist_enabled() {
  if (g_persist_override >= 0) return g_persist_override;
  static int v = -1;
  if (v < 0) { const char* e = getenv("ST355_GEMM_PERSIST"); v = (e && e[0] == '0') ? 0 : 1; }      // A/B: 0 = one tile per workgroup (k_gemm_pq) everywhere
  return v;
}
static int device_cus() {
  static int v = -1;
  if (v < 0) {
    int dev = 0; hipGetDevice(&dev);
    hipDeviceProp_t pr;
    v = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    if (const char* e = getenv("ST355_GEMM_PERSIST_WGS")) { const int o = atoi(e); if (o > 0) v = o; }
  }
  return v;
}
// the persistent kernel covers the plain NT bf16 problems made of full 256x256 tiles (k_gemm_pz header); everything else keeps k_gemm_pq
static bool pz_ok(const GemmP& p, int tiles) {
  if (!persist_enabled() || p.M % PQ_BM || p.N % PQ_BN || p.K % PQ_BK || p.K2 % PQ_BK) return false;
  if (p.K / PQ_BK + p.K2 / PQ_BK < 4 || p.K / PQ_BK < 2 || p.conv_taps || p.scale_b || p.partial || p.img_add) return false;
  if (tiles <= device_cus()) return false;            // one round of tiles: nothing to overlap
  // long plain-K problems (the feed-forward down projections, K = 12288 with no K-extension): 96+ k-iterations amortise the one-tile-per-workgroup
  // seam already and the dynamic workgroup dispatch balances the last round better — measured in-step (profiles/archive/r03_flux_step_gemm_persistent_ab.txt):
  // 1327 -> 1266 TFLOP/s under the persistent schedule, every other shape class +0.5 ... +8 %
  // (r5 A/B: a long contraction made of TWO segments — the Flux single block's proj_out, K = 3072 + K2 = 12288 — is the other way round: 1210-1214 TFLOP/s under the
  // persistent schedule, 1186-1200 on one tile per workgroup, same box, two runs each; it stays persistent)
  if (g_persist_override < 0 && p.K >= 12288 && p.K2 == 0) return false;      // (a forced schedule, set_persistent(1), ignores the heuristic)
  bool ok = (p.ldc % 8 == 0) && (((uintptr_t)p.C & 15) == 0) && (((uintptr_t)p.A & 15) == 0) && (((uintptr_t)p.B & 15) == 0) && p.lda % 8 == 0 && p.ldb % 8 == 0;
  if (p.bias) ok = ok && (((uintptr_t)p.bias & 15) == 0);
  if (p.aux_out) ok = ok && (p.ld_aux_out % 8 == 0) && (((uintptr_t)p.aux_out & 15) == 0);
  if (p.aux_in) ok = ok && (p.ld_aux_in % 8 == 0) && (((uintptr_t)p.aux_in & 15) == 0);
  if (p.gate) ok = ok && (p.gate_stride % 8 == 0) && (((uintptr_t)p.gate & 15) == 0);
  return ok;
}
template <int EPI>
static int launch_pz(void* stream, const GemmP& p, int tiles) {
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_pz<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, PZ_LDS); }
  const int wgs = tiles < device_cus() ? tiles : device_cus();
  hipLaunchKernelGGL((k_gemm_pz<EPI>), dim3(wgs), dim3(PQ_THREADS), PZ_LDS, (hipStream_t)stream, p, tiles);
  return st355_check_launch("gemm_pz");
}

template <int EPI>
static int launch_256(void* stream, const GemmGroup& g, int tiles) {
  if (tiles == g.tiles0 && pz_ok(g.p[0], tiles)) return launch_pz<EPI>(stream, g.p[0], tiles);     // (a two-problem grid stays one-tile-per-workgroup)
  return launch_pq<EPI>(stream, g, tiles);
}

#define DISPATCH_EPI(fn, epi, ...)                                                           \
  switch (epi) {                                                                             \
    case ST355_EPI_NONE: return fn<ST355_EPI_NONE>(__VA_ARGS__);                             \
    case ST355_EPI_GELU: return fn<ST355_EPI_GELU>(__VA_ARGS__);                             \
    case ST355_EPI_GATE_RESIDUAL: return fn<ST355_EPI_GATE_RESIDUAL>(__VA_ARGS__);           \
    case ST355_EPI_MUL_GELU_GRAD: return fn<ST355_EPI_MUL_GELU_GRAD>(__VA_ARGS__);           \
    default: return fn<ST355_EPI_ADD>(__VA_ARGS__);                                          \
  }

// =================================================================================================
// k_gemm_thin: C[M, N <= 128] = A[M, K] B[N, K]^T for the rank-space projections of the LoRA path (x A^T, dY (sB)): N = 64 / 128 output columns, K in the
// thousands, M = every token of the batch.  The problem is ONE streaming pass over A (42 MB at the SDXL 32^2 level) — HBM-bound, no reuse worth a ring: round 5 ran
// it as split-K 128x128 tiles + a slab reduce (15.8 + 5.4 us per launch against ~9 us of A traffic; 620 launches per SDXL-LoRA step, rocprofv3 r06).
// Here a workgroup owns 64 rows; its four waves split the K-tiles (wave w takes tiles w, w + 4, ...), each with the whole 64 x N accumulator; operands go straight
// from global memory to MFMA fragments (no LDS staging: nothing is reused inside a wave beyond the two row fragments per weight fragment).  A lane owns 64
// contiguous BYTES of its row per K-tile (k-half = lane >> 5): both operands use the same lane -> k mapping, so any order that covers the tile once is a correct
// contraction.  Next tile's loads are issued before the current tile's MFMAs.  The four partial accumulators meet in LDS (fixed order), one bf16 store.
// =================================================================================================
template <int NT, bool PREF>          // NT = N / 32 column tiles (2 or 4); PREF: next tile's fragments requested before this tile's MFMAs (N = 64; N = 128 keeps one
                                      // fragment set — its 128 accumulators leave no room for two — and hides the latency with two waves per SIMD instead)
__global__ void __launch_bounds__(256) k_gemm_thin(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb, bf16* __restrict__ C, int64_t ldc,
                                                  int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;                                     // [2 regions][64 rows][N] fp32
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int m0 = blockIdx.x * 64;
  const int nkt = K / 64;
  const bf16* arow[2];
#pragma unroll
  for (int j = 0; j < 2; j++) arow[j] = A + (int64_t)min(m0 + 32 * j + l31, M - 1) * lda + khalf * 32;
  const bf16* brow[NT];
#pragma unroll
  for (int i = 0; i < NT; i++) brow[i] = B + (int64_t)min(32 * i + l31, N - 1) * ldb + khalf * 32;
  f32x16 acc[NT][2];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  bf16x8 xa[2][4], wa[NT][4], xb[2][4], wb[NT][4];
  auto load = [&](int t, bf16x8 (&x)[2][4], bf16x8 (&w)[NT][4]) {
    const int k0 = t * 64;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) x[j][q] = *(const bf16x8*)(arow[j] + k0 + 8 * q);
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
      for (int q = 0; q < 4; q++) w[i][q] = *(const bf16x8*)(brow[i] + k0 + 8 * q);
  };
  auto mma = [&](bf16x8 (&x)[2][4], bf16x8 (&w)[NT][4]) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i][q], x[j][q], acc[i][j], 0, 0, 0);
  };
  int t = wv;
  if (PREF) {
    if (t < nkt) load(t, xa, wa);
    while (t < nkt) {
      if (t + 4 < nkt) load(t + 4, xb, wb);
      mma(xa, wa);
      t += 4;
      if (t >= nkt) break;
      if (t + 4 < nkt) load(t + 4, xa, wa);
      mma(xb, wb);
      t += 4;
    }
  } else {
    for (; t < nkt; t += 4) { load(t, xa, wa); mma(xa, wa); }
  }
  // the four K-partials meet in a fixed order through TWO 64 x N fp32 regions: (0 += 2, 1 += 3), then 0 += 1.  acc[i][j][4a + b] = D[feature 32 i + 8 a + 4 khalf + b][token 32 j + l31]
  auto put = [&](float* reg) {
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          f32x4 v;
#pragma unroll
          for (int b = 0; b < 4; b++) v[b] = acc[i][j][4 * a + b];
          *(f32x4*)(reg + (size_t)(32 * j + l31) * N + 32 * i + 8 * a + 4 * khalf) = v;
        }
  };
  auto add = [&](const float* reg) {
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          const f32x4 v = *(const f32x4*)(reg + (size_t)(32 * j + l31) * N + 32 * i + 8 * a + 4 * khalf);
#pragma unroll
          for (int b = 0; b < 4; b++) acc[i][j][4 * a + b] += v[b];
        }
  };
  float* reg0 = red;
  float* reg1 = red + (size_t)64 * N;
  if (wv >= 2) put(wv == 2 ? reg0 : reg1);
  __syncthreads();
  if (wv < 2) add(wv == 0 ? reg0 : reg1);
  __syncthreads();
  if (wv == 1) put(reg0);
  __syncthreads();
  if (wv == 0) { add(reg0); put(reg1); }
  __syncthreads();
  const int n8 = N / 8;
  for (int idx = threadIdx.x; idx < 64 * n8; idx += 256) {
    const int row = idx / n8, c = (idx % n8) * 8;
    if (m0 + row >= M) continue;
    const f32x4 lo = *(const f32x4*)(reg1 + (size_t)row * N + c), hi = *(const f32x4*)(reg1 + (size_t)row * N + c + 4);
    bf16x8 o;
#pragma unroll
    for (int b = 0; b < 4; b++) { o[b] = f2bf(lo[b]); o[4 + b] = f2bf(hi[b]); }
    *(bf16x8*)(C + (int64_t)(m0 + row) * ldc + c) = o;
  }
}
// k_gemm_rows (r6): the same problem with the ROWS split over the waves instead of K.  A workgroup of two waves owns 64 rows, each wave 32 of them over the whole K
// (no partial sums to meet: k_gemm_thin spends four barriers and two 64 x N fp32 round trips through LDS per 64 rows, a third of its time at K = 1280); the
// [N, 64] weight K-tile is staged ONCE per workgroup in LDS (XOR-swizzled 16-byte chunks: the 8 lanes a ds_read_b128 serves per clock hit 8 distinct chunks) and
// double-buffered — one barrier per K-tile —, the A fragments go from global memory straight to registers one K-tile ahead.  128-thread workgroups, 16 / 32 KiB
// of LDS: 4+ workgroups per CU keep the A stream's loads in flight.  Same lane -> k mapping as k_gemm_thin (a lane owns 64 contiguous bytes of its row per K-tile).
template <int NT>
__global__ void __launch_bounds__(128) k_gemm_rows(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb, bf16* __restrict__ C, int64_t ldc,
                                                  int M, int K) {
  constexpr int N = 32 * NT;
  constexpr int WCH = N * 8 / 128;                               // 16-byte chunks of a weight K-tile per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x [N rows x 128 B] weight K-tiles; afterwards 2 x [32 rows x N] fp32 output staging
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int m0 = blockIdx.x * 64 + wv * 32;
  const int nkt = K / 64;
  // A: COALESCED requests (8 lanes per 128-byte row segment of the K-tile; a row-per-lane request touches 32 lines for 16 bytes each) into registers three
  // K-tiles ahead, through the wave's own swizzled LDS slice one K-tile ahead, read back as MFMA fragments (row l31, chunks 4 khalf .. 4 khalf + 3)
  const bf16* asrc[4];
  int adst[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int idx = lane + 64 * c, row = idx >> 3, ch = idx & 7;
    asrc[c] = A + (int64_t)min(m0 + row, M - 1) * lda + ch * 8;
    adst[c] = row * 128 + ((ch ^ (row & 7)) << 4);
  }
  int ard[4];
#pragma unroll
  for (int q = 0; q < 4; q++) ard[q] = l31 * 128 + (((khalf * 4 + q) ^ (l31 & 7)) << 4);
  char* abuf = smem + 2 * (N * 128) + wv * 8192;                 // this wave's two 4-KiB A K-tiles
  const bf16* wsrc[WCH];
  int wdst[WCH];
#pragma unroll
  for (int c = 0; c < WCH; c++) {
    const int idx = tid + 128 * c, row = idx >> 3, ch = idx & 7;
    wsrc[c] = B + (int64_t)row * ldb + ch * 8;
    wdst[c] = row * 128 + ((ch ^ (row & 7)) << 4);
  }
  int wrd[NT][4];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = 32 * i + l31;
      wrd[i][q] = col * 128 + (((khalf * 4 + q) ^ (col & 7)) << 4);
    }
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  // rings: A chunks three K-tiles ahead in registers (x[t & 3]), weight chunks two ahead (wr[t & 1]); both are put into LDS one K-tile ahead (buffers t & 1)
  bf16x8 x[4][4], wr[2][WCH];
  auto load_x = [&](int t, bf16x8 (&xx)[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) xx[c] = *(const bf16x8*)(asrc[c] + t * 64);
  };
  auto load_w = [&](int t, bf16x8 (&ww)[WCH]) {
#pragma unroll
    for (int c = 0; c < WCH; c++) ww[c] = *(const bf16x8*)(wsrc[c] + t * 64);
  };
  auto put_x = [&](int buf, bf16x8 (&xx)[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) *(bf16x8*)(abuf + buf * 4096 + adst[c]) = xx[c];
  };
  auto put_w = [&](int buf, bf16x8 (&ww)[WCH]) {
#pragma unroll
    for (int c = 0; c < WCH; c++) *(bf16x8*)(smem + buf * (N * 128) + wdst[c]) = ww[c];
  };
  auto mma = [&](int buf) {
    const char* wb = smem + buf * (N * 128);
    const char* ab = abuf + buf * 4096;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bf16x8 xf = *(const bf16x8*)(ab + ard[q]);
#pragma unroll
      for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wb + wrd[i][q]), xf, acc[i], 0, 0, 0);
    }
  };
  // branch-free body (nkt % 4 == 0, the launcher checks): requests past the last K-tile are clamped to it — a conditional request makes the compiler's
  // s_waitcnt bookkeeping assume the shorter queue at every join and the counted waits collapse to vmcnt(0)
  const int last = nkt - 1;
  load_w(0, wr[0]);
  load_x(0, x[0]);
  load_w(min(1, last), wr[1]);
  load_x(min(1, last), x[1]);
  load_x(min(2, last), x[2]);
  put_w(0, wr[0]);
  put_x(0, x[0]);
  __syncthreads();
  for (int t0 = 0; t0 < nkt; t0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int t = t0 + u;
      load_w(min(t + 2, last), wr[u & 1]);
      load_x(min(t + 3, last), x[(u + 3) & 3]);
      mma(u & 1);
      put_w((u + 1) & 1, wr[(u + 1) & 1]);
      put_x((u + 1) & 1, x[(u + 1) & 3]);
      __syncthreads();
    }
  }
  // acc[i][4a + b] = D[feature 32 i + 8 a + 4 khalf + b][token l31]: through the wave's [32 tokens x N] fp32 slice (the weight buffers are idle: every wave passed
  // the last barrier), then 16-byte row stores
  float* st = (float*)smem + (size_t)wv * 32 * N;
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int a = 0; a < 4; a++) {
      f32x4 v;
#pragma unroll
      for (int b = 0; b < 4; b++) v[b] = acc[i][4 * a + b];
      const int cch = (32 * i + 8 * a + 4 * khalf) >> 2;                       // 16-byte chunk of the token's fp32 row, swizzled by the token
      *(f32x4*)(st + (size_t)l31 * N + ((cch ^ (l31 & 7)) << 2)) = v;
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);                                          // lgkmcnt(0): the wave reads back what its own lanes wrote
  __builtin_amdgcn_wave_barrier();
  constexpr int n8 = N / 8;
  for (int idx = lane; idx < 32 * n8; idx += 64) {
    const int row = idx / n8, c8 = idx % n8;
    if (m0 + row >= M) continue;
    const f32x4 lo = *(const f32x4*)(st + (size_t)row * N + (((2 * c8) ^ (row & 7)) << 2)), hi = *(const f32x4*)(st + (size_t)row * N + (((2 * c8 + 1) ^ (row & 7)) << 2));
    bf16x8 o;
#pragma unroll
    for (int b = 0; b < 4; b++) { o[b] = f2bf(lo[b]); o[4 + b] = f2bf(hi[b]); }
    *(bf16x8*)(C + (int64_t)(m0 + row) * ldc + c8 * 8) = o;
  }
}
// Measured r6 (tools/probes/rows_gemm_bench.py, operands flushed from the caches between launches; same box): 32768 x 64 x 1280 46.0 -> 35.7 us, 16384 x 64 x 1280
// 26.3 -> 21.6, 32768 x 128 x 1280 49.1 -> 40.9; from K ~ 3000 the tile schedules win at N = 128 (36864 x 128 x 9216: 211 vs 252 us) and k_gemm_thin is level at
// N = 64 (the Flux step did not move).  In-step, same box: SDXL-LoRA batch 32 695.4 -> 690.7 ms.  So: N = 64 / 128 with K < 2048.
static int rows_mode() {          // ST355_GEMM_ROWS: 0 = off, 1 (default) = K < 2048, 2 = every K, 64 / 128 = that width only (every K)
  static int v = -1;
  if (v < 0) { const char* e = getenv("ST355_GEMM_ROWS"); v = e ? atoi(e) : 1; }
  return v;
}
static bool rows_ok(const st355_gemm_args* a) {
  const int m = rows_mode();
  if (!m || !(a->N == 64 || a->N == 128) || (m > 2 && m != a->N) || (m == 1 && a->K >= 2048)) return false;
  return a->K2 == 0 && a->epilogue == ST355_EPI_NONE && !a->bias && !a->seg_rows && a->K >= 256 && a->K % 256 == 0 && a->M >= 1024 && a->ldc % 8 == 0 &&
         a->lda % 8 == 0 && a->ldb % 8 == 0 && ((uintptr_t)a->C % 16 == 0) && ((uintptr_t)a->A % 16 == 0) && ((uintptr_t)a->B % 16 == 0);
}
static int launch_rows(void* stream, const st355_gemm_args* a) {
  const dim3 grid((a->M + 63) / 64), block(128);
  if (a->N == 64)
    hipLaunchKernelGGL((k_gemm_rows<2>), grid, block, 2 * 64 * 128 + 16384, (hipStream_t)stream, (const bf16*)a->A, a->lda, (const bf16*)a->B, a->ldb, (bf16*)a->C, a->ldc, a->M, a->K);
  else
    hipLaunchKernelGGL((k_gemm_rows<4>), grid, block, 2 * 128 * 128 + 16384, (hipStream_t)stream, (const bf16*)a->A, a->lda, (const bf16*)a->B, a->ldb, (bf16*)a->C, a->ldc, a->M, a->K);
  return st355_check_launch("gemm_rows");
}
static bool thin_ok(const st355_gemm_args* a) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("ST355_GEMM_THIN"); on = (e && e[0] == '0') ? 0 : 1; }          // A/B: 0 = the split-K tile path
  // N = 128 (three adapters of a fused projection): a 64-row workgroup re-reads the whole [128, K] weight block — at Flux's K = 3072 that is 2x the bytes of its A
  // rows through the CU's L1, and the launch ran 262 us against ~130 on the tile schedules (rocprofv3 r06: +14 ms per Flux step); at the SDXL shapes it was a wash
  // (75.5 vs 74.1 us at K = 3840).  So N = 128 stays on the tile schedules (ST355_GEMM_THIN128=1: A/B)
  static int on128 = -1;
  if (on128 < 0) { const char* e = getenv("ST355_GEMM_THIN128"); on128 = (e && e[0] == '1') ? 1 : 0; }
  // N = 64: wins where the tile path needs a split-K + reduce pair or K is long (SDXL 32^2 level at batch 16: 19.7 vs 21.2 us; Flux M = 36 864, K = 3072: -3 ms per
  // step); with 32 768+ rows of a short K the tile schedules fill the chip by themselves (SDXL-LoRA batch 32: 659.8 ms without it vs 664.7, same box, r06)
  if (a->N == 64 && a->M > 24576 && a->K < 2048 && !on128) return false;
  return on && (a->N == 64 || (a->N == 128 && on128)) && a->K2 == 0 && a->epilogue == ST355_EPI_NONE && !a->bias && !a->seg_rows && a->K >= 256 && a->M >= 1024 &&
         a->ldc % 8 == 0 && ((uintptr_t)a->C % 16 == 0);
}
static int launch_thin(void* stream, const st355_gemm_args* a) {
  const int lds = 2 * 64 * a->N * 4;
  const dim3 grid((a->M + 63) / 64), block(256);
  if (a->N == 64) {
    static St355AttrOnce set;
    if (set.need()) { hipFuncSetAttribute((const void*)k_gemm_thin<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
    hipLaunchKernelGGL((k_gemm_thin<2, true>), grid, block, lds, (hipStream_t)stream, (const bf16*)a->A, a->lda, (const bf16*)a->B, a->ldb, (bf16*)a->C, a->ldc, a->M, a->N, a->K);
  } else {
    static St355AttrOnce set;
    if (set.need()) { hipFuncSetAttribute((const void*)k_gemm_thin<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
    hipLaunchKernelGGL((k_gemm_thin<4, false>), grid, block, lds, (hipStream_t)stream, (const bf16*)a->A, a->lda, (const bf16*)a->B, a->ldb, (bf16*)a->C, a->ldc, a->M, a->N, a->K);
  }
  return st355_check_launch("gemm_thin");
}

static int launch_splitk(void* stream, GemmP& p, const st355_gemm_args* a, int ksplit) {
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_s2<EPI_SPLITK>, hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS); }
  p.partial = (float*)a->workspace; p.ksplit = ksplit;
  const int tiles = ((p.M + S2_BM - 1) / S2_BM) * ((p.N + S2_BN - 1) / S2_BN);
  hipLaunchKernelGGL(k_gemm_s2<EPI_SPLITK>, dim3(tiles * ksplit), dim3(S2_THREADS), S2_LDS, (hipStream_t)stream, p);
  int rc = st355_check_launch("gemm_s2_splitk");
  if (rc) return rc;
  const int64_t n4 = (int64_t)p.M * (p.N / 4);
  hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)p.partial, ksplit,
                     p.bias, p.C, p.ldc, p.M, p.N, 0);
  return st355_check_launch("gemm_splitk_reduce");
}

// ---- stream-K tail plan (k_gemm_pq<..., SK>) ----
static int g_tail_override = -1;             // st355_gemm_set_tail_split: 0 / 1 force, -1 = environment (ST355_GEMM_TAIL=0 turns it off)
static int tail_enabled() {
  if (g_tail_override >= 0) return g_tail_override;
  static int v = -1;
  if (v < 0) { const char* e = getenv("ST355_GEMM_TAIL"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}
// The fix-up of the stream-K tail is only correct when the slices of a tile (block ids congruent mod 8) run on ONE XCD — the hardware's round-robin dispatch of
// consecutive workgroups over the 8 XCDs.  That is checked, not assumed: once per device, before the first cut launch, 512 probe workgroups record their XCC_ID
// register and the host compares them (one stream synchronisation in the life of the process; never inside a hipGraph capture — a capture that arrives before
// the probe has run simply keeps the uncut schedule).  A device that places workgroups differently (another partition mode) keeps the uncut schedule for good.
__global__ void k_xcc_probe(int* out) {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
static std::atomic<int> g_tail_placement[32];  // per device: 0 = not probed, 1 = round-robin confirmed, 2 = not confirmed
static int tail_placement_ok(void* stream, void* scratch) {
  std::atomic<int>* ok = g_tail_placement;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return 0;
  const int v = ok[d].load(std::memory_order_relaxed);
  if (v) return v == 1;
  if (const char* e = getenv("ST355_GEMM_TAIL_PROBE")) if (e[0] == '0') { ok[d] = 1; return 1; }      // lab: trust the placement
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return 0;
  constexpr int NB = 512;
  int host[NB];
  hipLaunchKernelGGL(k_xcc_probe, dim3(NB), dim3(64), 0, (hipStream_t)stream, (int*)scratch);
  if (hipMemcpyAsync(host, scratch, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    (void)hipGetLastError();
    ok[d] = 2;
    return 0;
  }
  bool good = true;
  for (int b = 8; b < NB && good; b++) good = host[b] == host[b & 7];
  for (int i = 0; i < 8 && good; i++)
    for (int j = 0; j < i && good; j++) good = host[i] != host[j];
  ok[d] = good ? 1 : 2;
  return good;
}

// true: run `p` as g (blocks = *blocks).  The last round of 256x256 tiles is cut when it fills at most half the chip and the problem is short enough for that
// round to matter (<= 6 full rounds before it); 2..4 slices of >= 4 K-tiles each, slabs in the caller's workspace.
static bool tail_plan(void* stream, const st355_gemm_args* a, const GemmP& p, GemmGroup& g, int* blocks) {
  if (!tail_enabled() || !a->workspace || !a->tile_flags || ((uintptr_t)a->workspace % 16) || p.partial || p.scale_b || p.conv_taps || p.img_add) return false;
  const int cus = device_cus();
  const int T = p4_tiles(p), rounds = T / cus, rem = T % cus;
  static int max_rounds = -1, min_tiles = -1;
  if (max_rounds < 0) { const char* e = getenv("ST355_GEMM_TAIL_ROUNDS"); max_rounds = e ? atoi(e) : 6; }
  if (min_tiles < 0) { const char* e = getenv("ST355_GEMM_TAIL_MIN_TILES"); min_tiles = e ? atoi(e) : 128; }       // problems below one round: from this many tiles
  if (rem == 0 || 2 * rem > cus || rounds > max_rounds || T < min_tiles || rem > 1024) return false;
  int s = cus / rem;
  if (s > 4) s = 4;
  const int nt_all = p.K / PQ_BK;
  // the fix-up costs ~10 us at two slices and ~40 at four (three slabs per cut tile exceed the XCD's L2: measured r06), a tile ~28 us per 1000 of K: four slices pay from K ~ 4000
  static int k4 = -1;
  if (k4 < 0) { const char* e = getenv("ST355_GEMM_TAIL_K4"); k4 = e ? atoi(e) : 4096; }
  if (p.K + p.K2 < k4 && s > 2) s = 2;
  // below K ~ 2000 the cut round does not pay at all (r06: 16384 x 1280 x 1280+64: 74.7 us cut in two vs 76.9 uncut, 93.7 vs 83.0 with the residual-add epilogue)
  static int kmin = -1;
  if (kmin < 0) { const char* e = getenv("ST355_GEMM_TAIL_KMIN"); kmin = e ? atoi(e) : 2048; }
  if (p.K + p.K2 < kmin) return false;
  if (s > nt_all / 4) s = nt_all / 4;
  if (s < 2) return false;
  if ((nt_all + s - 1) / s * (s - 1) >= nt_all) return false;               // (every slice owns at least one K-tile)
  if ((int64_t)rem * (s - 1) * PQ_BM * PQ_BN * 4 > a->workspace_bytes) return false;
  g.p[0] = p; g.p[1] = p; g.tiles0 = T;
  g.sk_first = rounds * cus; g.sk_s = s; g.sk_ws = (float*)a->workspace; g.sk_flags = (int*)a->tile_flags;
  *blocks = g.sk_first + (rem + 7) / 8 * 8 * s;
  return tail_placement_ok(stream, a->workspace) != 0;
}

static int run_one(void* stream, const st355_gemm_args* a) {
  if (rows_ok(a)) return launch_rows(stream, a);
  if (thin_ok(a)) return launch_thin(stream, a);
  GemmP p = to_p(a);
  // thin problems (the LoRA rank-space projections: N <= 128, K in the thousands) stream A once and have only M/128 tiles: split
  // K so that >= 2 workgroups per CU are in flight, partial sums through the caller's fp32 workspace (fixed-order reduce)
  static int thin_splitk = -1;
  if (thin_splitk < 0) { const char* e = getenv("ST355_THIN_SPLITK"); thin_splitk = (e && e[0] == '0') ? 0 : 1; }      // A/B: 0 = no split-K for thin problems
  if (thin_splitk && a->workspace && p.N <= S2_BN && p.K2 == 0 && a->epilogue == ST355_EPI_NONE && p.K >= 1024 && p.M >= 512 && !(a->seg_rows && a->seg_c)) {   // (the slab reduce writes compact C rows)
    const int tiles = (p.M + S2_BM - 1) / S2_BM;
    int ks = (512 + tiles - 1) / tiles;
    // more than one round of resident workgroups already (288 row tiles for the LoRA projections of a 36 864-token batch): a split only adds slab traffic
    // and a ragged second round — measured 83 us (ks = 2) / 77 (ks = 3) against 65 us on the 256x128 schedule without a split, which the code below picks
    if (tiles > 256) ks = 1;
    const int nt1 = p.K / BK;
    if (ks > nt1 / 4) ks = nt1 / 4;                       // >= 4 K-tiles per slice
    if (ks > 16) ks = 16;
    if (ks >= 2 && (int64_t)ks * p.M * p.N * 4 <= a->workspace_bytes && ((uintptr_t)a->workspace % 16) == 0) return launch_splitk(stream, p, a, ks);
  }
  // the deep-pipelined schedule needs enough tiles to fill 256 CUs; tiny problems stay on the 128x128 schedule
  // 256x256 tiles only when they (nearly) fill the 256 CUs at one workgroup each
  if (a->epilogue == ST355_EPI_QK_NORM_ROPE) {          // the fused q/k epilogue exists in the 256x256 schedule only
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p4_tiles(p);
    return launch_pq<ST355_EPI_QK_NORM_ROPE>(stream, g, g.tiles0);
  }
  if (a->epilogue == ST355_EPI_HEADS) {                 // ... and the head-splitting one
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p4_tiles(p);
    return launch_pq<ST355_EPI_HEADS>(stream, g, g.tiles0);
  }
  if (a->epilogue == ST355_EPI_GEGLU || a->epilogue == ST355_EPI_GEGLU_GRAD) {      // likewise the GEGLU pair
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p4_tiles(p);
    // the backward epilogue (reads the kept pre-activation, writes d value | d gate: 2.7 GB per launch at the SDXL 32^2 level of batch 32 against 0.43 PFLOP of
    // MFMA work) on the PERSISTENT schedule when the problem is made of full tiles (r6): the next tile's first K-tiles load under it.  ST355_GEGLU_GRAD_PZ=0: A/B
    static int gg_pz = -1;
    if (gg_pz < 0) { const char* e = getenv("ST355_GEGLU_GRAD_PZ"); gg_pz = (e && e[0] == '0') ? 0 : 1; }
    if (gg_pz && a->epilogue == ST355_EPI_GEGLU_GRAD && !a->bias && pz_ok(p, g.tiles0)) return launch_pz<ST355_EPI_GEGLU_GRAD>(stream, p, g.tiles0);
    return a->epilogue == ST355_EPI_GEGLU ? launch_pq<ST355_EPI_GEGLU>(stream, g, g.tiles0) : launch_pq<ST355_EPI_GEGLU_GRAD>(stream, g, g.tiles0);
  }
  // tile quantisation (r6 experiment knob ST355_GEMM_P3_WINDOW=lo,hi): a problem whose 256x256 tiles leave the chip's last round mostly empty — 320 tiles on
  // 256 CUs: two rounds for 1.25 rounds of work, the N = 1280 projections of the SDXL 32^2 level at batch 16 — may run better as 256x128 tiles (2.5 rounds of half the size)
  static int p3_lo = -1, p3_hi = -1;
  if (p3_lo < 0) { const char* e = getenv("ST355_GEMM_P3_WINDOW"); p3_lo = 0; p3_hi = 0; if (e) sscanf(e, "%d,%d", &p3_lo, &p3_hi); }
  const bool p3_window = p3_hi > 0 && p4_tiles(p) >= p3_lo && p4_tiles(p) < p3_hi;
  // N <= 128 (the N = 128 adapter projections of a long batch: 36 864 x 128 x 3072 in the Flux step, 95 launches): a 256x256 tile would carry 128 dead columns —
  // half of its staging and MFMAs — on 144 workgroups; the 256x128 schedule is the one that fits (lab r3: 65 us; the 256-wide rule took these problems over when its
  // tile threshold went from 200 to 128 in r3 and ran them at ~130 us in-step).  ST355_GEMM_N128_P3=0: A/B
  static int n128_p3 = -1;
  if (n128_p3 < 0) { const char* e = getenv("ST355_GEMM_N128_P3"); n128_p3 = (e && e[0] == '0') ? 0 : 1; }
  const bool narrow = n128_p3 && p.N <= P3_BN && p.K2 == 0 && gemm_impl_choice() >= 1 && p.M > 128 && p3_tiles(p) >= 128;
  if (gemm_impl_choice() >= 2 && !p3_window && !narrow) {
    GemmGroup g;
    int blocks = 0;
    if (tail_plan(stream, a, p, g, &blocks)) { DISPATCH_EPI(launch_pq_sk, a->epilogue, stream, g, blocks); }
  }
  if (gemm_impl_choice() >= 2 && p4_tiles(p) >= min_tiles_256() && !p3_window && !narrow) {
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p4_tiles(p);
    DISPATCH_EPI(launch_256, a->epilogue, stream, g, g.tiles0);
  }
  const bool big = gemm_impl_choice() >= 1 && p.M > 128 && p3_tiles(p) >= 128;
  if (big) {
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p3_tiles(p);
    DISPATCH_EPI(launch_p3, a->epilogue, stream, g, g.tiles0);
  }
  DISPATCH_EPI(launch_s2, a->epilogue, stream, p);
}

extern "C" int st355_gemm_tail_placement(void) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return 2;
  return g_tail_placement[d].load(std::memory_order_relaxed);
}

extern "C" int st355_gemm_set_tail_split(int mode) {
  const int prev = g_tail_override;
  g_tail_override = mode < 0 ? -1 : (mode ? 1 : 0);
  return prev;
}

extern "C" int st355_gemm_set_persistent(int mode) {
  const int prev = g_persist_override;
  g_persist_override = mode < 0 ? -1 : (mode ? 1 : 0);
  return prev;
}

extern "C" int st355_gemm_bf16(void* stream, const st355_gemm_args* a) {
  int rc = validate(a);
  if (rc) return rc;
  ProfScope ps(stream, ST355_K_GEMM, gemm_flops(a), gemm_bytes(a), "%dx%dx%d+%d e%d", a->M, a->N, a->K, a->K2, a->epilogue);
  return run_one(stream, a);
}

// ---- fp8-native Linear (fp8_native.py:41-119) -----------------------------------------------------------------------------------------
extern "C" int st355_linear_fp8(void* stream, const void* xq, int64_t ldx, const float* scale_a, const void* wq, int64_t ldw,
                                const float* w_scale, const void* bias, void* out, int64_t ldo, int M, int N, int K) {
  ST_REQUIRE(xq && wq && scale_a && w_scale && out && M > 0 && N > 0 && K > 0, "linear_fp8: bad args");
  ST_REQUIRE(K % 128 == 0 && N % 8 == 0 && ldx % 16 == 0 && ldw % 16 == 0 && ldo % 8 == 0, "linear_fp8: K must be a multiple of 128, N of 8, rows 16-byte aligned");
  ST_REQUIRE(((uintptr_t)xq % 16 == 0) && ((uintptr_t)wq % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)w_scale % 16 == 0), "linear_fp8: misaligned pointer");
  ST_REQUIRE(256 * (ldx > ldw ? ldx : ldw) + (int64_t)K < ((int64_t)1 << 31), "linear_fp8: a 256-row tile must fit 32-bit buffer offsets");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)xq; p.lda = ldx; p.B = (const bf16*)wq; p.ldb = ldw; p.C = (bf16*)out; p.ldc = ldo;
  p.M = M; p.N = N; p.K = K; p.K2 = 0; p.ksplit = 1;
  p.bias = (const bf16*)bias; p.scale_a = scale_a; p.scale_b = w_scale;
  GemmGroup g;
  g.p[0] = p; g.p[1] = p;
  g.tiles0 = ((M + PQ_BM - 1) / PQ_BM) * ((N + PQ_BN - 1) / PQ_BN);
  ProfScope ps(stream, ST355_K_GEMM, 2.0 * (double)M * N * K, (double)M * K + (double)N * K + 2.0 * (double)M * N, "F8 %dx%dx%d", M, N, K);
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_pq<ST355_EPI_NONE, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PQ_LDS); }
  hipLaunchKernelGGL((k_gemm_pq<ST355_EPI_NONE, false, true>), dim3(g.tiles0), dim3(PQ_THREADS), PQ_LDS, (hipStream_t)stream, g);
  return st355_check_launch("linear_fp8");
}

// ---- weight-gradient form -------------------------------------------------------------------------------------------------------
template <int EPI>
static int launch_tn(void* stream, const GemmGroup& g, int tiles) {
  static St355AttrOnce attr_set;
  if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_pq<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PQ_LDS); }
  hipLaunchKernelGGL((k_gemm_pq<EPI, true>), dim3(tiles), dim3(PQ_THREADS), PQ_LDS, (hipStream_t)stream, g);
  return st355_check_launch("gemm_tn");
}

// taps == 9: C is [P, 9*Q]; column block `tap` = L^T (R shifted by (tap/3)*wp + tap%3 rows)  — the 3x3 convolution weight gradient in one launch
static int gemm_tn_impl(void* stream, const void* L, int64_t ldl, const void* R, int64_t ldr, void* C, int64_t ldc,
                        int64_t Mc, int P, int Q, int accumulate, void* workspace, int64_t workspace_bytes, int taps, int wp,
                        int64_t seg_rows = 0, int64_t seg_l = 0, int64_t seg_r = 0) {
  ST_REQUIRE(L && R && C && Mc > 0 && P > 0 && Q > 0, "gemm_tn: bad args");
  if (seg_rows) {
    ST_REQUIRE(taps == 1 && seg_rows >= 2 * PQ_BK && seg_rows % PQ_BK == 0 && Mc % seg_rows == 0 && (seg_l == 0 || seg_l >= seg_rows) && (seg_r == 0 || seg_r >= seg_rows),
               "gemm_tn_seg: segments of %lld contraction rows (a multiple of 64, >= 128, that divides Mc = %lld), physical segment strides >= that", (long long)seg_rows, (long long)Mc);
    const int64_t nseg = Mc / seg_rows;
    const int64_t span_l = ((nseg - 1) * (seg_l ? seg_l : seg_rows) + seg_rows) * ldl * 2, span_r = ((nseg - 1) * (seg_r ? seg_r : seg_rows) + seg_rows) * ldr * 2;
    ST_REQUIRE(span_l < ((int64_t)1 << 31) && span_r < ((int64_t)1 << 31) && Mc / PQ_BK < 65536, "gemm_tn_seg: operand too large for the 32-bit buffer offsets (2 GiB)");
  }
  ST_REQUIRE(Mc % PQ_BK == 0, "gemm_tn: the contraction length (%lld rows) must be a multiple of 64 (pad the operands with zero rows)", (long long)Mc);
  ST_REQUIRE(P % 8 == 0 && Q % 8 == 0 && ldl % 8 == 0 && ldr % 8 == 0 && ldc % 8 == 0, "gemm_tn: P, Q and the leading dimensions must be multiples of 8");
  ST_REQUIRE(((uintptr_t)L % 16 == 0) && ((uintptr_t)R % 16 == 0) && ((uintptr_t)C % 16 == 0), "gemm_tn: misaligned pointer");
  ST_REQUIRE(Mc * (int64_t)(ldl > ldr ? ldl : ldr) * 2 < ((int64_t)1 << 31), "gemm_tn: operand too large for the 32-bit buffer offsets (2 GiB)");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)L; p.lda = ldl; p.B = (const bf16*)R; p.ldb = ldr; p.C = (bf16*)C; p.ldc = ldc;
  p.M = P; p.N = Q; p.K = (int)Mc; p.K2 = 0; p.ksplit = 1; p.part_ld = (int64_t)taps * Q;
  p.conv_taps = taps == 9 ? 9 : 0; p.conv_wp = wp;
  if (seg_rows && Mc > seg_rows && ((seg_l && seg_l != seg_rows) || (seg_r && seg_r != seg_rows))) {
    const uint32_t tps = (uint32_t)(seg_rows / PQ_BK);                                     // >= 2 (checked above): ceil(2^32 / tps) fits 32 bits
    p.tn_magic = (uint32_t)((((uint64_t)1 << 32) + tps - 1) / tps);
    p.tn_skip_a = (int)((seg_l ? seg_l - seg_rows : 0) * ldl * 2);
    p.tn_skip_b = (int)((seg_r ? seg_r - seg_rows : 0) * ldr * 2);
  }
  if (accumulate) { p.aux_in = (const bf16*)C; p.ld_aux_in = ldc; }
  GemmGroup g;
  g.p[0] = p; g.p[1] = p;
  g.tiles0 = ((P + PQ_BM - 1) / PQ_BM) * ((Q + PQ_BN - 1) / PQ_BN) * taps;
  ProfScope ps(stream, ST355_K_GEMM, 2.0 * (double)Mc * P * Q * taps, 2.0 * ((double)Mc * (P + Q) + (double)P * Q * taps * (accumulate ? 2 : 1)), "TN%d %dx%dx%lld", taps, P, Q, (long long)Mc);
  // weight matrices are small next to the token count: when the output has too few 256x256 tiles for 256 CUs, slice the contraction
  // (fp32 slabs in the caller's workspace, fixed-order reduce — deterministic)
  const int nt_all = (int)(Mc / PQ_BK);
  int ks = (384 + g.tiles0 - 1) / g.tiles0;
  if (ks > nt_all / 8) ks = nt_all / 8;                // >= 8 K-tiles per slice
  if (ks > 16) ks = 16;
  {
    // r5: the slice count from a cost model instead of "about 384 workgroups": rounds of workgroups over the CUs x (K-tiles per slice + ~6 K-tiles' worth of
    // prologue / slab epilogue), plus the reduce pass's read of one fp32 slab per slice (tiles0 * 256 KiB at ~4.8 TB/s ~= 0.026 K-tile times per tile).  The old rule
    // gave 144 output tiles 3 slices = 1.69 rounds (2 rounds for 84 % of the work); 5 or 7 slices fill their last round.  ST355_TN_KS: 0 = this model, -1 = the old
    // rule, n = n slices (lab).
    static int mode = -2;
    if (mode == -2) { const char* e = getenv("ST355_TN_KS"); mode = e ? atoi(e) : 0; }
    if (mode > 0) ks = mode;
    else if (mode == 0 && g.tiles0 < 2 * device_cus()) {
      const int cus = device_cus();
      double best = 1e30; int best_ks = 1;
      const int ks_max = nt_all / 8 < 16 ? (nt_all / 8 < 1 ? 1 : nt_all / 8) : 16;
      for (int k = 1; k <= ks_max; k++) {
        const int per = (nt_all + k - 1) / k;
        const int rounds = (g.tiles0 * k + cus - 1) / cus;
        const double cost = (double)rounds * (per + 6) + (k > 1 ? 0.026 * g.tiles0 * k : 0.0);
        if (cost < best - 1e-9) { best = cost; best_ks = k; }
      }
      ks = best_ks;
    }
  }
  while (ks >= 2 && workspace && (int64_t)ks * P * Q * taps * 4 > workspace_bytes) ks--;      // as many slices as the caller's slab space holds (never fall back to ONE round of full-K tiles)
  if (ks >= 2) { const int per = (nt_all + ks - 1) / ks; ks = (nt_all + per - 1) / per; }   // no empty K-slices (the ring prologue assumes >= 1 K-tile)
  if (ks >= 2 && workspace && ((uintptr_t)workspace % 16 == 0) && (int64_t)ks * P * Q * taps * 4 <= workspace_bytes) {
    g.p[0].partial = (float*)workspace; g.p[0].ksplit = ks; g.p[0].aux_in = nullptr;
    g.p[1] = g.p[0];
    static St355AttrOnce attr_set;
    if (attr_set.need()) { hipFuncSetAttribute((const void*)k_gemm_pq<EPI_SPLITK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PQ_LDS); }
    hipLaunchKernelGGL((k_gemm_pq<EPI_SPLITK, true>), dim3(g.tiles0 * ks), dim3(PQ_THREADS), PQ_LDS, (hipStream_t)stream, g);
    int rc = st355_check_launch("gemm_tn_splitk");
    if (rc) return rc;
    const int64_t n4 = (int64_t)P * (Q * taps / 4);
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, ks,
                       (const bf16*)nullptr, (bf16*)C, ldc, P, Q * taps, accumulate);
    return st355_check_launch("gemm_tn_splitk_reduce");
  }
  return accumulate ? launch_tn<ST355_EPI_ADD>(stream, g, g.tiles0) : launch_tn<ST355_EPI_NONE>(stream, g, g.tiles0);
}

extern "C" int st355_gemm_tn_bf16(void* stream, const void* L, int64_t ldl, const void* R, int64_t ldr, void* C, int64_t ldc,
                                  int64_t Mc, int P, int Q, int accumulate, void* workspace, int64_t workspace_bytes) {
  return gemm_tn_impl(stream, L, ldl, R, ldr, C, ldc, Mc, P, Q, accumulate, workspace, workspace_bytes, 1, 0);
}

// the same product with a SEGMENTED contraction axis: logical contraction row m of operand X in {L, R} lives at physical row (m / seg_rows) * seg_X + m % seg_rows
// (seg_X = 0: compact).  A stream's rows of a joint [B, S, *] buffer (the image rows of the attention output, of the projection gradient) are then contracted in
// place — round 5 gathered them into a compact copy first (100-300 MB read + written per operand and block).  seg_rows: a multiple of 64 (>= 128) that divides Mc.
extern "C" int st355_gemm_tn_seg_bf16(void* stream, const void* L, int64_t ldl, int64_t seg_l, const void* R, int64_t ldr, int64_t seg_r, void* C, int64_t ldc,
                                      int64_t Mc, int64_t seg_rows, int P, int Q, int accumulate, void* workspace, int64_t workspace_bytes) {
  ST_REQUIRE(seg_rows > 0, "gemm_tn_seg: seg_rows");
  return gemm_tn_impl(stream, L, ldl, R, ldr, C, ldc, Mc, P, Q, accumulate, workspace, workspace_bytes, 1, 0, seg_rows, seg_l, seg_r);
}

// ---- convolution over a zero-bordered NHWC grid (SDXL / SD1.5 UNet, VAE: diffusers ResnetBlock2D / Downsample2D / Upsample2D convs) -------
// grid buffer = [st355_conv_grid_rows(B,H,W), C] bf16: position (b, y, x) of the (H+2) x (W+2) padded image b is row (b*(H+2)+y)*(W+2)+x,
// border positions hold ZERO, and 64 zero rows follow the last image (so that shifted / 64-rounded reads stay inside the buffer).
// 3x3 stride-1 pad-1 conv == ONE GEMM over the grid whose K loop walks the 9 taps as row-shifted views of x (no im2col); 1x1 conv and
// pre-gathered columns (stride 2, tiny Cin) are the taps == 1 case.  Rows [Wp+1, rows - Wp - 1) are computed (every read stays inside
// the images); the first / last Wp+1 positions are border positions and are left untouched (zero in a grid buffer).
extern "C" int64_t st355_conv_grid_rows(int B, int H, int W) { return (int64_t)B * (H + 2) * (W + 2) + 64; }

extern "C" int st355_conv_bf16(void* stream, const void* x, const void* w, const void* bias, const void* img_add, int64_t img_add_stride,
                               const void* residual, void* out, int B, int H, int W, int Cin, int Cout, int taps) {
  ST_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0, "conv: bad args");
  ST_REQUIRE(taps == 1 || taps == 9, "conv: taps must be 1 or 9");
  ST_REQUIRE(Cin % BK == 0 && Cout % 8 == 0, "conv: Cin (%d) must be a multiple of 64 and Cout (%d) of 8", Cin, Cout);
  ST_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)out % 16 == 0), "conv: misaligned pointer");
  const int Hp = H + 2, Wp = W + 2;
  const int64_t Mtot = (int64_t)B * Hp * Wp, p0 = Wp + 1, Mc = Mtot - 2 * p0;
  ST_REQUIRE(Mc > 0 && Mc < ((int64_t)1 << 31), "conv: grid too large");
  ST_REQUIRE(256 * (int64_t)Cin * 2 + (2 * (int64_t)Wp + 2) * Cin * 2 + Cin * 2 < ((int64_t)1 << 31), "conv: a tile must fit 32-bit offsets");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16*)x + (taps == 9 ? 0 : p0 * Cin); p.lda = Cin;
  p.B = (const bf16*)w; p.ldb = (int64_t)taps * Cin;
  p.C = (bf16*)out + p0 * Cout; p.ldc = Cout;
  p.M = (int)Mc; p.N = Cout; p.K = taps * Cin; p.K2 = 0; p.ksplit = 1;
  p.bias = (const bf16*)bias;
  if (residual) { p.aux_in = (const bf16*)residual + p0 * Cout; p.ld_aux_in = Cout; }
  p.conv_taps = taps; p.conv_tpt = Cin / BK; p.conv_wp = Wp; p.conv_hp = Hp; p.conv_row0 = p0;
  { static int sd = -1; if (sd < 0) { const char* e = getenv("ST355_CONV_SKIP_DEAD"); sd = (e && e[0] == '0') ? 0 : 1; } p.skip_dead = sd; }
  p.img_add = (const bf16*)img_add; p.img_add_stride = img_add_stride;
  // the first / last Wp+1 positions are border positions that no GEMM row covers: they keep the caller's zeros (grid buffers are allocated
  // zero-filled and no kernel ever writes a border position non-zero), which saves two memset launches per convolution
  ProfScope ps(stream, ST355_K_GEMM, 2.0 * (double)Mc * Cout * taps * Cin, 2.0 * ((double)Mtot * Cin + (double)Cout * taps * Cin + (double)Mtot * Cout * (residual ? 2 : 1)),
               "CONV%d %dx%dx%d b%d %dx%d", taps, (int)Mc, Cout, taps * Cin, B, H, W);
  if (gemm_impl_choice() >= 4 && p4_tiles(p) >= min_tiles_256()) {
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p4_tiles(p);
    return residual ? launch_pq_conv<ST355_EPI_ADD>(stream, g, g.tiles0) : launch_pq_conv<ST355_EPI_NONE>(stream, g, g.tiles0);
  }
  return residual ? launch_s2<ST355_EPI_ADD>(stream, p) : launch_s2<ST355_EPI_NONE>(stream, p);
}

// weight gradient of the same convolution: dw[co, tap*Cin + ci] = sum_pos dy[pos, co] * x[pos + shift(tap), ci]  (one TN GEMM per tap;
// border rows of dy are zero, so rounding the contraction up to 64 rows only adds zero terms)
extern "C" int st355_conv_wgrad_bf16(void* stream, const void* x, const void* dy, void* dw, int B, int H, int W, int Cin, int Cout, int taps,
                                     int accumulate, void* workspace, int64_t workspace_bytes) {
  ST_REQUIRE(x && dy && dw && (taps == 1 || taps == 9), "conv_wgrad: bad args");
  const int Hp = H + 2, Wp = W + 2;
  const int64_t Mtot = (int64_t)B * Hp * Wp, p0 = Wp + 1;
  const int64_t Mc = ((Mtot - 2 * p0 + 63) / 64) * 64;
  if (taps == 9)      // all nine taps in one launch: R = x from the grid start (p0 + shift(tap) >= 0 for every tap)
    return gemm_tn_impl(stream, (const bf16*)dy + p0 * Cout, Cout, (const bf16*)x, Cin, dw, (int64_t)9 * Cin, Mc, Cout, Cin, accumulate, workspace, workspace_bytes, 9, Wp);
  int rc = st355_gemm_tn_bf16(stream, (const bf16*)dy + p0 * Cout, Cout, (const bf16*)x + p0 * Cin, Cin, dw, Cin, Mc, Cout, Cin, accumulate, workspace, workspace_bytes);
  if (rc) return rc;
  return ST355_OK;
}

extern "C" int st355_gemm_bf16_grouped(void* stream, const st355_gemm_args* args, int count) {
  ST_REQUIRE(args && count >= 1, "gemm_grouped: bad args");
  for (int i = 0; i < count; i++) {
    int rc = validate(&args[i]);
    if (rc) return rc;
    ST_REQUIRE(args[i].epilogue == args[0].epilogue, "gemm_grouped: all problems must share one epilogue kind");
  }
  int i = 0;
  while (i < count) {
    if (args[i].epilogue == ST355_EPI_GEGLU || args[i].epilogue == ST355_EPI_GEGLU_GRAD) {      // (one launch per problem: these two exist as plain problems only)
      ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]), gemm_bytes(&args[i]), "%dx%dx%d+%d e%d g", args[i].M, args[i].N, args[i].K, args[i].K2, args[i].epilogue);
      int rc = run_one(stream, &args[i]);
      if (rc) return rc;
      i += 1;
      continue;
    }
    if (args[i].epilogue == ST355_EPI_HEADS) {             // img + txt projections of one joint attention: one grid (or the last, odd problem alone)
      GemmGroup g;
      const int n2 = i + 1 < count ? 2 : 1;
      g.p[0] = to_p(&args[i]); g.p[1] = to_p(&args[i + n2 - 1]);
      g.tiles0 = p4_tiles(g.p[0]);
      const int tiles = g.tiles0 + (n2 == 2 ? p4_tiles(g.p[1]) : 0);
      ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]) + (n2 == 2 ? gemm_flops(&args[i + 1]) : 0.0), gemm_bytes(&args[i]) + (n2 == 2 ? gemm_bytes(&args[i + 1]) : 0.0),
                   "%d&%dx%dx%d+%d e%d", args[i].M, n2 == 2 ? args[i + 1].M : 0, args[i].N, args[i].K, args[i].K2, args[i].epilogue);
      int rc = launch_pq<ST355_EPI_HEADS>(stream, g, tiles);
      if (rc) return rc;
      i += n2;
      continue;
    }
    if (i + 1 < count && args[i].epilogue == ST355_EPI_QK_NORM_ROPE) {      // img + txt projections of one block: one grid
      GemmGroup g;
      g.p[0] = to_p(&args[i]); g.p[1] = to_p(&args[i + 1]);
      g.tiles0 = p4_tiles(g.p[0]);
      const int tiles = g.tiles0 + p4_tiles(g.p[1]);
      ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]) + gemm_flops(&args[i + 1]), gemm_bytes(&args[i]) + gemm_bytes(&args[i + 1]),
                   "%d&%dx%dx%d+%d e%d", args[i].M, args[i + 1].M, args[i].N, args[i].K, args[i].K2, args[i].epilogue);
      int rc = launch_pq<ST355_EPI_QK_NORM_ROPE>(stream, g, tiles);
      if (rc) return rc;
      i += 2;
      continue;
    }
    if (i + 1 < count && gemm_impl_choice() >= 2) {
      GemmGroup g;
      g.p[0] = to_p(&args[i]); g.p[1] = to_p(&args[i + 1]);
      g.tiles0 = p4_tiles(g.p[0]);
      const int tiles = g.tiles0 + p4_tiles(g.p[1]);
      if (pz_ok(g.p[0], g.tiles0) && pz_ok(g.p[1], tiles - g.tiles0)) {     // both fill the chip alone: nothing to gain from sharing a grid — two persistent launches
        for (int k = 0; k < 2; k++) {
          ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i + k]), gemm_bytes(&args[i + k]), "%dx%dx%d+%d e%d g", args[i + k].M, args[i + k].N, args[i + k].K,
                       args[i + k].K2, args[i + k].epilogue);
          int rc = run_one(stream, &args[i + k]);
          if (rc) return rc;
        }
        i += 2;
        continue;
      }
      // (measured r5, SD3-Medium full fine-tune at batch 8: launching a small text-stream partner apart from an image problem whose 768 tiles quantise to exactly 3
      // rounds — partner on the 128x128 schedule — did NOT pay: 319.6 vs 317.2 ms per step, GEMM class 152.2 vs 151.8 ms; the rule was removed.
      // profiles/r05_sd3_full_b8_ungroup_rule_ab.txt)
      if (tiles >= min_tiles_256()) {
        ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]) + gemm_flops(&args[i + 1]), gemm_bytes(&args[i]) + gemm_bytes(&args[i + 1]),
                     "%d&%dx%dx%d+%d e%d", args[i].M, args[i + 1].M, args[i].N, args[i].K, args[i].K2, args[i].epilogue);
        int rc;
        switch (args[i].epilogue) {
          case ST355_EPI_NONE: rc = launch_256<ST355_EPI_NONE>(stream, g, tiles); break;
          case ST355_EPI_GELU: rc = launch_256<ST355_EPI_GELU>(stream, g, tiles); break;
          case ST355_EPI_GATE_RESIDUAL: rc = launch_256<ST355_EPI_GATE_RESIDUAL>(stream, g, tiles); break;
          case ST355_EPI_MUL_GELU_GRAD: rc = launch_256<ST355_EPI_MUL_GELU_GRAD>(stream, g, tiles); break;
          default: rc = launch_256<ST355_EPI_ADD>(stream, g, tiles); break;
        }
        if (rc) return rc;
        i += 2;
        continue;
      }
    }
    if (i + 1 < count && gemm_impl_choice() >= 1) {
      GemmGroup g;
      g.p[0] = to_p(&args[i]); g.p[1] = to_p(&args[i + 1]);
      g.tiles0 = p3_tiles(g.p[0]);
      const int tiles = g.tiles0 + p3_tiles(g.p[1]);
      if (tiles >= 128) {
        ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]) + gemm_flops(&args[i + 1]), gemm_bytes(&args[i]) + gemm_bytes(&args[i + 1]),
                     "%d&%dx%dx%d+%d e%d", args[i].M, args[i + 1].M, args[i].N, args[i].K, args[i].K2, args[i].epilogue);
        int rc;
        switch (args[i].epilogue) {
          case ST355_EPI_NONE: rc = launch_p3<ST355_EPI_NONE>(stream, g, tiles); break;
          case ST355_EPI_GELU: rc = launch_p3<ST355_EPI_GELU>(stream, g, tiles); break;
          case ST355_EPI_GATE_RESIDUAL: rc = launch_p3<ST355_EPI_GATE_RESIDUAL>(stream, g, tiles); break;
          case ST355_EPI_MUL_GELU_GRAD: rc = launch_p3<ST355_EPI_MUL_GELU_GRAD>(stream, g, tiles); break;
          default: rc = launch_p3<ST355_EPI_ADD>(stream, g, tiles); break;
        }
        if (rc) return rc;
        i += 2;
        continue;
      }
    }
    ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]), gemm_bytes(&args[i]), "%dx%dx%d+%d e%d g", args[i].M, args[i].N, args[i].K, args[i].K2, args[i].epilogue);
    int rc = run_one(stream, &args[i]);
    if (rc) return rc;
    i += 1;
  }
  return ST355_OK;
}
