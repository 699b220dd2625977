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
#include "common.h"

#define BK 64

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
};

struct GemmGroup {
  GemmP p[2];
  int tiles0;  // tiles of problem 0 (blocks with id >= tiles0 work on problem 1)
};

__device__ __forceinline__ void glds16(const bf16* gsrc, char* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

// grouped tile order: GROUP consecutive m-tiles share one W panel
__device__ __forceinline__ void tile_coords(int id, int nbm, int nbn, int& pm, int& pn) {
  const int GROUP = 8;
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
        for (int b = 0; b < 4; b++) o[b] = f2bf(v[b]);
        *(bf16x4*)(p.C + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
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
  const GemmP& p = g.p[pi];
  const int nbm = (p.M + P3_BM - 1) / P3_BM, nbn = (p.N + P3_BN - 1) / P3_BN;
  int pm, pn;
  tile_coords(id, nbm, nbn, pm, pn);
  const int m0 = pm * P3_BM, n0 = pn * P3_BN;
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

// =================================================================================================
// k_gemm_p4: 256x256 tile, BK=32, FOUR-slot LDS-DMA ring (4 x 32 KiB), 8 waves (2 x 4), wave tile 128 tokens x 64 features.
// Doubling the wave tile halves the LDS bytes read per MFMA (0.75 -> 6 fragments per 8 MFMAs vs 4 per 4), which is what
// bounds k_gemm_p3 (LDS ~80 % busy at 34 % MFMA utilisation).  Loads are issued three K-slots ahead; vmcnt(8) retires the
// oldest slot (4 LDS-DMAs per wave per slot) and leaves two slots in flight across the raw barrier.
// LDS rows are 64 B: chunk' = chunk ^ ((row>>2)&3) spreads a 16-lane ds_read_b128 group over all 16 slots of a 256-B bank row.
// =================================================================================================
#define P4_BM 256
#define P4_BN 256
#define P4_BK 32
#define P4_THREADS 512
#define P4_XBYTES (P4_BM * P4_BK * 2)       // 16 KiB
#define P4_SLOT (2 * P4_XBYTES)             // 32 KiB
#define P4_LDS (4 * P4_SLOT)                // 128 KiB

template <int EPI>
__global__ void __launch_bounds__(P4_THREADS, 2) k_gemm_p4(GemmGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;     // 2 (tokens, 128 each) x 4 (features, 64 each)

  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int pi = id >= g.tiles0 ? 1 : 0;
  if (pi) id -= g.tiles0;
  const GemmP& p = g.p[pi];
  const int nbm = (p.M + P4_BM - 1) / P4_BM, nbn = (p.N + P4_BN - 1) / P4_BN;
  int pm, pn;
  tile_coords(id, nbm, nbn, pm, pn);
  const int m0 = pm * P4_BM, n0 = pn * P4_BN;
  const int nt1 = p.K / P4_BK;
  const int nt = nt1 + p.K2 / P4_BK;

  const bf16* A1 = p.A; const bf16* B1 = p.B; const bf16* A2 = p.A2; const bf16* B2 = p.B2;
  const int64_t la2 = p.lda2, lb2 = p.ldb2;
  const int M = p.M, N = p.N;
  // staging: one LDS-DMA instruction = 16 rows x 64 B; this wave issues instructions 2wv, 2wv+1 of the X and of the W tile
  const int st_row = lane >> 2, st_cp = lane & 3;
  int64_t xo[2], wo[2];
  int xrow[2], wrow[2], sc[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int row = (wv * 2 + j) * 16 + st_row;
    sc[j] = (st_cp ^ ((row >> 2) & 3)) * 8;
    xrow[j] = min(m0 + row, M - 1);
    wrow[j] = min(n0 + row, N - 1);
    xo[j] = (int64_t)xrow[j] * p.lda + sc[j];
    wo[j] = (int64_t)wrow[j] * p.ldb + sc[j];
  }
  auto stage = [&](int t, int slot) {
    char* xs = smem + slot * P4_SLOT;
    char* ws = xs + P4_XBYTES;
    if (t < nt1) {
      const int k0 = t * P4_BK;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        glds16(A1 + xo[j] + k0, xs + (wv * 2 + j) * 1024);
        glds16(B1 + wo[j] + k0, ws + (wv * 2 + j) * 1024);
      }
    } else {
      const int k0 = (t - nt1) * P4_BK;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        glds16(A2 + (int64_t)xrow[j] * la2 + sc[j] + k0, xs + (wv * 2 + j) * 1024);
        glds16(B2 + (int64_t)wrow[j] * lb2 + sc[j] + k0, ws + (wv * 2 + j) * 1024);
      }
    }
  };

  int w_off[2], w_sw[2], x_off[4], x_sw[4];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int wr = wn * 64 + i * 32 + (lane & 31);
    w_off[i] = wr * 64; w_sw[i] = (wr >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int xr = wm * 128 + j * 32 + (lane & 31);
    x_off[j] = xr * 64; x_sw[j] = (xr >> 2) & 3;
  }
  const int khalf = lane >> 5;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // ---- prologue: up to three slots in flight ----
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  if (nt > 2) stage(2, 2);
  if (nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int t = 0; t < nt; t++) {
    if (t + 3 < nt) stage(t + 3, (cur + 3) & 3);     // the slot read during iteration t-1
    const char* xs = smem + cur * P4_SLOT;
    const char* ws = xs + P4_XBYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = 2 * ks + khalf;
      bf16x8 wf[2], xf[4];
#pragma unroll
      for (int i = 0; i < 2; i++) wf[i] = *(const bf16x8*)(ws + w_off[i] + ((c ^ w_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < 4; j++) xf[j] = *(const bf16x8*)(xs + x_off[j] + ((c ^ x_sw[j]) << 4));
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      // retire slot t+1; slots t+2, t+3 (4 LDS-DMAs each) stay in flight across the barrier
      if (t + 3 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    cur = (cur + 1) & 3;
  }
  gemm_epilogue<EPI, 2, 4>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
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
__global__ void __launch_bounds__(S2_THREADS, 2) k_gemm_s2(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int nbm = (p.M + S2_BM - 1) / S2_BM, nbn = (p.N + S2_BN - 1) / S2_BN;
  int pm, pn;
  tile_coords(xcd_remap(blockIdx.x, nbm * nbn), nbm, nbn, pm, pn);
  const int m0 = pm * S2_BM, n0 = pn * S2_BN;
  const int nt1 = p.K / BK;
  const int nt = nt1 + p.K2 / BK;
  const int st_row = lane >> 3, st_cp = lane & 7;
  auto stage = [&](int t, int buf) {
    const bf16* Ap; const bf16* Bp; int64_t la, lb; int k0;
    if (t < nt1) { Ap = p.A; la = p.lda; Bp = p.B; lb = p.ldb; k0 = t * BK; }
    else { Ap = p.A2; la = p.lda2; Bp = p.B2; lb = p.ldb2; k0 = (t - nt1) * BK; }
    char* xs = smem + buf * S2_STAGE;
    char* ws = xs + S2_TILE;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int r0 = (wv * 4 + j) * 8;
      const int row = r0 + st_row;
      const int c = st_cp ^ ((row >> 1) & 7);
      glds16(Ap + (int64_t)min(m0 + row, p.M - 1) * la + k0 + c * 8, xs + r0 * 128);
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
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < nt; t++) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    const char* xs = smem + buf * S2_STAGE;
    mma_tile(xs, xs + S2_TILE, x_off, x_sw, w_off, w_sw, khalf, acc);
    __syncthreads();
  }
  gemm_epilogue<EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
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
  if (a->epilogue == ST355_EPI_GELU && a->aux_out) ST_REQUIRE(a->ld_aux_out % 4 == 0, "gemm: ld_aux_out must be a multiple of 4");
  ST_REQUIRE(a->epilogue >= 0 && a->epilogue <= ST355_EPI_ADD, "gemm: unknown epilogue %d", a->epilogue);
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
  return p;
}

static double gemm_flops(const st355_gemm_args* a) { return 2.0 * a->M * a->N * ((double)a->K + a->K2); }
static double gemm_bytes(const st355_gemm_args* a) {
  return 2.0 * ((double)a->M * (a->K + a->K2) + (double)a->N * (a->K + a->K2) + (double)a->M * a->N);
}

template <int EPI>
static int launch_s2(void* stream, const GemmP& p) {
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_gemm_s2<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS); attr_set = true; }
  const int nbm = (p.M + S2_BM - 1) / S2_BM, nbn = (p.N + S2_BN - 1) / S2_BN;
  hipLaunchKernelGGL(k_gemm_s2<EPI>, dim3(nbm * nbn), dim3(S2_THREADS), S2_LDS, (hipStream_t)stream, p);
  return st355_check_launch("gemm_s2");
}

template <int EPI>
static int launch_p3(void* stream, const GemmGroup& g, int tiles) {
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_gemm_p3<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, P3_LDS); attr_set = true; }
  hipLaunchKernelGGL(k_gemm_p3<EPI>, dim3(tiles), dim3(P3_THREADS), P3_LDS, (hipStream_t)stream, g);
  return st355_check_launch("gemm_p3");
}

template <int EPI>
static int launch_p4(void* stream, const GemmGroup& g, int tiles) {
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)k_gemm_p4<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS); attr_set = true; }
  hipLaunchKernelGGL(k_gemm_p4<EPI>, dim3(tiles), dim3(P4_THREADS), P4_LDS, (hipStream_t)stream, g);
  return st355_check_launch("gemm_p4");
}
static int p4_tiles(const GemmP& p) { return ((p.M + P4_BM - 1) / P4_BM) * ((p.N + P4_BN - 1) / P4_BN); }

static int p3_tiles(const GemmP& p) { return ((p.M + P3_BM - 1) / P3_BM) * ((p.N + P3_BN - 1) / P3_BN); }

static int gemm_impl_choice() {
  static int c = -1;
  if (c < 0) {
    const char* e = getenv("ST355_GEMM_IMPL");   // A/B testing: "s2" small tile only, "p3" no 256x256 schedule, default all
    c = (e && e[0] == 's') ? 0 : ((e && e[0] == 'p' && e[1] == '3') ? 1 : 2);
  }
  return c;
}

#define DISPATCH_EPI(fn, epi, ...)                                                           \
  switch (epi) {                                                                             \
    case ST355_EPI_NONE: return fn<ST355_EPI_NONE>(__VA_ARGS__);                             \
    case ST355_EPI_GELU: return fn<ST355_EPI_GELU>(__VA_ARGS__);                             \
    case ST355_EPI_GATE_RESIDUAL: return fn<ST355_EPI_GATE_RESIDUAL>(__VA_ARGS__);           \
    case ST355_EPI_MUL_GELU_GRAD: return fn<ST355_EPI_MUL_GELU_GRAD>(__VA_ARGS__);           \
    default: return fn<ST355_EPI_ADD>(__VA_ARGS__);                                          \
  }

static int run_one(void* stream, const st355_gemm_args* a) {
  GemmP p = to_p(a);
  // the deep-pipelined schedule needs enough tiles to fill 256 CUs; tiny problems stay on the 128x128 schedule
  // 256x256 tiles only when they (nearly) fill the 256 CUs at one workgroup each
  if (gemm_impl_choice() == 2 && p4_tiles(p) >= 200) {
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p4_tiles(p);
    DISPATCH_EPI(launch_p4, a->epilogue, stream, g, g.tiles0);
  }
  const bool big = gemm_impl_choice() >= 1 && p.M > 128 && p3_tiles(p) >= 128;
  if (big) {
    GemmGroup g;
    g.p[0] = p; g.p[1] = p; g.tiles0 = p3_tiles(p);
    DISPATCH_EPI(launch_p3, a->epilogue, stream, g, g.tiles0);
  }
  DISPATCH_EPI(launch_s2, a->epilogue, stream, p);
}

extern "C" int st355_gemm_bf16(void* stream, const st355_gemm_args* a) {
  int rc = validate(a);
  if (rc) return rc;
  ProfScope ps(stream, ST355_K_GEMM, gemm_flops(a), gemm_bytes(a));
  return run_one(stream, a);
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
    if (i + 1 < count && gemm_impl_choice() == 2) {
      GemmGroup g;
      g.p[0] = to_p(&args[i]); g.p[1] = to_p(&args[i + 1]);
      g.tiles0 = p4_tiles(g.p[0]);
      const int tiles = g.tiles0 + p4_tiles(g.p[1]);
      if (tiles >= 200) {
        ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]) + gemm_flops(&args[i + 1]), gemm_bytes(&args[i]) + gemm_bytes(&args[i + 1]));
        int rc;
        switch (args[i].epilogue) {
          case ST355_EPI_NONE: rc = launch_p4<ST355_EPI_NONE>(stream, g, tiles); break;
          case ST355_EPI_GELU: rc = launch_p4<ST355_EPI_GELU>(stream, g, tiles); break;
          case ST355_EPI_GATE_RESIDUAL: rc = launch_p4<ST355_EPI_GATE_RESIDUAL>(stream, g, tiles); break;
          case ST355_EPI_MUL_GELU_GRAD: rc = launch_p4<ST355_EPI_MUL_GELU_GRAD>(stream, g, tiles); break;
          default: rc = launch_p4<ST355_EPI_ADD>(stream, g, tiles); break;
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
        ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]) + gemm_flops(&args[i + 1]), gemm_bytes(&args[i]) + gemm_bytes(&args[i + 1]));
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
    ProfScope ps(stream, ST355_K_GEMM, gemm_flops(&args[i]), gemm_bytes(&args[i]));
    int rc = run_one(stream, &args[i]);
    if (rc) return rc;
    i += 1;
  }
  return ST355_OK;
}
