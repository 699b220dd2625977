// gemm.hip — bf16 MFMA GEMM for every dense contraction of the DiT step (K4, K8, K9, K11, K12):
//     C[M,N] = A[M,K] · B[N,K]^T  (+ A2[M,K2] · B2[N,K2]^T)   with fused epilogues.
//
// gfx950 design (cdna_hip_programming.md §5):
//   * 128(tokens) x 128(features) x 64(k) tile, 4 waves (2x2), each wave a 64x64 sub-tile built from
//     v_mfma_f32_32x32x16_bf16 (2x2 MFMA tiles, 64 fp32 accumulators / lane).
//   * operands are SWAPPED inside the MFMA (weights = MFMA "A", activations = MFMA "B") so that a lane's 4
//     consecutive accumulator registers are 4 consecutive FEATURES of one token: row-major C gets 8-byte
//     packed stores and the epilogue reads bias / gate / residual with the same contiguity.
//   * global -> LDS by direct LDS-DMA (global_load_lds_dwordx4, 16 B / lane); the LDS image stays
//     lane-linear and the bank-conflict swizzle  chunk' = chunk ^ ((row>>1)&7)  is applied on the SOURCE
//     address and again on the ds_read_b128 address (rule 21: both-sides-or-neither).
//   * double-buffered LDS (2 x 32 KiB), one barrier per K-tile, 2 workgroups resident per CU.
//   * the LoRA low-rank term is a K-extension of the same accumulators (K2 extra columns), i.e. the fused
//     base+low-rank GEMM of the north star: y = [x | xA^T] · [W | sB]^T.
//   * XCD-aware bijective block remap + grouped tile order for L2 locality.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256
#define TILE_BYTES (128 * BK * 2)        // 16 KiB per operand tile
#define STAGE_BYTES (2 * TILE_BYTES)     // X tile + W tile
#define GEMM_LDS (2 * STAGE_BYTES)       // double buffered: 64 KiB

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

__device__ __forceinline__ void glds16(const bf16* gsrc, char* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

template <int EPI>
__global__ void __launch_bounds__(GEMM_THREADS, 2) k_gemm_bf16(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int wm = wv >> 1, wn = wv & 1;

  // ---- tile coordinates: XCD remap, then grouped order (GROUP m-tiles share one W panel) ----
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int id = xcd_remap(blockIdx.x, nwg);
  const int GROUP = 8;
  const int width = GROUP * nbn;
  const int group_id = id / width;
  const int first_m = group_id * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  const int pm = first_m + (id % width) % gsz;
  const int pn = (id % width) / gsz;
  const int m0 = pm * BM, n0 = pn * BN;

  const int nt1 = p.K / BK;
  const int nt = nt1 + p.K2 / BK;

  // per-lane staging coordinates: instruction j of this wave covers tile rows (wv*4+j)*8 .. +8
  const int st_row = lane >> 3;                       // row inside the 8-row group
  const int st_cp = lane & 7;                         // physical 16-B chunk inside the 128-B row

  auto stage = [&](int t, int buf) {
    const bf16* Ap; const bf16* Bp; int64_t la, lb; int k0;
    if (t < nt1) { Ap = p.A; la = p.lda; Bp = p.B; lb = p.ldb; k0 = t * BK; }
    else { Ap = p.A2; la = p.lda2; Bp = p.B2; lb = p.ldb2; k0 = (t - nt1) * BK; }
    char* xs = smem + buf * STAGE_BYTES;
    char* ws = xs + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int r0 = (wv * 4 + j) * 8;
      const int row = r0 + st_row;
      const int c = st_cp ^ ((row >> 1) & 7);         // logical chunk fetched into physical slot st_cp
      const int gm = min(m0 + row, p.M - 1);
      const int gn = min(n0 + row, p.N - 1);
      glds16(Ap + (int64_t)gm * la + k0 + c * 8, xs + r0 * 128);
      glds16(Bp + (int64_t)gn * lb + k0 + c * 8, ws + r0 * 128);
    }
  };

  // per-lane fragment coordinates
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
    const char* xs = smem + buf * STAGE_BYTES;
    const char* ws = xs + TILE_BYTES;
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
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: D[n_local][m_local]; lane: token m = lane&31, features 8a + 4*(lane>>5) + b ----
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int m = m0 + wm * 64 + j * 32 + (lane & 31);
    if (m >= p.M) continue;
    const int64_t bidx = (EPI == ST355_EPI_GATE_RESIDUAL) ? (int64_t)(m / p.rows_per_batch) : 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int n = n0 + wn * 64 + i * 32 + 8 * a + 4 * khalf;
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

template <int EPI>
static int launch_gemm(void* stream, const GemmP& p) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_gemm_bf16<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    attr_set = true;
  }
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(k_gemm_bf16<EPI>, dim3(nbm * nbn), dim3(GEMM_THREADS), GEMM_LDS, (hipStream_t)stream, p);
  return st355_check_launch("gemm_bf16");
}

extern "C" int st355_gemm_bf16(void* stream, const st355_gemm_args* a) {
  ST_REQUIRE(a && a->A && a->B && a->C, "gemm: null pointer");
  ST_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "gemm: empty shape M=%d N=%d K=%d", a->M, a->N, a->K);
  ST_REQUIRE(a->K % BK == 0 && a->K2 % BK == 0, "gemm: K (%d) and K2 (%d) must be multiples of 64", a->K, a->K2);
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

  GemmP p;
  p.A = (const bf16*)a->A; p.lda = a->lda; p.B = (const bf16*)a->B; p.ldb = a->ldb;
  p.A2 = (const bf16*)a->A2; p.lda2 = a->lda2; p.B2 = (const bf16*)a->B2; p.ldb2 = a->ldb2;
  p.C = (bf16*)a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K; p.K2 = a->K2;
  p.bias = (const bf16*)a->bias;
  p.aux_out = (bf16*)a->aux_out; p.ld_aux_out = a->ld_aux_out;
  p.aux_in = (const bf16*)a->aux_in; p.ld_aux_in = a->ld_aux_in;
  p.gate = (const bf16*)a->gate; p.gate_stride = a->gate_stride; p.rows_per_batch = a->rows_per_batch;

  const double flops = 2.0 * a->M * a->N * ((double)a->K + a->K2);
  const double bytes = 2.0 * ((double)a->M * (a->K + a->K2) + (double)a->N * (a->K + a->K2) + (double)a->M * a->N);
  ProfScope ps(stream, ST355_K_GEMM, flops, bytes);
  switch (a->epilogue) {
    case ST355_EPI_NONE: return launch_gemm<ST355_EPI_NONE>(stream, p);
    case ST355_EPI_GELU: return launch_gemm<ST355_EPI_GELU>(stream, p);
    case ST355_EPI_GATE_RESIDUAL: return launch_gemm<ST355_EPI_GATE_RESIDUAL>(stream, p);
    case ST355_EPI_MUL_GELU_GRAD: return launch_gemm<ST355_EPI_MUL_GELU_GRAD>(stream, p);
    case ST355_EPI_ADD: return launch_gemm<ST355_EPI_ADD>(stream, p);
    default: st355_set_error("gemm: unknown epilogue %d", a->epilogue); return ST355_EINVAL;
  }
}
