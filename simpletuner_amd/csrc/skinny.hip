// skinny.hip — rank-space LoRA gradients (K12 backward):  out[P, r] = alpha * sum_m L[m,P] * R[m,r]
//   dB = s * dY^T (x A^T)      (L = dY [M,N],  R = T [M,r])
//   dA^T = x^T (dY sB)         (L = x  [M,K],  R = U [M,r])
// The contraction runs over the TOKEN axis, which is the slow axis of both operands: the MFMA fragments (8 consecutive
// tokens per lane) are gathered from the ROW-MAJOR LDS tiles by gfx950's transposing LDS read (ds_read_b64_tr_b16: in a 16-lane
// group, lane 4i+s points at 4 contiguous bf16 of tile row i and lane j receives column j of the 4x16 block), so nothing is
// transposed in memory.  r <= 64 makes this a bandwidth-class op (reads L once).  Deterministic: split-M partials + a
// fixed-order reduce.  (k_skinny_tn is the first-generation fp32-VALU version, kept for A/B: ST355_SKINNY=1.)
#include <stdlib.h>
#include "common.h"

#define SK_PT 128   // columns of L per workgroup
#define SK_MC 256   // smallest split-M chunk: rows of L per workgroup (the launch picks 256 / 512 / 1024, see skinny_chunk)
#define SK_MS 64    // rows per LDS sub-tile of the MFMA kernel (r6: 32 -> 64: half as many load -> LDS -> MFMA round trips per workgroup, twice the bytes in flight:
                    // the kernel was latency-bound per sub-tile — 2.7 TB/s at the SDXL 32^2 level, rocprofv3 r06)
#define SK_MS1 32   // ... of the first-generation fp32-VALU kernel (A/B only)

template <int RN>
__global__ void __launch_bounds__(256) k_skinny_tn(const bf16* __restrict__ L, int64_t ldl, const bf16* __restrict__ R, int64_t ldr,
                                                  float* __restrict__ ws, int64_t M, int64_t P, int64_t seg_rows, int64_t seg_xl, int64_t seg_xr, int mc) {
  constexpr int RT = RN / 16;
  __shared__ __attribute__((aligned(16))) bf16 Ls[SK_MS1][SK_PT];
  __shared__ __attribute__((aligned(16))) bf16 Rs[SK_MS1][RN];
  const int tid = threadIdx.x;
  const int pg = tid & 15, rg = tid >> 4;
  const int64_t p0 = (int64_t)blockIdx.x * SK_PT;
  const int64_t mbase = (int64_t)blockIdx.y * mc;
  if (seg_rows) {          // segmented rows (st355_skinny_tn_seg): this 256-row chunk lies inside segment mbase / seg_rows; shift both operand bases to it
    const int64_t sgi = mbase / seg_rows;
    L += sgi * seg_xl;
    R += sgi * seg_xr;
  }
  float acc[8][RT];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < RT; j++) acc[i][j] = 0.f;

  for (int ms = 0; ms < mc; ms += SK_MS1) {
    // stage L sub-tile: 32 x 128 -> 512 chunks of 16 B
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int id = k * 256 + tid;
      const int row = id >> 4, c = id & 15;
      const int64_t m = mbase + ms + row;
      bf16x8 v;
      if (m < M && p0 + c * 8 < P) v = *(const bf16x8*)(L + m * ldl + p0 + c * 8);
      else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = f2bf(0.f);
      }
      *(bf16x8*)(&Ls[row][c * 8]) = v;
    }
    if (tid < SK_MS1 * RN / 8) {
      const int row = tid / (RN / 8), c = tid % (RN / 8);
      const int64_t m = mbase + ms + row;
      bf16x8 v;
      if (m < M) v = *(const bf16x8*)(R + m * ldr + c * 8);
      else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = f2bf(0.f);
      }
      *(bf16x8*)(&Rs[row][c * 8]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int m = 0; m < SK_MS1; m++) {
      bf16x8 lv = *(const bf16x8*)(&Ls[m][pg * 8]);
      float rv[RT];
#pragma unroll
      for (int j = 0; j < RT; j++) rv[j] = bf2f(Rs[m][rg * RT + j]);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float lf = bf2f(lv[i]);
#pragma unroll
        for (int j = 0; j < RT; j++) acc[i][j] += lf * rv[j];
      }
    }
    __syncthreads();
  }
  float* w = ws + ((int64_t)blockIdx.y * P) * RN;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int64_t p = p0 + pg * 8 + i;
    if (p < P) {
#pragma unroll
      for (int j = 0; j < RT; j++) w[p * RN + rg * RT + j] = acc[i][j];
    }
  }
}


// LDS pitches = 64 B (mod 256 B): the four tile rows a 32-lane half touches sit in four disjoint 16-bank windows
#define SK_LP (SK_PT + 32)      // 160 elements = 320 B
template <int RN>
__global__ void __launch_bounds__(256) k_skinny_tn_mfma(const bf16* __restrict__ L, int64_t ldl, const bf16* __restrict__ R, int64_t ldr,
                                                       float* __restrict__ ws, int64_t M, int64_t P, int64_t seg_rows, int64_t seg_xl, int64_t seg_xr, int mc) {
  constexpr int RT = RN / 32;                  // 32-wide r blocks
  constexpr int RP = (RN == 32) ? 32 : (RN == 64 ? 96 : 160);     // R tile pitch in elements (64 B / 192 B / 320 B)
  __shared__ __attribute__((aligned(16))) bf16 Ls[SK_MS * SK_LP];
  __shared__ __attribute__((aligned(16))) bf16 Rs[SK_MS * RP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * SK_PT;
  const int64_t mbase = (int64_t)blockIdx.y * mc;
  if (seg_rows) {          // segmented rows (st355_skinny_tn_seg): this 256-row chunk lies inside segment mbase / seg_rows; shift both operand bases to it
    const int64_t sgi = mbase / seg_rows;
    L += sgi * seg_xl;
    R += sgi * seg_xr;
  }
  f32x16 acc[RT];
#pragma unroll
  for (int j = 0; j < RT; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  // transposing-read geometry: group g = lane>>4, i = (lane>>2)&3 (tile row within the 4-row block), s = lane&3 (4-column segment)
  const int g = lane >> 4, ti = (lane >> 2) & 3, tsg = lane & 3;
  const int row_l = 8 * (g >> 1) + ti;                                   // + 16*ks + 4*rd
  const int a_col = 32 * wv + 16 * (g & 1) + 4 * tsg;                    // this wave's 32 columns of the L tile
  const int b_col = 16 * (g & 1) + 4 * tsg;                              // + 32*j

  constexpr int RCH = (SK_MS * RN / 8 + 255) / 256;      // 16-byte R chunks per thread (1; 2 for RN = 128)
  constexpr int LCH = SK_MS / 16;                        // 16-byte L chunks per thread (SK_MS rows x 16 chunks over 256 threads)
  bf16x8 lreg[LCH], rreg[RCH];
  auto load = [&](int ms) {
#pragma unroll
    for (int k = 0; k < LCH; k++) {
      const int id = k * 256 + tid;
      const int row = id >> 4, c = id & 15;
      const int64_t m = mbase + ms + row;
      if (m < M && p0 + c * 8 < P) lreg[k] = *(const bf16x8*)(L + m * ldl + p0 + c * 8);
      else {
#pragma unroll
        for (int j = 0; j < 8; j++) lreg[k][j] = f2bf(0.f);
      }
    }
#pragma unroll
    for (int q = 0; q < RCH; q++) {
      const int id = q * 256 + tid;
      if (id < SK_MS * RN / 8) {
        const int row = id / (RN / 8), c = id % (RN / 8);
        const int64_t m = mbase + ms + row;
        if (m < M) rreg[q] = *(const bf16x8*)(R + m * ldr + c * 8);
        else {
#pragma unroll
          for (int j = 0; j < 8; j++) rreg[q][j] = f2bf(0.f);
        }
      }
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int k = 0; k < LCH; k++) {
      const int id = k * 256 + tid;
      *(bf16x8*)(&Ls[(id >> 4) * SK_LP + (id & 15) * 8]) = lreg[k];
    }
#pragma unroll
    for (int q = 0; q < RCH; q++) {
      const int id = q * 256 + tid;
      if (id < SK_MS * RN / 8) *(bf16x8*)(&Rs[(id / (RN / 8)) * RP + (id % (RN / 8)) * 8]) = rreg[q];
    }
  };
  load(0);
  for (int ms = 0; ms < mc; ms += SK_MS) {
    store();
    __syncthreads();
    if (ms + SK_MS < mc) load(ms + SK_MS);          // next sub-tile's global loads fly under the MFMAs
#pragma unroll
    for (int ks = 0; ks < SK_MS / 16; ks++) {
      const bf16* lp = &Ls[(16 * ks + row_l) * SK_LP + a_col];
      const s16x4 a0 = lds_tr16(lp), a1 = lds_tr16(lp + 4 * SK_LP);
      bf16x8 af;
      *(s16x4*)&af = a0;
      *((s16x4*)&af + 1) = a1;
#pragma unroll
      for (int j = 0; j < RT; j++) {
        const bf16* rp = &Rs[(16 * ks + row_l) * RP + b_col + 32 * j];
        const s16x4 b0 = lds_tr16(rp), b1 = lds_tr16(rp + 4 * RP);
        bf16x8 bfr;
        *(s16x4*)&bfr = b0;
        *((s16x4*)&bfr + 1) = b1;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D[i = p][j = r]: lane -> r = lane&31, register reg -> p = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* w = ws + ((int64_t)blockIdx.y * P) * RN;
#pragma unroll
  for (int j = 0; j < RT; j++)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int64_t pp = p0 + 32 * wv + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (pp < P) w[pp * RN + 32 * j + (lane & 31)] = acc[j][reg];
    }
}

// fixed-order reduction of the split-M partials: four lanes per output element, lane q sums chunks q, q + 4, q + 8, ... in order, then a two-step
// xor-shuffle tree (deterministic; a single thread walking 36-144 chunks one dependent load after the other made this kernel slower than the MFMA
// kernel it follows on small problems: 13.8 us against 11.9 us per call in the SDXL-LoRA step, rocprofv3 r02)
#define SK_RED_LANES 4
__global__ void __launch_bounds__(256) k_skinny_reduce(const float* __restrict__ ws, float* __restrict__ out, int64_t so_p, int64_t so_r,
                                                      int64_t P, int RN, int r_used, int nchunks, float alpha, int accumulate) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / SK_RED_LANES;
  const int q = (int)(t % SK_RED_LANES);
  const bool live = i < P * r_used;
  const int64_t p = live ? i / r_used : 0;
  const int r = live ? (int)(i % r_used) : 0;
  float s = 0.f;
  if (live)
    for (int c = q; c < nchunks; c += SK_RED_LANES) s += ws[((int64_t)c * P + p) * RN + r];
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (live && q == 0) {
    float* o = out + p * so_p + r * so_r;
    *o = (accumulate ? *o : 0.f) + alpha * s;
  }
}

// Rows per workgroup.  Every chunk leaves a [P, Rn] fp32 partial that is written and read once more by the reduce: at 256 rows the partials of a
// 36 864-row product are half as many bytes again as the L operand itself.  So: the largest chunk that (a) divides the segment length when the operands
// are segmented and (b) still leaves >= 512 workgroups (two per CU) in the grid.
static int skinny_chunk(int64_t M, int64_t P, int64_t seg_rows) {
  const int64_t pt = cdiv64(P, SK_PT);
  for (int mc = 1024; mc > SK_MC; mc >>= 1)
    if ((seg_rows == 0 || seg_rows % mc == 0) && cdiv64(M, mc) * pt >= 512) return mc;
  return SK_MC;
}
extern "C" size_t st355_skinny_tn_workspace(int64_t M, int64_t P, int Rn) {
  return (size_t)cdiv64(M, SK_MC) * (size_t)P * (size_t)Rn * sizeof(float);
}

// seg_rows > 0: logical row m of L / R lives at physical row (m / seg_rows) * seg_l + m % seg_rows (resp. seg_r; 0 = compact) — the same segmented-row
// view as st355_gemm_args.seg_rows, so the rank-space gradients read the per-sample row blocks of a joint [B, S, *] buffer in place.
extern "C" int st355_skinny_tn_seg(void* stream, const void* L, int64_t ldl, const void* R, int64_t ldr, float* out, int64_t so_p,
                                   int64_t so_r, int64_t M, int64_t P, int Rn, int r_used, float alpha, int accumulate, void* workspace,
                                   int64_t seg_rows, int64_t seg_l, int64_t seg_r) {
  ST_REQUIRE(L && R && out && workspace, "skinny_tn: null pointer");
  ST_REQUIRE(seg_rows == 0 || (seg_rows > 0 && seg_rows % SK_MC == 0 && M % seg_rows == 0 && (seg_l == 0 || seg_l >= seg_rows) && (seg_r == 0 || seg_r >= seg_rows)),
             "skinny_tn: seg_rows (%lld) must be a multiple of %d that divides M; strides >= seg_rows", (long long)seg_rows, SK_MC);
  const int64_t seg_xl = (seg_rows && seg_l) ? (seg_l - seg_rows) * ldl : 0, seg_xr = (seg_rows && seg_r) ? (seg_r - seg_rows) * ldr : 0;
  ST_REQUIRE((Rn == 32 || Rn == 64) && r_used > 0 && r_used <= Rn, "skinny_tn: Rn must be 32 or 64 (got %d)", Rn);
  ST_REQUIRE(M > 0 && P > 0 && P % 8 == 0 && ldl % 8 == 0 && ldr % 8 == 0, "skinny_tn: bad shape");
  const int mc = skinny_chunk(M, P, seg_rows);
  const int nchunks = (int)cdiv64(M, mc);
  ProfScope ps(stream, ST355_K_SKINNY, 2.0 * M * P * Rn, 2.0 * M * (P + Rn) + 8.0 * nchunks * P * Rn);
  dim3 grid((unsigned)cdiv64(P, SK_PT), nchunks);
  static int gen = -1;
  if (gen < 0) { const char* e = getenv("ST355_SKINNY"); gen = (e && e[0] == '1') ? 1 : 2; }
  if (gen == 2 && Rn == 32)
    hipLaunchKernelGGL(k_skinny_tn_mfma<32>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)L, ldl, (const bf16*)R, ldr,
                       (float*)workspace, M, P, seg_rows, seg_xl, seg_xr, mc);
  else if (gen == 2)
    hipLaunchKernelGGL(k_skinny_tn_mfma<64>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)L, ldl, (const bf16*)R, ldr,
                       (float*)workspace, M, P, seg_rows, seg_xl, seg_xr, mc);
  else if (Rn == 32)
    hipLaunchKernelGGL(k_skinny_tn<32>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)L, ldl, (const bf16*)R, ldr,
                       (float*)workspace, M, P, seg_rows, seg_xl, seg_xr, mc);
  else
    hipLaunchKernelGGL(k_skinny_tn<64>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)L, ldl, (const bf16*)R, ldr,
                       (float*)workspace, M, P, seg_rows, seg_xl, seg_xr, mc);
  int rc = st355_check_launch("skinny_tn");
  if (rc) return rc;
  const int64_t n = P * r_used * SK_RED_LANES;
  hipLaunchKernelGGL(k_skinny_reduce, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out,
                     so_p, so_r, P, Rn, r_used, nchunks, alpha, accumulate);
  return st355_check_launch("skinny_reduce");
}
// several adapters that share the L operand (the q / k / v adapters of one fused projection: dA_g = U_g^T x): ONE pass over L against the [M, 32 * nout]
// column blocks of R, block g reduced into its own output.  L is the big operand (226 MB at 36 864 x 3072): reading it once instead of nout times is the point.
struct SkinnyOuts { float* o[4]; };
__global__ void __launch_bounds__(256) k_skinny_reduce_multi(const float* __restrict__ ws, SkinnyOuts outs, int64_t so_p, int64_t so_r, int64_t P, int RN,
                                                            int r_used, int nout, int nchunks, float alpha, int accumulate) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / SK_RED_LANES;
  const int q = (int)(t % SK_RED_LANES);
  const bool live = i < P * r_used * nout;
  const int r = live ? (int)(i % r_used) : 0;
  const int g = live ? (int)((i / r_used) % nout) : 0;
  const int64_t p = live ? i / ((int64_t)r_used * nout) : 0;
  float s = 0.f;
  if (live)
    for (int c = q; c < nchunks; c += SK_RED_LANES) s += ws[((int64_t)c * P + p) * RN + 32 * g + r];      // the same lane / chunk order as k_skinny_reduce
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (live && q == 0) {
    float* o = outs.o[g] + p * so_p + r * so_r;
    *o = (accumulate ? *o : 0.f) + alpha * s;
  }
}
extern "C" int st355_skinny_tn_multi(void* stream, const void* L, int64_t ldl, const void* R, int64_t ldr, float* const* outs, int nout, int64_t so_p,
                                     int64_t so_r, int64_t M, int64_t P, int r_used, float alpha, int accumulate, void* workspace,
                                     int64_t seg_rows, int64_t seg_l, int64_t seg_r) {
  ST_REQUIRE(L && R && outs && workspace && nout >= 1 && nout <= 4, "skinny_tn_multi: bad arguments (1..4 outputs)");
  for (int g = 0; g < nout; g++) ST_REQUIRE(outs[g] != nullptr, "skinny_tn_multi: null output %d", g);
  ST_REQUIRE(seg_rows == 0 || (seg_rows > 0 && seg_rows % SK_MC == 0 && M % seg_rows == 0 && (seg_l == 0 || seg_l >= seg_rows) && (seg_r == 0 || seg_r >= seg_rows)),
             "skinny_tn_multi: seg_rows (%lld) must be a multiple of %d that divides M; strides >= seg_rows", (long long)seg_rows, SK_MC);
  ST_REQUIRE(r_used > 0 && r_used <= 32 && M > 0 && P > 0 && P % 8 == 0 && ldl % 8 == 0 && ldr % 8 == 0 && ldr >= 128, "skinny_tn_multi: bad shape (R needs 128 columns)");
  const int64_t seg_xl = (seg_rows && seg_l) ? (seg_l - seg_rows) * ldl : 0, seg_xr = (seg_rows && seg_r) ? (seg_r - seg_rows) * ldr : 0;
  const int mc = skinny_chunk(M, P, seg_rows);
  const int nchunks = (int)cdiv64(M, mc);
  ProfScope ps(stream, ST355_K_SKINNY, 2.0 * M * P * 32.0 * nout, 2.0 * M * (P + 128) + 8.0 * nchunks * P * 128);
  dim3 grid((unsigned)cdiv64(P, SK_PT), nchunks);
  hipLaunchKernelGGL(k_skinny_tn_mfma<128>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)L, ldl, (const bf16*)R, ldr, (float*)workspace, M, P,
                     seg_rows, seg_xl, seg_xr, mc);
  int rc = st355_check_launch("skinny_tn_multi");
  if (rc) return rc;
  SkinnyOuts so{};
  for (int g = 0; g < nout; g++) so.o[g] = outs[g];
  const int64_t n = P * r_used * nout * SK_RED_LANES;
  hipLaunchKernelGGL(k_skinny_reduce_multi, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, so, so_p, so_r, P, 128,
                     r_used, nout, nchunks, alpha, accumulate);
  return st355_check_launch("skinny_reduce_multi");
}
extern "C" int st355_skinny_tn(void* stream, const void* L, int64_t ldl, const void* R, int64_t ldr, float* out, int64_t so_p,
                               int64_t so_r, int64_t M, int64_t P, int Rn, int r_used, float alpha, int accumulate, void* workspace) {
  return st355_skinny_tn_seg(stream, L, ldl, R, ldr, out, so_p, so_r, M, P, Rn, r_used, alpha, accumulate, workspace, 0, 0, 0);
}
