// skinny.hip — rank-space LoRA gradients (K12 backward):  out[P, r] = alpha * sum_m L[m,P] * R[m,r]
//   dB = s * dY^T (x A^T)      (L = dY [M,N],  R = T [M,r])
//   dA^T = x^T (dY sB)         (L = x  [M,K],  R = U [M,r])
// The contraction runs over the TOKEN axis, which is the slow axis of both operands, so MFMA would need
// transposed operand images; r <= 64 makes this a bandwidth-class op (reads L once), done on the VALU from
// LDS tiles with fp32 accumulation.  Deterministic: split-M partials + a fixed-order reduce.
#include "common.h"

#define SK_PT 128   // columns of L per workgroup
#define SK_MC 256   // rows of L per workgroup (split-M chunk)
#define SK_MS 32    // rows per LDS sub-tile

template <int RN>
__global__ void __launch_bounds__(256) k_skinny_tn(const bf16* __restrict__ L, int64_t ldl, const bf16* __restrict__ R, int64_t ldr,
                                                  float* __restrict__ ws, int64_t M, int64_t P) {
  constexpr int RT = RN / 16;
  __shared__ __attribute__((aligned(16))) bf16 Ls[SK_MS][SK_PT];
  __shared__ __attribute__((aligned(16))) bf16 Rs[SK_MS][RN];
  const int tid = threadIdx.x;
  const int pg = tid & 15, rg = tid >> 4;
  const int64_t p0 = (int64_t)blockIdx.x * SK_PT;
  const int64_t mbase = (int64_t)blockIdx.y * SK_MC;
  float acc[8][RT];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < RT; j++) acc[i][j] = 0.f;

  for (int ms = 0; ms < SK_MC; ms += SK_MS) {
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
    if (tid < SK_MS * RN / 8) {
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
    for (int m = 0; m < SK_MS; m++) {
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

__global__ void __launch_bounds__(256) k_skinny_reduce(const float* __restrict__ ws, float* __restrict__ out, int64_t so_p, int64_t so_r,
                                                      int64_t P, int RN, int r_used, int nchunks, float alpha, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * r_used) return;
  const int64_t p = i / r_used;
  const int r = (int)(i % r_used);
  float s = 0.f;
  for (int c = 0; c < nchunks; c++) s += ws[((int64_t)c * P + p) * RN + r];
  float* o = out + p * so_p + r * so_r;
  *o = (accumulate ? *o : 0.f) + alpha * s;
}

extern "C" size_t st355_skinny_tn_workspace(int64_t M, int64_t P, int Rn) {
  return (size_t)cdiv64(M, SK_MC) * (size_t)P * (size_t)Rn * sizeof(float);
}

extern "C" int st355_skinny_tn(void* stream, const void* L, int64_t ldl, const void* R, int64_t ldr, float* out, int64_t so_p,
                               int64_t so_r, int64_t M, int64_t P, int Rn, int r_used, float alpha, int accumulate, void* workspace) {
  ST_REQUIRE(L && R && out && workspace, "skinny_tn: null pointer");
  ST_REQUIRE((Rn == 32 || Rn == 64) && r_used > 0 && r_used <= Rn, "skinny_tn: Rn must be 32 or 64 (got %d)", Rn);
  ST_REQUIRE(M > 0 && P > 0 && P % 8 == 0 && ldl % 8 == 0 && ldr % 8 == 0, "skinny_tn: bad shape");
  const int nchunks = (int)cdiv64(M, SK_MC);
  ProfScope ps(stream, ST355_K_SKINNY, 2.0 * M * P * Rn, 2.0 * M * (P + Rn) + 8.0 * nchunks * P * Rn);
  dim3 grid((unsigned)cdiv64(P, SK_PT), nchunks);
  if (Rn == 32)
    hipLaunchKernelGGL(k_skinny_tn<32>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)L, ldl, (const bf16*)R, ldr,
                       (float*)workspace, M, P);
  else
    hipLaunchKernelGGL(k_skinny_tn<64>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)L, ldl, (const bf16*)R, ldr,
                       (float*)workspace, M, P);
  int rc = st355_check_launch("skinny_tn");
  if (rc) return rc;
  const int64_t n = P * r_used;
  hipLaunchKernelGGL(k_skinny_reduce, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out,
                     so_p, so_r, P, Rn, r_used, nchunks, alpha, accumulate);
  return st355_check_launch("skinny_reduce");
}
