// reduce.hip — token-axis reductions of the full fine-tune backward (what autograd accumulates for biases and for the AdaLN
// modulation outputs, trainer.py:7126):
//     out[b, n] = sum over the tokens t of batch element b of  a[t, n] * (b_opt ? b_opt[t, n] : 1)
//   * bias gradients            db = colsum(dY)                       (one "batch" spanning all rows)
//   * modulation shift / gate   dshift_b = colsum_b(dY),  dgate_b = colsum_b(dOut * y_branch)
//   * modulation scale          dscale_b = colsum_b(dY * xhat);  with the saved n = xhat (1+scale) + shift this is
//                               (colsum_b(dY * n) - shift_b * dshift_b) / (1 + scale_b)          (finalize mode 1)
// HBM-bound streaming passes: a lane owns 8 consecutive columns (16-byte loads), a wave walks rows, partial sums go through a fixed
// two-level tree (waves -> LDS -> row chunks -> finalize kernel): deterministic, no atomics.
#include "common.h"

#define RD_ROWS 64      // rows per workgroup (16 per wave)
#define RD_COLS 512     // columns per workgroup (64 lanes x 8)

__global__ void __launch_bounds__(256) k_colsum_prod(const bf16* __restrict__ a, int64_t lda, const bf16* __restrict__ bm, int64_t ldb,
                                                    int64_t rows_per_batch, int N, float* __restrict__ ws, int nchunks) {
  __shared__ float part[4][RD_COLS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * RD_COLS + lane * 8;
  const int chunk = blockIdx.y, bi = blockIdx.z;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) acc[j] = 0.f;
  if (col < N) {
    const int64_t r_end = min((int64_t)(chunk + 1) * RD_ROWS, rows_per_batch);
    for (int64_t r = (int64_t)chunk * RD_ROWS + wv; r < r_end; r += 4) {
      const int64_t row = (int64_t)bi * rows_per_batch + r;
      const bf16x8 av = *(const bf16x8*)(a + row * lda + col);
      if (bm) {
        const bf16x8 bv = *(const bf16x8*)(bm + row * ldb + col);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += bf2f(av[j]) * bf2f(bv[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += bf2f(av[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; j++) part[wv][lane * 8 + j] = acc[j];
  __syncthreads();
  for (int i = threadIdx.x; i < RD_COLS; i += 256) {
    const int c = blockIdx.x * RD_COLS + i;
    if (c < N) ws[((int64_t)bi * nchunks + chunk) * N + c] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
  }
}

// mode 0: out = S ; mode 1: out = (S - shift * prev) / (1 + scale)   (prev = the matching dshift row, fp32)
__global__ void __launch_bounds__(256) k_colsum_finalize(const float* __restrict__ ws, int nchunks, int N, float* __restrict__ out,
                                                        int64_t out_stride, int mode, const float* __restrict__ prev, int64_t prev_stride,
                                                        const bf16* __restrict__ shift, const bf16* __restrict__ scale, int64_t mod_stride,
                                                        int accumulate) {
  // 32 columns x 8 chunk-slices per block: the chunk loop is 8x shorter and still summed in a fixed order
  __shared__ float red[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5, bi = blockIdx.y;
  float part = 0.f;
  if (c < N)
    for (int k = sl; k < nchunks; k += 8) part += ws[((int64_t)bi * nchunks + k) * N + c];
  red[sl][threadIdx.x & 31] = part;
  __syncthreads();
  if (sl != 0 || c >= N) return;
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 8; q++) s += red[q][threadIdx.x & 31];
  if (mode == 1) {
    const float sh = bf2f(shift[bi * mod_stride + c]), sc = 1.f + bf2f(scale[bi * mod_stride + c]);
    s = (s - sh * prev[bi * prev_stride + c]) / (fabsf(sc) > 1e-6f ? sc : (sc < 0.f ? -1e-6f : 1e-6f));
  }
  float* o = out + bi * out_stride + c;
  *o = accumulate ? *o + s : s;
}

extern "C" size_t st355_colsum_workspace(int64_t rows, int N, int64_t rows_per_batch) {
  const int64_t nb = rows / rows_per_batch;
  return (size_t)nb * (size_t)cdiv64(rows_per_batch, RD_ROWS) * (size_t)N * sizeof(float);
}

extern "C" int st355_colsum_prod(void* stream, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t rows, int N,
                                 int64_t rows_per_batch, float* out, int64_t out_stride, int mode, const float* prev, int64_t prev_stride,
                                 const void* shift, const void* scale, int64_t mod_stride, int accumulate, void* workspace) {
  ST_REQUIRE(a && out && workspace && rows > 0 && N > 0 && rows_per_batch > 0 && rows % rows_per_batch == 0, "colsum_prod: bad args");
  ST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && (!b || ldb % 8 == 0) && ((uintptr_t)a % 16 == 0) && (!b || (uintptr_t)b % 16 == 0), "colsum_prod: alignment");
  ST_REQUIRE(mode == 0 || (mode == 1 && prev && shift && scale), "colsum_prod: mode 1 needs prev / shift / scale");
  const int nb = (int)(rows / rows_per_batch);
  const int nchunks = (int)cdiv64(rows_per_batch, RD_ROWS);
  ProfScope ps(stream, ST355_K_ELEMENTWISE, (b ? 2.0 : 1.0) * rows * N, (b ? 4.0 : 2.0) * rows * N);
  hipLaunchKernelGGL(k_colsum_prod, dim3((N + RD_COLS - 1) / RD_COLS, nchunks, nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, lda,
                     (const bf16*)b, ldb, rows_per_batch, N, (float*)workspace, nchunks);
  int rc = st355_check_launch("colsum_prod");
  if (rc) return rc;
  hipLaunchKernelGGL(k_colsum_finalize, dim3((N + 31) / 32, nb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nchunks, N, out,
                     out_stride, mode, prev, prev_stride, (const bf16*)shift, (const bf16*)scale, mod_stride, accumulate);
  return st355_check_launch("colsum_finalize");
}

// ---- bf16 2-D transpose (the K-major copies W^T the dgrad GEMMs read must follow the weights in a full fine-tune) ----
__global__ void __launch_bounds__(256) k_transpose_bf16(const bf16* __restrict__ src, int64_t lds_, bf16* __restrict__ dst, int64_t ldd,
                                                       int R, int Cn) {
  __shared__ bf16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {        // 64 rows x 8 chunks of 8 columns
    const int r = i >> 3, ch = (i & 7) * 8;
    if (r0 + r < R && c0 + ch < Cn) {
      const bf16x8 v = *(const bf16x8*)(src + (int64_t)(r0 + r) * lds_ + c0 + ch);
#pragma unroll
      for (int j = 0; j < 8; j++) tile[r][ch + j] = v[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {        // dst rows = src columns
    const int c = i >> 3, rh = (i & 7) * 8;
    if (c0 + c < Cn && r0 + rh < R) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = tile[rh + j][c];
      *(bf16x8*)(dst + (int64_t)(c0 + c) * ldd + r0 + rh) = v;
    }
  }
}
extern "C" int st355_transpose_bf16(void* stream, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols) {
  ST_REQUIRE(src && dst && rows > 0 && cols > 0 && rows % 8 == 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "transpose_bf16: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0, 4.0 * rows * cols);
  hipLaunchKernelGGL(k_transpose_bf16, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src,
                     (bf16*)dst, ld_dst, rows, cols);
  return st355_check_launch("transpose_bf16");
}


// ---- fp32-accumulating reduce of the chunks a rank receives in the all-to-all form of the gradient reduce-scatter (training/grad_sync.py) ----------
// out[i] = bf16( float(in[0][i]) + float(in[1][i]) + ... + float(in[W-1][i]) ), chunks summed in rank order: deterministic, one rounding.
// HBM-bound: (W + 1) * 2 bytes per output element; 16-byte accesses, grid-stride.
__global__ void __launch_bounds__(256) k_sum_chunks_bf16(const bf16* __restrict__ in, int W, int64_t n, bf16* __restrict__ out) {
  const int64_t n8 = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    for (int w = 0; w < W; w++) {
      const bf16x8 v = *(const bf16x8*)(in + (int64_t)w * n + i * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) acc[j] += bf2f(v[j]);
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(acc[j]);
    *(bf16x8*)(out + i * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {          // ragged tail (< 8 elements)
    const int64_t i = (n8 << 3) + threadIdx.x;
    float a = 0.f;
    for (int w = 0; w < W; w++) a += bf2f(in[(int64_t)w * n + i]);
    out[i] = f2bf(a);
  }
}

extern "C" int st355_sum_chunks_bf16(void* stream, const void* chunks, int world, int64_t n, void* out) {
  ST_REQUIRE(chunks && out && world >= 1 && n >= 0, "sum_chunks_bf16: bad args");
  ST_REQUIRE(((uintptr_t)chunks & 15) == 0 && ((uintptr_t)out & 15) == 0, "sum_chunks_bf16: 16-byte aligned buffers");
  if (n == 0) return ST355_OK;
  ST_REQUIRE(world == 1 || n % 8 == 0, "sum_chunks_bf16: the chunk length must keep every chunk 16-byte aligned (n % 8 == 0)");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, (double)(world + 1) * n * 2, "sum_chunks %dx%lld", world, (long long)n);
  const int64_t n8 = n >> 3;
  const unsigned grid = (unsigned)(n8 < 1 ? 1 : (n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
  hipLaunchKernelGGL(k_sum_chunks_bf16, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)chunks, world, n, (bf16*)out);
  return st355_check_launch("sum_chunks_bf16");
}
