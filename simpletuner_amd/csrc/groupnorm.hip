// groupnorm.hip — GroupNorm(32 groups, affine) [+ SiLU] over a zero-bordered NHWC grid buffer (conv.hip), forward and backward.
//   reference seam: diffusers ResnetBlock2D (norm1 / norm2 + SiLU, eps 1e-5), Transformer2DModel.norm (eps 1e-6, no activation),
//   UNet conv_norm_out + conv_act — the UNet the reference calls at sdxl/model.py:350-367, sd1x/model.py:224-270 (un-vendored diffusers).
// HBM-bound: forward = stats pass (read x) + apply pass (read x, write y); backward = stats pass (read x, dy) + apply pass (read x, dy, write dx).
// Deterministic: a thread owns a fixed 8-channel chunk and walks rows; row-lanes are combined through LDS in a fixed order; chunk partials are
// combined by one wave per (image, group) in a fixed order.  Border positions hold zero, so statistics may sum over the whole padded image.
#include "common.h"

#define GN_THREADS 256
#define GN_MAXC 2560           // largest channel count (SDXL up-block concat 1280+1280)

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) { const float s = 1.f / (1.f + __expf(-z)); return s * (1.f + z * (1.f - s)); }

// token row of grid position pos (interior only); returns -1 on the border
__device__ __forceinline__ int64_t token_of(int64_t pos, int H, int W) {
  const int Wp = W + 2, Hp = H + 2;
  const int xx = (int)(pos % Wp), yy = (int)((pos / Wp) % Hp);
  const int64_t b = pos / ((int64_t)Wp * Hp);
  if (yy < 1 || yy > H || xx < 1 || xx > W) return -1;
  return (b * H + yy - 1) * W + xx - 1;
}

// MODE 0: partial[b][chunk][c] = (sum x, sum x^2)       MODE 1: (sum g, sum g*xhat),  g = dy * (silu ? silu'(z) : 1), z = xhat*gamma + beta
template <int MODE>
__global__ void __launch_bounds__(GN_THREADS) k_gn_stats(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ stats,
                                                       const bf16* __restrict__ gamma, const bf16* __restrict__ beta, float* __restrict__ partial, int H, int W,
                                                       int C, int rows_per_chunk, int nchunks, int silu, int dy_tokens) {
  extern __shared__ float red[];                  // [RT][cw*8][2] for the current channel window
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int rows_img = (H + 2) * (W + 2);
  const int r0 = chunk * rows_per_chunk, r1 = min(rows_img, r0 + rows_per_chunk);
  const int c8 = C / 8;
  const int tid = threadIdx.x;
  for (int w0 = 0; w0 < c8; w0 += GN_THREADS) {          // channel windows of <= 256 chunks
    const int cw = min(c8 - w0, GN_THREADS);
    const int RT = GN_THREADS / cw;
    const int ch = w0 + tid % cw, rl = tid / cw;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { s0[j] = 0.f; s1[j] = 0.f; }
    if (rl < RT) {
      float mu[8], rs[8], ga[8], be[8];
      if (MODE == 1) {
        const bf16x8 gv = *(const bf16x8*)(gamma + ch * 8), bv = *(const bf16x8*)(beta + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          mu[j] = stats[((int64_t)b * C + ch * 8 + j) * 2]; rs[j] = stats[((int64_t)b * C + ch * 8 + j) * 2 + 1];
          ga[j] = bf2f(gv[j]); be[j] = bf2f(bv[j]);
        }
      }
      // four rows per trip: all their requests leave before the first sum needs one (one request in flight per thread kept 12 KB per CU outstanding — the stats
      // passes ran at 2.0-3.5 TB/s, rocprofv3 r6); the sums are still taken row by row in the same order: bit-identical partials
      for (int r = r0 + rl; r < r1; r += 4 * RT) {
        bf16x8 xv[4], dv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int rr = r + u * RT;
          ok[u] = rr < r1;
          const int64_t pos = (int64_t)b * rows_img + (ok[u] ? rr : r);
          xv[u] = *(const bf16x8*)(x + pos * C + ch * 8);
          if (MODE == 1) {
            int64_t drow = pos;
            if (dy_tokens) { drow = token_of(pos, H, W); if (drow < 0) { ok[u] = false; drow = 0; } }
            dv[u] = *(const bf16x8*)(dy + drow * C + ch * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (!ok[u]) continue;
          if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) { const float v = bf2f(xv[u][j]); s0[j] += v; s1[j] += v * v; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const float xh = (bf2f(xv[u][j]) - mu[j]) * rs[j];
              float g = bf2f(dv[u][j]);
              if (silu) g *= silu_grad_f(xh * ga[j] + be[j]);
              s0[j] += g; s1[j] += g * xh;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) { red[((rl * cw + tid % cw) * 8 + j) * 2] = s0[j]; red[((rl * cw + tid % cw) * 8 + j) * 2 + 1] = s1[j]; }
    }
    __syncthreads();
    for (int c = tid; c < cw * 8; c += GN_THREADS) {
      float a0 = 0.f, a1 = 0.f;
      for (int q = 0; q < RT; q++) { a0 += red[((q * cw + c / 8) * 8 + (c & 7)) * 2]; a1 += red[((q * cw + c / 8) * 8 + (c & 7)) * 2 + 1]; }
      float* dst = partial + (((int64_t)b * nchunks + chunk) * C + w0 * 8 + c) * 2;
      dst[0] = a0; dst[1] = a1;
    }
    __syncthreads();
  }
}

// one wave per (image, group): mean / rstd  ->  stats[b][c] = (mean_g, rstd_g) expanded per channel
__global__ void __launch_bounds__(64) k_gn_finalize_fwd(const float* __restrict__ partial, float* __restrict__ stats, int C, int G, int nchunks, float count,
                                                      float eps) {
  const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int cg = C / G;
  float s0 = 0.f, s1 = 0.f;
  for (int i = lane; i < nchunks * cg; i += 64) {
    const int chunk = i / cg, c = g * cg + i % cg;
    const float* src = partial + (((int64_t)b * nchunks + chunk) * C + c) * 2;
    s0 += src[0]; s1 += src[1];
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  const float mean = s0 / count;
  const float var = fmaxf(s1 / count - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  for (int c = lane; c < cg; c += 64) { stats[((int64_t)b * C + g * cg + c) * 2] = mean; stats[((int64_t)b * C + g * cg + c) * 2 + 1] = rstd; }
}

// y = (x - mean) * rstd * gamma + beta  [-> SiLU]; grid output (border zero) or dense tokens
__global__ void __launch_bounds__(GN_THREADS) k_gn_apply_fwd(const bf16* __restrict__ x, const float* __restrict__ stats, const bf16* __restrict__ gamma,
                                                           const bf16* __restrict__ beta, bf16* __restrict__ y, int B, int H, int W, int C, int silu,
                                                           int out_tokens) {
  const int c8 = C / 8;
  const int rows_img = (H + 2) * (W + 2);
  const int64_t n = (int64_t)B * rows_img * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c8);
    const int64_t pos = i / c8;
    const int64_t tok = token_of(pos, H, W);
    const int b = (int)(pos / rows_img);
    bf16x8 o;
    if (tok < 0) {
      if (out_tokens) continue;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = f2bf(0.f);
    } else {
      const bf16x8 xv = *(const bf16x8*)(x + pos * C + ch * 8);
      const bf16x8 gv = *(const bf16x8*)(gamma + ch * 8), bv = *(const bf16x8*)(beta + ch * 8);
      const float* st = stats + ((int64_t)b * C + ch * 8) * 2;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float z = (bf2f(xv[j]) - st[2 * j]) * st[2 * j + 1] * bf2f(gv[j]) + bf2f(bv[j]);
        if (silu) z = silu_f(z);
        o[j] = f2bf(z);
      }
    }
    *(bf16x8*)(y + (out_tokens ? tok : pos) * C + ch * 8) = o;
  }
}

// one wave per (image, group): S1 = sum_c gamma_c A_c, S2 = sum_c gamma_c B_c  ->  coef[b][c] = (rstd*gamma_c, -rstd*S2/n, -rstd*S1/n)
__global__ void __launch_bounds__(64) k_gn_finalize_bwd(const float* __restrict__ partial, const float* __restrict__ stats, const bf16* __restrict__ gamma,
                                                      float* __restrict__ coef, int C, int G, int nchunks, float count) {
  const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int cg = C / G;
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < nchunks * cg; i += 64) {
    const int chunk = i / cg, c = g * cg + i % cg;
    const float* src = partial + (((int64_t)b * nchunks + chunk) * C + c) * 2;
    const float gm = bf2f(gamma[c]);
    s1 += gm * src[0]; s2 += gm * src[1];
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  const float rstd = stats[((int64_t)b * C + g * cg) * 2 + 1];
  for (int c = lane; c < cg; c += 64) {
    float* d = coef + ((int64_t)b * C + g * cg + c) * 3;
    d[0] = rstd * bf2f(gamma[g * cg + c]); d[1] = -rstd * s2 / count; d[2] = -rstd * s1 / count;
  }
}
// dgamma[c] (+)= sum_{b,chunk} B_c ; dbeta[c] (+)= sum_{b,chunk} A_c      (fixed order)
__global__ void __launch_bounds__(256) k_gn_param_grads(const float* __restrict__ partial, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                      int nchunks, int accumulate) {
  __shared__ float red[8][32][2];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  float pa = 0.f, pb = 0.f;
  if (c < C)
    for (int i = sl; i < B * nchunks; i += 8) { const float* src = partial + ((int64_t)i * C + c) * 2; pa += src[0]; pb += src[1]; }
  red[sl][threadIdx.x & 31][0] = pa; red[sl][threadIdx.x & 31][1] = pb;
  __syncthreads();
  if (sl != 0 || c >= C) return;
  float a = 0.f, bsum = 0.f;
#pragma unroll
  for (int q = 0; q < 8; q++) { a += red[q][threadIdx.x & 31][0]; bsum += red[q][threadIdx.x & 31][1]; }
  if (accumulate) { dgamma[c] += bsum; dbeta[c] += a; } else { dgamma[c] = bsum; dbeta[c] = a; }
}

// dx = c1 * g + c2 * xhat + c3 (+ dadd)   on interior positions, zero on the border
__global__ void __launch_bounds__(GN_THREADS) k_gn_apply_bwd(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ stats,
                                                           const bf16* __restrict__ gamma, const bf16* __restrict__ beta, const float* __restrict__ coef,
                                                           const bf16* __restrict__ dadd, bf16* __restrict__ dx, int B, int H, int W, int C, int silu,
                                                           int dy_tokens) {
  const int c8 = C / 8;
  const int rows_img = (H + 2) * (W + 2);
  const int64_t n = (int64_t)B * rows_img * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c8);
    const int64_t pos = i / c8;
    const int64_t tok = token_of(pos, H, W);
    const int b = (int)(pos / rows_img);
    bf16x8 o;
    if (tok < 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = f2bf(0.f);
    } else {
      const bf16x8 xv = *(const bf16x8*)(x + pos * C + ch * 8);
      const bf16x8 dv = *(const bf16x8*)(dy + (dy_tokens ? tok : pos) * C + ch * 8);
      const bf16x8 gv = *(const bf16x8*)(gamma + ch * 8), bv = *(const bf16x8*)(beta + ch * 8);
      const float* st = stats + ((int64_t)b * C + ch * 8) * 2;
      const float* cf = coef + ((int64_t)b * C + ch * 8) * 3;
      bf16x8 av;
      if (dadd) av = *(const bf16x8*)(dadd + pos * C + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float xh = (bf2f(xv[j]) - st[2 * j]) * st[2 * j + 1];
        float g = bf2f(dv[j]);
        if (silu) g *= silu_grad_f(xh * bf2f(gv[j]) + bf2f(bv[j]));
        float d = cf[3 * j] * g + cf[3 * j + 1] * xh + cf[3 * j + 2];
        if (dadd) d += bf2f(av[j]);
        o[j] = f2bf(d);
      }
    }
    *(bf16x8*)(dx + pos * C + ch * 8) = o;
  }
}


// ---- row-walking apply passes (r6) ----
// The first form (k_gn_apply_fwd / k_gn_apply_bwd above, kept for A/B: ST355_GN_APPLY=1) maps one flat index to (position, channel chunk): two integer divisions, the
// border test and 16 (forward) / 40 (backward) scalar loads of per-channel statistics and coefficients for every 16 bytes of payload — 3.3 TB/s in the SDXL-LoRA step
// (rocprofv3 r06: 23.6 ms of a 700 ms step at batch 32).  Here a thread OWNS its 8 channels (statistics, gamma / beta, coefficients in registers, loaded once) and
// walks the rows of its chunk of one image; the grid position advances incrementally (no division in the loop); loads are unconditional (a border position of a grid
// buffer holds zero; a token-layout operand is read at a clamped row) and the border is a select.  Same arithmetic in the same order: bit-identical outputs.
__global__ void __launch_bounds__(GN_THREADS) k_gn_apply_fwd_rows(const bf16* __restrict__ x, const float* __restrict__ stats, const bf16* __restrict__ gamma,
                                                                const bf16* __restrict__ beta, bf16* __restrict__ y, int H, int W, int C, int silu, int out_tokens,
                                                                int rows_per_chunk, int cw, int RT) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int Wp = W + 2, rows_img = (H + 2) * Wp;
  const int r0 = chunk * rows_per_chunk, r1 = min(rows_img, r0 + rows_per_chunk);
  const int c8 = C / 8;
  const int ch = blockIdx.z * cw + (int)threadIdx.x % cw, rl = (int)threadIdx.x / cw;
  if (ch >= c8 || rl >= RT) return;
  float mu[8], rs[8], ga[8], be[8];
  {
    const bf16x8 gv = *(const bf16x8*)(gamma + ch * 8), bv = *(const bf16x8*)(beta + ch * 8);
    const float* st = stats + ((int64_t)b * C + ch * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; j++) { mu[j] = st[2 * j]; rs[j] = st[2 * j + 1]; ga[j] = bf2f(gv[j]); be[j] = bf2f(bv[j]); }
  }
  int r = r0 + rl;
  int yy = r / Wp, xx = r - yy * Wp;
  const bf16* xp = x + ((int64_t)b * rows_img) * C + ch * 8;
  for (; r < r1; r += RT) {
    const bool inside = yy >= 1 && yy <= H && xx >= 1 && xx <= W;
    const bf16x8 xv = *(const bf16x8*)(xp + (int64_t)r * C);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float z = (bf2f(xv[j]) - mu[j]) * rs[j] * ga[j] + be[j];
      if (silu) z = silu_f(z);
      o[j] = f2bf(inside ? z : 0.f);
    }
    if (out_tokens) {
      if (inside) *(bf16x8*)(y + (((int64_t)b * H + yy - 1) * W + xx - 1) * C + ch * 8) = o;
    } else {
      *(bf16x8*)(y + ((int64_t)b * rows_img + r) * C + ch * 8) = o;
    }
    xx += RT;
    while (xx >= Wp) { xx -= Wp; yy++; }
  }
}

__global__ void __launch_bounds__(GN_THREADS) k_gn_apply_bwd_rows(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ stats,
                                                                const bf16* __restrict__ gamma, const bf16* __restrict__ beta, const float* __restrict__ coef,
                                                                const bf16* __restrict__ dadd, bf16* __restrict__ dx, int H, int W, int C, int silu, int dy_tokens,
                                                                int rows_per_chunk, int cw, int RT) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int Wp = W + 2, rows_img = (H + 2) * Wp;
  const int r0 = chunk * rows_per_chunk, r1 = min(rows_img, r0 + rows_per_chunk);
  const int c8 = C / 8;
  const int ch = blockIdx.z * cw + (int)threadIdx.x % cw, rl = (int)threadIdx.x / cw;
  if (ch >= c8 || rl >= RT) return;
  float mu[8], rs[8], ga[8], be[8], c1[8], c2[8], c3[8];
  {
    const bf16x8 gv = *(const bf16x8*)(gamma + ch * 8), bv = *(const bf16x8*)(beta + ch * 8);
    const float* st = stats + ((int64_t)b * C + ch * 8) * 2;
    const float* cf = coef + ((int64_t)b * C + ch * 8) * 3;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      mu[j] = st[2 * j]; rs[j] = st[2 * j + 1]; ga[j] = bf2f(gv[j]); be[j] = bf2f(bv[j]);
      c1[j] = cf[3 * j]; c2[j] = cf[3 * j + 1]; c3[j] = cf[3 * j + 2];
    }
  }
  int r = r0 + rl;
  int yy = r / Wp, xx = r - yy * Wp;
  const int64_t img0 = (int64_t)b * rows_img;
  for (; r < r1; r += RT) {
    const bool inside = yy >= 1 && yy <= H && xx >= 1 && xx <= W;
    const int64_t pos = img0 + r;
    const int64_t drow = dy_tokens ? (inside ? ((int64_t)b * H + yy - 1) * W + xx - 1 : 0) : pos;
    const bf16x8 xv = *(const bf16x8*)(x + pos * C + ch * 8);
    const bf16x8 dv = *(const bf16x8*)(dy + drow * C + ch * 8);
    bf16x8 av;
    if (dadd) av = *(const bf16x8*)(dadd + pos * C + ch * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float xh = (bf2f(xv[j]) - mu[j]) * rs[j];
      float g = bf2f(dv[j]);
      if (silu) g *= silu_grad_f(xh * ga[j] + be[j]);
      float d = c1[j] * g + c2[j] * xh + c3[j];
      if (dadd) d += bf2f(av[j]);
      o[j] = f2bf(inside ? d : 0.f);
    }
    *(bf16x8*)(dx + pos * C + ch * 8) = o;
    xx += RT;
    while (xx >= Wp) { xx -= Wp; yy++; }
  }
}
static int gn_apply_form() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("ST355_GN_APPLY"); v = (e && e[0] == '1') ? 1 : 2; }
  return v;
}
// launch geometry of the row-walking passes: channel windows of <= 256 chunks of equal width, RT rows per pass
static void gn_rows_geom(int C, int* nwin, int* cw, int* RT) {
  const int c8 = C / 8;
  *nwin = (c8 + GN_THREADS - 1) / GN_THREADS;
  *cw = (c8 + *nwin - 1) / *nwin;
  *RT = GN_THREADS / *cw;
  if (*RT < 1) *RT = 1;
}

static int gn_chunks(int B, int H, int W, int* rows_per_chunk) {
  const int rows_img = (H + 2) * (W + 2);
  int nch = (768 + B - 1) / B;
  if (nch > (rows_img + 31) / 32) nch = (rows_img + 31) / 32;
  if (nch < 1) nch = 1;
  *rows_per_chunk = (rows_img + nch - 1) / nch;
  return (rows_img + *rows_per_chunk - 1) / *rows_per_chunk;
}
extern "C" size_t st355_groupnorm_workspace(int B, int H, int W, int C) {
  int rpc;
  const int nch = gn_chunks(B, H, W, &rpc);
  return (size_t)B * nch * C * 2 * 4 + (size_t)B * C * 3 * 4 + 256;
}
static size_t gn_lds(int C) { const int c8 = C / 8; const int cw = c8 < GN_THREADS ? c8 : GN_THREADS; return (size_t)(GN_THREADS / cw) * cw * 8 * 2 * 4; }

extern "C" int st355_groupnorm_fwd(void* stream, const void* x, const void* gamma, const void* beta, void* y, float* stats /* [B,C,2] fp32 out */, int B, int H,
                                   int W, int C, int groups, float eps, int silu, int out_tokens, void* workspace) {
  ST_REQUIRE(x && gamma && beta && y && stats && workspace, "groupnorm_fwd: null pointer");
  ST_REQUIRE(C % 8 == 0 && C % groups == 0 && C <= GN_MAXC && B > 0, "groupnorm_fwd: bad shape C=%d groups=%d", C, groups);
  int rpc;
  const int nch = gn_chunks(B, H, W, &rpc);
  float* partial = (float*)workspace;
  ProfScope ps(stream, ST355_K_LN_MOD, 10.0 * B * H * W * C, 6.0 * B * (H + 2) * (W + 2) * C);
  hipLaunchKernelGGL(k_gn_stats<0>, dim3(nch, B), dim3(GN_THREADS), gn_lds(C), (hipStream_t)stream, (const bf16*)x, (const bf16*)nullptr, (const float*)nullptr,
                     (const bf16*)nullptr, (const bf16*)nullptr, partial, H, W, C, rpc, nch, 0, 0);
  hipLaunchKernelGGL(k_gn_finalize_fwd, dim3(groups, B), dim3(64), 0, (hipStream_t)stream, (const float*)partial, stats, C, groups, nch,
                     (float)((double)H * W * (C / groups)), eps);
  if (gn_apply_form() == 2) {
    int nwin, cw, RT;
    gn_rows_geom(C, &nwin, &cw, &RT);
    hipLaunchKernelGGL(k_gn_apply_fwd_rows, dim3(nch, B, nwin), dim3(cw * RT), 0, (hipStream_t)stream, (const bf16*)x, (const float*)stats, (const bf16*)gamma,
                       (const bf16*)beta, (bf16*)y, H, W, C, silu, out_tokens, rpc, cw, RT);
    return st355_check_launch("groupnorm_fwd");
  }
  const int64_t n = (int64_t)B * (H + 2) * (W + 2) * (C / 8);
  hipLaunchKernelGGL(k_gn_apply_fwd, dim3((unsigned)std::min<int64_t>(cdiv64(n, GN_THREADS), 65536)), dim3(GN_THREADS), 0, (hipStream_t)stream, (const bf16*)x,
                     (const float*)stats, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, B, H, W, C, silu, out_tokens);
  return st355_check_launch("groupnorm_fwd");
}

extern "C" int st355_groupnorm_bwd(void* stream, const void* dy, const void* x, const void* gamma, const void* beta, const float* stats, const void* dadd,
                                   void* dx, float* dgamma, float* dbeta, int B, int H, int W, int C, int groups, int silu, int dy_tokens, int accumulate_params,
                                   void* workspace) {
  ST_REQUIRE(dy && x && gamma && beta && stats && dx && workspace, "groupnorm_bwd: null pointer");
  ST_REQUIRE(C % 8 == 0 && C % groups == 0 && C <= GN_MAXC && B > 0, "groupnorm_bwd: bad shape C=%d groups=%d", C, groups);
  int rpc;
  const int nch = gn_chunks(B, H, W, &rpc);
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)B * nch * C * 2;
  ProfScope ps(stream, ST355_K_LN_MOD, 30.0 * B * H * W * C, 10.0 * B * (H + 2) * (W + 2) * C);
  hipLaunchKernelGGL(k_gn_stats<1>, dim3(nch, B), dim3(GN_THREADS), gn_lds(C), (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, stats, (const bf16*)gamma,
                     (const bf16*)beta, partial, H, W, C, rpc, nch, silu, dy_tokens);
  hipLaunchKernelGGL(k_gn_finalize_bwd, dim3(groups, B), dim3(64), 0, (hipStream_t)stream, (const float*)partial, stats, (const bf16*)gamma, coef, C, groups, nch,
                     (float)((double)H * W * (C / groups)));
  if (dgamma && dbeta)
    hipLaunchKernelGGL(k_gn_param_grads, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float*)partial, dgamma, dbeta, B, C, nch, accumulate_params);
  if (gn_apply_form() == 2) {
    int nwin, cw, RT;
    gn_rows_geom(C, &nwin, &cw, &RT);
    hipLaunchKernelGGL(k_gn_apply_bwd_rows, dim3(nch, B, nwin), dim3(cw * RT), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, stats, (const bf16*)gamma,
                       (const bf16*)beta, (const float*)coef, (const bf16*)dadd, (bf16*)dx, H, W, C, silu, dy_tokens, rpc, cw, RT);
    return st355_check_launch("groupnorm_bwd");
  }
  const int64_t n = (int64_t)B * (H + 2) * (W + 2) * (C / 8);
  hipLaunchKernelGGL(k_gn_apply_bwd, dim3((unsigned)std::min<int64_t>(cdiv64(n, GN_THREADS), 65536)), dim3(GN_THREADS), 0, (hipStream_t)stream, (const bf16*)x,
                     (const bf16*)dy, stats, (const bf16*)gamma, (const bf16*)beta, (const float*)coef, (const bf16*)dadd, (bf16*)dx, B, H, W, C, silu, dy_tokens);
  return st355_check_launch("groupnorm_bwd");
}
