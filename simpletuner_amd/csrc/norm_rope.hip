// norm_rope.hip — the memory-bound glue of an MMDiT block:
//   K5  AdaLN modulate          y = LN(x) * (1 + scale_b) + shift_b           (fwd + bwd)
//   K6  per-head RMSNorm(q,k) + RoPE + head-major re-layout of q,k,v          (fwd + bwd)
// Each kernel is one pass over HBM with 16-byte accesses; rows live in registers between the
// statistics pass and the normalise pass (no re-read).
#include "common.h"

// ================================================================================================
// K5: LayerNorm (no affine) + modulation. One wave per row, NC 16-byte chunks per lane.
// ================================================================================================
template <int NC>
__global__ void __launch_bounds__(256) k_ln_mod_fwd(const bf16* __restrict__ x, int64_t ldx, const bf16* __restrict__ scale,
                                                   const bf16* __restrict__ shift, int64_t mod_stride, int64_t rows_per_batch,
                                                   bf16* __restrict__ y, int64_t ldy, int64_t rows, int D, float eps, float one) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t b = row / rows_per_batch;
  const bf16* xr = x + row * ldx;
  float v[NC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
      bf16x8 t = *(const bf16x8*)(xr + idx);
#pragma unroll
      for (int j = 0; j < 8; j++) { v[c][j] = bf2f(t[j]); s += v[c][j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) v[c][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
#pragma unroll
      for (int j = 0; j < 8; j++) { float d = v[c][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  const bf16* sc = scale + b * mod_stride;
  const bf16* sh = shift + b * mod_stride;
  bf16* yr = y + row * ldy;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
      bf16x8 scv = *(const bf16x8*)(sc + idx);
      bf16x8 shv = *(const bf16x8*)(sh + idx);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = f2bf((v[c][j] - mean) * rstd * (one + bf2f(scv[j])) + bf2f(shv[j]));
      *(bf16x8*)(yr + idx) = o;
    }
  }
}

static int ln_fwd_impl(void* stream, const void* x, int64_t ldx, const void* scale, const void* shift,
                                     int64_t mod_stride, int64_t rows_per_batch, void* y, int64_t ldy, int64_t rows, int D,
                                     float eps, float one) {
  ST_REQUIRE(x && scale && shift && y, "ln_modulate_fwd: null pointer");
  ST_REQUIRE(D % 8 == 0 && D <= 4096 && ldx % 8 == 0 && ldy % 8 == 0 && mod_stride % 8 == 0 && rows > 0 && rows_per_batch > 0,
             "ln_modulate_fwd: bad shape D=%d", D);
  ProfScope ps(stream, ST355_K_LN_MOD, 8.0 * rows * D, 4.0 * rows * D);
  dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
#define LAUNCH(NC)                                                                                                       \
  hipLaunchKernelGGL(k_ln_mod_fwd<NC>, grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx, (const bf16*)scale,       \
                     (const bf16*)shift, mod_stride, rows_per_batch, (bf16*)y, ldy, rows, D, eps, one)
  if (D <= 512) LAUNCH(1);
  else if (D <= 1024) LAUNCH(2);
  else if (D <= 1536) LAUNCH(3);
  else if (D <= 2048) LAUNCH(4);
  else if (D <= 3072) LAUNCH(6);
  else LAUNCH(8);
#undef LAUNCH
  return st355_check_launch("ln_modulate_fwd");
}
extern "C" int st355_ln_modulate_fwd(void* stream, const void* x, int64_t ldx, const void* scale, const void* shift,
                                     int64_t mod_stride, int64_t rows_per_batch, void* y, int64_t ldy, int64_t rows, int D,
                                     float eps) {
  return ln_fwd_impl(stream, x, ldx, scale, shift, mod_stride, rows_per_batch, y, ldy, rows, D, eps, 1.f);
}
// plain affine LayerNorm  y = LN(x) * weight + bias  (the UNet's BasicTransformerBlock norm1/2/3, eps 1e-5): same kernel, weight in place of (1+scale)
extern "C" int st355_layernorm_fwd(void* stream, const void* x, int64_t ldx, const void* weight, const void* bias, void* y, int64_t ldy, int64_t rows, int D,
                                   float eps) {
  return ln_fwd_impl(stream, x, ldx, weight, bias, 0, rows, y, ldy, rows, D, eps, 0.f);
}

// backward: g = dy*(1+scale); dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) [+ dres];  dxg = gate*dx
template <int NC>
__global__ void __launch_bounds__(256) k_ln_mod_bwd(const bf16* __restrict__ dy, int64_t lddy, const bf16* __restrict__ x,
                                                   int64_t ldx, const bf16* __restrict__ scale, int64_t mod_stride,
                                                   int64_t rows_per_batch, const bf16* __restrict__ dres, int64_t lddres,
                                                   const bf16* __restrict__ gate, int64_t gate_stride, bf16* __restrict__ dx,
                                                   int64_t lddx, bf16* __restrict__ dxg, int64_t lddxg, int64_t rows, int D,
                                                   float eps, float one) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t b = row / rows_per_batch;
  const bf16* xr = x + row * ldx;
  const bf16* dyr = dy + row * lddy;
  const bf16* sc = scale + b * mod_stride;
  float v[NC][8], g[NC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
      bf16x8 t = *(const bf16x8*)(xr + idx);
      bf16x8 d = *(const bf16x8*)(dyr + idx);
      bf16x8 scv = *(const bf16x8*)(sc + idx);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        v[c][j] = bf2f(t[j]);
        s += v[c][j];
        g[c][j] = bf2f(d[j]) * (one + bf2f(scv[j]));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) { v[c][j] = 0.f; g[c][j] = 0.f; }
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
#pragma unroll
      for (int j = 0; j < 8; j++) { float d = v[c][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float xh = (v[c][j] - mean) * rstd;
        v[c][j] = xh;
        sg += g[c][j];
        sgx += g[c][j] * xh;
      }
    }
  }
  const float c1 = wave_sum(sg) / (float)D;
  const float c2 = wave_sum(sgx) / (float)D;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = rstd * (g[c][j] - c1 - v[c][j] * c2);
      if (dres) {
        bf16x8 r = *(const bf16x8*)(dres + row * lddres + idx);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] += bf2f(r[j]);
      }
      bf16x8 ov;
#pragma unroll
      for (int j = 0; j < 8; j++) ov[j] = f2bf(o[j]);
      *(bf16x8*)(dx + row * lddx + idx) = ov;
      if (dxg) {
        bf16x8 gv = *(const bf16x8*)(gate + b * gate_stride + idx);
        bf16x8 og;
#pragma unroll
        for (int j = 0; j < 8; j++) og[j] = f2bf(bf2f(ov[j]) * bf2f(gv[j]));
        *(bf16x8*)(dxg + row * lddxg + idx) = og;
      }
    }
  }
}

static int ln_bwd_impl(void* stream, const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* scale,
                                     int64_t mod_stride, int64_t rows_per_batch, const void* dres, int64_t lddres,
                                     const void* gate, int64_t gate_stride, void* dx, int64_t lddx, void* dxg, int64_t lddxg,
                                     int64_t rows, int D, float eps, float one) {
  ST_REQUIRE(dy && x && scale && dx, "ln_modulate_bwd: null pointer");
  ST_REQUIRE(D % 8 == 0 && D <= 4096 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && mod_stride % 8 == 0 && rows > 0 &&
                 rows_per_batch > 0, "ln_modulate_bwd: bad shape D=%d", D);
  if (dres) ST_REQUIRE(lddres % 8 == 0, "ln_modulate_bwd: lddres");
  if (dxg) ST_REQUIRE(gate && gate_stride % 8 == 0 && lddxg % 8 == 0, "ln_modulate_bwd: gate missing for dxg");
  ProfScope ps(stream, ST355_K_LN_MOD, 16.0 * rows * D, (6.0 + (dres ? 2.0 : 0.0) + (dxg ? 2.0 : 0.0)) * rows * D);
  dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
#define LAUNCH(NC)                                                                                                        \
  hipLaunchKernelGGL(k_ln_mod_bwd<NC>, grid, block, 0, (hipStream_t)stream, (const bf16*)dy, lddy, (const bf16*)x, ldx,     \
                     (const bf16*)scale, mod_stride, rows_per_batch, (const bf16*)dres, lddres, (const bf16*)gate,          \
                     gate_stride, (bf16*)dx, lddx, (bf16*)dxg, lddxg, rows, D, eps, one)
  if (D <= 512) LAUNCH(1);
  else if (D <= 1024) LAUNCH(2);
  else if (D <= 1536) LAUNCH(3);
  else if (D <= 2048) LAUNCH(4);
  else if (D <= 3072) LAUNCH(6);
  else LAUNCH(8);
#undef LAUNCH
  return st355_check_launch("ln_modulate_bwd");
}
extern "C" int st355_ln_modulate_bwd(void* stream, const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* scale,
                                     int64_t mod_stride, int64_t rows_per_batch, const void* dres, int64_t lddres,
                                     const void* gate, int64_t gate_stride, void* dx, int64_t lddx, void* dxg, int64_t lddxg,
                                     int64_t rows, int D, float eps) {
  return ln_bwd_impl(stream, dy, lddy, x, ldx, scale, mod_stride, rows_per_batch, dres, lddres, gate, gate_stride, dx, lddx, dxg, lddxg, rows, D, eps, 1.f);
}
// dx = dres + LNbwd(dy * weight)   (weight / bias gradients: st355_colsum_prod on dy and dy * xhat)
extern "C" int st355_layernorm_bwd(void* stream, const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* weight, const void* dres, int64_t lddres,
                                   void* dx, int64_t lddx, int64_t rows, int D, float eps) {
  return ln_bwd_impl(stream, dy, lddy, x, ldx, weight, 0, rows, dres, lddres, nullptr, 0, dx, lddx, nullptr, 0, rows, D, eps, 0.f);
}

// affine LayerNorm parameter gradients: dweight[c] = sum_rows dy*xhat, dbias[c] = sum_rows dy.  A wave walks rows (stride = waves in the grid) with the
// row in registers (statistics recomputed), a lane owns the same channels for every row, partials [wave][D][2] -> fixed-order reduce.
template <int NC>
__global__ void __launch_bounds__(256) k_ln_param_partials(const bf16* __restrict__ dy, int64_t lddy, const bf16* __restrict__ x, int64_t ldx, int64_t rows, int D,
                                                          float eps, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  float aw[NC][8], ab[NC][8];
#pragma unroll
  for (int c = 0; c < NC; c++)
#pragma unroll
    for (int j = 0; j < 8; j++) { aw[c][j] = 0.f; ab[c][j] = 0.f; }
  for (int64_t row = wave; row < rows; row += nw) {
    float v[NC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
        const bf16x8 t = *(const bf16x8*)(x + row * ldx + idx);
#pragma unroll
        for (int j = 0; j < 8; j++) { v[c][j] = bf2f(t[j]); s += v[c][j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[c][j] = 0.f;
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) { const float d = v[c][j] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
        const bf16x8 d = *(const bf16x8*)(dy + row * lddy + idx);
#pragma unroll
        for (int j = 0; j < 8; j++) { const float g = bf2f(d[j]); aw[c][j] += g * (v[c][j] - mean) * rstd; ab[c][j] += g; }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
#pragma unroll
      for (int j = 0; j < 8; j++) { partial[((int64_t)wave * D + idx + j) * 2] = aw[c][j]; partial[((int64_t)wave * D + idx + j) * 2 + 1] = ab[c][j]; }
    }
  }
}
__global__ void __launch_bounds__(256) k_ln_param_reduce(const float* __restrict__ partial, int nw, int D, float* __restrict__ dweight, float* __restrict__ dbias,
                                                        int accumulate) {
  __shared__ float red[8][32][2];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  float pa = 0.f, pb = 0.f;
  if (c < D)
    for (int w = sl; w < nw; w += 8) { pa += partial[((int64_t)w * D + c) * 2]; pb += partial[((int64_t)w * D + c) * 2 + 1]; }
  red[sl][threadIdx.x & 31][0] = pa; red[sl][threadIdx.x & 31][1] = pb;
  __syncthreads();
  if (sl != 0 || c >= D) return;
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int q = 0; q < 8; q++) { a += red[q][threadIdx.x & 31][0]; b += red[q][threadIdx.x & 31][1]; }
  if (accumulate) { dweight[c] += a; dbias[c] += b; } else { dweight[c] = a; dbias[c] = b; }
}
#define LNP_BLOCKS 128
extern "C" size_t st355_layernorm_param_grads_workspace(int D) { return (size_t)LNP_BLOCKS * 4 * D * 2 * 4; }
extern "C" int st355_layernorm_param_grads(void* stream, const void* dy, int64_t lddy, const void* x, int64_t ldx, int64_t rows, int D, float eps, float* dweight,
                                           float* dbias, int accumulate, void* workspace) {
  ST_REQUIRE(dy && x && dweight && dbias && workspace && rows > 0 && D % 8 == 0 && D <= 2048 && ldx % 8 == 0 && lddy % 8 == 0, "layernorm_param_grads: bad args");
  ProfScope ps(stream, ST355_K_LN_MOD, 8.0 * rows * D, 4.0 * rows * D);
#define LAUNCH(NC) hipLaunchKernelGGL(k_ln_param_partials<NC>, dim3(LNP_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, lddy, (const bf16*)x, ldx, rows, D, eps, (float*)workspace)
  if (D <= 512) LAUNCH(1);
  else if (D <= 1024) LAUNCH(2);
  else if (D <= 1536) LAUNCH(3);
  else LAUNCH(4);
#undef LAUNCH
  hipLaunchKernelGGL(k_ln_param_reduce, dim3((D + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, LNP_BLOCKS * 4, D, dweight, dbias, accumulate);
  return st355_check_launch("layernorm_param_grads");
}

// ================================================================================================
// K6: RMSNorm(q,k) + RoPE + re-layout.  grid = (ceil(S_part/64), H, B), 256 threads.
// A thread owns 8 contiguous head channels (4 rotation pairs) of one token; HD/8 threads share a
// token-head row and reduce sum(x^2) with xor-shuffles.  Transposed copies go through an LDS tile
// with an odd dword pitch (2-way conflicts at most).
// ================================================================================================
#define TP 130  // LDS pitch (elements) of the [64 tok][HD] staging tile: 65 dwords -> odd

template <int HD>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = HD / 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int HD>
__global__ void __launch_bounds__(256) k_qk_norm_rope_fwd(const bf16* __restrict__ qkv, int64_t ld, const bf16* __restrict__ wq,
                                                         const bf16* __restrict__ wk, const float* __restrict__ cosT,
                                                         const float* __restrict__ sinT, bf16* __restrict__ Q, bf16* __restrict__ K,
                                                         bf16* __restrict__ Qt, bf16* __restrict__ Kt, bf16* __restrict__ Vt, int H,
                                                         int S_part, int pos0, int S, int Sp, float eps) {
  constexpr int TPR = HD / 8;            // threads per token row
  constexpr int TOK_PER_PASS = 256 / TPR;
  __shared__ __attribute__((aligned(16))) bf16 tile[3][64 * TP];
  const int tid = threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * 64;
  const int c = tid % TPR;               // 16-byte chunk inside the head row
  const int64_t Dm = (int64_t)H * HD;    // model width
  const int64_t bh = (int64_t)b * H + h;

  float wqv[8], wkv[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    wqv[j] = wq ? bf2f(wq[c * 8 + j]) : 1.f;
    wkv[j] = wk ? bf2f(wk[c * 8 + j]) : 1.f;
  }

  for (int tl = tid / TPR; tl < 64; tl += TOK_PER_PASS) {
    const int t = t0 + tl;
    const bool valid = t < S_part;
    const int tt = valid ? t : S_part - 1;
    const bf16* row = qkv + ((int64_t)b * S + pos0 + tt) * ld + (int64_t)h * HD + c * 8;
    bf16x8 qv = *(const bf16x8*)(row);
    bf16x8 kv = *(const bf16x8*)(row + Dm);
    bf16x8 vv = *(const bf16x8*)(row + 2 * Dm);
    const int pos = pos0 + tt;
    const float* cp = cosT + (int64_t)pos * HD + c * 8;
    const float* sp = sinT + (int64_t)pos * HD + c * 8;
    float cs[8], sn[8];
    *(f32x4*)&cs[0] = *(const f32x4*)cp; *(f32x4*)&cs[4] = *(const f32x4*)(cp + 4);
    *(f32x4*)&sn[0] = *(const f32x4*)sp; *(f32x4*)&sn[4] = *(const f32x4*)(sp + 4);
    float qf[8], kf[8], sq = 0.f, sk = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) { qf[j] = bf2f(qv[j]); kf[j] = bf2f(kv[j]); sq += qf[j] * qf[j]; sk += kf[j] * kf[j]; }
    const float rq = wq ? rsqrtf(group_sum<HD>(sq) / (float)HD + eps) : 1.f;
    const float rk = wk ? rsqrtf(group_sum<HD>(sk) / (float)HD + eps) : 1.f;
#pragma unroll
    for (int j = 0; j < 8; j++) { qf[j] *= rq * wqv[j]; kf[j] *= rk * wkv[j]; }
    bf16x8 qo, ko;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      // out = x*cos + rot(x)*sin, rot = (-x_imag, x_real) on interleaved pairs (flux/transformer.py:91-98)
      qo[j] = f2bf(qf[j] * cs[j] - qf[j + 1] * sn[j]);
      qo[j + 1] = f2bf(qf[j + 1] * cs[j + 1] + qf[j] * sn[j + 1]);
      ko[j] = f2bf(kf[j] * cs[j] - kf[j + 1] * sn[j]);
      ko[j + 1] = f2bf(kf[j + 1] * cs[j + 1] + kf[j] * sn[j + 1]);
    }
    if (valid) {
      const int64_t o = (bh * S + pos) * HD + c * 8;
      *(bf16x8*)(Q + o) = qo;
      *(bf16x8*)(K + o) = ko;
    }
    // stage for the transposed copies (4-byte LDS writes: rows are only 4-byte aligned at pitch 130)
    uint32_t* tq = (uint32_t*)(&tile[0][tl * TP + c * 8]);
    uint32_t* tk = (uint32_t*)(&tile[1][tl * TP + c * 8]);
    uint32_t* tv = (uint32_t*)(&tile[2][tl * TP + c * 8]);
    const u32x4 qw = *(const u32x4*)&qo, kw = *(const u32x4*)&ko, vw = *(const u32x4*)&vv;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (Qt) tq[j] = qw[j];
      if (Kt) tk[j] = kw[j];
      tv[j] = vw[j];
    }
  }
  __syncthreads();
  // transposed rows: each thread emits 8 consecutive tokens of one channel
  const int pbase = pos0 + t0;
  const int nvalid = min(64, S_part - t0);
  for (int i = tid; i < HD * 8; i += 256) {
    const int d = i >> 3, tc = i & 7;
#pragma unroll
    for (int w = 0; w < 3; w++) {
      bf16* tbase = (w == 0 ? Qt : (w == 1 ? Kt : Vt));
      if (tbase == nullptr) continue;                 // Qt / Kt are optional: the head_dim-128 backward gathers Q^T / K^T fragments by transposing LDS reads
      bf16* dst = tbase + (bh * HD + d) * (int64_t)Sp + pbase + tc * 8;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = tile[w][(tc * 8 + e) * TP + d];
      if (tc * 8 + 8 <= nvalid && (((uintptr_t)dst) & 15) == 0) {
        *(bf16x8*)dst = o;
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++)
          if (tc * 8 + e < nvalid) dst[e] = o[e];
      }
    }
  }
}

extern "C" int st355_qk_norm_rope_fwd(void* stream, const void* qkv, int64_t ld_qkv, const void* wq, const void* wk,
                                      const float* cos, const float* sin, void* Q, void* K, void* Qt, void* Kt, void* Vt, int B,
                                      int H, int d, int S_part, int pos0, int S, int Sp, float eps) {
  ST_REQUIRE(qkv && cos && sin && Q && K && Vt, "qk_norm_rope_fwd: null pointer (Qt / Kt may be NULL: backward without transposed copies)");
  ST_REQUIRE(ld_qkv % 8 == 0 && Sp % 64 == 0 && Sp >= S && pos0 + S_part <= S && S_part > 0, "qk_norm_rope_fwd: bad shape");
  ST_REQUIRE(d == 128 || d == 64, "qk_norm_rope_fwd: head_dim %d not built", d);
  const double n = (double)B * S_part * H * d;
  ProfScope ps(stream, ST355_K_QK_ROPE, 20.0 * n, (6.0 + 6.0 + (Qt ? 2.0 : 0.0) + (Kt ? 2.0 : 0.0)) * n);
  dim3 grid((S_part + 63) / 64, H, B), block(256);
  if (d == 128)
    hipLaunchKernelGGL(k_qk_norm_rope_fwd<128>, grid, block, 0, (hipStream_t)stream, (const bf16*)qkv, ld_qkv, (const bf16*)wq,
                       (const bf16*)wk, cos, sin, (bf16*)Q, (bf16*)K, (bf16*)Qt, (bf16*)Kt, (bf16*)Vt, H, S_part, pos0, S, Sp, eps);
  else
    hipLaunchKernelGGL(k_qk_norm_rope_fwd<64>, grid, block, 0, (hipStream_t)stream, (const bf16*)qkv, ld_qkv, (const bf16*)wq,
                       (const bf16*)wk, cos, sin, (bf16*)Q, (bf16*)K, (bf16*)Qt, (bf16*)Kt, (bf16*)Vt, H, S_part, pos0, S, Sp, eps);
  return st355_check_launch("qk_norm_rope_fwd");
}

// backward of RoPE (transpose of the rotation) then RMSNorm:  y = x * r * w, r = rsqrt(mean(x^2)+eps)
//   dx = r*w*dy - x * r^3 * mean(x*w*dy)
template <int HD>
__global__ void __launch_bounds__(256) k_qk_norm_rope_bwd(const bf16* __restrict__ dQ, const bf16* __restrict__ dK,
                                                         const bf16* __restrict__ qkv, int64_t ld, const bf16* __restrict__ wq,
                                                         const bf16* __restrict__ wk, const float* __restrict__ cosT,
                                                         const float* __restrict__ sinT, bf16* __restrict__ dqkv, int64_t ldd, int H,
                                                         int S_part, int pos0, int S, float eps, float* __restrict__ wg_part) {
  constexpr int TPR = HD / 8;
  constexpr int TOK_PER_PASS = 256 / TPR;
  const int tid = threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * 64;
  const int c = tid % TPR;
  // wg_part != NULL (full fine-tune of a q/k-RMSNorm model, SD3.5): this workgroup's share of d loss / d norm weight, dw[ch] = sum over its tokens of
  // dy[ch] * x_hat[ch] (y = x_hat * w), reduced in a fixed order; k_qk_norm_wgrad_final adds the workgroups up in index order
  float wacc[2][8];
#pragma unroll
  for (int w = 0; w < 2; w++)
#pragma unroll
    for (int j = 0; j < 8; j++) wacc[w][j] = 0.f;
  const int64_t Dm = (int64_t)H * HD;
  const int64_t bh = (int64_t)b * H + h;
  float wv[2][8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    wv[0][j] = wq ? bf2f(wq[c * 8 + j]) : 1.f;
    wv[1][j] = wk ? bf2f(wk[c * 8 + j]) : 1.f;
  }
  for (int tl = tid / TPR; tl < 64; tl += TOK_PER_PASS) {
    const int t = t0 + tl;
    const bool valid = t < S_part;
    const int tt = valid ? t : S_part - 1;
    const int pos = pos0 + tt;
    const float* cp = cosT + (int64_t)pos * HD + c * 8;
    const float* sp = sinT + (int64_t)pos * HD + c * 8;
    float cs[8], sn[8];
    *(f32x4*)&cs[0] = *(const f32x4*)cp; *(f32x4*)&cs[4] = *(const f32x4*)(cp + 4);
    *(f32x4*)&sn[0] = *(const f32x4*)sp; *(f32x4*)&sn[4] = *(const f32x4*)(sp + 4);
    const bf16* xrow = qkv + ((int64_t)b * S + pos0 + tt) * ld + (int64_t)h * HD + c * 8;
    bf16* drow = dqkv + ((int64_t)b * S + pos0 + tt) * ldd + (int64_t)h * HD + c * 8;
    const int64_t go = (bh * S + pos) * HD + c * 8;
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const bool has_norm = (w == 0) ? (wq != nullptr) : (wk != nullptr);
      bf16x8 gv = *(const bf16x8*)((w == 0 ? dQ : dK) + go);
      bf16x8 xv = *(const bf16x8*)(xrow + w * Dm);
      float g[8], dy[8], xf[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { g[j] = bf2f(gv[j]); xf[j] = bf2f(xv[j]); }
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        // forward: o0 = y0*c0 - y1*s0 ; o1 = y1*c1 + y0*s1   =>  dy0 = g0*c0 + g1*s1 ; dy1 = g1*c1 - g0*s0
        dy[j] = g[j] * cs[j] + g[j + 1] * sn[j + 1];
        dy[j + 1] = g[j + 1] * cs[j + 1] - g[j] * sn[j];
      }
      bf16x8 o;
      if (has_norm) {
        float sx = 0.f, sxy = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) { sx += xf[j] * xf[j]; sxy += xf[j] * wv[w][j] * dy[j]; }
        const float r = rsqrtf(group_sum<HD>(sx) / (float)HD + eps);
        const float m = group_sum<HD>(sxy) / (float)HD;
        const float r3m = r * r * r * m;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = f2bf(r * wv[w][j] * dy[j] - xf[j] * r3m);
        if (wg_part != nullptr && valid) {
#pragma unroll
          for (int j = 0; j < 8; j++) wacc[w][j] += dy[j] * xf[j] * r;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = f2bf(dy[j]);
      }
      if (valid) *(bf16x8*)(drow + w * Dm) = o;
    }
  }
  if (wg_part != nullptr) {
    __shared__ float red[256 * 8];
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
    for (int w = 0; w < 2; w++) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; j++) red[tid * 8 + j] = wacc[w][j];
      __syncthreads();
      if (tid < HD) {
        const int cc = tid >> 3, jj = tid & 7;
        float sum = 0.f;
        for (int tl = 0; tl < TOK_PER_PASS; tl++) sum += red[(tl * TPR + cc) * 8 + jj];
        wg_part[((int64_t)w * nblk + blk) * HD + tid] = sum;
      }
    }
  }
}

// first level of the reduction over the workgroups' partials: slice `blockIdx.y` of weight `blockIdx.x` (q, k) sums its `per` consecutive partials (4 interleaved
// chains, fixed order) into part2[w][slice][ch].  A Flux.1 block at batch 8 leaves 13 824 partials per weight: walked by the two workgroups of the final kernel alone
// that was 0.96 ms per call (rocprofv3, r03r: 72 ms per full-rank step); 64 slices x 2 weights put it on 128 CUs.
template <int HD>
__global__ void k_qk_norm_wgrad_slices(const float* __restrict__ part, int nblk, int per, float* __restrict__ part2) {
  __shared__ float red[4][HD];
  const int ch = threadIdx.x % HD, k = threadIdx.x / HD, w = blockIdx.x, sl = blockIdx.y, ns = gridDim.y;
  const int i0 = sl * per, i1 = min(nblk, i0 + per);
  float s = 0.f;
  for (int i = i0 + k; i < i1; i += 4) s += part[((int64_t)w * nblk + i) * HD + ch];
  red[k][ch] = s;
  __syncthreads();
  if (k == 0) part2[((int64_t)w * ns + sl) * HD + ch] = ((red[0][ch] + red[1][ch]) + red[2][ch]) + red[3][ch];
}

// out[w][ch] (+)= sum over the slices (index order) of the sums above.  grid = 2 (q, k), HD threads x 4 interleaved partial chains
template <int HD>
__global__ void k_qk_norm_wgrad_final(const float* __restrict__ part, int nblk, bf16* __restrict__ gwq, bf16* __restrict__ gwk, int accumulate) {
  __shared__ float red[4][HD];
  const int ch = threadIdx.x % HD, k = threadIdx.x / HD, w = blockIdx.x;
  bf16* dst = w == 0 ? gwq : gwk;
  if (dst == nullptr) return;
  float s = 0.f;
  for (int i = k; i < nblk; i += 4) s += part[((int64_t)w * nblk + i) * HD + ch];
  red[k][ch] = s;
  __syncthreads();
  if (k == 0) {
    float t = ((red[0][ch] + red[1][ch]) + red[2][ch]) + red[3][ch];
    if (accumulate) t += bf2f(dst[ch]);
    dst[ch] = f2bf(t);
  }
}

extern "C" int st355_qk_norm_rope_bwd(void* stream, const void* dQ, const void* dK, const void* qkv, int64_t ld_qkv, const void* wq,
                                      const void* wk, const float* cos, const float* sin, void* dqkv, int64_t ld_dqkv, int B, int H,
                                      int d, int S_part, int pos0, int S, float eps) {
  ST_REQUIRE(dQ && dK && qkv && cos && sin && dqkv, "qk_norm_rope_bwd: null pointer");
  ST_REQUIRE(ld_qkv % 8 == 0 && ld_dqkv % 8 == 0 && pos0 + S_part <= S && S_part > 0, "qk_norm_rope_bwd: bad shape");
  ST_REQUIRE(d == 128 || d == 64, "qk_norm_rope_bwd: head_dim %d not built", d);
  const double n = (double)B * S_part * H * d;
  ProfScope ps(stream, ST355_K_QK_ROPE, 30.0 * n, 12.0 * n);
  dim3 grid((S_part + 63) / 64, H, B), block(256);
  if (d == 128)
    hipLaunchKernelGGL(k_qk_norm_rope_bwd<128>, grid, block, 0, (hipStream_t)stream, (const bf16*)dQ, (const bf16*)dK,
                       (const bf16*)qkv, ld_qkv, (const bf16*)wq, (const bf16*)wk, cos, sin, (bf16*)dqkv, ld_dqkv, H, S_part, pos0, S, eps, (float*)nullptr);
  else
    hipLaunchKernelGGL(k_qk_norm_rope_bwd<64>, grid, block, 0, (hipStream_t)stream, (const bf16*)dQ, (const bf16*)dK,
                       (const bf16*)qkv, ld_qkv, (const bf16*)wq, (const bf16*)wk, cos, sin, (bf16*)dqkv, ld_dqkv, H, S_part, pos0, S, eps, (float*)nullptr);
  return st355_check_launch("qk_norm_rope_bwd");
}

// the same backward, also producing d loss / d (norm_q.weight, norm_k.weight) (sd3/transformer.py:155-165 q/k RMSNorm of SD3.5; the weights train in a full
// fine-tune): workspace = st355_qk_norm_wgrad_workspace(B, H, d, S_part) bytes of fp32 partials; gwq / gwk bf16 [d] (NULL: that weight is absent / frozen)
extern "C" size_t st355_qk_norm_wgrad_workspace(int B, int H, int d, int S_part) {
  return ((size_t)2 * ((size_t)(S_part + 63) / 64) * H * B + (size_t)2 * 64) * d * sizeof(float);        // per-workgroup partials + the <= 64 slice sums per weight
}
extern "C" int st355_qk_norm_rope_bwd_wgrad(void* stream, const void* dQ, const void* dK, const void* qkv, int64_t ld_qkv, const void* wq,
                                            const void* wk, const float* cos, const float* sin, void* dqkv, int64_t ld_dqkv, int B, int H,
                                            int d, int S_part, int pos0, int S, float eps, void* gwq, void* gwk, int accumulate, void* workspace) {
  ST_REQUIRE(dQ && dK && qkv && cos && sin && dqkv && workspace, "qk_norm_rope_bwd_wgrad: null pointer");
  ST_REQUIRE((gwq == nullptr || wq != nullptr) && (gwk == nullptr || wk != nullptr), "qk_norm_rope_bwd_wgrad: a weight gradient needs its norm weight");
  ST_REQUIRE(ld_qkv % 8 == 0 && ld_dqkv % 8 == 0 && pos0 + S_part <= S && S_part > 0, "qk_norm_rope_bwd_wgrad: bad shape");
  ST_REQUIRE(d == 128 || d == 64, "qk_norm_rope_bwd_wgrad: head_dim %d not built", d);
  const double n = (double)B * S_part * H * d;
  ProfScope ps(stream, ST355_K_QK_ROPE, 34.0 * n, 12.0 * n);
  dim3 grid((S_part + 63) / 64, H, B), block(256);
  const int nblk = (int)(grid.x * grid.y * grid.z);
  // two-level fixed-order reduction of the nblk per-workgroup partials: <= 64 slices per weight, then the slices
  const int ns = nblk < 64 * 32 ? (nblk + 31) / 32 : 64, per = (nblk + ns - 1) / ns;
  float* part2 = (float*)workspace + (size_t)2 * nblk * d;
  if (d == 128) {
    hipLaunchKernelGGL(k_qk_norm_rope_bwd<128>, grid, block, 0, (hipStream_t)stream, (const bf16*)dQ, (const bf16*)dK, (const bf16*)qkv, ld_qkv,
                       (const bf16*)wq, (const bf16*)wk, cos, sin, (bf16*)dqkv, ld_dqkv, H, S_part, pos0, S, eps, (float*)workspace);
    hipLaunchKernelGGL(k_qk_norm_wgrad_slices<128>, dim3(2, ns), dim3(512), 0, (hipStream_t)stream, (const float*)workspace, nblk, per, part2);
    hipLaunchKernelGGL(k_qk_norm_wgrad_final<128>, dim3(2), dim3(512), 0, (hipStream_t)stream, (const float*)part2, ns, (bf16*)gwq, (bf16*)gwk, accumulate);
  } else {
    hipLaunchKernelGGL(k_qk_norm_rope_bwd<64>, grid, block, 0, (hipStream_t)stream, (const bf16*)dQ, (const bf16*)dK, (const bf16*)qkv, ld_qkv,
                       (const bf16*)wq, (const bf16*)wk, cos, sin, (bf16*)dqkv, ld_dqkv, H, S_part, pos0, S, eps, (float*)workspace);
    hipLaunchKernelGGL(k_qk_norm_wgrad_slices<64>, dim3(2, ns), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, per, part2);
    hipLaunchKernelGGL(k_qk_norm_wgrad_final<64>, dim3(2), dim3(256), 0, (hipStream_t)stream, (const float*)part2, ns, (bf16*)gwq, (bf16*)gwk, accumulate);
  }
  return st355_check_launch("qk_norm_rope_bwd_wgrad");
}

// backward of the FUSED projection epilogue (ST355_EPI_QK_NORM_ROPE): the pre-norm q / k are never stored, so the backward starts from what the
// attention backward keeps anyway — the roped head-major Q / K (z) — and the 1/rms the epilogue wrote:
//   y = R^T z (the rotation is orthogonal),  x_hat = y / w,  dy = R^T dz,  dx = r * (w * dy - x_hat * mean(dy * y))      [w * dy * x_hat = dy * y]
// With no norm weight (w == NULL): dx = dy.  Channels whose norm weight is exactly 0 have no recoverable x_hat (their y is 0): the caller keeps the
// unfused path for such weights (FluxTransformer2DModel checks min |w| when it prepares for training).
template <int HD>
__global__ void __launch_bounds__(256) k_qk_rope_norm_bwd_z(const bf16* __restrict__ dQ, const bf16* __restrict__ dK, const bf16* __restrict__ Qz,
                                                           const bf16* __restrict__ Kz, const float* __restrict__ rrms,
                                                           const bf16* __restrict__ wq, const bf16* __restrict__ wk,
                                                           const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                           bf16* __restrict__ dqkv, int64_t ldd, int H, int S_part, int pos0, int S) {
  constexpr int TPR = HD / 8;
  constexpr int TOK_PER_PASS = 256 / TPR;
  const int tid = threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * 64;
  const int c = tid % TPR;
  const int64_t Dm = (int64_t)H * HD;
  const int64_t bh = (int64_t)b * H + h;
  float wv[2][8], wi[2][8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    wv[0][j] = wq ? bf2f(wq[c * 8 + j]) : 1.f;
    wv[1][j] = wk ? bf2f(wk[c * 8 + j]) : 1.f;
    wi[0][j] = 1.f / wv[0][j];
    wi[1][j] = 1.f / wv[1][j];
  }
  for (int tl = tid / TPR; tl < 64; tl += TOK_PER_PASS) {
    const int t = t0 + tl;
    const bool valid = t < S_part;
    const int tt = valid ? t : S_part - 1;
    const int pos = pos0 + tt;
    const float* cp = cosT + (int64_t)pos * HD + c * 8;
    const float* sp = sinT + (int64_t)pos * HD + c * 8;
    float cs[8], sn[8];
    *(f32x4*)&cs[0] = *(const f32x4*)cp; *(f32x4*)&cs[4] = *(const f32x4*)(cp + 4);
    *(f32x4*)&sn[0] = *(const f32x4*)sp; *(f32x4*)&sn[4] = *(const f32x4*)(sp + 4);
    bf16* drow = dqkv + ((int64_t)b * S + pos) * ldd + (int64_t)h * HD + c * 8;
    const int64_t go = (bh * S + pos) * HD + c * 8;
    const float* rrow = rrms + ((int64_t)b * S + pos) * (2 * H) + h;
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const bool has_norm = (w == 0) ? (wq != nullptr) : (wk != nullptr);
      const bf16x8 gv = *(const bf16x8*)((w == 0 ? dQ : dK) + go);
      const bf16x8 zv = *(const bf16x8*)((w == 0 ? Qz : Kz) + go);
      float g[8], z[8], dy[8], y[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { g[j] = bf2f(gv[j]); z[j] = bf2f(zv[j]); }
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        // forward: o0 = y0*c0 - y1*s0 ; o1 = y1*c1 + y0*s1   =>  transpose (= inverse): v0 = o0*c0 + o1*s1 ; v1 = o1*c1 - o0*s0
        dy[j] = g[j] * cs[j] + g[j + 1] * sn[j + 1];
        dy[j + 1] = g[j + 1] * cs[j + 1] - g[j] * sn[j];
        y[j] = z[j] * cs[j] + z[j + 1] * sn[j + 1];
        y[j + 1] = z[j + 1] * cs[j + 1] - z[j] * sn[j];
      }
      bf16x8 o;
      if (has_norm) {
        float sdy = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) sdy += dy[j] * y[j];
        const float r = rrow[w * H];
        const float mdy = group_sum<HD>(sdy) / (float)HD;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = f2bf(r * (wv[w][j] * dy[j] - y[j] * wi[w][j] * mdy));
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = f2bf(dy[j]);
      }
      if (valid) *(bf16x8*)(drow + w * Dm) = o;
    }
  }
}

extern "C" int st355_qk_rope_norm_bwd(void* stream, const void* dQ, const void* dK, const void* Q, const void* K, const float* rrms, const void* wq,
                                      const void* wk, const float* cos, const float* sin, void* dqkv, int64_t ld_dqkv, int B, int H, int d,
                                      int S_part, int pos0, int S) {
  ST_REQUIRE(dQ && dK && Q && K && rrms && cos && sin && dqkv, "qk_rope_norm_bwd: null pointer");
  ST_REQUIRE(ld_dqkv % 8 == 0 && pos0 + S_part <= S && S_part > 0, "qk_rope_norm_bwd: bad shape");
  ST_REQUIRE(d == 128, "qk_rope_norm_bwd: head_dim %d not built (the fused projection epilogue is head_dim 128)", d);
  const double n = (double)B * S_part * H * d;
  ProfScope ps(stream, ST355_K_QK_ROPE, 34.0 * n, 12.0 * n);
  dim3 grid((S_part + 63) / 64, H, B), block(256);
  hipLaunchKernelGGL(k_qk_rope_norm_bwd_z<128>, grid, block, 0, (hipStream_t)stream, (const bf16*)dQ, (const bf16*)dK, (const bf16*)Q, (const bf16*)K,
                     rrms, (const bf16*)wq, (const bf16*)wk, cos, sin, (bf16*)dqkv, ld_dqkv, H, S_part, pos0, S);
  return st355_check_launch("qk_rope_norm_bwd");
}

// ================================================================================================
// plain head split / merge (no norm, no RoPE): the UNet's attention (diffusers Attention with AttnProcessor2_0: q,k,v -> [B,H,S,d]).
//   split: src [B*S, ld] token-major (the caller offsets the pointer to the q / k / v column block) -> X [B,H,S,d] and/or Xt [B,H,d,Sp]
//   merge: dX [B,H,S,d] -> dst [B*S, ld] token-major
// ================================================================================================
template <int HD>
__global__ void __launch_bounds__(256) k_head_split(const bf16* __restrict__ src, int64_t ld, bf16* __restrict__ X, bf16* __restrict__ Xt, int H, int S, int Sp, int dsrc) {
  constexpr int TPR = (HD / 8 <= 8) ? 8 : 16;          // lanes per token row (power of two; head_dim 96 leaves 4 of 16 idle)
  constexpr int TOK_PER_PASS = 256 / TPR;
  __shared__ __attribute__((aligned(16))) bf16 tile[64 * TP];
  const int tid = threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * 64;
  const int c = tid % TPR;
  const int64_t bh = (int64_t)b * H + h;
  for (int tl = tid / TPR; tl < 64; tl += TOK_PER_PASS) {
    if (c >= HD / 8) continue;
    const int t = t0 + tl;
    const bool valid = t < S;
    const int tt = valid ? t : S - 1;
    bf16x8 v;                                    // dsrc < HD: the source heads are dsrc wide, the rest of the HD-wide head is zero padding
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = f2bf(0.f);
    if (c * 8 < dsrc) v = *(const bf16x8*)(src + ((int64_t)b * S + tt) * ld + (int64_t)h * dsrc + c * 8);
    if (valid && X) *(bf16x8*)(X + (bh * S + t) * HD + c * 8) = v;
    uint32_t* tq = (uint32_t*)(&tile[tl * TP + c * 8]);
    const u32x4 w = *(const u32x4*)&v;
#pragma unroll
    for (int j = 0; j < 4; j++) tq[j] = w[j];
  }
  if (!Xt) return;
  __syncthreads();
  const int nvalid = min(64, S - t0);
  for (int i = tid; i < HD * 8; i += 256) {
    const int d = i >> 3, tc = i & 7;
    bf16* dst = Xt + (bh * HD + d) * (int64_t)Sp + t0 + tc * 8;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = tile[(tc * 8 + e) * TP + d];
    if (tc * 8 + 8 <= nvalid) {
      *(bf16x8*)dst = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++)
        if (tc * 8 + e < nvalid) dst[e] = o[e];
    }
  }
}
template <int HD>
__global__ void __launch_bounds__(256) k_head_merge(const bf16* __restrict__ dX, bf16* __restrict__ dst, int64_t ld, int H, int S, int dsrc) {
  constexpr int TPR = (HD / 8 <= 8) ? 8 : 16;
  const int b = blockIdx.z, h = blockIdx.y;
  const int t = blockIdx.x * (256 / TPR) + threadIdx.x / TPR, c = threadIdx.x % TPR;
  if (t >= S || c * 8 >= dsrc) return;
  const int64_t bh = (int64_t)b * H + h;
  *(bf16x8*)(dst + ((int64_t)b * S + t) * ld + (int64_t)h * dsrc + c * 8) = *(const bf16x8*)(dX + (bh * S + t) * HD + c * 8);
}
// d_src <= d: token-major heads of width d_src land zero-padded in d-wide head-major rows (SD1.5's 40 / 80-wide heads on the 64 / 96 kernels)
extern "C" int st355_head_split_pad(void* stream, const void* src, int64_t ld, void* X, void* Xt, int B, int H, int d_src, int d, int S, int Sp) {
  ST_REQUIRE(src && (X || Xt) && ld % 8 == 0 && S > 0 && (!Xt || (Sp % 64 == 0 && Sp >= S)) && d_src > 0 && d_src <= d && d_src % 8 == 0, "head_split: bad args");
  ST_REQUIRE(d == 128 || d == 64 || d == 96, "head_split: head_dim %d not built", d);
  ST_REQUIRE(((uintptr_t)src % 16) == 0, "head_split: misaligned source");
  const double n = (double)B * S * H * d;
  ProfScope ps(stream, ST355_K_QK_ROPE, 0.0, (2.0 + (X ? 2.0 : 0.0) + (Xt ? 2.0 : 0.0)) * n);
  dim3 grid((S + 63) / 64, H, B), block(256);
  if (d == 96) hipLaunchKernelGGL(k_head_split<96>, grid, block, 0, (hipStream_t)stream, (const bf16*)src, ld, (bf16*)X, (bf16*)Xt, H, S, Sp, d_src);
  else if (d == 128) hipLaunchKernelGGL(k_head_split<128>, grid, block, 0, (hipStream_t)stream, (const bf16*)src, ld, (bf16*)X, (bf16*)Xt, H, S, Sp, d_src);
  else hipLaunchKernelGGL(k_head_split<64>, grid, block, 0, (hipStream_t)stream, (const bf16*)src, ld, (bf16*)X, (bf16*)Xt, H, S, Sp, d_src);
  return st355_check_launch("head_split");
}
extern "C" int st355_head_split(void* stream, const void* src, int64_t ld, void* X, void* Xt, int B, int H, int d, int S, int Sp) {
  return st355_head_split_pad(stream, src, ld, X, Xt, B, H, d, d, S, Sp);
}
extern "C" int st355_head_merge_pad(void* stream, const void* dX, void* dst, int64_t ld, int B, int H, int d_src, int d, int S) {
  ST_REQUIRE(dX && dst && ld % 8 == 0 && S > 0 && d_src > 0 && d_src <= d && d_src % 8 == 0, "head_merge: bad args");
  ST_REQUIRE(d == 128 || d == 64 || d == 96, "head_merge: head_dim %d not built", d);
  ST_REQUIRE(((uintptr_t)dst % 16) == 0, "head_merge: misaligned destination");
  const double n = (double)B * S * H * d;
  ProfScope ps(stream, ST355_K_QK_ROPE, 0.0, 4.0 * n);
  const int tpb = 256 / (d / 8 <= 8 ? 8 : 16);
  dim3 grid((S + tpb - 1) / tpb, H, B), block(256);
  if (d == 96) hipLaunchKernelGGL(k_head_merge<96>, grid, block, 0, (hipStream_t)stream, (const bf16*)dX, (bf16*)dst, ld, H, S, d_src);
  else if (d == 128) hipLaunchKernelGGL(k_head_merge<128>, grid, block, 0, (hipStream_t)stream, (const bf16*)dX, (bf16*)dst, ld, H, S, d_src);
  else hipLaunchKernelGGL(k_head_merge<64>, grid, block, 0, (hipStream_t)stream, (const bf16*)dX, (bf16*)dst, ld, H, S, d_src);
  return st355_check_launch("head_merge");
}
extern "C" int st355_head_merge(void* stream, const void* dX, void* dst, int64_t ld, int B, int H, int d, int S) {
  return st355_head_merge_pad(stream, dX, dst, ld, B, H, d, d, S);
}
