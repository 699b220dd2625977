// elementwise.hip — HBM-streaming kernels of the step: noising + target (K1), MSE + dL/dpred (K13),
// Flux pack/unpack (K3), timestep projection / SiLU / add (K4), gated-residual backward scaling.
// All are bandwidth-bound: 16-byte vector accesses, grid-stride loops, no LDS staging.
#include "common.h"

// zero-fill as a KERNEL (not hipMemsetAsync): memset nodes inside a captured hipGraph were observed to misbehave on replay (ROCm 7.2), and the
// trainer replays the whole predict + loss + backward as one graph
static __global__ void k_zero_words(uint32_t* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
static inline void zero_words(void* stream, void* p, int n_words) {
  hipLaunchKernelGGL(k_zero_words, dim3((n_words + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint32_t*)p, n_words);
}


#define EW_THREADS 256
static inline int ew_blocks(int64_t work_items) {
  int64_t b = cdiv64(work_items, EW_THREADS);
  if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  float r = sqrtf(-2.0f * __logf(u1));
  float s, c;
  __sincosf(6.283185307179586f * u2, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

// ---- K1: flow noising + target --------------------------------------------------------------------
template <bool GEN>
__global__ void __launch_bounds__(EW_THREADS) k_flow_noise_mix(const bf16* __restrict__ x, const bf16* __restrict__ noise,
                                                              const float* __restrict__ sigma, bf16* __restrict__ xt,
                                                              bf16* __restrict__ target, bf16* __restrict__ noise_out,
                                                              int64_t batch, int64_t per_sample, uint64_t seed,
                                                              uint64_t offset) {
  const int64_t nvec = (batch * per_sample) >> 3;  // 8 elements per item
  const int64_t vec_per_sample = per_sample >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / vec_per_sample;
    const float s = sigma[b];
    bf16x8 xv = *(const bf16x8*)(x + i * 8);
    float n[8];
    if (GEN) {
      uint32_t r[4];
      philox4x32_10(2 * (uint64_t)i + offset, seed, r);
      box_muller(r[0], r[1], n[0], n[1]);
      box_muller(r[2], r[3], n[2], n[3]);
      philox4x32_10(2 * (uint64_t)i + 1 + offset, seed, r);
      box_muller(r[0], r[1], n[4], n[5]);
      box_muller(r[2], r[3], n[6], n[7]);
      // the reference's noise is a weight-dtype tensor (randn_like(latents)): round to bf16 first
#pragma unroll
      for (int j = 0; j < 8; j++) n[j] = bf2f(f2bf(n[j]));
    } else {
      bf16x8 nv = *(const bf16x8*)(noise + i * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) n[j] = bf2f(nv[j]);
    }
    bf16x8 o_xt, o_tg, o_n;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float xf = bf2f(xv[j]);
      o_xt[j] = f2bf((1.0f - s) * xf + s * n[j]);
      o_tg[j] = f2bf(n[j] - xf);
      o_n[j] = f2bf(n[j]);
    }
    *(bf16x8*)(xt + i * 8) = o_xt;
    if (target) *(bf16x8*)(target + i * 8) = o_tg;
    if (GEN && noise_out) *(bf16x8*)(noise_out + i * 8) = o_n;
  }
}

extern "C" int st355_flow_noise_mix(void* stream, const void* x, const void* noise, const float* sigma, void* x_t,
                                    void* target, void* noise_out, int64_t batch, int64_t per_sample, uint64_t seed,
                                    uint64_t offset) {
  ST_REQUIRE(x && sigma && x_t, "flow_noise_mix: null pointer");
  ST_REQUIRE(per_sample % 8 == 0 && batch > 0, "flow_noise_mix: per_sample (%ld) must be a multiple of 8", (long)per_sample);
  const int64_t n = batch * per_sample;
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 4.0 * n, (noise ? 8.0 : 6.0) * n);
  int blocks = ew_blocks(n / 8);
  if (noise)
    hipLaunchKernelGGL(k_flow_noise_mix<false>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)noise, sigma, (bf16*)x_t, (bf16*)target, (bf16*)noise_out, batch, per_sample, seed, offset);
  else
    hipLaunchKernelGGL(k_flow_noise_mix<true>, dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)noise, sigma, (bf16*)x_t, (bf16*)target, (bf16*)noise_out, batch, per_sample, seed, offset);
  return st355_check_launch("flow_noise_mix");
}

__global__ void __launch_bounds__(EW_THREADS) k_ddpm_noise_mix(const bf16* __restrict__ x, const bf16* __restrict__ noise,
                                                              const float* __restrict__ sa, const float* __restrict__ ss,
                                                              bf16* __restrict__ xt, bf16* __restrict__ vt, int64_t batch,
                                                              int64_t per_sample) {
  const int64_t nvec = (batch * per_sample) >> 3;
  const int64_t vec_per_sample = per_sample >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / vec_per_sample;
    const float a = sa[b], s = ss[b];
    bf16x8 xv = *(const bf16x8*)(x + i * 8);
    bf16x8 nv = *(const bf16x8*)(noise + i * 8);
    bf16x8 o_xt, o_v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float xf = bf2f(xv[j]), nf = bf2f(nv[j]);
      o_xt[j] = f2bf(a * xf + s * nf);
      o_v[j] = f2bf(a * nf - s * xf);
    }
    *(bf16x8*)(xt + i * 8) = o_xt;
    if (vt) *(bf16x8*)(vt + i * 8) = o_v;
  }
}
extern "C" int st355_ddpm_noise_mix(void* stream, const void* x, const void* noise, const float* sqrt_acp,
                                    const float* sqrt_1macp, void* x_t, void* v_target, int64_t batch, int64_t per_sample) {
  ST_REQUIRE(x && noise && sqrt_acp && sqrt_1macp && x_t, "ddpm_noise_mix: null pointer");
  ST_REQUIRE(per_sample % 8 == 0 && batch > 0, "ddpm_noise_mix: per_sample must be a multiple of 8");
  const int64_t n = batch * per_sample;
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 6.0 * n, 8.0 * n);
  hipLaunchKernelGGL(k_ddpm_noise_mix, dim3(ew_blocks(n / 8)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)x,
                     (const bf16*)noise, sqrt_acp, sqrt_1macp, (bf16*)x_t, (bf16*)v_target, batch, per_sample);
  return st355_check_launch("ddpm_noise_mix");
}

// ---- K13: MSE ---------------------------------------------------------------------------------
// optional per-element loss mask (common.py:6402-6424: the conditioning mask, [B,1,H,W] broadcast over the channels): the elementwise loss is
// multiplied by emask[b, i % mask_period] before the per-sample mean; mask_period (= H*W) is a multiple of 8
__device__ __forceinline__ void load_mask8(const float* __restrict__ em, int64_t off, float (&mk)[8]) {
  if (em) {
    const f32x4 a = *(const f32x4*)(em + off), b = *(const f32x4*)(em + off + 4);
#pragma unroll
    for (int j = 0; j < 4; j++) { mk[j] = a[j]; mk[j + 4] = b[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) mk[j] = 1.f;
  }
}
// grid = (1, B), LOSS_THREADS threads: ONE workgroup per sample reduces it in a fixed order (lane tree -> 16 wave partials summed in wave order): the logged
// loss is bit-reproducible run to run.  (r02 used several blocks per sample and an atomicAdd: the last bits then depended on block arrival order.)
#define LOSS_THREADS 1024
__global__ void __launch_bounds__(LOSS_THREADS) k_mse(const bf16* __restrict__ pred, const bf16* __restrict__ target,
                                                   const float* __restrict__ weight, float* __restrict__ per_sample_acc,
                                                   bf16* __restrict__ dpred, int64_t per_sample, float dscale,
                                                   const float* __restrict__ emask, int64_t mask_period) {
  const int b = blockIdx.y;
  const float w = weight ? weight[b] : 1.0f;
  const int64_t vecs = per_sample >> 3;
  const bf16* p = pred + (int64_t)b * per_sample;
  const bf16* t = target + (int64_t)b * per_sample;
  bf16* d = dpred ? dpred + (int64_t)b * per_sample : nullptr;
  const float* em = emask ? emask + (int64_t)b * mask_period : nullptr;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (int64_t)gridDim.x * blockDim.x) {
    bf16x8 pv = *(const bf16x8*)(p + i * 8);
    bf16x8 tv = *(const bf16x8*)(t + i * 8);
    bf16x8 dv;
    float mk[8];
    load_mask8(em, (i * 8) % (em ? mask_period : 8), mk);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float df = bf2f(pv[j]) - bf2f(tv[j]);
      acc += df * df * mk[j];
      dv[j] = f2bf(dscale * w * df * mk[j]);
    }
    if (d) *(bf16x8*)(d + i * 8) = dv;
  }
  acc = wave_sum(acc);
  __shared__ float red[LOSS_THREADS / WAVE];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x / WAVE); i++) s += red[i];
    per_sample_acc[b] = s * w;
  }
}
__global__ void k_mse_finalize(float* per_sample_acc, float* loss, int B, float inv_per_sample) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; b++) {
      float v = per_sample_acc[b] * inv_per_sample;
      per_sample_acc[b] = v;
      s += v;
    }
    loss[0] = s / (float)B;
  }
}
static int mse_loss_impl(void* stream, const void* pred, const void* target, const float* weight, float* loss_out,
                         float* per_sample_out, void* dpred, int64_t batch, int64_t per_sample, float grad_scale, const float* emask, int64_t mask_period) {
  ST_REQUIRE(pred && target && loss_out && per_sample_out, "mse_loss: null pointer (per_sample_out is required scratch)");
  ST_REQUIRE(per_sample % 8 == 0 && batch > 0 && batch < 65536, "mse_loss: bad shape");
  ST_REQUIRE(!emask || (mask_period > 0 && mask_period % 8 == 0 && per_sample % mask_period == 0 && ((uintptr_t)emask & 15) == 0),
             "mse_loss: element mask period must be a multiple of 8 dividing per_sample (16-byte aligned)");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 3.0 * batch * per_sample, (dpred ? 6.0 : 4.0) * batch * per_sample);
  const float dscale = grad_scale * 2.0f / ((float)per_sample * (float)batch);
  hipLaunchKernelGGL(k_mse, dim3(1, (unsigned)batch), dim3(LOSS_THREADS), 0, (hipStream_t)stream, (const bf16*)pred,
                     (const bf16*)target, weight, per_sample_out, (bf16*)dpred, per_sample, dscale, emask, mask_period);
  hipLaunchKernelGGL(k_mse_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, per_sample_out, loss_out, (int)batch,
                     1.0f / (float)per_sample);
  return st355_check_launch("mse_loss");
}
extern "C" int st355_mse_loss(void* stream, const void* pred, const void* target, const float* weight, float* loss_out,
                              float* per_sample_out, void* dpred, int64_t batch, int64_t per_sample, float grad_scale) {
  return mse_loss_impl(stream, pred, target, weight, loss_out, per_sample_out, dpred, batch, per_sample, grad_scale, nullptr, 0);
}

// ---- K13b: conditional_loss (common.py:6132-6166): l2 | huber | smooth_l1, reduction "none" -> per-sample mean -> batch mean --------
//   huber:     2c (sqrt(d^2 + c^2) - c)        d/dpred = 2c d / sqrt(d^2 + c^2)
//   smooth_l1: 2  (sqrt(d^2 + c^2) - c)        d/dpred = 2  d / sqrt(d^2 + c^2)
// huber_c is per sample (scheduled huber, common.py:6252-6272) or one value broadcast by the host; weight[b] as in st355_mse_loss.
template <int TYPE>
__global__ void __launch_bounds__(LOSS_THREADS) k_cond_loss(const bf16* __restrict__ pred, const bf16* __restrict__ target,
                                                         const float* __restrict__ weight, const float* __restrict__ huber_c,
                                                         float* __restrict__ per_sample_acc, bf16* __restrict__ dpred,
                                                         int64_t per_sample, float dscale, const float* __restrict__ emask, int64_t mask_period) {
  const int b = blockIdx.y;
  const float w = weight ? weight[b] : 1.0f;
  const float* em = emask ? emask + (int64_t)b * mask_period : nullptr;
  const float c = huber_c[b], c2 = c * c;
  const float k = (TYPE == 1) ? 2.f * c : 2.f;
  const int64_t vecs = per_sample >> 3;
  const bf16* p = pred + (int64_t)b * per_sample;
  const bf16* t = target + (int64_t)b * per_sample;
  bf16* d = dpred ? dpred + (int64_t)b * per_sample : nullptr;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (int64_t)gridDim.x * blockDim.x) {
    bf16x8 pv = *(const bf16x8*)(p + i * 8);
    bf16x8 tv = *(const bf16x8*)(t + i * 8);
    bf16x8 dv;
    float mk[8];
    load_mask8(em, (i * 8) % (em ? mask_period : 8), mk);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float df = bf2f(pv[j]) - bf2f(tv[j]);
      const float r = sqrtf(df * df + c2);
      acc += k * (r - c) * mk[j];
      dv[j] = f2bf(0.5f * dscale * w * k * df / r * mk[j]);      // dscale carries the 2/(n B) of the l2 convention: undo the 2
    }
    if (d) *(bf16x8*)(d + i * 8) = dv;
  }
  acc = wave_sum(acc);
  __shared__ float red[LOSS_THREADS / WAVE];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x / WAVE); i++) s += red[i];
    per_sample_acc[b] = s * w;
  }
}
extern "C" int st355_cond_loss_masked(void* stream, const void* pred, const void* target, const float* weight, const float* huber_c, int loss_type,
                                      const float* emask, int64_t mask_period, float* loss_out, float* per_sample_out, void* dpred, int64_t batch,
                                      int64_t per_sample, float grad_scale) {
  if (loss_type == 0) return mse_loss_impl(stream, pred, target, weight, loss_out, per_sample_out, dpred, batch, per_sample, grad_scale, emask, mask_period);
  ST_REQUIRE(pred && target && loss_out && per_sample_out && huber_c, "cond_loss: null pointer");
  ST_REQUIRE((loss_type == 1 || loss_type == 2) && per_sample % 8 == 0 && batch > 0 && batch < 65536, "cond_loss: bad args");
  ST_REQUIRE(!emask || (mask_period > 0 && mask_period % 8 == 0 && per_sample % mask_period == 0 && ((uintptr_t)emask & 15) == 0),
             "cond_loss: element mask period must be a multiple of 8 dividing per_sample (16-byte aligned)");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 6.0 * batch * per_sample, (dpred ? 6.0 : 4.0) * batch * per_sample);
  const float dscale = grad_scale * 2.0f / ((float)per_sample * (float)batch);
  if (loss_type == 1)
    hipLaunchKernelGGL(k_cond_loss<1>, dim3(1, (unsigned)batch), dim3(LOSS_THREADS), 0, (hipStream_t)stream, (const bf16*)pred,
                       (const bf16*)target, weight, huber_c, per_sample_out, (bf16*)dpred, per_sample, dscale, emask, mask_period);
  else
    hipLaunchKernelGGL(k_cond_loss<2>, dim3(1, (unsigned)batch), dim3(LOSS_THREADS), 0, (hipStream_t)stream, (const bf16*)pred,
                       (const bf16*)target, weight, huber_c, per_sample_out, (bf16*)dpred, per_sample, dscale, emask, mask_period);
  hipLaunchKernelGGL(k_mse_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, per_sample_out, loss_out, (int)batch,
                     1.0f / (float)per_sample);
  return st355_check_launch("cond_loss");
}
extern "C" int st355_cond_loss(void* stream, const void* pred, const void* target, const float* weight, const float* huber_c, int loss_type,
                               float* loss_out, float* per_sample_out, void* dpred, int64_t batch, int64_t per_sample, float grad_scale) {
  return st355_cond_loss_masked(stream, pred, target, weight, huber_c, loss_type, nullptr, 0, loss_out, per_sample_out, dpred, batch, per_sample, grad_scale);
}

// ---- K3: Flux pack / unpack -------------------------------------------------------------------------
// packed[b, h2*W2 + w2, c*4 + dh*2 + dw] = lat[b, c, 2*h2 + dh, 2*w2 + dw]
template <bool PACK, int ORDER>
__global__ void __launch_bounds__(EW_THREADS) k_patchify(const bf16* __restrict__ src, bf16* __restrict__ dst, int B, int C,
                                                        int H, int W) {
  // 2x2 patches.  ORDER 0: feature = c*4 + dh*2 + dw  (Flux pack_latents == Conv2d(k=2,s=2) im2col, weight.flatten(1) order)
  //               ORDER 1: feature = (dh*2 + dw)*C + c (the "nhwpqc->nchpwq" unpatchify of SD3 / PixArt, sd3/transformer.py:879-902)
  const int64_t total = (int64_t)B * C * H * W;
  const int H2 = H >> 1, W2 = W >> 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // i indexes the PACKED layout (coalesced on the packed side)
    int64_t r = i;
    const int ch4 = (int)(r % (4 * C)); r /= (4 * C);
    const int w2 = (int)(r % W2); r /= W2;
    const int h2 = (int)(r % H2); r /= H2;
    const int b = (int)r;
    int c, dh, dw;
    if (ORDER == 0) { c = ch4 >> 2; dh = (ch4 >> 1) & 1; dw = ch4 & 1; }
    else { c = ch4 % C; dh = (ch4 / C) >> 1; dw = (ch4 / C) & 1; }
    const int64_t li = (((int64_t)b * C + c) * H + (2 * h2 + dh)) * W + (2 * w2 + dw);
    if (PACK) dst[i] = src[li];
    else dst[li] = src[i];
  }
}
extern "C" int st355_patchify(void* stream, const void* latents, void* packed, int B, int C, int H, int W, int order) {
  ST_REQUIRE(latents && packed && (H % 2 == 0) && (W % 2 == 0) && (order == 0 || order == 1), "patchify: bad args");
  const int64_t n = (int64_t)B * C * H * W;
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0, 4.0 * n);
  if (order == 0)
    hipLaunchKernelGGL((k_patchify<true, 0>), dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)latents, (bf16*)packed, B, C, H, W);
  else
    hipLaunchKernelGGL((k_patchify<true, 1>), dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)latents, (bf16*)packed, B, C, H, W);
  return st355_check_launch("patchify");
}
extern "C" int st355_unpatchify(void* stream, const void* packed, void* latents, int B, int C, int H, int W, int order) {
  ST_REQUIRE(latents && packed && (H % 2 == 0) && (W % 2 == 0) && (order == 0 || order == 1), "unpatchify: bad args");
  const int64_t n = (int64_t)B * C * H * W;
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0, 4.0 * n);
  if (order == 0)
    hipLaunchKernelGGL((k_patchify<false, 0>), dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)packed, (bf16*)latents, B, C, H, W);
  else
    hipLaunchKernelGGL((k_patchify<false, 1>), dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)packed, (bf16*)latents, B, C, H, W);
  return st355_check_launch("unpatchify");
}
extern "C" int st355_flux_pack(void* stream, const void* latents, void* packed, int B, int C, int H, int W) {
  return st355_patchify(stream, latents, packed, B, C, H, W, 0);
}
extern "C" int st355_flux_unpack(void* stream, const void* packed, void* latents, int B, int C, int H, int W) {
  return st355_unpatchify(stream, packed, latents, B, C, H, W, 0);
}

// ---- K4 helpers ------------------------------------------------------------------------------------
// Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): out = [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(1e4) i / half)
__global__ void k_timestep_proj(const float* __restrict__ t, bf16* __restrict__ out, int B, int dim, float scale) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t[b] * scale * f;
  out[(int64_t)b * dim + k] = f2bf(cosf(a));
  out[(int64_t)b * dim + half + k] = f2bf(sinf(a));
}
extern "C" int st355_timestep_proj(void* stream, const float* t, void* out, int B, int dim, float scale) {
  ST_REQUIRE(t && out && dim % 2 == 0 && B > 0, "timestep_proj: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0, 2.0 * B * dim);
  int n = B * (dim / 2);
  hipLaunchKernelGGL(k_timestep_proj, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, (bf16*)out, B, dim, scale);
  return st355_check_launch("timestep_proj");
}

template <int OP>  // 0 silu, 1 add
__global__ void __launch_bounds__(EW_THREADS) k_unary_binary(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                            bf16* __restrict__ y, int64_t n) {
  const int64_t nvec = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    bf16x8 av = *(const bf16x8*)(a + i * 8), bv, o;
    if (OP == 1 || OP == 2) bv = *(const bf16x8*)(b + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float f = bf2f(av[j]);
      if (OP == 0) o[j] = f2bf(f / (1.f + __expf(-f)));
      else if (OP == 3) o[j] = f2bf(gelu_tanh(f));
      else if (OP == 1) o[j] = f2bf(f + bf2f(bv[j]));
      else { const float sg = 1.f / (1.f + __expf(-f)); o[j] = f2bf(bf2f(bv[j]) * sg * (1.f + f * (1.f - sg))); }   // OP 2: dy * silu'(x)
    }
    *(bf16x8*)(y + i * 8) = o;
  }
  // tail (n not a multiple of 8)
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    int64_t i = (nvec << 3) + threadIdx.x;
    float f = bf2f(a[i]);
    if (OP == 0) y[i] = f2bf(f / (1.f + __expf(-f)));
    else if (OP == 3) y[i] = f2bf(gelu_tanh(f));
    else if (OP == 1) y[i] = f2bf(f + bf2f(b[i]));
    else { const float sg = 1.f / (1.f + __expf(-f)); y[i] = f2bf(bf2f(b[i]) * sg * (1.f + f * (1.f - sg))); }
  }
}
extern "C" int st355_gelu_tanh(void* stream, const void* x, void* y, int64_t n) {       /* GELU(approximate="tanh") as its own pass (fp8 Linear path: no fused epilogue) */
  ST_REQUIRE(x && y && n > 0, "gelu_tanh: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 10.0 * n, 4.0 * n);
  hipLaunchKernelGGL(k_unary_binary<3>, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)nullptr, (bf16*)y, n);
  return st355_check_launch("gelu_tanh");
}
extern "C" int st355_silu(void* stream, const void* x, void* y, int64_t n) {
  ST_REQUIRE(x && y && n > 0, "silu: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 4.0 * n, 4.0 * n);
  hipLaunchKernelGGL(k_unary_binary<0>, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)x,
                     (const bf16*)nullptr, (bf16*)y, n);
  return st355_check_launch("silu");
}
/* dx = dy * silu'(x) */
extern "C" int st355_silu_bwd(void* stream, const void* x, const void* dy, void* dx, int64_t n) {
  ST_REQUIRE(x && dy && dx && n > 0, "silu_bwd: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 8.0 * n, 6.0 * n);
  hipLaunchKernelGGL(k_unary_binary<2>, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)x,
                     (const bf16*)dy, (bf16*)dx, n);
  return st355_check_launch("silu_bwd");
}
extern "C" int st355_add(void* stream, const void* a, const void* b, void* y, int64_t n) {
  ST_REQUIRE(a && b && y && n > 0, "add: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 1.0 * n, 6.0 * n);
  hipLaunchKernelGGL(k_unary_binary<1>, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)a,
                     (const bf16*)b, (bf16*)y, n);
  return st355_check_launch("add");
}

__global__ void __launch_bounds__(EW_THREADS) k_scale_cols(const bf16* __restrict__ in, int64_t ld_in, const bf16* __restrict__ gate,
                                                          int64_t gate_stride, int64_t rows_per_batch, bf16* __restrict__ out,
                                                          int64_t ld_out, int64_t M, int64_t N) {
  const int64_t nv = N >> 3;
  const int64_t total = M * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / nv, c = (i % nv) * 8;
    const int64_t b = m / rows_per_batch;
    bf16x8 v = *(const bf16x8*)(in + m * ld_in + c);
    bf16x8 g = *(const bf16x8*)(gate + b * gate_stride + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(bf2f(v[j]) * bf2f(g[j]));
    *(bf16x8*)(out + m * ld_out + c) = o;
  }
}
extern "C" int st355_scale_cols(void* stream, const void* in, int64_t ld_in, const void* gate, int64_t gate_stride,
                                int64_t rows_per_batch, void* out, int64_t ld_out, int64_t M, int64_t N) {
  ST_REQUIRE(in && gate && out && N % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && gate_stride % 8 == 0 && rows_per_batch > 0,
             "scale_cols: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 1.0 * M * N, 4.0 * M * N);
  hipLaunchKernelGGL(k_scale_cols, dim3(ew_blocks(M * N / 8)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)in, ld_in,
                     (const bf16*)gate, gate_stride, rows_per_batch, (bf16*)out, ld_out, M, N);
  return st355_check_launch("scale_cols");
}

// ---- GEGLU (diffusers FeedForward activation_fn="geglu" inside the UNet's BasicTransformerBlock): proj -> [value | gate], out = value * gelu(gate),
// gelu = the exact erf form (F.gelu default).  h: [M, 2F] row stride ldh; out: [M, F].  Backward writes dh = [dout*gelu(gate) | dout*value*gelu'(gate)].
template <bool BWD>
__global__ void __launch_bounds__(256) k_geglu(const bf16* __restrict__ h, int64_t ldh, const bf16* __restrict__ dout, bf16* __restrict__ out, int64_t ldo, int64_t M,
                                              int F) {
  const int f8 = F / 8;
  const int64_t n = M * f8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / f8;
    const int c = (int)(i % f8) * 8;
    const bf16x8 v = *(const bf16x8*)(h + m * ldh + c), g = *(const bf16x8*)(h + m * ldh + F + c);
    if (!BWD) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = f2bf(bf2f(v[j]) * gelu_erf(bf2f(g[j])));
      *(bf16x8*)(out + m * ldo + c) = o;
    } else {
      const bf16x8 d = *(const bf16x8*)(dout + m * F + c);
      bf16x8 dv, dg;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float gg = bf2f(g[j]), dd = bf2f(d[j]);
        dv[j] = f2bf(dd * gelu_erf(gg));
        dg[j] = f2bf(dd * bf2f(v[j]) * gelu_erf_grad(gg));
      }
      *(bf16x8*)(out + m * ldo + c) = dv;
      *(bf16x8*)(out + m * ldo + F + c) = dg;
    }
  }
}
extern "C" int st355_geglu_fwd(void* stream, const void* h, int64_t ldh, void* out, int64_t M, int F) {
  ST_REQUIRE(h && out && M > 0 && F % 8 == 0 && ldh % 8 == 0, "geglu_fwd: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 12.0 * M * F, 6.0 * M * F);
  hipLaunchKernelGGL(k_geglu<false>, dim3(ew_blocks(M * (F / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16*)h, ldh, (const bf16*)nullptr, (bf16*)out,
                     (int64_t)F, M, F);
  return st355_check_launch("geglu_fwd");
}
extern "C" int st355_geglu_bwd(void* stream, const void* h, int64_t ldh, const void* dout, void* dh, int64_t lddh, int64_t M, int F) {
  ST_REQUIRE(h && dout && dh && M > 0 && F % 8 == 0 && ldh % 8 == 0 && lddh % 8 == 0, "geglu_bwd: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 24.0 * M * F, 10.0 * M * F);
  hipLaunchKernelGGL(k_geglu<true>, dim3(ew_blocks(M * (F / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16*)h, ldh, (const bf16*)dout, (bf16*)dh, lddh, M, F);
  return st355_check_launch("geglu_bwd");
}

// ---- in-place row softmax (bf16 storage, fp32 math): the VAE mid-block attention is ONE head of dim 512 over H*W tokens (diffusers Attention with
// heads=1 inside UNetMidBlock2D of AutoencoderKL) — scores are a plain GEMM, this kernel turns them into probabilities, P.V is a plain GEMM.
// One 256-thread block per row; the row (<= 64 Ki elements) is kept in registers between the max / sum / write passes.
template <int MAXC>                            // 256 threads * 8 elements * MAXC chunks cover the row
__global__ void __launch_bounds__(256) k_softmax_rows(bf16* __restrict__ x, int64_t ldx, int n, float scale) {
  __shared__ float red[4];
  bf16* row = x + (int64_t)blockIdx.x * ldx;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float v[MAXC][8];
  const int nch = (n + 2047) / 2048;
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    if (c < nch) {
      const int i = (c * 256 + tid) * 8;
      if (i < n) {
        const bf16x8 t = *(const bf16x8*)(row + i);
#pragma unroll
        for (int j = 0; j < 8; j++) { v[c][j] = (i + j < n) ? bf2f(t[j]) * scale : -INFINITY; m = fmaxf(m, v[c][j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[c][j] = -INFINITY;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; c++)
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 8; j++) { v[c][j] = __expf(v[c][j] - m); s += v[c][j]; }
    }
  s = wave_sum(s);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int c = 0; c < MAXC; c++)
    if (c < nch) {
      const int i = (c * 256 + tid) * 8;
      if (i < n) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = f2bf(v[c][j] * inv);
        *(bf16x8*)(row + i) = o;
      }
    }
}
extern "C" int st355_softmax_rows(void* stream, void* x, int64_t ldx, int64_t rows, int n, float scale) {
  ST_REQUIRE(x && rows > 0 && n > 0 && n % 8 == 0 && n <= 65536 && ldx % 8 == 0, "softmax_rows: n must be a multiple of 8 and <= 65536");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 5.0 * rows * n, 4.0 * rows * n);
#define SM_LAUNCH(MC) hipLaunchKernelGGL(k_softmax_rows<MC>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (bf16*)x, ldx, n, scale)
  const int nch = (n + 2047) / 2048;
  if (nch <= 1) SM_LAUNCH(1);
  else if (nch <= 2) SM_LAUNCH(2);
  else if (nch <= 4) SM_LAUNCH(4);
  else if (nch <= 8) SM_LAUNCH(8);
  else if (nch <= 16) SM_LAUNCH(16);
  else SM_LAUNCH(32);
#undef SM_LAUNCH
  return st355_check_launch("softmax_rows");
}

// softmax backward in place on dp: ds[r,:] = scale * p[r,:] * (dp[r,:] - sum_j dp[r,j] p[r,j])   (the unfused attention of heads wider than 128:
// SD1.5's 160-wide heads at the 16^2 / 8^2 levels — tiny score matrices).  One block per row, n <= 16384.
__global__ void __launch_bounds__(256) k_softmax_rows_bwd(const bf16* __restrict__ p, bf16* __restrict__ dp, int64_t ld, int n, float scale) {
  __shared__ float red[4];
  const bf16* pr = p + (int64_t)blockIdx.x * ld;
  bf16* dr = dp + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float s = 0.f;
  for (int i = tid * 8; i < n; i += 2048) {
    const bf16x8 a = *(const bf16x8*)(pr + i), b = *(const bf16x8*)(dr + i);
#pragma unroll
    for (int j = 0; j < 8; j++) s += bf2f(a[j]) * bf2f(b[j]);
  }
  s = wave_sum(s);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const float dot = red[0] + red[1] + red[2] + red[3];
  for (int i = tid * 8; i < n; i += 2048) {
    const bf16x8 a = *(const bf16x8*)(pr + i), b = *(const bf16x8*)(dr + i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(scale * bf2f(a[j]) * (bf2f(b[j]) - dot));
    *(bf16x8*)(dr + i) = o;
  }
}
extern "C" int st355_softmax_rows_bwd(void* stream, const void* p, void* dp, int64_t ld, int64_t rows, int n, float scale) {
  ST_REQUIRE(p && dp && rows > 0 && n > 0 && n % 8 == 0 && ld % 8 == 0, "softmax_rows_bwd: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 4.0 * rows * n, 6.0 * rows * n);
  hipLaunchKernelGGL(k_softmax_rows_bwd, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)p, (bf16*)dp, ld, n, scale);
  return st355_check_launch("softmax_rows_bwd");
}


// ---- TREAD token routing (helpers/training/tread.py:118-159): per-sample row gather / scatter over [B, S, D] token buffers ---------------------------
// gather : out[b, j, :] = x[b, idx[b, j], :]      (start_route: kept tokens first, truncated to K; also the adjoint of the scatter below)
// scatter: dst[b, idx[b, j], :] = src[b, j, :]    (end_route: routed tokens back into their slots of a copy of the pre-route sequence; idx rows are
//                                                  duplicates-free — a permutation prefix — so plain stores, no atomics)
// HBM-bound row copies: one 16-byte chunk per lane, grid-stride over B*K*(D/8) chunks.
__global__ void __launch_bounds__(EW_THREADS) k_route_rows(const bf16* __restrict__ src, int64_t src_ld, int64_t src_bs, const int* __restrict__ idx,
                                                          bf16* __restrict__ dst, int64_t dst_ld, int64_t dst_bs, int B, int K, int D8, int scatter) {
  const int64_t total = (int64_t)B * K * D8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D8);
    const int64_t r = i / D8;
    const int j = (int)(r % K), b = (int)(r / K);
    const int t = idx[(int64_t)b * K + j];
    const int64_t so = scatter ? (int64_t)b * src_bs + (int64_t)j * src_ld : (int64_t)b * src_bs + (int64_t)t * src_ld;
    const int64_t dof = scatter ? (int64_t)b * dst_bs + (int64_t)t * dst_ld : (int64_t)b * dst_bs + (int64_t)j * dst_ld;
    *(bf16x8*)(dst + dof + c * 8) = *(const bf16x8*)(src + so + c * 8);
  }
}
static int route_rows(void* stream, const void* src, int64_t src_ld, int64_t src_bs, const int* idx, void* dst, int64_t dst_ld, int64_t dst_bs, int B, int K, int D,
                      int scatter, const char* what) {
  ST_REQUIRE(src && idx && dst && B > 0 && K > 0 && D > 0 && D % 8 == 0, "route_rows: bad args (D must be a multiple of 8)");
  ST_REQUIRE(src_ld % 8 == 0 && dst_ld % 8 == 0 && src_bs % 8 == 0 && dst_bs % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
             "route_rows: 16-byte aligned rows");
  const int64_t n = (int64_t)B * K * D;
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 4.0 * n);
  hipLaunchKernelGGL(k_route_rows, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16*)src, src_ld, src_bs, idx, (bf16*)dst, dst_ld,
                     dst_bs, B, K, D / 8, scatter);
  return st355_check_launch(what);
}
extern "C" int st355_gather_rows(void* stream, const void* x, int64_t ld_x, int64_t batch_stride_x, const int* idx, void* out, int64_t ld_out,
                                 int64_t batch_stride_out, int B, int K, int D) {
  return route_rows(stream, x, ld_x, batch_stride_x, idx, out, ld_out, batch_stride_out, B, K, D, 0, "gather_rows");
}
extern "C" int st355_scatter_rows(void* stream, const void* src, int64_t ld_src, int64_t batch_stride_src, const int* idx, void* dst, int64_t ld_dst,
                                  int64_t batch_stride_dst, int B, int K, int D) {
  return route_rows(stream, src, ld_src, batch_stride_src, idx, dst, ld_dst, batch_stride_dst, B, K, D, 1, "scatter_rows");
}
