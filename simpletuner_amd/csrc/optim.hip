// optim.hip — K15/K16/K17: one-pass fused AdamW (+EMA) over a flat parameter arena, standalone EMA,
// gradient norm, and the LoRA operand packer.  All HBM-streaming: 16-byte accesses, one read and one write
// of every state word per step (22-28 B/param depending on layout).
#include "common.h"

#define OP_THREADS 256
static inline int op_blocks(int64_t items) {
  int64_t b = cdiv64(items, OP_THREADS);
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

struct AdamC {
  float lr, beta1, beta2, eps, wd, step_size, bc2_sqrt, grad_scale, ema_omd;  // ema_omd = 1 - ema_decay
};

// torch.optim.AdamW (single-tensor path) order of operations:
//   p *= 1 - lr*wd ; m = lerp(m, g, 1-b1) ; v = v*b2 + (1-b2)*g*g ; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamC& c) {
  g *= c.grad_scale;
  p = p * (1.f - c.lr * c.wd);
  m = m + (g - m) * (1.f - c.beta1);
  v = v * c.beta2 + (1.f - c.beta2) * g * g;
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p - c.step_size * (m / denom);
}

__global__ void __launch_bounds__(OP_THREADS) k_adamw_f32(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, float* __restrict__ ema, bf16* __restrict__ pb,
                                                         int64_t n, AdamC c) {
  const int64_t nv = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = *(f32x4*)(p + i * 4), gv = *(const f32x4*)(g + i * 4), mv = *(f32x4*)(m + i * 4), vv = *(f32x4*)(v + i * 4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float pj = pv[j], mj = mv[j], vj = vv[j];
      adam_one(pj, gv[j], mj, vj, c);
      pv[j] = pj; mv[j] = mj; vv[j] = vj;
    }
    *(f32x4*)(p + i * 4) = pv;
    *(f32x4*)(m + i * 4) = mv;
    *(f32x4*)(v + i * 4) = vv;
    if (ema) {
      f32x4 ev = *(f32x4*)(ema + i * 4);
#pragma unroll
      for (int j = 0; j < 4; j++) ev[j] = ev[j] - c.ema_omd * (ev[j] - pv[j]);
      *(f32x4*)(ema + i * 4) = ev;
    }
    if (pb) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = f2bf(pv[j]);
      *(bf16x4*)(pb + i * 4) = o;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
    const int64_t i = (nv << 2) + threadIdx.x;
    float pv = p[i], mv = m[i], vv = v[i];
    adam_one(pv, g[i], mv, vv, c);
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (ema) ema[i] = ema[i] - c.ema_omd * (ema[i] - pv);
    if (pb) pb[i] = f2bf(pv);
  }
}

__global__ void __launch_bounds__(OP_THREADS) k_adamw_bf16(bf16* __restrict__ p, const bf16* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, bf16* __restrict__ ema, int64_t n, AdamC c) {
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    bf16x8 pb = *(bf16x8*)(p + i * 8), gb = *(const bf16x8*)(g + i * 8);
    f32x4 m0 = *(f32x4*)(m + i * 8), m1 = *(f32x4*)(m + i * 8 + 4), v0 = *(f32x4*)(v + i * 8), v1 = *(f32x4*)(v + i * 8 + 4);
    float pf[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      pf[j] = bf2f(pb[j]); pf[j + 4] = bf2f(pb[j + 4]);
      float ma = m0[j], va = v0[j], mb = m1[j], vb = v1[j];
      adam_one(pf[j], bf2f(gb[j]), ma, va, c);
      adam_one(pf[j + 4], bf2f(gb[j + 4]), mb, vb, c);
      m0[j] = ma; v0[j] = va; m1[j] = mb; v1[j] = vb;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) pb[j] = f2bf(pf[j]);
    *(bf16x8*)(p + i * 8) = pb;
    *(f32x4*)(m + i * 8) = m0; *(f32x4*)(m + i * 8 + 4) = m1;
    *(f32x4*)(v + i * 8) = v0; *(f32x4*)(v + i * 8 + 4) = v1;
    if (ema) {
      bf16x8 eb = *(bf16x8*)(ema + i * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) { float e = bf2f(eb[j]); eb[j] = f2bf(e - c.ema_omd * (e - bf2f(pb[j]))); }
      *(bf16x8*)(ema + i * 8) = eb;
    }
  }
}

static AdamC make_adam(float lr, float b1, float b2, float eps, float wd, int64_t step, float gs, float ema_decay) {
  AdamC c;
  c.lr = lr; c.beta1 = b1; c.beta2 = b2; c.eps = eps; c.wd = wd; c.grad_scale = gs;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  c.step_size = (float)((double)lr / bc1);
  c.bc2_sqrt = (float)sqrt(bc2);
  c.ema_omd = 1.f - ema_decay;
  return c;
}

extern "C" int st355_adamw_ema_step(void* stream, float* p, const float* g, float* m, float* v, float* ema, void* p_bf16, int64_t n,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                                    float ema_decay) {
  ST_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_ema_step: bad args");
  ST_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
             "adamw_ema_step: arena must be 16-byte aligned");
  ProfScope ps(stream, ST355_K_OPTIM, 12.0 * n, (28.0 + (ema ? 8.0 : 0.0) + (p_bf16 ? 2.0 : 0.0)) * n);
  AdamC c = make_adam(lr, beta1, beta2, eps, weight_decay, step, grad_scale, ema_decay);
  hipLaunchKernelGGL(k_adamw_f32, dim3(op_blocks(n / 4 + 1)), dim3(OP_THREADS), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16*)p_bf16, n, c);
  return st355_check_launch("adamw_ema_step");
}

extern "C" int st355_adamw_ema_step_bf16(void* stream, void* p, const void* g, float* m, float* v, void* ema, int64_t n, float lr,
                                         float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                                         float ema_decay) {
  ST_REQUIRE(p && g && m && v && n > 0 && step >= 1 && n % 8 == 0, "adamw_ema_step_bf16: bad args (n must be a multiple of 8)");
  ProfScope ps(stream, ST355_K_OPTIM, 12.0 * n, (22.0 + (ema ? 4.0 : 0.0)) * n);
  AdamC c = make_adam(lr, beta1, beta2, eps, weight_decay, step, grad_scale, ema_decay);
  hipLaunchKernelGGL(k_adamw_bf16, dim3(op_blocks(n / 8)), dim3(OP_THREADS), 0, (hipStream_t)stream, (bf16*)p, (const bf16*)g, m, v,
                     (bf16*)ema, n, c);
  return st355_check_launch("adamw_ema_step_bf16");
}

// s -= (1-d) (s - p)      (ema.py:423: torch._foreach_sub_(s, torch._foreach_sub(s, p), alpha=1-d))
template <typename T>
__global__ void __launch_bounds__(OP_THREADS) k_ema(T* __restrict__ s, const T* __restrict__ p, int64_t n, float omd) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float sf = (float)s[i], pf = (float)p[i];
    const T diff = (T)(sf - pf);                 // the reference materialises (s - p) in the parameter dtype
    s[i] = (T)(sf - omd * (float)diff);
  }
}
extern "C" int st355_ema_update(void* stream, void* shadow, const void* param, int64_t n, float decay, int elem_bytes) {
  ST_REQUIRE(shadow && param && n > 0 && (elem_bytes == 4 || elem_bytes == 2), "ema_update: bad args");
  ProfScope ps(stream, ST355_K_OPTIM, 3.0 * n, 3.0 * elem_bytes * n);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_ema<float>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (float*)shadow, (const float*)param, n, 1.f - decay);
  else
    hipLaunchKernelGGL(k_ema<bf16>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (bf16*)shadow, (const bf16*)param, n, 1.f - decay);
  return st355_check_launch("ema_update");
}

template <typename T>
__global__ void __launch_bounds__(OP_THREADS) k_grad_norm(const T* __restrict__ g, int64_t n, float* __restrict__ out2) {
  float ss = 0.f, mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float f = (float)g[i];
    ss += f * f;
    mx = fmaxf(mx, fabsf(f));
  }
  ss = wave_sum(ss);
  mx = wave_max(mx);
  __shared__ float rs[OP_THREADS / WAVE], rm[OP_THREADS / WAVE];
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = ss; rm[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < OP_THREADS / WAVE; i++) { a += rs[i]; b = fmaxf(b, rm[i]); }
    atomicAdd(&out2[0], a);
    atomicMax((unsigned int*)&out2[1], __float_as_uint(b));  // non-negative floats order like uints
  }
}
extern "C" int st355_grad_norm(void* stream, const void* g, int64_t n, int elem_bytes, float* out2) {
  ST_REQUIRE(g && out2 && n > 0 && (elem_bytes == 4 || elem_bytes == 2), "grad_norm: bad args");
  ProfScope ps(stream, ST355_K_OPTIM, 3.0 * n, (double)elem_bytes * n);
  hipMemsetAsync(out2, 0, 2 * sizeof(float), (hipStream_t)stream);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_grad_norm<float>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (const float*)g, n, out2);
  else
    hipLaunchKernelGGL(k_grad_norm<bf16>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (const bf16*)g, n, out2);
  return st355_check_launch("grad_norm");
}

// LoRA operand packer (block-structured; see st355.h)
__global__ void __launch_bounds__(OP_THREADS) k_lora_pack(const float* __restrict__ A, const float* __restrict__ Bm, int r, int K, int N,
                                                         float scale, bf16* __restrict__ A_cat, bf16* __restrict__ A_cat_T,
                                                         bf16* __restrict__ B_blk, bf16* __restrict__ B_blk_T, int K2, int k2_off,
                                                         int N_total, int n_off) {
  const int64_t nA = (int64_t)r * K, nB = (int64_t)r * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nA + nB; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nA) {
      const int j = (int)(i / K), k = (int)(i % K);
      const bf16 val = f2bf(A[i]);
      A_cat[(int64_t)(k2_off + j) * K + k] = val;
      A_cat_T[(int64_t)k * K2 + k2_off + j] = val;
    } else {
      const int64_t t = i - nA;
      const int j = (int)(t / N), n = (int)(t % N);
      const bf16 val = f2bf(scale * Bm[(int64_t)n * r + j]);
      B_blk_T[(int64_t)(k2_off + j) * N_total + n_off + n] = val;
      B_blk[(int64_t)(n_off + n) * K2 + k2_off + j] = val;
    }
  }
}
extern "C" int st355_lora_pack(void* stream, const float* A, const float* Bm, int r, int K, int N, float scale, void* A_cat, void* A_cat_T,
                               void* B_blk, void* B_blk_T, int K2, int k2_off, int N_total, int n_off) {
  ST_REQUIRE(A && Bm && A_cat && A_cat_T && B_blk && B_blk_T && r > 0 && K > 0 && N > 0, "lora_pack: bad args");
  ST_REQUIRE(K2 % 64 == 0 && k2_off >= 0 && k2_off + r <= K2 && n_off >= 0 && n_off + N <= N_total, "lora_pack: block out of range");
  ProfScope ps(stream, ST355_K_OPTIM, 0, 8.0 * r * (K + N));
  hipLaunchKernelGGL(k_lora_pack, dim3(op_blocks((int64_t)r * (K + N))), dim3(OP_THREADS), 0, (hipStream_t)stream, A, Bm, r, K, N, scale,
                     (bf16*)A_cat, (bf16*)A_cat_T, (bf16*)B_blk, (bf16*)B_blk_T, K2, k2_off, N_total, n_off);
  return st355_check_launch("lora_pack");
}
