// optim.hip — K15/K16/K17: one-pass fused AdamW (+EMA) over a flat parameter arena, standalone EMA,
// gradient norm, and the LoRA operand packer.  All HBM-streaming: 16-byte accesses, one read and one write
// of every state word per step (22-28 B/param depending on layout).
#include <math.h>
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
      for (int j = 0; j < 8; j++) { const float e = bf2f(eb[j]); const float diff = bf2f(f2bf(e - bf2f(pb[j]))); eb[j] = f2bf(e - c.ema_omd * diff); }   // as k_ema<bf16>: (s - p) materialised in bf16 (ema.py:393-433)
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

// ---- K16b: AdamWBF16 — the reference examples' default optimizer ------------------------------------------------------------
// optimizers/adamw_bfloat16/__init__.py:55-180 (+ stochastic/__init__.py:47-124): parameter, both moments and a compensation buffer
// ("shift": what should have been added to p but was lost to bf16 truncation) are ALL bf16; the first moment, the shift and the
// parameter are updated with stochastic rounding (a random 16-bit integer added to the fp32 bit pattern before truncation); weight
// decay is owed per tensor and applied to `shift` only once it exceeds 5e-3.  The reference runs ~25 small kernels per parameter
// tensor with fp32 temporaries; here it is ONE pass over a flat arena: 5 bf16 reads + 4 bf16 writes = 18 B/param.
// Every operation below keeps the reference's order and rounding points (no FMA contraction: the fp32 temporaries of the reference
// are separate multiply / add / divide results), so that with the same random draws the states are bit-identical.
struct Bf16OptC {
  float beta1, one_m_beta1, beta2, one_m_beta2, eps, value /* -lr * sqrt(1 - beta2^step) */, grad_scale;
};

__device__ __forceinline__ bf16 sr_bf16(float x, uint32_t r16) {           // copy_stochastic_
  const uint32_t bits = (__float_as_uint(x) + r16) & 0xFFFF0000u;
  return __builtin_bit_cast(bf16, (uint16_t)(bits >> 16));
}

#pragma clang fp contract(off)
__device__ __forceinline__ void adamw_bf16_one(bf16& p, bf16 g_in, bf16& m, bf16& v, bf16& sh, const Bf16OptC& c, float decay,
                                               const uint32_t (&r)[4]) {
  float g = bf2f(g_in);
  if (c.grad_scale != 1.f) g = bf2f(f2bf(g * c.grad_scale));              // a scaled gradient is a bf16 tensor in the reference too
  const bf16 m1 = f2bf(bf2f(m) * c.beta1);                                // exp_avg.mul_(beta1)
  const bf16 m_new = sr_bf16(g + c.one_m_beta1 * bf2f(m1), r[0]);         // add_stochastic_(exp_avg, grad, alpha): other + alpha*input (sic)
  const bf16 v1 = f2bf(bf2f(v) * c.beta2);                                // exp_avg_sq.mul_(beta2)
  const bf16 v_new = f2bf(bf2f(v1) + (c.one_m_beta2 * g) * g);            // .addcmul_(grad, grad, value=1-beta2)
  const bf16 denom = f2bf(bf2f(f2bf(sqrtf(bf2f(v_new)))) + c.eps);        // exp_avg_sq.sqrt().add_(eps)
  const bf16 sh1 = sr_bf16(bf2f(sh) + (c.value * bf2f(m_new)) / bf2f(denom), r[1]);   // addcdiv_stochastic_(shift, exp_avg, denom, value)
  const bf16 p_new = sr_bf16(bf2f(sh1) + bf2f(p), r[2]);                  // add_stochastic_(p, shift)
  const bf16 err = f2bf(bf2f(p) - bf2f(p_new));                           // buffer.sub_(p)
  bf16 sh2 = sr_bf16(bf2f(err) + bf2f(sh1), r[3]);                        // add_stochastic_(shift, buffer - p)
  if (decay > 0.f) sh2 = f2bf(bf2f(sh2) + (-decay) * bf2f(p_new));        // shift.add_(p, alpha=-decay): opmath (fp32) alpha, as ATen's GPU kernels
  p = p_new; m = m_new; v = v_new; sh = sh2;
}

// seg_end[nseg]: exclusive end offsets of the parameter tensors inside the arena; seg_decay[nseg]: this step's decay per tensor
__device__ __forceinline__ int seg_of(const int64_t* seg_end, int nseg, int64_t i) {
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (i < seg_end[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__global__ void __launch_bounds__(OP_THREADS) k_adamw_bf16_sr(bf16* __restrict__ p, const bf16* __restrict__ g, bf16* __restrict__ m,
                                                             bf16* __restrict__ v, bf16* __restrict__ sh, int64_t n, Bf16OptC c,
                                                             const int64_t* __restrict__ seg_end, const float* __restrict__ seg_decay,
                                                             int nseg, const int32_t* __restrict__ rand_bits, uint64_t seed,
                                                             uint64_t offset) {
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = i * 8;
    bf16x8 pb = *(bf16x8*)(p + e0), gb = *(const bf16x8*)(g + e0), mb = *(bf16x8*)(m + e0), vb = *(bf16x8*)(v + e0), sb = *(bf16x8*)(sh + e0);
    int s0 = 0, s1 = 0;
    if (nseg > 1) { s0 = seg_of(seg_end, nseg, e0); s1 = (e0 + 7 < seg_end[s0]) ? s0 : seg_of(seg_end, nseg, e0 + 7); }
    const float d0 = nseg > 0 ? seg_decay[s0] : 0.f;
    uint32_t rr[4][8];
    if (rand_bits) {                                     // injected draws (parity tests): rand_bits[k*n + e], values in [0, 65536)
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) rr[k][j] = (uint32_t)rand_bits[(int64_t)k * n + e0 + j];
    } else {                                             // 32 sixteen-bit draws from 4 Philox4x32-10 blocks (counter = group index)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t o[4];
        philox4x32_10(4 * (uint64_t)i + k + offset, seed, o);
#pragma unroll
        for (int q = 0; q < 4; q++) { rr[k][2 * q] = o[q] & 0xFFFFu; rr[k][2 * q + 1] = o[q] >> 16; }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float d = d0;
      if (s1 != s0) d = seg_decay[seg_of(seg_end, nseg, e0 + j)];
      const uint32_t r4[4] = {rr[0][j], rr[1][j], rr[2][j], rr[3][j]};
      bf16 pj = pb[j], mj = mb[j], vj = vb[j], sj = sb[j];
      adamw_bf16_one(pj, gb[j], mj, vj, sj, c, d, r4);
      pb[j] = pj; mb[j] = mj; vb[j] = vj; sb[j] = sj;
    }
    *(bf16x8*)(p + e0) = pb; *(bf16x8*)(m + e0) = mb; *(bf16x8*)(v + e0) = vb; *(bf16x8*)(sh + e0) = sb;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {      // tail elements
    const int64_t e = (nv << 3) + threadIdx.x;
    const float d = nseg > 0 ? seg_decay[seg_of(seg_end, nseg, e)] : 0.f;
    uint32_t r4[4];
    if (rand_bits) {
      for (int k = 0; k < 4; k++) r4[k] = (uint32_t)rand_bits[(int64_t)k * n + e];
    } else {
      uint32_t o[4];
      philox4x32_10(4 * (uint64_t)nv + 4 * (uint64_t)threadIdx.x + offset, seed ^ 0x9E3779B97F4A7C15ull, o);
      for (int k = 0; k < 4; k++) r4[k] = o[k] & 0xFFFFu;
    }
    bf16 pj = p[e], mj = m[e], vj = v[e], sj = sh[e];
    adamw_bf16_one(pj, g[e], mj, vj, sj, c, d, r4);
    p[e] = pj; m[e] = mj; v[e] = vj; sh[e] = sj;
  }
}
#pragma clang fp contract(fast)

extern "C" int st355_adamw_bf16_sr_step(void* stream, void* p, const void* g, void* m, void* v, void* shift, int64_t n, int64_t step,
                                        double lr, double beta1, double beta2, double eps, const int64_t* seg_end, const float* seg_decay,
                                        int nseg, const int32_t* rand_bits, uint64_t seed, uint64_t offset, float grad_scale) {
  ST_REQUIRE(p && g && m && v && shift && n > 0 && step >= 1, "adamw_bf16_sr_step: bad args");
  ST_REQUIRE(nseg == 0 || (seg_end && seg_decay), "adamw_bf16_sr_step: segment tables missing");
  ST_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)shift % 16 == 0),
             "adamw_bf16_sr_step: arena must be 16-byte aligned");
  ProfScope ps(stream, ST355_K_OPTIM, 20.0 * n, 18.0 * n);
  Bf16OptC c;          // the double -> float conversions happen where ATen converts its Python scalars
  c.beta1 = (float)beta1; c.one_m_beta1 = (float)(1.0 - beta1); c.beta2 = (float)beta2; c.one_m_beta2 = (float)(1.0 - beta2);
  c.eps = (float)eps; c.value = (float)(-lr * sqrt(1.0 - pow(beta2, (double)step))); c.grad_scale = grad_scale;
  hipLaunchKernelGGL(k_adamw_bf16_sr, dim3(op_blocks(n / 8 + 1)), dim3(OP_THREADS), 0, (hipStream_t)stream, (bf16*)p, (const bf16*)g,
                     (bf16*)m, (bf16*)v, (bf16*)shift, n, c, seg_end, seg_decay, nseg, rand_bits, seed, offset);
  return st355_check_launch("adamw_bf16_sr_step");
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

// Fixed-order two-stage reduction (r03; r02 added the block partials with atomicAdd, so the clip coefficient's last bits depended on block arrival order):
// GN_BLOCKS workgroups write one (sum of squares, max |g|) pair each into a library-owned scratch, a single wave then combines them in index order.
// The scratch is one buffer per process: st355_grad_norm calls are stream-ordered on the training stream (one training thread per rank).
#define GN_BLOCKS 1024
__device__ float g_gn_part[2 * GN_BLOCKS];          // st355_grad_norm's library-owned scratch; st355_grad_norm_ws takes the caller's (2 * GN_BLOCKS floats)
template <typename T>
__global__ void __launch_bounds__(OP_THREADS) k_grad_norm(const T* __restrict__ g, int64_t n, float* __restrict__ part) {
  if (part == nullptr) part = g_gn_part;
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
    part[blockIdx.x] = a;
    part[GN_BLOCKS + blockIdx.x] = b;
  }
}
__global__ void __launch_bounds__(WAVE) k_grad_norm_final(int nblocks, float* __restrict__ out2, const float* __restrict__ part) {
  if (part == nullptr) part = g_gn_part;
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += WAVE) { a += part[i]; b = fmaxf(b, part[GN_BLOCKS + i]); }     // lane l: blocks l, l + 64, ... in order
  a = wave_sum(a);                                                                                                 // fixed lane tree
  b = wave_max(b);
  if (threadIdx.x == 0) { out2[0] = a; out2[1] = b; }
}
// workspace: NULL = the library-owned scratch (ONE per process: calls must then be ordered on one stream), else 2 * 1024 floats of the caller's — one per stream
// that computes norms concurrently (a side-stream EMA / ControlNet norm, capture overlapping eager calls)
extern "C" int st355_grad_norm_ws(void* stream, const void* g, int64_t n, int elem_bytes, float* out2, float* workspace) {
  ST_REQUIRE(g && out2 && n > 0 && (elem_bytes == 4 || elem_bytes == 2), "grad_norm: bad args");
  ProfScope ps(stream, ST355_K_OPTIM, 3.0 * n, (double)elem_bytes * n);
  int64_t nb = cdiv64(n, OP_THREADS);
  if (nb > GN_BLOCKS) nb = GN_BLOCKS;
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_grad_norm<float>, dim3((unsigned)nb), dim3(OP_THREADS), 0, (hipStream_t)stream, (const float*)g, n, workspace);
  else
    hipLaunchKernelGGL(k_grad_norm<bf16>, dim3((unsigned)nb), dim3(OP_THREADS), 0, (hipStream_t)stream, (const bf16*)g, n, workspace);
  hipLaunchKernelGGL(k_grad_norm_final, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, (int)nb, out2, (const float*)workspace);
  return st355_check_launch("grad_norm");
}
extern "C" int st355_grad_norm(void* stream, const void* g, int64_t n, int elem_bytes, float* out2) { return st355_grad_norm_ws(stream, g, n, elem_bytes, out2, nullptr); }

// clip_grad_value_ (trainer.py:7209-7213): g <- clamp(g, -c, +c) in place over the flat gradient arena
template <typename T>
__global__ void __launch_bounds__(OP_THREADS) k_grad_clamp(T* __restrict__ g, int64_t n, float c) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = (float)g[i];
    g[i] = (T)fminf(fmaxf(v, -c), c);
  }
}
extern "C" int st355_grad_clamp(void* stream, void* g, int64_t n, int elem_bytes, float c) {
  ST_REQUIRE(g && n > 0 && c > 0.f && (elem_bytes == 4 || elem_bytes == 2), "grad_clamp: bad args");
  ProfScope ps(stream, ST355_K_OPTIM, 2.0 * n, 2.0 * elem_bytes * n);
  if (elem_bytes == 4) hipLaunchKernelGGL(k_grad_clamp<float>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (float*)g, n, c);
  else hipLaunchKernelGGL(k_grad_clamp<bf16>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (bf16*)g, n, c);
  return st355_check_launch("grad_clamp");
}

// accelerator.clip_grad_norm_ (trainer.py:7201-7208) without the host round trip: the coefficient min(1, max_norm / (||g|| + 1e-6)) is computed
// on the device from the statistics st355_grad_norm left in HBM, and applied in place (torch's clip_grad_norm_ does the same g.mul_(coef) pass).
// pre_scale folds the 1/world averaging that lives in the optimizer's grad_scale: the norm that is clipped is the norm of the AVERAGED gradient.
template <typename T>
__global__ void __launch_bounds__(OP_THREADS) k_grad_clip_norm(T* __restrict__ g, int64_t n, const float* __restrict__ stats2, float max_norm,
                                                              float pre_scale) {
  const float norm = sqrtf(stats2[0]) * pre_scale;
  const float coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  if (coef >= 1.f) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] = (T)((float)g[i] * coef);
}
extern "C" int st355_grad_clip_norm(void* stream, void* g, int64_t n, int elem_bytes, const float* stats2, float max_norm, float pre_scale) {
  ST_REQUIRE(g && stats2 && n > 0 && max_norm > 0.f && pre_scale > 0.f && (elem_bytes == 4 || elem_bytes == 2), "grad_clip_norm: bad args");
  ProfScope ps(stream, ST355_K_OPTIM, 1.0 * n, 2.0 * elem_bytes * n);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_grad_clip_norm<float>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (float*)g, n, stats2, max_norm, pre_scale);
  else
    hipLaunchKernelGGL(k_grad_clip_norm<bf16>, dim3(op_blocks(n)), dim3(OP_THREADS), 0, (hipStream_t)stream, (bf16*)g, n, stats2, max_norm, pre_scale);
  return st355_check_launch("grad_clip_norm");
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
