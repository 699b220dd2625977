// common.h — shared device helpers + host-side launch bookkeeping for libst355 (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/st355.h"

// ------------------------------------------------------------------------------------------------
// device types
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // RNE (v_cvt_pk_bf16_f32 on gfx950)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// tanh-approximated GELU (diffusers "gelu-approximate" / nn.GELU(approximate="tanh")):
//   0.5 x (1 + tanh(u)) = x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3)
// written as ONE v_exp_f32 + ONE v_rcp_f32 + 5 plain VALU ops.  The IEEE division of the textbook form (v_div_scale x2, v_rcp, 4 fma,
// v_div_fmas, v_div_fixup) made the GELU epilogues of the N = 12288 MLP GEMMs ~25 VALU instructions per element — 128 elements per lane per
// 256x256 tile, all of it with the matrix pipe idle.  v_rcp_f32 is accurate to 1 ulp; the result is rounded to bf16 right after.
__device__ __forceinline__ float sigmoid2u(float x, float x2) {
  // 2u * log2(e) = x * (c0 + c1 x^2):  c0 = 2 sqrt(2/pi) log2(e), c1 = c0 * 0.044715
  const float c0 = 2.302208198f, c1 = 0.1029432397f;
  const float e = __builtin_amdgcn_exp2f(-x * (c0 + c1 * x2));        // exp(-2u); inf for very negative x -> rcp(inf) = 0
  return __builtin_amdgcn_rcpf(1.f + e);
}
// exact (erf) GELU of the UNet's GEGLU feed-forward (F.gelu default) and its derivative.  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// rounding of every consumer) on ONE v_rcp_f32 + ONE v_exp_f32 + 8 plain VALU operations: libm's erff is ~60 instructions, which made the GEGLU passes — and, with
// the matrix pipe idle behind them, the GEGLU GEMM epilogues — VALU-bound (r6: the fused ff.net.0 epilogue cost 216 us per launch with erff).  e = exp(-g^2 / 2) is
// shared between erf(g / sqrt 2) and the Gaussian term of the derivative.
__device__ __forceinline__ void gelu_erf_parts(float g, float& phi, float& gauss) {      // phi = Phi(g) = 0.5 (1 + erf(g / sqrt 2));  gauss = exp(-g^2 / 2)
  const float ax = fabsf(g) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  gauss = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float half_erfc = 0.5f * poly * gauss;                   // 0.5 erfc(|x|)
  phi = g >= 0.f ? 1.f - half_erfc : half_erfc;
}
__device__ __forceinline__ float gelu_erf(float g) { float phi, e; gelu_erf_parts(g, phi, e); return g * phi; }
__device__ __forceinline__ float gelu_erf_grad(float g) { float phi, e; gelu_erf_parts(g, phi, e); return fmaf(g * 0.39894228040143268f, e, phi); }
#ifdef ST355_GELU_TEXTBOOK      // lab / A-B builds only: the IEEE-division form this replaced
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  float t = 1.f - 2.f / (1.f + __expf(2.f * u));
  return 0.5f * x * (1.f + t);
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float x2 = x * x;
  float u = k0 * (x + k1 * x * x2);
  float t = 1.f - 2.f / (1.f + __expf(2.f * u));
  float du = k0 * (1.f + 3.f * k1 * x2);
  return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
}
#else
__device__ __forceinline__ float gelu_tanh(float x) { return x * sigmoid2u(x, x * x); }
// d/dx [x s(x)] = s + x s (1 - s) * d(2u)/dx,   d(2u)/dx = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2)
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float k0 = 1.5957691216057308f, k1 = 0.134145f;
  const float x2 = x * x;
  const float s = sigmoid2u(x, x2);
  return s + x * s * (1.f - s) * (k0 + k0 * k1 * x2);
}
#endif

// ---- Philox4x32-10 ------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t seed, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; i++) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = c[i];
}
// gfx950 transposing LDS read (ds_read_b64_tr_b16): within each 16-lane group, lane 4i+s supplies the address of 4 contiguous bf16 of
// "row" i (i = 0..3) and lane j receives {row0[j], row1[j], row2[j], row3[j]} — column j of the 4x16 block (tools/probes/tr_probe.hip).
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ s16x4 lds_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 lds_tr16x2(const void* p0, const void* p1) {   // two transposed reads -> one MFMA k-fragment
  bf16x8 f;
  *(s16x4*)&f = lds_tr16(p0);
  *((s16x4*)&f + 1) = lds_tr16(p1);
  return f;
}

// XCD-aware, bijective block-id remap (8 XCDs, block b runs on XCD b%8): gives each XCD a
// contiguous chunk of the logical grid so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------
// host side: error text + launch profiler
// ------------------------------------------------------------------------------------------------
void st355_set_error(const char* fmt, ...);
int st355_check_launch(const char* what);

struct ProfScope {
  int idx;
  void* stream;
  ProfScope(void* stream, int klass, double flops, double bytes, const char* tag_fmt = nullptr, ...);
  ~ProfScope();
};

#define ST_REQUIRE(cond, ...)                \
  do {                                       \
    if (!(cond)) {                           \
      st355_set_error(__VA_ARGS__);          \
      return ST355_EINVAL;                   \
    }                                        \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE function attribute: one flag per (launch site, device), so a process that launches on a second GPU
// sets it there too (a single process-wide flag would skip it and the launch would fail)
// (devices >= 32 are not cached: the attribute is simply set at every launch there; the flags are relaxed atomics — two host threads racing on a first launch
// both set the attribute, which is idempotent)
struct St355AttrOnce {
  std::atomic<bool> done[32] = {};
  bool need() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return true;
    return !done[d].exchange(true, std::memory_order_relaxed);
  }
};
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
