// fp8.hip — K19: the fp8-native Linear of the reference (helpers/training/quantisation/fp8_native.py:25-119):
//   weights  : OCP e4m3fn, one fp32 scale per output row   (quantize_weight_to_fp8, :25-30)
//   inputs   : OCP e5m2, ONE scale per call = 57344 / amax(x), held as a bf16 scalar like the reference's tensor arithmetic (:58-60)
//   product  : out = (x_q W_q^T) * (1/input_scale) * w_scale[n] + bias   -> bf16   (torch._scaled_mm row-wise scaling, :67-75)
// The quantisers are HBM-streaming passes; the contraction is k_gemm_pq<EPI, false, /*F8=*/true> in gemm.hip (v_mfma_f32_32x32x16_fp8_bf8).
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


// fp32 -> e4m3fn / e5m2 bytes (gfx950 = OCP formats; v_cvt_pk_* round to nearest even and saturate to the finite maximum)
__device__ __forceinline__ uint8_t to_e4m3(float v) { return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0u, false) & 0xFFu); }
// e5m2: the hardware v_cvt_pk_bf8_f32 by default; -DST355_FP8_SOFT_CVT keeps an integer round-to-nearest-even restatement (normal range:
// re-bias the exponent, add the half-ulp bias + the parity bit, shift; below 2^-14: the magic-number float add that rounds into the
// subnormal grid; |v| <= 57344 here) that the parity test can be built against to cross-check the instruction.
#ifndef ST355_FP8_SOFT_CVT
__device__ __forceinline__ uint8_t to_e5m2(float v) { return (uint8_t)(__builtin_amdgcn_cvt_pk_bf8_f32(v, 0.f, 0u, false) & 0xFFu); }
#else
__device__ __forceinline__ uint8_t to_e5m2(float v) {
  uint32_t f = __float_as_uint(v);
  const uint32_t sign = f & 0x80000000u;
  f ^= sign;
  uint32_t r;
  if (f < (113u << 23)) {                                    // |v| < 2^-14: e5m2 subnormal grid (step 2^-16)
    const uint32_t magic = 134u << 23;                       // 2^(134-127) = 128.0f: adding it aligns the fp32 mantissa to that grid
    r = __float_as_uint(__uint_as_float(f) + __uint_as_float(magic)) - magic;
  } else {
    const uint32_t odd = (f >> 21) & 1u;
    f += ((uint32_t)(15 - 127) << 23) + 0xFFFFFu;
    f += odd;
    r = f >> 21;
  }
  return (uint8_t)(r | (sign >> 24));
}
#endif

// one wave per weight row: amax -> scale = max(amax, 1e-12) / 448 -> q = clamp(w / scale, +-448)
__global__ void __launch_bounds__(256) k_fp8_quant_weight(const bf16* __restrict__ w, int64_t ldw, uint8_t* __restrict__ q, float* __restrict__ scale,
                                                         int N, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const bf16* wr = w + (int64_t)row * ldw;
  float amax = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const bf16x8 v = *(const bf16x8*)(wr + k);
#pragma unroll
    for (int j = 0; j < 8; j++) amax = fmaxf(amax, fabsf(bf2f(v[j])));
  }
  amax = wave_max(amax);
  const float sc = fmaxf(amax, 1e-12f) / 448.0f;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + (int64_t)row * K;
  for (int k = lane * 8; k < K; k += 512) {
    const bf16x8 v = *(const bf16x8*)(wr + k);
    u32x2 o;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      lo |= (uint32_t)to_e4m3(fminf(fmaxf(bf2f(v[j]) / sc, -448.f), 448.f)) << (8 * j);
      hi |= (uint32_t)to_e4m3(fminf(fmaxf(bf2f(v[4 + j]) / sc, -448.f), 448.f)) << (8 * j);
    }
    o[0] = lo; o[1] = hi;
    *(u32x2*)(qr + k) = o;
  }
}

// |x| maximum of the whole tensor: non-negative floats order like their bit patterns, so an integer atomicMax is exact and deterministic
__global__ void __launch_bounds__(256) k_absmax(const bf16* __restrict__ x, int64_t ldx, int64_t M, int K, uint32_t* __restrict__ amax_bits) {
  float m = 0.f;
  const int64_t nv = M * (K / 8);
  if (ldx == K) {                                            // dense rows (every caller today): one linear 16-byte stream, no per-element index division
    const bf16x8* xv = (const bf16x8*)x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
      const bf16x8 v = xv[i];
#pragma unroll
      for (int j = 0; j < 8; j++) m = fmaxf(m, fabsf(bf2f(v[j])));
    }
  } else {
    const uint32_t kv = (uint32_t)(K / 8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = i / kv;
      const int c = (int)(i - r * kv) * 8;
      const bf16x8 v = *(const bf16x8*)(x + r * ldx + c);
#pragma unroll
      for (int j = 0; j < 8; j++) m = fmaxf(m, fabsf(bf2f(v[j])));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));
}

// input_scale = min( bf16( bf16(1 / max(amax, 1e-12)) * 57344 ), 57344 ) ; x_q = e5m2( clamp( bf16(x * input_scale), +-57344 ) ) ; scale_a = float(bf16(1 / input_scale))
__global__ void __launch_bounds__(256) k_fp8_quant_act(const bf16* __restrict__ x, int64_t ldx, int64_t M, int K, const uint32_t* __restrict__ amax_bits,
                                                      uint8_t* __restrict__ q, float* __restrict__ scale_a) {
  const float amax = bf2f(f2bf(__uint_as_float(*amax_bits)));                 // the reference's amax is a bf16 tensor
  const float amin = bf2f(f2bf(1e-12f));
  // `input_max / tensor` is Tensor.__rtruediv__ = tensor.reciprocal() * input_max: two bf16 roundings (e.g. amax 7.5 -> 7680, not 7648)
  const float rcp = bf2f(f2bf(1.0f / fmaxf(amax, amin)));
  const float is = fminf(bf2f(f2bf(rcp * 57344.0f)), 57344.0f);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_a[0] = bf2f(f2bf(1.0f / is));
  const int64_t nv = M * (K / 8);
  auto q8 = [&](const bf16x8& v) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      lo |= (uint32_t)to_e5m2(fminf(fmaxf(bf2f(f2bf(bf2f(v[j]) * is)), -57344.f), 57344.f)) << (8 * j);
      hi |= (uint32_t)to_e5m2(fminf(fmaxf(bf2f(f2bf(bf2f(v[4 + j]) * is)), -57344.f), 57344.f)) << (8 * j);
    }
    u32x2 o;
    o[0] = lo; o[1] = hi;
    return o;
  };
  if (ldx == K) {                                            // dense rows: a linear stream, 16 elements (two 16-byte loads, one 16-byte store) per thread and step
    const bf16x8* xv = (const bf16x8*)x;
    const int64_t np = nv / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (int64_t)gridDim.x * blockDim.x) {
      const u32x2 a = q8(xv[2 * i]), b = q8(xv[2 * i + 1]);
      u32x4 o;
      o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
      *(u32x4*)(q + 16 * i) = o;
    }
    if ((nv & 1) && blockIdx.x == 0 && threadIdx.x == 0) *(u32x2*)(q + 8 * (nv - 1)) = q8(xv[nv - 1]);
    return;
  }
  const uint32_t kv = (uint32_t)(K / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / kv;
    const int c = (int)(i - r * kv) * 8;
    *(u32x2*)(q + r * K + c) = q8(*(const bf16x8*)(x + r * ldx + c));
  }
}

extern "C" int st355_fp8_quantize_weight(void* stream, const void* w, int64_t ldw, void* q, float* scale, int N, int K) {
  ST_REQUIRE(w && q && scale && N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0, "fp8_quantize_weight: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 4.0 * N * K, 5.0 * N * K);
  hipLaunchKernelGGL(k_fp8_quant_weight, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)w, ldw, (uint8_t*)q, scale, N, K);
  return st355_check_launch("fp8_quantize_weight");
}

extern "C" int st355_fp8_quantize_act(void* stream, const void* x, int64_t ldx, void* q, float* scale_a, int64_t M, int K, void* workspace) {
  ST_REQUIRE(x && q && scale_a && workspace && M > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0, "fp8_quantize_act: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 4.0 * M * K, 5.0 * M * K);
  zero_words(stream, workspace, 1);
  int64_t blocks = cdiv64(M * (K / 8), 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_absmax, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, M, K, (uint32_t*)workspace);
  hipLaunchKernelGGL(k_fp8_quant_act, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, M, K,
                     (const uint32_t*)workspace, (uint8_t*)q, scale_a);
  return st355_check_launch("fp8_quantize_act");
}
