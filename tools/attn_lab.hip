// attn_lab — standalone attention kernel lab for libst355 (no torch: starts in milliseconds on a fresh GPU box).
//   tools/attn_lab [B H S d]      default 2 24 4608 128 (Flux.1 1024^2 at per-GPU batch 2)
// Times st355_attn_fwd (ST355_ATTN_FWD=1: the r01 kernel, otherwise the default r02 kernel; the lab re-executes itself once per generation, and the
// second child checks its output against the r01 kernel launched directly) and st355_attn_bwd with and
// without the pre-transposed Q^T / K^T copies (dkv2 + dq vs dkv3 + dq<TR>), per kernel class through the library's own hipEvent profiler, and checks
// that the two backward paths agree bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast tools/attn_lab.hip -o tools/attn_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>
#include "../simpletuner_amd/csrc/runtime.hip"
#include "../simpletuner_amd/csrc/attention.hip"
#include "../simpletuner_amd/csrc/attention_bwd.hip"
#include "attn_fwd_variants.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define RC(x) do { int r_ = (x); if (r_) { printf("st355 rc=%d: %s (%s:%d)\n", r_, st355_last_error(), __FILE__, __LINE__); exit(2); } } while (0)

// channels >= dvalid of every head are zero (LAB_DVALID: a head_dim zero-padded to the kernels' width — PixArt-Sigma's 72, SD 1.5's 80 inside 96; every buffer here
// has the head channel as i % d)
__global__ void k_fill(bf16* p, int64_t n, uint32_t seed, float scale, int d = 1, int dvalid = 1) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = (int)(i % d) < dvalid ? (bf16)(((h >> 8) * (1.f / 8388608.f) - 1.f) * scale) : (bf16)0.f;
  }
}
// Xt[b,h,c,s] = X[b,h,s,c] (zero padded to Sp)
__global__ void k_transpose_heads(const bf16* X, bf16* Xt, int64_t BH, int S, int Sp, int d) {
  const int64_t n = BH * (int64_t)d * Sp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(i % Sp); const int c = (int)((i / Sp) % d); const int64_t bh = i / ((int64_t)Sp * d);
    Xt[i] = s < S ? X[(bh * S + s) * d + c] : (bf16)0.f;
  }
}
// Vt[b,h,c,s] = vrows[(b*S+s)*ld + h*d + c]
__global__ void k_vt(const bf16* vrows, int64_t ld, bf16* Vt, int B, int H, int S, int Sp, int d) {
  const int64_t n = (int64_t)B * H * d * Sp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(i % Sp); const int c = (int)((i / Sp) % d); const int h = (int)((i / ((int64_t)Sp * d)) % H); const int b = (int)(i / ((int64_t)Sp * d * H));
    Vt[i] = s < S ? vrows[((int64_t)b * S + s) * ld + (int64_t)h * d + c] : (bf16)0.f;
  }
}
__global__ void k_diff(const bf16* a, const bf16* b, int64_t n, int64_t ld, int64_t cols, unsigned long long* bad, float* maxd) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = ld ? (i / cols) * ld + (i % cols) : i;
    const float x = (float)a[idx], y = (float)b[idx];
    if (!(x == y)) { atomicAdd(bad, 1ull); atomicMax((int*)maxd, __float_as_int(fabsf(x - y))); }
  }
}

static void prof_print(const char* what) {
  double ms[16]; int64_t la[16]; double fl[16], by[16];
  st355_prof_collect(ms, la, fl, by, 16);
  const char* names[] = {"gemm", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_prep"};
  for (int k = 1; k <= 4; k++)
    if (la[k]) printf("  %-28s %-13s %8.3f ms/launch  %7.1f TFLOP/s (%lld launches)\n", what, names[k], ms[k] / la[k], fl[k] / ms[k] / 1e9, (long long)la[k]);
  st355_prof_reset();
}

int main(int argc, char** argv) {
  const int B = argc > 4 ? atoi(argv[1]) : 2, H = argc > 4 ? atoi(argv[2]) : 24, S = argc > 4 ? atoi(argv[3]) : 4608, d = argc > 4 ? atoi(argv[4]) : 128;
  const int Sp = (S + 63) / 64 * 64;
  const int64_t D = (int64_t)H * d, BH = (int64_t)B * H;
  const char* gen = "4";
  printf("== attn_lab B%d H%d S%d d%d\n", B, H, S, d);
  hipStream_t st; CK(hipStreamCreate(&st));
  bf16 *Q, *K, *Qt, *Kt, *Vt, *qkv, *O, *dO, *dQ, *dK, *dqkv, *dQ2, *dK2, *dqkv2; float* lse2; void* ws;
  const size_t nh = (size_t)BH * S * d, nt = (size_t)BH * d * Sp, nr = (size_t)B * S * D;
  CK(hipMalloc(&Q, nh * 2)); CK(hipMalloc(&K, nh * 2)); CK(hipMalloc(&Qt, nt * 2)); CK(hipMalloc(&Kt, nt * 2)); CK(hipMalloc(&Vt, nt * 2));
  CK(hipMalloc(&qkv, nr * 3 * 2)); CK(hipMalloc(&O, nr * 2)); CK(hipMalloc(&dO, nr * 2));
  CK(hipMalloc(&dQ, nh * 2)); CK(hipMalloc(&dK, nh * 2)); CK(hipMalloc(&dqkv, nr * 3 * 2)); CK(hipMalloc(&dQ2, nh * 2)); CK(hipMalloc(&dK2, nh * 2)); CK(hipMalloc(&dqkv2, nr * 3 * 2));
  CK(hipMalloc(&lse2, (size_t)BH * S * 4)); CK(hipMalloc(&ws, st355_attn_bwd_workspace(B, H, S, Sp, d)));
  const float qs = getenv("LAB_QSCALE") ? (float)atof(getenv("LAB_QSCALE")) : 1.5f;      // 6: the tile maxima climb past the stale reference by more than 2^8 (the out-of-line rescale runs)
  const int dvalid = getenv("LAB_DVALID") ? atoi(getenv("LAB_DVALID")) : d;
  k_fill<<<2048, 256, 0, st>>>(Q, nh, 1u, qs, d, dvalid); k_fill<<<2048, 256, 0, st>>>(K, nh, 2u, 1.5f, d, dvalid);
  k_fill<<<2048, 256, 0, st>>>(qkv, nr * 3, 3u, 1.f, d, dvalid); k_fill<<<2048, 256, 0, st>>>(dO, nr, 4u, 1.f, d, dvalid);
  k_transpose_heads<<<4096, 256, 0, st>>>(Q, Qt, BH, S, Sp, d); k_transpose_heads<<<4096, 256, 0, st>>>(K, Kt, BH, S, Sp, d);
  bf16* vrows = qkv + 2 * D;
  k_vt<<<4096, 256, 0, st>>>(vrows, 3 * D, Vt, B, H, S, Sp, d);
  CK(hipMemsetAsync(dqkv, 0, nr * 3 * 2, st)); CK(hipMemsetAsync(dqkv2, 0, nr * 3 * 2, st));
  const float scale = 1.f / sqrtf((float)d);
  const int iters = getenv("LAB_ITERS") ? atoi(getenv("LAB_ITERS")) : 10;
  // ---- forward ----
  g_attn_fwd_impl = 32;                      // the 32-queries-per-wave kernels first (generation 1 or 4 by ST355_ATTN_FWD); fwd64 is timed and compared below
  if (getenv("LAB_FWD_TRACE")) { CK(hipMalloc(&g_attn_fwd_trace, 4 * 128)); CK(hipMemsetAsync(g_attn_fwd_trace, 0, 4 * 128, st)); }
  RC(st355_attn_fwd(st, Q, K, Vt, nullptr, O, D, lse2, B, H, S, Sp, d, scale));
  CK(hipStreamSynchronize(st));
  st355_prof_reset(); st355_prof_enable(1);
  for (int i = 0; i < iters; i++) RC(st355_attn_fwd(st, Q, K, Vt, nullptr, O, D, lse2, B, H, S, Sp, d, scale));
  CK(hipStreamSynchronize(st));
  st355_prof_enable(0); prof_print("forward");
  if (strcmp(gen, "1") != 0 && d == 128) {   // row-major V (transposing LDS reads, no V^T buffer): timed, and bit-compared with the V^T form above
    bf16* O2; float* lse3; CK(hipMalloc(&O2, nr * 2)); CK(hipMalloc(&lse3, (size_t)BH * S * 4));
    RC(st355_attn_fwd_vrows(st, Q, K, vrows, 3 * D, nullptr, O2, D, lse3, B, H, S, d, scale));
    CK(hipStreamSynchronize(st));
    st355_prof_reset(); st355_prof_enable(1);
    for (int i = 0; i < iters; i++) RC(st355_attn_fwd_vrows(st, Q, K, vrows, 3 * D, nullptr, O2, D, lse3, B, H, S, d, scale));
    CK(hipStreamSynchronize(st));
    st355_prof_enable(0); prof_print("forward, row-major V");
    unsigned long long* bad; float* maxd; CK(hipMalloc(&bad, 8)); CK(hipMalloc(&maxd, 4));
    CK(hipMemsetAsync(bad, 0, 8, st)); CK(hipMemsetAsync(maxd, 0, 4, st));
    k_diff<<<2048, 256, 0, st>>>(O, O2, (int64_t)nr, 0, 0, bad, maxd);
    unsigned long long hb; float hm;
    CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hm, maxd, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    printf("  O, row-major V vs V^T form: %llu of %lld elements differ (max |d| %.3e)  %s\n", hb, (long long)nr, hm, hb ? "MISMATCH" : "bit-identical");
  }
  if (strcmp(gen, "1") != 0 && d == 128) {   // other generations: compare O / lse2 with the generation-1 kernel launched directly
    bf16* O1; float* lse1; CK(hipMalloc(&O1, nr * 2)); CK(hipMalloc(&lse1, (size_t)BH * S * 4));
    const int lds = 2 * (KB * 256 + 128 * 128);
    CK(hipFuncSetAttribute((const void*)k_attn_fwd<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t g0, g1; CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1));
    for (int i = 0; i <= iters; i++) {
      if (i == 1) CK(hipEventRecord(g0, st));
      hipLaunchKernelGGL(k_attn_fwd<128>, dim3((S + QB - 1) / QB, H, B), dim3(ATT_THREADS), lds, st, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                         (const float*)nullptr, O1, D, lse1, H, S, S, Sp, scale * LOG2E);
    }
    CK(hipEventRecord(g1, st)); CK(hipStreamSynchronize(st));
    { float gms = 0.f; CK(hipEventElapsedTime(&gms, g0, g1)); const double fl = 4.0 * (double)BH * S * S * d;
      printf("  %-28s %-13s %8.3f ms/launch  %7.1f TFLOP/s (%d launches)\n", "forward, generation 1 (LAB)", "attn_fwd", gms / iters, fl / (gms / iters) / 1e9, iters); }
    unsigned long long* bad; float* maxd; CK(hipMalloc(&bad, 8)); CK(hipMalloc(&maxd, 4));
    CK(hipMemsetAsync(bad, 0, 8, st)); CK(hipMemsetAsync(maxd, 0, 4, st));
    k_diff<<<2048, 256, 0, st>>>(O, O1, (int64_t)nr, 0, 0, bad, maxd);
    unsigned long long hb; float hm;
    CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hm, maxd, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    printf("  O vs generation 1: %llu of %lld elements differ (max |d| %.3e)\n", hb, (long long)nr, hm);
  }
  if (strcmp(gen, "1") != 0 && S % 64 == 0) {   // k_attn_fwd64 (one wave per SIMD, 64 queries per wave): timed, O and lse2 compared with k_attn_fwd4
    bf16* O6; float* lse6; CK(hipMalloc(&O6, nr * 2)); CK(hipMalloc(&lse6, (size_t)BH * S * 4));
    CK(hipMemsetAsync(O6, 0, nr * 2, st)); CK(hipMemsetAsync(lse6, 0, (size_t)BH * S * 4, st));
    g_attn_fwd_impl = 64;
    RC(st355_attn_fwd(st, Q, K, Vt, nullptr, O6, D, lse6, B, H, S, Sp, d, scale));
    CK(hipStreamSynchronize(st));
    st355_prof_reset(); st355_prof_enable(1);
    for (int i = 0; i < iters; i++) RC(st355_attn_fwd(st, Q, K, Vt, nullptr, O6, D, lse6, B, H, S, Sp, d, scale));
    CK(hipStreamSynchronize(st));
    st355_prof_enable(0); prof_print("forward, fwd64");
    g_attn_fwd_impl = 32;
    float* maxd; double *sd, *sr; CK(hipMalloc(&maxd, 4)); CK(hipMalloc(&sd, 8)); CK(hipMalloc(&sr, 8));
    CK(hipMemsetAsync(maxd, 0, 4, st)); CK(hipMemsetAsync(sd, 0, 8, st)); CK(hipMemsetAsync(sr, 0, 8, st));
    k_absdiff<<<2048, 256, 0, st>>>(O6, O, (int64_t)nr, maxd, sd, sr);
    float hm; double hd, hr;
    CK(hipMemcpyAsync(&hm, maxd, 4, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hd, sd, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hr, sr, 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("  O, fwd64 vs fwd4: rel-L2 %.3e, max |d| %.3e  %s\n", sqrt(hd / (hr + 1e-30)), hm, sqrt(hd / (hr + 1e-30)) < 4e-3 ? "within bf16 rounding" : "MISMATCH");
    float* hl = (float*)malloc((size_t)BH * S * 4); float* hl6 = (float*)malloc((size_t)BH * S * 4);
    CK(hipMemcpy(hl, lse2, (size_t)BH * S * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl6, lse6, (size_t)BH * S * 4, hipMemcpyDeviceToHost));
    double ml = 0; for (size_t i = 0; i < (size_t)BH * S; i++) { double e = fabs((double)hl[i] - hl6[i]); if (!(e <= ml)) ml = e; }
    printf("  lse2, fwd64 vs fwd4: max |d| %.3e  %s\n", ml, ml < 1e-4 ? "ok" : "MISMATCH");
    {   // FNV-1a over the bytes of O and lse2: two lab binaries (tools/kgen/variants.sh) whose lines agree produced bit-identical results
      unsigned short* ho = (unsigned short*)malloc((size_t)nr * 2); CK(hipMemcpy(ho, O6, (size_t)nr * 2, hipMemcpyDeviceToHost));
      unsigned long long f = 1469598103934665603ull;
      for (size_t i = 0; i < (size_t)nr; i++) { f = (f ^ ho[i]) * 1099511628211ull; }
      unsigned long long g = 1469598103934665603ull;
      for (size_t i = 0; i < (size_t)BH * S; i++) { unsigned u; memcpy(&u, &hl6[i], 4); g = (g ^ u) * 1099511628211ull; }
      printf("  fwd64 checksums: O %016llx  lse2 %016llx\n", f, g);
      free(ho);
    }
    if (g_attn_fwd_trace) {
      unsigned long long t[64]; CK(hipMemcpy(t, g_attn_fwd_trace, sizeof(t), hipMemcpyDeviceToHost));
      for (int w = 0; w < 4; w++) printf("  fwd64 trace wave %d: A %llu  C %llu  (step %llu cycles)\n", w, t[16 * w + 1] - t[16 * w], t[16 * w + 2] - t[16 * w + 1], t[16 * w + 2] - t[16 * w]);
    }
  }
  if (strcmp(gen, "1") != 0 && d == 128 && S % 64 == 0 && getenv("LAB_STALE")) {   // LAB-ONLY variant: stale-maximum rescale (tools/attn_fwd_variants.hip), timed with events, compared with the product's O
    bf16* O3; float* lse4; CK(hipMalloc(&O3, nr * 2)); CK(hipMalloc(&lse4, (size_t)BH * S * 4));
    const int lds = 2 * (KB * 256 + 128 * 128);
    CK(hipFuncSetAttribute((const void*)k_attn_fwd4_stale<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid((S + QB - 1) / QB, H, B), block(ATT_THREADS);
    auto launch = [&]() {
      hipLaunchKernelGGL(k_attn_fwd4_stale<128>, grid, block, lds, st, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt, O3, D, lse4, H, S, S, Sp, scale * LOG2E);
    };
    launch(); CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 4.0 * (double)BH * S * S * d;
    printf("  %-28s %-13s %8.3f ms/launch  %7.1f TFLOP/s (%d launches)\n", "forward, stale-max (LAB)", "attn_fwd", ms / iters, flops / (ms / iters) / 1e9, iters);
    float* maxd; double *sd, *sr; CK(hipMalloc(&maxd, 4)); CK(hipMalloc(&sd, 8)); CK(hipMalloc(&sr, 8));
    CK(hipMemsetAsync(maxd, 0, 4, st)); CK(hipMemsetAsync(sd, 0, 8, st)); CK(hipMemsetAsync(sr, 0, 8, st));
    k_absdiff<<<2048, 256, 0, st>>>(O3, O, (int64_t)nr, maxd, sd, sr);
    float hm; double hd, hr;
    CK(hipMemcpyAsync(&hm, maxd, 4, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hd, sd, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hr, sr, 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("  O, stale-max variant vs product kernel: rel-L2 %.3e, max |d| %.3e  %s\n", sqrt(hd / (hr + 1e-30)), hm, sqrt(hd / (hr + 1e-30)) < 4e-3 ? "within bf16 rounding" : "MISMATCH");
  }
  if (getenv("LAB_FWD_ONLY")) { printf("  (LAB_FWD_ONLY: backward skipped)\n"); return 0; }
  // ---- backward: with the transposed copies (dkv2 + dq) and without (dkv3 + dq<TR>) ----
  bf16* dQ3; CK(hipMalloc(&dQ3, nh * 2)); CK(hipMemsetAsync(dQ3, 0, nh * 2, st));
  if (getenv("LAB_DQ_TRACE")) { CK(hipMalloc(&g_attn_dq_trace, 4 * 128)); CK(hipMemsetAsync(g_attn_dq_trace, 0, 4 * 128, st)); }
  bf16 *dK4, *dqkv4; CK(hipMalloc(&dK4, nh * 2)); CK(hipMalloc(&dqkv4, nr * 3 * 2)); CK(hipMemsetAsync(dK4, 0, nh * 2, st)); CK(hipMemsetAsync(dqkv4, 0, nr * 3 * 2, st));
  for (int pass = 0; pass < 4; pass++) {     // 0: transposed copies (dkv2 + dq)   1: no copies (dkv3 + dq<TR>)   2: dkv3 + dq64   3: dkv4 + dq64
    g_attn_dq_impl = pass >= 2 ? 64 : 32;
    g_attn_dkv_impl = pass == 3 ? 4 : 3;
    const bf16* qt = pass ? nullptr : Qt; const bf16* kt = pass ? nullptr : Kt;
    bf16 *dq_ = pass >= 2 ? dQ3 : pass ? dQ2 : dQ, *dk_ = pass == 3 ? dK4 : pass ? dK2 : dK, *dv_ = (pass == 3 ? dqkv4 : pass ? dqkv2 : dqkv) + 2 * D;
    RC(st355_attn_bwd(st, Q, K, qt, kt, vrows, 3 * D, O, D, dO, D, lse2, nullptr, dq_, dk_, dv_, 3 * D, B, H, S, Sp, d, scale, ws));
    CK(hipStreamSynchronize(st));
    st355_prof_reset(); st355_prof_enable(1);
    for (int i = 0; i < iters; i++) RC(st355_attn_bwd(st, Q, K, qt, kt, vrows, 3 * D, O, D, dO, D, lse2, nullptr, dq_, dk_, dv_, 3 * D, B, H, S, Sp, d, scale, ws));
    CK(hipStreamSynchronize(st));
    st355_prof_enable(0); prof_print(pass == 3 ? "bwd, dkv4 + dq64" : pass == 2 ? "bwd, no copies, dq64" : pass ? "bwd, no transposed copies" : "bwd, Q^T/K^T/dO^T copies");
  }
  if (g_attn_dq_trace) {      // s_memtime stamps of block 0's LAST main-loop iteration: loop top, after A / after C of step 1, after A / after C of step 2
    unsigned long long t[64]; CK(hipMemcpy(t, g_attn_dq_trace, sizeof(t), hipMemcpyDeviceToHost));
    for (int w = 0; w < 4; w++) {
      const unsigned long long* u = t + 16 * w;
      printf("  dq64 trace wave %d: A1 %llu  C1 %llu  A2 %llu  C2 %llu  (tile %llu cycles) | prologue %llu  loop+last %llu  park %llu  (statement %llu)\n", w, u[1] - u[0],
             u[2] - u[1], u[3] - u[2], u[4] - u[3], u[4] - u[0], u[6] - u[5], u[7] - u[6], u[8] - u[7], u[8] - u[5]);
    }
  }
  if (d == 128 || d == 96 || d == 64) {          // dkv4 vs dkv3: K is pre-scaled and re-rounded in dkv4 -> agreement to bf16 rounding
    float* maxd; double *sd, *sr; CK(hipMalloc(&maxd, 4)); CK(hipMalloc(&sd, 8)); CK(hipMalloc(&sr, 8));
    struct { const char* n; const bf16* a; const bf16* b; int64_t cnt; } c4[] = {{"dK", dK4, dK2, (int64_t)nh}, {"dV (whole dqkv rows)", dqkv4, dqkv2, (int64_t)nr * 3}};
    for (auto& c : c4) {
      CK(hipMemsetAsync(maxd, 0, 4, st)); CK(hipMemsetAsync(sd, 0, 8, st)); CK(hipMemsetAsync(sr, 0, 8, st));
      k_absdiff<<<2048, 256, 0, st>>>(c.a, c.b, c.cnt, maxd, sd, sr);
      float hm; double hd, hr;
      CK(hipMemcpyAsync(&hm, maxd, 4, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hd, sd, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hr, sr, 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      printf("  %s, dkv4 vs dkv3: rel-L2 %.3e, max |d| %.3e  %s\n", c.n, sqrt(hd / (hr + 1e-30)), hm, sqrt(hd / (hr + 1e-30)) < 6e-3 ? "within bf16 rounding" : "MISMATCH");
    }
  }
  if (d != 128 && d != 64 && d != 96) return 0;
  unsigned long long* bad; float* maxd; CK(hipMalloc(&bad, 8)); CK(hipMalloc(&maxd, 4));
  struct { const char* n; const bf16* a; const bf16* b; int64_t cnt, ld, cols; } cmp[] = {
      {"dQ (dq64 vs dq)", dQ2, dQ3, (int64_t)nh, 0, 0}, {"dQ", dQ, dQ2, (int64_t)nh, 0, 0}, {"dK", dK, dK2, (int64_t)nh, 0, 0}, {"dV", dqkv + 2 * D, dqkv2 + 2 * D, (int64_t)nr, 3 * D, D}};
  for (auto& c : cmp) {
    CK(hipMemsetAsync(bad, 0, 8, st)); CK(hipMemsetAsync(maxd, 0, 4, st));
    k_diff<<<2048, 256, 0, st>>>(c.a, c.b, c.cnt, c.ld, c.cols, bad, maxd);
    unsigned long long hb; float hm;
    CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hm, maxd, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    printf("  %s: copies vs no-copies path: %llu of %lld elements differ (max |d| %.3e)  %s\n", c.n, hb, (long long)c.cnt, hm, hb ? "MISMATCH" : "bit-identical");
  }
  return 0;
}
