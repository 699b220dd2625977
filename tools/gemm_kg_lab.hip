// gemm_kg_lab.hip — LAB: times and checks the generated four-wave 256 x 256 x 64 GEMM main loop (tools/kgen/gemm4w.py) against st355_gemm_bf16 of libst355.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_kg_lab.hip -o tools/gemm_kg_lab -ldl     (from the repo root; python -m tools.kgen.gemm4w first)
//   tools/gemm_kg_lab [M N]         default 8192 8192 (1024 tiles = four full rounds of 256 CUs); K in {3072, 12288}: the difference isolates the main loop
// Not part of the product: no epilogue variants, M, N % 256 == 0, K % 64 == 0, K >= 128.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../include/st355.h"

typedef __bf16 bf16;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
__device__ __forceinline__ void tile_coords(int id, int nbm, int nbn, int& pm, int& pn) {
  const int GROUP = 8;
  const int width = GROUP * nbn;
  const int group_id = id / width;
  const int first_m = group_id * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  pm = first_m + (id % width) % gsz;
  pn = (id % width) / gsz;
}

__global__ void __launch_bounds__(256, 1) k_gemm_kg(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                    int ldc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int pm, pn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), M / 256, N / 256, pm, pn);
  const int wm = wv >> 1, wn = wv & 1;
  auto voff = [&](int j, int ld) {
    const int row = j * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    return (uint32_t)((row * ld + chunk * 8) * 2);
  };
  const uint32_t voae = voff(wv * 8, lda), voao = voff(wv * 8 + 1, lda), vobe = voff(wv * 8, ldb), vobo = voff(wv * 8 + 1, ldb);
  const uint32_t stepa = (uint32_t)(16 * lda * 2), stepb = (uint32_t)(16 * ldb * 2);
  const int r = lane & 31, h = lane >> 5, f = (r >> 1) & 7;
  const uint32_t lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t fa = lds + wm * 16384 + r * 128 + ((h ^ f) << 4);
  const uint32_t fb = lds + 32768 + wn * 16384 + r * 128 + ((h ^ f) << 4);
  const bf16* abase = A + (size_t)pm * 256 * lda;
  const bf16* bbase = B + (size_t)pn * 256 * ldb;
  bf16* cp = C + (size_t)(pm * 256 + wm * 128 + r) * ldc + pn * 256 + wn * 128 + 4 * h;
  const uint32_t cplo = (uint32_t)(uintptr_t)cp, cphi = (uint32_t)((uintptr_t)cp >> 32);
  const uint32_t cstep = (uint32_t)(32 * ldc * 2), nkt = (uint32_t)(K / 64), wv8k = (uint32_t)wv * 8192u;
  asm volatile(
#include "gemm4w_body.inc"
      :
      : [voae] "v"(voae), [voao] "v"(voao), [vobe] "v"(vobe), [vobo] "v"(vobo), [fa] "v"(fa), [fb] "v"(fb), [cplo] "v"(cplo), [cphi] "v"(cphi), [stepa] "s"(stepa),
        [stepb] "s"(stepb), [abase] "s"(abase), [bbase] "s"(bbase), [lds] "s"(lds), [wv8k] "s"(wv8k), [nkt] "s"(nkt), [cstep] "s"(cstep)
      : "memory", "vcc", "scc",
#include "gemm4w_clobbers.inc"
  );
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192;
  void* lib = dlopen("simpletuner_amd/csrc/libst355.so", RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  auto gemm = (int (*)(void*, const st355_gemm_args*))dlsym(lib, "st355_gemm_bf16");
  const int lds_bytes = 2 * 65536;
  CK(hipFuncSetAttribute((const void*)k_gemm_kg, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double t_kg[2] = {0, 0}, t_lib[2] = {0, 0};
  const int Ks[2] = {3072, 12288};
  for (int ki = 0; ki < 2; ki++) {
    const int K = Ks[ki];
    std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
    uint32_t s = 12345u + ki;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float v = ((int)((s >> 9) & 0xffff) - 32768) / 32768.0f; uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x8000u) >> 16); };
    const bool zero = getenv("LAB_ZERO") != nullptr;        // zero-filled operands: no data toggling, the chip holds a higher clock (what the kernels do when power is not the limit)
    if (!zero) {
      for (auto& x : ha) x = rnd();
      for (auto& x : hb) x = rnd();
    }
    bf16 *A, *B, *C0, *C1;
    CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C0, (size_t)M * N * 2)); CK(hipMalloc(&C1, (size_t)M * N * 2));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(C0, 0xff, (size_t)M * N * 2)); CK(hipMemset(C1, 0x7f, (size_t)M * N * 2));
    st355_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = C0; g.ldc = N; g.M = M; g.N = N; g.K = K; g.epilogue = ST355_EPI_NONE;
    const dim3 grid((M / 256) * (N / 256));
    auto run_kg = [&]() { hipLaunchKernelGGL(k_gemm_kg, grid, dim3(256), lds_bytes, st, (const bf16*)A, (const bf16*)B, C1, M, N, K, K, K, N); };
    if (gemm(st, &g) != 0) { fprintf(stderr, "st355_gemm_bf16 failed\n"); return 1; }
    run_kg();
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> c0((size_t)M * N), c1((size_t)M * N);
    CK(hipMemcpy(c0.data(), C0, c0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), C1, c1.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0; double num = 0, den = 0;
    for (size_t i = 0; i < (zero ? (size_t)4096 : c0.size()); i++) {
      if (c0[i] != c1[i]) diff++;
      uint32_t u0 = (uint32_t)c0[i] << 16, u1 = (uint32_t)c1[i] << 16; float f0, f1; memcpy(&f0, &u0, 4); memcpy(&f1, &u1, 4);
      num += (double)(f0 - f1) * (f0 - f1); den += (double)f0 * f0;
    }
    printf("K %5d: generated 4-wave body vs st355_gemm_bf16: %zu of %zu elements differ, rel-L2 %.3e  %s\n", K, diff, c0.size(), sqrt(num / (den + 1e-30)), diff == 0 ? "bit-identical" : "MISMATCH");
    const int iters = 10;
    for (int w = 0; w < 2; w++) { gemm(st, &g); run_kg(); }
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) gemm(st, &g);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t_lib[ki] = ms / iters;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) run_kg();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); t_kg[ki] = ms / iters;
    const double fl = 2.0 * M * N * (double)K;
    printf("K %5d: st355_gemm_bf16 %8.1f us %7.1f TFLOP/s | generated 4-wave body (plain bf16 store epilogue) %8.1f us %7.1f TFLOP/s\n", K, t_lib[ki] * 1e3, fl / t_lib[ki] / 1e9,
           t_kg[ki] * 1e3, fl / t_kg[ki] / 1e9);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C0)); CK(hipFree(C1));
  }
  const double flk = 2.0 * M * N * (double)(Ks[1] - Ks[0]);
  printf("main loop alone (K %d minus K %d): st355_gemm_bf16 %7.1f TFLOP/s | generated 4-wave body %7.1f TFLOP/s\n", Ks[1], Ks[0], flk / (t_lib[1] - t_lib[0]) / 1e9,
         flk / (t_kg[1] - t_kg[0]) / 1e9);
  return 0;
}
