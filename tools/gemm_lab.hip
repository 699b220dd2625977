// gemm_lab — standalone GEMM schedule lab for libst355 (no torch: starts in milliseconds on a fresh GPU box).
//   tools/gemm_lab [impl ...]      impl in {s2,p3,pq}; default: all.  Re-executes itself per impl (the choice is cached per process).
// For every shape: checks st355_gemm_bf16 against a naive fp32-accumulate reference kernel on uniform random operands (every
// element, transpose-detecting), then times 30 launches with hipEvents and prints TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast [-DST355_TRACE] tools/gemm_lab.hip -o tools/gemm_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <vector>
#include "../simpletuner_amd/csrc/runtime.hip"   // the lab compiles the library sources in (so -DST355_TRACE can instrument them)
#include "../simpletuner_amd/csrc/gemm.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void k_fill(bf16* p, int64_t n, uint32_t seed, float scale) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = (bf16)(((h >> 8) * (1.f / 8388608.f) - 1.f) * scale);
  }
}
// C_ref[m,n] = sum_k A[m,k] B[n,k] (+ A2 B2), fp32
__global__ void k_ref(const bf16* A, const bf16* B, const bf16* A2, const bf16* B2, float* C, int M, int N, int K, int K2) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; k++) s += (float)A[(int64_t)m * K + k] * (float)B[(int64_t)n * K + k];
  for (int k = 0; k < K2; k++) s += (float)A2[(int64_t)m * K2 + k] * (float)B2[(int64_t)n * K2 + k];
  C[(int64_t)m * N + n] = s;
}
__global__ void k_cmp(const bf16* C, const float* R, int64_t n, float* out /*max abs err, max abs ref*/, unsigned long long* bad, float atol, float rtol) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float d = fabsf((float)C[i] - R[i]);
    if (!(d <= atol + rtol * fabsf(R[i]))) atomicAdd(bad, 1ull);
    atomicMax((int*)&out[0], __float_as_int(d));
    atomicMax((int*)&out[1], __float_as_int(fabsf(R[i])));
  }
}

__global__ void k_neq(const uint16_t* a, const uint16_t* b, int64_t n, unsigned long long* bad) {
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(bad, c);
}

struct Shape { int M, N, K, K2; };

__global__ void k_fill8(uint8_t* p, int64_t n, uint32_t seed) {   // pseudo-random FINITE fp8 bit patterns (mask keeps e5m2 < inf, e4m3 != NaN)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (uint8_t)(h & 0xB7u);
  }
}
int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "--child") != 0) {
    for (int i = 1; i < argc; i++) {
      setenv("ST355_GEMM_IMPL", argv[i], 1);
      fflush(stdout);
      if (fork() == 0) { execl(argv[0], argv[0], "--child", (char*)0); _exit(3); }
      int st; wait(&st);
    }
    return 0;
  }
  setenv("ST355_GEMM_MIN_TILES", "1", 0);   // force the 256x256 schedules on the small verification shapes too
  const char* impl = getenv("ST355_GEMM_IMPL");
  printf("== impl %s\n", impl ? impl : "default");
  const Shape check[] = {{512, 512, 256, 0}, {768, 1280, 192, 64}, {4608, 3072, 3072, 64}, {300, 520, 128, 0}};
  const Shape perf[] = {{4608, 3072, 3072, 0}, {4608, 9216, 3072, 0}, {4608, 12288, 3072, 0}, {4608, 3072, 12288, 0},
                        {18432, 3072, 3072, 0}, {18432, 12288, 3072, 0}, {18432, 3072, 12288, 0}, {8192, 8192, 8192, 0}, {4096, 4096, 4096, 0}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto run = [&](const Shape& s, bool verify) {
    bf16 *A, *B, *A2 = nullptr, *B2 = nullptr, *C;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&B, (size_t)s.N * s.K * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
    k_fill<<<1024, 256, 0, st>>>(A, (int64_t)s.M * s.K, 1u, 1.f);
    k_fill<<<1024, 256, 0, st>>>(B, (int64_t)s.N * s.K, 2u, 0.05f);
    if (s.K2) {
      CK(hipMalloc(&A2, (size_t)s.M * s.K2 * 2)); CK(hipMalloc(&B2, (size_t)s.N * s.K2 * 2));
      k_fill<<<256, 256, 0, st>>>(A2, (int64_t)s.M * s.K2, 3u, 1.f);
      k_fill<<<256, 256, 0, st>>>(B2, (int64_t)s.N * s.K2, 4u, 0.05f);
    }
    CK(hipMemsetAsync(C, 0xff, (size_t)s.M * s.N * 2, st));
    st355_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = s.K; a.B = B; a.ldb = s.K; a.A2 = A2; a.lda2 = s.K2; a.B2 = B2; a.ldb2 = s.K2;
    a.C = C; a.ldc = s.N; a.M = s.M; a.N = s.N; a.K = s.K; a.K2 = s.K2; a.epilogue = ST355_EPI_NONE;
    int rc = st355_gemm_bf16(st, &a);
    if (rc) { printf("  gemm rc=%d: %s\n", rc, st355_last_error()); exit(2); }
    if (verify) {
      float* R; float* out; unsigned long long* bad;
      CK(hipMalloc(&R, (size_t)s.M * s.N * 4)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&bad, 8));
      CK(hipMemsetAsync(out, 0, 8, st)); CK(hipMemsetAsync(bad, 0, 8, st));
      k_ref<<<dim3((s.N + 255) / 256, s.M), 256, 0, st>>>(A, B, A2, B2, R, s.M, s.N, s.K, s.K2);
      k_cmp<<<1024, 256, 0, st>>>(C, R, (int64_t)s.M * s.N, out, bad, 2e-2f, 1e-2f);
      float h[2]; unsigned long long hb;
      CK(hipMemcpyAsync(h, out, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      printf("  check %5d x %5d x %5d (+%d): max|err| %.4f  max|ref| %.3f  bad %llu  %s\n", s.M, s.N, s.K, s.K2, h[0], h[1], hb, hb ? "FAIL" : "ok");
      CK(hipFree(R)); CK(hipFree(out)); CK(hipFree(bad));
    } else {
      for (int i = 0; i < 5; i++) st355_gemm_bf16(st, &a);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = getenv("LAB_ITERS") ? atoi(getenv("LAB_ITERS")) : 30;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; i++) st355_gemm_bf16(st, &a);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("  %6d x %6d x %6d: %8.1f us  %8.1f TFLOP/s\n", s.M, s.N, s.K, ms * 1e3, 2.0 * s.M * s.N * (double)s.K / ms / 1e9);
#ifdef ST355_TRACE
      uint64_t tb[2][TR_MARKS];
      CK(hipMemcpyFromSymbol(tb, HIP_SYMBOL(st355_trace_buf), sizeof(tb)));
      for (int w = 0; w < 2; w++) {
        printf("    trace wave %d k-tile %d: start %+6lld  section cycles:", w * 4, TR_T0, (long long)(tb[w][0] - tb[0][0]));
        for (int k = 1; k < TR_MARKS; k++) printf(" %5lld", (long long)(tb[w][k] - tb[w][k - 1]));
        printf("   total %lld\n", (long long)(tb[w][TR_MARKS - 1] - tb[w][0]));
      }
#endif
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    if (A2) { CK(hipFree(A2)); CK(hipFree(B2)); }
  };
  if (getenv("LAB_PZ")) {         // persistent (k_gemm_pz) vs one-tile-per-workgroup (k_gemm_pq): bit equality of every output + interleaved timing
    struct ES { int M, N, K, K2, epi; const char* what; };
    const ES list[] = {{9216, 3072, 512, 64, ST355_EPI_NONE, "odd K-tile count (ring parity flips per tile) + K-ext"},
                       {9216, 3072, 256, 0, ST355_EPI_GELU, "4 K-tiles, GELU + pre-act store"},
                       {5120, 4096, 320, 0, ST355_EPI_GATE_RESIDUAL, "5 K-tiles, gate + residual"},
                       {36864, 12288, 3072, 0, ST355_EPI_NONE, "MLP up, plain"}, {36864, 12288, 3072, 0, ST355_EPI_GELU, "MLP up, GELU + pre-act store"},
                       {36864, 12288, 3072, 0, ST355_EPI_MUL_GELU_GRAD, "dgrad, x GELU'(pre-act)"}, {36864, 3072, 12288, 0, ST355_EPI_NONE, "MLP down, plain"},
                       {36864, 3072, 12288, 0, ST355_EPI_GATE_RESIDUAL, "MLP down, gate * y + residual"}, {36864, 9216, 3072, 128, ST355_EPI_NONE, "QKV + LoRA K-ext"},
                       {36864, 3072, 9216, 128, ST355_EPI_ADD, "dQKV dgrad + LoRA K-ext + add"}, {36864, 3072, 3072, 0, ST355_EPI_NONE, "3072^2, plain"},
                       {32768, 1536, 1536, 0, ST355_EPI_NONE, "SD3 1536^2"}, {32768, 6144, 1536, 0, ST355_EPI_GELU, "SD3 MLP up"}, {32768, 1536, 6144, 0, ST355_EPI_GATE_RESIDUAL, "SD3 MLP down"},
                       {8192, 8192, 8192, 0, ST355_EPI_NONE, "8192^3"}};
    for (const ES& e : list) {
      bf16 *A, *B, *A2 = nullptr, *B2 = nullptr, *C[2], *aux, *auxo[2], *gate, *bias;
      CK(hipMalloc(&A, (size_t)e.M * e.K * 2)); CK(hipMalloc(&B, (size_t)e.N * e.K * 2));
      for (int v = 0; v < 2; v++) { CK(hipMalloc(&C[v], (size_t)e.M * e.N * 2)); CK(hipMalloc(&auxo[v], (size_t)e.M * e.N * 2)); }
      CK(hipMalloc(&aux, (size_t)e.M * e.N * 2)); CK(hipMalloc(&gate, (size_t)8 * e.N * 2)); CK(hipMalloc(&bias, (size_t)e.N * 2));
      k_fill<<<1024, 256, 0, st>>>(A, (int64_t)e.M * e.K, 1u, 1.f); k_fill<<<1024, 256, 0, st>>>(B, (int64_t)e.N * e.K, 2u, 0.05f);
      k_fill<<<1024, 256, 0, st>>>(aux, (int64_t)e.M * e.N, 5u, 1.f); k_fill<<<64, 256, 0, st>>>(gate, (int64_t)8 * e.N, 6u, 1.f); k_fill<<<64, 256, 0, st>>>(bias, e.N, 7u, 0.1f);
      if (e.K2) {
        CK(hipMalloc(&A2, (size_t)e.M * e.K2 * 2)); CK(hipMalloc(&B2, (size_t)e.N * e.K2 * 2));
        k_fill<<<256, 256, 0, st>>>(A2, (int64_t)e.M * e.K2, 3u, 1.f); k_fill<<<256, 256, 0, st>>>(B2, (int64_t)e.N * e.K2, 4u, 0.05f);
      }
      st355_gemm_args a[2];
      for (int v = 0; v < 2; v++) {
        memset(&a[v], 0, sizeof(a[v]));
        a[v].A = A; a[v].lda = e.K; a[v].B = B; a[v].ldb = e.K; a[v].A2 = A2; a[v].lda2 = e.K2; a[v].B2 = B2; a[v].ldb2 = e.K2; a[v].C = C[v]; a[v].ldc = e.N;
        a[v].M = e.M; a[v].N = e.N; a[v].K = e.K; a[v].K2 = e.K2; a[v].epilogue = e.epi; a[v].bias = bias;
        if (e.epi == ST355_EPI_GELU) { a[v].aux_out = auxo[v]; a[v].ld_aux_out = e.N; }
        if (e.epi == ST355_EPI_MUL_GELU_GRAD || e.epi == ST355_EPI_ADD || e.epi == ST355_EPI_GATE_RESIDUAL) { a[v].aux_in = aux; a[v].ld_aux_in = e.N; }
        if (e.epi == ST355_EPI_GATE_RESIDUAL) { a[v].gate = gate; a[v].gate_stride = e.N; a[v].rows_per_batch = e.M / 8; }
        CK(hipMemsetAsync(C[v], 0xff, (size_t)e.M * e.N * 2, st)); CK(hipMemsetAsync(auxo[v], 0xee, (size_t)e.M * e.N * 2, st));
      }
      unsigned long long* bad; CK(hipMalloc(&bad, 16)); CK(hipMemsetAsync(bad, 0, 16, st));
      unsigned long long hb[2] = {0, 0};
      for (int rep = 0; rep < 3; rep++) {          // three launches each: a race shows up as a non-repeatable difference
        for (int v = 0; v < 2; v++) { g_persist_override = v; int rc = st355_gemm_bf16(st, &a[v]); if (rc) { printf("rc=%d %s\n", rc, st355_last_error()); exit(2); } }
        k_neq<<<1024, 256, 0, st>>>((const uint16_t*)C[0], (const uint16_t*)C[1], (int64_t)e.M * e.N, bad);
        if (e.epi == ST355_EPI_GELU) k_neq<<<1024, 256, 0, st>>>((const uint16_t*)auxo[0], (const uint16_t*)auxo[1], (int64_t)e.M * e.N, bad + 1);
      }
      CK(hipMemcpyAsync(hb, bad, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
      float best[2] = {1e30f, 1e30f};
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int round = 0; round < 4; round++)
        for (int v = 0; v < 2; v++) {
          g_persist_override = v;
          const int iters = 6;
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < iters; i++) st355_gemm_bf16(st, &a[v]);
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
          if (round && ms < best[v]) best[v] = ms;
        }
      const double fl = 2.0 * e.M * e.N * (double)(e.K + e.K2);
      printf("  PZ %-40s %6d x %6d x %6d+%3d: pq %8.1f us %7.1f TF | pz %8.1f us %7.1f TF  (%+5.1f %%)  differing C %llu aux %llu %s\n", e.what, e.M, e.N, e.K, e.K2,
             best[0] * 1e3, fl / best[0] / 1e9, best[1] * 1e3, fl / best[1] / 1e9, 100.0 * (best[0] / best[1] - 1.0), hb[0], hb[1], (hb[0] | hb[1]) ? "MISMATCH" : "bit-identical");
      CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(aux)); CK(hipFree(gate)); CK(hipFree(bias)); CK(hipFree(bad));
      for (int v = 0; v < 2; v++) { CK(hipFree(C[v])); CK(hipFree(auxo[v])); }
      if (A2) { CK(hipFree(A2)); CK(hipFree(B2)); }
    }
    g_persist_override = -1;
    return 0;
  }
  if (getenv("LAB_EPI")) {        // the step's big shapes with their fused epilogues (Flux.1 LoRA at per-GPU batch 8: 36 864 joint tokens)
    struct ES { int M, N, K, K2, epi; const char* what; };
    const ES list[] = {{36864, 12288, 3072, 0, ST355_EPI_NONE, "MLP up, plain"}, {36864, 12288, 3072, 0, ST355_EPI_GELU, "MLP up, GELU + pre-act store"},
                       {36864, 12288, 3072, 0, ST355_EPI_MUL_GELU_GRAD, "dgrad, x GELU'(pre-act)"}, {36864, 3072, 12288, 0, ST355_EPI_NONE, "MLP down, plain"},
                       {36864, 3072, 12288, 0, ST355_EPI_GATE_RESIDUAL, "MLP down, gate * y + residual"}, {36864, 9216, 3072, 128, ST355_EPI_NONE, "QKV + LoRA K-ext"},
                       {36864, 3072, 9216, 128, ST355_EPI_ADD, "dQKV dgrad + LoRA K-ext + add"}, {36864, 3072, 3072, 0, ST355_EPI_NONE, "3072^2, plain"}};
    for (const ES& e : list) {
      bf16 *A, *B, *A2 = nullptr, *B2 = nullptr, *C, *aux, *gate, *bias;
      CK(hipMalloc(&A, (size_t)e.M * e.K * 2)); CK(hipMalloc(&B, (size_t)e.N * e.K * 2)); CK(hipMalloc(&C, (size_t)e.M * e.N * 2));
      CK(hipMalloc(&aux, (size_t)e.M * e.N * 2)); CK(hipMalloc(&gate, (size_t)8 * e.N * 2)); CK(hipMalloc(&bias, (size_t)e.N * 2));
      k_fill<<<1024, 256, 0, st>>>(A, (int64_t)e.M * e.K, 1u, 1.f); k_fill<<<1024, 256, 0, st>>>(B, (int64_t)e.N * e.K, 2u, 0.05f);
      k_fill<<<1024, 256, 0, st>>>(aux, (int64_t)e.M * e.N, 5u, 1.f); k_fill<<<64, 256, 0, st>>>(gate, (int64_t)8 * e.N, 6u, 1.f); k_fill<<<64, 256, 0, st>>>(bias, e.N, 7u, 0.1f);
      if (e.K2) {
        CK(hipMalloc(&A2, (size_t)e.M * e.K2 * 2)); CK(hipMalloc(&B2, (size_t)e.N * e.K2 * 2));
        k_fill<<<256, 256, 0, st>>>(A2, (int64_t)e.M * e.K2, 3u, 1.f); k_fill<<<256, 256, 0, st>>>(B2, (int64_t)e.N * e.K2, 4u, 0.05f);
      }
      st355_gemm_args a;
      memset(&a, 0, sizeof(a));
      a.A = A; a.lda = e.K; a.B = B; a.ldb = e.K; a.A2 = A2; a.lda2 = e.K2; a.B2 = B2; a.ldb2 = e.K2; a.C = C; a.ldc = e.N;
      a.M = e.M; a.N = e.N; a.K = e.K; a.K2 = e.K2; a.epilogue = e.epi; a.bias = bias;
      if (e.epi == ST355_EPI_GELU) { a.aux_out = aux; a.ld_aux_out = e.N; }
      if (e.epi == ST355_EPI_MUL_GELU_GRAD || e.epi == ST355_EPI_ADD || e.epi == ST355_EPI_GATE_RESIDUAL) { a.aux_in = aux; a.ld_aux_in = e.N; }
      if (e.epi == ST355_EPI_GATE_RESIDUAL) { a.gate = gate; a.gate_stride = e.N; a.rows_per_batch = e.M / 8; }
      for (int i = 0; i < 3; i++) { int rc = st355_gemm_bf16(st, &a); if (rc) { printf("rc=%d %s\n", rc, st355_last_error()); exit(2); } }
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 10;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; i++) st355_gemm_bf16(st, &a);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      printf("  EPI %-34s %6d x %6d x %6d+%3d: %8.1f us  %8.1f TFLOP/s\n", e.what, e.M, e.N, e.K, e.K2, ms * 1e3, 2.0 * e.M * e.N * (double)(e.K + e.K2) / ms / 1e9);
      CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(aux)); CK(hipFree(gate)); CK(hipFree(bias));
      if (A2) { CK(hipFree(A2)); CK(hipFree(B2)); }
    }
    return 0;
  }
  if (const char* one = getenv("LAB_SHAPE")) {          // LAB_SHAPE=M,N,K : one timed shape only (for rocprofv3 --pmc runs)
    Shape s = {0, 0, 0, 0};
    sscanf(one, "%d,%d,%d", &s.M, &s.N, &s.K);
    run(s, false);
    return 0;
  }
  if (!getenv("LAB_F8")) {
  for (const Shape& s : check) run(s, true);
  for (const Shape& s : perf) run(s, false);
  }
  // fp8-native Linear (e5m2 activations x e4m3 weights, row scales in the epilogue): timing only, parity is tests/test_fp8_linear.py
  for (const Shape& s : {Shape{4608, 3072, 3072, 0}, Shape{18432, 3072, 3072, 0}, Shape{18432, 12288, 3072, 0}, Shape{18432, 3072, 12288, 0},
                         Shape{8192, 8192, 8192, 0}, Shape{16384, 16384, 8192, 0}}) {
    uint8_t *X, *W; float *sa, *ws; bf16* C;
    CK(hipMalloc(&X, (size_t)s.M * s.K)); CK(hipMalloc(&W, (size_t)s.N * s.K)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
    CK(hipMalloc(&sa, 4)); CK(hipMalloc(&ws, (size_t)s.N * 4));
    k_fill8<<<1024, 256, 0, st>>>(X, (int64_t)s.M * s.K, 7u); k_fill8<<<1024, 256, 0, st>>>(W, (int64_t)s.N * s.K, 8u);
    CK(hipMemsetAsync(sa, 0x3c, 4, st)); CK(hipMemsetAsync(ws, 0x3c, (size_t)s.N * 4, st));
    for (int i = 0; i < 3; i++) st355_linear_fp8(st, X, s.K, sa, W, s.K, ws, nullptr, C, s.N, s.M, s.N, s.K);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; i++) st355_linear_fp8(st, X, s.K, sa, W, s.K, ws, nullptr, C, s.N, s.M, s.N, s.K);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("  F8 %6d x %6d x %6d: %8.1f us  %8.1f TFLOP/s\n", s.M, s.N, s.K, ms * 1e3, 2.0 * s.M * s.N * (double)s.K / ms / 1e9);
    CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(sa)); CK(hipFree(ws));
  }
  if (getenv("LAB_F8")) return 0;
  if (const char* tn = getenv("LAB_TN")) {                  // LAB_TN=P,Q,Mc : one weight-gradient problem, timed (ST355_TN_KS picks the K-slice count: tools/r05 ks sweep)
    Shape s = {0, 0, 0, 0};
    sscanf(tn, "%d,%d,%d", &s.M, &s.N, &s.K);
    bf16 *Lm, *Rm, *C; void* tws;
    CK(hipMalloc(&tws, (size_t)1024 << 20));
    CK(hipMalloc(&Lm, (size_t)s.K * s.M * 2)); CK(hipMalloc(&Rm, (size_t)s.K * s.N * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
    k_fill<<<1024, 256, 0, st>>>(Lm, (int64_t)s.K * s.M, 5u, 1.f);
    k_fill<<<1024, 256, 0, st>>>(Rm, (int64_t)s.K * s.N, 6u, 0.05f);
    for (int i = 0; i < 3; i++) st355_gemm_tn_bf16(st, Lm, s.M, Rm, s.N, C, s.N, s.K, s.M, s.N, 0, tws, (int64_t)1024 << 20);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; i++) st355_gemm_tn_bf16(st, Lm, s.M, Rm, s.N, C, s.N, s.K, s.M, s.N, 0, tws, (int64_t)1024 << 20);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    // checksum of C: two slice counts must agree to fp32 summation order (bf16 output)
    unsigned short* hc = (unsigned short*)malloc((size_t)s.M * s.N * 2); CK(hipMemcpy(hc, C, (size_t)s.M * s.N * 2, hipMemcpyDeviceToHost));
    double sum = 0, sq = 0; for (size_t i = 0; i < (size_t)s.M * s.N; i++) { unsigned u = (unsigned)hc[i] << 16; float f; memcpy(&f, &u, 4); sum += f; sq += (double)f * f; }
    printf("  TN %6d x %6d over %6d (ST355_TN_KS=%s): %8.1f us  %8.1f TFLOP/s   sum %.6e  l2 %.6e\n", s.M, s.N, s.K, getenv("ST355_TN_KS") ? getenv("ST355_TN_KS") : "-", ms * 1e3,
           2.0 * s.M * s.N * (double)s.K / ms / 1e9, sum, sqrt(sq));
    return 0;
  }
  // weight-gradient (TN) form: C[P,Q] = L[M,P]^T R[M,Q]; Shape {M=P, N=Q, K=contraction}
  for (const Shape& s : {Shape{3072, 3072, 4608, 0}, Shape{12288, 3072, 4608, 0}, Shape{3072, 12288, 18432, 0}, Shape{1536, 1536, 16384, 0},
                         Shape{6144, 1536, 16384, 0}, Shape{8192, 8192, 8192, 0}}) {
    bf16 *Lm, *Rm, *C;
    static void* tws = nullptr;
    if (!tws) CK(hipMalloc(&tws, (size_t)512 << 20));
    CK(hipMalloc(&Lm, (size_t)s.K * s.M * 2)); CK(hipMalloc(&Rm, (size_t)s.K * s.N * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
    k_fill<<<1024, 256, 0, st>>>(Lm, (int64_t)s.K * s.M, 5u, 1.f);
    k_fill<<<1024, 256, 0, st>>>(Rm, (int64_t)s.K * s.N, 6u, 0.05f);
    for (int i = 0; i < 3; i++) st355_gemm_tn_bf16(st, Lm, s.M, Rm, s.N, C, s.N, s.K, s.M, s.N, 0, tws, (int64_t)512 << 20);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; i++) st355_gemm_tn_bf16(st, Lm, s.M, Rm, s.N, C, s.N, s.K, s.M, s.N, 0, tws, (int64_t)512 << 20);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("  TN %6d x %6d over %6d: %8.1f us  %8.1f TFLOP/s\n", s.M, s.N, s.K, ms * 1e3, 2.0 * s.M * s.N * (double)s.K / ms / 1e9);
    CK(hipFree(Lm)); CK(hipFree(Rm)); CK(hipFree(C));
  }
  return 0;
}
