// glds_probe — how fast can one CU pull GEMM operand tiles from L2 into LDS with global_load_lds_dwordx4, as a function of the
// row length of a piece (64-B rows = BK 32, 128-B rows = BK 64, 256-B rows = BK 128) and of the number of pieces in flight?
// Grid = 256 workgroups x 512 threads walking K exactly like the 256x256 GEMM does (same tile -> panel mapping, same XCD remap),
// no MFMA, no consumers: each wave keeps DEPTH pieces in flight with a counted vmcnt.  Prints bytes/clk/CU (at 2.4 GHz) and TB/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/glds_probe.hip -o tools/probes/glds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ void glds16(const bf16* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

template <int N> __device__ __forceinline__ void waitvm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
}

// ROWB = bytes of one row inside a piece (64 / 128 / 256); a piece = 1 KiB = (1024/ROWB) rows.  Per "step" every wave issues 2
// pieces of the X panel and 2 of the W panel (= 32 KiB per workgroup, the traffic of one BK=32 k-step of a 256x256 tile).
// DEPTH = pieces allowed in flight per wave after the wait.
template <int ROWB, int DEPTH>
__global__ void __launch_bounds__(512, 2) k_probe(const bf16* A, const bf16* B, int64_t ld, int K, int nbm, int nbn, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int GROUP = 8, width = GROUP * nbn, gid = id / width, first_m = gid * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  const int pm = first_m + (id % width) % gsz, pn = (id % width) / gsz;
  constexpr int RPP = 1024 / ROWB;      // rows per piece
  constexpr int LPR = ROWB / 16;        // lanes per row
  // a k-window of ROWB bytes covers 256 rows = 256*ROWB bytes = (ROWB/4) KiB = ROWB/4 pieces per operand; 8 waves x 2 pieces = 16
  // pieces per operand per step => one step advances K by 16*1024/(256*ROWB) windows... keep it simple: per step, wave wv loads
  // pieces (rows) [ (2wv+j)*RPP, +RPP ) for j=0,1 of a 16*RPP-row band; the band index cycles through the 256 rows of the tile.
  constexpr int BANDS = 256 / (16 * RPP);   // bands per k-window (ROWB=64: 1, 128: 2, 256: 4)
  const int r_in = lane / LPR, c_in = (lane % LPR) * 8;
  int it = 0;
  const int steps = (K * 2 / ROWB) * BANDS;
  for (int s = 0; s < steps; s++) {
    const int kw = s / BANDS, band = s % BANDS;
    const int64_t k0 = (int64_t)kw * (ROWB / 2);
    char* slot = smem + (s & 3) * 32768;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int row = band * 16 * RPP + (2 * wv + j) * RPP + r_in;
      glds16(A + (int64_t)(pm * 256 + row) * ld + k0 + c_in, slot + (2 * wv + j) * 1024);
      glds16(B + (int64_t)(pn * 256 + row) * ld + k0 + c_in, slot + 16384 + (2 * wv + j) * 1024);
    }
    waitvm<DEPTH>();
    it++;
  }
  waitvm<0>();
  __syncthreads();
  if (tid == 0 && smem[(blockIdx.x * 16) & 32767] == 77) sink[0] = it;
}

template <int ROWB, int DEPTH>
static void run(const bf16* A, const bf16* B, int n, int* sink, hipStream_t st) {
  CK(hipFuncSetAttribute((const void*)k_probe<ROWB, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  const int nb = n / 256;
  for (int grid : {256, 1024}) {
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_probe<ROWB, DEPTH>), dim3(grid), dim3(512), 131072, st, A, B, (int64_t)n, n, nb, nb, sink);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 5;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_probe<ROWB, DEPTH>), dim3(grid), dim3(512), 131072, st, A, B, (int64_t)n, n, nb, nb, sink);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double bytes = (double)grid * (2.0 * 256 * n * 2);
    printf("  row %3d B, depth %2d pieces, grid %4d: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU@2.4GHz\n", ROWB, DEPTH, grid, ms * 1e3, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / 256 / 2.4e9);
  }
}

int main() {
  const int n = 8192;
  bf16 *A, *B; int* sink;
  CK(hipMalloc(&A, (size_t)n * n * 2)); CK(hipMalloc(&B, (size_t)n * n * 2)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0x11, (size_t)n * n * 2)); CK(hipMemset(B, 0x22, (size_t)n * n * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  run<64, 4>(A, B, n, sink, st);
  run<64, 8>(A, B, n, sink, st);
  run<64, 16>(A, B, n, sink, st);
  run<64, 32>(A, B, n, sink, st);
  run<128, 4>(A, B, n, sink, st);
  run<128, 8>(A, B, n, sink, st);
  run<128, 16>(A, B, n, sink, st);
  run<128, 32>(A, B, n, sink, st);
  run<256, 8>(A, B, n, sink, st);
  run<256, 16>(A, B, n, sink, st);
  run<256, 32>(A, B, n, sink, st);
  return 0;
}
