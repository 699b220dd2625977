// atomic_probe — how fast can 256 workgroups accumulate fp32 tiles into shared rows?  (design input for a single-pass attention backward whose dQ tiles are
// summed over the key-owner workgroups of a head).  Pattern: HB "heads", each with NW workgroups placed on ONE XCD (block id % 8 == head % 8, as attn_wg_map does);
// every workgroup walks the head's T tiles of 32 KiB (64 x 128 fp32) and adds its contribution to each.  Modes: 0 plain stores (upper bound), 1 agent-scope
// atomic add, 2 workgroup-scope atomic add (executes in the XCD's L2 — only valid because the adders share an XCD), 3 agent-scope packed bf16 add (16 KiB tiles).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/atomic_probe.hip -o tools/probes/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_acc(float* __restrict__ buf, int T, int NW, int stagger, int spread) {
  const int L = blockIdx.x, slot = L >> 3, grp = slot / NW;
  int head = (L & 7) + 8 * grp, w = slot - grp * NW;
  if (spread) { head = L / NW; w = L - head * NW; }      // the adders of one head on all 8 XCDs
  float* base = buf + (size_t)head * T * 8192;
  for (int t0 = 0; t0 < T; t0++) {
    const int t = stagger ? (t0 + w * (T / NW)) % T : t0;
    float* p = base + (size_t)t * 8192 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const float v = (float)(i + w);
      if (MODE == 0) p[i * 256] = v;
      else if (MODE == 1) __hip_atomic_fetch_add(p + i * 256, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 2) __hip_atomic_fetch_add(p + i * 256, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // stand-in for the MFMA work between two tiles of a real kernel: ~20 us of dependent FMAs would hide everything; keep it short to expose the add rate
  }
}
int main(int argc, char** argv) {
  const int HB = 192, NW = 18, T = 72;
  float* buf; const size_t n = (size_t)HB * T * 8192;
  CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)HB * NW * T * 32768.0;
  for (int spread = 0; spread < 2; spread++)
  for (int stagger = 0; stagger < 2; stagger++)
    for (int mode = 0; mode < 3; mode++) {
      float best = 1e9;
      for (int it = 0; it < 3; it++) {
        CK(hipMemset(buf, 0, n * 4));
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_acc<0>, dim3(HB * NW), dim3(256), 0, 0, buf, T, NW, stagger, spread);
        if (mode == 1) hipLaunchKernelGGL(k_acc<1>, dim3(HB * NW), dim3(256), 0, 0, buf, T, NW, stagger, spread);
        if (mode == 2) hipLaunchKernelGGL(k_acc<2>, dim3(HB * NW), dim3(256), 0, 0, buf, T, NW, stagger, spread);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      // check: every element must be sum_w (i + w) for atomics
      float h[4]; CK(hipMemcpy(h, buf + 5 * 8192 + 256 * 3, 16, hipMemcpyDeviceToHost));
      const float want = mode ? NW * 3.f + NW * (NW - 1) / 2.f : -1.f;
      printf("spread %d mode %d (%s) stagger %d: %.3f ms  %.2f TB/s of tile bytes  sample %.1f (want %.1f)%s\n", spread, mode, mode == 0 ? "plain store" : mode == 1 ? "atomic agent" : "atomic wg-scope",
             stagger, best, bytes / best / 1e9, h[0], want, (mode && h[0] != want) ? "  WRONG" : "");
    }
  return 0;
}
