// attention.hip — K7 forward: joint non-causal attention over [txt || img] latent patch tokens.
//   O = softmax(scale * Q K^T + key_bias) V          (flux/transformer.py:200-207 → F.scaled_dot_product_attention)
//
// gfx950 design: flash-style, LDS-tiled over 64-key tiles, online softmax entirely in registers.
//   * workgroup = 4 waves x 32 queries; 2 workgroups resident per CU (64 KiB LDS each, <=256 VGPR).
//   * scores are computed TRANSPOSED (S^T = K Q^T: K tile = MFMA A from LDS, Q = MFMA B held in registers),
//     so a lane owns one query column: row max / row sum are register reductions + one xor-32 exchange.
//   * K rows are fetched in perm23 order, so exp(S^T) registers are already the B operand of
//     O^T += V^T P^T  (attn_common.h) — P never leaves the register file.
//   * V arrives pre-transposed (Vt [B,H,d,Sp], written by the QKV epilogue kernel), so the V^T tile is a
//     plain row-major LDS image read with ds_read_b128.
//   * both LDS tiles use XOR-swizzled 16-byte chunks (conflict-free for the 16-lane ds_read_b128 groups).
//   * register-staged double buffering: next tile's global loads are issued before the MFMA work and
//     written to the other LDS buffer after it (T14 async-stage split).
#include <stdlib.h>
#include <type_traits>
#include "attn_common.h"

#define ATT_THREADS 256
#define QB 128  // queries per workgroup
#define KB 64   // keys per tile

// NW = waves per workgroup (32 queries each): 4 = two independent workgroups per CU (default); 8 = ONE 256-query workgroup per CU sharing each K / V^T
// tile (half the LDS fill and half the global->LDS staging work per MFMA, one barrier domain of 8 waves) — A/B with ST355_ATTN_FWD=3
template <int HD, int NW = 4>
__global__ void __launch_bounds__(64 * NW, 2) k_attn_fwd(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                            const bf16* __restrict__ Vt, const float* __restrict__ key_bias,
                                                            bf16* __restrict__ O, int64_t ld_o, float* __restrict__ lse2, int H,
                                                            int Sq, int S, int Sp, float scale2) {   // Sq queries; S keys (padded Sp)
  constexpr int KROWB = HD * 2;          // bytes per K tile row
  constexpr int KT_BYTES = KB * KROWB;   // K tile
  constexpr int VT_BYTES = HD * 128;     // V^T tile: HD rows x 64 keys
  constexpr int BUF = KT_BYTES + VT_BYTES;
  constexpr int NKS = HD / 16;           // MFMA k-steps over the head dim
  constexpr int NDT = HD / 32;           // 32-row d tiles of O^T
  constexpr int ATT_T = 64 * NW;
  constexpr int KCH = KT_BYTES / 16 / ATT_T;  // 16-B chunks per thread
  constexpr int VCH = VT_BYTES / 16 / ATT_T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = blockIdx.x * (32 * NW) + wv * 32;
  const int qi = min(q0 + l31, Sq - 1);

  const bf16* Kg = K + bh * (int64_t)S * HD;
  const bf16* Vg = Vt + bh * (int64_t)HD * Sp;

  // Q fragments (MFMA B operand): lane -> query l31, head channels 16ks + 8h .. +8
  bf16x8 qf[NKS];
  {
    const bf16* qrow = Q + (bh * Sq + qi) * (int64_t)HD + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
  }

  f32x16 acc_o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc_o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  bf16x8 kreg[KCH], vreg[VCH];
  auto load_tile = [&](int kt) {
    const int key0 = kt * KB;
#pragma unroll
    for (int p = 0; p < KCH; p++) {
      const int id = p * ATT_T + tid;
      const int row = id / (HD / 8), c = id % (HD / 8);
      kreg[p] = *(const bf16x8*)(Kg + (int64_t)min(key0 + row, S - 1) * HD + c * 8);
    }
#pragma unroll
    for (int p = 0; p < VCH; p++) {
      const int id = p * ATT_T + tid;
      const int row = id >> 3, c = id & 7;
      vreg[p] = *(const bf16x8*)(Vg + (int64_t)row * Sp + key0 + c * 8);
    }
  };
  auto store_tile = [&](int buf) {
    char* ks = smem + buf * BUF;
    char* vs = ks + KT_BYTES;
#pragma unroll
    for (int p = 0; p < KCH; p++) {
      const int id = p * ATT_T + tid;
      const int row = id / (HD / 8), c = id % (HD / 8);
      *(bf16x8*)(ks + lds_off<KROWB>(row, c)) = kreg[p];
    }
#pragma unroll
    for (int p = 0; p < VCH; p++) {
      const int id = p * ATT_T + tid;
      const int row = id >> 3, c = id & 7;
      *(bf16x8*)(vs + lds_off<128>(row, c)) = vreg[p];
    }
  };

  const int nkt = (S + KB - 1) / KB;
  const int krow_p = perm23(l31);  // K tile row (within a 32-key sub-block) this lane feeds as MFMA A row l31

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* ks = smem + buf * BUF;
    const char* vs = ks + KT_BYTES;
    const int key0 = kt * KB;

    // ---- S^T = K Q^T for the two 32-key sub-blocks ----
    f32x16 sacc[2];
#pragma unroll
    for (int sb = 0; sb < 2; sb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) sacc[sb][r] = 0.f;
      const int row = 32 * sb + krow_p;
#pragma unroll
      for (int ks_ = 0; ks_ < NKS; ks_++) {
        bf16x8 kf = *(const bf16x8*)(ks + lds_off<KROWB>(row, 2 * ks_ + h));
        sacc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks_], sacc[sb], 0, 0, 0);
      }
    }
    // ---- scale, bias, mask; online softmax ----
    // ONE wave-uniform branch per tile picks the variant (plain / ragged last tile / per-key bias): a per-element `if (key_bias || tail)` inside the
    // unrolled 32-score loop compiled to 96 scalar branches and 32 separately guarded loads per tile, i.e. 32 tiny basic blocks the scheduler could
    // not interleave with anything (r2: found in the .s; the common case is now 32 v_mul + max3 chains in one block)
    float p[2][16];
    const bool tail = (key0 + KB > S);
    float mt = -INFINITY;
    auto scores = [&](auto bias_c, auto tail_c) {
      constexpr bool BIAS = decltype(bias_c)::value, TAIL = decltype(tail_c)::value;
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float s = sacc[sb][r];
          if (BIAS || TAIL) {
            s *= scale2;
            const int key = key0 + 32 * sb + acc_row(r, h);
            if (BIAS) s += key_bias[(int64_t)b * S + min(key, S - 1)] * LOG2E;
            if (key >= S) s = -INFINITY;
          }
          p[sb][r] = s;                      // plain tiles keep the RAW score: the scale rides in the exponent's fma below
          mt = fmaxf(mt, s);
        }
    };
    float psc = 1.f;                         // factor still to be applied to p[][] inside exp2(p * psc - m)
    if (key_bias != nullptr) scores(std::true_type{}, std::true_type{});
    else if (tail) scores(std::false_type{}, std::true_type{});
    else { scores(std::false_type{}, std::false_type{}); mt *= scale2; psc = scale2; }      // scale2 > 0: max(scale2 * s) = scale2 * max(s)
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float ls = 0.f;
#pragma unroll
    for (int sb = 0; sb < 2; sb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        p[sb][r] = fast_exp2(fmaf(p[sb][r], psc, -m_new));
        ls += p[sb][r];
      }
    l_run = l_run * alpha + ls;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {        // wave-uniform: once the running max is stable the 16*NDT multiplies are skipped
#pragma unroll
      for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc_o[dt][r] *= alpha;
    }

    // ---- O^T += V^T P^T ----
    bf16x8 pf[2][2];
#pragma unroll
    for (int sb = 0; sb < 2; sb++)
#pragma unroll
      for (int m = 0; m < 2; m++) pf[sb][m] = pack8(&p[sb][8 * m]);
#pragma unroll
    for (int dt = 0; dt < NDT; dt++) {
      const int row = 32 * dt + l31;
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int m = 0; m < 2; m++) {
          bf16x8 vf = *(const bf16x8*)(vs + lds_off<128>(row, 4 * sb + 2 * m + h));
          acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sb][m], acc_o[dt], 0, 0, 0);
        }
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- finish: combine the two half-lanes' partial sums, normalise, store ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  const int q = q0 + l31;
  if (q < Sq) {
    bf16* orow = O + ((int64_t)b * Sq + q) * ld_o + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
      for (int a = 0; a < 4; a++) {
        bf16x4 o;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) o[bb] = f2bf(acc_o[dt][4 * a + bb] * inv);
        *(bf16x4*)(orow + 32 * dt + 8 * a + 4 * h) = o;
      }
    if (h == 0) lse2[bh * Sq + q] = m_run + __log2f(l_tot);
  }
}

// ------------------------------------------------------------------------------------------------
// second generation (the recipe that took the dK/dV kernel from 553 to 950 TFLOP/s): 8 waves x 32 queries = 256 queries per
// workgroup share one K / V^T tile (half the LDS fill per MFMA of the 4-wave kernel), the tiles arrive by LDS-DMA into a double
// buffer (no staging VGPRs, no ds_write pass), address arithmetic is re-derived from the lane id each tile (nothing hoisted,
// nothing spilled), and the O accumulators are only rescaled when some lane's running max actually moved.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void f_glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// STAG: the two waves of every SIMD (wave w and w+4) run half a key tile apart — one does QK^T + softmax (MFMA then VALU) while the other does
// PV (MFMA) — so the matrix pipe and the VALU are both busy; two barriers per key tile; every half ends by retiring the LDS-DMA pieces
// the wave issued in its PREVIOUS half (counted vmcnt), which is exactly what the other group's next half reads.
template <int HD, bool STAG>
__global__ void __launch_bounds__(512, 2) k_attn_fwd2(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                     const bf16* __restrict__ Vt, const float* __restrict__ key_bias,
                                                     bf16* __restrict__ O, int64_t ld_o, float* __restrict__ lse2, int H, int S, int Sp,
                                                     float scale2) {
  constexpr int KROWB = HD * 2;
  constexpr int KT_BYTES = KB * KROWB;
  constexpr int VT_BYTES = HD * 128;
  constexpr int BUF = KT_BYTES + VT_BYTES;
  constexpr int NKS = HD / 16, NDT = HD / 32;
  constexpr int KPW = KT_BYTES / 1024 / 8;     // K pieces per wave (HD=128: 2, HD=64: 1)
  constexpr int VPW = VT_BYTES / 1024 / 8;
  constexpr int RPP = 1024 / KROWB;            // K rows per piece
  constexpr int CPR = KROWB / 16;              // 16-byte chunks per K row
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = blockIdx.x * 256 + wv * 32;
  const int qi = min(q0 + l31, S - 1);
  const bf16* Kg = K + bh * (int64_t)S * HD;
  const bf16* Vg = Vt + bh * (int64_t)HD * Sp;

  bf16x8 qf[NKS];
  {
    const bf16* qrow = Q + (bh * S + qi) * (int64_t)HD + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
  }
  f32x16 acc_o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc_o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  auto stage_k = [&](int kt) {
    char* ks = smem + (kt & 1) * BUF;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int p = 0; p < KPW; p++) {
      const int row = (wv * KPW + p) * RPP + ln / CPR;
      const int col = ((ln % CPR) ^ (HD == 128 ? (row & 15) : ((row >> 1) & 7))) * 8;
      f_glds16(Kg + (uint32_t)(min(kt * KB + row, S - 1) * HD + col), ks + (wv * KPW + p) * 1024);
    }
  };
  auto stage_v = [&](int kt) {
    char* vs = smem + (kt & 1) * BUF + KT_BYTES;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int p = 0; p < VPW; p++) {
      const int row = (wv * VPW + p) * 8 + (ln >> 3);
      f_glds16(Vg + (uint32_t)(row * Sp + ((ln & 7) ^ ((row >> 1) & 7)) * 8 + kt * KB), vs + (wv * VPW + p) * 1024);
    }
  };
  auto wait_pieces = [&](int n) {                 // n pieces may stay in flight (n in {0, 1, 2})
    if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto half_end = [&](int in_flight) {
    wait_pieces(in_flight);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  const int nkt = (S + KB - 1) / KB;
  const int krow_p = perm23(l31);
  const int k_base0 = (HD == 128) ? krow_p * 256 + ((h ^ (krow_p & 15)) << 4) : krow_p * 128 + ((h ^ ((krow_p >> 1) & 7)) << 4);
  const int v_base0 = KT_BYTES + l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4);
  const bool late = STAG && wv >= 4;               // group 1 runs one half behind
  stage_k(0); stage_v(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (late) {                                      // its "PV half of tile -1": only the K(1) issue
    int fl = 0;
    if (1 < nkt) { stage_k(1); fl = KPW; }
    half_end(fl);
  }
  bf16x8 pf[2][2];
  float alpha = 1.f;
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    int k_base = k_base0 + buf * BUF, v_base = v_base0 + buf * BUF;
    asm volatile("" : "+v"(k_base), "+v"(v_base));
    const int key0 = kt * KB;
    // ================= half 1: S^T = K Q^T, online softmax =================
    int fl = 0;
    if (!STAG) { if (kt + 1 < nkt) { stage_k(kt + 1); stage_v(kt + 1); } }
    else if (!late) { if (kt + 1 < nkt) { stage_k(kt + 1); fl = KPW; } }
    else { if (kt + 1 < nkt) { stage_v(kt + 1); fl = VPW; } }
    f32x16 sacc[2];
#pragma unroll
    for (int sb = 0; sb < 2; sb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) sacc[sb][r] = 0.f;
#pragma unroll
      for (int ks_ = 0; ks_ < NKS; ks_++) {
        const bf16x8 kf = *(const bf16x8*)(smem + (k_base ^ ((2 * ks_) << 4)) + sb * 32 * KROWB);
        sacc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks_], sacc[sb], 0, 0, 0);
      }
    }
    const bool tail = (key0 + KB > S);
    float mt = -INFINITY;
#pragma unroll
    for (int sb = 0; sb < 2; sb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float sv = sacc[sb][r] * scale2;
        if (key_bias != nullptr || tail) {
          const int key = key0 + 32 * sb + acc_row(r, h);
          if (key_bias != nullptr) sv += key_bias[(int64_t)b * S + min(key, S - 1)] * LOG2E;
          if (key >= S) sv = -INFINITY;
        }
        sacc[sb][r] = sv;
        mt = fmaxf(mt, sv);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float ls = 0.f;
#pragma unroll
    for (int sb = 0; sb < 2; sb++)
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const float e = fast_exp2(sacc[sb][8 * m + r] - m_new);
          ls += e;
          pf[sb][m][r] = f2bf(e);
        }
    l_run = l_run * alpha + ls;
    if (STAG) half_end(fl);
    // ================= half 2: O^T += V^T P^T =================
    fl = 0;
    if (STAG) {
      if (!late) { if (kt + 1 < nkt) { stage_v(kt + 1); fl = VPW; } }
      else { if (kt + 2 < nkt) { stage_k(kt + 2); fl = KPW; } }
    }
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {        // wave-uniform: skip the 64 multiplies once the running max is stable
#pragma unroll
      for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc_o[dt][r] *= alpha;
    }
#pragma unroll
    for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int m = 0; m < 2; m++) {
          const bf16x8 vf = *(const bf16x8*)(smem + (v_base ^ ((4 * sb + 2 * m) << 4)) + dt * 4096);
          acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sb][m], acc_o[dt], 0, 0, 0);
        }
    }
    half_end(STAG ? fl : 0);
  }
  if (STAG && !late) {                             // pairs with group 1's extra half
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  const int q = q0 + l31;
  if (q < S) {
    bf16* orow = O + ((int64_t)b * S + q) * ld_o + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
      for (int a = 0; a < 4; a++) {
        bf16x4 o;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) o[bb] = f2bf(acc_o[dt][4 * a + bb] * inv);
        *(bf16x4*)(orow + 32 * dt + 8 * a + 4 * h) = o;
      }
    if (h == 0) lse2[bh * S + q] = m_run + __log2f(l_tot);
  }
}

static int attn_fwd_impl(void* stream, const void* Q, const void* K, const void* Vt, const float* key_bias, void* O,
                         int64_t ld_o, float* lse2, int B, int H, int Sq, int S, int Sp, int d, float scale) {
  ST_REQUIRE(Q && K && Vt && O && lse2, "attn_fwd: null pointer");
  ST_REQUIRE(B > 0 && H > 0 && S > 0 && Sq > 0 && Sp % 64 == 0 && Sp >= S && ld_o % 4 == 0, "attn_fwd: bad shape S=%d Sp=%d", S, Sp);
  if (d != 128 && d != 64 && d != 96) { st355_set_error("attn_fwd: head_dim %d not built", d); return ST355_ENOSYS; }
  const double flops = 4.0 * (double)B * H * (double)Sq * S * d;
  const double bytes = 2.0 * (double)B * H * (Sq + S) * d * 2.0;
  ProfScope ps(stream, ST355_K_ATTN_FWD, flops, bytes);
  const float scale2 = scale * LOG2E;
  static int gen = -1;
  // A/B switch.  Default = the first-generation kernel (4 waves x 32 queries, 2 independent workgroups per CU, register staging):
  // measured 761-786 TFLOP/s in-step.  ST355_ATTN_FWD=2 selects k_attn_fwd2 (8 waves, LDS-DMA, half-tile stagger): correct (same
  // parity tests) but 635-700 TFLOP/s — the forward is VALU/latency-shaped and loses the decoupling of two independent workgroups.
  if (gen < 0) { const char* e = getenv("ST355_ATTN_FWD"); gen = (e && e[0] == '2') ? 2 : ((e && e[0] == '3') ? 3 : 1); }
  if (gen == 2 && Sq == S && d != 96) {
    dim3 grid2((S + 255) / 256, H, B);
    if (d == 128) {
      const int lds = 2 * (KB * 256 + 128 * 128);
      static bool set = false;
      if (!set) { hipFuncSetAttribute((const void*)k_attn_fwd2<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set = true; }
      hipLaunchKernelGGL((k_attn_fwd2<128, true>), grid2, dim3(512), lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                         key_bias, (bf16*)O, ld_o, lse2, H, S, Sp, scale2);
    } else {
      const int lds = 2 * (KB * 128 + 64 * 128);
      hipLaunchKernelGGL((k_attn_fwd2<64, true>), grid2, dim3(512), lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                         key_bias, (bf16*)O, ld_o, lse2, H, S, Sp, scale2);
    }
    return st355_check_launch("attn_fwd2");
  }
  if (gen == 3 && (d == 128 || d == 64)) {
    dim3 grid3((Sq + 255) / 256, H, B);
    if (d == 128) {
      const int lds = 2 * (KB * 256 + 128 * 128);
      static bool set = false;
      if (!set) { hipFuncSetAttribute((const void*)k_attn_fwd<128, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set = true; }
      hipLaunchKernelGGL((k_attn_fwd<128, 8>), grid3, dim3(512), lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                         key_bias, (bf16*)O, ld_o, lse2, H, Sq, S, Sp, scale2);
    } else {
      const int lds = 2 * (KB * 128 + 64 * 128);
      hipLaunchKernelGGL((k_attn_fwd<64, 8>), grid3, dim3(512), lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                         key_bias, (bf16*)O, ld_o, lse2, H, Sq, S, Sp, scale2);
    }
    return st355_check_launch("attn_fwd3");
  }
  dim3 grid((Sq + QB - 1) / QB, H, B), block(ATT_THREADS);
  if (d == 96) {                 // PixArt's head_dim 72 zero-padded to 96 (3 d-tiles of 32, 6 k-steps of 16)
    const int lds = 2 * (KB * 192 + 96 * 128);
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)k_attn_fwd<96>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set = true; }
    hipLaunchKernelGGL(k_attn_fwd<96>, grid, block, lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                       key_bias, (bf16*)O, ld_o, lse2, H, Sq, S, Sp, scale2);
  } else if (d == 128) {
    const int lds = 2 * (KB * 256 + 128 * 128);
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)k_attn_fwd<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set = true; }
    hipLaunchKernelGGL(k_attn_fwd<128>, grid, block, lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                       key_bias, (bf16*)O, ld_o, lse2, H, Sq, S, Sp, scale2);
  } else {
    const int lds = 2 * (KB * 128 + 64 * 128);
    hipLaunchKernelGGL(k_attn_fwd<64>, grid, block, lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt,
                       key_bias, (bf16*)O, ld_o, lse2, H, Sq, S, Sp, scale2);
  }
  return st355_check_launch("attn_fwd");
}
extern "C" int st355_attn_fwd(void* stream, const void* Q, const void* K, const void* Vt, const float* key_bias, void* O,
                              int64_t ld_o, float* lse2, int B, int H, int S, int Sp, int d, float scale) {
  return attn_fwd_impl(stream, Q, K, Vt, key_bias, O, ld_o, lse2, B, H, S, S, Sp, d, scale);
}
// cross-attention (UNet attn2 over the 77 text tokens, PixArt cross-attention): Sq queries [B,H,Sq,d] against Sk keys [B,H,Sk,d], Vt [B,H,d,Skp];
// O: [B*Sq, ld_o] token-major, lse2 [B,H,Sq]; key_bias [B,Sk] additive or NULL
extern "C" int st355_attn_cross_fwd(void* stream, const void* Q, const void* K, const void* Vt, const float* key_bias, void* O,
                                    int64_t ld_o, float* lse2, int B, int H, int Sq, int Sk, int Skp, int d, float scale) {
  return attn_fwd_impl(stream, Q, K, Vt, key_bias, O, ld_o, lse2, B, H, Sq, Sk, Skp, d, scale);
}
