// attn_fwd_variants.hip — LAB-ONLY forward attention variants (included by tools/attn_lab.hip after the product sources; never part of libst355).
//
// k_attn_fwd4_stale<128>: the product's k_attn_fwd4<128, false, false> with ONE change — the running maximum that the exponentials are taken against is allowed to
// go STALE.  The product re-references every tile: m_new = max(m_run, tile max), alpha = 2^(m_run - m_new), O *= alpha, l *= alpha (it already skips the 32 packed
// multiplies of O when no row maximum of the wave moved).  Here a row is re-referenced only when its tile maximum exceeds the reference by more than STALE_BOUND
// (log2 units), wave-uniformly: between re-references p = 2^(s - m_ref) may reach 2^STALE_BOUND instead of 1 — representable in bf16 at the same relative precision,
// and numerator (O) and denominator (l) share the reference, so the normalised output and lse2 = m_ref + log2(l) are the same quantities.
// Why: tools/isa_mix.py counts 1016 vector-ALU cycles against 1024 matrix-pipe cycles per wave and tile in the product loop (profiles/archive/r03_isa_mix_hot_loops.md);
// the accumulator rescale is 256 of them, the alpha / l_run bookkeeping a few more; with trained (peaked) score rows some row of the 32 moves in almost every tile,
// so the exact-equality skip of the product rarely fires in the early tiles.
// STATUS: written in round 3 after the GPU budget was spent — compiled for gfx950, NOT yet run.  tools/attn_lab times it and compares its O / lse2 with the product kernel.
// ---- the first-generation forward (r01), moved here from attention.hip in r04: the lab's A/B baseline only ----
template <int HD>
__global__ void __launch_bounds__(256, 2) k_attn_fwd(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                            const bf16* __restrict__ Vt, const float* __restrict__ key_bias,
                                                            bf16* __restrict__ O, int64_t ld_o, float* __restrict__ lse2, int H,
                                                            int Sq, int S, int Sp, float scale2) {   // Sq queries; S keys (padded Sp)
  constexpr int KROWB = HD * 2;          // bytes per K tile row
  constexpr int KT_BYTES = KB * KROWB;   // K tile
  constexpr int VT_BYTES = HD * 128;     // V^T tile: HD rows x 64 keys
  constexpr int BUF = KT_BYTES + VT_BYTES;
  constexpr int NKS = HD / 16;           // MFMA k-steps over the head dim
  constexpr int NDT = HD / 32;           // 32-row d tiles of O^T
  constexpr int NW = 4;
  constexpr int ATT_T = 64 * NW;
  constexpr int KCH = KT_BYTES / 16 / ATT_T;  // 16-B chunks per thread
  constexpr int VCH = VT_BYTES / 16 / ATT_T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = wg.tile * (32 * NW) + wv * 32;
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


#define STALE_BOUND 8.0f

template <int HD>
__global__ void __launch_bounds__(256, 2) k_attn_fwd4_stale(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vt,
                                                            bf16* __restrict__ O, int64_t ld_o, float* __restrict__ lse2, int H, int Sq, int S, int Sp,
                                                            float scale2) {
  static_assert(HD == 128, "lab variant: head_dim 128, S a multiple of 64, no bias");
  constexpr int NW = 4;
  constexpr int KROWB = HD * 2;
  constexpr int KT_BYTES = KB * KROWB;
  constexpr int VT_BYTES = HD * 128;
  constexpr int BUF = KT_BYTES + VT_BYTES;
  constexpr int NKS = HD / 16;
  constexpr int NDT = HD / 32;
  constexpr int ATT_T = 64 * NW;
  constexpr int KCH = KT_BYTES / 16 / ATT_T;
  constexpr int VCH = VT_BYTES / 16 / ATT_T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = wg.tile * (32 * NW) + wv * 32;
  const int qi = min(q0 + l31, Sq - 1);
  const bf16* Kg = K + bh * (int64_t)S * HD;
  const bf16* Vg = Vt + bh * (int64_t)HD * Sp;

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
  float m_ref = -INFINITY, l_run = 0.f;

  bf16x8 kreg[KCH], vreg[VCH];
  const uint32_t koff0 = (uint32_t)tid * 16u;
  const uint32_t voff0 = ((uint32_t)(tid >> 3) * (uint32_t)Sp + (uint32_t)(tid & 7) * 8u) * 2u;
  auto load_tile = [&](int kt) {
    const int key0 = kt * KB;
    const char* kb = (const char*)Kg + (size_t)key0 * KROWB;
#pragma unroll
    for (int p = 0; p < KCH; p++) kreg[p] = *(const bf16x8*)(kb + (koff0 + (uint32_t)(p * ATT_T * 16)));
    const char* vb = (const char*)Vg + (size_t)key0 * 2;
#pragma unroll
    for (int p = 0; p < VCH; p++) vreg[p] = *(const bf16x8*)(vb + (size_t)p * (ATT_T / 8) * Sp * 2 + voff0);
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

  const int nkt = S / KB;
  const int krow_p = perm23(l31);
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* ks = smem + buf * BUF;
    const char* vs = ks + KT_BYTES;
    f32x16 sacc[2];
    float mt = -INFINITY;
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
#pragma unroll
    for (int sb = 0; sb < 2; sb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mt = fmaxf(mt, sacc[sb][r]);
    mt = xhalf_max(mt * scale2);
    // re-reference only when some row of the wave ran away from its reference by more than the bound (first tile: m_ref = -inf, always)
    if (__builtin_amdgcn_ballot_w64(mt > m_ref + STALE_BOUND) != 0) {
      const float m_new = fmaxf(m_ref, mt);
      const float alpha = fast_exp2(m_ref - m_new);
      m_ref = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc_o[dt][r] *= alpha;
    }
    float ls[2] = {0.f, 0.f};
#pragma unroll
    for (int sb = 0; sb < 2; sb++)
#pragma unroll
      for (int m = 0; m < 2; m++) {
        float e[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          e[r] = fast_exp2(fmaf(sacc[sb][8 * m + r], scale2, -m_ref));
          ls[r & 1] += e[r];
        }
        const bf16x8 pf = pack8(e);
#pragma unroll
        for (int dt = 0; dt < NDT; dt++) {
          const bf16x8 vf = *(const bf16x8*)(vs + lds_off<128>(32 * dt + l31, 4 * sb + 2 * m + h));
          acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, acc_o[dt], 0, 0, 0);
        }
      }
    l_run += ls[0] + ls[1];
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  const float l_tot = xhalf_sum(l_run);
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
    if (h == 0) lse2[bh * Sq + q] = m_ref + __log2f(l_tot);
  }
}

// |a - b| statistics of two bf16 buffers (the variant's rounding differs from the product's: tolerance, not bit equality)
__global__ void k_absdiff(const bf16* a, const bf16* b, int64_t n, float* maxd, double* sumsq_d, double* sumsq_ref) {
  float md = 0.f; double sd = 0.0, sr = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = (float)a[i], y = (float)b[i];
    md = fmaxf(md, fabsf(x - y)); sd += (double)(x - y) * (x - y); sr += (double)y * y;
  }
  atomicMax((int*)maxd, __float_as_int(md)); atomicAdd(sumsq_d, sd); atomicAdd(sumsq_ref, sr);
}
