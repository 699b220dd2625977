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

// ------------------------------------------------------------------------------------------------
// Second generation (r02, the default): the generation-1 shape (4 waves x 32 queries, two independent workgroups per CU, register staging) with the per-tile
// VALU stream cut down and software-pipelined against the MFMAs INSIDE a wave.  Found in the generation-1 .s: per tile and wave 32 MFMAs (1024
// cycles) sat next to ~195 VALU instructions + 33 v_exp — of which 32 were v_mov phi copies (the three score variants merged into one p[] array),
// one ds_bpermute + s_waitcnt lgkmcnt(0) for the half-wave max (which also drains every V^T fragment read in flight), and the PV loop ran dt-outer
// so all 32 exponentials had to retire before the 2nd..4th MFMA.  Here:
//   * BIAS is a kernel template parameter and the ragged last tile is peeled (tile<TAIL>): the steady-state tile is straight-line code;
//   * scores stay in the MFMA accumulator registers (masked variants rewrite them in place), no phi copies;
//   * half-wave exchange by v_permlane32_swap (VALU, no LDS round trip, no lgkmcnt drain);
//   * PV runs (sb, m)-outer / dt-inner: the 8 exponentials + pack of k-slice (sb, m) are issued just before its NDT MFMAs, so the exponentials of
//     slice g+1 execute under the MFMAs of slice g (independent accumulators acc_o[0..NDT-1] back to back, no dependent-MFMA stalls).
// ------------------------------------------------------------------------------------------------

// VROW (head_dim 128): V comes ROW-major (token rows of a [B*S, ld_v] projection buffer, head h at columns h*128, e.g. the V third of a fused QKV
// output) instead of the pre-transposed head-major V^T copy: the V tile is staged as [64 keys][128 channels] with the swz_q chunk swizzle of the
// backward kernels and the V^T fragments of O^T += V^T P^T are gathered by transposing LDS reads (two ds_read_b64_tr_b16 per fragment, same LDS
// bytes as the one ds_read_b128 of the V^T form) — no V^T buffer, no transposing pass over V anywhere (Vt is then Vrows, Sp is ld_v).
__device__ __forceinline__ int swz_kv(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }      // = swz_q of attention_bwd.hip

template <int HD, bool BIAS, bool VROW = false>
__global__ void __launch_bounds__(256, 2) k_attn_fwd4(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                      const bf16* __restrict__ Vt, const float* __restrict__ key_bias,
                                                      bf16* __restrict__ O, int64_t ld_o, float* __restrict__ lse2, int H,
                                                      int Sq, int S, int Sp, float scale2, bf16* __restrict__ Ores = nullptr) {
  static_assert(!VROW || HD == 128, "row-major V is built for head_dim 128");
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
  const bf16* Vg = VROW ? Vt + (int64_t)b * S * Sp + (int64_t)head * HD      // VROW: Sp carries ld_v
                        : Vt + bh * (int64_t)HD * Sp;

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
  // Staging addresses: a K tile is KT_BYTES of CONTIGUOUS memory (rows key0..key0+63 of one head), so thread t's p-th 16-byte chunk sits at
  // tile base + (p * ATT_T + t) * 16; V^T chunks at a per-thread 32-bit offset from a wave-uniform base.  Uniform base (SGPR) + 32-bit lane offset
  // is the saddr form of global_load: no per-load 64-bit VALU arithmetic (generation 1 spent ~20 VALU per tile on it).  Only the ragged last tile
  // clamps rows (K rows past S would read the next head, or past the allocation for the last one).
  const uint32_t koff0 = (uint32_t)tid * 16u;
  const uint32_t voff0 = VROW ? ((uint32_t)(tid >> 4) * (uint32_t)Sp + (uint32_t)(tid & 15) * 8u) * 2u      // VROW: 16 chunks per key row, 16 rows per pass
                              : ((uint32_t)(tid >> 3) * (uint32_t)Sp + (uint32_t)(tid & 7) * 8u) * 2u;
  auto load_tile = [&](int kt) {
    const int key0 = kt * KB;
    const char* kb = (const char*)Kg + (size_t)key0 * KROWB;
    if (key0 + KB <= S) {
#pragma unroll
      for (int p = 0; p < KCH; p++) kreg[p] = *(const bf16x8*)(kb + (koff0 + (uint32_t)(p * ATT_T * 16)));
    } else {
#pragma unroll
      for (int p = 0; p < KCH; p++) {
        const int id = p * ATT_T + tid;
        const int row = id / (HD / 8), c = id % (HD / 8);
        kreg[p] = *(const bf16x8*)(Kg + (int64_t)min(key0 + row, S - 1) * HD + c * 8);
      }
    }
    if (VROW) {
      const char* vb = (const char*)Vg + (size_t)key0 * Sp * 2;
      if (key0 + KB <= S) {
#pragma unroll
        for (int p = 0; p < VCH; p++) vreg[p] = *(const bf16x8*)(vb + (size_t)p * (ATT_T / 16) * Sp * 2 + voff0);
      } else {                 // ragged last tile: clamp the key row (its P column is exactly 0, the value only has to be finite)
#pragma unroll
        for (int p = 0; p < VCH; p++) {
          const int row = p * (ATT_T / 16) + (tid >> 4);
          vreg[p] = *(const bf16x8*)(Vg + (int64_t)min(key0 + row, S - 1) * Sp + (tid & 15) * 8);
        }
      }
    } else {
      const char* vb = (const char*)Vg + (size_t)key0 * 2;
#pragma unroll
      for (int p = 0; p < VCH; p++) vreg[p] = *(const bf16x8*)(vb + (size_t)p * (ATT_T / 8) * Sp * 2 + voff0);
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
      if (VROW) {
        const int row = id >> 4, c = id & 15;
        *(bf16x8*)(vs + row * 256 + ((c ^ swz_kv(row)) << 4)) = vreg[p];
      } else {
        const int row = id >> 3, c = id & 7;
        *(bf16x8*)(vs + lds_off<128>(row, c)) = vreg[p];
      }
    }
  };

  const int nkt = (S + KB - 1) / KB;
  const int krow_p = perm23(l31);
  // VROW: transposed-fragment base (attention_bwd.hip dkv3): lane = 16 g + 4 r + s reads key row 8h + r (+4 for the second read), channels
  // 16 (g & 1) + 4 s .. + 3 of the 32-channel d tile: byte = key * 256 + ((chunk ^ f(key)) << 4) + (s & 1) * 8, chunk = 4 dt + 2 (g & 1) + (s >> 1)
  const int tr_r = (lane >> 2) & 3, tr_s = lane & 3, tr_ih = (lane >> 4) & 1;
  const int tr_base = (8 * h + tr_r) * 256 + ((((2 * tr_ih + (tr_s >> 1)) ^ (4 * tr_r + 2 * h))) << 4) + (tr_s & 1) * 8;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  auto tile = [&](int kt, auto masked_c) {
    constexpr bool MASKED = decltype(masked_c)::value;      // ragged last tile and/or per-key bias: scores are rewritten in place, already scaled
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* ks = smem + buf * BUF;
    const char* vs = ks + KT_BYTES;
    const int key0 = kt * KB;

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
      for (int r = 0; r < 16; r++) {
        if (MASKED) {
          float s = sacc[sb][r] * scale2;
          const int key = key0 + 32 * sb + acc_row(r, h);
          if (BIAS) s += key_bias[(int64_t)b * S + min(key, S - 1)] * LOG2E;
          if (key >= S) s = -INFINITY;
          sacc[sb][r] = s;
        }
        mt = fmaxf(mt, sacc[sb][r]);
      }
    const float psc = MASKED ? 1.f : scale2;               // plain tiles keep the RAW score: the scale rides in the exponent's fma
    if (!MASKED) mt *= scale2;                             // scale2 > 0: max(scale2 * s) = scale2 * max(s)
    mt = xhalf_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // wave-uniform: once the running max is stable the 16*NDT multiplies are skipped
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
          e[r] = fast_exp2(fmaf(sacc[sb][8 * m + r], psc, -m_new));
          ls[r & 1] += e[r];
        }
        const bf16x8 pf = pack8(e);
#pragma unroll
        for (int dt = 0; dt < NDT; dt++) {
          bf16x8 vf;
          if (VROW) {     // V^T[d = 32 dt + l31][keys 32 sb + 16 m + 8 h + 0..7]: two transposing reads (keys +0..3, +4..7)
            const char* r0 = vs + (tr_base ^ ((4 * dt) << 4)) + (32 * sb + 16 * m) * 256;
            const char* r1 = vs + (tr_base ^ (((4 * dt) ^ 1) << 4)) + (32 * sb + 16 * m + 4) * 256;
            vf = lds_tr16x2(r0, r1);
          } else {
            vf = *(const bf16x8*)(vs + lds_off<128>(32 * dt + l31, 4 * sb + 2 * m + h));
          }
          acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, acc_o[dt], 0, 0, 0);
        }
      }
    l_run = l_run * alpha + (ls[0] + ls[1]);
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  };
  const int nfull = BIAS ? 0 : S / KB;                     // tiles with all 64 keys valid and no bias take the plain path
  for (int kt = 0; kt < nfull; kt++) tile(kt, std::false_type{});
  for (int kt = nfull; kt < nkt; kt++) tile(kt, std::true_type{});

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
        if (Ores) {     // the rounding residual of O (st355_attn_fwd_res): O + Ores carries 16 mantissa bits of the fp32 output for the backward's delta = rowsum(dO * O)
          bf16x4 r;
#pragma unroll
          for (int bb = 0; bb < 4; bb++) r[bb] = f2bf(acc_o[dt][4 * a + bb] * inv - bf2f(o[bb]));
          *(bf16x4*)(Ores + (orow - O) + 32 * dt + 8 * a + 4 * h) = r;
        }
      }
    if (h == 0) lse2[bh * Sq + q] = m_run + __log2f(l_tot);
  }
}


// ------------------------------------------------------------------------------------------------
// Third generation (r04, head_dim 128, no bias, S % 64 == 0): 4 waves x 64 queries — ONE wave per SIMD with the whole 512-register file, every K / V^T fragment
// feeding TWO MFMAs (half the LDS operand traffic per MFMA of k_attn_fwd4), O^T and the Q fragments in AGPRs.  hipcc's allocator does not manage that budget
// (512 VGPRs + 180 spills), so the main loop is ONE asm statement with an asm-owned register map generated by tools/kgen/fwd64.py (pipeline,
// stale-reference softmax and wait-state rules are documented there); HIP code computes the lane addresses in front of it and stores O / lse2 behind it
// (O leaves through the idle LDS ring as whole 256-byte token rows).
// ------------------------------------------------------------------------------------------------
#define ST355_FWD64_OPERANDS                                                                                                                           \
        : [lse0] "=&v"(lse0), [lse1] "=&v"(lse1)                                                                                                   \
        : [qp0] "v"(qp0), [qp1] "v"(qp1), [koff] "v"(koff), [voff] "v"(voff), [rowb] "v"(rowb), [vtb] "v"(vtb), [park] "v"(park), [kbase] "s"(kbase), \
          [vbase] "s"(vbase), [lds] "s"(lds), [wvoff] "s"(wvoff), [nkt] "s"(nkt), [vrow32] "s"(vrow32), [scale2] "s"(scale2), [thr] "s"(thr), [blk0] "s"(blk0), \
          [tracelo] "s"(tracelo), [tracehi] "s"(tracehi)                                                                                            \
        : "memory", "vcc", "scc",

// HD = 128 (Flux), 96 (PixArt-Sigma's head_dim 72, zero padded: 6 k-steps, 3 d tiles; the K tile image keeps the 256-byte row pitch, the V^T image holds 96 rows)
// (head_dim 64 — SDXL / SD3 / SD 1.x — keeps k_attn_fwd4: a generated 64-row body exists, `FWD64_HD=64 python -m tools.kgen.fwd64`, but with half the MFMAs per tile under the
// same softmax work it is issue-bound at one wave per SIMD: 653 vs 740 TFLOP/s at B4 H10 S4096, 433 vs 515 at B4 H20 S1024 — profiles/r04_attn_lab_hd64_64_row_kernels.log)
template <int HD>
__global__ void __launch_bounds__(256, 1) k_attn_fwd64(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vt, bf16* __restrict__ O,
                                                       int64_t ld_o, float* __restrict__ lse2, int H, int Sq, int S, int Sp, float scale2,
                                                       unsigned long long* trace) {
  static_assert(HD == 128 || HD == 96, "k_attn_fwd64: head_dim 128 or 96");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = wg.tile * 256 + wv * 64;
  float lse0, lse1;
  {
    const int qi0 = min(q0 + l31, Sq - 1), qi1 = min(q0 + 32 + l31, Sq - 1);
    const bf16* qp0 = Q + (bh * Sq + qi0) * (int64_t)HD + 8 * h;
    const bf16* qp1 = Q + (bh * Sq + qi1) * (int64_t)HD + 8 * h;
    // LDS-DMA lane offsets of the wave's first K piece (rows 4 wave + lane / 16, chunk swizzle swz_kv) and first V^T piece (channel rows 8 wave + lane / 8)
    const int krow = wv * 4 + (lane >> 4);
    int kchunk = (lane & 15) ^ swz_kv(krow);
    if (kchunk >= HD / 8) kchunk = 0;                    // (head_dim 96: slots of source chunks >= 12 are never read; their lanes re-fetch chunk 0)
    const uint32_t koff = (uint32_t)(krow * HD + kchunk * 8) * 2u;
    const int vrow = wv * 8 + (lane >> 3);
    const uint32_t voff = (uint32_t)(vrow * Sp + ((lane & 7) ^ ((vrow >> 1) & 7)) * 8) * 2u;
    const uint32_t vrow32 = (uint32_t)(32 * Sp * 2);
    const int rowp = perm23(l31);
    const uint32_t rowb = rowp * 256 + ((h ^ swz_kv(rowp)) << 4);                     // K row fragments: chunk (2 ks + h) ^ f(row)
    const uint32_t vtb = l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4);                   // V^T fragments: row 32 dt + l31, chunk (4 sb + 2 m + h) ^ ((row >> 1) & 7)
    const uint32_t park = l31 * 256 + ((l31 & 15) << 4) + 8 * h;
    const bf16* kbase = K + bh * (int64_t)S * HD;
    const bf16* vbase = Vt + bh * (int64_t)HD * Sp;
    const uint32_t lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t wvoff = (uint32_t)wv * 1024u;
    const uint32_t nkt = (uint32_t)(S / 64);
    const uint32_t blk0 = blockIdx.x | blockIdx.y | blockIdx.z;
    const uint32_t tracelo = (uint32_t)(uintptr_t)trace, tracehi = (uint32_t)((uintptr_t)trace >> 32);
    const float thr = 8.0f / scale2;                    // the stale-reference bound (2^8 on the exponentials) in the accumulators' raw-score units
    if constexpr (HD == 128) {
      asm volatile(
#ifdef ST355_FWD64_BODY_INC       // tools/attn_lab builds: a generator variant under test
#include ST355_FWD64_BODY_INC
#else
#include "gen/attn_fwd64_body.inc"
#endif
          ST355_FWD64_OPERANDS
#include "gen/attn_fwd64_clobbers.inc"
      );
    } else {
      asm volatile(
#ifdef ST355_FWD64_HD96_BODY_INC
#include ST355_FWD64_HD96_BODY_INC
#else
#include "gen/attn_fwd64_hd96_body.inc"
#endif
          ST355_FWD64_OPERANDS
#include "gen/attn_fwd64_clobbers.inc"
      );
    }
  }
  const char* mine = smem + wv * 16384;
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    const int tok0 = q0 + 32 * qb;
    const int tl = lane >> 4, c = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int t = it * 4 + tl;
      if (tok0 + t < Sq && c < HD / 8)
        *(bf16x8*)(O + ((int64_t)b * Sq + tok0 + t) * ld_o + (int64_t)head * HD + c * 8) = *(const bf16x8*)(mine + qb * 8192 + t * 256 + ((c ^ (t & 15)) << 4));
    }
    const int q = tok0 + l31;
    if (h == 0 && q < Sq) lse2[bh * Sq + q] = qb ? lse1 : lse0;
  }
}

// forward kernel choice where k_attn_fwd64 applies: 64 (default) or 32 = k_attn_fwd4.  ST355_ATTN_FWD64=0 overrides; tools/attn_lab sets the variable directly.
int g_attn_fwd_impl = -1;
unsigned long long* g_attn_fwd_trace = nullptr;    // tools/attn_lab, trace builds of the fwd64 body only
static int attn_fwd_impl64() {
  if (g_attn_fwd_impl < 0) { const char* e = getenv("ST355_ATTN_FWD64"); g_attn_fwd_impl = (e && e[0] == '0') ? 32 : 64; }
  return g_attn_fwd_impl;
}

// vrow != 0: Vt is the ROW-major V (token rows, head h at columns h*d) and Sp its leading dimension (k_attn_fwd4<128, *, true>)
static int attn_fwd_impl(void* stream, const void* Q, const void* K, const void* Vt, const float* key_bias, void* O,
                         int64_t ld_o, float* lse2, int B, int H, int Sq, int S, int Sp, int d, float scale, int vrow = 0, void* O_res = nullptr) {
  ST_REQUIRE(Q && K && Vt && O && lse2, "attn_fwd: null pointer");
  ST_REQUIRE(B > 0 && H > 0 && S > 0 && Sq > 0 && ld_o % 4 == 0 && (vrow ? (Sp % 8 == 0 && Sp >= H * d && d == 128) : (Sp % 64 == 0 && Sp >= S)),
             "attn_fwd: bad shape S=%d Sp(ld_v)=%d", S, Sp);
  if (d != 128 && d != 64 && d != 96) { st355_set_error("attn_fwd: head_dim %d not built", d); return ST355_ENOSYS; }
  const double flops = 4.0 * (double)B * H * (double)Sq * S * d;
  const double bytes = 2.0 * (double)B * H * (Sq + S) * d * 2.0;
  ProfScope ps(stream, ST355_K_ATTN_FWD, flops, bytes);
  const float scale2 = scale * LOG2E;
  // k_attn_fwd4 (r02) is the general kernel; k_attn_fwd64 (r04) takes the head_dim-128, no-bias, S % 64 == 0 shapes.  The r01 kernel lives on in
  // tools/attn_fwd_variants.hip as the lab's A/B baseline (r02 lab, B8 H24 S4608 d128: 862 -> 927 TFLOP/s).  Measured and deleted in r02: an 8-wave /
  // 256-query workgroup variant (843 TFLOP/s) and an 8-wave LDS-DMA half-tile-stagger variant (738); logs under profiles/archive/r02_attn_lab_*.log.
  if (!vrow && !key_bias && !O_res && (d == 128 || d == 96) && S % 64 == 0 && attn_fwd_impl64() == 64) {      // hand-scheduled 64-queries-per-wave kernel
    dim3 grid64((Sq + 255) / 256, H, B);
    const int lds64 = 4 * 2 * 64 * 256;
#define ST355_FWD64_LAUNCH(HD_)                                                                                                           \
  do {                                                                                                                                   \
    static St355AttrOnce set64;                                                                                                           \
    if (set64.need()) { hipFuncSetAttribute((const void*)k_attn_fwd64<HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds64); } \
    hipLaunchKernelGGL(k_attn_fwd64<HD_>, grid64, dim3(256), lds64, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt, (bf16*)O, ld_o, lse2, \
                       H, Sq, S, Sp, scale2, g_attn_fwd_trace);                                                                          \
  } while (0)
    if (d == 128) ST355_FWD64_LAUNCH(128);
    else ST355_FWD64_LAUNCH(96);
#undef ST355_FWD64_LAUNCH
    return st355_check_launch("attn_fwd64");
  }
  dim3 grid((Sq + QB - 1) / QB, H, B), block(ATT_THREADS);
  const int lds = 2 * (KB * d * 2 + d * 128);
#define ST355_FWD_LAUNCH(KERN)                                                                                                           \
  do {                                                                                                                                   \
    static St355AttrOnce set;                                                                                                             \
    if (set.need()) { hipFuncSetAttribute((const void*)(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }                 \
    hipLaunchKernelGGL((KERN), grid, block, lds, (hipStream_t)stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt, key_bias,         \
                       (bf16*)O, ld_o, lse2, H, Sq, S, Sp, scale2, (bf16*)O_res);                                                       \
  } while (0)
  if (vrow) {
    if (key_bias) ST355_FWD_LAUNCH((k_attn_fwd4<128, true, true>));
    else ST355_FWD_LAUNCH((k_attn_fwd4<128, false, true>));
  } else if (key_bias) {
    if (d == 128) ST355_FWD_LAUNCH((k_attn_fwd4<128, true>));
    else if (d == 96) ST355_FWD_LAUNCH((k_attn_fwd4<96, true>));          // PixArt's head_dim 72 zero-padded to 96 (3 d-tiles of 32, 6 k-steps of 16)
    else ST355_FWD_LAUNCH((k_attn_fwd4<64, true>));
  } else {
    if (d == 128) ST355_FWD_LAUNCH((k_attn_fwd4<128, false>));
    else if (d == 96) ST355_FWD_LAUNCH((k_attn_fwd4<96, false>));
    else ST355_FWD_LAUNCH((k_attn_fwd4<64, false>));
  }
#undef ST355_FWD_LAUNCH
  return st355_check_launch("attn_fwd");
}
extern "C" int st355_attn_fwd(void* stream, const void* Q, const void* K, const void* Vt, const float* key_bias, void* O,
                              int64_t ld_o, float* lse2, int B, int H, int S, int Sp, int d, float scale) {
  return attn_fwd_impl(stream, Q, K, Vt, key_bias, O, ld_o, lse2, B, H, S, S, Sp, d, scale);
}
extern "C" int st355_attn_fwd_vrows(void* stream, const void* Q, const void* K, const void* v_rows, int64_t ld_v, const float* key_bias, void* O,
                                    int64_t ld_o, float* lse2, int B, int H, int S, int d, float scale) {
  ST_REQUIRE(ld_v > 0 && ld_v < ((int64_t)1 << 31), "attn_fwd_vrows: bad ld_v");
  return attn_fwd_impl(stream, Q, K, v_rows, key_bias, O, ld_o, lse2, B, H, S, S, (int)ld_v, d, scale, 1);
}
// st355_attn_fwd / st355_attn_cross_fwd that ALSO write the rounding residual of the output, O_res = bf16(O_fp32 - O) in O's layout (same ld_o): the backward's
// delta = rowsum(dO * O) is then taken from O + O_res (st355_attn_bwd_res).  With O alone every dS row carries the error P_ij * dO_i.(O_fp32 - O)_i, which the
// dQ = dS K and dK = dS^T Q products multiply by the COMMON component of K / Q over tokens — exact arithmetic cancels that component (sum_j dS_ij = 0); measured on the
// SDXL UNet's 32^2 self-attention (LayerNorm outputs with |mean_j x_j| / rms |x_j| = 0.97): dQ rel-L2 0.26 -> profiles/r05_sdxl_lora_outlier_probe.log
extern "C" int st355_attn_fwd_res(void* stream, const void* Q, const void* K, const void* Vt, const float* key_bias, void* O, int64_t ld_o, void* O_res,
                                  float* lse2, int B, int H, int Sq, int Sk, int Skp, int d, float scale) {
  ST_REQUIRE(O_res, "attn_fwd_res: null residual pointer");
  return attn_fwd_impl(stream, Q, K, Vt, key_bias, O, ld_o, lse2, B, H, Sq, Sk, Skp, d, scale, 0, O_res);
}
// cross-attention (UNet attn2 over the 77 text tokens, PixArt cross-attention): Sq queries [B,H,Sq,d] against Sk keys [B,H,Sk,d], Vt [B,H,d,Skp];
// O: [B*Sq, ld_o] token-major, lse2 [B,H,Sq]; key_bias [B,Sk] additive or NULL
extern "C" int st355_attn_cross_fwd(void* stream, const void* Q, const void* K, const void* Vt, const float* key_bias, void* O,
                                    int64_t ld_o, float* lse2, int B, int H, int Sq, int Sk, int Skp, int d, float scale) {
  return attn_fwd_impl(stream, Q, K, Vt, key_bias, O, ld_o, lse2, B, H, Sq, Sk, Skp, d, scale);
}
