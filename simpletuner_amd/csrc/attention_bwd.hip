// attention_bwd.hip — K7 backward (what autograd does through F.scaled_dot_product_attention in the reference,
// trainer.py:7126).  Three launches, no atomics, deterministic:
//   prep : delta[q] = sum_d O[q,d] dO[q,d]   and   dO^T (head-major, transposed, zero padded)
//   dKdV : per 128-key workgroup, loop over 64-query tiles:  P, dS recomputed;  dV^T += dO^T P ; dK^T += Q^T dS
//   dQ   : per 256-query workgroup, loop over 64-key tiles:                        dQ^T += K^T dS^T
// Both main kernels reuse the forward's register-resident softmax trick (attn_common.h): the tile whose rows
// become the next contraction index is fetched in perm23 order, so P / dS go straight from accumulator
// registers into the next MFMA's B operand.  Operands whose contraction index is the token axis (Q^T, K^T, dO^T)
// are read from pre-transposed head-major copies (written by the QKV epilogue kernel / prep), never transposed
// in LDS.
#include <stdlib.h>
#include <type_traits>
#include "attn_common.h"

#define TPB 130

// tile-image swizzle of the kernels that ALSO gather transposed fragments from a row-major tile (dq<.., TR>, dkv3): 16-byte chunk c of row q is stored
// at c ^ f(q), f(q) = 4 (q & 3) + ((q >> 2) & 3) — a bijection of q & 15 (ds_read_b128 row fragments stay conflict-free) under which 4 consecutive
// rows differ in bits 2-3 (the 32 eight-byte pieces of a ds_read_b64_tr_b16 lane group fall on 32 distinct bank slots)
__device__ __forceinline__ int swz_q(int q) { return ((q & 3) << 2) | ((q >> 2) & 3); }

// ------------------------------------------------------------------------------------------------
// prep
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(256) k_attn_bwd_prep(const bf16* __restrict__ O, int64_t ld_o, const bf16* __restrict__ dO,
                                                      int64_t ld_do, const float* __restrict__ lse2, float* __restrict__ delta,
                                                      float* __restrict__ lsep, bf16* __restrict__ dOt, int H, int S, int Sp, float lse_mul,
                                                      const bf16* __restrict__ Ores = nullptr) {
  constexpr int TPR = (HD / 8 <= 8) ? 8 : 16;          // lanes per token (power of two; head_dim 96 leaves 4 of 16 idle)
  constexpr int TOK_PER_PASS = 256 / TPR;
  __shared__ __attribute__((aligned(16))) bf16 tile[64 * TPB];
  const int tid = threadIdx.x;
  const int head = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * 64;
  const int c = tid % TPR;
  const bool cact = c < HD / 8;
  const int64_t bh = (int64_t)b * H + head;
  for (int tl = tid / TPR; tl < 64; tl += TOK_PER_PASS) {
    const int t = t0 + tl;
    const bool valid = t < S;
    const int tt = valid ? t : S - 1;
    bf16x8 ov, gv;
#pragma unroll
    for (int j = 0; j < 8; j++) { ov[j] = f2bf(0.f); gv[j] = f2bf(0.f); }
    if (cact) {
      ov = *(const bf16x8*)(O + ((int64_t)b * S + tt) * ld_o + (int64_t)head * HD + c * 8);
      gv = *(const bf16x8*)(dO + ((int64_t)b * S + tt) * ld_do + (int64_t)head * HD + c * 8);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) s += bf2f(ov[j]) * bf2f(gv[j]);
    if (Ores && cact) {          // + dO . (O_fp32 - O): delta from the un-rounded attention output (st355_attn_bwd_res)
      const bf16x8 rv = *(const bf16x8*)(Ores + ((int64_t)b * S + tt) * ld_o + (int64_t)head * HD + c * 8);
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) s2 += bf2f(rv[j]) * bf2f(gv[j]);
      s += s2;
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (c == 0) {   // stats are written in the padded [B*H, Sp] layout: padded queries get delta 0 and lse +inf (=> P = 0)
      delta[bh * Sp + t] = valid ? s : 0.f;
      lsep[bh * Sp + t] = valid ? lse2[bh * S + t] * lse_mul : INFINITY;      // lse_mul = 1 / scale2 for k_attn_bwd_dkv4 (its score chains start from it), else 1
    }
    if (!valid) {
#pragma unroll
      for (int j = 0; j < 8; j++) gv[j] = f2bf(0.f);
    }
    if (cact && dOt != nullptr) {
      uint32_t* tp = (uint32_t*)(&tile[tl * TPB + c * 8]);
      const u32x4 gw = *(const u32x4*)&gv;
#pragma unroll
      for (int j = 0; j < 4; j++) tp[j] = gw[j];
    }
  }
  if (dOt == nullptr) return;                         // dkv3 (head_dim 128) gathers dO^T fragments by transposing LDS reads: no transposed copy
  __syncthreads();
  for (int i = tid; i < HD * 8; i += 256) {
    const int d = i >> 3, tc = i & 7;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = tile[(tc * 8 + e) * TPB + d];
    *(bf16x8*)(dOt + (bh * HD + d) * (int64_t)Sp + t0 + tc * 8) = o;  // rows are 128-B aligned (Sp % 64 == 0)
  }
}

// ------------------------------------------------------------------------------------------------
// Fused RoPE + RMSNorm backward in the dQ / dK epilogues (st355_attn_bwd_rope, head_dim 128).  The accumulator layout gives a lane 64 of the 128
// channels of ONE token (register 4a + b <-> channel 32 dt + 8a + 4h + b): whole rotation pairs (2i, 2i+1) sit in one lane, the 128-channel RMSNorm
// reduction is an in-lane sum + one half-wave exchange.  So the gradient w.r.t. the roped head-major Q / K never goes to HBM: the epilogue un-rotates it
// (dy = R^T g), un-rotates the kept roped activation (y = R^T z), applies dx = r (w dy - (y / w) mean(dy y)) and writes the PROJECTION-gradient rows
// dqkv[(b S + pos), part D + head 128 + c] directly — the standalone pass (k_qk_rope_norm_bwd_z) read dQ, dK, Q, K and wrote dqkv once more.
// Two norm-weight sets: joint positions < split use w_lo (the text stream's norm_added_q / _k), the rest w_hi (single blocks: split = 0).
// ------------------------------------------------------------------------------------------------
struct RopeBwd {
  const float* rrms;                     // [B*S, 2H] 1/rms from the fused projection epilogue
  const bf16* w_lo; const bf16* w_hi;    // RMSNorm weights [128] (both NULL: no norm)
  const float* cos_p; const float* sin_p;   // [S, 64] one angle per rotation pair
  bf16* out; int64_t ldo;                // projection-gradient rows (already offset to the q or k column block); out == NULL: plain head-major store
  int split, H, S;
  int rr_off;                            // 0 for q heads, H for k heads inside a rrms row
};
// Layout change first: the accumulator layout gives a lane 8-byte pieces of one token at a 256-byte token pitch — read Q / cos / sin / w and write the
// output that way and every wave instruction touches 64 separate cache lines for 8 useful bytes each (measured: +47 us per workgroup, slower than the pass it
// replaced).  So the wave parks its 32 tokens x 128 channels (already scaled, rounded to bf16 exactly like the head-major dQ / dK of the unfused path) in
// its own 8 KiB slice of the idle LDS ring, XOR-swizzled by token, and reads it back with 16 lanes per token x 16 bytes: all global traffic is then whole
// 256-byte token rows, the RMSNorm reduction a 16-lane xor-shuffle tree.
// second half of rope_bwd_store: the wave's 32 tokens x 128 channels are parked in `stage` (bf16, token rows of 256 bytes, 16-byte chunk c of token t at
// c ^ (t & 15)); read them back 16 lanes per token and finish (k_attn_bwd_dq64 parks from its hand-scheduled body and calls this directly)
__device__ __forceinline__ void rope_bwd_finish(const RopeBwd& rp, const bf16* zhead, int b, int head, int tok0, int ntok, int lane, const char* stage) {
  const int tl = lane >> 4, c = lane & 15;                                // read side: token tl of each group of 4, 16-byte chunk c (channels 8c .. 8c+7)
  float wlo[8], whi[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { wlo[j] = rp.w_lo ? bf2f(rp.w_lo[c * 8 + j]) : 1.f; whi[j] = rp.w_hi ? bf2f(rp.w_hi[c * 8 + j]) : 1.f; }
  const bool has_norm = rp.w_hi != nullptr;
#pragma unroll 2
  for (int it = 0; it < 8; it++) {
    const int t = it * 4 + tl;
    const int tok = min(tok0 + t, ntok - 1);
    const bf16x8 gv = *(const bf16x8*)(stage + t * 256 + ((c ^ (t & 15)) << 4));
    const bf16x8 zv = *(const bf16x8*)(zhead + (int64_t)tok * 128 + c * 8);
    const f32x4 cs = *(const f32x4*)(rp.cos_p + (int64_t)tok * 64 + c * 4);
    const f32x4 sn = *(const f32x4*)(rp.sin_p + (int64_t)tok * 64 + c * 4);
    float dy[8], y[8], sdy = 0.f;
#pragma unroll
    for (int pr = 0; pr < 4; pr++) {
      const float g0 = bf2f(gv[2 * pr]), g1 = bf2f(gv[2 * pr + 1]), z0 = bf2f(zv[2 * pr]), z1 = bf2f(zv[2 * pr + 1]);
      // forward: o0 = y0 c - y1 s ; o1 = y1 c + y0 s   =>   R^T v = (v0 c + v1 s, v1 c - v0 s)
      dy[2 * pr] = g0 * cs[pr] + g1 * sn[pr]; dy[2 * pr + 1] = g1 * cs[pr] - g0 * sn[pr];
      y[2 * pr] = z0 * cs[pr] + z1 * sn[pr]; y[2 * pr + 1] = z1 * cs[pr] - z0 * sn[pr];
      sdy += dy[2 * pr] * y[2 * pr] + dy[2 * pr + 1] * y[2 * pr + 1];
    }
    bf16x8 o;
    if (has_norm) {
      sdy += __shfl_xor(sdy, 8, 64); sdy += __shfl_xor(sdy, 4, 64); sdy += __shfl_xor(sdy, 2, 64); sdy += __shfl_xor(sdy, 1, 64);
      const float m = sdy * (1.f / 128.f);
      const float r = rp.rrms[((int64_t)b * rp.S + tok) * (2 * rp.H) + rp.rr_off + head];
      const bool lo = tok < rp.split;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float wj = lo ? wlo[j] : whi[j];
        o[j] = f2bf(r * (wj * dy[j] - y[j] * __builtin_amdgcn_rcpf(wj) * m));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = f2bf(dy[j]);
    }
    if (tok0 + t < ntok) *(bf16x8*)(rp.out + ((int64_t)b * rp.S + tok) * rp.ldo + (int64_t)head * 128 + c * 8) = o;
  }
}
__device__ __forceinline__ void rope_bwd_store(const RopeBwd& rp, const f32x16 (&acc)[4], float scale, const bf16* zhead, int b, int head, int tok0, int ntok,
                                               int lane, char* stage) {   // zhead: roped head-major activations of (b, head); tok0: the wave's first token
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int a = 0; a < 4; a++) {
      bf16x4 o;
#pragma unroll
      for (int bb = 0; bb < 4; bb++) o[bb] = f2bf(acc[dt][4 * a + bb] * scale);
      const int ch16 = 4 * dt + a;                                        // 16-byte chunk of the token row; h picks its 8-byte half
      *(bf16x4*)(stage + l31 * 256 + ((ch16 ^ (l31 & 15)) << 4) + 8 * h) = o;
    }
  rope_bwd_finish(rp, zhead, b, head, tok0, ntok, lane, stage);
}

// ------------------------------------------------------------------------------------------------
// dQ kernel: 8 waves x 32 queries
// ------------------------------------------------------------------------------------------------
// TR (head_dim 128): no K^T tile — the K^T fragments of dQ^T += K^T dS^T are gathered from the row-major K tile by transposing LDS reads (the
// dkv3 recipe: tile image swizzled with f(row)), so the pre-transposed head-major K^T copy in HBM disappears and a key tile is 32 KiB, not 48.
// BIAS is a kernel template parameter and the ragged last key tile is peeled (tile<MASKED>), as in the forward: the steady-state tile is straight-line
// code (no merge copies of the score registers between variants, and the scheduler may run the dS arithmetic of one 32-key block under the MFMAs of
// the next).
template <int HD, bool TR = false, bool BIAS = false>
__global__ void __launch_bounds__(512, 2) k_attn_bwd_dq(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                       const bf16* __restrict__ Kt, const bf16* __restrict__ Vrows, int64_t ld_v,
                                                       const bf16* __restrict__ dO, int64_t ld_do, const float* __restrict__ lse2,
                                                       const float* __restrict__ delta, const float* __restrict__ key_bias,
                                                       bf16* __restrict__ dQ, int H, int Sq, int Sqp, int Sk, int Skp, float scale, float scale2,
                                                       RopeBwd rp, int kt0 = 0, int accumulate = 0) {
  // kt0 / accumulate: the RAGGED-TAIL form behind k_attn_bwd_dq64 — the backward is separable over keys once lse2 and delta are known
  // (dQ_i = sum_j P_ij (dP_ij - delta_i) K_j), so the 64-row kernel takes the full 64-key tiles [0, kt0) and this kernel adds the contribution of
  // the keys [64 kt0, Sk) to the dQ it left behind (one extra rounding of the sum to bf16)
  constexpr int NT = 512;
  constexpr int KROWB = HD * 2;
  constexpr int KT_BYTES = 64 * KROWB;   // K tile and V tile (row-major, 64 keys)
  constexpr int TT_BYTES = HD * 128;     // K^T tile (HD rows x 64 keys)
  // TR: the K tile image keeps a 256-byte row pitch at every head_dim (16 chunk slots per key row, HD / 8 of them used): the swz_q chunk swizzle and the
  // transposing-read lane layout of dkv3 were derived for that pitch; head_dim 64 / 96 leave slots empty instead of needing a second swizzle
  constexpr int KI_BYTES = TR ? 64 * 256 : KT_BYTES;
  constexpr int BUF = KI_BYTES + KT_BYTES + (TR ? 0 : TT_BYTES);
  constexpr int NKS = HD / 16, NDT = HD / 32;
  constexpr int KTOT = KT_BYTES / 16, TTOT = TT_BYTES / 16;             // 16-byte chunks per tile (head_dim 96: 768, not a multiple of 512)
  constexpr int KCH = (KTOT + NT - 1) / NT;
  constexpr int TCH = TR ? 1 : (TTOT + NT - 1) / NT;
  auto koff = [](int row, int chunk) { return TR ? row * 256 + ((chunk ^ swz_q(row)) << 4) : lds_off<KROWB>(row, chunk); };   // K tile image
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = wg.tile * 256 + wv * 32;
  const int q = q0 + l31;
  const int qi = min(q, Sq - 1);

  const bf16* Kg = K + bh * (int64_t)Sk * HD;
  const bf16* Ktg = Kt + bh * (int64_t)HD * Skp;
  const bf16* Vg = Vrows + (int64_t)b * Sk * ld_v + (int64_t)head * HD;

  bf16x8 qf[NKS], dof[NKS];
  {
    const bf16* qrow = Q + (bh * Sq + qi) * (int64_t)HD + 8 * h;
    const bf16* drow = dO + ((int64_t)b * Sq + qi) * ld_do + (int64_t)head * HD + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
      dof[ks] = *(const bf16x8*)(drow + 16 * ks);
    }
  }
  const float lse_q = lse2[bh * Sq + qi];
  const float delta_q = delta[bh * (int64_t)Sqp + qi];

  f32x16 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[dt][r] = 0.f;

  bf16x8 kreg[KCH], vreg[KCH], treg[TCH];
  auto load_tile = [&](int kt) {
    const int key0 = kt * 64;
#pragma unroll
    for (int p = 0; p < KCH; p++) {
      const int id = p * NT + tid;
      if (KTOT % NT != 0 && id >= KTOT) continue;
      const int row = id / (HD / 8), c = id % (HD / 8);
      const int key = min(key0 + row, Sk - 1);
      kreg[p] = *(const bf16x8*)(Kg + (int64_t)key * HD + c * 8);
      vreg[p] = *(const bf16x8*)(Vg + (int64_t)key * ld_v + c * 8);
    }
    if (!TR) {
#pragma unroll
      for (int p = 0; p < TCH; p++) {
        const int id = p * NT + tid;
        if (TTOT % NT != 0 && id >= TTOT) continue;
        const int row = id >> 3, c = id & 7;
        treg[p] = *(const bf16x8*)(Ktg + (int64_t)row * Skp + key0 + c * 8);
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* ks = smem + buf * BUF;
    char* vs = ks + KI_BYTES;
    char* ts = vs + KT_BYTES;
#pragma unroll
    for (int p = 0; p < KCH; p++) {
      const int id = p * NT + tid;
      if (KTOT % NT != 0 && id >= KTOT) continue;
      const int row = id / (HD / 8), c = id % (HD / 8);
      *(bf16x8*)(ks + koff(row, c)) = kreg[p];
      *(bf16x8*)(vs + lds_off<KROWB>(row, c)) = vreg[p];
    }
    if (!TR) {
#pragma unroll
      for (int p = 0; p < TCH; p++) {
        const int id = p * NT + tid;
        if (TTOT % NT != 0 && id >= TTOT) continue;
        const int row = id >> 3, c = id & 7;
        *(bf16x8*)(ts + lds_off<128>(row, c)) = treg[p];
      }
    }
  };

  const int nkt = (Sk + 63) / 64;
  const int krow_p = perm23(l31);
  // TR: transposed-fragment base (see dkv3): lane = 16 g + 4 r + s, key row 8h + r, chunk (2 ihalf + (s >> 1)) ^ (4 r + 2 h), + (s & 1) * 8
  const int tr_r = (lane >> 2) & 3, tr_s = lane & 3, tr_ih = (lane >> 4) & 1;
  const int tr_base = (8 * h + tr_r) * 256 + ((((2 * tr_ih + (tr_s >> 1)) ^ (4 * tr_r + 2 * h))) << 4) + (tr_s & 1) * 8;
  load_tile(kt0);
  store_tile(kt0 & 1);
  __syncthreads();
  auto tile = [&](int kt, auto masked_c) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* ks = smem + buf * BUF;
    const char* vs = ks + KI_BYTES;
    const char* ts = vs + KT_BYTES;
    const int key0 = kt * 64;
#pragma unroll
    for (int sb = 0; sb < 2; sb++) {
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; r++) { sacc[r] = 0.f; dpacc[r] = 0.f; }
      const int row = 32 * sb + krow_p;
#pragma unroll
      for (int ks_ = 0; ks_ < NKS; ks_++) {
        bf16x8 kf = *(const bf16x8*)(ks + koff(row, 2 * ks_ + h));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks_], sacc, 0, 0, 0);
        bf16x8 vf = *(const bf16x8*)(vs + lds_off<KROWB>(row, 2 * ks_ + h));
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks_], dpacc, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float pv;
        if (MASKED) {
          float s = sacc[r] * scale2;
          const int key = key0 + 32 * sb + acc_row(r, h);
          if (BIAS) s += key_bias[(int64_t)b * Sk + min(key, Sk - 1)] * LOG2E;
          pv = (key < Sk) ? fast_exp2(s - lse_q) : 0.f;
        } else {
          pv = fast_exp2(fmaf(sacc[r], scale2, -lse_q));
        }
        ds[r] = pv * (dpacc[r] - delta_q);
      }
      bf16x8 dsf[2];
      dsf[0] = pack8(&ds[0]);
      dsf[1] = pack8(&ds[8]);
#pragma unroll
      for (int dt = 0; dt < NDT; dt++) {
        const int trow = 32 * dt + l31;
#pragma unroll
        for (int m = 0; m < 2; m++) {
          bf16x8 tf;
          if (TR) {       // K^T[d = 32 dt + l31][keys 32 sb + 16 m + 8 h + 0..7] from the row-major K tile: two transposing reads (keys +0..3, +4..7)
            const char* r0 = ks + (tr_base ^ ((4 * dt) << 4)) + (32 * sb + 16 * m) * 256;
            const char* r1 = ks + (tr_base ^ (((4 * dt) ^ 1) << 4)) + (32 * sb + 16 * m + 4) * 256;
            tf = lds_tr16x2(r0, r1);
          } else {
            tf = *(const bf16x8*)(ts + lds_off<128>(trow, 4 * sb + 2 * m + h));
          }
          acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, dsf[m], acc[dt], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  };
  const int nfull = BIAS ? 0 : Sk / 64;                    // key tiles with all 64 keys valid and no bias take the plain path
  for (int kt = kt0; kt < nfull; kt++) tile(kt, std::false_type{});
  for (int kt = max(kt0, nfull); kt < nkt; kt++) tile(kt, std::true_type{});
  if (HD == 128 && rp.out != nullptr) {          // fused RoPE + RMSNorm backward: straight to the projection-gradient rows (all lanes take part in the exchange)
    // (the key-tile loop ended on a workgroup barrier: the LDS ring is idle, each wave takes its own 8 KiB slice)
    if constexpr (HD == 128) rope_bwd_store(rp, acc, scale, Q + bh * (int64_t)Sq * HD, b, head, q0, Sq, lane, smem + wv * 8192);
    return;
  }
  if (q < Sq) {
    bf16* orow = dQ + (bh * Sq + q) * (int64_t)HD;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
      for (int a = 0; a < 4; a++) {
        bf16x4 o;
        if (accumulate) {
          const bf16x4 prev = *(const bf16x4*)(orow + 32 * dt + 8 * a + 4 * h);
#pragma unroll
          for (int bb = 0; bb < 4; bb++) o[bb] = f2bf(fmaf(acc[dt][4 * a + bb], scale, bf2f(prev[bb])));
        } else {
#pragma unroll
          for (int bb = 0; bb < 4; bb++) o[bb] = f2bf(acc[dt][4 * a + bb] * scale);
        }
        *(bf16x4*)(orow + 32 * dt + 8 * a + 4 * h) = o;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// dK/dV kernel, second generation: 8 waves x 32 keys (256 keys per workgroup, two waves per SIMD so one wave's softmax VALU and
// LDS reads overlap the other's MFMAs), the four 16-KiB query-tile images (Q, dO row-major; Q^T, dO^T) and the 64 lse/delta
// pairs arrive by LDS-DMA (global_load_lds, swizzle applied to the SOURCE address) into a double buffer: no staging VGPRs and no
// ds_write pass (the first-generation kernel spent ~830 LDS cycles per tile on ds_write_b128 alone).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void a_glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void a_glds4(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

template <int HD>
__global__ void __launch_bounds__(512, 2) k_attn_bwd_dkv2(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                         const bf16* __restrict__ Qt, const bf16* __restrict__ Vrows, int64_t ld_v,
                                                         const bf16* __restrict__ dO, int64_t ld_do, const bf16* __restrict__ dOt,
                                                         const float* __restrict__ lsep, const float* __restrict__ delta,
                                                         const float* __restrict__ key_bias, bf16* __restrict__ dK,
                                                         bf16* __restrict__ dVrows, int64_t ld_dv, int H, int Sq, int Sqp, int Sk, int Skp, float scale,
                                                         float scale2) {
  constexpr int QROWB = HD * 2;
  constexpr int QT_BYTES = 64 * QROWB;   // Q tile / dO tile (64 queries, row-major)
  constexpr int TT_BYTES = HD * 128;     // Q^T tile / dO^T tile (HD rows x 64 queries)
  constexpr int STAT_BYTES = 2 * 64 * 4;
  constexpr int BUF = 2 * QT_BYTES + 2 * TT_BYTES + STAT_BYTES;
  constexpr int NKS = HD / 16, NDT = HD / 32;
  constexpr int QPIECES = QT_BYTES / 1024, TPIECES = TT_BYTES / 1024;   // 1-KiB DMA pieces per tile (HD=96: 12, spread over 8 waves as 2+1)
  constexpr int QPW = (QPIECES + 7) / 8;     // pieces per wave per row-major tile (HD=128: 2)
  constexpr int TPW = (TPIECES + 7) / 8;     // ... per transposed tile
  constexpr int CPR = QROWB / 16;            // 16-byte chunks per row (HD=128: 16, 96: 12, 64: 8)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  const int key = wg.tile * 256 + wv * 32 + l31;
  const int keyi = min(key, Sk - 1);

  const bf16* Qg = Q + bh * (int64_t)Sq * HD;
  const bf16* Qtg = Qt + bh * (int64_t)HD * Sqp;
  const bf16* dOtg = dOt + bh * (int64_t)HD * Sqp;
  const bf16* dOg = dO + (int64_t)b * Sq * ld_do + (int64_t)head * HD;
  const float* lse_g = lsep + bh * (int64_t)Sqp;
  const float* del_g = delta + bh * (int64_t)Sqp;

  bf16x8 kf[NKS], vf[NKS];
  {
    const bf16* krow = K + (bh * Sk + keyi) * (int64_t)HD + 8 * h;
    const bf16* vrow = Vrows + ((int64_t)b * Sk + keyi) * ld_v + (int64_t)head * HD + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      kf[ks] = *(const bf16x8*)(krow + 16 * ks);
      vf[ks] = *(const bf16x8*)(vrow + 16 * ks);
    }
  }
  const float kb2 = key_bias ? key_bias[(int64_t)b * Sk + keyi] * LOG2E : 0.f;

  f32x16 acc_dk[NDT], acc_dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) { acc_dk[dt][r] = 0.f; acc_dv[dt][r] = 0.f; }

  // DMA sources are rebuilt per tile from the lane id (a handful of VALU ops per piece): nothing address-like stays live
  // across the MFMA work, which needs every register it can get.
  auto stage = [&](int qt, int buf) {
    const int qq0 = qt * 64;
    char* qs = smem + buf * BUF;
    char* gs = qs + QT_BYTES;
    char* qts = gs + QT_BYTES;
    char* gts = qts + TT_BYTES;
    char* stat = gts + TT_BYTES;
    int ln = lane;
    asm volatile("" : "+v"(ln));     // everything below is re-derived from the lane id each tile (never spilled, never hoisted)
#pragma unroll
    for (int p = 0; p < QPW; p++) {
      const int piece = wv + 8 * p;                                // wave-uniform
      if (QPIECES % 8 != 0 && piece >= QPIECES) continue;
      const int idx = piece * 64 + ln;                             // linear 16-byte chunk index inside the tile image
      const int row = idx / CPR;                                   // tile row = query
      const int swz = HD == 128 ? (row & 15) : (HD == 96 ? ((row >> 2) & 3) : ((row >> 1) & 7));
      const int col = ((idx % CPR) ^ swz) * 8;
      const int qq = min(qq0 + row, Sq - 1);
      a_glds16(Qg + (uint32_t)(qq * HD + col), qs + piece * 1024);
      a_glds16(dOg + ((int64_t)qq * ld_do + col), gs + piece * 1024);
    }
#pragma unroll
    for (int p = 0; p < TPW; p++) {
      const int piece = wv + 8 * p;
      if (TPIECES % 8 != 0 && piece >= TPIECES) continue;
      const int row = piece * 8 + (ln >> 3);                       // tile row = head channel
      const uint32_t off = (uint32_t)(row * Sqp + ((ln & 7) ^ ((row >> 1) & 7)) * 8 + qq0);
      a_glds16(Qtg + off, qts + piece * 1024);
      a_glds16(dOtg + off, gts + piece * 1024);
    }
    if (wv == 0) a_glds4(lse_g + qq0 + ln, stat);
    if (wv == 1) a_glds4(del_g + qq0 + ln, stat + 256);
  };

  const int nqt = (Sq + 63) / 64;
  // per-lane LDS read bases; every fragment address is base ^ (chunk_pair << 4) + an immediate (the XOR swizzles act on bit 0
  // = lane half h, folded into the base, and on bits 1.. = the k-step, applied per read)
  const int qrow_p = perm23(l31);
  // head_dim 96: a 192-byte row pitch puts row bits into the chunk field (bits 6-7), so the chunk cannot be XOR-ed into the address: the
  // (2-bit) swizzle is XOR-ed into the chunk NUMBER and the chunk offset is added
  const int hs96 = h ^ ((qrow_p >> 2) & 3);
  const int q_base0 = (HD == 128) ? qrow_p * 256 + ((h ^ (qrow_p & 15)) << 4)
                      : (HD == 96 ? qrow_p * 192 : qrow_p * 128 + ((h ^ ((qrow_p >> 1) & 7)) << 4));
  const int t_base0 = QT_BYTES * 2 + l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4);
  const int s_base0 = 2 * QT_BYTES + 2 * TT_BYTES + 32 * h;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int qt = 0; qt < nqt; qt++) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) stage(qt + 1, buf ^ 1);      // the other buffer's last reads finished before the previous barrier
    // laundered through an empty asm so that the address arithmetic below is NOT hoisted out of the loop into (spilled) registers
    int q_base = q_base0 + buf * BUF, t_base = t_base0 + buf * BUF, s_base = s_base0 + buf * BUF;
    asm volatile("" : "+v"(q_base), "+v"(t_base), "+v"(s_base));
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
      // Register budget (<= 256 with two waves per SIMD): 128 dK/dV accumulators + 64 K/V fragments are fixed, so LDS fragments
      // are consumed in pairs (sched_barrier pins the order; the partner wave on the SIMD covers the LDS latency).
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; r++) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
      for (int k2 = 0; k2 < NKS; k2 += 2) {
        const char* q0p = (HD == 96) ? smem + q_base + (((2 * k2) ^ hs96) << 4) + qb * 32 * QROWB : smem + (q_base ^ ((2 * k2) << 4)) + qb * 32 * QROWB;
        const char* q1p = (HD == 96) ? smem + q_base + (((2 * k2 + 2) ^ hs96) << 4) + qb * 32 * QROWB
                                     : smem + (q_base ^ ((2 * k2 + 2) << 4)) + qb * 32 * QROWB;
        bf16x8 qf0 = *(const bf16x8*)(q0p);
        bf16x8 qf1 = *(const bf16x8*)(q1p);
        bf16x8 gf0 = *(const bf16x8*)(q0p + QT_BYTES);
        bf16x8 gf1 = *(const bf16x8*)(q1p + QT_BYTES);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf0, kf[k2], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf0, vf[k2], dpacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf1, kf[k2 + 1], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf1, vf[k2 + 1], dpacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // accumulator register r <-> query 32qb + 16(r>>3) + 8h + (r&7): stats are contiguous 4-float runs
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int m = 0; m < 2; m++) {
#pragma unroll
        for (int q4 = 0; q4 < 2; q4++) {
          const float* sp = (const float*)(smem + s_base) + 32 * qb + 16 * m + 4 * q4;
          const f32x4 lse4 = *(const f32x4*)sp;
          const f32x4 del4 = *(const f32x4*)(sp + 64);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int ri = 8 * m + 4 * q4 + r;
            const float pr = fast_exp2(sacc[ri] * scale2 + kb2 - lse4[r]);
            const float dsv = pr * (dpacc[ri] - del4[r]);
            pf[m][4 * q4 + r] = f2bf(pr);
            dsf[m][4 * q4 + r] = f2bf(dsv);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int dt = 0; dt < NDT; dt++) {
        const char* t0p = smem + (t_base ^ ((4 * qb) << 4)) + dt * 4096;
        const char* t1p = smem + (t_base ^ ((4 * qb + 2) << 4)) + dt * 4096;
        bf16x8 qtf0 = *(const bf16x8*)(t0p);
        bf16x8 gtf0 = *(const bf16x8*)(t0p + TT_BYTES);
        bf16x8 qtf1 = *(const bf16x8*)(t1p);
        bf16x8 gtf1 = *(const bf16x8*)(t1p + TT_BYTES);
        acc_dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf0, pf[0], acc_dv[dt], 0, 0, 0);
        acc_dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf0, dsf[0], acc_dk[dt], 0, 0, 0);
        acc_dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf1, pf[1], acc_dv[dt], 0, 0, 0);
        acc_dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf1, dsf[1], acc_dk[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // next tile landed (it was issued a whole tile of MFMA work ago) + every wave done reading this buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (key < Sk) {
    bf16* krow = dK + (bh * Sk + key) * (int64_t)HD;
    bf16* vrow = dVrows + ((int64_t)b * Sk + key) * ld_dv + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
      for (int a = 0; a < 4; a++) {
        bf16x4 ok, ov;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
          ok[bb] = f2bf(acc_dk[dt][4 * a + bb] * scale);
          ov[bb] = f2bf(acc_dv[dt][4 * a + bb]);
        }
        *(bf16x4*)(krow + 32 * dt + 8 * a + 4 * h) = ok;
        *(bf16x4*)(vrow + 32 * dt + 8 * a + 4 * h) = ov;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// dK/dV kernel, third generation (head_dim 128): the dkv2 schedule WITHOUT the pre-transposed operand copies.  dK^T += Q^T dS and dV^T += dO^T P
// contract over the QUERIES of the tile; dkv2 read those A operands from separate head-major transposed copies (Q^T from the QKV epilogue
// kernel, dO^T from the prep kernel), i.e. every query tile came in twice — 64 KiB of LDS-DMA per tile (8 one-KiB pieces per wave at 60-185 issue
// cycles each, and 1500 cycles of LDS fill at the measured 43.7 B/clk), next to 4096 MFMA cycles per SIMD.  Here the k-fragments are gathered from
// the SAME row-major Q / dO tile images that feed S = Q K^T and dP = dO V^T, by the transposing LDS read (two ds_read_b64_tr_b16 per fragment: a
// 16-lane group fetches a [4 queries][16 channels] block, lane j receives channel j of the 4 queries).  Half the DMA pieces, half the fill, no Q^T /
// dO^T buffers in HBM at all (the QKV epilogue kernel and the prep kernel stop writing them).
// Tile image: row = query (256 B), 16-byte chunk c stored at c ^ f(row), f = swz_q.
// ------------------------------------------------------------------------------------------------
// head_dim 64 / 96 (r3): same kernel, same 256-byte row pitch of the tile images (16 chunk slots per query row, HD / 8 used — the DMA lanes of the empty
// slots are masked off), NKS / NDT from the head_dim; the fused RoPE epilogue stays head_dim 128 only.
template <int HD>
__global__ void __launch_bounds__(512, 2) k_attn_bwd_dkv3(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                         const bf16* __restrict__ Vrows, int64_t ld_v,
                                                         const bf16* __restrict__ dO, int64_t ld_do,
                                                         const float* __restrict__ lsep, const float* __restrict__ delta,
                                                         const float* __restrict__ key_bias, bf16* __restrict__ dK,
                                                         bf16* __restrict__ dVrows, int64_t ld_dv, int H, int Sq, int Sqp, int Sk, float scale,
                                                         float scale2, RopeBwd rp) {
  constexpr int QROWB = 256;             // row pitch of the tile images (= the head_dim 128 row; narrower heads leave chunk slots empty)
  constexpr int QT_BYTES = 64 * QROWB;   // Q tile / dO tile (64 queries, row-major): 16 KiB each
  constexpr int STAT_BYTES = 2 * 64 * 4;
  constexpr int BUF = 2 * QT_BYTES + STAT_BYTES;
  constexpr int NKS = HD / 16, NDT = HD / 32;
  constexpr int QPW = QT_BYTES / 1024 / 8;   // 1-KiB DMA pieces per wave per tile image (2)
  constexpr int CPR = QROWB / 16;            // 16-byte chunks per row (16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  const int key = wg.tile * 256 + wv * 32 + l31;
  const int keyi = min(key, Sk - 1);

  const bf16* Qg = Q + bh * (int64_t)Sq * HD;
  const bf16* dOg = dO + (int64_t)b * Sq * ld_do + (int64_t)head * HD;
  const float* lse_g = lsep + bh * (int64_t)Sqp;
  const float* del_g = delta + bh * (int64_t)Sqp;

  bf16x8 kf[NKS], vf[NKS];
  {
    const bf16* krow = K + (bh * Sk + keyi) * (int64_t)HD + 8 * h;
    const bf16* vrow = Vrows + ((int64_t)b * Sk + keyi) * ld_v + (int64_t)head * HD + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      kf[ks] = *(const bf16x8*)(krow + 16 * ks);
      vf[ks] = *(const bf16x8*)(vrow + 16 * ks);
    }
  }
  const float kb2 = key_bias ? key_bias[(int64_t)b * Sk + keyi] * LOG2E : 0.f;

  f32x16 acc_dk[NDT], acc_dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) { acc_dk[dt][r] = 0.f; acc_dv[dt][r] = 0.f; }

  auto stage = [&](int qt, int buf) {
    const int qq0 = qt * 64;
    char* qs = smem + buf * BUF;
    char* gs = qs + QT_BYTES;
    char* stat = gs + QT_BYTES;
    int ln = lane;
    asm volatile("" : "+v"(ln));     // everything below is re-derived from the lane id each tile (never spilled, never hoisted)
#pragma unroll
    for (int p = 0; p < QPW; p++) {
      const int piece = wv + 8 * p;                                // wave-uniform
      const int idx = piece * 64 + ln;                             // linear 16-byte chunk index inside the tile image
      const int row = idx / CPR;                                   // tile row = query
      const int col = ((idx % CPR) ^ swz_q(row)) * 8;              // LDS chunk position c holds source chunk c ^ f(row)
      const int qq = min(qq0 + row, Sq - 1);
      if (HD == 128 || col < HD) {                                 // (head_dim 64 / 96: the slots of source chunks >= HD / 8 stay unwritten and unread)
        a_glds16(Qg + (uint32_t)(qq * HD + col), qs + piece * 1024);
        a_glds16(dOg + ((int64_t)qq * ld_do + col), gs + piece * 1024);
      }
    }
    if (wv == 0) a_glds4(lse_g + qq0 + ln, stat);
    if (wv == 1) a_glds4(del_g + qq0 + ln, stat + 256);
  };

  const int nqt = (Sq + 63) / 64;
  // row fragments (S and dP phases): row = perm23 order, chunk (2 ks + h) ^ f(row)
  const int qrow_p = perm23(l31);
  const int q_base0 = qrow_p * 256 + ((h ^ swz_q(qrow_p)) << 4);
  // transposed fragments (dV / dK phases): lane = 16 g + 4 r + s;  ihalf = g & 1 picks the 16-channel half of the 32-channel d tile, the lane's
  // address is query 8h + r (+ 4 for the second read, + 16 m + 32 qb) of the tile, channels 16 ihalf + 4 s .. + 3 of d tile dt:
  //   byte = query * 256 + ((chunk ^ f(query)) << 4) + (s & 1) * 8,  chunk = 4 dt + 2 ihalf + (s >> 1),  f(query) = 4 r + 2 h + jsel
  const int tr_r = (lane >> 2) & 3, tr_s = lane & 3, tr_ih = (lane >> 4) & 1;
  const int t_base0 = (8 * h + tr_r) * 256 + ((((2 * tr_ih + (tr_s >> 1)) ^ (4 * tr_r + 2 * h))) << 4) + (tr_s & 1) * 8;
  const int s_base0 = 2 * QT_BYTES + 32 * h;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int qt = 0; qt < nqt; qt++) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) stage(qt + 1, buf ^ 1);      // the other buffer's last reads finished before the previous barrier
    int q_base = q_base0 + buf * BUF, t_base = t_base0 + buf * BUF, s_base = s_base0 + buf * BUF;
    asm volatile("" : "+v"(q_base), "+v"(t_base), "+v"(s_base));
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; r++) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
      for (int k2 = 0; k2 < NKS; k2 += 2) {
        const char* q0p = smem + (q_base ^ ((2 * k2) << 4)) + qb * 32 * QROWB;
        const char* q1p = smem + (q_base ^ ((2 * k2 + 2) << 4)) + qb * 32 * QROWB;
        bf16x8 qf0 = *(const bf16x8*)(q0p);
        bf16x8 qf1 = *(const bf16x8*)(q1p);
        bf16x8 gf0 = *(const bf16x8*)(q0p + QT_BYTES);
        bf16x8 gf1 = *(const bf16x8*)(q1p + QT_BYTES);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf0, kf[k2], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf0, vf[k2], dpacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf1, kf[k2 + 1], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf1, vf[k2 + 1], dpacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int m = 0; m < 2; m++) {
#pragma unroll
        for (int q4 = 0; q4 < 2; q4++) {
          const float* sp = (const float*)(smem + s_base) + 32 * qb + 16 * m + 4 * q4;
          const f32x4 lse4 = *(const f32x4*)sp;
          const f32x4 del4 = *(const f32x4*)(sp + 64);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int ri = 8 * m + 4 * q4 + r;
            const float pr = fast_exp2(sacc[ri] * scale2 + kb2 - lse4[r]);
            const float dsv = pr * (dpacc[ri] - del4[r]);
            pf[m][4 * q4 + r] = f2bf(pr);
            dsf[m][4 * q4 + r] = f2bf(dsv);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // dV^T[d, key] += dO^T[d, q] P[q, key];  dK^T[d, key] += Q^T[d, q] dS[q, key]:  A fragments (rows d = 32 dt + l31, k-slots = queries
      // 32 qb + 16 m + 8 h + 0..7) by two transposing reads each (queries +0..3, +4..7).  One (dt, m) at a time: 8 fragment registers live, not 16
      // (the kernel sits at the 256-VGPR limit; the partner wave of the SIMD covers the LDS latency)
#pragma unroll
      for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
        for (int m = 0; m < 2; m++) {
          const char* r0 = smem + (t_base ^ ((4 * dt) << 4)) + (32 * qb + 16 * m) * QROWB;            // queries + 0..3
          const char* r1 = smem + (t_base ^ (((4 * dt) ^ 1) << 4)) + (32 * qb + 16 * m + 4) * QROWB;  // queries + 4..7  (f gains jsel = 1)
          const bf16x8 qtf = lds_tr16x2(r0, r1);
          const bf16x8 gtf = lds_tr16x2(r0 + QT_BYTES, r1 + QT_BYTES);
          acc_dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf, pf[m], acc_dv[dt], 0, 0, 0);
          acc_dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf[m], acc_dk[dt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (rp.out != nullptr) {       // dK through the fused RoPE + RMSNorm backward (straight to the k columns of the projection gradient); dV as before
    // every index is re-derived from the (laundered) thread id: nothing the epilogue needs may stay live across the main loop, which sits at the
    // 256-VGPR limit (a first version that reused key / keyi / h spilled eight fragment registers INTO the loop)
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int h = (t2 >> 5) & 1;
    const int wv2 = __builtin_amdgcn_readfirstlane(t2 >> 6);
    const int key = wg.tile * 256 + wv2 * 32 + (t2 & 31);
    if constexpr (HD == 128) rope_bwd_store(rp, acc_dk, scale, K + bh * (int64_t)Sk * HD, b, head, wg.tile * 256 + wv2 * 32, Sk, t2 & 63, smem + wv2 * 8192);
    if (key < Sk) {
      bf16* vrow = dVrows + ((int64_t)b * Sk + key) * ld_dv + (int64_t)head * HD;
#pragma unroll
      for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          bf16x4 ov;
#pragma unroll
          for (int bb = 0; bb < 4; bb++) ov[bb] = f2bf(acc_dv[dt][4 * a + bb]);
          *(bf16x4*)(vrow + 32 * dt + 8 * a + 4 * h) = ov;
        }
    }
    return;
  }
  if (key < Sk) {
    bf16* krow = dK + (bh * Sk + key) * (int64_t)HD;
    bf16* vrow = dVrows + ((int64_t)b * Sk + key) * ld_dv + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
      for (int a = 0; a < 4; a++) {
        bf16x4 ok, ov;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
          ok[bb] = f2bf(acc_dk[dt][4 * a + bb] * scale);
          ov[bb] = f2bf(acc_dv[dt][4 * a + bb]);
        }
        *(bf16x4*)(krow + 32 * dt + 8 * a + 4 * h) = ok;
        *(bf16x4*)(vrow + 32 * dt + 8 * a + 4 * h) = ov;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// dK/dV kernel, fourth generation (head_dim 128, no key bias, r4): the geometry of dkv3 (8 waves x 32 keys, two waves per SIMD, 256 registers per wave, the
// same LDS images and transposing reads) with a HAND-SCHEDULED body generated by tools/kgen/dkv.py: the statistics ride in the MFMA chains (S chains start
// from lse / scale2 and accumulate q . (-k), dP chains start from +delta and accumulate dO . (-v): four VALU instructions per score instead of five, no
// statistics registers), P / dS are packed in place, fragments are requested two MFMAs ahead into a four-deep ring with counted waits.  The scores are the
// same fp32 sums as dkv3's (K and V only change sign); dK / dV differ from dkv3 by fp32 summation order only (bf16-rounding agreement, not bit for bit).  HIP code computes the lane
// addresses in front of the statement and finishes behind it (dK through the fused RoPE + RMSNorm backward or as head-major rows, dV as token rows).
// ------------------------------------------------------------------------------------------------
#define ST355_DKV4_OPERANDS                                                                                                                            \
        :                                                                                                                                          \
        : [koffs] "v"(koffs), [voffs] "v"(voffs), [rowb] "v"(rowb), [trb] "v"(trb), [statb] "v"(statb), [drow] "v"(drow), [dcol] "v"(dcol),          \
          [kbase] "s"(kbase), [vbase] "s"(vbase), [qbase] "s"(qbase), [gbase] "s"(gbase), [lbase] "s"(lbase), [dbase] "s"(dbase), [lds] "s"(lds), [wv] "s"(wvu), \
          [nqt] "s"(nqt), [sq] "s"(sq), [stmax] "s"(stmax), [ldo2] "s"(ldo2), [q2] "s"(q2), [scale] "s"(scale), [nscale2] "s"(nscale2)                \
        : "memory", "vcc", "scc",

// HD = 128 (Flux) or 96 (PixArt-Sigma's head_dim 72, zero padded); the fused RoPE epilogue is head_dim 128 only
template <int HD>
__global__ void __launch_bounds__(512, 2) k_attn_bwd_dkv4(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vrows, int64_t ld_v,
                                                          const bf16* __restrict__ dO, int64_t ld_do, const float* __restrict__ lsep,
                                                          const float* __restrict__ delta, bf16* __restrict__ dK, bf16* __restrict__ dVrows, int64_t ld_dv,
                                                          int H, int Sq, int Sqp, int Sk, float scale, float scale2, RopeBwd rp) {
  static_assert(HD == 128 || HD == 96 || HD == 64, "k_attn_bwd_dkv4: head_dim 128, 96 or 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  {
    const int h = lane >> 5, l31 = lane & 31;
    const int keyi = min(wg.tile * 256 + wv * 32 + l31, Sk - 1);
    const bf16* kbase = K + bh * (int64_t)Sk * HD;
    const bf16* vbase = Vrows + (int64_t)b * Sk * ld_v + (int64_t)head * HD;
    const uint32_t koffs = (uint32_t)(keyi * HD + 8 * h) * 2u;
    const uint32_t voffs = (uint32_t)((int64_t)keyi * ld_v + 8 * h) * 2u;
    const uint32_t lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int qrow_p = perm23(l31);
    const uint32_t rowb = lds + qrow_p * 256 + ((h ^ swz_q(qrow_p)) << 4);            // row fragments: chunk (2 ks + h) ^ f(row)
    const int tr_r = (lane >> 2) & 3, tr_s = lane & 3, tr_ih = (lane >> 4) & 1;
    const uint32_t trb = lds + (8 * h + tr_r) * 256 + ((((2 * tr_ih + (tr_s >> 1)) ^ (4 * tr_r + 2 * h))) << 4) + (tr_s & 1) * 8;
    const uint32_t statb = lds + 32 * h;
    // LDS-DMA: piece = wave + 8 p, chunk idx = piece * 64 + lane: row = piece * 4 + lane / 16, LDS chunk position c holds source chunk c ^ f(row)
    const int drow_i = wv * 4 + (lane >> 4);
    const uint32_t drow = (uint32_t)drow_i;
    int dchunk = (lane & 15) ^ swz_q(drow_i);
    if (dchunk >= HD / 8) dchunk = 0;                    // (head_dim 96: slots of source chunks >= 12 are never read; their lanes re-fetch chunk 0)
    const uint32_t dcol = (uint32_t)(dchunk * 8) * 2u;
    const bf16* qbase = Q + bh * (int64_t)Sq * HD;
    const bf16* gbase = dO + (int64_t)b * Sq * ld_do + (int64_t)head * HD;
    const float* lbase = lsep + bh * (int64_t)Sqp;
    const float* dbase = delta + bh * (int64_t)Sqp;
    const uint32_t nqt = (uint32_t)((Sq + 63) / 64), sq = (uint32_t)Sq, stmax = (uint32_t)(Sqp - 64), ldo2 = (uint32_t)(ld_do * 2), q2 = (uint32_t)(HD * 2);
    const uint32_t wvu = (uint32_t)wv;
    const float nscale2 = -scale2;
    if constexpr (HD == 128) {
      asm volatile(
#ifdef ST355_DKV4_BODY_INC        // tools/attn_lab builds: a generator variant under test
#include ST355_DKV4_BODY_INC
#else
#include "gen/attn_dkv4_body.inc"
#endif
          ST355_DKV4_OPERANDS
#include "gen/attn_dkv4_clobbers.inc"
      );
    } else if constexpr (HD == 96) {
      asm volatile(
#ifdef ST355_DKV4_HD96_BODY_INC
#include ST355_DKV4_HD96_BODY_INC
#else
#include "gen/attn_dkv4_hd96_body.inc"
#endif
          ST355_DKV4_OPERANDS
#include "gen/attn_dkv4_clobbers.inc"
      );
    } else {
      asm volatile(
#include "gen/attn_dkv4_hd64_body.inc"
          ST355_DKV4_OPERANDS
#include "gen/attn_dkv4_clobbers.inc"
      );
    }
  }
  // every index below is re-derived: nothing needs to live across the statement above
  const char* mine = smem + wv * 16384;
  const int key0 = wg.tile * 256 + wv * 32;
  if (HD == 128 && rp.out != nullptr) {
    if constexpr (HD == 128) rope_bwd_finish(rp, K + bh * (int64_t)Sk * HD, b, head, key0, Sk, lane, mine);
  } else {
    const int tl = lane >> 4, c = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int t = it * 4 + tl;
      if (key0 + t < Sk && c < HD / 8) *(bf16x8*)(dK + (bh * Sk + key0 + t) * (int64_t)HD + c * 8) = *(const bf16x8*)(mine + t * 256 + ((c ^ (t & 15)) << 4));
    }
  }
  {
    const int tl = lane >> 4, c = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int t = it * 4 + tl;
      if (key0 + t < Sk && c < HD / 8)
        *(bf16x8*)(dVrows + ((int64_t)b * Sk + key0 + t) * ld_dv + (int64_t)head * HD + c * 8) = *(const bf16x8*)(mine + 8192 + t * 256 + ((c ^ (t & 15)) << 4));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dQ kernel, second generation (head_dim 128, r4): 4 waves x 64 queries — ONE wave per SIMD with the whole 512-register file.
// Why: in k_attn_bwd_dq every MFMA fetches a fresh 1-KiB operand fragment from LDS (one resident operand, 32 queries per wave), so the LDS pipe has to
// run at the matrix pipe's rate; with 64 queries per wave each K / V / K^T fragment feeds TWO MFMAs (0.5 fragment reads per MFMA).  That needs dQ^T
// 64 q x 128 d (128 registers) + the Q and dO fragments of 64 queries (128) + two generations of score accumulators (128): a one-wave-per-SIMD budget
// that hipcc's allocator does not manage (the plain-HIP form of this kernel compiled to 256 + 256 registers, 117 spills and ~540 v_accvgpr copies per
// tile).  So the main loop is ONE asm statement with an asm-owned register map, generated by tools/kgen/dq64.py (structure, pipeline and the
// wait-state rules it keeps are documented there); HIP code computes the lane addresses in front of it and finishes behind it.
//   pipeline over 32-key blocks j:   A(j) S^T, dP^T (32 MFMAs)   B(j) dS (VALU)   C(j) dQ^T += K^T dS^T (16 MFMAs);  step j = [A(j+1) | B(j)] ; C(j)
//   K / V tiles (row-major, 256-byte pitch, swz_q chunk swizzle) arrive by LDS-DMA into a ring of three 32-KiB slots, one tile ahead, one barrier per tile
//   dQ leaves through the (idle) ring: parked as bf16 token rows, read back 16 lanes per token by rope_bwd_finish (fused RoPE + RMSNorm backward) or
//   stored as whole 256-byte rows (plain head-major dQ)
// Same arithmetic, same accumulation order as k_attn_bwd_dq<128, true, false>: results are bit-identical (tools/attn_lab checks).
// Built for Sk % 64 == 0 and no key bias (the Flux / PixArt-2K self-attention shapes); everything else keeps k_attn_bwd_dq.
// ------------------------------------------------------------------------------------------------
#define ST355_DQ64_OPERANDS                                                                                                                            \
        :                                                                                                                                          \
        : [qp0] "v"(qp0), [qp1] "v"(qp1), [dp0] "v"(dp0), [dp1] "v"(dp1), [nlse0] "v"(nlse0), [nlse1] "v"(nlse1), [del0] "v"(del0), [del1] "v"(del1), \
          [koff] "v"(koff), [voff] "v"(voff), [rowb] "v"(rowb), [trb] "v"(trb), [park] "v"(park), [kbase] "s"(kbase), [vbase] "s"(vbase),           \
          [lds] "s"(lds), [wvoff] "s"(wvoff), [nkt] "s"(nkt), [vstep] "s"(vstep), [vrow16] "s"(vrow16), [scale2] "s"(scale2), [scale] "s"(scale),   \
          [blk0] "s"(blk0), [tracelo] "s"(tracelo), [tracehi] "s"(tracehi)                                                                          \
        : "memory", "vcc", "scc",

// HD = 128 (Flux) or 96 (PixArt-Sigma's head_dim 72, zero padded: 6 k-steps, 3 d tiles; tile images keep the 256-byte row pitch, the DMA lanes of the unused
// chunk slots fetch chunk 0 again); the fused RoPE epilogue is head_dim 128 only
template <int HD>
__global__ void __launch_bounds__(256, 1) k_attn_bwd_dq64(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vrows, int64_t ld_v,
                                                          const bf16* __restrict__ dO, int64_t ld_do, const float* __restrict__ lse2,
                                                          const float* __restrict__ delta, bf16* __restrict__ dQ, int H, int Sq, int Sqp, int Sk,
                                                          float scale, float scale2, RopeBwd rp, unsigned long long* trace) {
  static_assert(HD == 128 || HD == 96 || HD == 64, "k_attn_bwd_dq64: head_dim 128, 96 or 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const WgMap wg = attn_wg_map();
  const int head = wg.head, b = wg.b;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = wg.tile * 256 + wv * 64;
  {
    // (lab builds of the body stamp s_memtime at the phase boundaries and dump them through `trace`; the product body ignores these three operands)
    const uint32_t blk0 = blockIdx.x | blockIdx.y | blockIdx.z;
    const uint32_t tracelo = (uint32_t)(uintptr_t)trace, tracehi = (uint32_t)((uintptr_t)trace >> 32);
    const int qi0 = min(q0 + l31, Sq - 1), qi1 = min(q0 + 32 + l31, Sq - 1);
    const bf16* qp0 = Q + (bh * Sq + qi0) * (int64_t)HD + 8 * h;
    const bf16* qp1 = Q + (bh * Sq + qi1) * (int64_t)HD + 8 * h;
    const bf16* dp0 = dO + ((int64_t)b * Sq + qi0) * ld_do + (int64_t)head * HD + 8 * h;
    const bf16* dp1 = dO + ((int64_t)b * Sq + qi1) * ld_do + (int64_t)head * HD + 8 * h;
    const float nlse0 = -lse2[bh * Sq + qi0], nlse1 = -lse2[bh * Sq + qi1];
    const float del0 = delta[bh * (int64_t)Sqp + qi0], del1 = delta[bh * (int64_t)Sqp + qi1];
    // LDS-DMA lane offsets of the wave's first piece (piece = wave + 4 p; chunk idx = piece * 64 + lane): LDS chunk position c of row r holds source chunk c ^ f(r)
    const int row = wv * 4 + (lane >> 4);
    int chunk = (lane & 15) ^ swz_q(row);
    if (chunk >= HD / 8) chunk = 0;                    // (head_dim 96: slots of source chunks >= 12 are never read; their lanes re-fetch chunk 0)
    const int col = chunk * 8;
    const uint32_t koff = (uint32_t)(row * HD + col) * 2u;
    const uint32_t voff = (uint32_t)((int64_t)row * ld_v + col) * 2u;
    const uint32_t vrow16 = (uint32_t)(16 * ld_v * 2), vstep = (uint32_t)(64 * ld_v * 2);
    const int rowp = perm23(l31);
    const uint32_t rowb = rowp * 256 + ((h ^ swz_q(rowp)) << 4);                     // row fragments: chunk (2 ks + h) ^ f(row)
    const int tr_r = (lane >> 2) & 3, tr_s = lane & 3, tr_ih = (lane >> 4) & 1;
    const uint32_t trb = (8 * h + tr_r) * 256 + ((((2 * tr_ih + (tr_s >> 1)) ^ (4 * tr_r + 2 * h))) << 4) + (tr_s & 1) * 8;   // transposed fragments (dq<TR>)
    const uint32_t park = l31 * 256 + ((l31 & 15) << 4) + 8 * h;
    const bf16* kbase = K + bh * (int64_t)Sk * HD;
    const bf16* vbase = Vrows + (int64_t)b * Sk * ld_v + (int64_t)head * HD;
    const uint32_t lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t wvoff = (uint32_t)wv * 1024u;
    const uint32_t nkt = (uint32_t)(Sk / 64);
    if constexpr (HD == 128) {
      asm volatile(
#ifdef ST355_DQ64_BODY_INC       // tools/attn_lab builds: a generator variant under test
#include ST355_DQ64_BODY_INC
#else
#include "gen/attn_dq64_body.inc"
#endif
          ST355_DQ64_OPERANDS
#include "gen/attn_dq64_clobbers.inc"
      );
    } else if constexpr (HD == 96) {
      asm volatile(
#ifdef ST355_DQ64_HD96_BODY_INC
#include ST355_DQ64_HD96_BODY_INC
#else
#include "gen/attn_dq64_hd96_body.inc"
#endif
          ST355_DQ64_OPERANDS
#include "gen/attn_dq64_clobbers.inc"
      );
    } else {
      asm volatile(
#include "gen/attn_dq64_hd64_body.inc"
          ST355_DQ64_OPERANDS
#include "gen/attn_dq64_clobbers.inc"
      );
    }
  }
  // every index below is re-derived: nothing needs to live across the statement above
  const char* mine = smem + wv * 16384;
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    const int tok0 = q0 + 32 * qb;
    if (HD == 128 && rp.out != nullptr) {
      if constexpr (HD == 128) rope_bwd_finish(rp, Q + bh * (int64_t)Sq * HD, b, head, tok0, Sq, lane, mine + qb * 8192);
    } else {
      const int tl = lane >> 4, c = lane & 15;
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int t = it * 4 + tl;
        if (tok0 + t < Sq && c < HD / 8)
          *(bf16x8*)(dQ + (bh * Sq + tok0 + t) * (int64_t)HD + c * 8) = *(const bf16x8*)(mine + qb * 8192 + t * 256 + ((c ^ (t & 15)) << 4));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
static inline size_t round256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t st355_attn_bwd_workspace(int B, int H, int S, int Sp, int d) {   // S, Sp: the QUERY length and its padding
  return 2 * round256((size_t)B * H * Sp * sizeof(float)) + round256((size_t)B * H * d * Sp * 2);   // delta, lse (padded) + dO^T
}

// dQ kernel choice: 64 = k_attn_bwd_dq64 where it applies (default), 32 = always k_attn_bwd_dq.  ST355_ATTN_DQ overrides; tools/attn_lab sets the variable directly.
int g_attn_dq_impl = -1;
unsigned long long* g_attn_dq_trace = nullptr;     // tools/attn_lab, trace builds of the dq64 body only
// dK/dV kernel choice for head_dim 128 without key bias: 4 = k_attn_bwd_dkv4 (hand-scheduled body), 3 = k_attn_bwd_dkv3.  ST355_ATTN_DKV=3 overrides.
int g_attn_dkv_impl = -1;
static int attn_dkv_impl() {
  if (g_attn_dkv_impl < 0) { const char* e = getenv("ST355_ATTN_DKV"); g_attn_dkv_impl = (e && atoi(e) == 3) ? 3 : 4; }
  return g_attn_dkv_impl;
}
static int attn_dq_impl() {
  if (g_attn_dq_impl < 0) { const char* e = getenv("ST355_ATTN_DQ"); g_attn_dq_impl = (e && atoi(e) == 32) ? 32 : 64; }
  return g_attn_dq_impl;
}

extern int g_attn_fwd_impl;                        // attention.hip
extern "C" int st355_attn_set_impl(int fwd, int dq, int dkv) {
  if (g_attn_fwd_impl < 0) { const char* e = getenv("ST355_ATTN_FWD64"); g_attn_fwd_impl = (e && e[0] == '0') ? 32 : 64; }
  const int prev = g_attn_fwd_impl * 65536 + attn_dq_impl() * 256 + attn_dkv_impl();
  if (fwd == 32 || fwd == 64) g_attn_fwd_impl = fwd;
  if (dq == 32 || dq == 64) g_attn_dq_impl = dq;
  if (dkv == 3 || dkv == 4) g_attn_dkv_impl = dkv;
  return prev;
}

static int attn_bwd_impl(void* stream, const void* Q, const void* K, const void* Qt, const void* Kt, const void* v_rows,
                              int64_t ld_v, const void* O, int64_t ld_o, const void* dO, int64_t ld_do, const float* lse2,
                              const float* key_bias, void* dQ, void* dK, void* dv_rows, int64_t ld_dv, int B, int H, int S, int Sp, int Sk, int Skp,
                              int d, float scale, void* workspace, const RopeBwd* rope_q = nullptr, const RopeBwd* rope_k = nullptr, const void* O_res = nullptr) {
  const RopeBwd rq = rope_q ? *rope_q : RopeBwd{}, rk = rope_k ? *rope_k : RopeBwd{};       // out == NULL: head-major dQ / dK as before
  ST_REQUIRE(Q && K && v_rows && O && dO && lse2 && (dQ || rq.out) && (dK || rk.out) && dv_rows && workspace, "attn_bwd: null pointer");
  // Qt == NULL selects the third-generation dK/dV kernel, Kt == NULL the transposing-read dQ kernel (head_dim 128): Q^T / dO^T / K^T fragments come
  // from the row-major tiles by ds_read_b64_tr_b16 instead of pre-transposed head-major copies
  ST_REQUIRE((rq.out == nullptr && rk.out == nullptr) || d == 128, "attn_bwd: the fused RoPE epilogues are built for head_dim 128 (got %d)", d);
  ST_REQUIRE(B > 0 && H > 0 && S > 0 && Sp % 64 == 0 && Sp >= S && Sk > 0 && Skp % 64 == 0 && Skp >= Sk, "attn_bwd: bad shape S=%d Sp=%d Sk=%d Skp=%d", S, Sp, Sk, Skp);
  ST_REQUIRE(ld_v % 8 == 0 && ld_o % 8 == 0 && ld_do % 8 == 0 && ld_dv % 4 == 0, "attn_bwd: leading dimensions must be multiples of 8");
  ST_REQUIRE(((uintptr_t)workspace & 255) == 0, "attn_bwd: workspace must be 256-byte aligned");
  if (d != 128 && d != 64 && d != 96) { st355_set_error("attn_bwd: head_dim %d not built", d); return ST355_ENOSYS; }
  float* delta = (float*)workspace;
  float* lsep = (float*)((char*)workspace + round256((size_t)B * H * Sp * sizeof(float)));
  bf16* dOt = Qt ? (bf16*)((char*)workspace + 2 * round256((size_t)B * H * Sp * sizeof(float))) : nullptr;
  const float scale2 = scale * LOG2E;
  const double fl_unit = 2.0 * (double)B * H * (double)S * Sk * d;  // one Sq x Sk x d contraction
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const bool use_dkv4 = !Qt && !key_bias && attn_dkv_impl() == 4;                // head_dim 128 / 96 / 64 (every head_dim built)
  {
    ProfScope ps(stream, ST355_K_ATTN_PREP, 2.0 * B * H * (double)S * d, 6.0 * B * H * (double)S * d);
    dim3 grid(Sp / 64, H, B);
    const float lse_mul = use_dkv4 ? 1.f / scale2 : 1.f;
    if (d == 96)
      hipLaunchKernelGGL(k_attn_bwd_prep<96>, grid, dim3(256), 0, st, (const bf16*)O, ld_o, (const bf16*)dO, ld_do, lse2, delta, lsep, dOt, H, S, Sp, lse_mul, (const bf16*)O_res);
    else if (d == 128)
      hipLaunchKernelGGL(k_attn_bwd_prep<128>, grid, dim3(256), 0, st, (const bf16*)O, ld_o, (const bf16*)dO, ld_do, lse2, delta, lsep, dOt, H, S, Sp, lse_mul, (const bf16*)O_res);
    else
      hipLaunchKernelGGL(k_attn_bwd_prep<64>, grid, dim3(256), 0, st, (const bf16*)O, ld_o, (const bf16*)dO, ld_do, lse2, delta, lsep, dOt, H, S, Sp, lse_mul, (const bf16*)O_res);
    if ((rc = st355_check_launch("attn_bwd_prep")) != 0) return rc;
  }
  {
    ProfScope ps(stream, ST355_K_ATTN_BWD_DKV, 4.0 * fl_unit, 2.0 * (double)B * H * (S + Sk) * d * 4.0);
    if (use_dkv4) {                                                    // hand-scheduled body (k_attn_bwd_dkv4)
      dim3 grid((Sk + 255) / 256, H, B);
      const int lds = 8 * 16384;                                       // the two ring slots (66.5 KiB); 16 KiB per wave for the parked dK / dV rows
#define ST355_DKV4_LAUNCH(HD_)                                                                                                            \
  do {                                                                                                                                   \
    static St355AttrOnce set;                                                                                                             \
    if (set.need()) { hipFuncSetAttribute((const void*)k_attn_bwd_dkv4<HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }   \
    hipLaunchKernelGGL(k_attn_bwd_dkv4<HD_>, grid, dim3(512), lds, st, (const bf16*)Q, (const bf16*)K, (const bf16*)v_rows, ld_v, (const bf16*)dO, ld_do,  \
                       (const float*)lsep, (const float*)delta, (bf16*)dK, (bf16*)dv_rows, ld_dv, H, S, Sp, Sk, scale, scale2, rk);      \
  } while (0)
      if (d == 128) ST355_DKV4_LAUNCH(128);
      else if (d == 96) ST355_DKV4_LAUNCH(96);
      else ST355_DKV4_LAUNCH(64);
#undef ST355_DKV4_LAUNCH
    } else if (!Qt) {
      dim3 grid((Sk + 255) / 256, H, B);
      const int lds = 2 * (2 * 64 * 256 + 512);
#define ST355_DKV3_LAUNCH(HD_)                                                                                                            \
  do {                                                                                                                                   \
    static St355AttrOnce set;                                                                                                             \
    if (set.need()) { hipFuncSetAttribute((const void*)k_attn_bwd_dkv3<HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }   \
    hipLaunchKernelGGL(k_attn_bwd_dkv3<HD_>, grid, dim3(512), lds, st, (const bf16*)Q, (const bf16*)K, (const bf16*)v_rows, ld_v,        \
                       (const bf16*)dO, ld_do, (const float*)lsep, (const float*)delta, key_bias, (bf16*)dK, (bf16*)dv_rows, ld_dv, H, S,  \
                       Sp, Sk, scale, scale2, rk);                                                                                        \
  } while (0)
      if (d == 128) ST355_DKV3_LAUNCH(128);
      else if (d == 96) ST355_DKV3_LAUNCH(96);
      else ST355_DKV3_LAUNCH(64);
#undef ST355_DKV3_LAUNCH
    } else {                                   // head-major Q^T / dO^T copies supplied: the LDS-DMA kernel over four tile images (head_dim 64 / 96 / 128)
      dim3 grid((Sk + 255) / 256, H, B);
      const int lds = 2 * (2 * 64 * d * 2 + 2 * d * 128 + 512);
#define ST355_DKV2_LAUNCH(KERN)                                                                                                          \
  do {                                                                                                                                   \
    static St355AttrOnce set;                                                                                                             \
    if (set.need()) { hipFuncSetAttribute((const void*)(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }                 \
    hipLaunchKernelGGL((KERN), grid, dim3(512), lds, st, (const bf16*)Q, (const bf16*)K, (const bf16*)Qt, (const bf16*)v_rows, ld_v,     \
                       (const bf16*)dO, ld_do, (const bf16*)dOt, (const float*)lsep, (const float*)delta, key_bias, (bf16*)dK,           \
                       (bf16*)dv_rows, ld_dv, H, S, Sp, Sk, Skp, scale, scale2);                                                        \
  } while (0)
      if (d == 128) ST355_DKV2_LAUNCH(k_attn_bwd_dkv2<128>);
      else if (d == 96) ST355_DKV2_LAUNCH(k_attn_bwd_dkv2<96>);     // PixArt's 72, zero-padded: 12 DMA pieces per tile image over the 8 waves
      else ST355_DKV2_LAUNCH(k_attn_bwd_dkv2<64>);
#undef ST355_DKV2_LAUNCH
    }
    if ((rc = st355_check_launch("attn_bwd_dkv")) != 0) return rc;
  }
  // the 64-row kernel takes the full 64-key tiles; a ragged key tail (SD3's S = 4096 + 231, every mixed-aspect bucket) is added by the general kernel restricted
  // to the last tile (kt0 / accumulate above).  The fused-RoPE epilogue (head_dim 128, Flux: S % 64 == 0 always) writes projection rows, not dQ: no tail form.
  const bool dq_tail = Sk % 64 != 0;
  // a SHORT ragged key axis (the UNets' cross-attention: 77 text keys = one full tile + 13 keys) goes through the general kernel in ONE pass: as 64-row kernel + tail
  // it was two memory-bound passes over Q / dO / dQ with a bf16 round trip of dQ between them (r6; ST355_ATTN_DQ_SHORT=0: A/B)
  static int dq_short = -1;
  if (dq_short < 0) { const char* e = getenv("ST355_ATTN_DQ_SHORT"); dq_short = (e && e[0] == '0') ? 0 : 1; }
  const bool short_ragged = dq_short && dq_tail && Sk < 128;
  if (!Kt && !key_bias && attn_dq_impl() == 64 && Sk >= 64 && !short_ragged && (!dq_tail || (rq.out == nullptr && dQ != nullptr))) {      // hand-scheduled 64-queries-per-wave kernel (k_attn_bwd_dq64)
    ProfScope ps(stream, ST355_K_ATTN_BWD_DQ, 3.0 * fl_unit, 2.0 * (double)B * H * (S + Sk) * d * 3.0);
    dim3 grid((S + 255) / 256, H, B);
    const int lds = 3 * 2 * 64 * 256;
#define ST355_DQ64_LAUNCH(HD_)                                                                                                            \
  do {                                                                                                                                   \
    static St355AttrOnce set;                                                                                                             \
    if (set.need()) { hipFuncSetAttribute((const void*)k_attn_bwd_dq64<HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }   \
    hipLaunchKernelGGL(k_attn_bwd_dq64<HD_>, grid, dim3(256), lds, st, (const bf16*)Q, (const bf16*)K, (const bf16*)v_rows, ld_v, (const bf16*)dO, ld_do, lse2, \
                       (const float*)delta, (bf16*)dQ, H, S, Sp, Sk, scale, scale2, rq, g_attn_dq_trace);                                \
  } while (0)
    if (d == 128) ST355_DQ64_LAUNCH(128);
    else if (d == 96) ST355_DQ64_LAUNCH(96);
    else ST355_DQ64_LAUNCH(64);
#undef ST355_DQ64_LAUNCH
    if ((rc = st355_check_launch("attn_bwd_dq64")) != 0) return rc;
    if (dq_tail) {
      const int lds_t = 2 * (64 * 256 + 64 * d * 2);
#define ST355_DQT_LAUNCH(HD_)                                                                                                             \
  do {                                                                                                                                   \
    static St355AttrOnce set;                                                                                                             \
    if (set.need()) { hipFuncSetAttribute((const void*)k_attn_bwd_dq<HD_, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_t); } \
    hipLaunchKernelGGL((k_attn_bwd_dq<HD_, true, false>), grid, dim3(512), lds_t, st, (const bf16*)Q, (const bf16*)K, (const bf16*)nullptr, (const bf16*)v_rows, ld_v, \
                       (const bf16*)dO, ld_do, lse2, (const float*)delta, (const float*)nullptr, (bf16*)dQ, H, S, Sp, Sk, Skp, scale, scale2, rq, Sk / 64, 1);   \
  } while (0)
      if (d == 128) ST355_DQT_LAUNCH(128);
      else if (d == 96) ST355_DQT_LAUNCH(96);
      else ST355_DQT_LAUNCH(64);
#undef ST355_DQT_LAUNCH
      if ((rc = st355_check_launch("attn_bwd_dq_tail")) != 0) return rc;
    }
  } else {
    ProfScope ps(stream, ST355_K_ATTN_BWD_DQ, 3.0 * fl_unit, 2.0 * (double)B * H * (S + Sk) * d * 3.0);
    dim3 grid((S + 255) / 256, H, B);
    const int ktb = 64 * d * 2;                                   // one row-major 64-key tile
    const bool tr = !Kt;                                          // no K^T copy: transposing-read kernel (K tile image at a 256-byte row pitch)
    const int lds = 2 * ((tr ? 64 * 256 : ktb) + ktb + (tr ? 0 : d * 128));
#define ST355_DQ_LAUNCH(KERN)                                                                                                            \
  do {                                                                                                                                   \
    static St355AttrOnce set;                                                                                                             \
    if (set.need()) { hipFuncSetAttribute((const void*)(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }                 \
    hipLaunchKernelGGL((KERN), grid, dim3(512), lds, st, (const bf16*)Q, (const bf16*)K, (const bf16*)Kt, (const bf16*)v_rows, ld_v,     \
                       (const bf16*)dO, ld_do, lse2, (const float*)delta, key_bias, (bf16*)dQ, H, S, Sp, Sk, Skp, scale, scale2, rq);    \
  } while (0)
#define ST355_DQ_PICK(HD_)                                                                   \
  do {                                                                                       \
    if (key_bias) {                                                                          \
      if (tr) ST355_DQ_LAUNCH((k_attn_bwd_dq<HD_, true, true>));                             \
      else ST355_DQ_LAUNCH((k_attn_bwd_dq<HD_, false, true>));                               \
    } else {                                                                                 \
      if (tr) ST355_DQ_LAUNCH((k_attn_bwd_dq<HD_, true, false>));                            \
      else ST355_DQ_LAUNCH((k_attn_bwd_dq<HD_, false, false>));                              \
    }                                                                                        \
  } while (0)
    if (d == 128) ST355_DQ_PICK(128);
    else if (d == 96) ST355_DQ_PICK(96);
    else ST355_DQ_PICK(64);
#undef ST355_DQ_PICK
#undef ST355_DQ_LAUNCH
    if ((rc = st355_check_launch("attn_bwd_dq")) != 0) return rc;
  }
  return ST355_OK;
}
extern "C" int st355_attn_bwd(void* stream, const void* Q, const void* K, const void* Qt, const void* Kt, const void* v_rows,
                              int64_t ld_v, const void* O, int64_t ld_o, const void* dO, int64_t ld_do, const float* lse2,
                              const float* key_bias, void* dQ, void* dK, void* dv_rows, int64_t ld_dv, int B, int H, int S, int Sp,
                              int d, float scale, void* workspace) {
  return attn_bwd_impl(stream, Q, K, Qt, Kt, v_rows, ld_v, O, ld_o, dO, ld_do, lse2, key_bias, dQ, dK, dv_rows, ld_dv, B, H, S, Sp, S, Sp, d, scale, workspace);
}
// st355_attn_bwd / st355_attn_cross_bwd with delta = rowsum(dO * (O + O_res)): O_res is what st355_attn_fwd_res wrote (attention.hip has the why)
extern "C" int st355_attn_bwd_res(void* stream, const void* Q, const void* K, const void* Qt, const void* Kt, const void* v_rows, int64_t ld_v, const void* O,
                                  int64_t ld_o, const void* O_res, const void* dO, int64_t ld_do, const float* lse2, const float* key_bias, void* dQ, void* dK,
                                  void* dv_rows, int64_t ld_dv, int B, int H, int Sq, int Sqp, int Sk, int Skp, int d, float scale, void* workspace) {
  ST_REQUIRE(O_res, "attn_bwd_res: null residual pointer");
  return attn_bwd_impl(stream, Q, K, Qt, Kt, v_rows, ld_v, O, ld_o, dO, ld_do, lse2, key_bias, dQ, dK, dv_rows, ld_dv, B, H, Sq, Sqp, Sk, Skp, d, scale, workspace,
                       nullptr, nullptr, O_res);
}
// Self-attention backward with the RoPE + RMSNorm backward fused into the dQ / dK epilogues (head_dim 128, the fused-projection form): dq, dk and dv all
// land in the rows of the projection gradient dqkv [B*S, ld] (column blocks q | k | v of width D = H*128); no head-major dQ / dK exists.
extern "C" int st355_attn_bwd_rope(void* stream, const void* Q, const void* K, const void* v_rows, int64_t ld_v, const void* O, int64_t ld_o,
                                   const void* dO, int64_t ld_do, const float* lse2, const float* key_bias, const float* rrms, const void* wq_lo,
                                   const void* wk_lo, const void* wq_hi, const void* wk_hi, int split, const float* cos_p, const float* sin_p,
                                   void* dqkv, int64_t ld_dqkv, int B, int H, int S, int Sp, int d, float scale, void* workspace) {
  ST_REQUIRE(d == 128, "attn_bwd_rope: head_dim %d not built (128 only)", d);
  ST_REQUIRE(rrms && cos_p && sin_p && dqkv && ld_dqkv % 8 == 0 && ld_dqkv >= 3 * (int64_t)H * d && split >= 0 && split <= S, "attn_bwd_rope: bad arguments");
  ST_REQUIRE((wq_lo == nullptr) == (wq_hi == nullptr) && (wk_lo == nullptr) == (wk_hi == nullptr), "attn_bwd_rope: a norm weight needs both position ranges");
  const int64_t D = (int64_t)H * d;
  RopeBwd rq{rrms, (const bf16*)wq_lo, (const bf16*)wq_hi, cos_p, sin_p, (bf16*)dqkv, ld_dqkv, split, H, S, 0};
  RopeBwd rk{rrms, (const bf16*)wk_lo, (const bf16*)wk_hi, cos_p, sin_p, (bf16*)dqkv + D, ld_dqkv, split, H, S, H};
  return attn_bwd_impl(stream, Q, K, nullptr, nullptr, v_rows, ld_v, O, ld_o, dO, ld_do, lse2, key_bias, nullptr, nullptr, (bf16*)dqkv + 2 * D, ld_dqkv,
                       B, H, S, Sp, S, Sp, d, scale, workspace, &rq, &rk);
}
// cross-attention backward: Q,Qt over Sq (padded Sqp) queries; K,Kt, v_rows / dv_rows ([B*Sk, ld]) over Sk (padded Skp) keys; workspace sized for (Sq, Sqp)
extern "C" int st355_attn_cross_bwd(void* stream, const void* Q, const void* K, const void* Qt, const void* Kt, const void* v_rows,
                                    int64_t ld_v, const void* O, int64_t ld_o, const void* dO, int64_t ld_do, const float* lse2,
                                    const float* key_bias, void* dQ, void* dK, void* dv_rows, int64_t ld_dv, int B, int H, int Sq, int Sqp,
                                    int Sk, int Skp, int d, float scale, void* workspace) {
  return attn_bwd_impl(stream, Q, K, Qt, Kt, v_rows, ld_v, O, ld_o, dO, ld_do, lse2, key_bias, dQ, dK, dv_rows, ld_dv, B, H, Sq, Sqp, Sk, Skp, d, scale, workspace);
}
