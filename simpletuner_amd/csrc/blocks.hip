// blocks.hip — block-level entry points (`st355_block_flux_single_{fwd,bwd}`): one FluxSingleTransformerBlock forward / backward as ONE
// C call.  reference seam: FluxSingleTransformerBlock.forward (simpletuner/helpers/models/flux/transformer.py:473-510) and the autograd through it.
//
// These functions only SEQUENCE entry points that libst355 already exports (AdaLN modulate, the fused QKV projection with RMSNorm + RoPE in its epilogue,
// attention, the GELU / gated-residual GEMMs and their backward forms, the rank-space adapter gradients): same launches, same order and same operands as
// the host-side sequencing in simpletuner_amd/flux/transformer.py (`_single_fwd` / `_single_bwd`), so the results are bit-identical to it — which the GPU
// suite checks.  Every buffer is the caller's (activations kept for the backward, scratch, split-K / attention / skinny workspaces): nothing is allocated.
// Built for the production form of the block: head_dim 128, the fused projection epilogue (token count per sample a multiple of 256), optional LoRA
// adapters on to_q / to_k / to_v riding in the K-extension.
#include <string.h>

#include "common.h"

namespace {
struct Seq {
  void* st; int rc;
  void run(int r) { if (rc == 0 && r != 0) rc = r; }
  bool ok() const { return rc == 0; }
  void gemm(const st355_gemm_args& a) { if (ok()) run(st355_gemm_bf16(st, &a)); }
};
st355_gemm_args G(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K) {
  st355_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.epilogue = ST355_EPI_NONE;
  return a;
}
// the host wrapper hands thin problems (N <= 128, M >= 1024: the LoRA down projections) the caller's fp32 split-K scratch
void thin_ws(st355_gemm_args& a, void* ws, int64_t bytes) {
  if (a.N <= 128 && a.M >= 1024) { a.workspace = ws; a.workspace_bytes = bytes; }
}
}  // namespace

extern "C" int st355_block_flux_single_fwd(void* stream, const st355_flux_single_fwd_args* p) {
  ST_REQUIRE(p && p->x && p->x_out && p->n && p->V && p->rrms && p->Q && p->K && p->Vt && p->O && p->lse2 && p->hpre && p->hact, "block_flux_single_fwd: null pointer");
  ST_REQUIRE(p->B > 0 && p->S > 0 && p->S % 256 == 0 && p->H > 0 && p->H % 2 == 0 && p->D == p->H * 128, "block_flux_single_fwd: built for head_dim 128, even H, S %% 256 == 0");
  ST_REQUIRE((p->K2 == 0) == (p->A_cat == nullptr) && (p->K2 == 0 || (p->B_blk && p->T && p->K2 % 64 == 0)), "block_flux_single_fwd: inconsistent adapter operands");
  const int B = p->B, S = p->S, H = p->H, D = p->D, M = B * S;
  Seq q{stream, 0};
  // norm_hidden = LN(x) * (1 + scale) + shift                                          (flux/transformer.py:396-403 AdaLayerNormZeroSingle)
  q.run(st355_ln_modulate_fwd(stream, p->x, D, p->mod_scale, p->mod_shift, p->mod_stride, S, p->n, D, M, D, 1e-6f));
  // q | k | v = norm_hidden Wqkv^T + b (+ adapters in the K-extension); RMSNorm(q), RMSNorm(k), RoPE and the head-major re-layout in the epilogue (:140-207)
  if (p->K2) {
    st355_gemm_args t = G(p->n, D, p->A_cat, D, p->T, p->K2, M, p->K2, D);
    thin_ws(t, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(t);
  }
  st355_qk_rope rope;
  memset(&rope, 0, sizeof(rope));
  rope.Q = p->Q; rope.K = p->K; rope.rrms = p->rrms; rope.wq = p->norm_q; rope.wk = p->norm_k; rope.cos = p->cos_p; rope.sin = p->sin_p;
  rope.H = H; rope.S = S; rope.pos0 = 0; rope.eps = 1e-6f; rope.Vt = p->Vt; rope.Sp = S;
  {
    st355_gemm_args a = G(p->n, D, p->w_qkv, D, p->V, D, M, 3 * D, D);
    a.bias = p->b_qkv; a.epilogue = ST355_EPI_QK_NORM_ROPE; a.rope = &rope; a.rows_per_batch = S;
    if (p->K2) { a.A2 = p->T; a.lda2 = p->K2; a.B2 = p->B_blk; a.ldb2 = p->K2; a.K2 = p->K2; a.K2_real = p->k2_real; }
    q.gemm(a);
  }
  if (q.ok()) q.run(st355_attn_fwd(stream, p->Q, p->K, p->Vt, p->key_bias, p->O, D, p->lse2, B, H, S, S, 128, p->scale));
  // mlp = GELU(norm_hidden Wmlp^T + b), pre-activation kept for the backward (:489-490)
  {
    st355_gemm_args a = G(p->n, D, p->w_mlp, D, p->hact, 4 * D, M, 4 * D, D);
    a.bias = p->b_mlp; a.epilogue = ST355_EPI_GELU; a.aux_out = p->hpre; a.ld_aux_out = 4 * D;
    q.gemm(a);
  }
  // x' = x + gate * (cat[attn, mlp] Wout^T + b): the concat is a two-segment K loop (:498-503)
  {
    st355_gemm_args a = G(p->O, D, p->w_out, p->ld_w_out, p->x_out, D, M, D, D);
    a.bias = p->b_out; a.A2 = p->hact; a.lda2 = 4 * D; a.B2 = (const char*)p->w_out + (size_t)D * 2; a.ldb2 = p->ld_w_out; a.K2 = 4 * D;
    a.epilogue = ST355_EPI_GATE_RESIDUAL; a.aux_in = p->x; a.ld_aux_in = D; a.gate = p->mod_gate; a.gate_stride = p->mod_stride; a.rows_per_batch = S;
    q.gemm(a);
  }
  return q.rc;
}

extern "C" int st355_block_flux_single_bwd(void* stream, const st355_flux_single_bwd_args* p) {
  ST_REQUIRE(p && p->x && p->n && p->V && p->rrms && p->Q && p->K && p->O && p->lse2 && p->hpre && p->dx && p->dx_out, "block_flux_single_bwd: null pointer");
  ST_REQUIRE((p->dxg || p->g) && p->dO && p->dhpre && p->dn_mlp && p->dqkv && p->dn && p->attn_ws, "block_flux_single_bwd: null scratch pointer");
  ST_REQUIRE(p->B > 0 && p->S > 0 && p->S % 256 == 0 && p->H > 0 && p->H % 2 == 0 && p->D == p->H * 128, "block_flux_single_bwd: built for head_dim 128, even H, S %% 256 == 0");
  ST_REQUIRE((p->K2 == 0) == (p->A_cat_T == nullptr) && (p->K2 == 0 || (p->B_blk_T && p->T && p->U && p->skinny_ws && p->n_targets >= 1 && p->n_targets <= 4)),
             "block_flux_single_bwd: inconsistent adapter operands");
  ST_REQUIRE((p->dxg_out == nullptr) == (p->gate_prev == nullptr), "block_flux_single_bwd: the gated output needs the previous block's gate");
  const int B = p->B, S = p->S, H = p->H, D = p->D, M = B * S;
  Seq q{stream, 0};
  // g = gate * d(x')   (handed over pre-gated by the next block's backward, else computed here)
  const void* g = p->dxg;
  if (g == nullptr) {
    q.run(st355_scale_cols(stream, p->dx, D, p->mod_gate, p->mod_stride, S, p->g, D, M, D));
    g = p->g;
  }
  // d cat[attn, mlp] = g Wout: the attention part straight to dO, the mlp part through GELU' of the kept pre-activation
  q.gemm(G(g, D, p->wT_out, D, p->dO, D, M, D, D));
  {
    st355_gemm_args a = G(g, D, (const char*)p->wT_out + (size_t)D * D * 2, D, p->dhpre, 4 * D, M, 4 * D, D);
    a.epilogue = ST355_EPI_MUL_GELU_GRAD; a.aux_in = p->hpre; a.ld_aux_in = 4 * D;
    q.gemm(a);
  }
  q.gemm(G(p->dhpre, 4 * D, p->wT_mlp, 4 * D, p->dn_mlp, D, M, D, 4 * D));
  // attention backward with the RoPE + RMSNorm backward in the dQ / dK kernels' epilogues: dq | dk | dv rows of the projection gradient
  if (q.ok())
    q.run(st355_attn_bwd_rope(stream, p->Q, p->K, p->V, D, p->O, D, p->dO, D, p->lse2, p->key_bias, p->rrms, p->norm_q, p->norm_k, p->norm_q, p->norm_k, 0,
                              p->cos_p, p->sin_p, p->dqkv, 3 * D, B, H, S, S, 128, p->scale, p->attn_ws));
  // d norm_hidden = dqkv Wqkv (+ (dqkv sB) A) + d norm_hidden(mlp)
  if (p->K2) {
    st355_gemm_args u = G(p->dqkv, 3 * D, p->B_blk_T, 3 * D, p->U, p->K2, M, p->K2, 3 * D);
    thin_ws(u, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(u);
  }
  {
    st355_gemm_args a = G(p->dqkv, 3 * D, p->wT_qkv, 3 * D, p->dn, D, M, D, 3 * D);
    if (p->K2) { a.A2 = p->U; a.lda2 = p->K2; a.B2 = p->A_cat_T; a.ldb2 = p->K2; a.K2 = p->K2; a.K2_real = p->k2_real; }
    a.epilogue = ST355_EPI_ADD; a.aux_in = p->dn_mlp; a.ld_aux_in = D;
    q.gemm(a);
  }
  // rank-space adapter gradients: dB_t = s dy_t^T T_t, dA_t = U_t^T norm_hidden (LoraGroup.grads of the host side)
  if (p->K2 && q.ok()) {
    const int rank = p->rank, r_pad = p->r_pad, cw = r_pad < 64 ? r_pad : 64;
    const bool multi = p->n_targets > 1 && r_pad == 32 && p->K2 >= 128;
    for (int t = 0; t < p->n_targets && q.ok(); t++) {
      const int Nt = D;                                                          // to_q / to_k / to_v: D output columns each, at column t * D
      for (int s0 = 0; s0 < rank && q.ok(); s0 += cw) {
        const int c0 = t * r_pad + s0, r_used = (rank - s0) < cw ? (rank - s0) : cw;
        q.run(st355_skinny_tn_seg(stream, (const char*)p->dqkv + (size_t)t * D * 2, 3 * D, (const char*)p->T + (size_t)c0 * 2, p->K2, p->gB[t] + s0, rank, 1, M, Nt, cw,
                                  r_used, p->lora_scale, p->accumulate, p->skinny_ws, 0, 0, 0));
        if (!multi && q.ok())
          q.run(st355_skinny_tn_seg(stream, p->n, D, (const char*)p->U + (size_t)c0 * 2, p->K2, p->gA[t] + (size_t)s0 * D, 1, D, M, D, cw, r_used, 1.0f, p->accumulate,
                                    p->skinny_ws, 0, 0, 0));
      }
    }
    if (multi && q.ok())
      q.run(st355_skinny_tn_multi(stream, p->n, D, p->U, p->K2, (float* const*)p->gA, p->n_targets, 1, D, M, D, rank, 1.0f, p->accumulate, p->skinny_ws, 0, 0, 0));
  }
  // d x = d x' + LN'(d norm_hidden * (1 + scale)); the previous block's gate applied in the same pass when it has one
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn, D, p->x, D, p->mod_scale, p->mod_stride, S, p->dx, D, p->gate_prev, p->gate_prev ? p->mod_stride : 0, p->dx_out, D,
                                p->dxg_out, D, M, D, 1e-6f));
  return q.rc;
}
