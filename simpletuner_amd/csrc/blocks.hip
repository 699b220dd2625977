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

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// FluxTransformerBlock ("double" block, flux/transformer.py:607-687): the image and the text stream use different weights but the same epilogues; every
// pair of projections goes out as ONE grouped launch.  Joint buffers ([B*S, *], sample b = rows [b*S, (b+1)*S), text rows first) are read and written in
// place through segmented-row operands (st355_gemm_args.seg_rows): the image rows of every sample are one problem, the text rows another.
// ---------------------------------------------------------------------------------------------------------------------------------------------------
namespace {
struct Rows { const void* p; int64_t ld; int64_t seg; };                 // seg: physical rows from one sample's block to the next (0 = compact rows)
Rows joint_rows(const void* base, int64_t ld, int lo, int S) { return Rows{(const char*)base + (size_t)lo * ld * 2, ld, S}; }
Rows compact_rows(const void* base, int64_t ld) { return Rows{base, ld, 0}; }
// one stream's problem over `rows` rows per sample: plain 2-D when B == 1, else ONE segmented problem (the host side's FluxTransformer2DModel._problems)
st355_gemm_args GS(int B, int rows, Rows A, const void* W, int64_t ldw, Rows C, int N, int K) {
  st355_gemm_args a = G(A.p, A.ld, W, ldw, (void*)C.p, C.ld, B * rows, N, K);
  if (B > 1) { a.seg_rows = rows; a.seg_a = A.seg; a.seg_c = C.seg; }
  return a;
}
void ext(st355_gemm_args& a, int B, Rows A2, const void* B2, int K2, int k2_real) {
  if (!K2) return;
  a.A2 = A2.p; a.lda2 = A2.ld; a.B2 = B2; a.ldb2 = K2; a.K2 = K2; a.K2_real = k2_real;
  if (B > 1) a.seg_a2 = A2.seg;
}
}  // namespace

extern "C" int st355_block_flux_double_fwd(void* stream, const st355_flux_double_fwd_args* p) {
  ST_REQUIRE(p && p->img && p->txt && p->n_img && p->n_txt && p->V && p->rrms && p->Q && p->K && p->Vt && p->O && p->lse2 && p->x1_img && p->x1_txt && p->hpre_img &&
             p->hpre_txt && p->n2_img && p->n2_txt && p->h_img && p->h_txt, "block_flux_double_fwd: null pointer");
  ST_REQUIRE((p->out_joint != nullptr) != (p->out_img != nullptr && p->out_txt != nullptr), "block_flux_double_fwd: give either the joint output or the two stream outputs");
  ST_REQUIRE(p->B > 0 && p->Si > 0 && p->St > 0 && p->Si % 256 == 0 && p->St % 256 == 0 && p->H > 0 && p->H % 2 == 0 && p->D == p->H * 128,
             "block_flux_double_fwd: built for head_dim 128, even H, both streams a multiple of 256 rows per sample");
  ST_REQUIRE((p->K2_qkv == 0) == (p->A_qkv == nullptr) && (p->K2_qkv == 0 || (p->Bb_qkv && p->T_img)) && (p->K2_out == 0) == (p->A_out == nullptr) &&
             (p->K2_out == 0 || (p->Bb_out && p->T_o)), "block_flux_double_fwd: inconsistent adapter operands");
  const int B = p->B, Si = p->Si, St = p->St, S = Si + St, H = p->H, D = p->D;
  const int64_t ms = p->mod_stride;
  const bf16* mi = (const bf16*)p->mod_img; const bf16* mt = (const bf16*)p->mod_txt;       // [B, 6D] slices: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
  Seq q{stream, 0};
  q.run(st355_ln_modulate_fwd(stream, p->img, D, mi + D, mi, ms, Si, p->n_img, D, (int64_t)B * Si, D, 1e-6f));
  if (q.ok()) q.run(st355_ln_modulate_fwd(stream, p->txt, D, mt + D, mt, ms, St, p->n_txt, D, (int64_t)B * St, D, 1e-6f));
  if (p->K2_qkv) {
    st355_gemm_args t = G(p->n_img, D, p->A_qkv, D, p->T_img, p->K2_qkv, B * Si, p->K2_qkv, D);
    thin_ws(t, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(t);
  }
  // both streams project into the joint [txt || img] sequence: q / k head-major with RMSNorm + RoPE in the epilogue, v rows + V^T (flux/transformer.py:140-207)
  st355_qk_rope ri, rt;
  memset(&ri, 0, sizeof(ri));
  ri.Q = p->Q; ri.K = p->K; ri.rrms = (float*)p->rrms; ri.cos = (const float*)p->cos_p; ri.sin = (const float*)p->sin_p; ri.H = H; ri.S = S; ri.eps = 1e-6f; ri.Vt = p->Vt; ri.Sp = S;
  rt = ri;
  ri.wq = p->norm_q; ri.wk = p->norm_k; ri.pos0 = St;
  rt.wq = p->norm_added_q; rt.wk = p->norm_added_k; rt.pos0 = 0;
  if (q.ok()) {
    st355_gemm_args g2[2];
    g2[0] = GS(B, Si, compact_rows(p->n_img, D), p->w_qkv, D, joint_rows(p->V, D, St, S), 3 * D, D);
    g2[0].bias = p->b_qkv; g2[0].epilogue = ST355_EPI_QK_NORM_ROPE; g2[0].rope = &ri; g2[0].rows_per_batch = Si;
    ext(g2[0], B, compact_rows(p->T_img, p->K2_qkv), p->Bb_qkv, p->K2_qkv, p->k2r_qkv);
    g2[1] = GS(B, St, compact_rows(p->n_txt, D), p->w_add_qkv, D, joint_rows(p->V, D, 0, S), 3 * D, D);
    g2[1].bias = p->b_add_qkv; g2[1].epilogue = ST355_EPI_QK_NORM_ROPE; g2[1].rope = &rt; g2[1].rows_per_batch = St;
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  if (q.ok()) q.run(st355_attn_fwd(stream, p->Q, p->K, p->Vt, (const float*)p->key_bias, p->O, D, (float*)p->lse2, B, H, S, S, 128, p->scale));
  // attention output projections, gated onto the two residual streams (the attention output is split back by rows, in place)
  const Rows O_i = joint_rows(p->O, D, St, S), O_t = joint_rows(p->O, D, 0, S);
  if (p->K2_out) {
    st355_gemm_args t = GS(B, Si, O_i, p->A_out, D, compact_rows(p->T_o, p->K2_out), p->K2_out, D);
    thin_ws(t, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(t);
  }
  if (q.ok()) {
    st355_gemm_args g2[2];
    g2[0] = GS(B, Si, O_i, p->w_out, D, compact_rows(p->x1_img, D), D, D);
    g2[0].bias = p->b_out; g2[0].epilogue = ST355_EPI_GATE_RESIDUAL; g2[0].aux_in = p->img; g2[0].ld_aux_in = D; g2[0].gate = mi + 2 * D; g2[0].gate_stride = ms;
    g2[0].rows_per_batch = Si;
    ext(g2[0], B, compact_rows(p->T_o, p->K2_out), p->Bb_out, p->K2_out, p->k2r_out);
    g2[1] = GS(B, St, O_t, p->w_add_out, D, compact_rows(p->x1_txt, D), D, D);
    g2[1].bias = p->b_add_out; g2[1].epilogue = ST355_EPI_GATE_RESIDUAL; g2[1].aux_in = p->txt; g2[1].ld_aux_in = D; g2[1].gate = mt + 2 * D; g2[1].gate_stride = ms;
    g2[1].rows_per_batch = St;
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  // the two MLPs
  if (q.ok()) q.run(st355_ln_modulate_fwd(stream, p->x1_img, D, mi + 4 * D, mi + 3 * D, ms, Si, p->n2_img, D, (int64_t)B * Si, D, 1e-6f));
  if (q.ok()) q.run(st355_ln_modulate_fwd(stream, p->x1_txt, D, mt + 4 * D, mt + 3 * D, ms, St, p->n2_txt, D, (int64_t)B * St, D, 1e-6f));
  if (q.ok()) {
    st355_gemm_args g2[2];
    g2[0] = G(p->n2_img, D, p->w_ff1, D, p->h_img, 4 * D, B * Si, 4 * D, D);
    g2[0].bias = p->b_ff1; g2[0].epilogue = ST355_EPI_GELU; g2[0].aux_out = p->hpre_img; g2[0].ld_aux_out = 4 * D;
    g2[1] = G(p->n2_txt, D, p->w_ffc1, D, p->h_txt, 4 * D, B * St, 4 * D, D);
    g2[1].bias = p->b_ffc1; g2[1].epilogue = ST355_EPI_GELU; g2[1].aux_out = p->hpre_txt; g2[1].ld_aux_out = 4 * D;
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  if (q.ok()) {
    st355_gemm_args g2[2];
    if (p->out_joint) {      // the LAST double block writes the joint [txt || img] sequence of the single blocks in place (flux/transformer.py:1332 `torch.cat`)
      g2[0] = GS(B, Si, compact_rows(p->h_img, 4 * D), p->w_ff2, 4 * D, joint_rows(p->out_joint, D, St, S), D, 4 * D);
      g2[1] = GS(B, St, compact_rows(p->h_txt, 4 * D), p->w_ffc2, 4 * D, joint_rows(p->out_joint, D, 0, S), D, 4 * D);
    } else {
      g2[0] = G(p->h_img, 4 * D, p->w_ff2, 4 * D, p->out_img, D, B * Si, D, 4 * D);
      g2[1] = G(p->h_txt, 4 * D, p->w_ffc2, 4 * D, p->out_txt, D, B * St, D, 4 * D);
    }
    g2[0].bias = p->b_ff2; g2[0].epilogue = ST355_EPI_GATE_RESIDUAL; g2[0].aux_in = p->x1_img; g2[0].ld_aux_in = D; g2[0].gate = mi + 5 * D; g2[0].gate_stride = ms;
    g2[0].rows_per_batch = Si;
    g2[1].bias = p->b_ffc2; g2[1].epilogue = ST355_EPI_GATE_RESIDUAL; g2[1].aux_in = p->x1_txt; g2[1].ld_aux_in = D; g2[1].gate = mt + 5 * D; g2[1].gate_stride = ms;
    g2[1].rows_per_batch = St;
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  return q.rc;
}

namespace {
// LoraGroup.grads of the host side for ONE stream's projection group: dB_t = s dy_t^T T_t, dA_t = U_t^T x.  dy: the projection gradient (columns t * Nt),
// x: the projection input; either may be rows of a joint buffer (seg != 0, B > 1).
void lora_grads(Seq& q, int B, int rows, Rows dy, int Nt, Rows x, int K, const void* T, const void* U, int K2, int n_targets, int rank, int r_pad, float s,
                float* const* gA, float* const* gB, int accumulate, void* ws) {
  const int cw = r_pad < 64 ? r_pad : 64;
  const bool multi = n_targets > 1 && r_pad == 32 && K2 >= 128;
  const int64_t M = (int64_t)B * rows;
  const int64_t seg = B > 1 && (dy.seg || x.seg) ? rows : 0;
  for (int t = 0; t < n_targets && q.ok(); t++)
    for (int s0 = 0; s0 < rank && q.ok(); s0 += cw) {
      const int c0 = t * r_pad + s0, r_used = (rank - s0) < cw ? (rank - s0) : cw;
      q.run(st355_skinny_tn_seg(q.st, (const char*)dy.p + (size_t)t * Nt * 2, dy.ld, (const char*)T + (size_t)c0 * 2, K2, gB[t] + s0, rank, 1, M, Nt, cw, r_used, s,
                                accumulate, ws, (B > 1 && dy.seg) ? rows : 0, (B > 1) ? dy.seg : 0, 0));
      if (!multi && q.ok())
        q.run(st355_skinny_tn_seg(q.st, x.p, x.ld, (const char*)U + (size_t)c0 * 2, K2, gA[t] + (size_t)s0 * K, 1, K, M, K, cw, r_used, 1.0f, accumulate, ws,
                                  (B > 1 && x.seg) ? rows : 0, (B > 1) ? x.seg : 0, 0));
    }
  if (multi && q.ok())
    q.run(st355_skinny_tn_multi(q.st, x.p, x.ld, U, K2, gA, n_targets, 1, K, M, K, rank, 1.0f, accumulate, ws, (B > 1 && x.seg) ? rows : 0, (B > 1) ? x.seg : 0, 0));
  (void)seg;
}
}  // namespace

// backward of a double block that is NOT the first one (its input gradients are needed; block 0's embedders are frozen and the host keeps that special case)
extern "C" int st355_block_flux_double_bwd(void* stream, const st355_flux_double_bwd_args* p) {
  ST_REQUIRE(p && p->img && p->txt && p->n_img && p->V && p->rrms && p->Q && p->K && p->O && p->lse2 && p->x1_img && p->x1_txt && p->hpre_img && p->hpre_txt &&
             p->d_img && p->d_txt && p->d_img_out && p->d_txt_out, "block_flux_double_bwd: null pointer");
  ST_REQUIRE(p->g_img && p->g_txt && p->dh_img && p->dh_txt && p->dn2_img && p->dn2_txt && p->dx1_img && p->dx1g_img && p->dx1_txt && p->dx1g_txt && p->dO && p->dqkv &&
             p->dn_img && p->dn_txt && p->attn_ws, "block_flux_double_bwd: null scratch pointer");
  ST_REQUIRE(p->B > 0 && p->Si > 0 && p->St > 0 && p->Si % 256 == 0 && p->St % 256 == 0 && p->H > 0 && p->H % 2 == 0 && p->D == p->H * 128,
             "block_flux_double_bwd: built for head_dim 128, even H, both streams a multiple of 256 rows per sample");
  ST_REQUIRE((p->K2_qkv == 0) == (p->At_qkv == nullptr) && (p->K2_qkv == 0 || (p->Bbt_qkv && p->T_img && p->U_qkv && p->skinny_ws)) &&
             (p->K2_out == 0) == (p->At_out == nullptr) && (p->K2_out == 0 || (p->Bbt_out && p->T_o && p->U_out && p->skinny_ws)), "block_flux_double_bwd: inconsistent adapter operands");
  const int B = p->B, Si = p->Si, St = p->St, S = Si + St, H = p->H, D = p->D;
  const int64_t ms = p->mod_stride, Mi = (int64_t)B * Si, Mt = (int64_t)B * St;
  const bf16* mi = (const bf16*)p->mod_img; const bf16* mt = (const bf16*)p->mod_txt;
  Seq q{stream, 0};
  // ---- the two MLPs ----
  q.run(st355_scale_cols(stream, p->d_img, D, mi + 5 * D, ms, Si, p->g_img, D, Mi, D));
  if (q.ok()) q.run(st355_scale_cols(stream, p->d_txt, D, mt + 5 * D, ms, St, p->g_txt, D, Mt, D));
  if (q.ok()) {
    st355_gemm_args g2[2];
    g2[0] = G(p->g_img, D, p->wT_ff2, D, p->dh_img, 4 * D, (int)Mi, 4 * D, D);
    g2[0].epilogue = ST355_EPI_MUL_GELU_GRAD; g2[0].aux_in = p->hpre_img; g2[0].ld_aux_in = 4 * D;
    g2[1] = G(p->g_txt, D, p->wT_ffc2, D, p->dh_txt, 4 * D, (int)Mt, 4 * D, D);
    g2[1].epilogue = ST355_EPI_MUL_GELU_GRAD; g2[1].aux_in = p->hpre_txt; g2[1].ld_aux_in = 4 * D;
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  if (q.ok()) {
    st355_gemm_args g2[2];
    g2[0] = G(p->dh_img, 4 * D, p->wT_ff1, 4 * D, p->dn2_img, D, (int)Mi, D, 4 * D);
    g2[1] = G(p->dh_txt, 4 * D, p->wT_ffc1, 4 * D, p->dn2_txt, D, (int)Mt, D, 4 * D);
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn2_img, D, p->x1_img, D, mi + 4 * D, ms, Si, p->d_img, D, mi + 2 * D, ms, p->dx1_img, D, p->dx1g_img, D, Mi, D, 1e-6f));
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn2_txt, D, p->x1_txt, D, mt + 4 * D, ms, St, p->d_txt, D, mt + 2 * D, ms, p->dx1_txt, D, p->dx1g_txt, D, Mt, D, 1e-6f));
  // ---- attention output projections -> dO rows of both streams (+ the to_out.0 adapter gradients) ----
  if (p->K2_out) {
    st355_gemm_args u = G(p->dx1g_img, D, p->Bbt_out, D, p->U_out, p->K2_out, (int)Mi, p->K2_out, D);
    thin_ws(u, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(u);
  }
  if (q.ok()) {
    st355_gemm_args g2[2];
    g2[0] = GS(B, Si, compact_rows(p->dx1g_img, D), p->wT_out, D, joint_rows(p->dO, D, St, S), D, D);
    ext(g2[0], B, compact_rows(p->U_out, p->K2_out), p->At_out, p->K2_out, p->k2r_out);
    g2[1] = GS(B, St, compact_rows(p->dx1g_txt, D), p->wT_add_out, D, joint_rows(p->dO, D, 0, S), D, D);
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  if (p->K2_out && q.ok())
    lora_grads(q, B, Si, compact_rows(p->dx1g_img, D), D, joint_rows(p->O, D, St, S), D, p->T_o, p->U_out, p->K2_out, 1, p->rank_out, p->rpad_out, p->scale_out,
               p->gA_out, p->gB_out, p->accumulate, p->skinny_ws);
  // ---- attention (+ RoPE / RMSNorm backward in its epilogues): text positions < St carry the norm_added weights ----
  if (q.ok())
    q.run(st355_attn_bwd_rope(stream, p->Q, p->K, p->V, D, p->O, D, p->dO, D, (const float*)p->lse2, (const float*)p->key_bias, (const float*)p->rrms, p->norm_added_q,
                              p->norm_added_k, p->norm_q, p->norm_k, St, (const float*)p->cos_p, (const float*)p->sin_p, p->dqkv, 3 * D, B, H, S, S, 128, p->scale, p->attn_ws));
  // ---- input projections: the two streams' rows of the joint dqkv, in place ----
  const Rows dq_i = joint_rows(p->dqkv, 3 * D, St, S), dq_t = joint_rows(p->dqkv, 3 * D, 0, S);
  if (p->K2_qkv) {
    st355_gemm_args u = GS(B, Si, dq_i, p->Bbt_qkv, 3 * D, compact_rows(p->U_qkv, p->K2_qkv), p->K2_qkv, 3 * D);
    thin_ws(u, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(u);
  }
  if (q.ok()) {
    st355_gemm_args g2[2];
    g2[0] = GS(B, Si, dq_i, p->wT_qkv, 3 * D, compact_rows(p->dn_img, D), D, 3 * D);
    ext(g2[0], B, compact_rows(p->U_qkv, p->K2_qkv), p->At_qkv, p->K2_qkv, p->k2r_qkv);
    g2[1] = GS(B, St, dq_t, p->wT_add_qkv, 3 * D, compact_rows(p->dn_txt, D), D, 3 * D);
    q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  if (p->K2_qkv && q.ok())
    lora_grads(q, B, Si, dq_i, D, compact_rows(p->n_img, D), D, p->T_img, p->U_qkv, p->K2_qkv, 3, p->rank_qkv, p->rpad_qkv, p->scale_qkv, p->gA_qkv, p->gB_qkv,
               p->accumulate, p->skinny_ws);
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn_img, D, p->img, D, mi + D, ms, Si, p->dx1_img, D, nullptr, 0, p->d_img_out, D, nullptr, D, Mi, D, 1e-6f));
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn_txt, D, p->txt, D, mt + D, ms, St, p->dx1_txt, D, nullptr, 0, p->d_txt_out, D, nullptr, D, Mt, D, 1e-6f));
  return q.rc;
}

// =====================================================================================================================================================
// PixArt BasicTransformerBlock(ada_norm_single): reference seam helpers/models/pixart/transformer.py:95-145 (`_pixart_apply_block`) and the autograd
// through it.  Same launches, order and operands as simpletuner_amd/pixart/transformer.py `_block_fwd` / `_block_bwd` (which calls these by default).
// =====================================================================================================================================================
extern "C" int st355_block_pixart_fwd(void* stream, const st355_pixart_block_fwd_args* p) {
  ST_REQUIRE(p && p->h && p->ctx && p->mod && p->n1 && p->qkv && p->Q && p->K && p->O && p->lse && p->h1 && p->q2 && p->kv && p->Q2 && p->K2 && p->O2 && p->lse_x
                 && p->h2 && p->n2 && p->act && p->Vt && p->V2t && p->out, "block_pixart_fwd: null pointer");
  ST_REQUIRE(p->w_qkv && p->w_out1 && p->w_q2 && p->w_kv2 && p->w_out2 && p->w_ff1 && p->w_ff2, "block_pixart_fwd: null weight pointer");
  ST_REQUIRE(p->B > 0 && p->S > 0 && p->Sk > 0 && p->H > 0 && p->D > 0 && p->D % 64 == 0 && (p->d_pad == 64 || p->d_pad == 96 || p->d_pad == 128),
             "block_pixart_fwd: D %% 64 == 0 and a padded head width of 64 / 96 / 128");
  const int B = p->B, S = p->S, Sk = p->Sk, H = p->H, D = p->D, hd = p->d_pad, Dp = H * hd, M = B * S, Mk = B * Sk;
  const int Sp = (S + 63) / 64 * 64, Skp = (Sk + 63) / 64 * 64;
  const int64_t ms = p->mod_stride;
  const char* m = (const char*)p->mod;
  auto chunk = [&](int k) { return (const void*)(m + (size_t)k * D * 2); };
  char* qkv = (char*)p->qkv; char* kv = (char*)p->kv;
  Seq q{stream, 0};
  // norm_hidden = LN(h) * (1 + scale_msa) + shift_msa;  q | k | v                                              (pixart/transformer.py:107-113)
  q.run(st355_ln_modulate_fwd(stream, p->h, D, chunk(1), chunk(0), ms, S, p->n1, D, M, D, 1e-6f));
  {
    st355_gemm_args a = G(p->n1, D, p->w_qkv, D, p->qkv, 3 * Dp, M, 3 * Dp, D);
    a.bias = p->b_qkv;
    q.gemm(a);
  }
  if (q.ok()) q.run(st355_head_split_pad(stream, qkv, 3 * Dp, p->Q, nullptr, B, H, hd, hd, S, Sp));
  if (q.ok()) q.run(st355_head_split_pad(stream, qkv + (size_t)Dp * 2, 3 * Dp, p->K, nullptr, B, H, hd, hd, S, Sp));
  if (q.ok()) q.run(st355_head_split_pad(stream, qkv + (size_t)2 * Dp * 2, 3 * Dp, nullptr, p->Vt, B, H, hd, hd, S, Sp));
  if (q.ok()) q.run(st355_attn_fwd(stream, p->Q, p->K, p->Vt, nullptr, p->O, Dp, (float*)p->lse, B, H, S, Sp, hd, p->scale));
  // h1 = h + gate_msa * (attn W_o^T + b)                                                                        (:114-115)
  {
    st355_gemm_args a = G(p->O, Dp, p->w_out1, Dp, p->h1, D, M, D, Dp);
    a.bias = p->b_out1; a.epilogue = ST355_EPI_GATE_RESIDUAL; a.gate = chunk(2); a.gate_stride = ms; a.rows_per_batch = S; a.aux_in = p->h; a.ld_aux_in = D;
    if (p->ya) { a.aux_out = p->ya; a.ld_aux_out = D; }
    q.gemm(a);
  }
  // cross-attention over the caption tokens: no pre-norm, no gate                                               (:117-129)
  {
    st355_gemm_args a = G(p->h1, D, p->w_q2, D, p->q2, Dp, M, Dp, D);
    a.bias = p->b_q2;
    q.gemm(a);
    st355_gemm_args b = G(p->ctx, D, p->w_kv2, D, p->kv, 2 * Dp, Mk, 2 * Dp, D);
    b.bias = p->b_kv2;
    q.gemm(b);
  }
  if (q.ok()) q.run(st355_head_split_pad(stream, p->q2, Dp, p->Q2, nullptr, B, H, hd, hd, S, Sp));
  if (q.ok()) q.run(st355_head_split_pad(stream, kv, 2 * Dp, p->K2, nullptr, B, H, hd, hd, Sk, Skp));
  if (q.ok()) q.run(st355_head_split_pad(stream, kv + (size_t)Dp * 2, 2 * Dp, nullptr, p->V2t, B, H, hd, hd, Sk, Skp));
  if (q.ok())
    q.run(st355_attn_cross_fwd(stream, p->Q2, p->K2, p->V2t, (const float*)p->key_bias, p->O2, Dp, (float*)p->lse_x, B, H, S, Sk, Skp, hd, p->scale));
  {
    st355_gemm_args a = G(p->O2, Dp, p->w_out2, Dp, p->h2, D, M, D, Dp);
    a.bias = p->b_out2; a.epilogue = ST355_EPI_ADD; a.aux_in = p->h1; a.ld_aux_in = D;
    q.gemm(a);
  }
  // feed-forward: LN(h2) * (1 + scale_mlp) + shift_mlp -> GELU(tanh) MLP -> gated residual                      (:131-143)
  if (q.ok()) q.run(st355_ln_modulate_fwd(stream, p->h2, D, chunk(4), chunk(3), ms, S, p->n2, D, M, D, 1e-6f));
  {
    st355_gemm_args a = G(p->n2, D, p->w_ff1, D, p->act, 4 * D, M, 4 * D, D);
    a.bias = p->b_ff1; a.epilogue = ST355_EPI_GELU;
    if (p->pre) { a.aux_out = p->pre; a.ld_aux_out = 4 * D; }
    q.gemm(a);
  }
  {
    st355_gemm_args a = G(p->act, 4 * D, p->w_ff2, 4 * D, p->out, D, M, D, 4 * D);
    a.bias = p->b_ff2; a.epilogue = ST355_EPI_GATE_RESIDUAL; a.gate = chunk(5); a.gate_stride = ms; a.rows_per_batch = S; a.aux_in = p->h2; a.ld_aux_in = D;
    if (p->yf) { a.aux_out = p->yf; a.ld_aux_out = D; }
    q.gemm(a);
  }
  return q.rc;
}

extern "C" int st355_block_pixart_bwd(void* stream, const st355_pixart_block_bwd_args* p) {
  ST_REQUIRE(p && p->h && p->mod && p->qkv && p->Q && p->K && p->O && p->lse && p->q2 && p->kv && p->Q2 && p->K2 && p->O2 && p->lse_x && p->h2 && p->pre && p->d_out,
             "block_pixart_bwd: null pointer");
  ST_REQUIRE(p->wT_qkv && p->wT_out1 && p->wT_q2 && p->wT_out2 && p->wT_ff1 && p->wT_ff2, "block_pixart_bwd: null weight pointer");
  ST_REQUIRE(p->dyf && p->dpre && p->dn2 && p->d2 && p->dO2 && p->dq2 && p->dkv && p->d1 && p->dya && p->dO && p->dqkv && p->dn1 && p->dQ && p->dK && p->attn_ws && p->d_in,
             "block_pixart_bwd: null scratch pointer");
  ST_REQUIRE(p->B > 0 && p->S > 0 && p->Sk > 0 && p->H > 0 && p->D > 0 && p->D % 64 == 0 && (p->d_pad == 64 || p->d_pad == 96 || p->d_pad == 128),
             "block_pixart_bwd: D %% 64 == 0 and a padded head width of 64 / 96 / 128");
  const int B = p->B, S = p->S, Sk = p->Sk, H = p->H, D = p->D, hd = p->d_pad, Dp = H * hd, M = B * S;
  const int Sp = (S + 63) / 64 * 64, Skp = (Sk + 63) / 64 * 64;
  const int64_t ms = p->mod_stride;
  const char* m = (const char*)p->mod;
  auto chunk = [&](int k) { return (const void*)(m + (size_t)k * D * 2); };
  const char* qkv = (const char*)p->qkv; const char* kv = (const char*)p->kv;
  char* dqkv = (char*)p->dqkv; char* dkv = (char*)p->dkv;
  Seq q{stream, 0};
  // ---- feed-forward ----
  q.run(st355_scale_cols(stream, p->d_out, D, chunk(5), ms, S, p->dyf, D, M, D));
  {
    st355_gemm_args a = G(p->dyf, D, p->wT_ff2, D, p->dpre, 4 * D, M, 4 * D, D);
    a.epilogue = ST355_EPI_MUL_GELU_GRAD; a.aux_in = p->pre; a.ld_aux_in = 4 * D;
    q.gemm(a);
  }
  q.gemm(G(p->dpre, 4 * D, p->wT_ff1, 4 * D, p->dn2, D, M, D, 4 * D));
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn2, D, p->h2, D, chunk(4), ms, S, p->d_out, D, nullptr, 0, p->d2, D, nullptr, D, M, D, 1e-6f));
  // ---- cross-attention (no pre-norm, no gate) ----
  q.gemm(G(p->d2, D, p->wT_out2, D, p->dO2, Dp, M, Dp, D));
  if (q.ok())
    q.run(st355_attn_cross_bwd(stream, p->Q2, p->K2, nullptr, nullptr, kv + (size_t)Dp * 2, 2 * Dp, p->O2, Dp, p->dO2, Dp, (const float*)p->lse_x,
                               (const float*)p->key_bias, p->dQ, p->dK, dkv + (size_t)Dp * 2, 2 * Dp, B, H, S, Sp, Sk, Skp, hd, p->scale, p->attn_ws));
  if (q.ok()) q.run(st355_head_merge_pad(stream, p->dQ, p->dq2, Dp, B, H, hd, hd, S));
  if (q.ok()) q.run(st355_head_merge_pad(stream, p->dK, dkv, 2 * Dp, B, H, hd, hd, Sk));
  {
    st355_gemm_args a = G(p->dq2, Dp, p->wT_q2, Dp, p->d1, D, M, D, Dp);
    a.epilogue = ST355_EPI_ADD; a.aux_in = p->d2; a.ld_aux_in = D;
    q.gemm(a);
  }
  // ---- self-attention ----
  if (q.ok()) q.run(st355_scale_cols(stream, p->d1, D, chunk(2), ms, S, p->dya, D, M, D));
  q.gemm(G(p->dya, D, p->wT_out1, D, p->dO, Dp, M, Dp, D));
  if (q.ok())
    q.run(st355_attn_bwd(stream, p->Q, p->K, nullptr, nullptr, qkv + (size_t)2 * Dp * 2, 3 * Dp, p->O, Dp, p->dO, Dp, (const float*)p->lse, nullptr, p->dQ, p->dK,
                         dqkv + (size_t)2 * Dp * 2, 3 * Dp, B, H, S, Sp, hd, p->scale, p->attn_ws));
  if (q.ok()) q.run(st355_head_merge_pad(stream, p->dQ, dqkv, 3 * Dp, B, H, hd, hd, S));
  if (q.ok()) q.run(st355_head_merge_pad(stream, p->dK, dqkv + (size_t)Dp * 2, 3 * Dp, B, H, hd, hd, S));
  q.gemm(G(p->dqkv, 3 * Dp, p->wT_qkv, 3 * Dp, p->dn1, D, M, D, 3 * Dp));
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn1, D, p->h, D, chunk(1), ms, S, p->d1, D, nullptr, 0, p->d_in, D, nullptr, D, M, D, 1e-6f));
  return q.rc;
}

// =====================================================================================================================================================
// SD3 JointTransformerBlock: reference seam helpers/models/sd3/transformer.py:145-241 (`_sd3_apply_joint_transformer_block`) and the autograd through it.
// Same launches, order and operands as simpletuner_amd/sd3/transformer.py `_block_fwd` / the block body of `_engine_backward` (which call these by default).
// Joint buffers hold [img || txt] per sample.  A stream's row block is ONE problem when B == 1 or the block is tile-aligned (segmented rows); other blocks
// follow `_stream_problems` of the host side: >= 1024 rows -> one problem per sample, fewer -> compact copies (gathered before / scattered after the launch).
// =====================================================================================================================================================
#include <vector>

namespace {
enum StreamMode { SM_SINGLE, SM_PER_SAMPLE, SM_COMPACT };
StreamMode stream_mode(int B, int rows) { return (B == 1 || rows % 256 == 0) ? SM_SINGLE : (rows >= 1024 ? SM_PER_SAMPLE : SM_COMPACT); }

struct Scatter { void* dst; size_t dpitch; const void* src; size_t spitch; size_t width; int B; };
int copy2d(void* st, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int B) {
  return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)st) == hipSuccess ? 0 : 1;
}
// rows of a joint buffer -> compact [B * rows, ld] (what `t.reshape(B * rows, -1)` of a [B, rows, C] view copies on the host side)
int gather_block(void* st, void* dst, Rows src, int rows, int B) {
  return copy2d(st, dst, (size_t)rows * src.ld * 2, src.p, (size_t)src.seg * src.ld * 2, (size_t)rows * src.ld * 2, B);
}

struct StreamOps { Rows A, A2, C, In, Out; };            // .p == nullptr: absent
// Append one stream's projection (row operands in `o`, everything else in `proto`) to `v`.  COMPACT mode: joint inputs are gathered into `scratch` now (A only —
// the call sites never have two joint inputs), a joint output goes to `scratch` and is scattered back by the entry appended to `sc` (run after the launch).
void stream_probs(Seq& q, std::vector<st355_gemm_args>& v, std::vector<Scatter>& sc, int B, int rows, const st355_gemm_args& proto, StreamOps o, void* scratch) {
  const StreamMode mode = stream_mode(B, rows);
  auto fill = [&](st355_gemm_args& a, const StreamOps& s, int64_t off_rows_scale) {
    (void)off_rows_scale;
    a.A = s.A.p; a.lda = s.A.ld; a.C = (void*)s.C.p; a.ldc = s.C.ld;
    if (s.A2.p) { a.A2 = s.A2.p; a.lda2 = s.A2.ld; }
    if (s.In.p) { a.aux_in = s.In.p; a.ld_aux_in = s.In.ld; }
    if (s.Out.p) { a.aux_out = (void*)s.Out.p; a.ld_aux_out = s.Out.ld; }
  };
  if (mode == SM_COMPACT) {
    if (o.A.seg) {
      q.run(gather_block(q.st, scratch, o.A, rows, B));
      o.A = compact_rows(scratch, o.A.ld);
    } else if (o.C.seg) {
      sc.push_back(Scatter{(void*)o.C.p, (size_t)o.C.seg * o.C.ld * 2, scratch, (size_t)rows * o.C.ld * 2, (size_t)rows * o.C.ld * 2, B});
      o.C = compact_rows(scratch, o.C.ld);
    }
  }
  if (mode == SM_SINGLE || mode == SM_COMPACT) {
    st355_gemm_args a = proto;
    fill(a, o, 0);
    a.M = B * rows;
    if (B > 1 && mode == SM_SINGLE) { a.seg_rows = rows; a.seg_a = o.A.seg; a.seg_a2 = o.A2.seg; a.seg_c = o.C.seg; a.seg_in = o.In.seg; a.seg_out = o.Out.seg; }
    v.push_back(a);
    return;
  }
  for (int b = 0; b < B; b++) {
    auto at = [&](Rows r) { return r.p ? Rows{(const char*)r.p + (size_t)b * (r.seg ? r.seg : rows) * r.ld * 2, r.ld, 0} : r; };
    st355_gemm_args a = proto;
    fill(a, StreamOps{at(o.A), at(o.A2), at(o.C), at(o.In), at(o.Out)}, 0);
    a.M = rows;
    if (a.gate) a.gate = (const char*)a.gate + (size_t)b * a.gate_stride * 2;
    v.push_back(a);
  }
}
void run_scatters(Seq& q, std::vector<Scatter>& sc) {
  for (const Scatter& s : sc)
    if (q.ok()) q.run(copy2d(q.st, s.dst, s.dpitch, s.src, s.spitch, s.width, s.B));
  sc.clear();
}
void run_grouped(Seq& q, std::vector<st355_gemm_args>& v) {
  if (q.ok() && !v.empty()) q.run(st355_gemm_bf16_grouped(q.st, v.data(), (int)v.size()));
  v.clear();
}
st355_gemm_args proto_of(const void* W, int64_t ldw, int N, int K) {
  st355_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.B = W; a.ldb = ldw; a.N = N; a.K = K; a.epilogue = ST355_EPI_NONE;
  return a;
}
void proto_ext(st355_gemm_args& a, const void* B2, int K2, int k2_real) {
  if (!K2) return;
  a.B2 = B2; a.ldb2 = K2; a.K2 = K2; a.K2_real = k2_real;
}
}  // namespace

extern "C" int st355_block_sd3_joint_fwd(void* stream, const st355_sd3_joint_fwd_args* p) {
  ST_REQUIRE(p && p->img && p->txt && p->mod_img && p->mod_txt && p->n_img && p->n_txt && p->qkv && p->Q && p->K && p->O && p->lse2 && p->x1_img && p->hpre_img &&
             p->n2_img && p->h_img && p->Vt && p->out_img && p->cos && p->sin, "block_sd3_joint_fwd: null pointer");
  ST_REQUIRE(p->last || (p->x1_txt && p->hpre_txt && p->n2_txt && p->h_txt && p->out_txt), "block_sd3_joint_fwd: null text-stream pointer");
  ST_REQUIRE(p->B > 0 && p->Si > 0 && p->St > 0 && p->H > 0 && (p->hd == 64 || p->hd == 128) && p->D == p->H * p->hd, "block_sd3_joint_fwd: D = H * hd, hd 64 / 128");
  ST_REQUIRE((p->K2_qkv == 0 || (p->A_qkv && p->Bb_qkv && p->T_img)) && (p->K2_aqkv == 0 || (p->A_aqkv && p->Bb_aqkv && p->T_txt)) &&
             (p->K2_out == 0 || (p->A_out && p->Bb_out && p->T_o)) && (p->K2_aout == 0 || p->last || (p->A_aout && p->Bb_aout && p->T_ao)),
             "block_sd3_joint_fwd: inconsistent adapter operands");
  const int B = p->B, Si = p->Si, St = p->St, S = Si + St, Sp = (S + 63) / 64 * 64, H = p->H, D = p->D, hd = p->hd;
  const bool last = p->last != 0;
  ST_REQUIRE((stream_mode(B, Si) != SM_COMPACT || p->c_img) && (stream_mode(B, St) != SM_COMPACT || p->c_txt), "block_sd3_joint_fwd: a stream needs its compact-copy scratch");
  const int64_t ms = p->mod_stride;
  const bf16* mi = (const bf16*)p->mod_img; const bf16* mt = (const bf16*)p->mod_txt;
  Seq q{stream, 0};
  std::vector<st355_gemm_args> v;
  std::vector<Scatter> sc;
  // norm_hidden = LN(x) * (1 + scale_msa) + shift_msa; the context_pre_only block's text norm is AdaLayerNormContinuous: (scale, shift)      (:150-160)
  q.run(st355_ln_modulate_fwd(stream, p->img, D, mi + D, mi, ms, Si, p->n_img, D, (int64_t)B * Si, D, 1e-6f));
  if (q.ok()) q.run(st355_ln_modulate_fwd(stream, p->txt, D, last ? mt : mt + D, last ? mt + D : mt, ms, St, p->n_txt, D, (int64_t)B * St, D, 1e-6f));
  if (p->K2_qkv) {
    st355_gemm_args t = G(p->n_img, D, p->A_qkv, D, p->T_img, p->K2_qkv, B * Si, p->K2_qkv, D);
    thin_ws(t, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(t);
  }
  if (p->K2_aqkv) {
    st355_gemm_args t = G(p->n_txt, D, p->A_aqkv, D, p->T_txt, p->K2_aqkv, B * St, p->K2_aqkv, D);
    thin_ws(t, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(t);
  }
  // q | k | v of both streams into the joint [img || txt] rows (+ adapters in the K-extension)                                                (:162-176)
  {
    st355_gemm_args a = proto_of(p->w_qkv, D, 3 * D, D);
    a.bias = p->b_qkv; proto_ext(a, p->Bb_qkv, p->K2_qkv, p->k2r_qkv);
    stream_probs(q, v, sc, B, Si, a, StreamOps{compact_rows(p->n_img, D), p->K2_qkv ? compact_rows(p->T_img, p->K2_qkv) : Rows{nullptr, 0, 0},
                                                 joint_rows(p->qkv, 3 * D, 0, S), Rows{nullptr, 0, 0}, Rows{nullptr, 0, 0}}, p->c_img);
    st355_gemm_args t = proto_of(p->w_add_qkv, D, 3 * D, D);
    t.bias = p->b_add_qkv; proto_ext(t, p->Bb_aqkv, p->K2_aqkv, p->k2r_aqkv);
    stream_probs(q, v, sc, B, St, t, StreamOps{compact_rows(p->n_txt, D), p->K2_aqkv ? compact_rows(p->T_txt, p->K2_aqkv) : Rows{nullptr, 0, 0},
                                                 joint_rows(p->qkv, 3 * D, Si, S), Rows{nullptr, 0, 0}, Rows{nullptr, 0, 0}}, p->c_txt);
    run_grouped(q, v);
    run_scatters(q, sc);
  }
  // per-head RMSNorm of q / k (SD3.5), head-major re-layout, V^T                                                                               (:178-186)
  if (q.ok())
    q.run(st355_qk_norm_rope_fwd(stream, p->qkv, 3 * D, p->norm_q, p->norm_k, (const float*)p->cos, (const float*)p->sin, p->Q, p->K, nullptr, nullptr, p->Vt, B, H, hd, Si, 0,
                                 S, Sp, 1e-6f));
  if (q.ok())
    q.run(st355_qk_norm_rope_fwd(stream, p->qkv, 3 * D, p->norm_added_q, p->norm_added_k, (const float*)p->cos, (const float*)p->sin, p->Q, p->K, nullptr, nullptr, p->Vt, B,
                                 H, hd, St, Si, S, Sp, 1e-6f));
  if (q.ok()) q.run(st355_attn_fwd(stream, p->Q, p->K, p->Vt, nullptr, p->O, D, (float*)p->lse2, B, H, S, Sp, hd, p->scale));
  // attention output projections, gated onto the residual streams                                                                               (:199-214)
  Rows O_i = joint_rows(p->O, D, 0, S), O_t = joint_rows(p->O, D, Si, S);
  if (B > 1 && St % 256 != 0) {                                            // the text rows of the attention output, gathered once for both uses below
    if (q.ok()) q.run(gather_block(stream, p->c_txt, O_t, St, B));
    O_t = compact_rows(p->c_txt, D);
  }
  if (stream_mode(B, Si) == SM_COMPACT) {                                  // (the host side gathers inside each use; the bytes are the same)
    if (q.ok()) q.run(gather_block(stream, p->c_img, O_i, Si, B));
    O_i = compact_rows(p->c_img, D);
  }
  const Rows none{nullptr, 0, 0};
  if (p->K2_out) {                                                         // thin adapter down-projections: one launch per problem, as on the host side
    st355_gemm_args a = proto_of(p->A_out, D, p->K2_out, D);
    stream_probs(q, v, sc, B, Si, a, StreamOps{O_i, none, compact_rows(p->T_o, p->K2_out), none, none}, nullptr);
    for (auto& g : v) { thin_ws(g, p->gemm_ws, p->gemm_ws_bytes); q.gemm(g); }
    v.clear();
  }
  if (!last && p->K2_aout) {
    st355_gemm_args a = proto_of(p->A_aout, D, p->K2_aout, D);
    stream_probs(q, v, sc, B, St, a, StreamOps{O_t, none, compact_rows(p->T_ao, p->K2_aout), none, none}, nullptr);
    for (auto& g : v) { thin_ws(g, p->gemm_ws, p->gemm_ws_bytes); q.gemm(g); }
    v.clear();
  }
  {
    st355_gemm_args a = proto_of(p->w_out, D, D, D);
    a.bias = p->b_out; a.epilogue = ST355_EPI_GATE_RESIDUAL; a.gate = mi + 2 * D; a.gate_stride = ms; a.rows_per_batch = Si; proto_ext(a, p->Bb_out, p->K2_out, p->k2r_out);
    stream_probs(q, v, sc, B, Si, a, StreamOps{O_i, p->K2_out ? compact_rows(p->T_o, p->K2_out) : none, compact_rows(p->x1_img, D), compact_rows(p->img, D),
                                                 p->ya_img ? compact_rows(p->ya_img, D) : none}, nullptr);
    if (!last) {
      st355_gemm_args t = proto_of(p->w_add_out, D, D, D);
      t.bias = p->b_add_out; t.epilogue = ST355_EPI_GATE_RESIDUAL; t.gate = mt + 2 * D; t.gate_stride = ms; t.rows_per_batch = St; proto_ext(t, p->Bb_aout, p->K2_aout, p->k2r_aout);
      stream_probs(q, v, sc, B, St, t, StreamOps{O_t, p->K2_aout ? compact_rows(p->T_ao, p->K2_aout) : none, compact_rows(p->x1_txt, D), compact_rows(p->txt, D),
                                                   p->ya_txt ? compact_rows(p->ya_txt, D) : none}, nullptr);
    }
    run_grouped(q, v);
  }
  // the feed-forward branches                                                                                                                  (:216-239)
  if (q.ok()) q.run(st355_ln_modulate_fwd(stream, p->x1_img, D, mi + 4 * D, mi + 3 * D, ms, Si, p->n2_img, D, (int64_t)B * Si, D, 1e-6f));
  st355_gemm_args f1 = G(p->n2_img, D, p->w_ff1, D, p->h_img, 4 * D, B * Si, 4 * D, D);
  f1.bias = p->b_ff1; f1.epilogue = ST355_EPI_GELU; f1.aux_out = p->hpre_img; f1.ld_aux_out = 4 * D;
  st355_gemm_args f2 = G(p->h_img, 4 * D, p->w_ff2, 4 * D, p->out_img, D, B * Si, D, 4 * D);
  f2.bias = p->b_ff2; f2.epilogue = ST355_EPI_GATE_RESIDUAL; f2.aux_in = p->x1_img; f2.ld_aux_in = D; f2.gate = mi + 5 * D; f2.gate_stride = ms; f2.rows_per_batch = Si;
  if (p->yf_img) { f2.aux_out = p->yf_img; f2.ld_aux_out = D; }
  if (last) {
    q.gemm(f1);
    q.gemm(f2);
    return q.rc;
  }
  if (q.ok()) q.run(st355_ln_modulate_fwd(stream, p->x1_txt, D, mt + 4 * D, mt + 3 * D, ms, St, p->n2_txt, D, (int64_t)B * St, D, 1e-6f));
  st355_gemm_args c1 = G(p->n2_txt, D, p->w_ffc1, D, p->h_txt, 4 * D, B * St, 4 * D, D);
  c1.bias = p->b_ffc1; c1.epilogue = ST355_EPI_GELU; c1.aux_out = p->hpre_txt; c1.ld_aux_out = 4 * D;
  st355_gemm_args c2 = G(p->h_txt, 4 * D, p->w_ffc2, 4 * D, p->out_txt, D, B * St, D, 4 * D);
  c2.bias = p->b_ffc2; c2.epilogue = ST355_EPI_GATE_RESIDUAL; c2.aux_in = p->x1_txt; c2.ld_aux_in = D; c2.gate = mt + 5 * D; c2.gate_stride = ms; c2.rows_per_batch = St;
  if (p->yf_txt) { c2.aux_out = p->yf_txt; c2.ld_aux_out = D; }
  if (q.ok()) { st355_gemm_args g2[2] = {f1, c1}; q.run(st355_gemm_bf16_grouped(stream, g2, 2)); }
  if (q.ok()) { st355_gemm_args g2[2] = {f2, c2}; q.run(st355_gemm_bf16_grouped(stream, g2, 2)); }
  return q.rc;
}

extern "C" int st355_block_sd3_joint_bwd(void* stream, const st355_sd3_joint_bwd_args* p) {
  ST_REQUIRE(p && p->img && p->txt && p->mod_img && p->mod_txt && p->qkv && p->Q && p->K && p->O && p->lse2 && p->x1_img && p->hpre_img && p->d_img && p->cos && p->sin,
             "block_sd3_joint_bwd: null pointer");
  ST_REQUIRE(p->g_img && p->dh_img && p->dn2_img && p->dx1_img && p->dx1g_img && p->dO && p->dqkv && p->dQ && p->dK && p->attn_ws, "block_sd3_joint_bwd: null scratch pointer");
  ST_REQUIRE(p->last || (p->x1_txt && p->hpre_txt && p->d_txt && p->g_txt && p->dh_txt && p->dn2_txt && p->dx1_txt && p->dx1g_txt), "block_sd3_joint_bwd: null text-stream pointer");
  ST_REQUIRE(!p->need_input_grads || (p->dn_img && p->dn_txt && p->d_img_out && p->d_txt_out && p->wT_qkv && p->wT_add_qkv), "block_sd3_joint_bwd: null input-gradient pointer");
  ST_REQUIRE(p->B > 0 && p->Si > 0 && p->St > 0 && p->H > 0 && (p->hd == 64 || p->hd == 128) && p->D == p->H * p->hd, "block_sd3_joint_bwd: D = H * hd, hd 64 / 128");
  ST_REQUIRE((p->K2_qkv == 0 || (p->Bbt_qkv && p->U_qkv && (!p->need_input_grads || p->At_qkv))) && (p->K2_aqkv == 0 || (p->Bbt_aqkv && p->U_aqkv && (!p->need_input_grads || p->At_aqkv))) &&
             (p->K2_out == 0 || (p->Bbt_out && p->At_out && p->U_o)) && (p->K2_aout == 0 || p->last || (p->Bbt_aout && p->At_aout && p->U_ao)),
             "block_sd3_joint_bwd: inconsistent adapter operands");
  const int B = p->B, Si = p->Si, St = p->St, S = Si + St, Sp = (S + 63) / 64 * 64, H = p->H, D = p->D, hd = p->hd;
  const bool last = p->last != 0;
  const bool cp_i = B > 1 && Si % 256 != 0, cp_t = B > 1 && St % 256 != 0;          // the streams' rows of dqkv go through compact copies (`_compact` of the host side)
  ST_REQUIRE((!(cp_i || stream_mode(B, Si) == SM_COMPACT) || p->c_img) && (!(cp_t || stream_mode(B, St) == SM_COMPACT) || p->c_txt), "block_sd3_joint_bwd: a stream needs its compact-copy scratch");
  const int64_t ms = p->mod_stride;
  const bf16* mi = (const bf16*)p->mod_img; const bf16* mt = (const bf16*)p->mod_txt;
  const int64_t Mi = (int64_t)B * Si, Mt = (int64_t)B * St;
  const Rows none{nullptr, 0, 0};
  Seq q{stream, 0};
  std::vector<st355_gemm_args> v;
  std::vector<Scatter> sc;
  // full fine-tune: the modulation / gate / bias gradients ride in the passes below (csrc/stats.hip)
  const bool stats = p->dmod_img != nullptr;
  ST_REQUIRE(!stats || (p->need_input_grads && p->dmod_txt && p->dmod_stride > 0 && p->ya_img && p->yf_img && p->gb_ff2 && p->gb_ff1 && p->gb_out && p->gb_qkv &&
                        p->gb_add_qkv && p->stats_ws && (last || (p->ya_txt && p->yf_txt && p->gb_ffc2 && p->gb_ffc1 && p->gb_add_out))),
             "block_sd3_joint_bwd: the fused-statistics form needs every modulation / bias gradient destination, the kept branch outputs and stats_ws");
  float* dmi = (float*)p->dmod_img; float* dmt = (float*)p->dmod_txt;
  auto per_batch = [&](float* base, int chunk) { st355_stat_out o; memset(&o, 0, sizeof(o)); o.out = base + (int64_t)chunk * D; o.stride = p->dmod_stride; return o; };
  auto bias_row = [&](void* gb) { st355_stat_out o; memset(&o, 0, sizeof(o)); o.out = gb; o.reduce_batches = 1; o.out_bf16 = 1; return o; };
  // ---- feed-forward branches: g = gate_mlp * d, d h = (g W2) * GELU'(pre), d norm2 = d h W1; d x1 = d + LN'(.), and gate_msa * d x1 in the same pass ----
  if (stats) {
    const st355_stat_out dg = per_batch(dmi, 5), db = bias_row(p->gb_ff2);            // d gate_mlp = sum d * y_ff, d b_ff2 = sum g
    q.run(st355_scale_cols_stats(stream, p->d_img, D, mi + 5 * D, ms, Si, p->g_img, D, Mi, D, p->yf_img, D, &dg, &db, p->stats_ws));
  } else
  q.run(st355_scale_cols(stream, p->d_img, D, mi + 5 * D, ms, Si, p->g_img, D, Mi, D));
  st355_gemm_args h_i = G(p->g_img, D, p->wT_ff2, D, p->dh_img, 4 * D, B * Si, 4 * D, D);
  h_i.epilogue = ST355_EPI_MUL_GELU_GRAD; h_i.aux_in = p->hpre_img; h_i.ld_aux_in = 4 * D;
  st355_gemm_args n_i = G(p->dh_img, 4 * D, p->wT_ff1, 4 * D, p->dn2_img, D, B * Si, D, 4 * D);
  if (last) {
    q.gemm(h_i);
    q.gemm(n_i);
  } else {
    if (stats) {
      const st355_stat_out dg = per_batch(dmt, 5), db = bias_row(p->gb_ffc2);
      if (q.ok()) q.run(st355_scale_cols_stats(stream, p->d_txt, D, mt + 5 * D, ms, St, p->g_txt, D, Mt, D, p->yf_txt, D, &dg, &db, p->stats_ws));
    } else
    if (q.ok()) q.run(st355_scale_cols(stream, p->d_txt, D, mt + 5 * D, ms, St, p->g_txt, D, Mt, D));
    st355_gemm_args h_t = G(p->g_txt, D, p->wT_ffc2, D, p->dh_txt, 4 * D, B * St, 4 * D, D);
    h_t.epilogue = ST355_EPI_MUL_GELU_GRAD; h_t.aux_in = p->hpre_txt; h_t.ld_aux_in = 4 * D;
    st355_gemm_args n_t = G(p->dh_txt, 4 * D, p->wT_ffc1, 4 * D, p->dn2_txt, D, B * St, D, 4 * D);
    if (q.ok()) { st355_gemm_args g2[2] = {h_i, h_t}; q.run(st355_gemm_bf16_grouped(stream, g2, 2)); }
    if (q.ok()) { st355_gemm_args g2[2] = {n_i, n_t}; q.run(st355_gemm_bf16_grouped(stream, g2, 2)); }
    if (stats) {          // + d shift_mlp, d scale_mlp, d gate_msa = sum d x1 * y_attn, d b_add_out = sum gate_msa * d x1;  d b_ffc1 = sum d h
      const st355_stat_out dsh = per_batch(dmt, 3), dsc = per_batch(dmt, 4), dg = per_batch(dmt, 2), db = bias_row(p->gb_add_out), dbh = bias_row(p->gb_ffc1);
      if (q.ok())
        q.run(st355_ln_modulate_bwd_stats(stream, p->dn2_txt, D, p->x1_txt, D, mt + 4 * D, ms, St, p->d_txt, D, mt + 2 * D, ms, p->dx1_txt, D, p->dx1g_txt, D, Mt, D, 1e-6f,
                                          p->ya_txt, D, &dsh, &dsc, &dg, &db, p->stats_ws));
      if (q.ok()) q.run(st355_colsum_rows(stream, p->dh_txt, 4 * D, St, St, B, 4 * D, &dbh, p->stats_ws));
    } else
    if (q.ok())
      q.run(st355_ln_modulate_bwd(stream, p->dn2_txt, D, p->x1_txt, D, mt + 4 * D, ms, St, p->d_txt, D, mt + 2 * D, ms, p->dx1_txt, D, p->dx1g_txt, D, Mt, D, 1e-6f));
  }
  if (stats) {
    const st355_stat_out dsh = per_batch(dmi, 3), dsc = per_batch(dmi, 4), dg = per_batch(dmi, 2), db = bias_row(p->gb_out), dbh = bias_row(p->gb_ff1);
    if (q.ok())
      q.run(st355_ln_modulate_bwd_stats(stream, p->dn2_img, D, p->x1_img, D, mi + 4 * D, ms, Si, p->d_img, D, mi + 2 * D, ms, p->dx1_img, D, p->dx1g_img, D, Mi, D, 1e-6f,
                                        p->ya_img, D, &dsh, &dsc, &dg, &db, p->stats_ws));
    if (q.ok()) q.run(st355_colsum_rows(stream, p->dh_img, 4 * D, Si, Si, B, 4 * D, &dbh, p->stats_ws));
  } else
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn2_img, D, p->x1_img, D, mi + 4 * D, ms, Si, p->d_img, D, mi + 2 * D, ms, p->dx1_img, D, p->dx1g_img, D, Mi, D, 1e-6f));
  // ---- attention output projections -> the dO rows of both streams (a context_pre_only block writes no text rows: the caller zero-filled dO) ----
  if (p->K2_out) {
    st355_gemm_args u = G(p->dx1g_img, D, p->Bbt_out, D, p->U_o, p->K2_out, B * Si, p->K2_out, D);
    thin_ws(u, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(u);
  }
  if (!last && p->K2_aout) {
    st355_gemm_args u = G(p->dx1g_txt, D, p->Bbt_aout, D, p->U_ao, p->K2_aout, B * St, p->K2_aout, D);
    thin_ws(u, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(u);
  }
  {
    st355_gemm_args a = proto_of(p->wT_out, D, D, D);
    proto_ext(a, p->At_out, p->K2_out, p->k2r_out);
    stream_probs(q, v, sc, B, Si, a, StreamOps{compact_rows(p->dx1g_img, D), p->K2_out ? compact_rows(p->U_o, p->K2_out) : none, joint_rows(p->dO, D, 0, S), none, none}, p->c_img);
    if (!last) {
      st355_gemm_args t = proto_of(p->wT_add_out, D, D, D);
      proto_ext(t, p->At_aout, p->K2_aout, p->k2r_aout);
      stream_probs(q, v, sc, B, St, t, StreamOps{compact_rows(p->dx1g_txt, D), p->K2_aout ? compact_rows(p->U_ao, p->K2_aout) : none, joint_rows(p->dO, D, Si, S), none, none}, p->c_txt);
    }
    run_grouped(q, v);
    run_scatters(q, sc);
  }
  // ---- attention, then the RMSNorm (+ identity RoPE) backward of each stream's q / k into the projection-gradient rows ----
  const char* qkv = (const char*)p->qkv; char* dqkv = (char*)p->dqkv;
  if (q.ok())
    q.run(st355_attn_bwd(stream, p->Q, p->K, nullptr, nullptr, qkv + (size_t)2 * D * 2, 3 * D, p->O, D, p->dO, D, (const float*)p->lse2, nullptr, p->dQ, p->dK,
                         dqkv + (size_t)2 * D * 2, 3 * D, B, H, S, Sp, hd, p->scale, p->attn_ws));
  if (q.ok())
    q.run(st355_qk_norm_rope_bwd(stream, p->dQ, p->dK, p->qkv, 3 * D, p->norm_q, p->norm_k, (const float*)p->cos, (const float*)p->sin, p->dqkv, 3 * D, B, H, hd, Si, 0, S, 1e-6f));
  if (q.ok())
    q.run(st355_qk_norm_rope_bwd(stream, p->dQ, p->dK, p->qkv, 3 * D, p->norm_added_q, p->norm_added_k, (const float*)p->cos, (const float*)p->sin, p->dqkv, 3 * D, B, H, hd, St,
                                 Si, S, 1e-6f));
  if (stats) {          // d b_qkv / d b_add_qkv: column sums of each stream's rows of the joint dqkv, in place
    const st355_stat_out dbi = bias_row(p->gb_qkv), dbt = bias_row(p->gb_add_qkv);
    if (q.ok()) q.run(st355_colsum_rows(stream, p->dqkv, 3 * D, Si, S, B, 3 * D, &dbi, p->stats_ws));
    if (q.ok()) q.run(st355_colsum_rows(stream, (const char*)p->dqkv + (size_t)Si * 3 * D * 2, 3 * D, St, S, B, 3 * D, &dbt, p->stats_ws));
  }
  // ---- input projections: each stream's rows of dqkv in place (segmented) when tile-aligned or B == 1, else a compact copy ----
  Rows dq_i = joint_rows(p->dqkv, 3 * D, 0, S), dq_t = joint_rows(p->dqkv, 3 * D, Si, S);
  if (cp_i) { if (q.ok()) q.run(gather_block(stream, p->c_img, dq_i, Si, B)); dq_i = compact_rows(p->c_img, 3 * D); }
  if (cp_t) { if (q.ok()) q.run(gather_block(stream, p->c_txt, dq_t, St, B)); dq_t = compact_rows(p->c_txt, 3 * D); }
  auto whole = [&](Rows A, const void* W, int64_t ldw, void* C, int64_t ldc, int rows, int N, int K) {       // ONE problem over the stream's rows (2-D or segmented)
    st355_gemm_args a = G(A.p, A.ld, W, ldw, C, ldc, B * rows, N, K);
    if (B > 1 && A.seg) { a.seg_rows = rows; a.seg_a = A.seg; }
    return a;
  };
  if (p->K2_qkv) {
    st355_gemm_args u = whole(dq_i, p->Bbt_qkv, 3 * D, p->U_qkv, p->K2_qkv, Si, p->K2_qkv, 3 * D);
    thin_ws(u, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(u);
  }
  if (p->K2_aqkv) {
    st355_gemm_args u = whole(dq_t, p->Bbt_aqkv, 3 * D, p->U_aqkv, p->K2_aqkv, St, p->K2_aqkv, 3 * D);
    thin_ws(u, p->gemm_ws, p->gemm_ws_bytes);
    q.gemm(u);
  }
  if (!p->need_input_grads) return q.rc;
  {
    st355_gemm_args g2[2];
    g2[0] = whole(dq_i, p->wT_qkv, 3 * D, p->dn_img, D, Si, D, 3 * D);
    if (p->K2_qkv) { g2[0].A2 = p->U_qkv; g2[0].lda2 = p->K2_qkv; g2[0].B2 = p->At_qkv; g2[0].ldb2 = p->K2_qkv; g2[0].K2 = p->K2_qkv; g2[0].K2_real = p->k2r_qkv; }
    g2[1] = whole(dq_t, p->wT_add_qkv, 3 * D, p->dn_txt, D, St, D, 3 * D);
    if (p->K2_aqkv) { g2[1].A2 = p->U_aqkv; g2[1].lda2 = p->K2_aqkv; g2[1].B2 = p->At_aqkv; g2[1].ldb2 = p->K2_aqkv; g2[1].K2 = p->K2_aqkv; g2[1].K2_real = p->k2r_aqkv; }
    if (q.ok()) q.run(st355_gemm_bf16_grouped(stream, g2, 2));
  }
  if (stats) {          // + d shift_msa, d scale_msa (the `last` block's text norm is AdaLayerNormContinuous: chunks (scale, shift))
    const st355_stat_out ish = per_batch(dmi, 0), isc = per_batch(dmi, 1), tsh = per_batch(dmt, last ? 1 : 0), tsc = per_batch(dmt, last ? 0 : 1);
    if (q.ok())
      q.run(st355_ln_modulate_bwd_stats(stream, p->dn_img, D, p->img, D, mi + D, ms, Si, p->dx1_img, D, nullptr, 0, p->d_img_out, D, nullptr, D, Mi, D, 1e-6f, nullptr, 0,
                                        &ish, &isc, nullptr, nullptr, p->stats_ws));
    if (q.ok())
      q.run(st355_ln_modulate_bwd_stats(stream, p->dn_txt, D, p->txt, D, last ? mt : mt + D, ms, St, last ? nullptr : p->dx1_txt, last ? 0 : D, nullptr, 0, p->d_txt_out, D,
                                        nullptr, D, Mt, D, 1e-6f, nullptr, 0, &tsh, &tsc, nullptr, nullptr, p->stats_ws));
    return q.rc;
  }
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn_img, D, p->img, D, mi + D, ms, Si, p->dx1_img, D, nullptr, 0, p->d_img_out, D, nullptr, D, Mi, D, 1e-6f));
  if (q.ok())
    q.run(st355_ln_modulate_bwd(stream, p->dn_txt, D, p->txt, D, last ? mt : mt + D, ms, St, last ? nullptr : p->dx1_txt, last ? 0 : D, nullptr, 0, p->d_txt_out, D, nullptr, D, Mt,
                                D, 1e-6f));
  return q.rc;
}
