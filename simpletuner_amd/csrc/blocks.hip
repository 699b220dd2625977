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
