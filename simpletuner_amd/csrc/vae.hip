// vae.hip — st355_vae_encode: AutoencoderKL.encode (the VAE latent encode of the hot path) as ONE C entry point.
//
// reference seam: VAECache.encode_images -> model.encode_with_vae -> vae.encode(samples).latent_dist  (simpletuner/helpers/caching/vae.py:1238-1396,
// models/common.py:2767-2772); the network is diffusers' AutoencoderKL encoder: conv_in -> DownEncoderBlock2D x n (ResnetBlock2D x layers, Downsample2D
// = F.pad(0,1,0,1) + 3x3 stride-2 pad-0 conv) -> UNetMidBlock2D (resnet, single-head attention of width C, resnet) -> GroupNorm -> SiLU -> conv_out
// (with the 1x1 quant_conv folded into conv_out by the host at load time).
//
// This file only SEQUENCES kernels that already are C entry points of libst355 (grid layout passes, GroupNorm(+SiLU), the convolution-as-GEMM, the row
// softmax, the NT GEMM): same launches, same order and same operands as the Python sequencing it replaces (vae/autoencoder_kl.py until round 3), so the
// result is bit-identical to it.  No allocation: every intermediate lives in the caller's workspace (st355_vae_encode_workspace bytes), handed out by a
// stack allocator — a block's output is allocated before its temporaries, the temporaries are released when the block ends.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "common.h"

namespace {
struct Stack {
  char* base; size_t cap, top, peak; bool dry;
  void* alloc(size_t bytes) {
    const size_t o = (top + 255) & ~(size_t)255;
    top = o + bytes;
    if (top > peak) peak = top;
    return dry ? (void*)(uintptr_t)256 : (void*)(base + o);          // dry run: sizes only, nothing is dereferenced or launched
  }
};

struct Enc {
  hipStream_t st; const st355_vae_encoder* e; Stack m; int B; void* gn_ws; int rc; int wi;
  const void* next() { return e->tensors[wi++]; }
  bool ok() const { return rc == 0; }
  void chk(int r) { if (rc == 0 && r != 0) rc = r; }

  // a grid buffer the next kernel fills: only the rows that kernel does not write are zeroed (ops._grid_out of the Python side)
  void* grid(int H, int W, int C, bool conv) {
    const int64_t n = (int64_t)B * (H + 2) * (W + 2);
    char* p = (char*)m.alloc((size_t)(n + 64) * C * 2);
    if (!m.dry && ok()) {
      if (conv) {
        hipMemsetAsync(p, 0, (size_t)(W + 3) * C * 2, st);
        hipMemsetAsync(p + (size_t)(n - (W + 3)) * C * 2, 0, (size_t)(W + 3 + 64) * C * 2, st);
      } else {
        hipMemsetAsync(p + (size_t)n * C * 2, 0, (size_t)64 * C * 2, st);
      }
    }
    return p;
  }
  void* gn(const void* x, int H, int W, int C, bool silu, bool tokens) {
    const void* g = next(); const void* b = next();
    void* y = tokens ? m.alloc((size_t)B * H * W * C * 2) : grid(H, W, C, false);
    float* stats = (float*)m.alloc((size_t)B * C * 2 * 4);
    if (!m.dry && ok()) chk(st355_groupnorm_fwd(st, x, g, b, y, stats, B, H, W, C, e->norm_num_groups, 1e-6f, silu ? 1 : 0, tokens ? 1 : 0, gn_ws));
    return y;
  }
  void conv_into(const void* x, void* out, int H, int W, int Cin, int Cout, int taps, const void* residual) {
    const void* w = next(); const void* b = next();
    if (!m.dry && ok()) chk(st355_conv_bf16(st, x, w, b, nullptr, 0, residual, out, B, H, W, Cin, Cout, taps));
  }
  // ResnetBlock2D without a time embedding: GN-SiLU-conv3x3-GN-SiLU-conv3x3 + (1x1 shortcut | identity)
  void* res(const void* x, int H, int W, int ci, int co) {
    void* out = grid(H, W, co, true);
    const size_t mark = m.top;
    void* h1 = gn(x, H, W, ci, true, false);
    void* h2 = grid(H, W, co, true);
    conv_into(h1, h2, H, W, ci, co, 9, nullptr);
    void* h3 = gn(h2, H, W, co, true, false);
    const void* w2 = next(); const void* b2 = next();                  // conv2 comes before the shortcut in the tensor walk
    const void* sc = x;
    if (ci != co) {
      void* s = grid(H, W, co, true);
      conv_into(x, s, H, W, ci, co, 1, nullptr);
      sc = s;
    }
    if (!m.dry && ok()) chk(st355_conv_bf16(st, h3, w2, b2, nullptr, 0, sc, out, B, H, W, co, co, 9));
    m.top = mark;
    return out;
  }
  void gemm(const void* A, int64_t lda, const void* Bm, int64_t ldb, const void* bias, void* C, int64_t ldc, int M, int N, int K) {
    if (m.dry || !ok()) return;
    st355_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.B = Bm; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.bias = bias; a.epilogue = ST355_EPI_NONE;
    chk(st355_gemm_bf16(st, &a));
  }
  // UNetMidBlock2D attention: one head of width C over the S = H*W positions of each image: scores = q k^T as a plain [S, S] GEMM, row softmax, P V
  void* mid_attention(const void* x, int H, int W, int C) {
    void* out = grid(H, W, C, false);
    const size_t mark = m.top;
    const int S = H * W, Sp = (S + 63) / 64 * 64;
    void* n = gn(x, H, W, C, false, true);                            // tokens [B*S, C]
    const void* wqkv = next(); const void* bqkv = next(); const void* wo = next(); const void* bo = next();
    bf16* qkv = (bf16*)m.alloc((size_t)B * S * 3 * C * 2);
    gemm(n, C, wqkv, C, bqkv, qkv, 3 * C, B * S, 3 * C, C);
    bf16* o = (bf16*)m.alloc((size_t)B * S * C * 2);
    bf16* scores = (bf16*)m.alloc((size_t)S * Sp * 2);
    bf16* sc_raw = Sp != S ? (bf16*)m.alloc((size_t)S * S * 2) : scores;
    bf16* vt = (bf16*)m.alloc((size_t)C * Sp * 2);
    for (int b = 0; b < B; b++) {
      const bf16* q = qkv + (size_t)b * S * 3 * C; const bf16* k = q + C; const bf16* v = q + 2 * C;
      gemm(q, 3 * C, k, 3 * C, nullptr, sc_raw, S, S, S, C);
      if (!m.dry && ok()) chk(st355_softmax_rows(st, sc_raw, S, S, S, 1.0f / sqrtf((float)C)));
      if (Sp != S && !m.dry && ok()) {                                // contraction granule 64: zero-padded probabilities / V^T
        hipMemsetAsync(scores, 0, (size_t)S * Sp * 2, st);
        hipMemcpy2DAsync(scores, (size_t)Sp * 2, sc_raw, (size_t)S * 2, (size_t)S * 2, S, hipMemcpyDeviceToDevice, st);
        hipMemsetAsync(vt, 0, (size_t)C * Sp * 2, st);
      }
      if (!m.dry && ok()) chk(st355_transpose_bf16(st, v, 3 * C, vt, Sp, S, C));
      gemm(scores, Sp, vt, Sp, nullptr, o + (size_t)b * S * C, C, S, C, Sp);
    }
    bf16* proj = (bf16*)m.alloc((size_t)B * S * C * 2);
    gemm(o, C, wo, C, bo, proj, C, B * S, C, C);
    if (!m.dry && ok()) chk(st355_tokens_to_grid(st, proj, x, out, B, H, W, C));
    m.top = mark;
    return out;
  }

  int run(const void* pixels, void* moments, int H, int W) {
    const int nb = e->n_levels, L2 = 2 * e->latent_channels;
    const int* ch = e->block_out_channels;
    size_t gnb = 0;
    { int h = H, w = W; for (int i = 0; i < nb; i++) { size_t s = st355_groupnorm_workspace(B, h, w, ch[i]); if (s > gnb) gnb = s; if (i < nb - 1) { h /= 2; w /= 2; } } }
    gn_ws = m.alloc(gnb);
    void* g0 = grid(H, W, 8, false);
    if (!m.dry && ok()) chk(st355_grid_from_nchw(st, pixels, g0, B, e->in_channels, H, W, 8));
    void* col = grid(H, W, 128, false);
    if (!m.dry && ok()) chk(st355_im2col3x3(st, g0, col, B, H, W, 8, 1, 128, 1));
    void* h = grid(H, W, ch[0], true);
    conv_into(col, h, H, W, 128, ch[0], 1, nullptr);                 // conv_in over pre-gathered columns: [ch0, 128] weights (9 taps x 8 padded channels)
    int cin = ch[0];
    for (int i = 0; i < nb; i++) {
      for (int j = 0; j < e->layers_per_block; j++) { h = res(h, H, W, cin, ch[i]); cin = ch[i]; }
      if (i < nb - 1) {                                               // Downsample2D (padding 0): columns on the output grid, then a 1-tap GEMM
        const int Kpad = (9 * cin + 63) / 64 * 64;
        void* out = grid(H / 2, W / 2, cin, true);
        const size_t mark = m.top;
        void* c2 = grid(H / 2, W / 2, Kpad, false);
        if (!m.dry && ok()) chk(st355_im2col3x3(st, h, c2, B, H, W, cin, 2, Kpad, 0));
        H /= 2; W /= 2;
        conv_into(c2, out, H, W, Kpad, cin, 1, nullptr);
        m.top = mark;
        h = out;
      }
    }
    h = res(h, H, W, cin, cin);
    h = mid_attention(h, H, W, cin);
    h = res(h, H, W, cin, cin);
    void* hn = gn(h, H, W, cin, true, false);
    void* y = grid(H, W, L2, true);
    conv_into(hn, y, H, W, cin, L2, 9, nullptr);
    if (!m.dry && ok()) chk(st355_grid_to_nchw(st, y, moments, B, L2, H, W, L2));
    if (wi != e->n_tensors) { st355_set_error("vae_encode: the tensor table holds %d entries, the architecture walks %d", e->n_tensors, wi); return ST355_EINVAL; }
    return rc;
  }
};

int check(const st355_vae_encoder* e, int B, int H, int W) {
  ST_REQUIRE(e && e->tensors && B > 0 && H > 0 && W > 0, "vae_encode: bad args");
  ST_REQUIRE(e->n_levels >= 1 && e->n_levels <= 8 && e->layers_per_block >= 1 && e->in_channels >= 1 && e->in_channels <= 8, "vae_encode: bad architecture");
  ST_REQUIRE((2 * e->latent_channels) % 8 == 0 && e->norm_num_groups > 0, "vae_encode: 2 * latent_channels must be a multiple of 8");
  ST_REQUIRE(H % (1 << (e->n_levels - 1)) == 0 && W % (1 << (e->n_levels - 1)) == 0, "vae_encode: image sides must be divisible by 2^(levels-1)");
  for (int i = 0; i < e->n_levels; i++) ST_REQUIRE(e->block_out_channels[i] % 64 == 0, "vae_encode: block widths must be multiples of 64 (got %d)", e->block_out_channels[i]);
  return ST355_OK;
}
}  // namespace

extern "C" size_t st355_vae_encode_workspace(const st355_vae_encoder* enc, int B, int H, int W) {
  if (check(enc, B, H, W) != ST355_OK) return 0;
  Enc r{nullptr, enc, Stack{nullptr, 0, 0, 0, true}, B, nullptr, 0, 0};
  r.run(nullptr, nullptr, H, W);
  return r.m.peak + 256;
}

extern "C" int st355_vae_encode(void* stream, const st355_vae_encoder* enc, const void* pixels, void* moments, int B, int H, int W, void* workspace,
                                size_t workspace_bytes) {
  int rc = check(enc, B, H, W);
  if (rc) return rc;
  ST_REQUIRE(pixels && moments && workspace && ((uintptr_t)workspace % 256) == 0, "vae_encode: null / misaligned pointer");
  const size_t need = st355_vae_encode_workspace(enc, B, H, W);
  ST_REQUIRE(workspace_bytes >= need, "vae_encode: workspace of %zu bytes, %zu needed (st355_vae_encode_workspace)", workspace_bytes, need);
  Enc r{(hipStream_t)stream, enc, Stack{(char*)workspace, workspace_bytes, 0, 0, false}, B, nullptr, 0, 0};
  return r.run(pixels, moments, H, W);
}
