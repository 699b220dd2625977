// stats.hip — the token-axis reductions of the full fine-tune backward FUSED into the passes that already stream the tensors they reduce.
// What autograd accumulates for the AdaLN modulation outputs and the biases (trainer.py:7126 through sd3/transformer.py:145-241, flux/transformer.py:607-687):
//     d shift_b = sum_t dY,   d scale_b = sum_t dY * LN(x),   d gate_b = sum_t dOut * y_branch,   d bias = sum_{b,t} dY
// Round 5 ran each of them as its own st355_colsum_prod pass (+ one LN(x) materialisation per AdaLN instance): 20 launches and 2.3 GB of HBM traffic per SD3-Medium
// block at batch 8 (rocprofv3: k_colsum_prod + k_colsum_finalize = 15 ms of a 325 ms step).  Here the sums ride in the kernels that produce / consume the operands:
//   * k_ln_mod_bwd_stats   = k_ln_mod_bwd (LayerNorm + modulation backward, residual add, next gate) + d shift, d scale [+ d gate of the branch whose residual
//                            gradient it writes, + the bias gradient of the Linear that consumes its gated output]
//   * k_cols_stats         = k_scale_cols (g = gate_b * d) + d gate = sum d * y_branch + d bias = sum g;  without a gate: a plain (strided-batch) column sum
// A workgroup owns 64 consecutive rows of ONE batch element: waves accumulate in registers (a lane owns the same columns for every row), meet through LDS in a fixed
// order and leave one fp32 partial row per sum; k_stats_finalize adds the partial rows in index order — deterministic, no atomics.
#include "common.h"

#define SR_ROWS 64

struct StatOutD {
  float* f32; bf16* b16;
  int64_t stride;
  int reduce_batches, accumulate;
};
struct StatOuts { StatOutD o[4]; };

static StatOutD to_d(const st355_stat_out* s) {
  StatOutD d;
  memset(&d, 0, sizeof(d));
  if (s && s->out) {
    if (s->out_bf16) d.b16 = (bf16*)s->out; else d.f32 = (float*)s->out;
    d.stride = s->stride; d.reduce_batches = s->reduce_batches; d.accumulate = s->accumulate;
  }
  return d;
}

// grid (ceil(N / 32), nb, NS): 32 columns x 8 interleaved chains over the partial rows, summed in a fixed order
__global__ void __launch_bounds__(256) k_stats_finalize(const float* __restrict__ ws, int nchunks, int nb, int NS, int N, StatOuts outs) {
  __shared__ float red[8][32];
  const int k = blockIdx.z, bi = blockIdx.y;
  const StatOutD o = outs.o[k];
  if (o.f32 == nullptr && o.b16 == nullptr) return;
  if (o.reduce_batches && bi != 0) return;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  const int first = o.reduce_batches ? 0 : bi * nchunks, count = o.reduce_batches ? nb * nchunks : nchunks;
  float part = 0.f;
  if (c < N)
    for (int i = sl; i < count; i += 8) part += ws[((int64_t)(first + i) * NS + k) * N + c];
  red[sl][threadIdx.x & 31] = part;
  __syncthreads();
  if (sl != 0 || c >= N) return;
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 8; q++) s += red[q][threadIdx.x & 31];
  const int64_t at = (o.reduce_batches ? 0 : (int64_t)bi * o.stride) + c;
  if (o.b16) o.b16[at] = f2bf(o.accumulate ? bf2f(o.b16[at]) + s : s);
  else o.f32[at] = o.accumulate ? o.f32[at] + s : s;
}

static int finalize(void* stream, const float* ws, int nchunks, int nb, int NS, int N, const StatOuts& outs, const char* what) {
  hipLaunchKernelGGL(k_stats_finalize, dim3((N + 31) / 32, nb, NS), dim3(256), 0, (hipStream_t)stream, ws, nchunks, nb, NS, N, outs);
  return st355_check_launch(what);
}

extern "C" size_t st355_stats_workspace(int64_t rows, int N, int64_t rows_per_batch, int nsums) {
  if (rows <= 0 || rows_per_batch <= 0 || N <= 0 || nsums <= 0) return 0;
  const int64_t nb = rows / rows_per_batch;
  return (size_t)nb * (size_t)cdiv64(rows_per_batch, SR_ROWS) * (size_t)nsums * (size_t)N * sizeof(float);
}

// ================================================================================================
// LayerNorm + modulation backward with the modulation / gate / bias sums.  grid (nchunks, nb), 4 waves; wave w takes rows chunk*64 + w + 4 i.
//   sums: 0 = d shift (sum dy)   1 = d scale (sum dy * xhat)   GS: 2 = d gate (sum dx * y_branch, dx as stored)   3 = d bias (sum dxg, as stored)
// ================================================================================================
template <int NC, bool GS>
__global__ void __launch_bounds__(256) k_ln_mod_bwd_stats(const bf16* __restrict__ dy, int64_t lddy, const bf16* __restrict__ x, int64_t ldx,
                                                         const bf16* __restrict__ scale, int64_t mod_stride, int64_t rows_per_batch,
                                                         const bf16* __restrict__ dres, int64_t lddres, const bf16* __restrict__ gate, int64_t gate_stride,
                                                         bf16* __restrict__ dx, int64_t lddx, bf16* __restrict__ dxg, int64_t lddxg,
                                                         const bf16* __restrict__ yb, int64_t ldyb, int D, float eps, float one, float* __restrict__ ws, int nchunks) {
  constexpr int NS = GS ? 4 : 2;
  __shared__ float red[4][NC * 512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int chunk = blockIdx.x, bi = blockIdx.y;
  float scm[NC][8], a_sh[NC][8], a_sc[NC][8], a_g[GS ? NC : 1][8], a_b[GS ? NC : 1][8];
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (idx < D) {
      const bf16x8 scv = *(const bf16x8*)(scale + (int64_t)bi * mod_stride + idx);
#pragma unroll
      for (int j = 0; j < 8; j++) scm[c][j] = one + bf2f(scv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) scm[c][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { a_sh[c][j] = 0.f; a_sc[c][j] = 0.f; }
    if (GS) {
#pragma unroll
      for (int j = 0; j < 8; j++) { a_g[c][j] = 0.f; a_b[c][j] = 0.f; }
    }
  }
  // One memory round trip per row, issued ONE ROW AHEAD: every operand of row r + 4 (x, dy, residual, branch output) is requested before row r's reductions start,
  // so the four dependent wave reductions of a row run under the next row's loads (first form: two dependent round trips per row and nothing in flight during the
  // reductions — 2.7 TB/s, rocprofv3 r06; the unfused k_ln_mod_bwd hides the same latency with 4x the resident waves, which the accumulators here do not leave room for)
  const int64_t r_end = min((int64_t)(chunk + 1) * SR_ROWS, rows_per_batch);
  bf16x8 rx[NC], rdy[NC], rres[NC], rya[NC], gvr[NC];
  const bool has_y = GS && yb != nullptr;
  auto issue = [&](int64_t r) {
    const int64_t row = (int64_t)bi * rows_per_batch + r;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
        rx[c] = *(const bf16x8*)(x + row * ldx + idx);
        rdy[c] = *(const bf16x8*)(dy + row * lddy + idx);
        if (NC <= 4) {                      // (NC = 6: these two are fetched where they are used, as k_ln_mod_bwd does)
          if (dres) rres[c] = *(const bf16x8*)(dres + row * lddres + idx);
          if (has_y) rya[c] = *(const bf16x8*)(yb + row * ldyb + idx);
        }
      }
    }
  };
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int idx = (c * 64 + lane) * 8;
    if (NC <= 4 && dxg && idx < D) gvr[c] = *(const bf16x8*)(gate + (int64_t)bi * gate_stride + idx);
  }
  constexpr bool PREF = NC <= 4;          // D = 3072 (NC = 6): the second row's operands do not fit the register file next to the accumulators — plain order there
  int64_t r = (int64_t)chunk * SR_ROWS + wv;
  if (PREF && r < r_end) issue(r);
  for (; r < r_end; r += 4) {
    const int64_t row = (int64_t)bi * rows_per_batch + r;
    if (!PREF) issue(r);
    float v[NC][8], dv[NC][8];
    bf16x8 cres[NC], cya[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) { v[c][j] = bf2f(rx[c][j]); s += v[c][j]; dv[c][j] = bf2f(rdy[c][j]); }
        if (NC <= 4) { cres[c] = rres[c]; cya[c] = rya[c]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) { v[c][j] = 0.f; dv[c][j] = 0.f; }
      }
    }
    if (PREF && r + 4 < r_end) issue(r + 4);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) { const float d = v[c][j] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float xh = (v[c][j] - mean) * rstd;
          v[c][j] = xh;
          const float g = dv[c][j] * scm[c][j];
          sg += g;
          sgx += g * xh;
          a_sh[c][j] += dv[c][j];
          a_sc[c][j] += dv[c][j] * xh;
        }
      }
    }
    const float c1 = wave_sum(sg) / (float)D;
    const float c2 = wave_sum(sgx) / (float)D;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int idx = (c * 64 + lane) * 8;
      if (idx < D) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = rstd * (dv[c][j] * scm[c][j] - c1 - v[c][j] * c2);
        if (dres) {
          if (NC > 4) cres[c] = *(const bf16x8*)(dres + row * lddres + idx);
#pragma unroll
          for (int j = 0; j < 8; j++) o[j] += bf2f(cres[c][j]);
        }
        bf16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; j++) ov[j] = f2bf(o[j]);
        *(bf16x8*)(dx + row * lddx + idx) = ov;
        if (GS) {
          if (has_y) {
            if (NC > 4) cya[c] = *(const bf16x8*)(yb + row * ldyb + idx);
#pragma unroll
            for (int j = 0; j < 8; j++) a_g[c][j] += bf2f(ov[j]) * bf2f(cya[c][j]);
          }
        }
        if (dxg) {
          bf16x8 og;
          if (NC > 4) gvr[c] = *(const bf16x8*)(gate + (int64_t)bi * gate_stride + idx);
#pragma unroll
          for (int j = 0; j < 8; j++) og[j] = f2bf(bf2f(ov[j]) * bf2f(gvr[c][j]));
          *(bf16x8*)(dxg + row * lddxg + idx) = og;
          if (GS) {
#pragma unroll
            for (int j = 0; j < 8; j++) a_b[c][j] += bf2f(og[j]);
          }
        }
      }
    }
  }
  float* wrow = ws + ((int64_t)bi * nchunks + chunk) * NS * D;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; c++) {
      f32x4 lo, hi;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        lo[j] = k == 0 ? a_sh[c][j] : k == 1 ? a_sc[c][j] : k == 2 ? a_g[GS ? c : 0][j] : a_b[GS ? c : 0][j];
        hi[j] = k == 0 ? a_sh[c][4 + j] : k == 1 ? a_sc[c][4 + j] : k == 2 ? a_g[GS ? c : 0][4 + j] : a_b[GS ? c : 0][4 + j];
      }
      *(f32x4*)&red[wv][(c * 64 + lane) * 8] = lo;
      *(f32x4*)&red[wv][(c * 64 + lane) * 8 + 4] = hi;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NC * 512; i += 256)
      if (i < D) wrow[(int64_t)k * D + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

extern "C" int st355_ln_modulate_bwd_stats(void* stream, const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* scale, int64_t mod_stride,
                                           int64_t rows_per_batch, const void* dres, int64_t lddres, const void* gate, int64_t gate_stride, void* dx, int64_t lddx,
                                           void* dxg, int64_t lddxg, int64_t rows, int D, float eps, const void* y_branch, int64_t ld_y,
                                           const st355_stat_out* d_shift, const st355_stat_out* d_scale, const st355_stat_out* d_gate, const st355_stat_out* d_bias,
                                           void* workspace) {
  ST_REQUIRE(dy && x && scale && dx && workspace && d_shift && d_scale, "ln_modulate_bwd_stats: null pointer");
  ST_REQUIRE(D % 8 == 0 && D <= 3072 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && mod_stride % 8 == 0 && rows > 0 && rows_per_batch > 0 &&
                 rows % rows_per_batch == 0, "ln_modulate_bwd_stats: bad shape D=%d", D);
  if (dres) ST_REQUIRE(lddres % 8 == 0, "ln_modulate_bwd_stats: lddres");
  if (dxg) ST_REQUIRE(gate && gate_stride % 8 == 0 && lddxg % 8 == 0, "ln_modulate_bwd_stats: gate missing for dxg");
  const bool gs = (d_gate && d_gate->out) || (d_bias && d_bias->out);
  if (d_gate && d_gate->out) ST_REQUIRE(y_branch && ld_y % 8 == 0, "ln_modulate_bwd_stats: the gate gradient needs the branch output");
  if (d_bias && d_bias->out) ST_REQUIRE(dxg, "ln_modulate_bwd_stats: the bias gradient is the column sum of the gated output");
  const int nb = (int)(rows / rows_per_batch), nchunks = (int)cdiv64(rows_per_batch, SR_ROWS);
  const int NS = gs ? 4 : 2;
  {
    ProfScope ps(stream, ST355_K_LN_MOD, (16.0 + 2.0 * NS) * rows * D, (6.0 + (dres ? 2.0 : 0.0) + (dxg ? 2.0 : 0.0) + ((gs && y_branch) ? 2.0 : 0.0)) * rows * D);
    dim3 grid(nchunks, nb), block(256);
#define LAUNCH(NC, GS_)                                                                                                                         \
  hipLaunchKernelGGL((k_ln_mod_bwd_stats<NC, GS_>), grid, block, 0, (hipStream_t)stream, (const bf16*)dy, lddy, (const bf16*)x, ldx, (const bf16*)scale,     \
                     mod_stride, rows_per_batch, (const bf16*)dres, lddres, (const bf16*)gate, gate_stride, (bf16*)dx, lddx, (bf16*)dxg, lddxg,             \
                     (const bf16*)((d_gate && d_gate->out) ? y_branch : nullptr), ld_y, D, eps, 1.f, (float*)workspace, nchunks)
#define PICK(GS_)                        \
  do {                                   \
    if (D <= 512) LAUNCH(1, GS_);        \
    else if (D <= 1024) LAUNCH(2, GS_);  \
    else if (D <= 1536) LAUNCH(3, GS_);  \
    else if (D <= 2048) LAUNCH(4, GS_);  \
    else LAUNCH(6, GS_);                 \
  } while (0)
    if (gs) PICK(true); else PICK(false);
#undef PICK
#undef LAUNCH
    int rc = st355_check_launch("ln_modulate_bwd_stats");
    if (rc) return rc;
  }
  StatOuts outs;
  outs.o[0] = to_d(d_shift); outs.o[1] = to_d(d_scale); outs.o[2] = to_d(d_gate); outs.o[3] = to_d(d_bias);
  return finalize(stream, (const float*)workspace, nchunks, nb, NS, D, outs, "ln_modulate_bwd_stats_finalize");
}

// ================================================================================================
// g = gate_b * a (optional) with  sum 0 = sum_t a * y (or, with neither y nor an output, sum_t a: the plain column sum)  and  sum 1 = sum_t g (as stored).
// Rows of batch element b start at physical row b * batch_stride (batch_stride == rows_per_batch: compact).  grid (ceil(N/512), nchunks, nb).
// ================================================================================================
__global__ void __launch_bounds__(256) k_cols_stats(const bf16* __restrict__ a, int64_t lda, const bf16* __restrict__ gate, int64_t gate_stride,
                                                   const bf16* __restrict__ y, int64_t ldy, bf16* __restrict__ out, int64_t ldo, int64_t rows_per_batch,
                                                   int64_t batch_stride, int N, float* __restrict__ ws, int nchunks, int NS) {
  __shared__ float part[4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + lane * 8;
  const int chunk = blockIdx.y, bi = blockIdx.z;
  float acc0[8], acc1[8], gf[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { acc0[j] = 0.f; acc1[j] = 0.f; gf[j] = 1.f; }
  if (col < N) {
    if (gate) {
      const bf16x8 gv = *(const bf16x8*)(gate + (int64_t)bi * gate_stride + col);
#pragma unroll
      for (int j = 0; j < 8; j++) gf[j] = bf2f(gv[j]);
    }
    const int64_t r_end = min((int64_t)(chunk + 1) * SR_ROWS, rows_per_batch);
    for (int64_t r = (int64_t)chunk * SR_ROWS + wv; r < r_end; r += 4) {
      const int64_t row = (int64_t)bi * batch_stride + r;
      const int64_t lrow = (int64_t)bi * rows_per_batch + r;               // y and out are compact [nb * rows_per_batch, .]
      const bf16x8 av = *(const bf16x8*)(a + row * lda + col);
      if (y) {
        const bf16x8 yv = *(const bf16x8*)(y + lrow * ldy + col);
#pragma unroll
        for (int j = 0; j < 8; j++) acc0[j] += bf2f(av[j]) * bf2f(yv[j]);
      } else if (!out) {                                                   // plain column sum (st355_colsum_rows): sum 0 = sum_t a
#pragma unroll
        for (int j = 0; j < 8; j++) acc0[j] += bf2f(av[j]);
      }
      if (out) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) { o[j] = f2bf(bf2f(av[j]) * gf[j]); acc1[j] += bf2f(o[j]); }
        *(bf16x8*)(out + lrow * ldo + col) = o;
      }
    }
  }
  float* wrow = ws + ((int64_t)bi * nchunks + chunk) * NS * N;
  for (int k = 0; k < NS; k++) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) part[wv][lane * 8 + j] = k == 0 ? acc0[j] : acc1[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
      const int c = blockIdx.x * 512 + i;
      if (c < N) wrow[(int64_t)k * N + c] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    }
  }
}

extern "C" int st355_scale_cols_stats(void* stream, const void* in, int64_t ld_in, const void* gate, int64_t gate_stride, int64_t rows_per_batch, void* out,
                                      int64_t ld_out, int64_t M, int N, const void* y_branch, int64_t ld_y, const st355_stat_out* d_gate,
                                      const st355_stat_out* d_bias, void* workspace) {
  ST_REQUIRE(in && gate && out && workspace && N % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && gate_stride % 8 == 0 && rows_per_batch > 0 && M > 0 &&
                 M % rows_per_batch == 0, "scale_cols_stats: bad args");
  const bool want_gate = d_gate && d_gate->out;
  if (want_gate) ST_REQUIRE(y_branch && ld_y % 8 == 0, "scale_cols_stats: the gate gradient needs the branch output");
  const int nb = (int)(M / rows_per_batch), nchunks = (int)cdiv64(rows_per_batch, SR_ROWS);
  {
    ProfScope ps(stream, ST355_K_ELEMENTWISE, 3.0 * M * N, (4.0 + (want_gate ? 2.0 : 0.0)) * M * N);
    hipLaunchKernelGGL(k_cols_stats, dim3((N + 511) / 512, nchunks, nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, ld_in, (const bf16*)gate, gate_stride,
                       (const bf16*)(want_gate ? y_branch : nullptr), ld_y, (bf16*)out, ld_out, rows_per_batch, rows_per_batch, N, (float*)workspace, nchunks, 2);
    int rc = st355_check_launch("scale_cols_stats");
    if (rc) return rc;
  }
  StatOuts outs;
  memset(&outs, 0, sizeof(outs));
  outs.o[0] = to_d(d_gate); outs.o[1] = to_d(d_bias);
  return finalize(stream, (const float*)workspace, nchunks, nb, 2, N, outs, "scale_cols_stats_finalize");
}

// out (+)= column sums of a [nb x rows_per_batch rows, N] whose batch element b starts at physical row b * batch_stride_rows (a per-stream row block of a joint
// [B, S, *] buffer is summed in place: no gathered copy); per-batch rows or ONE row over all batches, fp32 or bf16 (st355_stat_out)
extern "C" int st355_colsum_rows(void* stream, const void* a, int64_t lda, int64_t rows_per_batch, int64_t batch_stride_rows, int nb, int N,
                                 const st355_stat_out* out, void* workspace) {
  ST_REQUIRE(a && out && out->out && workspace && N % 8 == 0 && lda % 8 == 0 && rows_per_batch > 0 && batch_stride_rows >= rows_per_batch && nb > 0 &&
                 ((uintptr_t)a % 16 == 0), "colsum_rows: bad args");
  const int nchunks = (int)cdiv64(rows_per_batch, SR_ROWS);
  {
    ProfScope ps(stream, ST355_K_ELEMENTWISE, 1.0 * nb * rows_per_batch * N, 2.0 * nb * rows_per_batch * N);
    hipLaunchKernelGGL(k_cols_stats, dim3((N + 511) / 512, nchunks, nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, lda, (const bf16*)nullptr, 0,
                       (const bf16*)nullptr, 0, (bf16*)nullptr, 0, rows_per_batch, batch_stride_rows, N, (float*)workspace, nchunks, 1);
    int rc = st355_check_launch("colsum_rows");
    if (rc) return rc;
  }
  StatOuts outs;
  memset(&outs, 0, sizeof(outs));
  outs.o[0] = to_d(out);
  return finalize(stream, (const float*)workspace, nchunks, nb, 1, N, outs, "colsum_rows_finalize");
}
