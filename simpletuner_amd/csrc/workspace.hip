// workspace.hip — st355_workspace_bytes: the per-op scratch sizes behind one entry point (kept out of runtime.hip so the standalone labs can
// compile runtime.hip without the rest of the library).
#include "common.h"

// ---- one workspace query for every op that needs caller-provided scratch (`st355_workspace_bytes(op, shape)`) ----
extern "C" size_t st355_attn_bwd_workspace(int B, int H, int S, int Sp, int d);
extern "C" size_t st355_colsum_workspace(int64_t rows, int N, int64_t rows_per_batch);
extern "C" size_t st355_skinny_tn_workspace(int64_t M, int64_t P, int R);
extern "C" size_t st355_groupnorm_workspace(int B, int H, int W, int C);
extern "C" size_t st355_layernorm_param_grads_workspace(int D);
extern "C" int64_t st355_workspace_bytes(int op, const int64_t* dims, int ndims) {
  auto need = [&](int n) { return dims != nullptr && ndims >= n; };
  switch (op) {
    case ST355_WS_ATTN_BWD: return need(5) ? (int64_t)st355_attn_bwd_workspace((int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4]) : -1;   // B, H, Sq, Sqp, d
    case ST355_WS_COLSUM: return need(3) ? (int64_t)st355_colsum_workspace(dims[0], (int)dims[1], dims[2]) : -1;                                          // rows, N, rows_per_batch
    case ST355_WS_SKINNY_TN: return need(3) ? (int64_t)st355_skinny_tn_workspace(dims[0], dims[1], (int)dims[2]) : -1;                                    // M, P, R
    case ST355_WS_GROUPNORM: return need(4) ? (int64_t)st355_groupnorm_workspace((int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3]) : -1;             // B, H, W, C
    case ST355_WS_LAYERNORM_PARAM_GRADS: return need(1) ? (int64_t)st355_layernorm_param_grads_workspace((int)dims[0]) : -1;                               // D
    case ST355_WS_GEMM_SPLITK: return need(3) ? 16 * dims[0] * dims[1] * 4 : -1;        // M, N, (K): up to 16 fp32 K-slice slabs (thin GEMMs, st355_gemm_bf16)
    case ST355_WS_GEMM_TN: return need(2) ? 16 * dims[0] * dims[1] * (ndims >= 4 && dims[3] == 9 ? 9 : 1) * 4 : -1;   // P, Q, (contraction), (taps): slabs of the weight-gradient form
    default: return -1;
  }
}
