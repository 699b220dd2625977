// runtime.hip — library bookkeeping: version, error text, hipEvent launch profiler.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void st355_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int st355_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    st355_set_error("%s: %s", what, hipGetErrorString(e));
    return ST355_EFAULT;
  }
  return ST355_OK;
}

extern "C" int st355_version(void) { return 1000 * 0 + 1; }
extern "C" const char* st355_arch(void) { return "gfx950"; }
extern "C" const char* st355_last_error(void) { return g_err; }

// ---- profiler -----------------------------------------------------------------------------------
struct ProfRec {
  hipEvent_t a, b;
  int klass;
  double flops, bytes;
  char tag[48];
};
static std::mutex g_pm;
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(void* s, int klass, double flops, double bytes, const char* tag_fmt, ...) : idx(-1), stream(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_pm);
  ProfRec r;
  r.tag[0] = 0;
  if (tag_fmt) {
    va_list ap;
    va_start(ap, tag_fmt);
    vsnprintf(r.tag, sizeof(r.tag), tag_fmt, ap);
    va_end(ap);
  }
  r.a = get_event();
  r.b = get_event();
  r.klass = klass;
  r.flops = flops;
  r.bytes = bytes;
  hipEventRecord(r.a, (hipStream_t)s);
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_pm);
  if (idx < (int)g_recs.size()) hipEventRecord(g_recs[idx].b, (hipStream_t)stream);
}

extern "C" int st355_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_pm);
  g_prof_on = on != 0;
  return ST355_OK;
}
extern "C" int st355_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_pm);
  for (auto& r : g_recs) {
    hipEventSynchronize(r.b);
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return ST355_OK;
}
extern "C" int st355_prof_collect(double* ms, int64_t* launches, double* flops, double* bytes, int n) {
  std::lock_guard<std::mutex> lk(g_pm);
  for (int i = 0; i < n; i++) {
    ms[i] = 0;
    launches[i] = 0;
    flops[i] = 0;
    bytes[i] = 0;
  }
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    if (r.klass < n) {
      ms[r.klass] += t;
      launches[r.klass] += 1;
      flops[r.klass] += r.flops;
      bytes[r.klass] += r.bytes;
    }
  }
  return ST355_OK;
}

// one CSV line per recorded launch: class,ms,flops,bytes,tag  (call after the stream is idle; bench.py --prof-dump)
extern "C" int st355_prof_dump(const char* path) {
  std::lock_guard<std::mutex> lk(g_pm);
  FILE* f = fopen(path, "w");
  if (!f) { st355_set_error("prof_dump: cannot open %s", path); return ST355_EINVAL; }
  fprintf(f, "class,ms,flops,bytes,tag\n");
  for (auto& r : g_recs) {
    float t = 0;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    fprintf(f, "%d,%.5f,%.6g,%.6g,%s\n", r.klass, t, r.flops, r.bytes, r.tag);
  }
  fclose(f);
  return ST355_OK;
}

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
