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
