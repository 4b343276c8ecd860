// pp_runtime.hip -- context, error string, buffers and timing ring of
// libpp_hip.so (host code only).

#include "pp_common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace pp {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int DevBuf::ensure(size_t need, bool *grew) {
  if (grew) *grew = false;
  if (need <= bytes && ptr) return PP_OK;
  if (ptr) {
    // the previous buffer may still be in use by queued kernels
    PP_HIP_TRY(hipDeviceSynchronize());
    PP_HIP_TRY(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
  }
  if (need == 0) need = 256;
  hipError_t e = hipMalloc(&ptr, need);
  if (e != hipSuccess) {
    ptr = nullptr;
    set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? PP_ERR_NOMEM : PP_ERR_HIP;
  }
  bytes = need;
  if (grew) *grew = true;
  return PP_OK;
}

void DevBuf::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
}

int PinBuf::ensure(size_t need) {
  if (need <= bytes && ptr) return PP_OK;
  if (ptr) {
    PP_HIP_TRY(hipDeviceSynchronize());
    PP_HIP_TRY(hipHostFree(ptr));
    ptr = nullptr;
    bytes = 0;
  }
  if (need == 0) need = 256;
  need = (need + 4095) / 4096 * 4096;
  hipError_t e = hipHostMalloc(&ptr, need, hipHostMallocDefault);
  if (e != hipSuccess) {
    ptr = nullptr;
    set_error("hipHostMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    return PP_ERR_NOMEM;
  }
  bytes = need;
  return PP_OK;
}

void PinBuf::release() {
  if (ptr) (void)hipHostFree(ptr);
  ptr = nullptr;
  bytes = 0;
}

struct HostPool::Impl {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<unsigned long long> gen{0};   // bumped once per job
  std::atomic<int> pending{0};              // worker parts of the current job still running
  std::atomic<bool> stop{false};
  const std::function<void(int, int)> *job = nullptr;
  int first_part = 0, parts = 0;            // worker w runs part first_part + w of `parts`
  const std::function<void(int, int)> *deferred = nullptr;  // start() without workers: run at wait()

  void worker(int w) {
    unsigned long long seen = 0;
    for (;;) {
      // between the jobs of one call: spin (a job follows the previous one within microseconds); then sleep
      int spins = 0;
      while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed)) {
        if (++spins < 4000) {
          __builtin_ia32_pause();
        } else {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen || stop.load(); });
        }
      }
      if (stop.load()) return;
      seen = gen.load(std::memory_order_acquire);
      (*job)(first_part + w, parts);
      pending.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  void post(const std::function<void(int, int)> &fn, int first, int total) {
    job = &fn;
    first_part = first;
    parts = total;
    pending.store((int)workers.size(), std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(m);
      gen.fetch_add(1, std::memory_order_acq_rel);
    }
    cv.notify_all();
  }
  void join() {
    while (pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }
};

HostPool::HostPool(int threads) : impl_(new Impl), n_(std::max(1, threads)) {
  for (int w = 0; w + 1 < n_; ++w) impl_->workers.emplace_back([this, w] { impl_->worker(w); });
}

HostPool::~HostPool() {
  {
    std::lock_guard<std::mutex> lk(impl_->m);
    impl_->stop.store(true);
  }
  impl_->cv.notify_all();
  for (auto &t : impl_->workers) t.join();
  delete impl_;
}

void HostPool::run(const std::function<void(int, int)> &fn) {
  if (n_ == 1) {
    fn(0, 1);
    return;
  }
  impl_->post(fn, 1, n_);
  fn(0, n_);
  impl_->join();
}

void HostPool::start(const std::function<void(int, int)> &fn) {
  if (n_ == 1) {
    impl_->deferred = &fn;
    return;
  }
  impl_->post(fn, 0, n_ - 1);
}

void HostPool::wait() {
  if (n_ == 1) {
    if (impl_->deferred) (*impl_->deferred)(0, 1);
    impl_->deferred = nullptr;
    return;
  }
  impl_->join();
}

HostPool *host_pool(pp_ctx *ctx) {
  if (!ctx->pool) {
    int n = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("PP_HOST_THREADS")) n = std::max(1, std::min(64, atoi(e)));
    ctx->pool = new HostPool(n);
  }
  return ctx->pool;
}

}  // namespace pp

using namespace pp;

extern "C" const char *pp_last_error(void) { return g_last_error.c_str(); }

extern "C" const char *pp_version(void) { return "pp_hip 0.1 (gfx950)"; }

extern "C" int pp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int pp_ctx_create(int device, pp_ctx_t **out) {
  if (!out) {
    set_error("pp_ctx_create: out is NULL");
    return PP_ERR_VALUE;
  }
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (%s): libpp_hip has no CPU fallback",
              e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range [0,%d)", device, n);
    return PP_ERR_VALUE;
  }
  pp_ctx *c = new (std::nothrow) pp_ctx();
  if (!c) return PP_ERR_NOMEM;
  c->device = device;
  if (const char *e = getenv("PP_TILE_WAVES")) {  // development knobs, see pp_common.h
    const int v = atoi(e);
    if (v == 4 || v == 8 || v == 16) c->force_tile_waves = v;
  }
  *out = c;
  return PP_OK;
}

extern "C" void pp_ctx_destroy(pp_ctx_t *ctx) {
  if (!ctx) return;
  int prev = -1;
  if (hipGetDevice(&prev) == hipSuccess && prev != ctx->device) (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto &w : ctx->vox_ws) w.release();
  ctx->stage_in.release();
  ctx->stage_out.release();
  ctx->stage_out2.release();
  ctx->iou_ws.release();
  ctx->decode_ws.release();
  ctx->pfn_ws.release();
  ctx->pin_in.release();
  ctx->pin_out.release();
  ctx->pin_meta.release();
  ctx->anchors_pin.release();
  ctx->anchors_dev.release();
  delete ctx->pool;
  for (auto &e : ctx->chunk_ev)
    if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < 3; ++k) {
    for (auto &e : ctx->ev_start[k]) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_stop[k]) (void)hipEventDestroy(e);
  }
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  delete ctx;
}

extern "C" int pp_host_pool_selftest(int threads, int jobs, int n) {
  if (threads < 1 || threads > 64 || jobs < 0 || n < 0) {
    set_error("pp_host_pool_selftest: bad argument");
    return PP_ERR_VALUE;
  }
  HostPool pool(threads);
  std::vector<long long> slot(64);
  std::vector<int> seen(64);
  for (int j = 0; j < jobs; ++j) {
    // every 256th job finds the workers ASLEEP (a pause beyond their spin): the condition-variable wake-up path
    if ((j & 255) == 255) std::this_thread::sleep_for(std::chrono::microseconds(400));
    std::fill(slot.begin(), slot.end(), 0ll);
    std::fill(seen.begin(), seen.end(), 0);
    int expect_parts = 0;
    const std::function<void(int, int)> fn = [&](int part, int parts) {
      const long long i0 = (long long)n * part / parts, i1 = (long long)n * (part + 1) / parts;
      long long acc = 0;
      for (long long i = i0; i < i1; ++i) acc += i + 1;
      slot[part] += acc;          // disjoint slots: no two parts share one
      seen[part] += parts;        // (the parts count every part was told)
    };
    if (j & 1) {                  // workers only, the caller free meanwhile
      expect_parts = threads == 1 ? 1 : threads - 1;
      pool.start(fn);
      pool.wait();
    } else {
      expect_parts = threads;
      pool.run(fn);
    }
    long long total = 0;
    for (int p = 0; p < 64; ++p) {
      total += slot[p];
      if (seen[p] != (p < expect_parts ? expect_parts : 0)) return j + 1;
    }
    if (total != (long long)n * (n + 1) / 2) return j + 1;
  }
  return 0;
}

extern "C" int pp_ctx_set_timing(pp_ctx_t *ctx, int slots) {
  if (!ctx || slots < 0 || slots > 4096) {
    set_error("pp_ctx_set_timing: bad argument");
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  for (int k = 0; k < 3; ++k) {
    for (auto &e : ctx->ev_start[k]) (void)hipEventDestroy(e);
    for (auto &e : ctx->ev_stop[k]) (void)hipEventDestroy(e);
    ctx->ev_start[k].clear();
    ctx->ev_stop[k].clear();
  }
  ctx->ev_slots = 0;
  ctx->ev_next = 0;
  ctx->ev_count = 0;
  int rc = PP_OK;
  for (int i = 0; i < slots && rc == PP_OK; ++i) {
    for (int k = 0; k < 3; ++k) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        set_error("hipEventCreate failed");
        rc = PP_ERR_HIP;
        break;
      }
      ctx->ev_start[k].push_back(a);
      ctx->ev_stop[k].push_back(b);
    }
  }
  if (rc == PP_OK) ctx->ev_slots = slots;
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  return rc;
}

extern "C" int pp_ctx_read_kernel_ms(pp_ctx_t *ctx, int which, float *ms, int cap, int *count) {
  if (!ctx || !ms || !count || which < 0 || which > 2) {
    set_error("pp_ctx_read_kernel_ms: bad argument");
    return PP_ERR_VALUE;
  }
  if (!((ctx->ev_columns >> which) & 1)) {  // k_step launches: one kernel, recorded in the EMIT column
    *count = 0;
    return PP_OK;
  }
  const int n = ctx->ev_count < cap ? ctx->ev_count : cap;
  const int slots = ctx->ev_slots > 0 ? ctx->ev_slots : 1;
  int idx = (ctx->ev_next - ctx->ev_count + 2 * slots) % slots;  // oldest first
  for (int i = 0; i < n; ++i) {
    PP_HIP_TRY(hipEventSynchronize(ctx->ev_stop[which][idx]));
    PP_HIP_TRY(hipEventElapsedTime(&ms[i], ctx->ev_start[which][idx], ctx->ev_stop[which][idx]));
    idx = (idx + 1) % slots;
  }
  *count = n;
  if (which == PP_KERNEL_EMIT) ctx->ev_count = 0;   // read; the ring keeps turning
  return PP_OK;
}

extern "C" int pp_ctx_read_emit_ms(pp_ctx_t *ctx, float *ms, int cap, int *count) {
  return pp_ctx_read_kernel_ms(ctx, PP_KERNEL_EMIT, ms, cap, count);
}
