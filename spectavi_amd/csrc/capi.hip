// capi.hip -- extern "C" surface of libspectavi.so (declared in include/spectavi_amd.h).
//
// Host-pointer entry points own the H2D/D2H traffic and scratch allocation and
// call the same device-pointer runners (l1k2_run / cascade_run / dlt_run) the
// section-3 symbols expose.  No compute happens on the host and there is no
// CPU fallback: every path ends in a HIP kernel launch or in an error status.

#include "common.h"
#include "records.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <initializer_list>
#include <map>
#include <memory>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <random>
#include <vector>

namespace spv {

namespace {
thread_local int g_status = SPV_OK;
thread_local char g_message[512] = "";

std::mutex g_cfg_mutex;
int g_device = -1;  // -1: not chosen yet
std::vector<int> g_devices;  // empty: not chosen yet
bool g_seed_fixed = false;
uint32_t g_seed = 0;
int g_gather_mode = -1;  // -1: SPECTAVI_GATHER, else automatic; SPV_GATHER_DIRECT / SPV_GATHER_RCCL
}  // namespace

int set_error(int status, const char *fmt, ...) {
  g_status = status;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_message, sizeof(g_message), fmt, ap);
  va_end(ap);
  return status;
}

void clear_error() {
  g_status = SPV_OK;
  g_message[0] = '\0';
}

// No C++ exception may cross the extern "C" boundary (the reference's own symbols let them
// escape into libffi and abort the process, src/BruteForceNnL1K2.h:75,79): every entry point
// runs its body through this.
template <typename Fn>
static int guard(Fn fn) {
  try {
    return fn();
  } catch (const std::bad_alloc &) {
    return set_error(SPV_ERR_NOMEM, "host allocation failed");
  } catch (const std::exception &e) {
    return set_error(SPV_ERR_INTERNAL, "unexpected C++ exception: %s", e.what());
  } catch (...) {
    return set_error(SPV_ERR_INTERNAL, "unexpected C++ exception");
  }
}

// ---- optional kernel timing ----------------------------------------------------------
namespace {
std::mutex g_prof_mutex;
std::atomic<bool> g_prof_on{false};
// Per kernel name: running totals of the launches whose events have completed, and the event
// pairs still in flight.  Completed pairs are folded into the totals (and their events destroyed)
// whenever the pending list grows past kProfFoldAt, when profiling is switched off, and on every
// read, so a long-running caller that leaves profiling enabled holds a bounded number of events.
struct ProfEntry {
  long long launches = 0;
  double total_ms = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
std::map<std::string, ProfEntry> g_prof;
constexpr size_t kProfFoldAt = 64;

// g_prof_mutex held.  wait = true: block on every pending pair; false: fold only finished ones.
void prof_fold(ProfEntry &e, bool wait) {
  size_t keep = 0;
  for (auto &ev : e.pending) {
    const hipError_t q = wait ? hipEventSynchronize(ev.second) : hipEventQuery(ev.second);
    if (q == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
        e.total_ms += ms;
        e.launches += 1;
      }
    } else if (q == hipErrorNotReady) {
      e.pending[keep++] = ev;
      continue;
    }
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  e.pending.resize(keep);
}
}  // namespace

ProfScope::ProfScope(const char *name, hipStream_t stream) : name_(name), stream_(stream) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  if (hipEventCreate(&start_) != hipSuccess) {
    start_ = nullptr;
    return;
  }
  (void)hipEventRecord(start_, stream_);
}

ProfScope::~ProfScope() {
  if (!start_) return;
  hipEvent_t stop = nullptr;
  if (hipEventCreate(&stop) != hipSuccess) {
    (void)hipEventDestroy(start_);
    return;
  }
  (void)hipEventRecord(stop, stream_);
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  ProfEntry &e = g_prof[name_];
  e.pending.emplace_back(start_, stop);
  if (e.pending.size() >= kProfFoldAt) prof_fold(e, false);
}

// The host-pointer entry points select their device(s) with hipSetDevice on the caller's thread
// (and on their own shard threads).  The caller's current device is part of ITS state -- a
// framework's tensors and streams hang off it -- so every such entry point restores it on exit.
struct DeviceRestore {
  int prev = -1;
  DeviceRestore() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceRestore() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceRestore(const DeviceRestore &) = delete;
  DeviceRestore &operator=(const DeviceRestore &) = delete;
};

// guard() for entry points that may switch the current device.
template <typename Fn>
static int host_guard(Fn fn) {
  DeviceRestore restore;
  return guard(fn);
}

static int use_device(int dev) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return set_error(SPV_ERR_HIP, "no HIP device available (%s); libspectavi has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (dev < 0 || dev >= count)
    return set_error(SPV_ERR_HIP, "device %d requested but only %d devices are visible", dev, count);
  SPV_HIP_CHECK(hipSetDevice(dev));
  return SPV_OK;
}

// Devices the host-pointer entry points shard over.  One entry unless
// spv_set_devices() / SPECTAVI_DEVICES ("0,1,2,3" or "all") asked for more.
static std::vector<int> device_list() {
  std::lock_guard<std::mutex> lk(g_cfg_mutex);
  if (g_devices.empty()) {
    const char *e = getenv("SPECTAVI_DEVICES");
    if (e && *e) {
      if (!strcmp(e, "all")) {
        int count = 0;
        if (hipGetDeviceCount(&count) == hipSuccess)
          for (int d = 0; d < count; ++d) g_devices.push_back(d);
      } else {
        for (const char *c = e; *c;) {
          char *end = nullptr;
          const long v = strtol(c, &end, 10);
          if (end == c) break;
          g_devices.push_back((int)v);
          c = (*end == ',') ? end + 1 : end;
        }
      }
    }
    if (g_devices.empty()) {
      if (g_device < 0) {
        const char *d = getenv("SPECTAVI_DEVICE");
        g_device = (d && *d) ? atoi(d) : 0;
      }
      g_devices.push_back(g_device);
    }
  }
  return g_devices;
}

// How the shards of a host-pointer call reach the caller's arrays: each shard copied straight
// into its slice (direct), or gathered on the first listed GPU over RCCL and copied from there
// (north_star's "RCCL gather of (idx0, idx1, d0, d1)").  spv_set_gather_mode / SPECTAVI_GATHER
// choose; left alone, RCCL is used exactly when more than one distinct device is configured.
// Automatic mode only: can a clique over exactly these devices be built?  (librccl opened,
// ncclCommInitAll done; both cached, a refusal too, so a box without a usable RCCL pays once.)
static bool rccl_clique_usable(const std::vector<int> &devs) {
  static std::mutex mu;
  static std::map<std::vector<int>, bool> known;
  std::lock_guard<std::mutex> lk(mu);
  auto it = known.find(devs);
  if (it != known.end()) return it->second;
  bool ok;
  {
    std::lock_guard<std::mutex> glk(gather_mutex());
    GatherCtx *ctx = nullptr;
    ok = gather_ctx_get(devs, true, &ctx) == SPV_OK;
  }
  if (!ok) {
    fprintf(stderr, "libspectavi: RCCL gather unavailable (%s); sharding with direct copies instead\n", g_message);
    clear_error();
  }
  known[devs] = ok;
  return ok;
}

static int gather_transport(const std::vector<int> &devs, long long total = -1) {
  int mode;
  {
    std::lock_guard<std::mutex> lk(g_cfg_mutex);
    mode = g_gather_mode;
  }
  if (mode < 0) {
    const char *e = getenv("SPECTAVI_GATHER");
    if (e && !strcmp(e, "rccl")) mode = SPV_GATHER_RCCL;
    if (e && !strcmp(e, "direct")) mode = SPV_GATHER_DIRECT;
    if (e && !strcmp(e, "copy")) mode = SPV_GATHER_PEERCOPY;
  }
  if (mode >= 0) return mode;  // asked for by name: a failure of that transport is the caller's error
  if (devs.size() < 2) return SPV_GATHER_DIRECT;
  for (size_t a = 0; a < devs.size(); ++a)
    for (size_t b = a + 1; b < devs.size(); ++b)
      if (devs[a] == devs[b]) return SPV_GATHER_DIRECT;  // a clique needs distinct devices
  // the clique a call of `total` rows would use (run_gathered lists no more ranks than rows)
  const size_t G = total < 0 ? devs.size() : (size_t)std::min<long long>((long long)devs.size(), std::max<long long>(total, 1));
  if (!rccl_clique_usable(std::vector<int>(devs.begin(), devs.begin() + G))) return SPV_GATHER_DIRECT;
  return SPV_GATHER_RCCL;
}
static bool use_rccl_gather(const std::vector<int> &devs, long long total) { return gather_transport(devs, total) != SPV_GATHER_DIRECT; }

int ensure_device() { return use_device(device_list()[0]); }

int device_cu_count() {
  static std::atomic<int> cache[64];  // per device, 0 = not asked yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  int cus = (dev >= 0 && dev < 64) ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (cus <= 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1;
    if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
  }
  return cus;
}

// Splits [0, total) into contiguous balanced shards, one per configured device, and runs
// fn(device, lo, hi) on a host thread per shard (every row of the hot path is independent:
// the reference parallelises the same loop with OpenMP, src/BruteForceNnL1K2.h:92).  Each
// shard writes straight into its slice of the caller's output, so no gather is needed
// inside one process.
template <typename Fn>
static int run_sharded(long long total, Fn fn) {
  const std::vector<int> devs = device_list();
  const int G = (int)std::min<long long>((long long)devs.size(), std::max<long long>(total, 1));
  if (G <= 1) return fn(devs[0], 0LL, total);
  std::vector<int> status(G, SPV_OK);
  std::vector<std::string> message(G);
  std::vector<std::thread> threads;
  const long long base = total / G, extra = total % G;
  threads.reserve(G);
  for (int r = 0; r < G; ++r) {
    const long long lo = r * base + std::min<long long>(r, extra);
    const long long hi = lo + base + (r < extra ? 1 : 0);
    auto shard = [&, r, lo, hi] {
      status[r] = guard([&] {
        const int st = fn(devs[r], lo, hi);
        if (st != SPV_OK) message[r] = g_message;
        return st;
      });
    };
    // a thread that cannot be started (std::system_error) must not unwind through joinable
    // threads (std::terminate): run that shard on this thread instead
    try {
      threads.emplace_back(shard);
    } catch (...) {
      shard();
    }
  }
  for (auto &t : threads) t.join();
  for (int r = 0; r < G; ++r)
    if (status[r] != SPV_OK)
      return set_error(status[r], "device %d: %s", devs[r], message[r].c_str());
  return SPV_OK;
}

namespace {

// Per-device cache of freed device buffers for the host-pointer paths: repeated calls
// (a front-end matching image pairs, RANSAC-style loops over dlt_triangulate) would
// otherwise pay five hipMalloc/hipFree pairs each.  Grow-only up to kPoolCapBytes per
// device; spv_release_cached_memory() empties it.  A buffer goes back to the pool only after
// the owning thread's stream has drained (DevBuf's destructor synchronises it: a no-op on the
// normal path, which has already synchronised, and the safety net on error paths), so a
// later owner on another stream never sees work in flight.
class DevicePool {
 public:
  static constexpr size_t kPoolCapBytes = (size_t)4 << 30;
  void *acquire(int dev, size_t bytes, size_t *got) {
    std::lock_guard<std::mutex> lk(mu_);
    auto &fl = free_[dev];
    size_t best = fl.size();
    for (size_t i = 0; i < fl.size(); ++i)
      if (fl[i].second >= bytes && fl[i].second <= 2 * bytes + 4096 &&
          (best == fl.size() || fl[i].second < fl[best].second))
        best = i;
    if (best == fl.size()) return nullptr;
    void *p = fl[best].first;
    *got = fl[best].second;
    held_[dev] -= fl[best].second;
    fl.erase(fl.begin() + best);
    return p;
  }
  void release(int dev, void *p, size_t bytes) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (held_[dev] + bytes <= kPoolCapBytes) {
        free_[dev].emplace_back(p, bytes);
        held_[dev] += bytes;
        return;
      }
    }
    (void)hipFree(p);
  }
  void clear() {
    std::lock_guard<std::mutex> lk(mu_);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : free_) {
      (void)hipSetDevice(kv.first);
      for (auto &b : kv.second) (void)hipFree(b.first);
      kv.second.clear();
    }
    held_.clear();
    (void)hipSetDevice(cur);
  }

 private:
  std::mutex mu_;
  std::map<int, std::vector<std::pair<void *, size_t>>> free_;
  std::map<int, size_t> held_;
};
DevicePool g_pool;

// RAII device buffer for the host-pointer paths (allocated on the current device).
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int dev = 0;
  ~DevBuf() {
    if (!p) return;
    (void)hipStreamSynchronize(hipStreamPerThread);
    g_pool.release(dev, p, cap);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    (void)hipGetDevice(&dev);
    p = g_pool.acquire(dev, bytes, &cap);
    if (p) return SPV_OK;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      // the cache may be what is exhausting the device: drop it and retry once
      g_pool.clear();
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
      p = nullptr;
      return set_error(SPV_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    cap = bytes;
    return SPV_OK;
  }
  template <typename T>
  T *as() {
    return static_cast<T *>(p);
  }
};

// A caller's fresh result array (np.empty) has no pages yet: the device-to-host copy then
// runs at the page-fault rate (~11 GB/s measured) instead of the link rate.  Large outputs are
// therefore touched -- one byte per page, from a few threads -- while the inputs travel and the
// kernels run; the contents of an output buffer are undefined before the call returns, so
// writing to it early is allowed.  Skipped when the output overlaps an input.
class HostPrefault {
 public:
  // Pages are touched in 2 MB blocks dealt round-robin to the threads (block j belongs to thread
  // j % T), so the front of the array is ready first and wait_range() can release a consumer
  // that only needs a prefix while the rest is still being touched.
  static constexpr size_t kBlock = (size_t)2 << 20;
  HostPrefault(void *dst, size_t bytes, std::initializer_list<std::pair<const void *, size_t>> inputs) {
    constexpr size_t kMin = (size_t)16 << 20, kPage = 4096;
    if (!dst || bytes < kMin) return;
    const char *lo = static_cast<const char *>(dst), *hi = lo + bytes;
    for (const auto &in : inputs) {
      const char *a = static_cast<const char *>(in.first);
      if (a && a < hi && a + in.second > lo) return;
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int T = (int)std::min<size_t>(std::min<unsigned>(8u, hw), bytes / kMin + 1);
    const size_t nblocks = (bytes + kBlock - 1) / kBlock;
    done_.reset(new std::atomic<size_t>[T]);
    for (int i = 0; i < T; ++i) done_[i].store(0, std::memory_order_relaxed);
    nthreads_ = T;
    try {
      for (int i = 0; i < T; ++i)
        threads_.emplace_back([=] {
          volatile char *p = static_cast<volatile char *>(dst);
          size_t mine = 0;
          for (size_t b = (size_t)i; b < nblocks; b += (size_t)T) {
            for (size_t off = b * kBlock; off < std::min(bytes, (b + 1) * kBlock); off += kPage) p[off] = 0;
            done_[i].store(++mine, std::memory_order_release);
          }
          done_[i].store(SIZE_MAX, std::memory_order_release);
        });
    } catch (...) {  // could not start a thread: the copy simply faults the pages itself
    }
    // a thread that never started owns blocks nobody touches: they count as done (the copy into
    // them then faults the pages itself, which is only slower)
    for (int i = (int)threads_.size(); i < T; ++i) done_[i].store(SIZE_MAX, std::memory_order_release);
  }
  // Returns once no prefault store can land in [off, off + len) any more: a consumer must call
  // this (or wait()) before it writes real data there -- a late `p[off] = 0` would otherwise
  // overwrite one byte per page of the result.
  void wait_range(size_t off, size_t len) {
    if (nthreads_ == 0 || len == 0) return;
    const size_t b0 = off / kBlock, b1 = (off + len - 1) / kBlock;
    for (size_t b = b0; b <= b1; ++b) {
      const int owner = (int)(b % (size_t)nthreads_);
      const size_t need = b / (size_t)nthreads_ + 1;  // blocks the owner must have finished
      while (done_[owner].load(std::memory_order_acquire) < need) std::this_thread::yield();
    }
  }
  void wait() {
    for (auto &t : threads_)
      if (t.joinable()) t.join();
    threads_.clear();
  }
  ~HostPrefault() { wait(); }

 private:
  std::vector<std::thread> threads_;
  std::unique_ptr<std::atomic<size_t>[]> done_;
  int nthreads_ = 0;
};

// ---- results back to pageable host memory ------------------------------------------------
// A device-to-host copy into a caller's ordinary (pageable) array runs at ~16 GB/s: the runtime
// stages it through pinned memory and copies out of the staging buffer with one host thread.  For
// large results the library does that staging itself with several threads: each worker owns a slice
// of every chunk, two pinned bounce buffers and its own stream; it waits for the chunk's producer
// event on the device side, copies device -> pinned at the link rate, and copies pinned -> the
// caller's array while its next slice is already in flight.  Chunks become available as the caller
// announces them (ready()), so a chunked computation overlaps its uploads and kernels with the
// download of the chunks before.
class PinnedPool {
 public:
  static constexpr size_t kCapBytes = (size_t)256 << 20;  // kept for reuse; more is handed back
  void *acquire(size_t bytes) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].second >= bytes && free_[i].second <= 2 * bytes + 4096) {
          void *p = free_[i].first;
          held_ -= free_[i].second;
          free_.erase(free_.begin() + i);
          return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu_);
    cap_[p] = bytes;
    return p;
  }
  void release(void *p) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      const size_t bytes = cap_[p];
      if (held_ + bytes <= kCapBytes) {
        free_.emplace_back(p, bytes);
        held_ += bytes;
        return;
      }
      cap_.erase(p);
    }
    (void)hipHostFree(p);
  }
  void clear() {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto &b : free_) {
      cap_.erase(b.first);
      (void)hipHostFree(b.first);
    }
    free_.clear();
    held_ = 0;
  }

 private:
  std::mutex mu_;
  std::vector<std::pair<void *, size_t>> free_;
  std::map<void *, size_t> cap_;
  size_t held_ = 0;
};
PinnedPool g_pinned;

class D2HPipeline {
 public:
  static constexpr size_t kMinBytes = (size_t)16 << 20;  // below this a plain copy is as good
  D2HPipeline(int dev, const void *d_src, void *h_dst, size_t bytes, size_t chunk_bytes)
      : dev_(dev), src_(static_cast<const char *>(d_src)), dst_(static_cast<char *>(h_dst)), bytes_(bytes),
        chunk_(std::max<size_t>(chunk_bytes, 1)), nchunks_((int)((bytes + chunk_ - 1) / chunk_)),
        events_(nchunks_, nullptr) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    workers_ = (int)std::min<size_t>(std::min<unsigned>(6u, hw), std::max<size_t>(1, std::min(chunk_, bytes) >> 20));
    piece_ = round_up((std::min(chunk_, bytes) + workers_ - 1) / workers_, 4096);
    try {
      for (int t = 0; t < workers_; ++t) threads_.emplace_back([this, t] { work(t); });
    } catch (...) {  // fewer workers than planned: the missing slices are copied by finish()
    }
    started_ = (int)threads_.size();
  }
  // The caller's array is still being pre-touched by `touch`: every slice waits for its own pages
  // before the pinned -> caller copy (call before the first ready(); touch must outlive finish()).
  void set_prefault(HostPrefault *touch) { prefault_ = touch; }
  // chunk k (bytes [k * chunk, (k+1) * chunk) of the source) is final once `ev` has passed;
  // chunks must be announced in order.  ev must outlive finish().
  void ready(int k, hipEvent_t ev) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      events_[k] = ev;
      ready_ = k + 1;
    }
    cv_.notify_all();
  }
  void abort() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      aborted_ = true;
    }
    cv_.notify_all();
  }
  int finish() {
    for (auto &t : threads_)
      if (t.joinable()) t.join();
    threads_.clear();
    if (aborted_) return SPV_OK;  // the caller reports its own error
    if (status_.load() != SPV_OK) return set_error(status_.load(), "%s", message_.c_str());
    for (int t = started_; t < workers_; ++t) work(t);  // slices of workers that never started
    if (status_.load() != SPV_OK) return set_error(status_.load(), "%s", message_.c_str());
    return SPV_OK;
  }
  ~D2HPipeline() {
    abort();
    for (auto &t : threads_)
      if (t.joinable()) t.join();
  }

 private:
  void fail(int st, const char *what, hipError_t e) {
    std::lock_guard<std::mutex> lk(mu_);
    if (status_.load() == SPV_OK) {
      message_ = std::string(what) + ": " + hipGetErrorString(e);
      status_.store(st);
    }
  }
  void work(int t) {
    if (hipSetDevice(dev_) != hipSuccess) return fail(SPV_ERR_HIP, "hipSetDevice", hipGetLastError());
    char *pin[2] = {static_cast<char *>(g_pinned.acquire(piece_)), static_cast<char *>(g_pinned.acquire(piece_))};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t st = hipStreamPerThread;
    bool ok = pin[0] && pin[1];
    if (!ok) fail(SPV_ERR_NOMEM, "hipHostMalloc", hipErrorOutOfMemory);
    for (int i = 0; ok && i < 2; ++i)
      if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) {
        ok = false;
        fail(SPV_ERR_HIP, "hipEventCreate", hipGetLastError());
      }
    size_t prev_off = 0, prev_len = 0;
    int prev_slot = -1;
    auto drain = [&] {  // the slice whose download is in flight: pinned -> the caller's array
      if (prev_slot < 0) return;
      const hipError_t e = hipEventSynchronize(done[prev_slot]);
      if (e != hipSuccess) {
        ok = false;
        fail(SPV_ERR_HIP, "device-to-host copy", e);
      } else {
        if (prefault_) prefault_->wait_range(prev_off, prev_len);
        memcpy(dst_ + prev_off, pin[prev_slot], prev_len);
      }
      prev_slot = -1;
    };
    for (int k = 0; ok && k < nchunks_; ++k) {
      hipEvent_t ev;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return ready_ > k || aborted_; });
        if (aborted_) {
          ok = false;
          break;
        }
        ev = events_[k];
      }
      const size_t c0 = (size_t)k * chunk_, c1 = std::min(bytes_, c0 + chunk_);
      const size_t off = c0 + (size_t)t * piece_;
      if (off >= c1) {
        continue;  // this chunk is shorter than t slices
      }
      const size_t len = std::min(piece_, c1 - off);
      const int slot = k & 1;
      if (prev_slot == slot) drain();  // (a skipped chunk in between) never overwrite an undrained buffer
      if (!ok) break;
      hipError_t e = ev ? hipStreamWaitEvent(st, ev, 0) : hipSuccess;
      if (e == hipSuccess) e = hipMemcpyAsync(pin[slot], src_ + off, len, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipEventRecord(done[slot], st);
      if (e != hipSuccess) {
        ok = false;
        fail(SPV_ERR_HIP, "device-to-host copy", e);
        break;
      }
      drain();  // the previous slice, while this one travels
      prev_off = off;
      prev_len = len;
      prev_slot = slot;
    }
    if (ok) drain();
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < 2; ++i) {
      if (done[i]) (void)hipEventDestroy(done[i]);
      g_pinned.release(pin[i]);
    }
  }

  int dev_;
  const char *src_;
  char *dst_;
  size_t bytes_, chunk_;
  int nchunks_;
  std::vector<hipEvent_t> events_;
  int workers_ = 1, started_ = 0;
  size_t piece_ = 0;
  HostPrefault *prefault_ = nullptr;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  int ready_ = 0;
  bool aborted_ = false;
  std::atomic<int> status_{SPV_OK};
  std::string message_;
};

// An event recorded on `st` now, destroyed with the holder.
struct ScopedEvent {
  hipEvent_t ev = nullptr;
  int record(hipStream_t st) {
    SPV_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    SPV_HIP_CHECK(hipEventRecord(ev, st));
    return SPV_OK;
  }
  ~ScopedEvent() {
    if (ev) (void)hipEventDestroy(ev);
  }
};

// Device result -> caller's array: through the threaded pinned pipeline when large, else one copy.
// `st` is the stream the producing kernels were enqueued on; returns after the data has arrived.
int download(int dev, void *h_dst, const void *d_src, size_t bytes, hipStream_t st) {
  if (bytes < D2HPipeline::kMinBytes) {
    SPV_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
    SPV_HIP_CHECK(hipStreamSynchronize(st));
    return SPV_OK;
  }
  ScopedEvent produced;
  SPV_TRY(produced.record(st));
  const size_t chunk = (size_t)8 << 20;
  D2HPipeline pipe(dev, d_src, h_dst, bytes, chunk);
  const int n = (int)((bytes + chunk - 1) / chunk);
  for (int k = 0; k < n; ++k) pipe.ready(k, produced.ev);
  const int status = pipe.finish();
  SPV_HIP_CHECK(hipStreamSynchronize(st));
  return status;
}

int check_l1k2_args(const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim, const uint64_t *idx,
                    const int32_t *dist) {
  if (xrows < 0 || yrows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  if (dim <= 0 || dim % 16 != 0)
    return set_error(SPV_ERR_INVALID,
                     "Input matrix inner dimensions must be 16-byte aligned (dim=%d).", dim);
  if (yrows > 0 && (!y || !idx || !dist || (xrows > 0 && !x))) return set_error(SPV_ERR_INVALID, "null pointer");
  return SPV_OK;
}

int host_l1k2_one(int dev, const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim,
                  uint64_t *idx, int32_t *dist) {
  SPV_TRY(check_l1k2_args(x, y, xrows, yrows, dim, idx, dist));
  if (yrows == 0) return SPV_OK;
  SPV_TRY(use_device(dev));
  const size_t xb = (size_t)xrows * dim, yb = (size_t)yrows * dim;
  const size_t wsb = spv_l1k2_workspace_bytes(xrows, yrows, dim);
  HostPrefault touch_idx(idx, (size_t)yrows * 2 * sizeof(uint64_t), {{x, xb}, {y, yb}});
  HostPrefault touch_dist(dist, (size_t)yrows * 2 * sizeof(int32_t), {{x, xb}, {y, yb}});
  DevBuf dx, dy, di, dd, ws;
  SPV_TRY(dx.alloc(xb));
  SPV_TRY(dy.alloc(yb));
  SPV_TRY(di.alloc((size_t)yrows * 2 * sizeof(uint64_t)));
  SPV_TRY(dd.alloc((size_t)yrows * 2 * sizeof(int32_t)));
  SPV_TRY(ws.alloc(wsb));
  hipStream_t st = hipStreamPerThread;  // concurrent callers (ctypes drops the GIL) do not serialise on the null stream
  if (xb) SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x, xb, hipMemcpyHostToDevice, st));
  SPV_HIP_CHECK(hipMemcpyAsync(dy.p, y, yb, hipMemcpyHostToDevice, st));
  SPV_TRY(l1k2_run(dx.as<uint8_t>(), dy.as<uint8_t>(), xrows, yrows, dim, di.as<uint64_t>(),
                   dd.as<int32_t>(), ws.p, wsb, st));
  touch_idx.wait();
  touch_dist.wait();
  SPV_HIP_CHECK(hipMemcpyAsync(dist, dd.p, (size_t)yrows * 2 * sizeof(int32_t),
                               hipMemcpyDeviceToHost, st));
  return download(dev, idx, di.p, (size_t)yrows * 2 * sizeof(uint64_t), st);
}

int host_l1k2_gathered(const std::vector<int> &devs, const uint8_t *x, const uint8_t *y, int xrows, int yrows,
                       int dim, uint64_t *idx, int32_t *dist);
int host_cascade_gathered(const std::vector<int> &devs, const float *x, const float *y, int xrows, int yrows,
                          int dim, int m, int n, int g, const float *dict, uint64_t *idx, float *dist,
                          int32_t *ncand);
int host_dlt_gathered(const std::vector<int> &devs, const double *P0, const double *P1, int npt, const double *x,
                      const double *xp, double *dst, bool want_error);

int host_l1k2(const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim, uint64_t *idx,
              int32_t *dist) {
  if (yrows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  const std::vector<int> devs = device_list();
  if (yrows > 0 && use_rccl_gather(devs, yrows)) return host_l1k2_gathered(devs, x, y, xrows, yrows, dim, idx, dist);
  return run_sharded(yrows, [&](int dev, long long lo, long long hi) {
    return host_l1k2_one(dev, x, y ? y + (size_t)lo * dim : y, xrows, (int)(hi - lo), dim,
                         idx ? idx + 2 * lo : idx, dist ? dist + 2 * lo : dist);
  });
}

int check_cascade_args(int xrows, int yrows, int dim, int m, int n, int g) {
  if (xrows < 0 || yrows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  if (dim <= 0 || dim % 16 != 0)
    return set_error(SPV_ERR_INVALID,
                     "Input matrix inner dimensions must be 16-byte aligned (dim=%d).", dim);
  if (m < 1 || m > 31) return set_error(SPV_ERR_INVALID, "hash_bit_rate m=%d must be in [1,31]", m);
  if (n < 1) return set_error(SPV_ERR_INVALID, "num_hash_tables n=%d must be >= 1", n);
  if (g < 0 || g > m || g > 16)
    return set_error(SPV_ERR_INVALID, "num_candidate_neighbours g=%d must be in [0,min(m,16)]", g);
  return SPV_OK;
}

int host_cascade_one(int dev, const float *x, const float *y, int xrows, int yrows, int dim, int m,
                     int n, int g, const float *dict, uint64_t *idx, float *dist, int32_t *ncand) {
  SPV_TRY(check_cascade_args(xrows, yrows, dim, m, n, g));
  if (yrows == 0) return SPV_OK;
  if (!y || !idx || !dist || !dict || (xrows > 0 && !x))
    return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(use_device(dev));
  const size_t xb = (size_t)xrows * dim * sizeof(float), yb = (size_t)yrows * dim * sizeof(float);
  const size_t db = (size_t)n * dim * m * sizeof(float);
  const size_t wsb = cascade_workspace_bytes(xrows, yrows, dim, m, n, g);
  HostPrefault touch_idx(idx, (size_t)yrows * 2 * sizeof(uint64_t), {{x, xb}, {y, yb}});
  DevBuf dx, dy, dd, di, dds, dn, ws;
  SPV_TRY(dx.alloc(xb));
  SPV_TRY(dy.alloc(yb));
  SPV_TRY(dd.alloc(db));
  SPV_TRY(di.alloc((size_t)yrows * 2 * sizeof(uint64_t)));
  SPV_TRY(dds.alloc((size_t)yrows * 2 * sizeof(float)));
  SPV_TRY(dn.alloc((size_t)yrows * sizeof(int32_t)));
  SPV_TRY(ws.alloc(wsb));
  hipStream_t st = hipStreamPerThread;  // concurrent callers (ctypes drops the GIL) do not serialise on the null stream
  if (xb) SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x, xb, hipMemcpyHostToDevice, st));
  SPV_HIP_CHECK(hipMemcpyAsync(dy.p, y, yb, hipMemcpyHostToDevice, st));
  SPV_HIP_CHECK(hipMemcpyAsync(dd.p, dict, db, hipMemcpyHostToDevice, st));
  SPV_TRY(cascade_run(dx.as<float>(), dy.as<float>(), xrows, yrows, dim, m, n, g, dd.as<float>(),
                      di.as<uint64_t>(), dds.as<float>(), dn.as<int32_t>(), ws.p, wsb, st));
  touch_idx.wait();
  SPV_HIP_CHECK(hipMemcpyAsync(dist, dds.p, (size_t)yrows * 2 * sizeof(float),
                               hipMemcpyDeviceToHost, st));
  if (ncand)
    SPV_HIP_CHECK(hipMemcpyAsync(ncand, dn.p, (size_t)yrows * sizeof(int32_t),
                                 hipMemcpyDeviceToHost, st));
  return download(dev, idx, di.p, (size_t)yrows * 2 * sizeof(uint64_t), st);
}

int host_cascade(const float *x, const float *y, int xrows, int yrows, int dim, int m, int n,
                 int g, const float *dict, uint64_t *idx, float *dist, int32_t *ncand) {
  SPV_TRY(check_cascade_args(xrows, yrows, dim, m, n, g));
  const std::vector<int> devs = device_list();
  if (yrows > 0 && use_rccl_gather(devs, yrows))
    return host_cascade_gathered(devs, x, y, xrows, yrows, dim, m, n, g, dict, idx, dist, ncand);
  return run_sharded(yrows, [&](int dev, long long lo, long long hi) {
    return host_cascade_one(dev, x, y ? y + (size_t)lo * dim : y, xrows, (int)(hi - lo), dim, m, n, g,
                            dict, idx ? idx + 2 * lo : idx, dist ? dist + 2 * lo : dist,
                            ncand ? ncand + lo : ncand);
  });
}

int host_dlt_one(int dev, const double *P0, const double *P1, int npt, const double *x,
                 const double *xp, double *dst, bool want_error) {
  if (npt < 0) return set_error(SPV_ERR_INVALID, "negative point count");
  if (npt == 0) return SPV_OK;
  if (!P0 || !P1 || !x || !xp || !dst) return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(use_device(dev));
  const size_t row = (want_error ? 1 : 4) * sizeof(double);
  const size_t ib = (size_t)npt * 3 * sizeof(double);
  const size_t ob = (size_t)npt * row;
  DevBuf dx, dxp, dd;
  SPV_TRY(dx.alloc(ib));
  SPV_TRY(dxp.alloc(ib));
  SPV_TRY(dd.alloc(ob));
  hipStream_t st = hipStreamPerThread;  // concurrent callers (ctypes drops the GIL) do not serialise on the null stream
  if (ob < D2HPipeline::kMinBytes) {
    HostPrefault touch(dst, ob, {{x, ib}, {xp, ib}});
    SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x, ib, hipMemcpyHostToDevice, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dxp.p, xp, ib, hipMemcpyHostToDevice, st));
    SPV_TRY(dlt_run(P0, P1, npt, dx.as<double>(), dxp.as<double>(), dd.as<double>(), want_error, st));
    touch.wait();
    SPV_HIP_CHECK(hipMemcpyAsync(dst, dd.p, ob, hipMemcpyDeviceToHost, st));
    SPV_HIP_CHECK(hipStreamSynchronize(st));
    return SPV_OK;
  }
  // Large batches are transfer-bound at this boundary (480 MB in, 320 MB out for 10M points against
  // 0.2 ms of kernel): the points are independent, so the call runs in chunks -- upload and solve
  // chunk k while the pinned pipeline brings chunk k-1's rows back over the other PCIe direction.
  // (measured on 10M points: 1M-point chunks 14.7 ms per call, 256k 16.5, 4M 18; a fresh result
  // array is touched by HostPrefault's threads meanwhile: the kernel zeroes 320 MB of new pages)
  const long long chunk_pts = 1 << 20;
  const int nchunks = (int)((npt + chunk_pts - 1) / chunk_pts);
  std::vector<ScopedEvent> produced(nchunks);
  HostPrefault touch(dst, ob, {{x, ib}, {xp, ib}});
  D2HPipeline pipe(dev, dd.p, dst, ob, (size_t)chunk_pts * row);
  pipe.set_prefault(&touch);  // a slice lands only after its own pages have been touched
  int status = SPV_OK;
  for (int k = 0; k < nchunks && status == SPV_OK; ++k) {
    const long long p0 = (long long)k * chunk_pts, cnt = std::min<long long>(chunk_pts, npt - p0);
    status = [&] {
      SPV_HIP_CHECK(hipMemcpyAsync(dx.as<double>() + 3 * p0, x + 3 * p0, (size_t)cnt * 24, hipMemcpyHostToDevice, st));
      SPV_HIP_CHECK(hipMemcpyAsync(dxp.as<double>() + 3 * p0, xp + 3 * p0, (size_t)cnt * 24, hipMemcpyHostToDevice, st));
      SPV_TRY(dlt_run(P0, P1, cnt, dx.as<double>() + 3 * p0, dxp.as<double>() + 3 * p0,
                      reinterpret_cast<double *>(dd.as<char>() + (size_t)p0 * row), want_error, st));
      SPV_TRY(produced[k].record(st));
      return SPV_OK;
    }();
    if (status == SPV_OK) pipe.ready(k, produced[k].ev);
  }
  if (status != SPV_OK) {
    const std::string msg = g_message;  // finish() may overwrite the thread's message
    pipe.abort();
    (void)pipe.finish();
    (void)hipStreamSynchronize(st);
    return set_error(status, "%s", msg.c_str());
  }
  status = pipe.finish();
  SPV_HIP_CHECK(hipStreamSynchronize(st));
  return status;
}

int host_dlt(const double *P0, const double *P1, int npt, const double *x, const double *xp,
             double *dst, bool want_error) {
  if (npt < 0) return set_error(SPV_ERR_INVALID, "negative point count");
  const int cols = want_error ? 1 : 4;
  const std::vector<int> devs = device_list();
  if (npt > 0 && use_rccl_gather(devs, npt)) return host_dlt_gathered(devs, P0, P1, npt, x, xp, dst, want_error);
  return run_sharded(npt, [&](int dev, long long lo, long long hi) {
    return host_dlt_one(dev, P0, P1, (int)(hi - lo), x ? x + 3 * lo : x, xp ? xp + 3 * lo : xp,
                        dst ? dst + cols * lo : dst, want_error);
  });
}

int host_dlt_score(const double *P0, const double *P1s, int nhyp, int npt, const double *x,
                   const double *xp, double max_error, int32_t *counts, uint8_t *mask) {
  if (npt < 0 || nhyp < 0) return set_error(SPV_ERR_INVALID, "negative count");
  if (nhyp == 0) return SPV_OK;
  if (!P0 || !P1s || !counts || (npt > 0 && (!x || !xp))) return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(ensure_device());
  const size_t ib = (size_t)npt * 3 * sizeof(double);
  DevBuf dx, dxp, dp, dc, dm, ws;
  const size_t wsb = dlt_score_workspace_bytes(nhyp, npt);
  SPV_TRY(dx.alloc(ib));
  SPV_TRY(dxp.alloc(ib));
  SPV_TRY(dp.alloc((size_t)nhyp * 12 * sizeof(double)));
  SPV_TRY(dc.alloc((size_t)nhyp * sizeof(int32_t)));
  if (mask) SPV_TRY(dm.alloc((size_t)nhyp * npt));
  SPV_TRY(ws.alloc(wsb));
  hipStream_t st = hipStreamPerThread;  // concurrent callers (ctypes drops the GIL) do not serialise on the null stream
  if (ib) {
    SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x, ib, hipMemcpyHostToDevice, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dxp.p, xp, ib, hipMemcpyHostToDevice, st));
  }
  SPV_HIP_CHECK(hipMemcpyAsync(dp.p, P1s, (size_t)nhyp * 12 * sizeof(double), hipMemcpyHostToDevice, st));
  SPV_TRY(dlt_score_run(P0, dp.as<double>(), nhyp, npt, dx.as<double>(), dxp.as<double>(), max_error,
                        dc.as<int>(), mask ? dm.as<unsigned char>() : nullptr, ws.p, wsb, st));
  SPV_HIP_CHECK(hipMemcpyAsync(counts, dc.p, (size_t)nhyp * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (mask && npt > 0)
    SPV_HIP_CHECK(hipMemcpyAsync(mask, dm.p, (size_t)nhyp * npt, hipMemcpyDeviceToHost, st));
  SPV_HIP_CHECK(hipStreamSynchronize(st));
  return SPV_OK;
}

int host_ransac_process(const double *Fs, int nF, const double *x0, const double *x1, int npt,
                        double ratio_allowed, double required_percent, double max_error, int find_best,
                        int32_t *success, int32_t *inlier_count, int32_t *best_cam, double *best_P, double *ratio,
                        double *E, int32_t *counts4, uint8_t *mask) {
  if (nF < 0 || npt < 0) return set_error(SPV_ERR_INVALID, "negative count");
  if (nF == 0) return SPV_OK;
  if (npt == 0) return set_error(SPV_ERR_INVALID, "no correspondences");
  if (!Fs || !x0 || !x1 || !success || !inlier_count || !best_cam) return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(ensure_device());
  // candidates per launch: the scoring grid takes 65535 hypotheses (4 per candidate) and the
  // per-camera masks are kept under 1 GiB
  int chunk = 16383;
  if (mask) chunk = (int)std::max<long long>(1, std::min<long long>(chunk, ((long long)1 << 28) / npt));
  chunk = std::min(chunk, nF);
  const size_t ib = (size_t)npt * 3 * sizeof(double);
  const size_t wsb = ransac_workspace_bytes(chunk, npt, mask != nullptr);
  DevBuf dx, dxp, dF, ds, dc, db, dP, dr, dE, d4, dm, ws;
  SPV_TRY(dx.alloc(ib));
  SPV_TRY(dxp.alloc(ib));
  SPV_TRY(dF.alloc((size_t)chunk * 9 * sizeof(double)));
  SPV_TRY(ds.alloc((size_t)chunk * sizeof(int32_t)));
  SPV_TRY(dc.alloc((size_t)chunk * sizeof(int32_t)));
  SPV_TRY(db.alloc((size_t)chunk * sizeof(int32_t)));
  SPV_TRY(dP.alloc((size_t)chunk * 12 * sizeof(double)));
  SPV_TRY(dr.alloc((size_t)chunk * sizeof(double)));
  SPV_TRY(dE.alloc((size_t)chunk * 9 * sizeof(double)));
  SPV_TRY(d4.alloc((size_t)chunk * 4 * sizeof(int32_t)));
  if (mask) SPV_TRY(dm.alloc((size_t)chunk * npt));
  SPV_TRY(ws.alloc(wsb));
  hipStream_t st = hipStreamPerThread;
  SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x0, ib, hipMemcpyHostToDevice, st));
  SPV_HIP_CHECK(hipMemcpyAsync(dxp.p, x1, ib, hipMemcpyHostToDevice, st));
  for (int f0 = 0; f0 < nF; f0 += chunk) {
    const int nf = std::min(chunk, nF - f0);
    SPV_HIP_CHECK(hipMemcpyAsync(dF.p, Fs + (size_t)f0 * 9, (size_t)nf * 9 * sizeof(double), hipMemcpyHostToDevice, st));
    SPV_TRY(ransac_process_run(dF.as<double>(), nf, npt, dx.as<double>(), dxp.as<double>(), ratio_allowed,
                               required_percent, max_error, find_best, ds.as<int>(), dc.as<int>(), db.as<int>(),
                               dP.as<double>(), dr.as<double>(), dE.as<double>(), d4.as<int>(),
                               mask ? dm.as<unsigned char>() : nullptr, ws.p, wsb, st));
    SPV_HIP_CHECK(hipMemcpyAsync(success + f0, ds.p, (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SPV_HIP_CHECK(hipMemcpyAsync(inlier_count + f0, dc.p, (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SPV_HIP_CHECK(hipMemcpyAsync(best_cam + f0, db.p, (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (best_P) SPV_HIP_CHECK(hipMemcpyAsync(best_P + (size_t)f0 * 12, dP.p, (size_t)nf * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (ratio) SPV_HIP_CHECK(hipMemcpyAsync(ratio + f0, dr.p, (size_t)nf * sizeof(double), hipMemcpyDeviceToHost, st));
    if (E) SPV_HIP_CHECK(hipMemcpyAsync(E + (size_t)f0 * 9, dE.p, (size_t)nf * 9 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (counts4) SPV_HIP_CHECK(hipMemcpyAsync(counts4 + (size_t)f0 * 4, d4.p, (size_t)nf * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (mask) SPV_HIP_CHECK(hipMemcpyAsync(mask + (size_t)f0 * npt, dm.p, (size_t)nf * npt, hipMemcpyDeviceToHost, st));
    SPV_HIP_CHECK(hipStreamSynchronize(st));  // the staging buffers are reused by the next chunk
  }
  return SPV_OK;
}

// ---- seven-point solver + RANSAC loop (ransac.hip) -----------------------------------
int host_seven_point(const double *x, const double *xp, int n, int32_t *nroot, double *Fs, double *basis) {
  if (n < 0) return set_error(SPV_ERR_INVALID, "negative count");
  if (n == 0) return SPV_OK;
  if (!x || !xp || !nroot || !Fs) return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(ensure_device());
  DevBuf dx, dxp, dF, dn, db;
  const size_t ib = (size_t)n * 14 * sizeof(double);
  SPV_TRY(dx.alloc(ib));
  SPV_TRY(dxp.alloc(ib));
  SPV_TRY(dF.alloc((size_t)n * 27 * sizeof(double)));
  SPV_TRY(dn.alloc((size_t)n * sizeof(int32_t)));
  if (basis) SPV_TRY(db.alloc((size_t)n * 18 * sizeof(double)));
  hipStream_t st = hipStreamPerThread;
  SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x, ib, hipMemcpyHostToDevice, st));
  SPV_HIP_CHECK(hipMemcpyAsync(dxp.p, xp, ib, hipMemcpyHostToDevice, st));
  SPV_TRY(seven_point_run(dx.as<double>(), dxp.as<double>(), n, dF.as<double>(), dn.as<int>(),
                          basis ? db.as<double>() : nullptr, st));
  SPV_HIP_CHECK(hipMemcpyAsync(Fs, dF.p, (size_t)n * 27 * sizeof(double), hipMemcpyDeviceToHost, st));
  SPV_HIP_CHECK(hipMemcpyAsync(nroot, dn.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (basis) SPV_HIP_CHECK(hipMemcpyAsync(basis, db.p, (size_t)n * 18 * sizeof(double), hipMemcpyDeviceToHost, st));
  SPV_HIP_CHECK(hipStreamSynchronize(st));
  return SPV_OK;
}

// One 7-subset of [0, N) the way the reference draws it (floyd_sample, src/RansacFitter.h:120-132):
// for r = N-7 .. N-1 a value v uniform on [1, r] is taken unless already present, else r.  (Row 0 is
// therefore never drawn: the reference's range starts at 1.)  The reference seeds a fresh mt19937 from
// std::random_device for every subset and walks an unordered_set; here one generator serves all the
// tries of a call and the rows keep their insertion order.
void floyd_sample(std::mt19937 &gen, int N, int *out) {
  int n = 0;
  for (int r = N - 7; r < N; ++r) {
    const int v = std::uniform_int_distribution<>(1, r)(gen);
    bool present = false;
    for (int i = 0; i < n; ++i) present |= (out[i] == v);
    out[n++] = present ? r : v;
  }
}

uint32_t ransac_seed(uint64_t seed) {
  if (seed) return (uint32_t)(seed ^ (seed >> 32));
  if (const char *e = getenv("SPECTAVI_RANSAC_SEED")) return (uint32_t)strtoul(e, nullptr, 0);
  return std::random_device{}();
}

int check_fit_args(const double *x0, const double *x1, int npt, int max_tries) {
  if (!x0 || !x1) return set_error(SPV_ERR_INVALID, "null pointer");
  // RansacFitter's constructor, src/RansacFitter.h:146-149
  if (npt < 10) return set_error(SPV_ERR_INVALID, "Supplied less than 10 point matches, unsupported.");
  if (max_tries < 0) return set_error(SPV_ERR_INVALID, "negative maximum_tries");
  if (max_tries > 700000000) return set_error(SPV_ERR_INVALID, "maximum_tries above 7e8 (candidate ids are 32-bit: 3 per try)");
  return SPV_OK;
}

// samples != NULL: the 7-subsets of the tries, int32[max_tries,7]; otherwise drawn from `seed`.
// inputs_on_device: x0 / x1 are device pointers (the caller's current device) and `st` the caller's stream.
int host_ransac_fit(const double *x0, const double *x1, int npt, double required_percent, double max_error,
                    int max_tries, int find_best, double ratio_allowed, const int32_t *samples, uint64_t seed,
                    int32_t *success, double *essential, double *camera, double *inlier_percent,
                    int32_t *inlier_idx, int32_t *n_inliers, int32_t *best_try, int32_t *best_root,
                    int32_t *tries_run, bool inputs_on_device = false, hipStream_t st = hipStreamPerThread) {
  SPV_TRY(check_fit_args(x0, x1, npt, max_tries));
  if (!success || !essential || !camera || !inlier_percent || !inlier_idx || !n_inliers)
    return set_error(SPV_ERR_INVALID, "null pointer");
  if (!inputs_on_device) SPV_TRY(ensure_device());
  const int batch = std::min(ransac_fit_batch_limit(npt), std::max(max_tries, 1));
  const size_t wsb = ransac_fit_workspace_bytes(batch, npt);
  const size_t ib = (size_t)npt * 3 * sizeof(double);
  DevBuf dx, dxp, ws;
  SPV_TRY(ws.alloc(wsb));
  const double *d_x0 = x0, *d_x1 = x1;
  if (!inputs_on_device) {
    SPV_TRY(dx.alloc(ib));
    SPV_TRY(dxp.alloc(ib));
    SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x0, ib, hipMemcpyHostToDevice, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dxp.p, x1, ib, hipMemcpyHostToDevice, st));
    d_x0 = dx.as<double>();
    d_x1 = dxp.as<double>();
  }
  std::mt19937 gen(samples ? 0u : ransac_seed(seed));
  auto next = [&](int first, int n, int *dst) {
    if (samples) {
      memcpy(dst, samples + (size_t)first * 7, (size_t)n * 7 * sizeof(int));
    } else {
      for (int t = 0; t < n; ++t) floyd_sample(gen, npt, dst + 7 * (size_t)t);
    }
  };
  std::vector<unsigned char> mask((size_t)npt, 0);
  int ok = 0, ninl = 0, bt = -1, br = -1, ran = 0;
  const int rc = ransac_fit_run(d_x0, d_x1, npt, required_percent, max_error, max_tries, find_best, ratio_allowed, next,
                                &ok, essential, camera, &ninl, mask.data(), &bt, &br, &ran, ws.p, wsb, batch, st);
  // the workspace goes back to the pool below: nothing may still be queued on a caller's stream
  // (DevBuf's destructor only drains this thread's own stream; error paths and a call with zero
  // tries return without the per-batch synchronisation)
  if (st != hipStreamPerThread) (void)hipStreamSynchronize(st);
  SPV_TRY(rc);
  *success = ok;
  *inlier_percent = (double)ninl / (double)npt;
  int n = 0;
  if (bt >= 0)
    for (int i = 0; i < npt; ++i)
      if (mask[i]) inlier_idx[n++] = i;
  if (n != ninl) return set_error(SPV_ERR_HIP, "inlier list has %d entries, the count was %d", n, ninl);
  *n_inliers = n;
  if (best_try) *best_try = bt;
  if (best_root) *best_root = br;
  if (tries_run) *tries_run = ran;
  return SPV_OK;
}

int host_ratio(const uint64_t *idx, const void *dist, int dist_is_float, int yrows, double min_ratio,
               int32_t *matches, int32_t *count) {
  if (yrows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  if (!count) return set_error(SPV_ERR_INVALID, "null pointer");
  *count = 0;
  if (yrows == 0) return SPV_OK;
  if (!idx || !dist || !matches) return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(ensure_device());
  DevBuf di, dd, dm, dc, ws;
  const size_t wsb = ratio_workspace_bytes(yrows);
  SPV_TRY(di.alloc((size_t)yrows * 2 * sizeof(uint64_t)));
  SPV_TRY(dd.alloc((size_t)yrows * 2 * 4));
  SPV_TRY(dm.alloc((size_t)yrows * 2 * sizeof(int32_t)));
  SPV_TRY(dc.alloc(sizeof(int32_t)));
  SPV_TRY(ws.alloc(wsb));
  hipStream_t st = hipStreamPerThread;  // concurrent callers (ctypes drops the GIL) do not serialise on the null stream
  SPV_HIP_CHECK(hipMemcpyAsync(di.p, idx, (size_t)yrows * 2 * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  SPV_HIP_CHECK(hipMemcpyAsync(dd.p, dist, (size_t)yrows * 2 * 4, hipMemcpyHostToDevice, st));
  SPV_TRY(ratio_run(di.as<uint64_t>(), dd.p, dist_is_float, yrows, min_ratio, dm.as<int>(), dc.as<int>(),
                    ws.p, wsb, st));
  SPV_HIP_CHECK(hipMemcpyAsync(count, dc.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SPV_HIP_CHECK(hipStreamSynchronize(st));
  if (*count > 0)
    SPV_HIP_CHECK(hipMemcpy(matches, dm.p, (size_t)*count * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SPV_OK;
}

int host_sift_split(const float *table, int rows, float *geom, uint8_t *desc) {
  if (rows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  if (rows == 0) return SPV_OK;
  if (!table || !geom || !desc) return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(ensure_device());
  DevBuf dt, dg, dd;
  SPV_TRY(dt.alloc((size_t)rows * 132 * sizeof(float)));
  SPV_TRY(dg.alloc((size_t)rows * 4 * sizeof(float)));
  SPV_TRY(dd.alloc((size_t)rows * 128));
  hipStream_t st = hipStreamPerThread;  // concurrent callers (ctypes drops the GIL) do not serialise on the null stream
  SPV_HIP_CHECK(hipMemcpyAsync(dt.p, table, (size_t)rows * 132 * sizeof(float), hipMemcpyHostToDevice, st));
  SPV_TRY(sift_split_run(dt.as<float>(), rows, dg.as<float>(), dd.as<uint8_t>(), st));
  SPV_HIP_CHECK(hipMemcpyAsync(geom, dg.p, (size_t)rows * 4 * sizeof(float), hipMemcpyDeviceToHost, st));
  SPV_HIP_CHECK(hipMemcpyAsync(desc, dd.p, (size_t)rows * 128, hipMemcpyDeviceToHost, st));
  SPV_HIP_CHECK(hipStreamSynchronize(st));
  return SPV_OK;
}

int host_normalize(const float *x, int rows, int dim, float *out_f32, uint8_t *out_u8) {
  if (rows < 0 || dim <= 0) return set_error(SPV_ERR_INVALID, "bad shape");
  if (rows == 0) return SPV_OK;
  if (dim == 1 && rows > 1)
    return set_error(SPV_ERR_INVALID, "normalisation of a single-column table is not supported (dim=1)");
  if (!x || (!out_f32 && !out_u8)) return set_error(SPV_ERR_INVALID, "null pointer");
  SPV_TRY(ensure_device());
  const int dim16 = (dim + 15) / 16 * 16;
  DevBuf dx, df, du, ws;
  SPV_TRY(dx.alloc((size_t)rows * dim * sizeof(float)));
  if (out_f32) SPV_TRY(df.alloc((size_t)rows * dim16 * sizeof(float)));
  if (out_u8) SPV_TRY(du.alloc((size_t)rows * dim16));
  const size_t wsb = normalize_workspace_bytes_rows(rows, dim);
  SPV_TRY(ws.alloc(wsb));
  hipStream_t st = hipStreamPerThread;  // concurrent callers (ctypes drops the GIL) do not serialise on the null stream
  SPV_HIP_CHECK(hipMemcpyAsync(dx.p, x, (size_t)rows * dim * sizeof(float), hipMemcpyHostToDevice, st));
  SPV_TRY(normalize_run(dx.as<float>(), rows, dim, out_f32 ? df.as<float>() : nullptr,
                        out_u8 ? du.as<unsigned char>() : nullptr, ws.p, wsb, st));
  if (out_f32)
    SPV_HIP_CHECK(hipMemcpyAsync(out_f32, df.p, (size_t)rows * dim16 * sizeof(float), hipMemcpyDeviceToHost, st));
  if (out_u8) SPV_HIP_CHECK(hipMemcpyAsync(out_u8, du.p, (size_t)rows * dim16, hipMemcpyDeviceToHost, st));
  SPV_HIP_CHECK(hipStreamSynchronize(st));
  return SPV_OK;
}

// ---- RCCL-gathered sharding -----------------------------------------------------------
// Device buffers of one shard (or of the root), alive until every stream of the call has drained.
struct BufList {
  std::vector<std::unique_ptr<DevBuf>> v;
  int add(size_t bytes, DevBuf **out) {
    v.emplace_back(new DevBuf);
    *out = v.back().get();
    return (*out)->alloc(bytes);
  }
};

// total rows over the listed devices: rank r runs `produce` on a host thread of its own (device
// devs[r] current, work enqueued on the clique's stream r) and leaves its rows, row_bytes[k] each,
// in K send buffers of max_cnt rows; the calling thread then gathers each of the K buffers on rank 0
// with one ncclGather per rank (gather.hip) and runs `consume` on the root: recv[k] holds
// [G][max_cnt] rows in rank order.  Everything is synchronised before the buffers are released.
template <typename Produce, typename Consume>
int run_gathered(const std::vector<int> &all_devs, long long total, const std::vector<size_t> &row_bytes,
                 Produce produce, Consume consume, int transport = SPV_GATHER_AUTO) {
  const int G = (int)std::min<long long>((long long)all_devs.size(), std::max<long long>(total, 1));
  const std::vector<int> devs(all_devs.begin(), all_devs.begin() + G);
  const size_t K = row_bytes.size();
  // (decided before the clique lock is taken: the automatic rule may itself build a clique under it)
  const bool want_rccl = (transport == SPV_GATHER_AUTO ? gather_transport(all_devs, total) : transport) == SPV_GATHER_RCCL;
  std::lock_guard<std::mutex> lk(gather_mutex());  // one clique user at a time
  for (int d : devs) SPV_TRY(use_device(d));         // fail early on a bad device number
  GatherCtx *ctx = nullptr;
  SPV_TRY(gather_ctx_get(devs, want_rccl, &ctx));
  const long long max_cnt = shard_lo(total, G, 1);   // = size of shard 0, the largest
  std::vector<BufList> bufs(G + 1);                  // [G] = the root's receive / staging buffers
  std::vector<std::vector<const void *>> send(K, std::vector<const void *>(G, nullptr));
  std::vector<int> status(G, SPV_OK);
  std::vector<std::string> message(G);
  {
    std::vector<std::thread> threads;
    threads.reserve(G);
    for (int r = 0; r < G; ++r) {
      auto shard = [&, r] {
        status[r] = guard([&] {
          int st = use_device(devs[r]);
          if (st == SPV_OK) {
            std::vector<const void *> mine(K, nullptr);
            st = produce(r, shard_lo(total, G, r), shard_lo(total, G, r + 1), max_cnt, gather_stream(ctx, r),
                         bufs[r], mine);
            for (size_t k = 0; k < K; ++k) send[k][r] = mine[k];
          }
          if (st != SPV_OK) message[r] = g_message;
          return st;
        });
      };
      try {
        threads.emplace_back(shard);
      } catch (...) {
        shard();
      }
    }
    for (auto &t : threads) t.join();
  }
  int st = SPV_OK;
  for (int r = 0; r < G && st == SPV_OK; ++r)
    if (status[r] != SPV_OK) st = set_error(status[r], "device %d: %s", devs[r], message[r].c_str());
  std::vector<const void *> recv(K, nullptr);
  if (st == SPV_OK) st = use_device(devs[0]);
  for (size_t k = 0; k < K && st == SPV_OK; ++k) {
    DevBuf *rb = nullptr;
    st = bufs[G].add((size_t)G * max_cnt * row_bytes[k], &rb);
    if (st == SPV_OK) {
      recv[k] = rb->p;
      st = gather_bytes_run(ctx, send[k], rb->p, (size_t)max_cnt * row_bytes[k]);
    }
  }
  if (st == SPV_OK) st = use_device(devs[0]);
  if (st == SPV_OK) st = consume(recv, G, max_cnt, gather_stream(ctx, 0), bufs[G]);
  // drain every rank's stream before its buffers go back to the pool, error or not
  for (int r = 0; r < G; ++r)
    if (hipSetDevice(devs[r]) == hipSuccess) {
      const hipError_t e = hipStreamSynchronize(gather_stream(ctx, r));
      if (e != hipSuccess && st == SPV_OK)
        st = set_error(SPV_ERR_HIP, "device %d: %s", devs[r], hipGetErrorString(e));
    }
  return st;
}

int host_l1k2_gathered(const std::vector<int> &devs, const uint8_t *x, const uint8_t *y, int xrows, int yrows,
                       int dim, uint64_t *idx, int32_t *dist) {
  SPV_TRY(check_l1k2_args(x, y, xrows, yrows, dim, idx, dist));
  const size_t xb = (size_t)xrows * dim;
  HostPrefault touch_idx(idx, (size_t)yrows * 2 * sizeof(uint64_t), {{x, xb}, {y, (size_t)yrows * dim}});
  auto produce = [&](int, long long lo, long long hi, long long max_cnt, hipStream_t st, BufList &b,
                     std::vector<const void *> &send) {
    const int cnt = (int)(hi - lo);
    const size_t wsb = spv_l1k2_workspace_bytes(xrows, cnt, dim);
    DevBuf *dx, *dy, *di, *dd, *ws, *rec;
    SPV_TRY(b.add(xb, &dx));
    SPV_TRY(b.add((size_t)cnt * dim, &dy));
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(uint64_t), &di));
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(int32_t), &dd));
    SPV_TRY(b.add(wsb, &ws));
    SPV_TRY(b.add((size_t)max_cnt * sizeof(Record), &rec));
    if (xb) SPV_HIP_CHECK(hipMemcpyAsync(dx->p, x, xb, hipMemcpyHostToDevice, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dy->p, y + (size_t)lo * dim, (size_t)cnt * dim, hipMemcpyHostToDevice, st));
    SPV_TRY(l1k2_run(dx->as<uint8_t>(), dy->as<uint8_t>(), xrows, cnt, dim, di->as<uint64_t>(), dd->as<int32_t>(),
                     ws->p, wsb, st));
    SPV_TRY(gather_pack_run(di->as<uint64_t>(), dd->p, cnt, rec->p, st));
    send[0] = rec->p;
    return SPV_OK;
  };
  auto consume = [&](const std::vector<const void *> &recv, int G, long long max_cnt, hipStream_t st, BufList &b) {
    DevBuf *di, *dd;
    SPV_TRY(b.add((size_t)yrows * 2 * sizeof(uint64_t), &di));
    SPV_TRY(b.add((size_t)yrows * 2 * sizeof(int32_t), &dd));
    SPV_TRY(gather_widen_run(recv[0], yrows, G, max_cnt, di->as<uint64_t>(), dd->p, st));
    touch_idx.wait();
    SPV_HIP_CHECK(hipMemcpyAsync(idx, di->p, (size_t)yrows * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dist, dd->p, (size_t)yrows * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    return SPV_OK;
  };
  return run_gathered(devs, yrows, {sizeof(Record)}, produce, consume);
}

// The same exchange with everything resident: device devs[r] already holds a replica of the database
// (d_x[r]) and its contiguous balanced query shard (d_y[r]); the widened result lands in d_idx /
// d_dist on devs[0].  Scratch comes from the per-device buffer cache, so repeated calls allocate
// nothing.  Returns after every rank's stream has drained.
int device_l1k2_gathered(const std::vector<int> &devs, const uint8_t *const *d_x, const uint8_t *const *d_y,
                         int xrows, long long yrows, int dim, uint64_t *d_idx, int32_t *d_dist, int transport) {
  auto produce = [&](int r, long long lo, long long hi, long long max_cnt, hipStream_t st, BufList &b,
                     std::vector<const void *> &send) {
    const int cnt = (int)(hi - lo);
    const size_t wsb = spv_l1k2_workspace_bytes(xrows, cnt, dim);
    DevBuf *di, *dd, *ws, *rec;
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(uint64_t), &di));
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(int32_t), &dd));
    SPV_TRY(b.add(wsb, &ws));
    SPV_TRY(b.add((size_t)max_cnt * sizeof(Record), &rec));
    SPV_TRY(l1k2_run(d_x[r], d_y[r], xrows, cnt, dim, di->as<uint64_t>(), dd->as<int32_t>(), ws->p, wsb, st));
    SPV_TRY(gather_pack_run(di->as<uint64_t>(), dd->p, cnt, rec->p, st));
    send[0] = rec->p;
    return SPV_OK;
  };
  auto consume = [&](const std::vector<const void *> &recv, int G, long long max_cnt, hipStream_t st, BufList &) {
    return gather_widen_run(recv[0], yrows, G, max_cnt, d_idx, d_dist, st);
  };
  return run_gathered(devs, yrows, {sizeof(Record)}, produce, consume, transport);
}

// Cascade hash and DLT with everything resident, the same way: per-device replicas of the database
// and of the hyperplanes (every rank rebuilds identical codes and bucket tables), query / point shards
// per device, results gathered and (cascade) widened on devs[0].
int device_cascade_gathered(const std::vector<int> &devs, const float *const *d_x, const float *const *d_y,
                            int xrows, long long yrows, int dim, int m, int n, int g, const float *const *d_dict,
                            uint64_t *d_idx, float *d_dist, int32_t *d_ncand, int transport) {
  auto produce = [&](int r, long long lo, long long hi, long long max_cnt, hipStream_t st, BufList &b,
                     std::vector<const void *> &send) {
    const int cnt = (int)(hi - lo);
    const size_t wsb = cascade_workspace_bytes(xrows, cnt, dim, m, n, g);
    DevBuf *di, *dds, *dn, *ws, *rec;
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(uint64_t), &di));
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(float), &dds));
    SPV_TRY(b.add((size_t)max_cnt * sizeof(int32_t), &dn));
    SPV_TRY(b.add(wsb, &ws));
    SPV_TRY(b.add((size_t)max_cnt * sizeof(Record), &rec));
    SPV_TRY(cascade_run(d_x[r], d_y[r], xrows, cnt, dim, m, n, g, d_dict[r], di->as<uint64_t>(), dds->as<float>(),
                        d_ncand ? dn->as<int32_t>() : nullptr, ws->p, wsb, st));
    SPV_TRY(gather_pack_run(di->as<uint64_t>(), dds->p, cnt, rec->p, st));  // float32 distances as their bits
    send[0] = rec->p;
    if (d_ncand) send[1] = dn->p;
    return SPV_OK;
  };
  auto consume = [&](const std::vector<const void *> &recv, int G, long long max_cnt, hipStream_t st, BufList &) {
    SPV_TRY(gather_widen_run(recv[0], yrows, G, max_cnt, d_idx, d_dist, st));
    if (d_ncand)
      for (int r = 0; r < G; ++r) {  // the shards' segments to their slices of the root's array
        const long long lo = shard_lo(yrows, G, r), hi = shard_lo(yrows, G, r + 1);
        SPV_HIP_CHECK(hipMemcpyAsync(d_ncand + lo, static_cast<const int32_t *>(recv[1]) + (size_t)r * max_cnt,
                                     (size_t)(hi - lo) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
      }
    return SPV_OK;
  };
  std::vector<size_t> rows = {sizeof(Record)};
  if (d_ncand) rows.push_back(sizeof(int32_t));
  return run_gathered(devs, yrows, rows, produce, consume, transport);
}

int device_dlt_gathered(const std::vector<int> &devs, const double *P0, const double *P1, long long npt,
                        const double *const *d_x, const double *const *d_xp, double *d_dst, bool want_error,
                        int transport) {
  const size_t row = (want_error ? 1 : 4) * sizeof(double);
  auto produce = [&](int r, long long lo, long long hi, long long max_cnt, hipStream_t st, BufList &b,
                     std::vector<const void *> &send) {
    DevBuf *dd;
    SPV_TRY(b.add((size_t)max_cnt * row, &dd));
    SPV_TRY(dlt_run(P0, P1, hi - lo, d_x[r], d_xp[r], dd->as<double>(), want_error, st));
    send[0] = dd->p;
    return SPV_OK;
  };
  auto consume = [&](const std::vector<const void *> &recv, int G, long long max_cnt, hipStream_t st, BufList &) {
    for (int r = 0; r < G; ++r) {  // rows are already in the ABI layout: shard segments to their slices
      const long long lo = shard_lo(npt, G, r), hi = shard_lo(npt, G, r + 1);
      SPV_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(d_dst) + (size_t)lo * row,
                                   static_cast<const char *>(recv[0]) + (size_t)r * max_cnt * row,
                                   (size_t)(hi - lo) * row, hipMemcpyDeviceToDevice, st));
    }
    return SPV_OK;
  };
  return run_gathered(devs, npt, {row}, produce, consume, transport);
}

int host_cascade_gathered(const std::vector<int> &devs, const float *x, const float *y, int xrows, int yrows,
                          int dim, int m, int n, int g, const float *dict, uint64_t *idx, float *dist,
                          int32_t *ncand) {
  if (!y || !idx || !dist || !dict || (xrows > 0 && !x)) return set_error(SPV_ERR_INVALID, "null pointer");
  const size_t xb = (size_t)xrows * dim * sizeof(float);
  const size_t db = (size_t)n * dim * m * sizeof(float);
  HostPrefault touch_idx(idx, (size_t)yrows * 2 * sizeof(uint64_t), {{x, xb}, {y, (size_t)yrows * dim * sizeof(float)}});
  auto produce = [&](int, long long lo, long long hi, long long max_cnt, hipStream_t st, BufList &b,
                     std::vector<const void *> &send) {
    const int cnt = (int)(hi - lo);
    const size_t yb = (size_t)cnt * dim * sizeof(float);
    const size_t wsb = cascade_workspace_bytes(xrows, cnt, dim, m, n, g);
    DevBuf *dx, *dy, *dd, *di, *dds, *dn, *ws, *rec;
    SPV_TRY(b.add(xb, &dx));
    SPV_TRY(b.add(yb, &dy));
    SPV_TRY(b.add(db, &dd));
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(uint64_t), &di));
    SPV_TRY(b.add((size_t)cnt * 2 * sizeof(float), &dds));
    SPV_TRY(b.add((size_t)max_cnt * sizeof(int32_t), &dn));
    SPV_TRY(b.add(wsb, &ws));
    SPV_TRY(b.add((size_t)max_cnt * sizeof(Record), &rec));
    if (xb) SPV_HIP_CHECK(hipMemcpyAsync(dx->p, x, xb, hipMemcpyHostToDevice, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dy->p, y + (size_t)lo * dim, yb, hipMemcpyHostToDevice, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dd->p, dict, db, hipMemcpyHostToDevice, st));
    SPV_TRY(cascade_run(dx->as<float>(), dy->as<float>(), xrows, cnt, dim, m, n, g, dd->as<float>(),
                        di->as<uint64_t>(), dds->as<float>(), ncand ? dn->as<int32_t>() : nullptr, ws->p, wsb, st));
    SPV_TRY(gather_pack_run(di->as<uint64_t>(), dds->p, cnt, rec->p, st));  // float32 distances as their bits
    send[0] = rec->p;
    if (ncand) send[1] = dn->p;
    return SPV_OK;
  };
  auto consume = [&](const std::vector<const void *> &recv, int G, long long max_cnt, hipStream_t st, BufList &b) {
    DevBuf *di, *dd;
    SPV_TRY(b.add((size_t)yrows * 2 * sizeof(uint64_t), &di));
    SPV_TRY(b.add((size_t)yrows * 2 * sizeof(float), &dd));
    SPV_TRY(gather_widen_run(recv[0], yrows, G, max_cnt, di->as<uint64_t>(), dd->p, st));
    touch_idx.wait();
    SPV_HIP_CHECK(hipMemcpyAsync(idx, di->p, (size_t)yrows * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dist, dd->p, (size_t)yrows * 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    if (ncand)
      for (int r = 0; r < G; ++r) {  // 4 bytes per query: the shards' segments straight to their slices
        const long long lo = shard_lo(yrows, G, r), hi = shard_lo(yrows, G, r + 1);
        SPV_HIP_CHECK(hipMemcpyAsync(ncand + lo, static_cast<const int32_t *>(recv[1]) + (size_t)r * max_cnt,
                                     (size_t)(hi - lo) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      }
    return SPV_OK;
  };
  std::vector<size_t> rows = {sizeof(Record)};
  if (ncand) rows.push_back(sizeof(int32_t));
  return run_gathered(devs, yrows, rows, produce, consume);
}

int host_dlt_gathered(const std::vector<int> &devs, const double *P0, const double *P1, int npt, const double *x,
                      const double *xp, double *dst, bool want_error) {
  if (!P0 || !P1 || !x || !xp || !dst) return set_error(SPV_ERR_INVALID, "null pointer");
  const size_t row = (want_error ? 1 : 4) * sizeof(double);
  HostPrefault touch(dst, (size_t)npt * row, {{x, (size_t)npt * 24}, {xp, (size_t)npt * 24}});
  auto produce = [&](int, long long lo, long long hi, long long max_cnt, hipStream_t st, BufList &b,
                     std::vector<const void *> &send) {
    const long long cnt = hi - lo;
    const size_t ib = (size_t)cnt * 3 * sizeof(double);
    DevBuf *dx, *dxp, *dd;
    SPV_TRY(b.add(ib, &dx));
    SPV_TRY(b.add(ib, &dxp));
    SPV_TRY(b.add((size_t)max_cnt * row, &dd));
    SPV_HIP_CHECK(hipMemcpyAsync(dx->p, x + 3 * lo, ib, hipMemcpyHostToDevice, st));
    SPV_HIP_CHECK(hipMemcpyAsync(dxp->p, xp + 3 * lo, ib, hipMemcpyHostToDevice, st));
    SPV_TRY(dlt_run(P0, P1, cnt, dx->as<double>(), dxp->as<double>(), dd->as<double>(), want_error, st));
    send[0] = dd->p;
    return SPV_OK;
  };
  auto consume = [&](const std::vector<const void *> &recv, int G, long long max_cnt, hipStream_t st, BufList &) {
    touch.wait();
    for (int r = 0; r < G; ++r) {  // rows are already in the ABI layout: shard segments to their slices
      const long long lo = shard_lo(npt, G, r), hi = shard_lo(npt, G, r + 1);
      SPV_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(dst) + (size_t)lo * row,
                                   static_cast<const char *>(recv[0]) + (size_t)r * max_cnt * row,
                                   (size_t)(hi - lo) * row, hipMemcpyDeviceToHost, st));
    }
    return SPV_OK;
  };
  return run_gathered(devs, npt, {row}, produce, consume);
}

// Hyperplanes as the reference draws them (src/CascadingHashNn.h:86-100):
// one std::mt19937 stream, std::normal_distribution<float>(0,1), table-major,
// then dim (i), then bit (j).
void fill_hash_dict(uint32_t seed, int dim, int m, int n, float *dict) {
  std::mt19937 gen(seed);
  std::normal_distribution<float> normal(0.f, 1.f);
  const size_t total = (size_t)n * dim * m;
  for (size_t e = 0; e < total; ++e) dict[e] = normal(gen);
}

// The only place that touches NdArray members besides the callers' m_data reads.  With the in-repo
// header (include/NdArray.h) the item size fixed by the Python constructor is cross-checked; when
// built against the upstream ctypes_ndarray header (make NDARRAY_INC=...) define
// SPV_NDARRAY_ITEMSIZE(arr) to its item-size member if it has one, else nothing is checked and the
// upstream ndarray_alloc sizes the buffer as it always did for the reference.
#if !defined(SPECTAVI_EXTERNAL_NDARRAY) && !defined(SPV_NDARRAY_ITEMSIZE)
#define SPV_NDARRAY_ITEMSIZE(arr) ((arr)->m_itemsize)
#endif
int alloc_out(NdArray *arr, size_t rows, size_t cols, int itemsize) {
  if (!arr) return set_error(SPV_ERR_INVALID, "null NdArray");
#ifdef SPV_NDARRAY_ITEMSIZE
  if ((int)SPV_NDARRAY_ITEMSIZE(arr) != itemsize)
    return set_error(SPV_ERR_INVALID, "NdArray itemsize %d, expected %d", (int)SPV_NDARRAY_ITEMSIZE(arr), itemsize);
#else
  (void)itemsize;
#endif
  ndarray_set_size(arr, rows, cols);
  ndarray_alloc(arr);
  if (!arr->m_data) return set_error(SPV_ERR_NOMEM, "ndarray_alloc failed");
  return SPV_OK;
}

}  // namespace
}  // namespace spv

using namespace spv;

extern "C" {

// ---- NdArray ------------------------------------------------------------------------
// In-repo implementation of the helpers the reference takes from its ctypes_ndarray submodule;
// compiled out when the library is built against the upstream header and library
// (make NDARRAY_INC=... NDARRAY_LIB=...), which then provide them.
#ifndef SPECTAVI_EXTERNAL_NDARRAY
void ndarray_set_size(NdArray *arr, size_t d0, size_t d1) {
  arr->m_ndim = 2;
  arr->m_shape[0] = d0;
  arr->m_shape[1] = d1;
  arr->m_shape[2] = arr->m_shape[3] = 1;
}
void ndarray_set_size3(NdArray *arr, size_t d0, size_t d1, size_t d2) {
  arr->m_ndim = 3;
  arr->m_shape[0] = d0;
  arr->m_shape[1] = d1;
  arr->m_shape[2] = d2;
  arr->m_shape[3] = 1;
}
int ndarray_alloc(NdArray *arr) {
  size_t n = (size_t)(arr->m_itemsize > 0 ? arr->m_itemsize : 1);
  for (int i = 0; i < arr->m_ndim; ++i) n *= arr->m_shape[i];
  if (arr->m_data) free(arr->m_data);
  arr->m_data = malloc(n ? n : 1);
  return arr->m_data ? 0 : 1;
}
void ndarray_free(NdArray *arr) {
  if (arr && arr->m_data) {
    free(arr->m_data);
    arr->m_data = nullptr;
  }
}
#endif  // !SPECTAVI_EXTERNAL_NDARRAY

// ---- status -------------------------------------------------------------------------
int spv_last_status(void) { return g_status; }
const char *spv_last_error(void) { return g_message; }
const char *spv_version(void) { return "spectavi_amd 0.1 (gfx950)"; }
int spv_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}
int spv_set_device(int device) {
  clear_error();
  if (device < 0) return set_error(SPV_ERR_INVALID, "device %d", device);
  std::lock_guard<std::mutex> lk(g_cfg_mutex);
  g_device = device;
  g_devices.assign(1, device);
  return SPV_OK;
}
int spv_set_devices(const int *devices, int count) {
  clear_error();
  if (count < 1 || !devices) return set_error(SPV_ERR_INVALID, "need at least one device");
  for (int i = 0; i < count; ++i)
    if (devices[i] < 0) return set_error(SPV_ERR_INVALID, "device %d", devices[i]);
  std::lock_guard<std::mutex> lk(g_cfg_mutex);
  g_devices.assign(devices, devices + count);
  g_device = devices[0];
  return SPV_OK;
}

int spv_set_gather_mode(int mode) {
  clear_error();
  if (mode != SPV_GATHER_AUTO && mode != SPV_GATHER_DIRECT && mode != SPV_GATHER_RCCL && mode != SPV_GATHER_PEERCOPY)
    return set_error(SPV_ERR_INVALID, "gather mode %d", mode);
  std::lock_guard<std::mutex> lk(g_cfg_mutex);
  g_gather_mode = mode;
  return SPV_OK;
}

void spv_release_cached_memory(void) {
  g_pool.clear();
  g_pinned.clear();
}

void spv_profile_enable(int on) {
  g_prof_on.store(on != 0);
  if (!on) {  // release the events of everything that has finished; totals stay readable
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    for (auto &kv : g_prof) prof_fold(kv.second, false);
  }
}

void spv_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  for (auto &kv : g_prof)
    for (auto &ev : kv.second.pending) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
  g_prof.clear();
}

int spv_profile_read(const char *kernel, long long *launches, double *total_ms) {
  clear_error();
  if (!kernel || !launches || !total_ms) return set_error(SPV_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  *launches = 0;
  *total_ms = 0.0;
  auto it = g_prof.find(kernel);
  if (it == g_prof.end()) return SPV_OK;
  prof_fold(it->second, true);
  *launches = it->second.launches;
  *total_ms = it->second.total_ms;
  return SPV_OK;
}

// ---- reference-compatible symbols ---------------------------------------------------
void nn_bruteforcel1k2(const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim,
                       int nthreads, NdArray *outidx, NdArray *outdist) {
  (void)nthreads;  // OpenMP team size in the reference; the GPU path has no use for it
  clear_error();
  if (yrows < 0) {
    set_error(SPV_ERR_INVALID, "negative row count");
    return;
  }
  (void)host_guard([&] {
    SPV_TRY(alloc_out(outidx, (size_t)yrows, 2, (int)sizeof(size_t)));
    SPV_TRY(alloc_out(outdist, (size_t)yrows, 2, (int)sizeof(int)));
    return host_l1k2(x, y, xrows, yrows, dim, static_cast<uint64_t *>(outidx->m_data),
                     static_cast<int32_t *>(outdist->m_data));
  });
}

void nn_cascading_hash(const float *x, const float *y, int xrows, int yrows, int dim, int k,
                       int hash_bit_rate, int num_hash_tables, int num_candidate_neighbours,
                       NdArray *outidx, NdArray *outdist) {
  clear_error();
  if (k != 2) {
    set_error(SPV_ERR_INVALID, "k=%d: only k=2 is defined (reference writes exactly two columns)", k);
    return;
  }
  (void)host_guard([&] {
    SPV_TRY(check_cascade_args(xrows, yrows, dim, hash_bit_rate, num_hash_tables, num_candidate_neighbours));
    SPV_TRY(alloc_out(outidx, (size_t)yrows, 2, (int)sizeof(size_t)));
    SPV_TRY(alloc_out(outdist, (size_t)yrows, 2, (int)sizeof(float)));
    uint32_t seed;
    {
      std::lock_guard<std::mutex> lk(g_cfg_mutex);
      const char *e = getenv("SPECTAVI_HASH_SEED");
      if (g_seed_fixed)
        seed = g_seed;
      else if (e && *e)
        seed = (uint32_t)strtoul(e, nullptr, 0);
      else
        seed = std::random_device{}();  // as reference src/CascadingHashNn.h:87-88
    }
    std::vector<float> dict((size_t)num_hash_tables * dim * hash_bit_rate);
    fill_hash_dict(seed, dim, hash_bit_rate, num_hash_tables, dict.data());
    return host_cascade(x, y, xrows, yrows, dim, hash_bit_rate, num_hash_tables, num_candidate_neighbours,
                        dict.data(), static_cast<uint64_t *>(outidx->m_data),
                        static_cast<float *>(outdist->m_data), nullptr);
  });
}

void dlt_triangulate(const double *P0, const double *P1, int npt, const double *x,
                     const double *xp, double *dst) {
  clear_error();
  (void)host_guard([&] { return host_dlt(P0, P1, npt, x, xp, dst, false); });
}

void dlt_reprojection_error(const double *P0, const double *P1, int npt, const double *x,
                            const double *xp, double *dst) {
  clear_error();
  (void)host_guard([&] { return host_dlt(P0, P1, npt, x, xp, dst, true); });
}

// ---- host-pointer status variants ---------------------------------------------------
int spv_nn_bruteforcel1k2(const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim,
                          uint64_t *idx, int32_t *dist) {
  clear_error();
  return host_guard([&] { return host_l1k2(x, y, xrows, yrows, dim, idx, dist); });
}

int spv_nn_cascading_hash(const float *x, const float *y, int xrows, int yrows, int dim, int m,
                          int n, int g, const float *dict, uint64_t *idx, float *dist,
                          int32_t *ncand) {
  clear_error();
  return host_guard([&] { return host_cascade(x, y, xrows, yrows, dim, m, n, g, dict, idx, dist, ncand); });
}

int spv_generate_hash_dict(uint32_t seed, int dim, int m, int n, float *dict) {
  clear_error();
  if (dim <= 0 || m <= 0 || n <= 0 || !dict) return set_error(SPV_ERR_INVALID, "bad dict shape");
  fill_hash_dict(seed, dim, m, n, dict);
  return SPV_OK;
}

void spv_set_hash_seed(uint32_t seed, int use_fixed) {
  std::lock_guard<std::mutex> lk(g_cfg_mutex);
  g_seed = seed;
  g_seed_fixed = use_fixed != 0;
}

int spv_dlt_triangulate(const double *P0, const double *P1, int npt, const double *x,
                        const double *xp, double *dst) {
  clear_error();
  return host_guard([&] { return host_dlt(P0, P1, npt, x, xp, dst, false); });
}
int spv_normalize(const float *x, int rows, int dim, float *out_f32, uint8_t *out_u8) {
  clear_error();
  return host_guard([&] { return host_normalize(x, rows, dim, out_f32, out_u8); });
}
size_t spv_normalize_workspace_bytes(int dim) { return dim <= 0 ? 0 : normalize_workspace_bytes(dim); }
size_t spv_normalize_workspace_bytes_rows(int rows, int dim) {
  return (dim <= 0 || rows < 0) ? 0 : normalize_workspace_bytes_rows(rows, dim);
}
int spv_normalize_device(const float *d_x, int rows, int dim, float *d_out_f32, uint8_t *d_out_u8,
                         void *d_ws, size_t ws_bytes, void *stream) {
  clear_error();
  return guard([&] { return normalize_run(d_x, rows, dim, d_out_f32, d_out_u8, d_ws, ws_bytes, static_cast<hipStream_t>(stream)); });
}
int spv_sift_split(const float *table, int rows, float *geom, uint8_t *desc) {
  clear_error();
  return host_guard([&] { return host_sift_split(table, rows, geom, desc); });
}
int spv_sift_split_device(const float *d_table, int rows, float *d_geom, uint8_t *d_desc,
                          void *stream) {
  clear_error();
  return guard([&] { return sift_split_run(d_table, rows, d_geom, d_desc, static_cast<hipStream_t>(stream)); });
}
int spv_gather_match_coords_device(const float *d_geom_x, const float *d_geom_y,
                                   const int32_t *d_matches, const int32_t *d_count, int capacity,
                                   double *d_x0, double *d_x1, void *stream) {
  clear_error();
  return guard([&] { return gather_match_coords_run(d_geom_x, d_geom_y, d_matches, d_count, capacity, d_x0, d_x1,
                                 static_cast<hipStream_t>(stream)); });
}
int spv_ratio_test(const uint64_t *idx, const void *dist, int dist_is_float, int yrows,
                   double min_ratio, int32_t *matches, int32_t *count) {
  clear_error();
  return host_guard([&] { return host_ratio(idx, dist, dist_is_float, yrows, min_ratio, matches, count); });
}
size_t spv_ratio_test_workspace_bytes(int yrows) { return yrows < 0 ? 0 : ratio_workspace_bytes(yrows); }
int spv_ratio_test_device(const uint64_t *d_idx, const void *d_dist, int dist_is_float, int yrows,
                          double min_ratio, int32_t *d_matches, int32_t *d_count, void *d_ws,
                          size_t ws_bytes, void *stream) {
  clear_error();
  return guard([&] { return ratio_run(d_idx, d_dist, dist_is_float, yrows, min_ratio, d_matches, d_count, d_ws, ws_bytes,
                   static_cast<hipStream_t>(stream)); });
}
int spv_dlt_score_hypotheses(const double *P0, const double *P1s, int nhyp, int npt,
                             const double *x, const double *xp, double max_error,
                             int32_t *counts, uint8_t *mask) {
  clear_error();
  return host_guard([&] { return host_dlt_score(P0, P1s, nhyp, npt, x, xp, max_error, counts, mask); });
}
int spv_ransac_process_candidates(const double *Fs, int nF, const double *x0, const double *x1, int npt,
                                  double singular_value_ratio_allowed, double required_percent_inliers,
                                  double reprojection_error_allowed, int find_best_even_in_failure,
                                  int32_t *success, int32_t *inlier_count, int32_t *best_camera, double *best_P,
                                  double *gate_ratio, double *E, int32_t *counts4, uint8_t *inlier_mask) {
  clear_error();
  return host_guard([&] {
    return host_ransac_process(Fs, nF, x0, x1, npt, singular_value_ratio_allowed, required_percent_inliers,
                               reprojection_error_allowed, find_best_even_in_failure, success, inlier_count,
                               best_camera, best_P, gate_ratio, E, counts4, inlier_mask);
  });
}
int spv_seven_point(const double *x, const double *xp, int n, int32_t *nroot, double *Fs, double *basis) {
  clear_error();
  return host_guard([&] { return host_seven_point(x, xp, n, nroot, Fs, basis); });
}
int spv_seven_point_device(const double *d_x, const double *d_xp, int n, double *d_Fs, int32_t *d_nroot,
                           double *d_basis, void *stream) {
  clear_error();
  return guard([&] { return seven_point_run(d_x, d_xp, n, d_Fs, d_nroot, d_basis, static_cast<hipStream_t>(stream)); });
}
int spv_ransac_sample(unsigned long long seed, int npt, int ntries, int32_t *samples) {
  clear_error();
  if (npt < 10) return set_error(SPV_ERR_INVALID, "Supplied less than 10 point matches, unsupported.");
  if (ntries < 0 || (ntries > 0 && !samples)) return set_error(SPV_ERR_INVALID, "bad arguments");
  std::mt19937 gen(ransac_seed(seed));
  for (int t = 0; t < ntries; ++t) floyd_sample(gen, npt, samples + 7 * (size_t)t);
  return SPV_OK;
}
int spv_ransac_fit(const double *x0, const double *x1, int npt, double required_percent_inliers,
                   double reprojection_error_allowed, int maximum_tries, int find_best_even_in_failure,
                   double singular_value_ratio_allowed, unsigned long long seed, int32_t *success,
                   double *essential, double *camera, double *inlier_percent, int32_t *inlier_idx,
                   int32_t *n_inliers, int32_t *best_try, int32_t *best_root, int32_t *tries_run) {
  clear_error();
  return host_guard([&] {
    return host_ransac_fit(x0, x1, npt, required_percent_inliers, reprojection_error_allowed, maximum_tries,
                           find_best_even_in_failure, singular_value_ratio_allowed, nullptr, seed, success, essential,
                           camera, inlier_percent, inlier_idx, n_inliers, best_try, best_root, tries_run);
  });
}
int spv_ransac_fit_device(const double *d_x0, const double *d_x1, int npt, double required_percent_inliers,
                          double reprojection_error_allowed, int maximum_tries, int find_best_even_in_failure,
                          double singular_value_ratio_allowed, unsigned long long seed, const int32_t *samples,
                          int32_t *success, double *essential, double *camera, double *inlier_percent,
                          int32_t *inlier_idx, int32_t *n_inliers, int32_t *best_try, int32_t *best_root,
                          int32_t *tries_run, void *stream) {
  clear_error();
  return guard([&] {
    return host_ransac_fit(d_x0, d_x1, npt, required_percent_inliers, reprojection_error_allowed, maximum_tries,
                           find_best_even_in_failure, singular_value_ratio_allowed, samples, seed, success, essential,
                           camera, inlier_percent, inlier_idx, n_inliers, best_try, best_root, tries_run, true,
                           static_cast<hipStream_t>(stream));
  });
}
int spv_ransac_fit_samples(const double *x0, const double *x1, int npt, double required_percent_inliers,
                           double reprojection_error_allowed, const int32_t *samples, int ntries,
                           int find_best_even_in_failure, double singular_value_ratio_allowed, int32_t *success,
                           double *essential, double *camera, double *inlier_percent, int32_t *inlier_idx,
                           int32_t *n_inliers, int32_t *best_try, int32_t *best_root, int32_t *tries_run) {
  clear_error();
  if (ntries > 0 && !samples) return set_error(SPV_ERR_INVALID, "null samples");
  return host_guard([&] {
    return host_ransac_fit(x0, x1, npt, required_percent_inliers, reprojection_error_allowed, ntries,
                           find_best_even_in_failure, singular_value_ratio_allowed, samples, 0, success, essential,
                           camera, inlier_percent, inlier_idx, n_inliers, best_try, best_root, tries_run);
  });
}

// ---- the reference's own symbols for this path (src/Spectavi.cpp:14-36, :70-87) ------------
void seven_point_algorithm(const double *x, const double *xp, int *nroot, double *dst) {
  clear_error();
  if (!nroot || !dst) {
    set_error(SPV_ERR_INVALID, "null pointer");
    return;
  }
  *nroot = 0;
  int32_t nr = 0;
  double Fs[27];
  if (host_guard([&] { return host_seven_point(x, xp, 1, &nr, Fs, nullptr); }) != SPV_OK) return;
  *nroot = nr;
  memcpy(dst, Fs, (size_t)nr * 9 * sizeof(double));  // only the roots found are written, as in the reference
}

void ransac_fitter(const double *x0, const double *x1, int npt, double required_percent_inliers,
                   double reprojection_error_allowed, int maximum_tries, bool find_best_even_in_failure,
                   double singular_value_ratio_allowed, bool progressbar, bool *success, NdArray *essential,
                   NdArray *camera, double *inlier_percent, NdArray *inlier_idx) {
  (void)progressbar;  // the reference draws a text bar on stdout per try; tries run in batches here
  clear_error();
  if (!success || !inlier_percent) {
    set_error(SPV_ERR_INVALID, "null pointer");
    return;
  }
  *success = false;
  *inlier_percent = 0.0;
  host_guard([&] {
    SPV_TRY(check_fit_args(x0, x1, npt, maximum_tries));
    int32_t ok = 0, n = 0, bt = -1;
    double F[9], P[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, pct = 0.0;  // Camera(): Identity(3,4), src/Camera.h:27
    std::vector<int32_t> idx((size_t)npt);
    SPV_TRY(host_ransac_fit(x0, x1, npt, required_percent_inliers, reprojection_error_allowed, maximum_tries,
                            find_best_even_in_failure ? 1 : 0, singular_value_ratio_allowed, nullptr, 0, &ok, F, P, &pct,
                            idx.data(), &n, &bt, nullptr, nullptr));
    // ndarray_copy_matrix of the fitter's members (src/Spectavi.cpp:82-86): an untouched
    // m_best_fit_essential_matrix / m_inlier_idx is a 0 x 0 matrix
    const bool found = bt >= 0;
    SPV_TRY(alloc_out(essential, found ? 3 : 0, found ? 3 : 0, sizeof(double)));
    if (found) memcpy(essential->m_data, F, sizeof(F));
    SPV_TRY(alloc_out(camera, 3, 4, sizeof(double)));
    memcpy(camera->m_data, P, sizeof(P));
    SPV_TRY(alloc_out(inlier_idx, found ? (size_t)n : 0, found ? 1 : 0, sizeof(int32_t)));
    if (found && n > 0) memcpy(inlier_idx->m_data, idx.data(), (size_t)n * sizeof(int32_t));
    *success = ok != 0;
    *inlier_percent = pct;
    return (int)SPV_OK;
  });
}

size_t spv_ransac_workspace_bytes(int nF, long long npt, int want_mask) {
  return (nF < 0 || npt < 0) ? 0 : ransac_workspace_bytes(nF, npt, want_mask != 0);
}
int spv_ransac_process_candidates_device(const double *d_Fs, int nF, long long npt, const double *d_x0,
                                         const double *d_x1, double singular_value_ratio_allowed,
                                         double required_percent_inliers, double reprojection_error_allowed,
                                         int find_best_even_in_failure, int32_t *d_success, int32_t *d_inlier_count,
                                         int32_t *d_best_camera, double *d_best_P, double *d_gate_ratio, double *d_E,
                                         int32_t *d_counts4, uint8_t *d_inlier_mask, void *d_ws, size_t ws_bytes,
                                         void *stream) {
  clear_error();
  return guard([&] {
    return ransac_process_run(d_Fs, nF, npt, d_x0, d_x1, singular_value_ratio_allowed, required_percent_inliers,
                              reprojection_error_allowed, find_best_even_in_failure, d_success, d_inlier_count,
                              d_best_camera, d_best_P, d_gate_ratio, d_E, d_counts4, d_inlier_mask, d_ws, ws_bytes,
                              static_cast<hipStream_t>(stream));
  });
}
int spv_dlt_score_hypotheses_device(const double *P0, const double *d_P1s, int nhyp,
                                    long long npt, const double *d_x, const double *d_xp,
                                    double max_error, int32_t *d_counts, uint8_t *d_mask,
                                    void *stream) {
  clear_error();
  return guard([&] { return dlt_score_run(P0, d_P1s, nhyp, npt, d_x, d_xp, max_error, d_counts, d_mask, nullptr, 0,
                       static_cast<hipStream_t>(stream)); });
}
size_t spv_dlt_score_workspace_bytes(int nhyp, long long npt) { return dlt_score_workspace_bytes(nhyp, npt); }
int spv_dlt_score_hypotheses_device_ws(const double *P0, const double *d_P1s, int nhyp, long long npt,
                                       const double *d_x, const double *d_xp, double max_error,
                                       int32_t *d_counts, uint8_t *d_mask, void *d_ws, size_t ws_bytes,
                                       void *stream) {
  clear_error();
  return guard([&] { return dlt_score_run(P0, d_P1s, nhyp, npt, d_x, d_xp, max_error, d_counts, d_mask, d_ws,
                       ws_bytes, static_cast<hipStream_t>(stream)); });
}
int spv_dlt_reprojection_error(const double *P0, const double *P1, int npt, const double *x,
                               const double *xp, double *dst) {
  clear_error();
  return host_guard([&] { return host_dlt(P0, P1, npt, x, xp, dst, true); });
}

// ---- device-pointer variants --------------------------------------------------------
size_t spv_l1k2_workspace_bytes(int xrows, int yrows, int dim) {
  if (dim <= 0 || dim % 16 != 0 || xrows < 0 || yrows < 0) return 0;
  const L1K2Plan p = l1k2_plan(xrows, yrows, dim);
  return p.dim_pad < 0 ? 0 : p.total_bytes;
}

int spv_l1k2_device(const uint8_t *d_x, const uint8_t *d_y, int xrows, int yrows, int dim,
                    uint64_t *d_idx, int32_t *d_dist, void *d_ws, size_t ws_bytes, void *stream) {
  clear_error();
  return guard([&] { return l1k2_run(d_x, d_y, xrows, yrows, dim, d_idx, d_dist, d_ws, ws_bytes,
                  static_cast<hipStream_t>(stream)); });
}

size_t spv_cascade_workspace_bytes(int xrows, int yrows, int dim, int m, int n, int g) {
  if (dim <= 0 || dim % 16 != 0 || m < 1 || m > 31 || n < 1 || g < 0 || g > m || xrows < 0 ||
      yrows < 0)
    return 0;
  return cascade_workspace_bytes(xrows, yrows, dim, m, n, g);
}

int spv_l1k2_gathered_device(int ndev, const int *devices, const uint8_t *const *d_x, const uint8_t *const *d_y,
                             int xrows, long long yrows_total, int dim, uint64_t *d_idx, int32_t *d_dist,
                             int transport) {
  clear_error();
  return host_guard([&] {
    if (ndev < 1 || ndev > 64 || !devices || !d_x || !d_y) return set_error(SPV_ERR_INVALID, "bad device list");
    if (transport != SPV_GATHER_RCCL && transport != SPV_GATHER_PEERCOPY)
      return set_error(SPV_ERR_INVALID, "transport must be SPV_GATHER_RCCL or SPV_GATHER_PEERCOPY");
    if (xrows < 0 || yrows_total < 0 || yrows_total > (long long)INT32_MAX * ndev)
      return set_error(SPV_ERR_INVALID, "bad row count");
    if (dim <= 0 || dim % 16 != 0)
      return set_error(SPV_ERR_INVALID, "Input matrix inner dimensions must be 16-byte aligned (dim=%d).", dim);
    if (yrows_total == 0) return (int)SPV_OK;
    if (!d_idx || !d_dist) return set_error(SPV_ERR_INVALID, "null pointer");
    const std::vector<int> devs(devices, devices + ndev);
    const int G = (int)std::min<long long>(ndev, yrows_total);
    for (int r = 0; r < G; ++r) {
      if (!d_y[r] || (xrows > 0 && !d_x[r])) return set_error(SPV_ERR_INVALID, "null pointer (rank %d)", r);
      if (((uintptr_t)d_y[r] | (uintptr_t)d_x[r]) & 15)
        return set_error(SPV_ERR_INVALID, "device pointers must be 16-byte aligned (rank %d)", r);
    }
    if (((uintptr_t)d_idx | (uintptr_t)d_dist) & 15) return set_error(SPV_ERR_INVALID, "device pointers must be 16-byte aligned");
    return device_l1k2_gathered(devs, d_x, d_y, xrows, yrows_total, dim, d_idx, d_dist, transport);
  });
}

static int check_gathered_devices(int ndev, const int *devices, int transport) {
  if (ndev < 1 || ndev > 64 || !devices) return set_error(SPV_ERR_INVALID, "bad device list");
  if (transport != SPV_GATHER_RCCL && transport != SPV_GATHER_PEERCOPY)
    return set_error(SPV_ERR_INVALID, "transport must be SPV_GATHER_RCCL or SPV_GATHER_PEERCOPY");
  return SPV_OK;
}

int spv_cascade_gathered_device(int ndev, const int *devices, const float *const *d_x, const float *const *d_y,
                                int xrows, long long yrows_total, int dim, int m, int n, int g,
                                const float *const *d_dict, uint64_t *d_idx, float *d_dist, int32_t *d_ncand,
                                int transport) {
  clear_error();
  return host_guard([&] {
    SPV_TRY(check_gathered_devices(ndev, devices, transport));
    if (!d_x || !d_y || !d_dict) return set_error(SPV_ERR_INVALID, "null pointer");
    if (yrows_total < 0 || yrows_total > (long long)INT32_MAX * ndev) return set_error(SPV_ERR_INVALID, "bad row count");
    SPV_TRY(check_cascade_args(xrows, 0, dim, m, n, g));
    if (yrows_total == 0) return (int)SPV_OK;
    if (!d_idx || !d_dist) return set_error(SPV_ERR_INVALID, "null pointer");
    const int G = (int)std::min<long long>(ndev, yrows_total);
    for (int r = 0; r < G; ++r)
      if (!d_y[r] || !d_dict[r] || (xrows > 0 && !d_x[r])) return set_error(SPV_ERR_INVALID, "null pointer (rank %d)", r);
    return device_cascade_gathered(std::vector<int>(devices, devices + ndev), d_x, d_y, xrows, yrows_total, dim, m, n, g,
                                   d_dict, d_idx, d_dist, d_ncand, transport);
  });
}

int spv_dlt_gathered_device(int ndev, const int *devices, const double *P0, const double *P1, long long npt_total,
                            const double *const *d_x, const double *const *d_xp, double *d_dst, int want_error,
                            int transport) {
  clear_error();
  return host_guard([&] {
    SPV_TRY(check_gathered_devices(ndev, devices, transport));
    if (npt_total < 0) return set_error(SPV_ERR_INVALID, "negative point count");
    if (npt_total == 0) return (int)SPV_OK;
    if (!P0 || !P1 || !d_x || !d_xp || !d_dst) return set_error(SPV_ERR_INVALID, "null pointer");
    const int G = (int)std::min<long long>(ndev, npt_total);
    for (int r = 0; r < G; ++r)
      if (!d_x[r] || !d_xp[r]) return set_error(SPV_ERR_INVALID, "null pointer (rank %d)", r);
    return device_dlt_gathered(std::vector<int>(devices, devices + ndev), P0, P1, npt_total, d_x, d_xp, d_dst,
                               want_error != 0, transport);
  });
}

long long spv_shard_lo(long long total, int shards, int r) {
  if (shards < 1 || r < 0) return 0;
  if (r >= shards) return total;
  return shard_lo(total, shards, r);
}

int spv_cascade_device(const float *d_x, const float *d_y, int xrows, int yrows, int dim, int m,
                       int n, int g, const float *d_dict, uint64_t *d_idx, float *d_dist,
                       int32_t *d_ncand, void *d_ws, size_t ws_bytes, void *stream) {
  clear_error();
  return guard([&] {
    int s = check_cascade_args(xrows, yrows, dim, m, n, g);
    if (s != SPV_OK) return s;
    return cascade_run(d_x, d_y, xrows, yrows, dim, m, n, g, d_dict, d_idx, d_dist, d_ncand, d_ws,
                       ws_bytes, static_cast<hipStream_t>(stream));
  });
}

int spv_dlt_triangulate_device(const double *P0, const double *P1, long long npt,
                               const double *d_x, const double *d_xp, double *d_dst,
                               void *stream) {
  clear_error();
  return guard([&] { return dlt_run(P0, P1, npt, d_x, d_xp, d_dst, false, static_cast<hipStream_t>(stream)); });
}
int spv_dlt_reprojection_error_device(const double *P0, const double *P1, long long npt,
                                      const double *d_x, const double *d_xp, double *d_dst,
                                      void *stream) {
  clear_error();
  return guard([&] { return dlt_run(P0, P1, npt, d_x, d_xp, d_dst, true, static_cast<hipStream_t>(stream)); });
}

}  // extern "C"
