// gather.hip -- the one exchange step of the sharded hot path, inside libspectavi.so.
//
// The reference parallelises the query loop with OpenMP threads that own whole output rows
// (src/BruteForceNnL1K2.h:92-93); sharded over the GPUs of one node the same loop needs exactly
// one collective: the per-GPU (idx0, idx1, d0, d1) records gathered on the root GPU.
// This file is that step for the single-process host-pointer entry points
// (spv_set_devices / SPECTAVI_DEVICES with SPECTAVI_GATHER=rccl): one communicator clique from
// ncclCommInitAll (cached per device list), one ncclGather per rank inside a group call, then on
// the root one widening kernel into the ABI layout (size_t idx[N,2], 32-bit dist[N,2]) and one
// copy to the caller's arrays.  A second transport, SPECTAVI_GATHER=copy, moves the same blocks
// into the same root layout with hipMemcpyPeerAsync (no RCCL in the process; also the only
// gathering transport that accepts a device listed twice, which is how the multi-rank layout --
// ragged shards, padding, rank order -- is exercised on a one-GPU box).  The one-process-per-GPU form of the same exchange is
// spectavi_amd/sharded.py (torch.distributed, backend nccl = RCCL).
//
// librccl is ~570 MB: it is opened on first use (dlopen by soname, so a copy already mapped by
// PyTorch is reused), not at library load.

#include "common.h"
#include "records.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace spv {

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mutex;  // guards the loader, the communicator cache and every use of a clique
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.handle) return SPV_OK;
  const char *env = getenv("SPECTAVI_RCCL_LIB");
  const char *names[] = {env && *env ? env : "librccl.so.1", "librccl.so.1", "librccl.so",
                         "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h)
    return set_error(SPV_ERR_HIP, "cannot open librccl (%s); set SPECTAVI_RCCL_LIB, or SPECTAVI_GATHER=direct "
                                  "to shard without the RCCL gather", dlerror());
  Rccl r;
  r.handle = h;
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.Gather = reinterpret_cast<decltype(r.Gather)>(dlsym(h, "ncclGather"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!r.CommInitAll || !r.CommDestroy || !r.Gather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
    dlclose(h);
    return set_error(SPV_ERR_HIP, "librccl lacks ncclCommInitAll / ncclGather / ncclGroupStart");
  }
  g_rccl = r;
  return SPV_OK;
}

#define SPV_NCCL_CHECK(expr)                                                                  \
  do {                                                                                        \
    ncclResult_t _r = (expr);                                                                 \
    if (_r != ncclSuccess)                                                                    \
      return set_error(SPV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), \
                       __FILE__, __LINE__);                                                   \
  } while (0)

}  // namespace

// One stream per rank (rank r = position r in the device list) and, for the RCCL transport, one
// communicator per rank; for the peer-copy transport one event per rank instead.
struct GatherCtx {
  std::vector<int> devs;
  bool rccl = true;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> arrived;
};

namespace {
std::map<std::pair<std::vector<int>, bool>, std::unique_ptr<GatherCtx>> g_ctx;

__global__ __launch_bounds__(256) void pack_records_kernel(const uint64_t *__restrict__ idx,
                                                           const uint32_t *__restrict__ d32, long long cnt,
                                                           Record *__restrict__ rec) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const ulonglong2 id = reinterpret_cast<const ulonglong2 *>(idx)[i];
  const uint2 d = reinterpret_cast<const uint2 *>(d32)[i];
  rec[i] = record_pack(id.x, id.y, d.x, d.y);
}

// recv: [G][max_cnt] records in rank order -> the ABI layout over all `total` rows.
__global__ __launch_bounds__(256) void widen_records_kernel(const Record *__restrict__ recv, long long total,
                                                            int G, long long max_cnt,
                                                            uint64_t *__restrict__ idx,
                                                            uint32_t *__restrict__ d32) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= total) return;
  int r;
  long long i;
  shard_locate(q, total, G, &r, &i);
  const Record v = recv[(size_t)r * max_cnt + i];
  reinterpret_cast<ulonglong2 *>(idx)[q] = make_ulonglong2(record_widen_idx(v.idx0), record_widen_idx(v.idx1));
  reinterpret_cast<uint2 *>(d32)[q] = make_uint2(v.d0, v.d1);
}
}  // namespace

std::mutex &gather_mutex() { return g_rccl_mutex; }

// Caller holds gather_mutex().  The clique (ncclCommInitAll takes seconds) and its streams live
// until the process ends.
int gather_ctx_get(const std::vector<int> &devs, bool use_rccl, GatherCtx **out) {
  const auto key = std::make_pair(devs, use_rccl);
  auto it = g_ctx.find(key);
  if (it != g_ctx.end()) {
    *out = it->second.get();
    return SPV_OK;
  }
  std::unique_ptr<GatherCtx> ctx(new GatherCtx);
  ctx->devs = devs;
  ctx->rccl = use_rccl;
  if (use_rccl) {
    SPV_TRY(rccl_load());
    for (size_t a = 0; a < devs.size(); ++a)
      for (size_t b = a + 1; b < devs.size(); ++b)
        if (devs[a] == devs[b])
          return set_error(SPV_ERR_INVALID, "device %d is listed twice: an RCCL clique needs distinct devices "
                                            "(use SPECTAVI_GATHER=copy or direct for such a list)", devs[a]);
    ctx->comms.assign(devs.size(), nullptr);
    SPV_NCCL_CHECK(g_rccl.CommInitAll(ctx->comms.data(), (int)devs.size(), devs.data()));
  }
  ctx->streams.assign(devs.size(), nullptr);
  ctx->arrived.assign(devs.size(), nullptr);
  for (size_t r = 0; r < devs.size(); ++r) {
    SPV_HIP_CHECK(hipSetDevice(devs[r]));
    SPV_HIP_CHECK(hipStreamCreateWithFlags(&ctx->streams[r], hipStreamNonBlocking));
    if (!use_rccl) SPV_HIP_CHECK(hipEventCreateWithFlags(&ctx->arrived[r], hipEventDisableTiming));
  }
  *out = ctx.get();
  g_ctx[key] = std::move(ctx);
  return SPV_OK;
}

hipStream_t gather_stream(GatherCtx *ctx, int r) { return ctx->streams[r]; }

int gather_pack_run(const uint64_t *d_idx, const void *d_d32, long long cnt, void *d_rec, hipStream_t stream) {
  if (cnt <= 0) return SPV_OK;
  const long long blocks = (cnt + 255) / 256;
  hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_idx,
                     static_cast<const uint32_t *>(d_d32), cnt, static_cast<Record *>(d_rec));
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

// Every rank r sends `bytes_per_rank` bytes from d_send[r] (a buffer on devs[r]); the root
// (rank 0) receives them in rank order into d_recv_root.  Enqueued on the ranks' clique streams;
// the caller synchronises.  Caller holds gather_mutex().
int gather_bytes_run(GatherCtx *ctx, const std::vector<const void *> &d_send, void *d_recv_root,
                     size_t bytes_per_rank) {
  const int G = (int)ctx->devs.size();
  if ((int)d_send.size() != G) return set_error(SPV_ERR_INTERNAL, "gather: %zu buffers for %d ranks", d_send.size(), G);
  SPV_HIP_CHECK(hipSetDevice(ctx->devs[0]));
  ProfScope prof("gather", ctx->streams[0]);
  if (!ctx->rccl) {
    // peer-copy transport: rank r's stream copies its block into slot r of the root's buffer (the
    // same layout ncclGather produces); the root's stream then waits for every rank's copy
    for (int r = 0; r < G; ++r) {
      SPV_HIP_CHECK(hipSetDevice(ctx->devs[r]));
      SPV_HIP_CHECK(hipMemcpyPeerAsync(static_cast<char *>(d_recv_root) + (size_t)r * bytes_per_rank, ctx->devs[0],
                                       d_send[r], ctx->devs[r], bytes_per_rank, ctx->streams[r]));
      SPV_HIP_CHECK(hipEventRecord(ctx->arrived[r], ctx->streams[r]));
    }
    SPV_HIP_CHECK(hipSetDevice(ctx->devs[0]));
    for (int r = 1; r < G; ++r) SPV_HIP_CHECK(hipStreamWaitEvent(ctx->streams[0], ctx->arrived[r], 0));
    return SPV_OK;
  }
  SPV_NCCL_CHECK(g_rccl.GroupStart());
  for (int r = 0; r < G; ++r) {
    const ncclResult_t res = g_rccl.Gather(d_send[r], r == 0 ? d_recv_root : nullptr, bytes_per_rank, ncclInt8, 0,
                                           ctx->comms[r], ctx->streams[r]);
    if (res != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return set_error(SPV_ERR_HIP, "ncclGather (rank %d) failed: %s", r, g_rccl.GetErrorString(res));
    }
  }
  SPV_NCCL_CHECK(g_rccl.GroupEnd());
  return SPV_OK;
}

int gather_widen_run(const void *d_recv, long long total, int G, long long max_cnt, uint64_t *d_idx, void *d_d32,
                     hipStream_t stream) {
  if (total <= 0) return SPV_OK;
  ProfScope prof("gather_widen", stream);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(widen_records_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     static_cast<const Record *>(d_recv), total, G, max_cnt, d_idx, static_cast<uint32_t *>(d_d32));
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv

extern "C" {

// Host statement of the record format (no GPU involved): for callers that run their own
// collective on raw records, and for the unit tests of the pack / widen arithmetic.
int spv_records_pack(const uint64_t *idx, const void *dist32, long long n, int32_t *rec) {
  spv::clear_error();
  if (n < 0 || (n > 0 && (!idx || !dist32 || !rec))) return spv::set_error(SPV_ERR_INVALID, "bad arguments");
  const uint32_t *d = static_cast<const uint32_t *>(dist32);
  spv::Record *out = reinterpret_cast<spv::Record *>(rec);
  for (long long i = 0; i < n; ++i) out[i] = spv::record_pack(idx[2 * i], idx[2 * i + 1], d[2 * i], d[2 * i + 1]);
  return SPV_OK;
}

// rec: [G][max_cnt] records in rank order (ragged shards padded to max_cnt) -> idx uint64[total,2],
// dist32 [total,2].
int spv_records_unpack(const int32_t *rec, long long total, int G, long long max_cnt, uint64_t *idx, void *dist32) {
  spv::clear_error();
  if (total < 0 || G < 1 || max_cnt < 0 || (total > 0 && (!rec || !idx || !dist32)))
    return spv::set_error(SPV_ERR_INVALID, "bad arguments");
  if (total > 0 && spv::shard_lo(total, G, 1) > max_cnt)
    return spv::set_error(SPV_ERR_INVALID, "max_cnt %lld is smaller than the largest shard", max_cnt);
  const spv::Record *in = reinterpret_cast<const spv::Record *>(rec);
  uint32_t *d = static_cast<uint32_t *>(dist32);
  for (long long q = 0; q < total; ++q) {
    int r;
    long long i;
    spv::shard_locate(q, total, G, &r, &i);
    const spv::Record v = in[(size_t)r * max_cnt + i];
    idx[2 * q] = spv::record_widen_idx(v.idx0);
    idx[2 * q + 1] = spv::record_widen_idx(v.idx1);
    d[2 * q] = v.d0;
    d[2 * q + 1] = v.d1;
  }
  return SPV_OK;
}

}  // extern "C"
