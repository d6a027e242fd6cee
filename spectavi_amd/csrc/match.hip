// match.hip -- ratio test + ordered match compaction on device.
//
// The step immediately after the NN path in the reference's pipeline
// (example/ex01_essential_estimation.py:102-106):
//     ratio    = nn_dist[:, 1] / nn_dist[:, 0].astype('float64')
//     pass_idx = ratio >= min_ratio
//     idx0     = nn_idx[:, 0];  xd = x[idx0[pass_idx]];  yd = y[pass_idx]
// Doing it on device means only the surviving (query, database) index pairs leave
// HBM instead of all N result rows.  Semantics follow numpy's IEEE division:
// d1/0 = inf passes, 0/0 = NaN fails; additionally a query without any
// neighbour (idx0 == (size_t)-1, which would be an IndexError in the
// reference's fancy indexing) never passes.  Output order = ascending query
// index, exactly the order boolean-mask indexing produces.
//
// Three small kernels: per-block pass counts (wave ballots; a block takes 1024 queries, four per
// lane), an exclusive scan of the block counts (one workgroup), an ordered scatter.

#include "common.h"

namespace spv {
namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = 4;                       // queries per lane: a block takes 1024 consecutive queries
constexpr int kBlockQ = kThreads * kPerThread;

// One 16-byte load for the index pair and one 8-byte load for the distance pair of a query
// (round 3; before: three scalar loads per query at strides of 16 and 8 bytes).
template <typename DistT>
__device__ __forceinline__ bool passes(const uint64_t *idx, const DistT *dist, int q, double min_ratio, uint64_t &i0) {
  const ulonglong2 id = reinterpret_cast<const ulonglong2 *>(idx)[q];
  typedef DistT pair_t __attribute__((ext_vector_type(2)));
  const pair_t d = reinterpret_cast<const pair_t *>(dist)[q];
  i0 = id.x;
  if (id.x == ~0ull) return false;
  const double ratio = (double)d[1] / (double)d[0];
  return ratio >= min_ratio;
}

// Lane t of a block owns queries base + t + 256 k, k < 4 (coalesced loads per k); the pass flags of a
// wave's four rounds are four ballots.  Output order = ascending query index: round k of wave w comes
// after every earlier (k', w') pair of the block in (k, w) order, so the per-round ballot counts are
// kept as a [4 rounds][4 waves] table.
template <typename DistT>
__global__ __launch_bounds__(kThreads) void ratio_count_kernel(const uint64_t *__restrict__ idx,
                                                               const DistT *__restrict__ dist, int n,
                                                               double min_ratio,
                                                               int *__restrict__ block_counts) {
  __shared__ int wsum[kThreads / 64];
  const int base = blockIdx.x * kBlockQ;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int q = base + k * kThreads + threadIdx.x;
    uint64_t i0;
    const bool p = q < n && passes(idx, dist, q, min_ratio, i0);
    cnt += __popcll(__ballot(p));
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// in-place exclusive scan of counts[0..n); counts[n] and *total receive the sum
__global__ __launch_bounds__(1024) void block_scan_kernel(int *__restrict__ counts, int n,
                                                          int *__restrict__ total) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int e = base + t;
    const int v = e < n ? counts[e] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    const int excl = carry + woff + incl - v;
    if (e < n) counts[e] = excl;
    __syncthreads();
    if (t == 1023) carry = excl + v;
    __syncthreads();
  }
  if (t == 0) {
    counts[n] = carry;
    *total = carry;
  }
}

template <typename DistT>
__global__ __launch_bounds__(kThreads) void ratio_scatter_kernel(const uint64_t *__restrict__ idx,
                                                                 const DistT *__restrict__ dist, int n,
                                                                 double min_ratio,
                                                                 const int *__restrict__ block_offsets,
                                                                 int *__restrict__ matches) {
  __shared__ int wsum[kPerThread][kThreads / 64];
  const int base = blockIdx.x * kBlockQ;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  bool p[kPerThread];
  uint64_t i0[kPerThread];
  unsigned long long bal[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int q = base + k * kThreads + threadIdx.x;
    p[k] = q < n && passes(idx, dist, q, min_ratio, i0[k]);
    bal[k] = __ballot(p[k]);
    if (lane == 0) wsum[k][w] = __popcll(bal[k]);
  }
  __syncthreads();
  int off = block_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    int mine = off;
    for (int ww = 0; ww < w; ++ww) mine += wsum[k][ww];
    if (p[k]) {
      const int pos = mine + __popcll(bal[k] & ((1ull << lane) - 1ull));
      *reinterpret_cast<int2 *>(matches + 2 * (size_t)pos) = make_int2(base + k * kThreads + threadIdx.x, (int)i0[k]);
    }
    off += wsum[k][0] + wsum[k][1] + wsum[k][2] + wsum[k][3];
  }
}

}  // namespace

size_t ratio_workspace_bytes(int yrows) {
  const size_t blocks = ((size_t)std::max(yrows, 1) + kBlockQ - 1) / kBlockQ;
  return round_up((blocks + 1) * sizeof(int), 256);
}

int ratio_run(const uint64_t *d_idx, const void *d_dist, int dist_is_float, int yrows,
              double min_ratio, int *d_matches, int *d_count, void *d_ws, size_t ws_bytes,
              hipStream_t stream) {
  if (yrows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  if (!d_count) return set_error(SPV_ERR_INVALID, "null device pointer");
  if (yrows == 0) {
    SPV_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(int), stream));
    return SPV_OK;
  }
  if (!d_idx || !d_dist || !d_matches) return set_error(SPV_ERR_INVALID, "null device pointer");
  if (!d_ws || ws_bytes < ratio_workspace_bytes(yrows))
    return set_error(SPV_ERR_INVALID, "workspace too small");
  // rows are read as 16-byte (index pair) / 8-byte (distance pair) vectors, matches written as 8-byte pairs
  if ((reinterpret_cast<uintptr_t>(d_idx) & 15) || (reinterpret_cast<uintptr_t>(d_dist) & 7) ||
      (reinterpret_cast<uintptr_t>(d_matches) & 7))
    return set_error(SPV_ERR_INVALID, "device pointers must be aligned to a row (idx 16, dist 8, matches 8 bytes)");
  int *counts = static_cast<int *>(d_ws);
  const int blocks = (yrows + kBlockQ - 1) / kBlockQ;
  ProfScope prof("ratio_test", stream);
  if (dist_is_float) {
    const float *d = static_cast<const float *>(d_dist);
    hipLaunchKernelGGL((ratio_count_kernel<float>), dim3(blocks), dim3(kThreads), 0, stream, d_idx, d,
                       yrows, min_ratio, counts);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, blocks, d_count);
    hipLaunchKernelGGL((ratio_scatter_kernel<float>), dim3(blocks), dim3(kThreads), 0, stream, d_idx,
                       d, yrows, min_ratio, counts, d_matches);
  } else {
    const int32_t *d = static_cast<const int32_t *>(d_dist);
    hipLaunchKernelGGL((ratio_count_kernel<int32_t>), dim3(blocks), dim3(kThreads), 0, stream, d_idx,
                       d, yrows, min_ratio, counts);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, blocks, d_count);
    hipLaunchKernelGGL((ratio_scatter_kernel<int32_t>), dim3(blocks), dim3(kThreads), 0, stream,
                       d_idx, d, yrows, min_ratio, counts, d_matches);
  }
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv
