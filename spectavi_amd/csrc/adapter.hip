// adapter.hip -- data-format adapters either side of the NN path.
//
// SIFT table -> (geometry, descriptors): the reference's SiftFilter emits rows of 132 floats
// = x, y, sigma, angle + 128 descriptor values already quantised to uint8(512*d) but stored
// as float (src/Sift.h:13, 115-123).  The example pipeline runs the matcher on all 132
// columns after normalisation and padding to 144 (example/ex01_essential_estimation.py:92-93);
// splitting the table lets the L1 kernel run on the true 128-D uint8 descriptors (one
// 128-byte line each) with no normalisation pass.
//
// matches -> homogeneous coordinates: gathers (x, y, 1) of both keypoints of every match
// produced by the ratio test (example/ex01_essential_estimation.py:104-106), as the float64
// [n,3] arrays the DLT / RANSAC entry points take.

#include "common.h"

namespace spv {
namespace {

constexpr int kThreads = 256;
constexpr int kSiftCols = 132;

// one wave per 2 table rows: lane l of a half-wave converts descriptor dwords
__global__ __launch_bounds__(kThreads) void sift_split_kernel(const float *__restrict__ table,
                                                              int rows, float *__restrict__ geom,
                                                              uint8_t *__restrict__ desc) {
  // 32 lanes per row: lane j packs descriptor values 4j..4j+3 into one dword
  const long long gt = (long long)blockIdx.x * kThreads + threadIdx.x;
  const long long row = gt >> 5;
  const int j = (int)(gt & 31);
  if (row >= rows) return;
  const float *src = table + (size_t)row * kSiftCols;
  if (j < 4) geom[(size_t)row * 4 + j] = src[j];
  const float *d = src + 4 + 4 * j;
  // descriptor values are integers in [0,255] stored as float: plain truncating cast
  const uint32_t pk = ((uint32_t)(int)d[0] & 0xFFu) | (((uint32_t)(int)d[1] & 0xFFu) << 8) |
                      (((uint32_t)(int)d[2] & 0xFFu) << 16) | (((uint32_t)(int)d[3] & 0xFFu) << 24);
  reinterpret_cast<uint32_t *>(desc)[(size_t)row * 32 + j] = pk;
}

__global__ __launch_bounds__(kThreads) void gather_match_coords_kernel(
    const float *__restrict__ geom_x, const float *__restrict__ geom_y,
    const int *__restrict__ matches, const int *__restrict__ count, double *__restrict__ x0,
    double *__restrict__ x1) {
  const int n = *count;
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int q = matches[2 * (size_t)i], k = matches[2 * (size_t)i + 1];
  x0[3 * (size_t)i + 0] = (double)geom_x[4 * (size_t)k + 0];
  x0[3 * (size_t)i + 1] = (double)geom_x[4 * (size_t)k + 1];
  x0[3 * (size_t)i + 2] = 1.0;
  x1[3 * (size_t)i + 0] = (double)geom_y[4 * (size_t)q + 0];
  x1[3 * (size_t)i + 1] = (double)geom_y[4 * (size_t)q + 1];
  x1[3 * (size_t)i + 2] = 1.0;
}

}  // namespace

int sift_split_run(const float *d_table, int rows, float *d_geom, uint8_t *d_desc, hipStream_t stream) {
  if (rows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  if (rows == 0) return SPV_OK;
  if (!d_table || !d_geom || !d_desc) return set_error(SPV_ERR_INVALID, "null device pointer");
  const long long threads = (long long)rows * 32;
  const long long blocks = (threads + kThreads - 1) / kThreads;
  ProfScope prof("sift_split", stream);
  hipLaunchKernelGGL(sift_split_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, d_table, rows,
                     d_geom, d_desc);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

int gather_match_coords_run(const float *d_geom_x, const float *d_geom_y, const int *d_matches,
                            const int *d_count, int capacity, double *d_x0, double *d_x1,
                            hipStream_t stream) {
  if (capacity < 0) return set_error(SPV_ERR_INVALID, "negative capacity");
  if (capacity == 0) return SPV_OK;
  if (!d_geom_x || !d_geom_y || !d_matches || !d_count || !d_x0 || !d_x1)
    return set_error(SPV_ERR_INVALID, "null device pointer");
  const int blocks = (capacity + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(gather_match_coords_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_geom_x,
                     d_geom_y, d_matches, d_count, d_x0, d_x1);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv

// ---------------------------------------------------------------------------------
// normalize_to_ubyte_and_multiple_16_dim on device, bit for bit what
// numpy computes for a float32 input (reference spectavi/feature.py:384-407):
//   mean_c  = (sequential float32 sum of column c over the rows) / rows      [np.mean, axis 0:
//             numpy accumulates row by row in the input dtype -- verified against numpy 2.2]
//   x0      = x - mean_c;  norm_c = max(max(x0), -min(x0))
//   out     = clip(rint(x0 / norm_c * 128), -128, 127), zero-padded to a multiple of 16 columns
// The column sums are inherently serial in the rows (float addition is not associative), so the
// statistics kernel parallelises over 16-column blocks only -- one workgroup each, four waves
// streaming row tiles into LDS, 16 lanes of a fifth running the chains; everything else is parallel.  Optionally
// also emits the +128 uint8 image the brute-force path takes
// (example/ex01_essential_estimation.py:96-99).
// ---------------------------------------------------------------------------------
namespace spv {
namespace {

constexpr int kStatCols = 16;   // columns per workgroup: one 64-byte sector of every row
constexpr int kStatRows = 512;  // rows per LDS tile (double buffered: 2 x 32 KB)

// One workgroup of five waves per block of 16 columns.  Waves 1..4 (256 lanes) stream the
// block's row tiles into LDS, register-prefetched one tile ahead, and fold max / min of what
// they load (order-independent).  Wave 0 does nothing but the serial per-column chains, lanes
// 0..15, over the tile the loaders finished in the previous round: s += v in row order
// (float32, exactly numpy's order), its LDS reads issued one 16-row batch ahead of the adds so
// the chain runs at the dependent-add rate.  One barrier per tile; the loaders' round (LDS
// writes + issuing the next prefetch) hides behind the chain's.
// stats[0][c] = mean, stats[1][c] = max(max - mean, -(min - mean)).
constexpr int kStatThreads = 320;
__global__ __launch_bounds__(kStatThreads) void column_stats_kernel(const float *__restrict__ x, int rows,
                                                                    int dim, float *__restrict__ stats) {
  __shared__ float tile[2][kStatRows * kStatCols];  // [row][column]
  constexpr int NL = kStatRows * kStatCols / 256;   // 32 floats per loader lane per tile
  const int t = threadIdx.x;
  const bool loader = t >= 64;
  const int lt = t - 64;
  const int c0 = blockIdx.x * kStatCols;
  const int ncol = min(kStatCols, dim - c0);
  const int col = lt & 15, rsub = lt >> 4;  // loader lane -> (column, row within a group of 16 rows)
  const int ntiles = (rows + kStatRows - 1) / kStatRows;
  float pre[NL];
  float s = 0.f, mx = -__builtin_inff(), mn = __builtin_inff();
  const int colc = min(col, ncol - 1);
  auto prefetch = [&](int row0) {  // clamped addresses, raw values: nothing here waits for the loads
#pragma unroll
    for (int i = 0; i < NL; ++i)
      pre[i] = x[(size_t)min(row0 + rsub + 16 * i, rows - 1) * dim + c0 + colc];
  };
  auto publish = [&](int tl) {  // registers -> LDS tile tl, folding max / min on the way
    float *buf = tile[tl & 1];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      // rows past the end and columns past the block become +0 (bitwise, so the load above
      // stays unconditional and the 32 of them stay in flight together)
      const bool live = tl * kStatRows + rsub + 16 * i < rows && col < ncol;
      const float v = __uint_as_float(__float_as_uint(pre[i]) & (live ? 0xFFFFFFFFu : 0u));
      buf[(rsub + 16 * i) * kStatCols + col] = v;
      mx = live ? fmaxf(mx, v) : mx;
      mn = live ? fminf(mn, v) : mn;
    }
  };
  if (loader && ntiles > 0) {
    prefetch(0);
    publish(0);
    if (ntiles > 1) prefetch(kStatRows);
  }
  __syncthreads();
  for (int tl = 0; tl < ntiles; ++tl) {
    if (loader) {
      // tile tl + 1 goes into the buffer the chain finished with before the last barrier
      if (tl + 1 < ntiles) {
        publish(tl + 1);
        if (tl + 2 < ntiles) prefetch((tl + 2) * kStatRows);
      }
    } else if (t < kStatCols) {
      // rows past the end of the input were published as +0, which leaves a float32 sum
      // unchanged, so every tile is walked in full: 16 batches of 32 rows, no conditionals.
      // The scheduling barriers pin "reads of the next half-batch, then adds of this one".
      const float *buf = tile[tl & 1] + t;
      float va[16], vb[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) va[k] = buf[k * kStatCols];
      for (int r = 0; r < kStatRows; r += 32) {
#pragma unroll
        for (int k = 0; k < 16; ++k) vb[k] = buf[(r + 16 + k) * kStatCols];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) s += va[k];  // strictly in row order
        __builtin_amdgcn_sched_barrier(0);
        const int rn = (r + 32) & (kStatRows - 1);  // the last round re-reads rows 0..15, unused
#pragma unroll
        for (int k = 0; k < 16; ++k) va[k] = buf[(rn + k) * kStatCols];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) s += vb[k];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  // combine the 16 row-groups' max / min per column
  float *red = tile[0];
  if (loader) {
    red[lt] = mx;
    red[256 + lt] = mn;
  }
  __syncthreads();
  if (t < ncol) {
    float cmx = red[t], cmn = red[256 + t];
    for (int k = 1; k < 16; ++k) {
      cmx = fmaxf(cmx, red[16 * k + t]);
      cmn = fminf(cmn, red[256 + 16 * k + t]);
    }
    const float mean = s / (float)rows;
    stats[c0 + t] = mean;
    stats[dim + c0 + t] = fmaxf(cmx - mean, -(cmn - mean));
  }
}

__global__ __launch_bounds__(256) void normalize_apply_kernel(const float *__restrict__ x, int rows,
                                                              int dim, int dim16,
                                                              const float *__restrict__ stats,
                                                              float *__restrict__ out_f32,
                                                              unsigned char *__restrict__ out_u8) {
  const size_t total = (size_t)rows * dim16;
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / dim16;
    const int c = (int)(e % dim16);
    float v = 0.f;
    if (c < dim) {
      v = (x[r * dim + c] - stats[c]) / stats[dim + c] * 128.f;
      v = rintf(v);  // numpy round: half to even
      if (v > 127.f) v = 127.f;
      if (v < -128.f) v = -128.f;
    }
    if (out_f32) out_f32[e] = v;
    if (out_u8) out_u8[e] = (unsigned char)((int)(v + 128.f));
  }
}

}  // namespace

size_t normalize_workspace_bytes(int dim) { return round_up((size_t)2 * std::max(dim, 1) * sizeof(float), 256); }

int normalize_run(const float *d_x, int rows, int dim, float *d_out_f32, unsigned char *d_out_u8,
                  void *d_ws, size_t ws_bytes, hipStream_t stream) {
  if (rows < 0 || dim <= 0) return set_error(SPV_ERR_INVALID, "bad shape");
  if (rows == 0) return SPV_OK;
  // numpy sums a single contiguous column pairwise, not row by row: that order is not
  // reproduced here, so the one-column case is refused rather than answered differently
  if (dim == 1 && rows > 1)
    return set_error(SPV_ERR_INVALID, "normalisation of a single-column table is not supported (dim=1)");
  if (!d_x || (!d_out_f32 && !d_out_u8)) return set_error(SPV_ERR_INVALID, "null device pointer");
  if (!d_ws || ws_bytes < normalize_workspace_bytes(dim)) return set_error(SPV_ERR_INVALID, "workspace too small");
  float *stats = static_cast<float *>(d_ws);
  const int dim16 = (dim + 15) / 16 * 16;
  ProfScope prof("normalize", stream);
  hipLaunchKernelGGL(column_stats_kernel, dim3((dim + kStatCols - 1) / kStatCols), dim3(kStatThreads), 0, stream, d_x,
                     rows, dim, stats);
  hipLaunchKernelGGL(normalize_apply_kernel, dim3(2048), dim3(256), 0, stream, d_x, rows, dim, dim16, stats,
                     d_out_f32, d_out_u8);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv
