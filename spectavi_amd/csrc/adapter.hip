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
// statistics kernel parallelises over 16-column blocks only -- one workgroup each, all lanes
// streaming row tiles into LDS, 16 lanes running the chains; everything else is parallel.  Optionally
// also emits the +128 uint8 image the brute-force path takes
// (example/ex01_essential_estimation.py:96-99).
// ---------------------------------------------------------------------------------
namespace spv {
namespace {

constexpr int kStatCols = 16;   // columns per workgroup: one 64-byte sector of every row
constexpr int kStatRows = 512;  // rows per LDS tile (double buffered: 2 x 32 KB)

// One workgroup per block of 16 columns.  All 256 lanes stream the block's row tiles into
// LDS (register-prefetched, one tile ahead); lanes 0..15 of wave 0 then run the serial
// per-column chains over the tile in row order: s += v (float32, exactly numpy's order),
// running max and min.  stats[0][c] = mean, stats[1][c] = max(max - mean, -(min - mean)).
__global__ __launch_bounds__(256) void column_stats_kernel(const float *__restrict__ x, int rows,
                                                           int dim, float *__restrict__ stats) {
  __shared__ float tile[2][kStatRows * kStatCols];  // [row][column]
  constexpr int NL = kStatRows * kStatCols / 256;  // 32 floats per lane per tile
  const int t = threadIdx.x;
  const int c0 = blockIdx.x * kStatCols;
  const int ncol = min(kStatCols, dim - c0);
  const int col = t & 15, rsub = t >> 4;  // lane -> (column, row within a group of 16 rows)
  float pre[NL];
  auto prefetch = [&](int row0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = row0 + rsub + 16 * i;
      pre[i] = (r < rows && col < ncol) ? x[(size_t)r * dim + c0 + col] : 0.f;
    }
  };
  // max / min are order-independent: every lane folds the values it loads itself; only the
  // float32 sum needs the row-ordered serial chain (lanes 0..15)
  float s = 0.f, mx = -__builtin_inff(), mn = __builtin_inff();
  const int ntiles = (rows + kStatRows - 1) / kStatRows;
  auto fold_minmax = [&](int row0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (row0 + rsub + 16 * i < rows) {
        mx = fmaxf(mx, pre[i]);
        mn = fminf(mn, pre[i]);
      }
    }
  };
  if (ntiles > 0) prefetch(0);
  for (int tl = 0; tl < ntiles; ++tl) {
    float *buf = tile[tl & 1];
#pragma unroll
    for (int i = 0; i < NL; ++i) buf[(rsub + 16 * i) * kStatCols + col] = pre[i];
    fold_minmax(tl * kStatRows);
    __syncthreads();  // tile tl is complete; tile tl-1's chain finished before its own barrier
    if (tl + 1 < ntiles) prefetch((tl + 1) * kStatRows);
    if (t < kStatCols) {
      const int nr = min(kStatRows, rows - tl * kStatRows);
      int r = 0;
      for (; r + 16 <= nr; r += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = buf[(r + k) * kStatCols + t];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];  // strictly in row order
      }
      for (; r < nr; ++r) s += buf[r * kStatCols + t];
    }
    // the next iteration writes the OTHER buffer; this buffer is rewritten two iterations
    // later, after the barrier of the next iteration, which the chain lanes reach only
    // once they are done reading it
  }
  // combine the 16 row-groups' max / min per column
  __syncthreads();
  float *red = tile[0];
  red[t] = mx;
  red[256 + t] = mn;
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
  if (!d_x || (!d_out_f32 && !d_out_u8)) return set_error(SPV_ERR_INVALID, "null device pointer");
  if (!d_ws || ws_bytes < normalize_workspace_bytes(dim)) return set_error(SPV_ERR_INVALID, "workspace too small");
  float *stats = static_cast<float *>(d_ws);
  const int dim16 = (dim + 15) / 16 * 16;
  ProfScope prof("normalize", stream);
  hipLaunchKernelGGL(column_stats_kernel, dim3((dim + kStatCols - 1) / kStatCols), dim3(256), 0, stream, d_x,
                     rows, dim, stats);
  hipLaunchKernelGGL(normalize_apply_kernel, dim3(2048), dim3(256), 0, stream, d_x, rows, dim, dim16, stats,
                     d_out_f32, d_out_u8);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv
