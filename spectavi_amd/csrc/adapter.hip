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

#include <cstdlib>

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
                                                                    int dim, float *__restrict__ stats,
                                                                    const int *__restrict__ only_blocks) {
  __shared__ float tile[2][kStatRows * kStatCols];  // [row][column]
  if (only_blocks && !only_blocks[blockIdx.x]) return;  // this block's sums come from the folded chain
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

// ---------------------------------------------------------------------------------
// The same column sums -- the same bits -- without walking every column's 10^6 dependent adds
// (9.25 cycles each, tools/exp/dep_add.hip: 3.85 ms per million rows however many CUs there are).
//
// fl(S + x) rounds the exact sum E to a multiple of Q = 2^(floor(log2|E|) - 23), ties to even.
// While S and E stay inside ONE binade, Q is constant and S = T Q with an integer T, so a step is
//     x = a Q + b  (a = floor(x / Q), 0 <= b < Q),   T' = T + a + c,
//     c = 0 if b < Q/2,  1 if b > Q/2,  and on a tie whatever makes T' even, i.e. parity(T + a):
// the only thing a step needs to know about T is its PARITY.  A run of rows is therefore a function
// parity -> (sum of a + c, parity after), two table entries, and such functions compose
// associatively: rows can be folded in any grouping as long as the order is kept.  (All integer
// arithmetic on the float's mantissa; nothing here is approximate.)
//
//   colsum_stats_kernel      per (1024-row chunk, column): double sum, min / max of its prefix sums,
//                            max, min, all-finite flag                                  [parallel]
//   colsum_plan_kernel       per column, over the chunks: the binade each chunk is EXPECTED to
//                            stay in (from the double prefix), or "no guess"            [1 lane / column]
//   colsum_fold_kernel       per (chunk, column) with a guess: the two-entry function   [parallel]
//   colsum_chain_kernel      per column, over the chunks, carrying the TRUE float S: where S is in
//                            the guessed binade and S + (prefix min / max -+ a margin for the
//                            rounding so far) cannot leave it, apply the function: one table
//                            look-up for 1024 rows, all in integers on S's significand; anywhere
//                            else -- binade crossings, S = 0, cancellation, inf / nan -- add the
//                            chunk's rows one by one (wave-cooperative loads, runs of such chunks
//                            double buffered, the batch handed to the chain through LDS)  [1 wave / column]
// A guess is only ever a guess: it is the chain kernel's check against the true S that licenses
// the shortcut.  Columns whose chunks would mostly fall back (symmetric mixed-sign data, an early
// nan) are left to column_stats_kernel, which walks them at the dependent-add rate.
// ---------------------------------------------------------------------------------
constexpr int kFoldChunk = 1024;   // rows per chunk
constexpr int kFoldTile = 256;     // rows staged per step
constexpr int kFoldCols = 16;      // columns per workgroup: one 64-byte sector of every row
constexpr int kFoldThreads = 256;
constexpr int kFoldStride = kFoldCols + 1;
static_assert(kFoldCols == kStatCols, "normalize_run sizes one column-block grid and `blockserial` for both the statistics and the fold kernels");
constexpr int kNoGuess = -1000;

struct FoldBufs {
  // per (column, chunk) records, [dim][chunks]: a column's chunks are contiguous, so the kernels
  // that walk a column's chunks in order fetch 64 of them with one coalesced load per field
  double *csum, *cpmin, *cpmax;
  float *cvmax, *cvmin;
  int *cfinite;
  int *eguess;
  int *d0, *d1;
  int *tlo, *thi;                // the chunk keeps T + tlo .. T + thi inside the binade or is walked
  int *qq;                       // bit 0 = parity after from 0, bit 1 = from 1; -1 = unusable
  int nchunks;
  float *colvmax, *colvmin;      // [dim]
  int *blockserial;              // [column blocks]: 1 = column_stats_kernel takes this block
};

// Stage rows [row0, row0 + 256) x 16 columns of x into LDS (coalesced 64-byte pieces), +0 outside.
__device__ __forceinline__ void fold_stage(const float *__restrict__ x, int rows, int dim, long long row0, int c0,
                                           float *tile) {
  const int t = threadIdx.x, col = t & 15, rs = t >> 4;
  const bool colok = c0 + col < dim;
#pragma unroll
  for (int i = 0; i < kFoldTile / 16; ++i) {
    const long long r = row0 + rs + 16 * i;
    tile[(rs + 16 * i) * kFoldStride + col] = (colok && r < rows) ? x[(size_t)r * dim + c0 + col] : 0.f;
  }
}

__global__ __launch_bounds__(kFoldThreads) void colsum_stats_kernel(const float *__restrict__ x, int rows, int dim,
                                                                    FoldBufs fb) {
  __shared__ float tile[kFoldTile * kFoldStride];
  __shared__ double r_sum[16][16], r_min[16][16], r_max[16][16];
  __shared__ float r_vmax[16][16], r_vmin[16][16];
  __shared__ int r_fin[16][16];
  const int t = threadIdx.x, col = t & 15, rb = t >> 4;
  // column block fastest: the workgroups that share a row's 128-byte lines run side by side
  const int chunk = blockIdx.y, c0 = blockIdx.x * kFoldCols;
  double run_sum = 0.0, run_min = 0.0, run_max = 0.0;
  float run_vmax = -__builtin_inff(), run_vmin = __builtin_inff();
  int run_fin = 1;
  for (int tl = 0; tl < kFoldChunk / kFoldTile; ++tl) {
    const long long row0 = (long long)chunk * kFoldChunk + (long long)tl * kFoldTile;
    if (row0 >= rows) break;  // uniform
    fold_stage(x, rows, dim, row0, c0, tile);
    __syncthreads();
    double ls = 0.0, lmin = 0.0, lmax = 0.0;
    float vmx = -__builtin_inff(), vmn = __builtin_inff();
    int fin = 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (row0 + rb * 16 + r < rows) {
        const float v = tile[(rb * 16 + r) * kFoldStride + col];
        ls += (double)v;
        lmin = fmin(lmin, ls);
        lmax = fmax(lmax, ls);
        vmx = fmaxf(vmx, v);
        vmn = fminf(vmn, v);
        fin &= (__float_as_uint(v) & 0x7F800000u) != 0x7F800000u;
      }
    }
    r_sum[rb][col] = ls;
    r_min[rb][col] = lmin;
    r_max[rb][col] = lmax;
    r_vmax[rb][col] = vmx;
    r_vmin[rb][col] = vmn;
    r_fin[rb][col] = fin;
    __syncthreads();
    if (t < 16) {
      for (int k = 0; k < 16; ++k) {
        run_min = fmin(run_min, run_sum + r_min[k][t]);
        run_max = fmax(run_max, run_sum + r_max[k][t]);
        run_sum += r_sum[k][t];
        run_vmax = fmaxf(run_vmax, r_vmax[k][t]);
        run_vmin = fminf(run_vmin, r_vmin[k][t]);
        run_fin &= r_fin[k][t];
      }
    }
    __syncthreads();
  }
  if (t < 16 && c0 + t < dim) {
    const size_t o = (size_t)(c0 + t) * fb.nchunks + chunk;
    fb.csum[o] = run_sum;
    fb.cpmin[o] = run_min;
    fb.cpmax[o] = run_max;
    fb.cvmax[o] = run_vmax;
    fb.cvmin[o] = run_vmin;
    fb.cfinite[o] = run_fin;
  }
}

// floor(log2 |v|) of a finite nonzero double
__device__ __forceinline__ int dexp(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7FF) - 1023; }

__device__ __forceinline__ bool same_binade(double lo, double hi, int &e) {
  if (!(lo == lo) || !(hi == hi) || lo == 0.0 || hi == 0.0 || (lo > 0.0) != (hi > 0.0)) return false;
  const int el = dexp(lo), eh = dexp(hi);
  if (el != eh || el < -100 || el > 100) return false;  // (well inside the float32 normal range)
  e = el;
  return true;
}

// One wave per column, 64 chunks per step: the prefix P of the chunk sums is only a guess (any
// summation order will do), so it is a wave scan; so are "has an inf / nan entered the sum" and the
// count of chunks without a guess.
__global__ __launch_bounds__(64) void colsum_plan_kernel(int nchunks, int dim, FoldBufs fb, int *__restrict__ colserial) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double carry = 0.0;
  int poisoned = 0, fallbacks = 0;
  float vmx = -__builtin_inff(), vmn = __builtin_inff();
  for (int j0 = 0; j0 < nchunks; j0 += 64) {
    const int j = j0 + lane;
    const bool live = j < nchunks;
    const size_t o = (size_t)c * nchunks + (live ? j : nchunks - 1);
    const double s = live ? fb.csum[o] : 0.0;
    const int bad = live ? !fb.cfinite[o] : 0;
    // inclusive scans over the 64 chunks of this step
    double incl = s;
    int pbad = bad;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double up = __shfl_up(incl, d, 64);
      const int ub = __shfl_up(pbad, d, 64);
      if (lane >= d) {
        incl += up;
        pbad |= ub;
      }
    }
    const double P = carry + incl - s;  // exclusive
    int e = kNoGuess;
    if (live && !(poisoned | pbad)) {
      int ee;
      if (same_binade(P + fb.cpmin[o], P + fb.cpmax[o], ee)) e = ee;
    }
    if (live) {
      fb.eguess[o] = e;
      vmx = fmaxf(vmx, fb.cvmax[o]);
      vmn = fminf(vmn, fb.cvmin[o]);
    }
    fallbacks += __popcll(__ballot(live && e == kNoGuess));
    carry += __shfl(incl, 63, 64);
    poisoned |= __shfl(pbad, 63, 64);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    vmx = fmaxf(vmx, __shfl_down(vmx, d, 64));
    vmn = fminf(vmn, __shfl_down(vmn, d, 64));
  }
  if (lane == 0) {
    fb.colvmax[c] = vmx;
    fb.colvmin[c] = vmn;
    colserial[c] = fallbacks * 10 > nchunks * 6;
  }
}

// a 16-column block goes to column_stats_kernel when any of its columns would mostly be walked
__global__ void colsum_blockmode_kernel(int dim, const int *__restrict__ colserial, int *__restrict__ blockserial) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b * kFoldCols >= dim) return;
  int any = 0;
  for (int c = b * kFoldCols; c < min(dim, (b + 1) * kFoldCols); ++c) any |= colserial[c];
  blockserial[b] = any;
}

// one row relative to the quantum 2^kq: a = floor(x / Q) and how b = x - a Q compares with Q / 2
// (-1 below or b = 0, 0 tie, +1 above); false if a does not fit (the row is far too big for this
// binade: the chunk cannot stay in it, and the chain kernel's bounds will say so as well)
__device__ __forceinline__ bool fold_elem(uint32_t bits, int kq, int &a, int &bc) {
  const uint32_t ef = (bits >> 23) & 0xFFu, man = bits & 0x7FFFFFu;
  const bool neg = (bits >> 31) != 0;
  a = 0;
  bc = -1;
  if (ef == 0 && man == 0) return true;
  const uint32_t m = man | (ef ? 0x800000u : 0u);
  const int exu = (int)(ef ? ef : 1u) - 127;
  const int sh = kq - (exu - 23);
  if (sh <= 0) {
    if (-sh > 6) return false;
    const int av = (int)(m << (-sh));
    a = neg ? -av : av;
    return true;
  }
  if (sh >= 26) {  // |x| < Q / 4
    if (neg) {
      a = -1;
      bc = 1;
    }
    return true;
  }
  const uint32_t apos = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
  uint32_t b;
  if (!neg) {
    a = (int)apos;
    b = rem;
  } else if (rem == 0) {
    a = -(int)apos;
    b = 0;
  } else {
    a = -(int)apos - 1;
    b = (1u << sh) - rem;
  }
  bc = b == 0 ? -1 : (b < half ? -1 : (b == half ? 0 : 1));
  return true;
}

__global__ __launch_bounds__(kFoldThreads) void colsum_fold_kernel(const float *__restrict__ x, int rows, int dim,
                                                                   FoldBufs fb) {
  __shared__ float tile[kFoldTile * kFoldStride];
  __shared__ int r_d0[16][16], r_d1[16][16], r_q[16][16];
  __shared__ int s_e[16];
  __shared__ int s_any;
  const int t = threadIdx.x, col = t & 15, rb = t >> 4;
  const int chunk = blockIdx.y, c0 = blockIdx.x * kFoldCols;
  if (fb.blockserial[blockIdx.x]) return;  // uniform
  if (t == 0) s_any = 0;
  __syncthreads();
  if (t < 16) {
    const int e = c0 + t < dim ? fb.eguess[(size_t)(c0 + t) * fb.nchunks + chunk] : kNoGuess;
    s_e[t] = e;
    if (e != kNoGuess) atomicOr(&s_any, 1);
  }
  __syncthreads();
  if (!s_any) return;  // uniform: nothing to fold in this (chunk, column block)
  const int e = s_e[col];
  const int kq = e - 23;
  long long run_d0 = 0, run_d1 = 0;
  int run_q0 = 0, run_q1 = 1, run_ok = 1;
  for (int tl = 0; tl < kFoldChunk / kFoldTile; ++tl) {
    const long long row0 = (long long)chunk * kFoldChunk + (long long)tl * kFoldTile;
    if (row0 >= rows) break;  // uniform
    fold_stage(x, rows, dim, row0, c0, tile);
    __syncthreads();
    int d0 = 0, d1 = 0, q0 = 0, q1 = 1, ok = 1;
    if (e != kNoGuess) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (row0 + rb * 16 + r < rows) {
          int a, bc;
          ok &= fold_elem(__float_as_uint(tile[(rb * 16 + r) * kFoldStride + col]), kq, a, bc) ? 1 : 0;
          const int par0 = q0 ^ (a & 1), par1 = q1 ^ (a & 1);
          const int c0_ = bc < 0 ? 0 : (bc > 0 ? 1 : par0), c1_ = bc < 0 ? 0 : (bc > 0 ? 1 : par1);
          d0 += a + c0_;
          d1 += a + c1_;
          q0 = par0 ^ c0_;
          q1 = par1 ^ c1_;
        }
      }
    }
    r_d0[rb][col] = d0;
    r_d1[rb][col] = d1;
    r_q[rb][col] = q0 | (q1 << 1) | (ok ? 0 : 4);
    __syncthreads();
    if (t < 16) {
      for (int k = 0; k < 16; ++k) {
        const int q = r_q[k][t];
        const long long g0 = r_d0[k][t], g1 = r_d1[k][t];
        const int gq0 = q & 1, gq1 = (q >> 1) & 1;
        run_ok &= (q & 4) ? 0 : 1;
        // (running) then (this run of 16 rows)
        run_d0 += run_q0 ? g1 : g0;
        run_d1 += run_q1 ? g1 : g0;
        run_q0 = run_q0 ? gq1 : gq0;
        run_q1 = run_q1 ? gq1 : gq0;
      }
    }
    __syncthreads();
  }
  if (t < 16 && c0 + t < dim && s_e[t] != kNoGuess) {
    const size_t o = (size_t)(c0 + t) * fb.nchunks + chunk;
    bool fits = run_ok && run_d0 > -(1ll << 26) && run_d0 < (1ll << 26) && run_d1 > -(1ll << 26) && run_d1 < (1ll << 26);
    // how far, in units of Q, the running sum can move away from its value at the chunk's start:
    // the exact prefix range, widened by the rounding the chain can accumulate (<= Q / 2 per row)
    const double Q = __builtin_ldexp(1.0, s_e[t] - 23);
    const double lo = __builtin_floor((fb.cpmin[o]) / Q) - (double)kFoldChunk;
    const double hi = __builtin_ceil((fb.cpmax[o]) / Q) + (double)kFoldChunk;
    fits = fits && lo > -33554432.0 && hi < 33554432.0;  // 2^25: nothing that could stay in the binade is lost
    fb.d0[o] = (int)run_d0;
    fb.d1[o] = (int)run_d1;
    fb.tlo[o] = fits ? (int)lo : 0;
    fb.thi[o] = fits ? (int)hi : 0;
    fb.qq[o] = fits ? (run_q0 | (run_q1 << 1)) : -1;
  }
}

// One wave per column.  Every lane carries the same S.  The records of 64 chunks are fetched at
// once (one per lane, coalesced) and handed to the sequential part with v_readlane, so the chain
// never waits for memory between two chunks it can fold; lanes only differ when a chunk is walked
// (each loads its own rows, see walk_rows).
__device__ __forceinline__ int rl_i(int v, int k) { return __builtin_amdgcn_readlane(v, k); }

// Rows [r0, r0 + n) of column c added to S one by one, in order: every lane loads its own row of a
// 64-row batch and the batch is handed to the chain through LDS.  Groups of 1024 rows are double buffered, so
// only the first group of a run of walked chunks waits for memory.
__device__ __forceinline__ void walk_rows(const float *__restrict__ x, int dim, int c, long long r0, long long n,
                                          int lane, float *wbuf, float &S) {
  constexpr int NB = kFoldChunk / 64;
  float va[NB], vb[NB];
  const long long end = r0 + n;
  auto load = [&](float (&v)[NB], long long base) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const long long r = base + 64 * b + lane;
      v[b] = r < end ? x[(size_t)r * dim + c] : 0.f;
    }
  };
  // a batch goes through a 256-byte LDS slot (two slots, alternating): every lane reads the 64 values
  // back with 16-byte broadcast reads, so the chain's operands are plain VGPRs.  (v_readlane feeding
  // the adds through one SGPR costs ~17 cycles per row instead of the 9.25 of the dependent add: each
  // readlane has to wait for the add before it to release the register, and the add after it for
  // the readlane's hazard slots.)
  auto chain = [&](const float (&v)[NB], long long base) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const long long left = end - (base + 64 * b);
      float *slot = wbuf + 64 * (b & 1);
      slot[lane] = v[b];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (left >= 64) {
        float4 r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = reinterpret_cast<const float4 *>(slot)[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          S += r[k].x;
          S += r[k].y;
          S += r[k].z;
          S += r[k].w;
        }
      } else {
        for (int k = 0; k < left; ++k) S += slot[k];
      }
    }
  };
  load(va, r0);
  for (long long g = r0; g < end; g += 2 * kFoldChunk) {
    const bool more1 = g + kFoldChunk < end;
    if (more1) load(vb, g + kFoldChunk);
    chain(va, g);
    if (more1) {
      if (g + 2 * kFoldChunk < end) load(va, g + 2 * kFoldChunk);
      chain(vb, g + kFoldChunk);
    }
  }
}

__global__ __launch_bounds__(64) void colsum_chain_kernel(const float *__restrict__ x, int rows, int dim, int nchunks,
                                                          FoldBufs fb, float *__restrict__ stats) {
  __shared__ __attribute__((aligned(16))) float wbuf[128];
  const int c = blockIdx.x, lane = threadIdx.x;
  if (fb.blockserial[c / kFoldCols]) return;
  float S = 0.f;
  for (int j0 = 0; j0 < nchunks; j0 += 64) {
    const int jl = min(j0 + lane, nchunks - 1);
    const size_t o = (size_t)c * nchunks + jl;
    const int ev = fb.eguess[o], qv = fb.qq[o], d0v = fb.d0[o], d1v = fb.d1[o];
    const int lov = fb.tlo[o], hiv = fb.thi[o];
    const int nk = min(64, nchunks - j0);
    // chunks without a guess are walked for certain: consecutive ones as ONE run
    const unsigned long long noguess = __ballot(lane < nk && ev == kNoGuess);
    for (int k = 0; k < nk;) {
      const long long r0 = (long long)(j0 + k) * kFoldChunk;
      if ((noguess >> k) & 1ull) {
        const unsigned long long rest = ~(noguess >> k);
        const int run = min(rest ? __builtin_ctzll(rest) : 64, nk - k);
        walk_rows(x, dim, c, r0, min((long long)run * kFoldChunk, (long long)rows - r0), lane, wbuf, S);
        k += run;
        continue;
      }
      const int e = rl_i(ev, k);
      bool fast = false;
      const int q = rl_i(qv, k);
      // S is the same in every lane: telling the compiler so keeps this whole test on the scalar unit
      const uint32_t sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(S));
      const int es = (int)((sb >> 23) & 0xFFu) - 127;  // S = 0 / subnormal / inf / nan never match a guess
      if (q >= 0 && es == e) {
        // S = T Q with Q = 2^(e - 23): T is S's 24-bit significand with its sign
        const int mag = (int)((sb & 0x7FFFFFu) | 0x800000u);
        const bool neg = (sb >> 31) != 0;
        int T = neg ? -mag : mag;
        // the whole chunk inside the binade: 2^23 <= |T + t| < 2^24 for every t in [tlo, thi]
        const int lo = T + rl_i(lov, k), hi = T + rl_i(hiv, k);
        const bool inside = neg ? (lo > -(1 << 24) && hi <= -(1 << 23)) : (lo >= (1 << 23) && hi < (1 << 24));
        if (inside) {
          T += (T & 1) ? rl_i(d1v, k) : rl_i(d0v, k);
          const uint32_t am = (uint32_t)(T < 0 ? -T : T);  // in [2^23, 2^24) by the bounds
          S = __uint_as_float((sb & 0x80000000u) | ((uint32_t)(e + 127) << 23) | (am & 0x7FFFFFu));
          fast = true;
        }
      }
      if (!fast) walk_rows(x, dim, c, r0, min((long long)kFoldChunk, (long long)rows - r0), lane, wbuf, S);
      ++k;
    }
  }
  if (lane == 0) {
    const float mean = S / (float)rows;
    stats[c] = mean;
    stats[dim + c] = fmaxf(fb.colvmax[c] - mean, -(fb.colvmin[c] - mean));
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

static int fold_chunks(int rows) { return (rows + kFoldChunk - 1) / kFoldChunk; }

// workspace that lets normalize_run fold the column sums (tables of at least kFoldMinRows rows)
constexpr int kFoldMinRows = 64 * kFoldChunk;  // below this the launches and the always-walked first chunks (the sum
                                               // doubles through a binade per chunk at first) cost more than walking
size_t normalize_workspace_bytes_rows(int rows, int dim) {
  size_t b = normalize_workspace_bytes(dim);
  if (rows < kFoldMinRows) return b;
  const size_t n = (size_t)fold_chunks(rows) * std::max(dim, 1);
  b += 3 * round_up(n * sizeof(double), 256) + 9 * round_up(n * sizeof(int), 256);
  b += 3 * round_up((size_t)dim * sizeof(float), 256) + round_up((size_t)((dim + kFoldCols - 1) / kFoldCols) * sizeof(int), 256);
  return b;
}

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
  const int nblk = (dim + kStatCols - 1) / kStatCols;
  static const bool serial_only = [] {
    const char *e = getenv("SPECTAVI_NORMALIZE_SERIAL");
    return e && *e == '1';
  }();
  const int *only_blocks = nullptr;
  if (!serial_only && rows >= kFoldMinRows && ws_bytes >= normalize_workspace_bytes_rows(rows, dim)) {
    const int nch = fold_chunks(rows);
    const size_t n = (size_t)nch * dim;
    unsigned char *p = static_cast<unsigned char *>(d_ws) + normalize_workspace_bytes(dim);
    auto take = [&](size_t bytes) {
      unsigned char *q = p;
      p += round_up(bytes, 256);
      return q;
    };
    FoldBufs fb;
    fb.csum = reinterpret_cast<double *>(take(n * sizeof(double)));
    fb.cpmin = reinterpret_cast<double *>(take(n * sizeof(double)));
    fb.cpmax = reinterpret_cast<double *>(take(n * sizeof(double)));
    fb.cvmax = reinterpret_cast<float *>(take(n * sizeof(float)));
    fb.cvmin = reinterpret_cast<float *>(take(n * sizeof(float)));
    fb.cfinite = reinterpret_cast<int *>(take(n * sizeof(int)));
    fb.eguess = reinterpret_cast<int *>(take(n * sizeof(int)));
    fb.d0 = reinterpret_cast<int *>(take(n * sizeof(int)));
    fb.d1 = reinterpret_cast<int *>(take(n * sizeof(int)));
    fb.tlo = reinterpret_cast<int *>(take(n * sizeof(int)));
    fb.thi = reinterpret_cast<int *>(take(n * sizeof(int)));
    fb.qq = reinterpret_cast<int *>(take(n * sizeof(int)));
    fb.colvmax = reinterpret_cast<float *>(take((size_t)dim * sizeof(float)));
    fb.colvmin = reinterpret_cast<float *>(take((size_t)dim * sizeof(float)));
    fb.blockserial = reinterpret_cast<int *>(take((size_t)nblk * sizeof(int)));
    int *colserial = reinterpret_cast<int *>(take((size_t)dim * sizeof(int)));
    fb.nchunks = nch;
    const dim3 grid((unsigned)nblk, (unsigned)nch);
    hipLaunchKernelGGL(colsum_stats_kernel, grid, dim3(kFoldThreads), 0, stream, d_x, rows, dim, fb);
    hipLaunchKernelGGL(colsum_plan_kernel, dim3(dim), dim3(64), 0, stream, nch, dim, fb, colserial);
    hipLaunchKernelGGL(colsum_blockmode_kernel, dim3((nblk + 63) / 64), dim3(64), 0, stream, dim, colserial, fb.blockserial);
    hipLaunchKernelGGL(colsum_fold_kernel, grid, dim3(kFoldThreads), 0, stream, d_x, rows, dim, fb);
    hipLaunchKernelGGL(colsum_chain_kernel, dim3(dim), dim3(64), 0, stream, d_x, rows, dim, nch, fb, stats);
    only_blocks = fb.blockserial;
  }
  hipLaunchKernelGGL(column_stats_kernel, dim3(nblk), dim3(kStatThreads), 0, stream, d_x, rows, dim, stats,
                     only_blocks);
  hipLaunchKernelGGL(normalize_apply_kernel, dim3(2048), dim3(256), 0, stream, d_x, rows, dim, dim16, stats,
                     d_out_f32, d_out_u8);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv
