// cascade.hip -- cascade-hash candidate prefilter + exact L1 refine for gfx950 (MI355X).
//
// Replaces reference CascadingHashNn (src/CascadingHashNn.h:86-245) behind
// nn_cascading_hash (src/Spectavi.cpp:321-336).  The reference builds
// unordered_map<code, list<idx>> buckets on the host, and per query unions the
// buckets of 2^g multi-probe codes per table into an unordered_set that feeds
// the L1 brute force as a candidate filter.  Here everything is flat arrays in
// HBM and these kernels:
//
//   1 repack_dict     dict[n][dim][m] -> dictp[n][dim][MC], MC = roundup(m,4), zero padded
//   2 project<false>  database rows: n*m random-hyperplane projections (on the matrix cores
//                     when n*m <= 64, project_mfma_kernel; else the VALU project_kernel) as a
//                     dim-ordered fp32 FMA chain (the oracle runs the identical
//                     chain, so sign bits match bit for bit), sign-packed codes
//                     (bit b set iff proj >= 0, src/CascadingHashNn.h:102-111), the
//                     uint8 refine image  u8 = (int(trunc v) + 128) & 255
//                     (src/CascadingHashNn.h:236-239) and the bucket histogram
//   3 project<true>   query rows: sign code + mask of the g least-confident bits
//                     (smallest (|proj|, bit) pairs, src/CascadingHashNn.h:150-160)
//                     + uint8 image
//   4 bucket_segsum / segscan / scan, 5 bucket_fill   counting sort of database
//                     indices by bucket = code & (2^HB - 1), per table
//   6 probe_table     one launch per table, one 8-lane group per query (8 queries per wave), the
//                     queries walked in the order of that table's sign code (a second counting
//                     sort, large inputs only) so that a bucket's rows stay in the XCD's L2: 2^g
//                     probe codes (src/CascadingHashNn.h:170-179) -> bucket ranges -> the group's
//                     candidate list in LDS -> per round each group gathers one 128-byte
//                     candidate row (8 different queries' lines per wave instruction), v_sad_u8 +
//                     DPP reduce, branch-free two smallest (dist, idx) keys carried from table
//                     to table
//   7 probe_refine    one wave per query (the first design, issue-bound): used when
//                     kernel 6 does not apply -- full-code check (m > 22) or rows wider
//                     than 256 bytes (up to 2048)
//
// Candidate semantics (closed form of filter_potential_neighbours, :208-227):
// database row k is a candidate of query i iff for some table j
//   ((code_j(x_k) ^ sign_j(y_i)) & ~mask_j(y_i)) == 0.
// The reference visits the candidate set in unordered_set order, so its
// tie-break on equal distance is unspecified; here the result is the two
// smallest (dist, idx) pairs of the candidate set, lexicographic.

#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace spv {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBucketBits = 22;
constexpr int kListCap = 512;  // candidate indices staged in LDS per wave
constexpr uint64_t kNone64 = ~0ull;

inline int bucket_bits(int m) { return std::min(m, kMaxBucketBits); }

// ---------------------------------------------------------------------------------
// 1. dictionary repack
// ---------------------------------------------------------------------------------
__global__ void repack_dict_kernel(const float *__restrict__ dict, float *__restrict__ dictp, int n,
                                   int dim, int m, int mc) {
  const int total = n * dim * mc;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int b = e % mc;
    const int ji = e / mc;  // j*dim + i
    dictp[e] = b < m ? dict[(size_t)ji * m + b] : 0.f;
  }
}

// ---------------------------------------------------------------------------------
// 2/3. projections, codes, uint8 image.  A workgroup owns 256 rows, one per lane
// (kProjRows).  Per 16-dim step the row tile is staged in LDS by coalesced 16-byte loads
// (the uint8 image is written from the same registers), the hyperplanes of that step
// are staged next to it and read by broadcast ds_read_b128 -- one LDS dword feeds
// 64 lanes.  Accumulators of up to two tables stay in registers, so float32 rows are
// read from HBM once.  The projection of (row, bit) is ONE
// dim-ordered fp32 FMA chain (mirrored by the oracle).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u8(float v) {
  return (uint32_t)((int)v + 128) & 0xFFu;  // trunc toward zero, wrap mod 256
}

constexpr int kProjRows = 1;                 // rows per lane (2 halves the broadcast reads per FMA but its
                                             // 77 KB of LDS allow only two workgroups per CU: measured slower)
constexpr int kProjTile = kThreads * kProjRows;  // rows per workgroup
constexpr int kProjChunk = 16;               // dims staged per step (dim % 16 == 0 always)
constexpr int kProjXStride = kProjChunk * 4 + 16;  // bytes per row in LDS: 80 -> conflict-free b128 reads

// NT tables are accumulated per pass over the rows (NT*R*MC accumulators per lane); NT = 2 (the
// default for n >= 2, see launch_project) reads the float32 rows from HBM exactly once.
template <int MC, int NT, bool IS_QUERY, int GMAX>
__global__ __launch_bounds__(kThreads) void project_kernel(
    const float *__restrict__ rows, int nrows, int dim, int m, int n, int g,
    const float *__restrict__ dictp,     // [n][dim][MC]
    uint32_t *__restrict__ codes,        // [n][nrows] sign codes
    uint32_t *__restrict__ masks,        // [n][nrows] (queries only)
    uint8_t *__restrict__ u8img,         // [nrows][dim]
    uint32_t *__restrict__ counts,       // [n][nb+1] bucket histogram (may be NULL: none wanted)
    uint32_t *__restrict__ ranks,        // [n][nrows] rank of the row inside its bucket (with counts)
    uint32_t hbmask, int nb) {
  constexpr int R = kProjRows;
  __shared__ __attribute__((aligned(16))) uint8_t xs[kProjTile * kProjXStride];  // row tile, 16 dims
  __shared__ __attribute__((aligned(16))) uint8_t u8s[kProjTile * 64];  // uint8 image, four steps
  __shared__ float4 sd[NT][kProjChunk * MC / 4];  // hyperplanes [table][dim of the chunk][MC]
  // accumulator start values: +0 for real hyperplanes, +inf for the zero-padded ones so
  // that they are never picked as "least confident" (their code bits are masked off)
  __shared__ float sinit[MC];
  const int t = threadIdx.x;
  const int base = blockIdx.x * kProjTile;
  if (t < MC) sinit[t] = t < m ? 0.f : __builtin_inff();
  __syncthreads();

  for (int jt = 0; jt < n; jt += NT) {
    float acc[NT][R][MC];
#pragma unroll
    for (int b = 0; b < MC; ++b) {
      const float a0 = sinit[b];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int k = 0; k < R; ++k) acc[tj][k][b] = a0;
    }
    // register-prefetched staging: the global loads of step c0+16 are in flight while
    // step c0 is being computed out of LDS
    constexpr int NX = kProjTile * (kProjChunk / 4) / kThreads;           // row-tile vectors per lane
    constexpr int ND = (NT * kProjChunk * MC / 4 + kThreads - 1) / kThreads;  // hyperplane vectors per lane
    float4 px[NX], pd[ND];
    auto prefetch = [&](int c0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int e = t + i * kThreads;
        const int grow = base + (e >> 2);
        px[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grow < nrows)
          px[i] = *reinterpret_cast<const float4 *>(rows + (size_t)grow * dim + c0 + 4 * (e & 3));
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int e = t + i * kThreads;                 // [table of the pass][dim][MC/4]
        const int tj = e / (kProjChunk * MC / 4), w = e % (kProjChunk * MC / 4);
        pd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tj < NT && jt + tj < n)
          pd[i] = reinterpret_cast<const float4 *>(dictp + ((size_t)(jt + tj) * dim + c0) * MC)[w];
      }
    };
    prefetch(0);
    for (int c0 = 0; c0 < dim; c0 += kProjChunk) {
      __syncthreads();  // everyone is done reading the previous step out of LDS
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int e = t + i * kThreads;
        const int row = e >> 2, c = e & 3;
        *reinterpret_cast<float4 *>(xs + row * kProjXStride + 16 * c) = px[i];
        if (jt == 0) {
          // uint8 image: four 16-dim steps are collected per row in LDS (64 bytes per row,
          // unpadded: 16*e addressing) and flushed below as 64-byte runs = full sectors
          const float4 v = px[i];
          const uint32_t pk = f2u8(v.x) | (f2u8(v.y) << 8) | (f2u8(v.z) << 16) | (f2u8(v.w) << 24);
          *reinterpret_cast<uint32_t *>(u8s + row * 64 + ((c0 >> 4) & 3) * 16 + 4 * c) = pk;
        }
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int e = t + i * kThreads;
        if (e < NT * kProjChunk * MC / 4) (&sd[0][0])[e] = pd[i];
      }
      __syncthreads();
      if (jt == 0 && ((((c0 >> 4) & 3) == 3) || c0 + kProjChunk >= dim)) {
        const int ngrp = ((c0 >> 4) & 3) + 1;
        const int col0 = c0 - (ngrp - 1) * kProjChunk;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const int e = t + i * kThreads;
          const int row = e >> 2, q = e & 3;
          if (q < ngrp && base + row < nrows)
            *reinterpret_cast<uint4 *>(u8img + (size_t)(base + row) * dim + col0 + 16 * q) =
                *reinterpret_cast<const uint4 *>(u8s + 16 * e);
        }
      }
      if (c0 + kProjChunk < dim) prefetch(c0 + kProjChunk);
#pragma unroll
      for (int i0 = 0; i0 < kProjChunk; i0 += 4) {
        float4 xv[R];
#pragma unroll
        for (int k = 0; k < R; ++k)
          xv[k] = *reinterpret_cast<const float4 *>(xs + (t + k * kThreads) * kProjXStride + 4 * i0);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
#pragma unroll
          for (int tj = 0; tj < NT; ++tj) {
            // all lanes read the same hyperplane row: LDS broadcast, MC/4 x ds_read_b128
            float dv[MC];
#pragma unroll
            for (int b4 = 0; b4 < MC / 4; ++b4) {
              const float4 d4 = sd[tj][(i0 + ii) * (MC / 4) + b4];
              dv[4 * b4 + 0] = d4.x;
              dv[4 * b4 + 1] = d4.y;
              dv[4 * b4 + 2] = d4.z;
              dv[4 * b4 + 3] = d4.w;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
              const float xsv =
                  ii == 0 ? xv[k].x : (ii == 1 ? xv[k].y : (ii == 2 ? xv[k].z : xv[k].w));
#pragma unroll
              for (int b = 0; b < MC; ++b) acc[tj][k][b] = __builtin_fmaf(xsv, dv[b], acc[tj][k][b]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      const int j = jt + tj;
      if (j >= n) break;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int r = base + t + k * kThreads;
        uint32_t code = 0;
#pragma unroll
        for (int b = 0; b < MC; ++b) code |= (acc[tj][k][b] >= 0.f ? 1u : 0u) << b;
        code &= 0xFFFFFFFFu >> (32 - m);  // padded hyperplanes: drop their bits
        if (r < nrows) {
          codes[(size_t)j * nrows + r] = code;
          // database rows: bucket histogram for the counting sort, fused here
          // (the value the atomic returns is the row's rank inside its bucket: the fill pass
          // needs no second round of atomics; the matrix-core kernels' epilogue does the same for
          // query rows too, this kernel leaves those to query_rank_kernel: the atomic in this
          // epilogue's query form cost its widest instantiation its full unrolling)
          if (!IS_QUERY && counts)
            ranks[(size_t)j * nrows + r] = atomicAdd(&counts[(size_t)j * (nb + 1) + (code & hbmask)], 1u);
        }
        if (IS_QUERY) {
          // g smallest (|proj|, bit) pairs, lexicographic like the reference's max-heap of
          // pairs (src/CascadingHashNn.h:153-159): sorted insertion with the bit as tie-break,
          // which matters for an entry displaced down the list past an equal magnitude
          float best[GMAX];
          int bbit[GMAX];
#pragma unroll
          for (int q = 0; q < GMAX; ++q) {
            best[q] = __builtin_inff();
            bbit[q] = -1;
          }
#pragma unroll
          for (int b = 0; b < MC; ++b) {
            float v = fabsf(acc[tj][k][b]);  // +inf for padded hyperplanes: never inserted
            int vb = b;
#pragma unroll
            for (int q = 0; q < GMAX; ++q) {
              const bool lt = q < g && (v < best[q] || (v == best[q] && vb < bbit[q]));
              const float tv = best[q];
              const int tb = bbit[q];
              best[q] = lt ? v : tv;
              bbit[q] = lt ? vb : tb;
              v = lt ? tv : v;
              vb = lt ? tb : vb;
            }
          }
          uint32_t mask = 0;
#pragma unroll
          for (int q = 0; q < GMAX; ++q)
            if (q < g && bbit[q] >= 0) mask |= 1u << bbit[q];
          if (r < nrows) masks[(size_t)j * nrows + r] = mask;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// 2b/3b. the same projections on the matrix cores (used when all n*m hyperplanes fit 64
// columns, i.e. every default configuration).  The projection is the one GEMM-shaped step of
// the path: P[rows][n*m] = X[rows][dim] * H[dim][n*m].  v_mfma_f32_16x16x4_f32 accumulates each
// output element as a sequential fp32 FMA chain in k order -- checked against std::fmaf on
// 64 000 outputs, tools/exp/mfma_order.hip: 100 % identical -- so the result is bit for bit the
// chain the oracle and the VALU kernel above compute, at four times the FMA rate and without
// the LDS broadcast traffic that bounds the VALU kernel.
//
// One wave owns 64 rows (waves are independent: no workgroup barriers).  Per 32-dim step
// the wave's float4 loads (eight lanes cover one full 128-byte line of a row; prefetched one
// step ahead; the uint8 image is packed from the same registers) are staged in a
// wave-private LDS tile, from which lane
// (r = lane % 16, q = lane / 16) reads A[row r of row-tile rt][k + q]; the hyperplane operand
// B[k + q][column] comes straight from the repacked dictionary dictm[dim][16*CT] (L1/L2
// resident).  4 x CT accumulator tiles of 16 x 16.  Afterwards the tiles are transposed
// through LDS so that one lane holds one row's n*m projections and runs the same epilogue as
// the VALU kernel: sign codes, the g least-confident bits, bucket histogram.
// ---------------------------------------------------------------------------------
constexpr int kMfmaChunkC = 32;  // = kMfmaChunk (declared below): dims per step of the MFMA kernels
__global__ void repack_dict_mfma_kernel(const float *__restrict__ dict, float *__restrict__ dictm, int n,
                                        int dim, int m, int nc) {
  const int dimpad = (dim + kMfmaChunkC - 1) / kMfmaChunkC * kMfmaChunkC;  // whole chunks: zero rows past dim
  const int total = dimpad * nc;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int c = e % nc, i = e / nc;  // column c = table * m + bit
    const int tj = c / m, b = c % m;
    dictm[e] = (tj < n && i < dim) ? dict[((size_t)tj * dim + i) * m + b] : 0.f;
  }
}

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaChunk = kMfmaChunkC;  // dims per step: 8 lanes x 16 B = one full 128-byte line per row

// Row part of the matrix-core projection kernels' epilogue: lane l holds the n*m projections of row
// row0 + l at `mine` (LDS, column c = table*m + bit; column `pad` is a harmless filler for the slots
// past m): sign codes, the g least-confident bits and the bucket histogram as in the VALU kernel.
// Rows at and past row_limit are not written.
template <bool IS_QUERY, int GMAX>
__device__ __forceinline__ void project_rows_epilogue(
    const float *mine, int pad, int lane, long long row0, long long row_limit, long long nrows_total,
    int m, int n, int g, uint32_t *__restrict__ codes, uint32_t *__restrict__ masks,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ ranks, uint32_t hbmask, int nb) {
  const long long r = row0 + lane;
  const long long nrows = row_limit;  // rows at and past it belong to nobody here
  for (int j = 0; j < n; ++j) {
    uint32_t code = 0;
    float best[GMAX];
    int bbit[GMAX];
#pragma unroll
    for (int q = 0; q < GMAX; ++q) {
      best[q] = __builtin_inff();
      bbit[q] = -1;
    }
    // eight projections per round: the LDS reads of a round are independent, so their latency
    // is paid once per round instead of once per bit
    for (int b0 = 0; b0 < m; b0 += 8) {
      float p8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p8[i] = mine[min(j * m + b0 + i, pad)];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = b0 + i;
        const bool live = b < m;
        const float p = p8[i];
        code |= (live && p >= 0.f ? 1u : 0u) << (b & 31);
        if (IS_QUERY) {
          // g smallest (|proj|, bit) pairs, lexicographic (see the VALU kernel's epilogue);
          // slots past m enter as +inf and are never inserted
          float v = live ? fabsf(p) : __builtin_inff();
          int vb = b;
#pragma unroll
          for (int q = 0; q < GMAX; ++q) {
            const bool lt = q < g && (v < best[q] || (v == best[q] && vb < bbit[q]));
            const float tv = best[q];
            const int tb = bbit[q];
            best[q] = lt ? v : tv;
            bbit[q] = lt ? vb : tb;
            v = lt ? tv : v;
            vb = lt ? tb : vb;
          }
        }
      }
    }
    if (r < nrows) {
      codes[(size_t)j * nrows_total + r] = code;
      // bucket histogram, fused (database rows: the buckets; query rows: the order the probe walks
      // them in); the value the atomic returns is the row's rank inside its bucket, so the fill pass
      // needs no second round of atomics.  (Issuing the atomics of all tables together after this
      // loop, one round trip per wave instead of one per table, measured 0.385 -> 0.405 ms for the
      // 1M + 1M rows of the sorted form, no change for the database pass alone: kept per table.)
      if (counts)
        ranks[(size_t)j * nrows_total + r] = atomicAdd(&counts[(size_t)j * (nb + 1) + (code & hbmask)], 1u);
    }
    if (IS_QUERY) {
      uint32_t mask = 0;
#pragma unroll
      for (int q = 0; q < GMAX; ++q)
        if (q < g && bbit[q] >= 0) mask |= 1u << bbit[q];
      if (r < nrows) masks[(size_t)j * nrows_total + r] = mask;
    }
  }
}

// Epilogue of project_mfma_kernel: the wave's 4 x CT accumulator tiles are transposed through its LDS
// tile so that lane l holds the projections of row row0 + l, then the row part above.
template <int CT, bool IS_QUERY, int GMAX>
__device__ __forceinline__ void project_mfma_epilogue(
    const mfma_f4 (&acc)[4][CT], float *ws, int lane, long long row0, long long row_limit, long long nrows_total,
    int m, int n, int g, uint32_t *__restrict__ codes, uint32_t *__restrict__ masks,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ ranks, uint32_t hbmask, int nb) {
  constexpr int NC = 16 * CT;
  constexpr int ES = NC + 1;
  const int r16 = lane & 15, q4 = lane >> 4;
  // transpose: D tile element v of lane (r16, q4) is row 16*rt + 4*q4 + v, column 16*ct + r16
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int v = 0; v < 4; ++v) ws[(16 * rt + 4 * q4 + v) * ES + 16 * ct + r16] = acc[rt][ct][v];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  project_rows_epilogue<IS_QUERY, GMAX>(ws + lane * ES, NC, lane, row0, row_limit, nrows_total, m, n, g, codes, masks,
                                        counts, ranks, hbmask, nb);
}

template <int CT, bool IS_QUERY, int GMAX, bool FULL>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(CT <= 3 ? 3 : 2))) void project_mfma_kernel(
    const float *__restrict__ rows, int nrows, int dim, int m, int n, int g,
    const float *__restrict__ dictm,     // [roundup(dim, 32)][16*CT], column = table*m + bit, zero padded
    uint32_t *__restrict__ codes, uint32_t *__restrict__ masks, uint8_t *__restrict__ u8img,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ ranks, uint32_t hbmask, int nb) {
  constexpr int NC = 16 * CT;
  constexpr int KS = kMfmaChunk / 4;   // MFMA k-steps per chunk
  constexpr int NX = kMfmaChunk / 4;   // float4 loads per lane per chunk (64 rows x 32 dims / 64 lanes / 4)
  constexpr int XS = kMfmaChunk + 1;   // floats per staged row (+1: spreads the rows over the banks)
  constexpr int ES = NC + 1;           // floats per row of the transposed projections
  constexpr int kWaveFloats = 64 * (ES > XS ? ES : XS);
  __shared__ float lds[(kThreads / 64) * kWaveFloats];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *ws = lds + wv * kWaveFloats;
  const long long row0 = ((long long)blockIdx.x * (kThreads / 64) + wv) * 64;
  if (row0 >= nrows) return;  // whole wave past the end (nothing below synchronises across waves)
  const int r16 = lane & 15, q4 = lane >> 4;
  const int quad = lane & 3;

  mfma_f4 acc[4][CT];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = mfma_f4{0.f, 0.f, 0.f, 0.f};

  float4 px[NX];
  // FULL: dim is a whole number of 32-dim chunks, every load is unconditional.  Otherwise
  // (dim % 32 == 16) the dims past the end of a row are staged as zeros and multiply the zero
  // rows the repacked dictionary is padded with: fmaf(0, 0, acc) leaves every accumulator as it is
  auto prefetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int e = lane + 64 * j;  // (row of the tile, 4-dim part of the step): 8 lanes per row
      const long long grow = min(row0 + (e >> 3), (long long)nrows - 1);
      const int d0 = c0 + 4 * (e & 7);
      if (FULL || d0 < dim) {  // read once: keep the rows out of the way of the hyperplanes in L1
        const mfma_f4 t4 = __builtin_nontemporal_load(reinterpret_cast<const mfma_f4 *>(rows + (size_t)grow * dim + d0));
        px[j] = make_float4(t4[0], t4[1], t4[2], t4[3]);
      } else {
        px[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  // Hyperplane operands of a whole chunk (KS k-steps x CT column tiles, L1 / L2 resident), requested
  // right after the previous chunk's MFMAs -- i.e. AFTER that chunk's row prefetch and BEFORE this
  // chunk's stores and the next row prefetch.  Vector-memory operations of a wave complete in issue
  // order (s_waitcnt vmcnt counts the youngest N): fetched inside the MFMA loop, behind the
  // prefetch, every operand wait would also wait for the HBM stream.
  float bq[KS][CT];
  auto fetch_b_chunk = [&](int c0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bq[ks][ct] = dictm[(size_t)(c0 + 4 * ks + q4) * NC + 16 * ct + r16];
  };
  prefetch(0);
  fetch_b_chunk(0);
  for (int c0 = 0; c0 < dim; c0 += kMfmaChunk) {
    // stage the chunk in LDS (A operand order) and assemble the uint8 image: the four lanes of a
    // quad hold 16 consecutive bytes of a row; quad broadcasts give every lane all four dwords,
    // lane t of the quad keeps those of load j = t (mod 4), so that two 16-byte-per-lane stores
    // per chunk replace eight 4-byte ones
    uint4 keep = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int e = lane + 64 * j;
      const int row = e >> 3, part = e & 7;
      const float4 v = px[j];
      float *dstp = ws + row * XS + 4 * part;
      dstp[0] = v.x;
      dstp[1] = v.y;
      dstp[2] = v.z;
      dstp[3] = v.w;
      const uint32_t pk = f2u8(v.x) | (f2u8(v.y) << 8) | (f2u8(v.z) << 16) | (f2u8(v.w) << 24);
      const uint32_t g0 = __builtin_amdgcn_mov_dpp(pk, 0x00, 0xF, 0xF, true);  // quad_perm [0,0,0,0]
      const uint32_t g1 = __builtin_amdgcn_mov_dpp(pk, 0x55, 0xF, 0xF, true);  // [1,1,1,1]
      const uint32_t g2 = __builtin_amdgcn_mov_dpp(pk, 0xAA, 0xF, 0xF, true);  // [2,2,2,2]
      const uint32_t g3 = __builtin_amdgcn_mov_dpp(pk, 0xFF, 0xF, 0xF, true);  // [3,3,3,3]
      if (quad == (j & 3)) keep = make_uint4(g0, g1, g2, g3);
      if ((j & 3) == 3) {
        const int krow = 8 * ((j & ~3) + quad) + (lane >> 3);  // row of the load this lane kept
        const int kbyte = c0 + 16 * ((lane & 7) >> 2);         // first byte of the quad's 16
        if (row0 + krow < nrows && (FULL || kbyte < dim))
          *reinterpret_cast<uint4 *>(u8img + (size_t)(row0 + krow) * dim + kbyte) = keep;
      }
    }
    if (c0 + kMfmaChunk < dim) prefetch(c0 + kMfmaChunk);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float a[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) a[rt] = ws[(16 * rt + r16) * XS + 4 * ks + q4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], bq[ks][ct], acc[rt][ct], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (c0 + kMfmaChunk < dim) fetch_b_chunk(c0 + kMfmaChunk);
  }

  project_mfma_epilogue<CT, IS_QUERY, GMAX>(acc, ws, lane, row0, nrows, nrows, m, n, g, codes, masks, counts, ranks,
                                            hbmask, nb);
}

// The same projection when n*m is not a whole number of 16-column tiles and 1..8 columns are left
// over (the default configuration: n*m = 2 x 17 = 34 = two tiles + 2): the left-over columns go
// through v_mfma_f32_4x4x1_16B_f32 -- 16 blocks of D[4x4] += A[4x1] B[1x4], i.e. 64 rows x 4 columns
// x ONE k per instruction at ~11 cycles per SIMD -- instead of a third, 7/8-empty 16-column tile at
// 32 cycles per 4 k and 16 rows (measured, tools/exp/mfma_4x4_exp.hip: lane l feeds A[row l] and
// B[column l % 4], D[i] of lane l is row 4 (l / 4) + i, column l % 4; 51 200 of 51 200 outputs equal the
// std::fmaf chain in k order).  -21 % matrix-pipe time per row at n*m = 34.  Differences from
// project_mfma_kernel besides that: whole-chunk dims only (dim % 32 == 0, dim <= 512; anything else
// takes project_mfma_kernel, as does SPECTAVI_CASCADE_MFMA4=0 for A/B runs); staged rows
// are 16-byte aligned (36 floats: one ds_write_b128 per load, one ds_read_b128 per k-step for the
// 4x4x1 A operand); the left-over hyperplanes sit in a workgroup-shared LDS table [dim/4][4][4].
// workgroups per CU by LDS (4 wave tiles + the left-over table): three for the default shape
constexpr int mfma4_waves_per_simd(int ct, int ng) {
  // wave tiles + uint8 staging (static) + the left-over table at its largest (dim 512, dynamic)
  return 4 * 64 * 4 * (16 * ct + 4 * ng + 1 > 36 ? 16 * ct + 4 * ng + 1 : 36) + 4 * 2048 + ng * 8192 <= 160 * 1024 / 3 ? 3 : 2;
}

template <int CT, int NG, bool IS_QUERY, int GMAX>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(mfma4_waves_per_simd(CT, NG)))) void project_mfma4_kernel(
    const float *__restrict__ rows, int nrows, int dim, int m, int n, int g,
    const float *__restrict__ dictm,     // [dim][NCD], column = table*m + bit, zero padded; NCD = 16*(CT+1)
    uint32_t *__restrict__ codes, uint32_t *__restrict__ masks, uint8_t *__restrict__ u8img,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ ranks, uint32_t hbmask, int nb) {
  constexpr int NCD = 16 * (CT + 1);     // row length of dictm (the layout the 16-column kernel uses)
  constexpr int NCOL = 16 * CT + 4 * NG; // columns computed here
  constexpr int KS = kMfmaChunk / 4;
  constexpr int NX = kMfmaChunk / 4;
  constexpr int XS = kMfmaChunk + 4;     // floats per staged row: 16-byte aligned rows
  constexpr int ES = NCOL + 1;           // floats per row of the transposed projections
  constexpr int kWaveFloats = 64 * (ES > XS ? ES : XS);
  __shared__ __attribute__((aligned(16))) float lds[(kThreads / 64) * kWaveFloats];
  // uint8 image of the chunk being staged, per wave: 64 rows x 32 bytes, written one packed dword per
  // lane and load (byte offset 256 j + 4 lane = row-major) and read back 16 bytes per lane for the
  // stores.  Round 2 assembled those 16 bytes with four quad broadcasts and a predicated copy per
  // load: 80 VALU instructions per chunk, a fifth of the database pass's.
  __shared__ __attribute__((aligned(16))) uint32_t u8s[kThreads / 64][64 * kMfmaChunk / 4];
  // left-over hyperplanes [group][k / 4][column][k % 4], sized by the launch: NG * dim * 16 bytes
  // (8 KB per group at dim 512, 2 KB at 128: statically sized for 512 they cost dim 128 its third
  // workgroup per CU once the uint8 staging above was added)
  extern __shared__ __attribute__((aligned(16))) float hl_dyn[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *ws = lds + wv * kWaveFloats;
  uint32_t *u8w = u8s[wv];
  const int hl_group = dim * 4;  // floats per group
  const long long row0 = ((long long)blockIdx.x * (kThreads / 64) + wv) * 64;
  const int r16 = lane & 15, q4 = lane >> 4;
  const int quad = lane & 3;

  mfma_f4 acc[4][CT > 0 ? CT : 1];
  mfma_f4 accl[NG];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = mfma_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) accl[gq] = mfma_f4{0.f, 0.f, 0.f, 0.f};

  float4 px[NX];
  auto prefetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int e = lane + 64 * j;  // (row of the tile, 4-dim part of the step): 8 lanes per row
      const long long grow = min(row0 + (e >> 3), (long long)nrows - 1);
      const mfma_f4 t4 = __builtin_nontemporal_load(reinterpret_cast<const mfma_f4 *>(rows + (size_t)grow * dim + c0 + 4 * (e & 7)));
      px[j] = make_float4(t4[0], t4[1], t4[2], t4[3]);
    }
  };
  // operands of the 16-column tiles for a whole chunk, requested ahead of the next row prefetch
  // (see project_mfma_kernel)
  float bq[KS][CT > 0 ? CT : 1];
  // one per-lane base pointer, the (k-step, tile) offsets as immediates: the sixteen separately
  // indexed loads of round 2 kept sixteen 64-bit addresses (32 VGPRs) alive across the chunk loop
  const float *bp = dictm + (size_t)q4 * NCD + r16;
  auto fetch_b_chunk = [&](int c0) {
    const float *bc = bp + (size_t)c0 * NCD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bq[ks][ct] = bc[4 * ks * NCD + 16 * ct];
  };
  // left-over hyperplanes -> LDS (whole workgroup; the only workgroup-level step).  (Requesting the
  // wave's first row chunk BEFORE this staging was tried: __syncthreads() waits for vmcnt(0), so the
  // HBM round trip moved in front of the barrier instead of under it: 0.385 -> 0.41 ms.)
  for (int e = threadIdx.x; e < NG * dim * 4; e += kThreads) {
    const int gq = e / (dim * 4), rem = e - gq * dim * 4;
    const int k = rem >> 2, j = rem & 3;
    hl_dyn[gq * hl_group + ((k >> 2) * 4 + j) * 4 + (k & 3)] = dictm[(size_t)k * NCD + 16 * CT + 4 * gq + j];
  }
  __syncthreads();
  if (row0 >= nrows) return;  // whole wave past the end (nothing below synchronises across waves)
  prefetch(0);
  fetch_b_chunk(0);
  for (int c0 = 0; c0 < dim; c0 += kMfmaChunk) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int e = lane + 64 * j;
      const int row = e >> 3, part = e & 7;
      const float4 v = px[j];
      *reinterpret_cast<float4 *>(ws + row * XS + 4 * part) = v;
      // row (lane >> 3) + 8 j, bytes 4 (lane & 7) .. +3 of its 32: dword 64 j + lane of the image
      u8w[64 * j + lane] = f2u8(v.x) | (f2u8(v.y) << 8) | (f2u8(v.z) << 16) | (f2u8(v.w) << 24);
    }
    if (c0 + kMfmaChunk < dim) prefetch(c0 + kMfmaChunk);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the chunk's uint8 image out: 16-byte piece p = lane + 64 k is half (p & 1) of row p >> 1
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int piece = lane + 64 * k;
      const uint4 img = *reinterpret_cast<const uint4 *>(u8w + 4 * piece);
      if (row0 + (piece >> 1) < nrows)
        *reinterpret_cast<uint4 *>(u8img + (size_t)(row0 + (piece >> 1)) * dim + c0 + 16 * (piece & 1)) = img;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // left-over columns first in program order (any order is fine: separate accumulators)
      const float4 al = *reinterpret_cast<const float4 *>(ws + lane * XS + 4 * ks);  // row = lane, 4 consecutive k
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const float4 bl = *reinterpret_cast<const float4 *>(hl_dyn + gq * hl_group + (((c0 >> 2) + ks) * 4 + quad) * 4);
        accl[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(al.x, bl.x, accl[gq], 0, 0, 0);
        accl[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(al.y, bl.y, accl[gq], 0, 0, 0);
        accl[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(al.z, bl.z, accl[gq], 0, 0, 0);
        accl[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(al.w, bl.w, accl[gq], 0, 0, 0);
      }
      if constexpr (CT > 0) {
        float a[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) a[rt] = ws[(16 * rt + r16) * XS + 4 * ks + q4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], bq[ks][ct], acc[rt][ct], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (c0 + kMfmaChunk < dim) fetch_b_chunk(c0 + kMfmaChunk);
  }
  // transpose to one row per lane: 16-column tiles (D element v of lane (r16, q4) is row
  // 16 rt + 4 q4 + v, column 16 ct + r16), then the 4-column groups (D[i] of lane l is row
  // 4 (l / 4) + i, column l % 4)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int v = 0; v < 4; ++v) ws[(16 * rt + 4 * q4 + v) * ES + 16 * ct + r16] = acc[rt][ct][v];
#pragma unroll
  for (int gq = 0; gq < NG; ++gq)
#pragma unroll
    for (int i = 0; i < 4; ++i) ws[(4 * (lane >> 2) + i) * ES + 16 * CT + 4 * gq + quad] = accl[gq][i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  project_rows_epilogue<IS_QUERY, GMAX>(ws + lane * ES, NCOL, lane, row0, nrows, nrows, m, n, g, codes, masks, counts,
                                        ranks, hbmask, nb);
}

// ---------------------------------------------------------------------------------
// 4-6. counting sort of database indices by bucket, per table
// ---------------------------------------------------------------------------------
// Exclusive scan of counts[j][0..nb] in place, three small kernels: per-segment sums
// (1024 entries each), a scan of the segment sums (one workgroup per table), and the
// segment-local scans with the carried offset.
constexpr int kScanSeg = 1024;

__global__ __launch_bounds__(256) void bucket_segsum_kernel(const uint32_t *__restrict__ counts,
                                                            int nb, int nseg,
                                                            uint32_t *__restrict__ segsum) {
  __shared__ uint32_t wsum[4];
  const int j = blockIdx.y, seg = blockIdx.x;
  const uint32_t *c = counts + (size_t)j * (nb + 1);
  uint32_t v = 0;
  for (int e = seg * kScanSeg + threadIdx.x; e < min((seg + 1) * kScanSeg, nb + 1); e += 256) v += c[e];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) segsum[(size_t)j * nseg + seg] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// in-place exclusive scan of segsum[j][0..nseg) (nseg <= 4097 for 2^22 buckets)
__global__ __launch_bounds__(1024) void bucket_segscan_kernel(uint32_t *__restrict__ segsum, int nseg) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  uint32_t *c = segsum + (size_t)blockIdx.x * nseg;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nseg; base += 1024) {
    const int e = base + t;
    const uint32_t v = e < nseg ? c[e] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    const uint32_t excl = carry + woff + incl - v;
    if (e < nseg) c[e] = excl;
    __syncthreads();
    if (t == 1023) carry = excl + v;
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void bucket_scan_kernel(uint32_t *__restrict__ counts, int nb,
                                                           int nseg,
                                                           const uint32_t *__restrict__ segoff) {
  __shared__ uint32_t wsum[16];
  const int j = blockIdx.y, seg = blockIdx.x;
  uint32_t *c = counts + (size_t)j * (nb + 1);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int e = seg * kScanSeg + t;
  const uint32_t v = e <= nb ? c[e] : 0u;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t woff = segoff[(size_t)j * nseg + seg];
  for (int k = 0; k < w; ++k) woff += wsum[k];
  const uint32_t excl = woff + incl - v;
  if (e <= nb) c[e] = excl;
}

// Histogram + ranks of the QUERY sign codes, per table, for the VALU projection path (the
// matrix-core epilogue has it fused): what the probe's query order is built from.
__global__ __launch_bounds__(256) void query_rank_kernel(const uint32_t *__restrict__ codes, int N, int n,
                                                         uint32_t hbmask, int nb, uint32_t *__restrict__ counts,
                                                         uint32_t *__restrict__ ranks) {
  const size_t total = (size_t)n * N;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t j = e / N;
    ranks[e] = atomicAdd(&counts[j * (nb + 1) + (codes[e] & hbmask)], 1u);
  }
}

__global__ void bucket_fill_kernel(const uint32_t *__restrict__ codes,
                                   const uint32_t *__restrict__ ranks, int M, int n, uint32_t hbmask,
                                   int nb, const uint32_t *__restrict__ bstart,
                                   uint32_t *__restrict__ order) {
  const size_t total = (size_t)n * M;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(e / M);
    const uint32_t idx = (uint32_t)(e - (size_t)j * M);
    const uint32_t pos = bstart[(size_t)j * (nb + 1) + (codes[e] & hbmask)] + ranks[e];
    order[(size_t)j * M + pos] = idx;
  }
}

// ---------------------------------------------------------------------------------
// 7. probe + gather + refine, one wave per query
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_sad_u8(a, b, c);
}

// sum over the 8 lanes of an aligned lane group (DPP: half-mirror, xor 1, xor 2)
__device__ __forceinline__ uint32_t group8_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141 /*row_half_mirror*/, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true);
  return v;
}

__device__ __forceinline__ void top2_insert_distinct(uint64_t &k1, uint64_t &k2, uint64_t k) {
  if (k == k1 || k == k2) return;  // same database row reached through another table
  if (k < k1) {
    k2 = k1;
    k1 = k;
  } else if (k < k2) {
    k2 = k;
  }
}

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int mask) {
  const uint32_t lo = __shfl_xor((uint32_t)v, mask, 64);
  const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), mask, 64);
  return ((uint64_t)hi << 32) | lo;
}

// two smallest DISTINCT keys of {a1,a2,b1,b2}
__device__ __forceinline__ void merge_distinct(uint64_t &a1, uint64_t &a2, uint64_t b1, uint64_t b2) {
  const uint64_t lo = a1 < b1 ? a1 : b1;
  uint64_t second = kNone64;
  if (a1 > lo && a1 < second) second = a1;
  if (a2 > lo && a2 < second) second = a2;
  if (b1 > lo && b1 < second) second = b1;
  if (b2 > lo && b2 < second) second = b2;
  a1 = lo;
  a2 = second;
}

// Spread the low bits of v over the set bits of mask (software pdep).
__device__ __forceinline__ uint32_t deposit_bits(uint32_t v, uint32_t mask) {
  uint32_t out = 0;
  while (mask) {
    const uint32_t low = mask & (0u - mask);
    if (v & 1u) out |= low;
    v >>= 1;
    mask ^= low;
  }
  return out;
}

// CPL: 16-byte chunks of the row per lane of an 8-lane group (dim <= 128*CPL);
// RU: rounds of 8 candidate rows in flight per wave
template <int CPL, int RU>
__global__ __launch_bounds__(kThreads) void probe_refine_kernel(
    const uint8_t *__restrict__ ux, const uint8_t *__restrict__ uy, int M, int N, int dim, int m,
    int n, int g, int hb, const uint32_t *__restrict__ xcodes, const uint32_t *__restrict__ ysign,
    const uint32_t *__restrict__ ymask, const uint32_t *__restrict__ bstart,
    const uint32_t *__restrict__ order,
    uint64_t *__restrict__ out_idx, float *__restrict__ out_dist, int32_t *__restrict__ out_ncand) {
  __shared__ uint32_t lists[kThreads / 64][kListCap];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // one query per wave
  auto process_query = [&](const int query) {
  uint32_t *list = lists[wave];
  const int sub = lane & 7;    // chunk owner inside the 8-lane group
  const int grp = lane >> 3;   // candidate slot 0..7
  const int nchunk = dim / 16;
  const uint32_t nb = 1u << hb;
  const uint32_t hbmask = nb - 1;
  const bool check_code = m > hb;

  // this lane's share of the query row
  uint4 qv[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = sub + 8 * c;
    qv[c] = ch < nchunk ? *reinterpret_cast<const uint4 *>(uy + (size_t)query * dim + 16 * ch)
                        : make_uint4(0, 0, 0, 0);
  }

  uint64_t k1 = kNone64, k2 = kNone64;
  int visited = 0;

  auto refine = [&](int count) {
    // `count` candidate indices in list[0..count).  Each 8-lane group takes one candidate
    // per round (8 rows = 8 full 128-byte lines per wave instruction); four rounds of row
    // gathers are issued before the first is consumed so the random-row latency overlaps.
    for (int c0 = 0; c0 < count; c0 += 8 * RU) {
      uint4 xv[RU][CPL];
      uint32_t cand[RU];
      bool live[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int ci = c0 + 8 * u + grp;
        live[u] = ci < count;
        cand[u] = list[live[u] ? ci : 0];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int ch = sub + 8 * c;
          xv[u][c] = make_uint4(0, 0, 0, 0);
          if (live[u] && ch < nchunk)
            xv[u][c] = *reinterpret_cast<const uint4 *>(ux + (size_t)cand[u] * dim + 16 * ch);
        }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        uint32_t d = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          d = sad_u8(qv[c].x, xv[u][c].x, d);
          d = sad_u8(qv[c].y, xv[u][c].y, d);
          d = sad_u8(qv[c].z, xv[u][c].z, d);
          d = sad_u8(qv[c].w, xv[u][c].w, d);
        }
        d = group8_sum(d);
        if (live[u]) top2_insert_distinct(k1, k2, ((uint64_t)d << 32) | cand[u]);
      }
    }
  };

  const int nvar = 1 << g;
  const int nprobe = n * nvar;
  // lanes per probe: with few probes (n * 2^g = 8 by default) several lanes share one
  // bucket and copy it cooperatively
  int lpp = 1;
  while (lpp * 2 * nprobe <= 64) lpp *= 2;
  const int ppc = 64 / lpp;  // probes per pass of the wave
  for (int p0 = 0; p0 < nprobe; p0 += ppc) {
    const int p = p0 + lane / lpp;
    const int psub = lane % lpp;
    uint32_t s = 0, len = 0, pcode = 0;
    int tj = 0;
    if (p < nprobe) {
      tj = p >> g;
      const uint32_t var = (uint32_t)(p & (nvar - 1));
      const uint32_t sg = ysign[(size_t)tj * N + query];
      const uint32_t mk = ymask[(size_t)tj * N + query];
      pcode = (sg & ~mk) | deposit_bits(var, mk);
      const uint32_t *bs = bstart + (size_t)tj * (nb + 1) + (pcode & hbmask);
      s = bs[0];
      len = bs[1] - s;
    }
    // exclusive offsets of the buckets in the list: scan over the first lane of each probe
    const uint32_t mylen = psub == 0 ? len : 0u;
    uint32_t incl = mylen;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (!check_code && total <= (uint32_t)kListCap) {
      // fast path: the lanes of a probe copy its bucket into the shared list
      const uint32_t dst = __shfl(incl - mylen, lane - psub, 64);
      const uint32_t *src = order + (size_t)tj * M + s;
      for (uint32_t i = psub; i < len; i += lpp) list[dst + i] = src[i];
      __builtin_amdgcn_wave_barrier();
      refine((int)total);
      visited += (int)total;
      __builtin_amdgcn_wave_barrier();
    } else {
      // general path: buckets one at a time, 64 entries per step, optional full-code check
      const int np = min(ppc, nprobe - p0);
      for (int pp = 0; pp < np; ++pp) {
        const uint32_t ps = __shfl(s, pp * lpp, 64);
        const uint32_t pl = __shfl(len, pp * lpp, 64);
        const uint32_t pc = __shfl(pcode, pp * lpp, 64);
        const int pj = __shfl(tj, pp * lpp, 64);
        for (uint32_t off = 0; off < pl; off += 64) {
          const uint32_t e = off + lane;
          bool keep = e < pl;
          uint32_t cand = 0;
          if (keep) {
            cand = order[(size_t)pj * M + ps + e];
            if (check_code) keep = xcodes[(size_t)pj * M + cand] == pc;
          }
          const unsigned long long bal = __ballot(keep);
          const int pos = __popcll(bal & ((1ull << lane) - 1ull));
          if (keep) list[pos] = cand;
          const int cnt = __popcll(bal);
          __builtin_amdgcn_wave_barrier();
          refine(cnt);
          visited += cnt;
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }

  // argmin-2 butterfly over the 8 lane groups (keys are uniform inside a group)
#pragma unroll
  for (int msk = 8; msk < 64; msk <<= 1) {
    const uint64_t b1 = shfl_xor64(k1, msk);
    const uint64_t b2 = shfl_xor64(k2, msk);
    merge_distinct(k1, k2, b1, b2);
  }
  if (lane == 0) {
    const bool n1 = k1 == kNone64, n2 = k2 == kNone64;
    out_idx[2 * (size_t)query + 0] = n1 ? ~0ull : (k1 & 0xFFFFFFFFull);
    out_idx[2 * (size_t)query + 1] = n2 ? ~0ull : (k2 & 0xFFFFFFFFull);
    // int -> float as reference src/CascadingHashNn.h:244; sentinel INT_MAX -> 2147483648.0f
    out_dist[2 * (size_t)query + 0] = n1 ? 2147483648.0f : (float)(uint32_t)(k1 >> 32);
    out_dist[2 * (size_t)query + 1] = n2 ? 2147483648.0f : (float)(uint32_t)(k2 >> 32);
    if (out_ncand) out_ncand[query] = visited;
  }
  };  // process_query

  const int nwaves = gridDim.x * (kThreads / 64);
  const int wave_id = blockIdx.x * (kThreads / 64) + wave;
  for (int query = wave_id; query < N; query += nwaves) process_query(query);
}


// ---------------------------------------------------------------------------------
// 7b. group-per-query probe + refine, one table per launch: one 8-lane group per query, 8 queries
// per wave, 32 per workgroup.  The one-wave-per-query kernel above spends ~1100 wave instructions
// per query and is issue-bound; here every wave instruction serves 8 queries: lane s of a group owns
// probe s of each round of 8 probes (the default 2^g is 4), the group lays the probed buckets'
// entries out in a private LDS list, and in every round each group gathers ONE candidate row
// (8 lanes x 16 bytes = one 128-byte line, 8 different queries' lines per wave instruction),
// v_sad_u8 + DPP sum, branch-free top-2.  All 8 lanes of a group hold the same keys, so no
// cross-lane merge is needed at the end.  A round whose candidate stream is longer than 128
// entries is processed in several windows of the list.
//
// Round 3: the kernel runs once per TABLE, over the queries in the order of that table's sign code
// (counting sort fused into the query projection like the database's), with the query blocks dealt
// so that the blocks sharing an XCD (b % 8 under the dispatcher's round-robin; speed only) walk one
// contiguous eighth of the sorted order: the ~7.6 queries of a bucket, and the queries of the
// buckets one low bit away, are then in flight together on one XCD and find that bucket's rows in
// its L2 instead of gathering them again over the fabric (1M x 1M: FETCH_SIZE 13.3 -> 3.7 GB per
// step, L2 hit rate 5 -> 60 %, profiles/r03_cascade_pmc.md).  Between the passes a query's two best
// keys and its candidate count wait in `partial` / `pvisited`; the last pass writes the ABI outputs.
// Small inputs skip the sort (qorder = NULL, input order) but still go table by table.  Results do
// not depend on the order of the queries or of the tables: the two smallest distinct keys.
//
// That made the kernel issue-bound (SQ_INSTS_VALU 2 800 per wave and pass = 0.57 of the pass's
// 0.64 ms), so the instruction count per candidate went from ~54 to ~30:
//   * keys ordered by v_min_f64 / v_max_f64: a key (distance << 32 | index) is a positive double
//     whose order is the integer order of its bits (distances < 2^20: denormals, which the f64 units
//     keep), "none" = the largest finite double; the top-2 network is 3 instructions instead of
//     3 x (v_cmp_u64 + 2 v_cndmask) + hazard nops;
//   * "same row again" (a row reached through an earlier table, or twice when nan projections
//     leave a query with fewer than g probe bits) is a 32-bit index comparison with the two kept rows
//     at insertion instead of two 64-bit key comparisons;
//   * the candidate list of a round of probes is filled flat: list position p finds its bucket by a
//     packed byte comparison against the 8 bucket offsets (4 instructions) and loads its entry, four
//     independent loads per lane in flight -- rounds 1-2 walked the buckets one by one, one dependent
//     global load per 8 entries, every one of those round trips serial;
//   * four list entries per ds_read_b128; row addresses as a uniform base + 32-bit shifted offset
//     (dim a power of two, the image below 4 GiB) instead of two v_mad_u64_u32 per row; unconditional
//     loads (row 0 / entry 0 for the slots past the end) instead of an exec-mask branch per load;
//   * RU = 8 (rows of up to 128 bytes): eight rows in flight per group, and instead of all eight lanes
//     reducing every candidate's eight partial sums (3 DPP adds each) a transposing butterfly leaves
//     lane s with the whole sum of candidate s (7 DPP adds + 14 selects per 8 candidates); each lane
//     keeps the best two of ITS candidates, the eight pairs are merged once at the end of the pass
//     (equal keys = the same row, counted once).  ~12 instructions per candidate.
// ---------------------------------------------------------------------------------
constexpr uint64_t kNoneD = 0x7FEFFFFFFFFFFFFFull;

__device__ __forceinline__ uint64_t key_min(uint64_t a, uint64_t b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  return (uint64_t)__double_as_longlong(r);
}
__device__ __forceinline__ uint64_t key_max(uint64_t a, uint64_t b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  return (uint64_t)__double_as_longlong(r);
}

// (k1 <= k2) of this lane and of the lane CTRL pairs it with -> the two smallest distinct keys of the four
template <int CTRL>
__device__ __forceinline__ void merge_keys_with(uint64_t &k1, uint64_t &k2) {
  const uint64_t o1 = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k1 >> 32), CTRL, 0xF, 0xF, true) << 32) |
                      (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k1, CTRL, 0xF, 0xF, true);
  const uint64_t o2 = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k2 >> 32), CTRL, 0xF, 0xF, true) << 32) |
                      (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k2, CTRL, 0xF, 0xF, true);
  const uint64_t lo = key_min(k1, o1);
  uint64_t mid = key_max(k1, o1);
  if (mid == lo) mid = kNoneD;
  k2 = key_min(mid, key_min(k2, o2));
  k1 = lo;
}

template <int CPL, int RU, int WPE, bool SHIFT, bool FULL>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void probe_table_kernel(
    const uint8_t *__restrict__ ux, const uint8_t *__restrict__ uy, int M, int N, int dim, int dshift, int g, int hb,
    const uint32_t *__restrict__ ysign,   // [N] sign codes of THIS table
    const uint32_t *__restrict__ ymask,   // [N]
    const uint32_t *__restrict__ bstart,  // [2^hb + 1] bucket offsets of this table
    const uint32_t *__restrict__ order,   // [M] database rows grouped by bucket
    const uint32_t *__restrict__ qorder,  // [N] queries in this pass's order, or NULL (identity)
    int nblk, int per_xcd,                // query blocks; blocks per XCD range (0: block b takes slot block b)
    uint64_t *__restrict__ partial, int32_t *__restrict__ pvisited, int first_pass, int last_pass,
    uint64_t *__restrict__ out_idx, float *__restrict__ out_dist, int32_t *__restrict__ out_ncand) {
  constexpr int CAP = 128;
  static_assert(RU == 4 || RU == 8, "list entries come four per ds_read_b128");
  __shared__ __attribute__((aligned(16))) uint32_t lists[kThreads / 8][CAP];
  __shared__ __attribute__((aligned(16))) uint32_t gtab[kThreads / 8][16];  // [0..7] bucket offsets, [8..15] start - offset
  const int t = threadIdx.x;
  const int sub = t & 7;
  int slotblk = blockIdx.x;
  if (per_xcd > 0) {
    slotblk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (slotblk >= nblk) return;  // whole workgroup (nothing below synchronises across waves)
  }
  const int slot = slotblk * (kThreads / 8) + (t >> 3);
  const bool valid = slot < N;
  const int sq = valid ? slot : N - 1;  // keep every lane alive for the cross-lane ops
  const int q = qorder ? (int)qorder[sq] : sq;
  uint32_t *list = lists[t >> 3];
  uint32_t *tab = gtab[t >> 3];
  const int nchunk = dim / 16;
  const uint32_t hbmask = (1u << hb) - 1;
  const int nvar = 1 << g;

  uint4 qv[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = sub + 8 * c;
    qv[c] = ch < nchunk ? *reinterpret_cast<const uint4 *>(uy + (size_t)q * dim + 16 * ch)
                        : make_uint4(0, 0, 0, 0);
  }
  const uint32_t sg = ysign[q], mk = ymask[q];
  uint64_t k1 = kNoneD, k2 = kNoneD;
  int visited = 0;
  if (!first_pass) {
    k1 = partial[2 * (size_t)q];
    k2 = partial[2 * (size_t)q + 1];
    visited = pvisited[q];
  }

  for (int p0 = 0; p0 < nvar; p0 += 8) {
    // lane `sub` owns probe p0 + sub of this round
    const int p = p0 + sub;
    uint32_t s = 0, len = 0;
    if (p < nvar) {
      const uint32_t pcode = (sg & ~mk) | deposit_bits((uint32_t)p, mk);
      const uint32_t *bs = bstart + (pcode & hbmask);
      s = bs[0];
      len = bs[1] - s;
    }
    // exclusive offsets of the 8 buckets inside the round's candidate stream
    uint32_t incl = len;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 8);
      if (sub >= d) incl += o;
    }
    const uint32_t total = __shfl(incl, 7, 8);
    const uint32_t excl = incl - len;
    visited += (int)total;
    tab[sub] = excl;
    tab[8 + sub] = s - excl;  // entry of stream position x of this bucket: order[x + (s - excl)]
    __builtin_amdgcn_wave_barrier();
    // the stream [0, total) is processed in windows of CAP entries (one almost always)
    for (uint32_t w0 = 0; __any(w0 < total); w0 += CAP) {
      const uint32_t T = w0 < total ? min((uint32_t)CAP, total - w0) : 0u;
      // bucket offsets relative to the window, clamped to [0, 128], one byte each
      uint32_t thrA, thrB;
      {
        const uint4 ea = *reinterpret_cast<const uint4 *>(tab), eb = *reinterpret_cast<const uint4 *>(tab + 4);
        auto rel = [&](uint32_t e) { return (uint32_t)min(max((int)(e - w0), 0), 128); };
        thrA = rel(ea.x) | (rel(ea.y) << 8) | (rel(ea.z) << 16) | (rel(ea.w) << 24);
        thrB = rel(eb.x) | (rel(eb.y) << 8) | (rel(eb.z) << 16) | (rel(eb.w) << 24);
      }
      // (not unrolled: unrolled, the sixteen lane-constant positions and their byte-replicated forms
      // were hoisted out of every loop and held 31 VGPRs through the gather loop -- 88 in all, five
      // waves per SIMD)
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {  // 32 list positions per step, four independent loads per lane
        if (!__any(T > 32u * it)) break;
        uint32_t ent[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t pos = (uint32_t)sub + 8u * (4 * it + i);  // < 128
          // bucket of position pos: the last one whose offset is <= pos.  Per byte, bit 7 of
          // 0x80 + pos - offset is set iff offset <= pos (pos <= 127, offset <= 128: no borrow)
          const uint32_t rep = (pos * 0x01010101u) | 0x80808080u;
          const uint32_t cnt = __builtin_popcount((rep - thrA) & 0x80808080u) + __builtin_popcount((rep - thrB) & 0x80808080u);
          const uint32_t delta = tab[8 + ((cnt - 1) & 7)];
          // unconditional load (entry 0 for the positions past the end): no exec-mask branch per entry
          ent[i] = order[pos < T ? delta + w0 + pos : 0u];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) list[sub + 8 * (4 * it + i)] = ent[i];
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr (RU == 8) {
        // Eight candidate rows per round; afterwards lane s of the group OWNS candidate s: the eight
        // partial sums of the eight lanes are reduced by a transposing butterfly (7 DPP adds + 14
        // selects per 8 candidates instead of 3 DPP adds per candidate), and the top-2 update runs once
        // per lane and round on that lane's candidate.  The lanes' pairs are merged after the last round.
        const bool up4 = (sub & 4) != 0, up2 = (sub & 2) != 0, up1 = (sub & 1) != 0;
        for (uint32_t f0 = 0; __any(f0 < T); f0 += 8) {
          const uint32_t fb = f0 < T ? f0 : 0u;
          const uint4 ca = *reinterpret_cast<const uint4 *>(list + fb);
          const uint4 cb = *reinterpret_cast<const uint4 *>(list + fb + 4);
          const uint32_t cand[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
          const uint32_t mine = list[fb + sub];
          uint4 xv[8][CPL];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uint32_t row = f0 + u < T ? cand[u] : 0u;  // slots past the end gather row 0, dropped below
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              const int ch = sub + 8 * c;
              const int chc = FULL ? ch : min(ch, nchunk - 1);
              if (SHIFT) {
                const uint32_t off = (row << dshift) + 16u * (uint32_t)chc;
                xv[u][c] = *reinterpret_cast<const uint4 *>(ux + off);
              } else {
                xv[u][c] = *reinterpret_cast<const uint4 *>(ux + (size_t)row * dim + 16 * chc);
              }
              if (!FULL && ch >= nchunk) xv[u][c] = make_uint4(0, 0, 0, 0);
            }
          }
          uint32_t d[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            d[u] = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              d[u] = sad_u8(qv[c].x, xv[u][c].x, d[u]);
              d[u] = sad_u8(qv[c].y, xv[u][c].y, d[u]);
              d[u] = sad_u8(qv[c].z, xv[u][c].z, d[u]);
              d[u] = sad_u8(qv[c].w, xv[u][c].w, d[u]);
            }
          }
          // lanes i and 7 - i exchange halves (upper lanes keep candidates 4..7), then i and i ^ 2, then
          // i and i ^ 1: lane i ends with the whole sum of candidate i
          uint32_t e[4], f[2];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t keep = up4 ? d[j + 4] : d[j], send = up4 ? d[j] : d[j + 4];
            e[j] = keep + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0x141 /*row_half_mirror*/, 0xF, 0xF, true);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t keep = up2 ? e[j + 2] : e[j], send = up2 ? e[j] : e[j + 2];
            f[j] = keep + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true);
          }
          const uint32_t keep = up1 ? f[1] : f[0], send = up1 ? f[0] : f[1];
          const uint32_t dsum = keep + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true);
          const bool dead = !(f0 + (uint32_t)sub < T) || mine == (uint32_t)k1 || mine == (uint32_t)k2;
          const uint64_t k = dead ? kNoneD : (((uint64_t)dsum << 32) | mine);
          const uint64_t hi = key_max(k, k1);
          k1 = key_min(k, k1);
          k2 = key_min(hi, k2);
        }
      } else
      // rounds: one candidate row per group per round, RU rounds in flight
      for (uint32_t f0 = 0; __any(f0 < T); f0 += RU) {
        const uint4 c4 = *reinterpret_cast<const uint4 *>(list + (f0 < T ? f0 : 0u));
        const uint32_t cand[RU] = {c4.x, c4.y, c4.z, c4.w};
        uint4 xv[RU][CPL];
        bool live[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          live[u] = f0 + u < T;
          // the slots past the end of the list gather row 0 (unconditional loads: no exec-mask branch
          // per row; at most RU - 1 such rows per window, L2 hits) and are dropped at insertion
          const uint32_t row = live[u] ? cand[u] : 0u;
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            // FULL: the row fills all 8 * CPL chunks (dim 128, 256); otherwise the lanes past the row's
            // end re-read its last chunk and are zeroed (their query chunk is zero too)
            const int ch = sub + 8 * c;
            const int chc = FULL ? ch : min(ch, nchunk - 1);
            if (SHIFT) {
              const uint32_t off = (row << dshift) + 16u * (uint32_t)chc;
              xv[u][c] = *reinterpret_cast<const uint4 *>(ux + off);
            } else {
              xv[u][c] = *reinterpret_cast<const uint4 *>(ux + (size_t)row * dim + 16 * chc);
            }
            if (!FULL && ch >= nchunk) xv[u][c] = make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          uint32_t d = 0;
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            d = sad_u8(qv[c].x, xv[u][c].x, d);
            d = sad_u8(qv[c].y, xv[u][c].y, d);
            d = sad_u8(qv[c].z, xv[u][c].z, d);
            d = sad_u8(qv[c].w, xv[u][c].w, d);
          }
          d = group8_sum(d);
          // a row that already holds a place (an earlier table's pass, or a bucket probed twice)
          // would only repeat its own key
          const bool dead = !live[u] || cand[u] == (uint32_t)k1 || cand[u] == (uint32_t)k2;
          const uint64_t k = dead ? kNoneD : (((uint64_t)d << 32) | cand[u]);
          const uint64_t hi = key_max(k, k1);
          k1 = key_min(k, k1);
          k2 = key_min(hi, k2);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
  }

  if constexpr (RU == 8) {
    // the eight lanes' pairs -> the group's pair (every lane ends with it).  Equal keys are the same
    // row (the carried pair every lane started from, a row met again): counted once.
    merge_keys_with<0x141>(k1, k2);  // i <-> 7 - i
    merge_keys_with<0x4E>(k1, k2);   // i <-> i ^ 2
    merge_keys_with<0xB1>(k1, k2);   // i <-> i ^ 1
  }
  if (valid && sub == 0) {
    if (!last_pass) {
      partial[2 * (size_t)q] = k1;
      partial[2 * (size_t)q + 1] = k2;
      pvisited[q] = visited;
    } else {
      const bool n1 = k1 == kNoneD, n2 = k2 == kNoneD;
      out_idx[2 * (size_t)q + 0] = n1 ? ~0ull : (k1 & 0xFFFFFFFFull);
      out_idx[2 * (size_t)q + 1] = n2 ? ~0ull : (k2 & 0xFFFFFFFFull);
      out_dist[2 * (size_t)q + 0] = n1 ? 2147483648.0f : (float)(uint32_t)(k1 >> 32);
      out_dist[2 * (size_t)q + 1] = n2 ? 2147483648.0f : (float)(uint32_t)(k2 >> 32);
      if (out_ncand) out_ncand[q] = visited;
    }
  }
}

struct CascadeLayout {
  int mc, hb;
  size_t off_dictp, off_dictm, off_ux, off_uy, off_xcodes, off_ysign, off_ymask, off_bstart,
      off_order, off_ranks, off_segsum, off_qbstart, off_qorder, off_qranks, off_partial, off_pvisited, total;
};

CascadeLayout cascade_layout(int xrows, int yrows, int dim, int m, int n) {
  CascadeLayout L{};
  L.mc = (m + 3) / 4 * 4;  // accumulators per table: multiples of 4 (one ds_read_b128 each)
  L.hb = bucket_bits(m);
  const size_t nb1 = ((size_t)1 << L.hb) + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += round_up(std::max<size_t>(bytes, 16), 256);
    return o;
  };
  L.off_dictp = take((size_t)n * dim * L.mc * sizeof(float));
  L.off_dictm = take((size_t)(dim + 16) * 64 * sizeof(float));  // MFMA layout, at most 64 columns, rows padded to a multiple of 32
  L.off_ux = take((size_t)xrows * dim);
  L.off_uy = take((size_t)yrows * dim);
  L.off_xcodes = take((size_t)n * xrows * sizeof(uint32_t));
  L.off_ysign = take((size_t)n * yrows * sizeof(uint32_t));
  L.off_ymask = take((size_t)n * yrows * sizeof(uint32_t));
  L.off_bstart = take((size_t)2 * n * nb1 * sizeof(uint32_t));  // [n] database tables, then [n] query tables (one scan)
  L.off_order = take((size_t)n * xrows * sizeof(uint32_t));
  L.off_ranks = take((size_t)n * xrows * sizeof(uint32_t));
  L.off_segsum = take((size_t)2 * n * ((nb1 + kScanSeg - 1) / kScanSeg) * sizeof(uint32_t));
  // the probe's per-table query order (counting sort of the queries by sign code) and what a
  // query carries from one table's pass to the next
  L.off_qbstart = L.off_bstart + (size_t)n * nb1 * sizeof(uint32_t);
  L.off_qorder = take((size_t)n * yrows * sizeof(uint32_t));
  L.off_qranks = take((size_t)n * yrows * sizeof(uint32_t));
  L.off_partial = take((size_t)yrows * 2 * sizeof(uint64_t));
  L.off_pvisited = take((size_t)yrows * sizeof(int32_t));
  L.total = off;
  return L;
}

template <bool IS_QUERY>
void launch_project(int mc, int g, const float *rows, int nrows, int dim, int m, int n,
                    const float *dictp, uint32_t *codes, uint32_t *masks, uint8_t *img,
                    uint32_t *counts, uint32_t *ranks, uint32_t hbmask, int nb, hipStream_t stream) {
  if (nrows <= 0) return;
  const dim3 grid((nrows + kProjTile - 1) / kProjTile), block(kThreads);
  constexpr int G1 = IS_QUERY ? 4 : 1, G2 = IS_QUERY ? 16 : 1;
  // two tables per pass while the accumulators fit (MC <= 24), else one
#define SPV_LAUNCH_PROJECT(MCV, NTV)                                                           \
  if (g <= G1)                                                                                  \
    hipLaunchKernelGGL((project_kernel<MCV, NTV, IS_QUERY, G1>), grid, block, 0, stream, rows,  \
                       nrows, dim, m, n, g, dictp, codes, masks, img, counts, ranks, hbmask, nb); \
  else                                                                                          \
    hipLaunchKernelGGL((project_kernel<MCV, NTV, IS_QUERY, G2>), grid, block, 0, stream, rows,  \
                       nrows, dim, m, n, g, dictp, codes, masks, img, counts, ranks, hbmask, nb);
  // two tables per pass while 2 x 2 x MC accumulators fit the register budget (MC <= 24)
  // Two tables per pass (rows read from HBM once) with one row per lane: 0.47 ms for 1M + 1M rows;
  // measured alternatives: one table per pass 0.54 ms; two rows per lane 0.58 (one table) / 0.65
  // (two tables) -- its 77 KB of LDS per workgroup halve the occupancy.
  // SPECTAVI_CASCADE_NT=1 selects one table per pass.
  static const bool nt1_env = [] {
    const char *e = getenv("SPECTAVI_CASCADE_NT");
    return e && e[0] == '1';
  }();
  const bool two = n >= 2 && !nt1_env;
#define SPV_PROJECT_CASE(MCV)                                             \
  case MCV:                                                               \
    if (two && MCV <= 24) {                                               \
      SPV_LAUNCH_PROJECT(MCV, (MCV <= 24 ? 2 : 1))                        \
    } else {                                                              \
      SPV_LAUNCH_PROJECT(MCV, 1)                                          \
    }                                                                     \
    break;
  switch (mc) {
    SPV_PROJECT_CASE(4)
    SPV_PROJECT_CASE(8)
    SPV_PROJECT_CASE(12)
    SPV_PROJECT_CASE(16)
    SPV_PROJECT_CASE(20)
    SPV_PROJECT_CASE(24)
    SPV_PROJECT_CASE(28)
    default:
      SPV_LAUNCH_PROJECT(32, 1)
      break;
  }
#undef SPV_PROJECT_CASE
#undef SPV_LAUNCH_PROJECT
}

// Matrix-core projection: applies when the n*m hyperplanes fit 64 columns.
inline bool project_mfma_applies(int m, int n) {
  static const bool off = [] {
    const char *e = getenv("SPECTAVI_CASCADE_MFMA");
    return e && e[0] == '0';
  }();
  return !off && (long long)n * m <= 64;
}

template <bool IS_QUERY>
void launch_project_mfma(int g, const float *rows, int nrows, int dim, int m, int n, const float *dictm,
                         uint32_t *codes, uint32_t *masks, uint8_t *img, uint32_t *counts,
                         uint32_t *ranks, uint32_t hbmask, int nb, hipStream_t stream) {
  if (nrows <= 0) return;
  const int ct = (n * m + 15) / 16;
  const dim3 grid((nrows + kThreads - 1) / kThreads), block(kThreads);
  constexpr int G0 = IS_QUERY ? 2 : 1, G1 = IS_QUERY ? 4 : 1, G2 = IS_QUERY ? 16 : 1;
  const bool full = dim % kMfmaChunk == 0;
  // 1..8 columns beyond whole 16-column tiles (the default 2 x 17): left-over columns on 4x4x1 MFMAs
  {
    const int nm = n * m, ctm = nm / 16, left = nm % 16, ng = (left + 3) / 4;
    static const bool off4 = [] {
      const char *e = getenv("SPECTAVI_CASCADE_MFMA4");
      return e && e[0] == '0';
    }();
    if (!off4 && full && dim <= 512 && left >= 1 && left <= 8) {
#define SPV_LAUNCH_M4(CTV, NGV)                                                                        \
  if (g <= G0)                                                                                         \
    hipLaunchKernelGGL((project_mfma4_kernel<CTV, NGV, IS_QUERY, G0>), grid, block,                    \
                       (size_t)(NGV) * dim * 16, stream, rows,                                         \
                       nrows, dim, m, n, g, dictm, codes, masks, img, counts, ranks, hbmask, nb);      \
  else                                                                                                 \
    hipLaunchKernelGGL((project_mfma4_kernel<CTV, NGV, IS_QUERY, G2>), grid, block,                    \
                       (size_t)(NGV) * dim * 16, stream, rows,                                         \
                       nrows, dim, m, n, g, dictm, codes, masks, img, counts, ranks, hbmask, nb)
      switch (ctm * 2 + (ng - 1)) {
        case 0: SPV_LAUNCH_M4(0, 1); break;
        case 1: SPV_LAUNCH_M4(0, 2); break;
        case 2: SPV_LAUNCH_M4(1, 1); break;
        case 3: SPV_LAUNCH_M4(1, 2); break;
        case 4: SPV_LAUNCH_M4(2, 1); break;
        case 5: SPV_LAUNCH_M4(2, 2); break;
        case 6: SPV_LAUNCH_M4(3, 1); break;
        default: SPV_LAUNCH_M4(3, 2); break;
      }
#undef SPV_LAUNCH_M4
      return;
    }
  }
#define SPV_LAUNCH_ONE(CTV, GV, FULLV)                                                               \
  hipLaunchKernelGGL((project_mfma_kernel<CTV, IS_QUERY, GV, FULLV>), grid, block, 0, stream, rows,  \
                     nrows, dim, m, n, g, dictm, codes, masks, img, counts, ranks, hbmask, nb)
#define SPV_LAUNCH_MFMA(CTV)                                                                         \
  case CTV:                                                                                          \
    if (g <= G0) {                                                                                   \
      if (full) SPV_LAUNCH_ONE(CTV, G0, true);                                                       \
      else SPV_LAUNCH_ONE(CTV, G0, false);                                                           \
    } else if (g <= G1) {                                                                            \
      if (full) SPV_LAUNCH_ONE(CTV, G1, true);                                                       \
      else SPV_LAUNCH_ONE(CTV, G1, false);                                                           \
    } else {                                                                                         \
      if (full) SPV_LAUNCH_ONE(CTV, G2, true);                                                       \
      else SPV_LAUNCH_ONE(CTV, G2, false);                                                           \
    }                                                                                                \
    break;
  switch (ct) {
    SPV_LAUNCH_MFMA(1)
    SPV_LAUNCH_MFMA(2)
    SPV_LAUNCH_MFMA(3)
    default:
      SPV_LAUNCH_MFMA(4)
  }
#undef SPV_LAUNCH_ONE
#undef SPV_LAUNCH_MFMA
}

}  // namespace

size_t cascade_workspace_bytes(int xrows, int yrows, int dim, int m, int n, int g) {
  (void)g;
  return cascade_layout(xrows, yrows, dim, m, n).total;
}

int cascade_run(const float *d_x, const float *d_y, int xrows, int yrows, int dim, int m, int n,
                int g, const float *d_dict, uint64_t *d_idx, float *d_dist, int32_t *d_ncand,
                void *d_ws, size_t ws_bytes, hipStream_t stream) {
  if (yrows == 0) return SPV_OK;
  if (!d_y || !d_dict || !d_idx || !d_dist || (xrows > 0 && !d_x))
    return set_error(SPV_ERR_INVALID, "null device pointer");
  if (g > 16) return set_error(SPV_ERR_INVALID, "num_candidate_neighbours g=%d > 16", g);
  // before anything is enqueued: rows wider than the refine kernels take, misaligned bases
  const int cpl = (dim / 16 + 7) / 8;
  if (cpl > 16)
    return set_error(SPV_ERR_INVALID, "dim=%d > 2048 is not supported by the cascade refine kernel",
                     dim);
  if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y) |
       reinterpret_cast<uintptr_t>(d_dict) | reinterpret_cast<uintptr_t>(d_ws)) & 15)
    return set_error(SPV_ERR_INVALID, "device pointers must be 16-byte aligned (x %p, y %p, dict %p, ws %p)",
                     (const void *)d_x, (const void *)d_y, (const void *)d_dict, d_ws);
  if ((reinterpret_cast<uintptr_t>(d_idx) & 7) || (reinterpret_cast<uintptr_t>(d_dist) & 3) ||
      (reinterpret_cast<uintptr_t>(d_ncand) & 3))
    return set_error(SPV_ERR_INVALID, "output pointers must be aligned to their element size");
  const CascadeLayout L = cascade_layout(xrows, yrows, dim, m, n);
  if (!d_ws || ws_bytes < L.total)
    return set_error(SPV_ERR_INVALID, "workspace too small: %zu < %zu", ws_bytes, L.total);
  uint8_t *ws = static_cast<uint8_t *>(d_ws);
  float *dictp = reinterpret_cast<float *>(ws + L.off_dictp);
  uint8_t *ux = ws + L.off_ux;
  uint8_t *uy = ws + L.off_uy;
  uint32_t *xcodes = reinterpret_cast<uint32_t *>(ws + L.off_xcodes);
  uint32_t *ysign = reinterpret_cast<uint32_t *>(ws + L.off_ysign);
  uint32_t *ymask = reinterpret_cast<uint32_t *>(ws + L.off_ymask);
  uint32_t *bstart = reinterpret_cast<uint32_t *>(ws + L.off_bstart);
  uint32_t *order = reinterpret_cast<uint32_t *>(ws + L.off_order);
  uint32_t *ranks = reinterpret_cast<uint32_t *>(ws + L.off_ranks);
  const int nb = 1 << L.hb;
  const uint32_t hbmask = (uint32_t)nb - 1;

  // The probe walks the queries table by table in the order of that table's sign code when the
  // group kernel applies and the input is large enough for the order to matter (the per-table
  // counting sorts and passes cost a dozen small launches); SPECTAVI_CASCADE_SORT=0 / 1 force it
  // off / on (A/B runs, and the fuzz run covers both forms: same results).
  const int sort_env = [] {  // read per call: tests switch it inside one process
    const char *e = getenv("SPECTAVI_CASCADE_SORT");
    return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1;
  }();
  static const bool group_env = [] {
    const char *e = getenv("SPECTAVI_CASCADE_GROUP");
    return !(e && e[0] == '0');
  }();
  const bool use_group = group_env && m <= L.hb && cpl <= 2;
  const bool sorted = use_group && xrows > 0 &&
                      (sort_env >= 0 ? sort_env == 1 : (n <= 8 && (long long)yrows >= 65536));
  uint32_t *qbstart = reinterpret_cast<uint32_t *>(ws + L.off_qbstart);
  uint32_t *qorder = reinterpret_cast<uint32_t *>(ws + L.off_qorder);
  uint32_t *qranks = reinterpret_cast<uint32_t *>(ws + L.off_qranks);

  SPV_HIP_CHECK(hipMemsetAsync(bstart, 0, (size_t)(sorted ? 2 : 1) * n * (nb + 1) * sizeof(uint32_t), stream));
  // query histogram + ranks: fused into the matrix-core projection's epilogue (+0.05 ms on the query
  // pass at 1M rows: every wave ends on the round trip of its returning atomics) or, with
  // SPECTAVI_CASCADE_QHIST=0 and always on the VALU path, a kernel of their own after it (+0.10 ms)
  static const bool qhist_fused = [] {
    const char *e = getenv("SPECTAVI_CASCADE_QHIST");
    return !(e && e[0] == '0');
  }();
  uint32_t *qcounts = sorted && qhist_fused ? qbstart : nullptr;
  uint32_t *qrk = sorted && qhist_fused ? qranks : nullptr;
  {
  ProfScope prof("cascade_project", stream);
  if (project_mfma_applies(m, n)) {
    float *dictm = reinterpret_cast<float *>(ws + L.off_dictm);
    const int nc = (n * m + 15) / 16 * 16;
    hipLaunchKernelGGL(repack_dict_mfma_kernel, dim3(64), dim3(kThreads), 0, stream, d_dict, dictm, n, dim,
                       m, nc);
    launch_project_mfma<false>(0, d_x, xrows, dim, m, n, dictm, xcodes, nullptr, ux, bstart, ranks, hbmask,
                               nb, stream);
    launch_project_mfma<true>(g, d_y, yrows, dim, m, n, dictm, ysign, ymask, uy, qcounts, qrk, hbmask,
                              nb, stream);
    if (sorted && !qhist_fused)
      hipLaunchKernelGGL(query_rank_kernel, dim3(1024), dim3(256), 0, stream, ysign, yrows, n, hbmask, nb, qbstart,
                         qranks);
  } else {
    hipLaunchKernelGGL(repack_dict_kernel, dim3(64), dim3(kThreads), 0, stream, d_dict, dictp, n, dim,
                       m, L.mc);
    launch_project<false>(L.mc, 0, d_x, xrows, dim, m, n, dictp, xcodes, nullptr, ux, bstart, ranks, hbmask,
                          nb, stream);
    launch_project<true>(L.mc, g, d_y, yrows, dim, m, n, dictp, ysign, ymask, uy, nullptr, nullptr, hbmask,
                         nb, stream);
    if (sorted)
      hipLaunchKernelGGL(query_rank_kernel, dim3(1024), dim3(256), 0, stream, ysign, yrows, n, hbmask, nb, qbstart,
                         qranks);
  }
  }
  SPV_HIP_CHECK(hipGetLastError());

  {
  ProfScope prof("cascade_buckets", stream);
  {
    // database and (sorted probe) query histograms sit back to back: one scan over 2n "tables"
    const int nseg = (nb + 1 + kScanSeg - 1) / kScanSeg;
    const int ntab = sorted ? 2 * n : n;
    uint32_t *segsum = reinterpret_cast<uint32_t *>(ws + L.off_segsum);
    hipLaunchKernelGGL(bucket_segsum_kernel, dim3(nseg, ntab), dim3(256), 0, stream, bstart, nb, nseg, segsum);
    hipLaunchKernelGGL(bucket_segscan_kernel, dim3(ntab), dim3(1024), 0, stream, segsum, nseg);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(nseg, ntab), dim3(1024), 0, stream, bstart, nb, nseg, segsum);
  }
  if (xrows > 0)
    hipLaunchKernelGGL(bucket_fill_kernel, dim3(2048), dim3(kThreads), 0, stream, xcodes, ranks, xrows, n,
                       hbmask, nb, bstart, order);
  if (sorted)  // the queries' own order, per table, by sign code
    hipLaunchKernelGGL(bucket_fill_kernel, dim3(2048), dim3(kThreads), 0, stream, ysign, qranks, yrows, n,
                       hbmask, nb, qbstart, qorder);
  }
  SPV_HIP_CHECK(hipGetLastError());

  const dim3 grid((yrows + kThreads / 64 - 1) / (kThreads / 64)), block(kThreads);
  ProfScope prof_probe("cascade_probe_refine", stream);
  // group-per-query kernel unless the full-code check is needed (m > bucket bits) or rows
  // are wider than 256 bytes; otherwise the wave-per-query kernel
  if (use_group) {
    const int nblk = (yrows + kThreads / 8 - 1) / (kThreads / 8);
    uint64_t *partial = reinterpret_cast<uint64_t *>(ws + L.off_partial);
    int32_t *pvisited = reinterpret_cast<int32_t *>(ws + L.off_pvisited);
    const int per_xcd = sorted ? (nblk + 7) / 8 : 0;
    const dim3 ggrid(sorted ? 8 * per_xcd : nblk);
    {
      int dshift = 0;
      while ((1 << dshift) < dim) ++dshift;
      const bool shift = (1 << dshift) == dim && (unsigned long long)xrows * (unsigned)dim < (1ull << 32);
      const int nb1 = nb + 1;
      for (int ps = 0; ps < n; ++ps) {
        const uint32_t *qo = sorted ? qorder + (size_t)ps * yrows : nullptr;
#define SPV_RU(...) SPV_RU_(__VA_ARGS__ __VA_OPT__(,) 4)
#define SPV_RU_(A, ...) A
#define SPV_LAUNCH_LEAN(C, W, S, F, ...)                                                                              \
  hipLaunchKernelGGL((probe_table_kernel<C, SPV_RU(__VA_ARGS__), W, S, F>), ggrid, block, 0, stream, ux, uy, xrows, yrows, dim, dshift, \
                     g, L.hb, ysign + (size_t)ps * yrows, ymask + (size_t)ps * yrows, bstart + (size_t)ps * nb1,   \
                     order + (size_t)ps * xrows, qo, nblk, per_xcd, partial, pvisited, ps == 0, ps == n - 1,       \
                     d_idx, d_dist, d_ncand)
        if (cpl == 1) {
          // waves per SIMD: 59 VGPRs and 18 KB of LDS per workgroup = eight waves.  (A first build
          // wanted 88 VGPRs -- the list fill's unrolled lane constants, see the kernel -- and measured
          // 1.11 ms per 1M queries at five unspilled waves, 1.21-1.24 spilling at 6-8:
          // profiles/r03_cascade_variants.txt)
          // rows of up to 128 bytes: eight rows in flight per group, lane s owning candidate s after the
          // transposing reduce (69 VGPRs, seven waves): 0.90 ms per 1M queries against 0.98 for four rows
          // in flight with every lane reducing every candidate (59 VGPRs, eight waves);
          // SPECTAVI_CASCADE_RU8=0 selects the latter (A/B runs)
          const bool ru4_env = [] {  // read per call: the tests run both forms in one process
            const char *e = getenv("SPECTAVI_CASCADE_RU8");
            return e && e[0] == '0';
          }();
          if (ru4_env) {
            if (shift && dim == 128)
              SPV_LAUNCH_LEAN(1, 8, true, true);
            else if (shift)
              SPV_LAUNCH_LEAN(1, 8, true, false);
            else
              SPV_LAUNCH_LEAN(1, 8, false, false);
          } else if (shift && dim == 128)
            SPV_LAUNCH_LEAN(1, 7, true, true, 8);     // SIFT-128, the benchmark's shape
          else if (shift)
            SPV_LAUNCH_LEAN(1, 7, true, false, 8);    // dim 16, 32, 64
          else
            SPV_LAUNCH_LEAN(1, 7, false, false, 8);   // dim 48, 80, 96, 112, or an image of 4 GiB and more
        } else {
          SPV_LAUNCH_LEAN(2, 6, false, false);     // dim 144 .. 256 (80 VGPRs)
        }
#undef SPV_LAUNCH_LEAN
#undef SPV_RU
      }
      SPV_HIP_CHECK(hipGetLastError());
      return SPV_OK;
    }
  }
  const dim3 pgrid(grid.x);
#define SPV_LAUNCH_PROBE(C, U)                                                                       \
  hipLaunchKernelGGL((probe_refine_kernel<C, U>), pgrid, block, 0, stream, ux, uy, xrows, yrows, dim, \
                     m, n, g, L.hb, xcodes, ysign, ymask, bstart, order, d_idx, d_dist,             \
                     d_ncand)
  static const int ru_env = [] {
    const char *e = getenv("SPECTAVI_CASCADE_RU");
    return e ? atoi(e) : 0;
  }();
  if (cpl == 1) {
    if (ru_env == 2)
      SPV_LAUNCH_PROBE(1, 2);
    else if (ru_env == 8)
      SPV_LAUNCH_PROBE(1, 8);
    else
      SPV_LAUNCH_PROBE(1, 4);
  } else if (cpl == 2) {
    SPV_LAUNCH_PROBE(2, 4);
  } else if (cpl <= 4) {
    SPV_LAUNCH_PROBE(4, 2);
  } else if (cpl <= 8) {
    SPV_LAUNCH_PROBE(8, 1);   // rows up to 1024 bytes
  } else {
    SPV_LAUNCH_PROBE(16, 1);  // rows up to 2048 bytes, the widest the L1 kernels take as well
  }
#undef SPV_LAUNCH_PROBE
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv
