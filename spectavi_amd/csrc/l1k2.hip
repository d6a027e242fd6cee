// l1k2.hip -- exact all-pairs L1 (sum of absolute differences) 2-nearest-neighbour on
// uint8 descriptors for gfx950 (MI355X).
//
// Replaces the loop nest of the reference's BruteForceNnL1K2::find_neighbours
// <IdentityFilter> (reference src/BruteForceNnL1K2.h:84-145; SAD primitive
// sad_16 at :43-48).  This is a new design, not a translation: the reference
// keeps one query row resident and streams database rows past it with an
// early-exit prune; here
//
//   * every lane owns Q query rows entirely in VGPRs (Q*dim/4 dwords),
//   * a workgroup streams its database slice through LDS in 64-row tiles filled
//     by coalesced 16-byte global loads (one descriptor = one 128-byte line),
//   * all lanes of a wave read the SAME database row from LDS (broadcast
//     ds_read_b128), so the LDS cost per row is amortised over 64*Q pairs,
//   * the distance is accumulated by v_sad_hi_u8, which adds SAD<<16 into an
//     accumulator pre-loaded with the tile-local row index: the accumulator IS
//     the packed sort key (dist<<16 | idx16), no pack instruction needed,
//   * the running two smallest keys per query: one compare per pair against the
//     current second best and a wave-uniform branch; only when some lane improves
//     (rare after the first few hundred rows of a slice) v_min_u32 + v_med3_u32,
//   * no early-exit prune: it is result-neutral in the reference
//     (src/BruteForceNnL1K2.h:118-121 only skips pairs that could not be
//     inserted) and would diverge the wave.
//
// Keys are unique per database row, so "two smallest keys" is exactly the
// reference's streaming strict-< update visited in ascending row order
// (src/BruteForceNnL1K2.h:129-139): lexicographic (dist, idx).
//
// The database is cut into slices of <= 65536 rows (16-bit local index); a
// second kernel merges the per-slice partial top-2 keys of each query with a
// wavefront-shuffle argmin-2 butterfly and writes the ABI layout
// (size_t idx[N,2], int dist[N,2]; sentinels INT_MAX / (size_t)-1 as
// src/BruteForceNnL1K2.h:100-103).
//
// Roofline: sum-of-abs-diff is not a contraction (no MFMA); the bound is the issue
// rate of v_sad_hi_u8 -- one wave64 instruction per 4 cycles per SIMD, measured -- at
// 32 lane-ops per 128-D pair.  The kernel runs at 0.90 of that peak.  See DESIGN.md.

#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace spv {
namespace {

constexpr int kThreads = 256;
constexpr int kTileRows = 64;               // database rows per LDS tile
constexpr uint32_t kKeyNone = 0xFFFFFFFFu;  // > any real key (dist <= 65280)
constexpr uint64_t kKey64None = ~0ull;
constexpr int kMaxGenericDim = 2048;        // wide-row kernel: 16 rows x dim bytes of LDS

__device__ __forceinline__ uint32_t sad_hi(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_sad_hi_u8(a, b, c);  // (SAD_U8(a,b) << 16) + c
}

// Insert key k into the sorted pair (k1 <= k2).  min + med3.
__device__ __forceinline__ void top2_insert(uint32_t &k1, uint32_t &k2, uint32_t k) {
  k2 = max(min(k1, k), min(max(k1, k), k2));  // median of (k1, k, k2) -> v_med3_u32
  k1 = min(k1, k);
}

template <int V4>
__device__ __forceinline__ void lds_row(uint4 (&dst)[V4], const uint4 *row) {
#pragma unroll
  for (int c = 0; c < V4; ++c) dst[c] = row[c];
}

template <int D4, int Q>
__device__ __forceinline__ void row_update(const uint32_t (&qreg)[Q][D4], const uint4 (&xr)[D4 / 4],
                                           uint32_t j, uint32_t (&k1)[Q], uint32_t (&k2)[Q]) {
  // Q independent accumulator chains, interleaved so consecutive v_sad_hi_u8
  // never depend on each other.
  uint32_t acc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = j;
#pragma unroll
  for (int c = 0; c < D4 / 4; ++c) {
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][4 * c + 0], xr[c].x, acc[q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][4 * c + 1], xr[c].y, acc[q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][4 * c + 2], xr[c].z, acc[q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][4 * c + 3], xr[c].w, acc[q]);
  }
  // Lazy top-2: a new key enters only if it beats the current second best of
  // its query.  After the first few hundred rows of a slice that is rare, so the
  // common case is Q compares and one wave-uniform branch instead of Q x
  // (v_min + v_med3).  Result-identical to the eager update.
  bool any = false;
#pragma unroll
  for (int q = 0; q < Q; ++q) any |= acc[q] < k2[q];
  if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
    for (int q = 0; q < Q; ++q) top2_insert(k1[q], k2[q], acc[q]);
  }
}

// Chunked-row form for wide descriptors: the row is consumed as NCH chunks of CH4 = D4/NCH
// dwords (two for dim 192, four for dim 256) so that only one chunk-sized buffer pair is live
// next to the 2 x D4 query registers (two queries per lane keep the LDS broadcast amortised)
// and the kernel keeps three waves per SIMD.
template <int D4, int Q, int NCH, int C>
__device__ __forceinline__ void chunk_accumulate(const uint32_t (&qreg)[Q][D4],
                                                 const uint4 (&xc)[D4 / NCH / 4], uint32_t (&acc)[Q]) {
  constexpr int CH4 = D4 / NCH;
#pragma unroll
  for (int c = 0; c < CH4 / 4; ++c) {
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][C * CH4 + 4 * c + 0], xc[c].x, acc[q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][C * CH4 + 4 * c + 1], xc[c].y, acc[q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][C * CH4 + 4 * c + 2], xc[c].z, acc[q]);
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][C * CH4 + 4 * c + 3], xc[c].w, acc[q]);
  }
}

template <int Q>
__device__ __forceinline__ void lazy_top2(const uint32_t (&acc)[Q], uint32_t (&k1)[Q], uint32_t (&k2)[Q]) {
  bool any = false;
#pragma unroll
  for (int q = 0; q < Q; ++q) any |= acc[q] < k2[q];
  if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
    for (int q = 0; q < Q; ++q) top2_insert(k1[q], k2[q], acc[q]);
  }
}

// Partial-key layout: part[(query * S + slice) * 2 + {0,1}], key = dist<<32 | global idx.
__device__ __forceinline__ uint64_t widen_key(uint32_t k, uint32_t slice_base) {
  if (k == kKeyNone) return kKey64None;
  return ((uint64_t)(k >> 16) << 32) | (uint64_t)(slice_base + (k & 0xFFFFu));
}

// ---------------------------------------------------------------------------------
// Tile kernel.  grid = (query blocks, slices); block = 256 threads = 4 waves.
// Thread t of query block qb owns queries qb*256*Q + q*256 + t, q = 0..Q-1.
// ---------------------------------------------------------------------------------
template <int D4, int Q>
__global__ __launch_bounds__(kThreads, 3) void l1k2_tile_kernel(
    const uint4 *__restrict__ x, const uint4 *__restrict__ y, int M, int N, int slice_rows,
    int S, uint64_t *__restrict__ part) {
  constexpr int V4 = D4 / 4;                                   // 16-byte vectors per row
  constexpr int TILE_V4 = kTileRows * V4;                      // vectors per tile
  constexpr int NL = (TILE_V4 + kThreads - 1) / kThreads;      // staging loads per thread
  __shared__ uint4 tile[2][TILE_V4];

  const int t = threadIdx.x;
  const int qb = blockIdx.x;
  const int s = blockIdx.y;
  const int row_begin = s * slice_rows;
  const int row_end = min(M, row_begin + slice_rows);

  // ---- this lane's queries -> registers
  uint32_t qreg[Q][D4];
  int qi[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    qi[q] = qb * (kThreads * Q) + q * kThreads + t;
    const int src = min(qi[q], N - 1);
    const uint4 *yr = y + (size_t)src * V4;
#pragma unroll
    for (int c = 0; c < V4; ++c) {
      const uint4 v = yr[c];
      qreg[q][4 * c + 0] = v.x;
      qreg[q][4 * c + 1] = v.y;
      qreg[q][4 * c + 2] = v.z;
      qreg[q][4 * c + 3] = v.w;
    }
  }
  uint32_t k1[Q], k2[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) k1[q] = k2[q] = kKeyNone;

  // ---- database slice through LDS, register-prefetched double buffer
  uint4 stage[NL];
  auto stage_load = [&](int row0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = t + i * kThreads;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < TILE_V4 && row0 + e / V4 < row_end) v = x[(size_t)row0 * V4 + e];
      stage[i] = v;
    }
  };
  auto stage_store = [&](uint4 *dst) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = t + i * kThreads;
      if (e < TILE_V4) dst[e] = stage[i];
    }
  };

  const int ntiles = (row_end - row_begin + kTileRows - 1) / kTileRows;
  if (ntiles > 0) {
    stage_load(row_begin);
    stage_store(tile[0]);
  }
  __syncthreads();

  for (int tl = 0; tl < ntiles; ++tl) {
    const int row0 = row_begin + tl * kTileRows;
    const bool has_next = tl + 1 < ntiles;
    if (has_next) stage_load(row0 + kTileRows);

    const int nrows = min(kTileRows, row_end - row0);
    const uint4 *buf = tile[tl & 1];
    const uint32_t jbase = (uint32_t)(row0 - row_begin);

    if constexpr (D4 >= 40) {
      // wide rows: NCH chunks per row, the next chunk's LDS reads issued before the current
      // chunk's SAD chain (xa / xb alternate; the last chunk prefetches the next row's first)
      constexpr int NCH = D4 >= 64 ? 4 : 2;
      constexpr int CV = V4 / NCH;
      static_assert(V4 % NCH == 0, "the row must split into NCH chunks of whole 16-byte vectors");
      uint4 xa[CV], xb[CV];
      lds_row<CV>(xa, buf);
      for (int r = 0; r < nrows; ++r) {
        uint32_t acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = jbase + r;
        const uint4 *row = buf + r * V4;
        const uint4 *nxt = buf + min(r + 1, kTileRows - 1) * V4;
        lds_row<CV>(xb, row + CV);
        chunk_accumulate<D4, Q, NCH, 0>(qreg, xa, acc);
        if constexpr (NCH == 2) {
          lds_row<CV>(xa, nxt);
          chunk_accumulate<D4, Q, NCH, 1>(qreg, xb, acc);
        } else {
          lds_row<CV>(xa, row + 2 * CV);
          chunk_accumulate<D4, Q, NCH, 1>(qreg, xb, acc);
          lds_row<CV>(xb, row + 3 * CV);
          chunk_accumulate<D4, Q, NCH, 2>(qreg, xa, acc);
          lds_row<CV>(xa, nxt);
          chunk_accumulate<D4, Q, NCH, 3>(qreg, xb, acc);
        }
        lazy_top2<Q>(acc, k1, k2);
      }
    } else {
      // two rows per iteration, next row's LDS reads issued before the current
      // row's SAD chain so the broadcast reads hide behind VALU work
      uint4 xa[V4], xb[V4];
      lds_row<V4>(xa, buf);
      for (int r = 0; r < nrows; r += 2) {
        lds_row<V4>(xb, buf + min(r + 1, kTileRows - 1) * V4);
        row_update<D4, Q>(qreg, xa, jbase + r, k1, k2);
        lds_row<V4>(xa, buf + min(r + 2, kTileRows - 1) * V4);
        if (r + 1 < nrows) row_update<D4, Q>(qreg, xb, jbase + r + 1, k1, k2);
      }
    }

    if (has_next) stage_store(tile[(tl + 1) & 1]);
    __syncthreads();
  }

#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (qi[q] < N) {
      uint64_t *dst = part + ((size_t)qi[q] * S + s) * 2;
      dst[0] = widen_key(k1[q], (uint32_t)row_begin);
      dst[1] = widen_key(k2[q], (uint32_t)row_begin);
    }
  }
}

// ---------------------------------------------------------------------------------
// Wide rows (256 < dim <= 2048, any multiple of 16 bytes): a row no longer fits the
// register file next to a second one, so the roles of the tile kernel are kept but the row is
// consumed in 128-byte chunks.  A workgroup stages kWideRows database rows (full width) in LDS;
// for every chunk each lane fetches that chunk of its Q query rows (one 128-byte line per query,
// L2 resident: the queries of a workgroup are re-read once per database tile), reads the tile's
// rows by broadcast ds_read_b128 exactly as the tile kernel does, and adds the chunk's SADs into
// kWideRows x Q accumulators.  After the last chunk the tile's rows enter the running top-2 in
// ascending row order (strict <, as src/BruteForceNnL1K2.h:129-139).  Distances need more than
// 16 bits here (2048 x 255), so keys are (32-bit distance, 32-bit index) pairs and the partial
// result is the merge kernel's 64-bit key.
// ---------------------------------------------------------------------------------
constexpr int kWideRows = 16;
constexpr int kWideChunk4 = 32;  // dwords per chunk

template <int Q>
__global__ __launch_bounds__(kThreads, 2) void l1k2_wide_kernel(const uint4 *__restrict__ x,
                                                                const uint4 *__restrict__ y, int M, int N,
                                                                int D4, int slice_rows, int S,
                                                                uint64_t *__restrict__ part) {
  extern __shared__ uint4 wtile[];  // [kWideRows][D4 / 4]
  const int V4 = D4 / 4;
  const int t = threadIdx.x;
  // XCD-aware block -> (query block, slice): every database tile makes a workgroup re-read its
  // queries' chunks, 256 Q rows against 16 database rows, so the query block must stay in the L2 of
  // the XCD that runs it.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8): all slices
  // of query block qb are given ids = qb (mod 8), so they run on one XCD, one after the other, and an
  // XCD works on ~3 query blocks (<= 1 MB at dim 2048) at a time.  (Speed only: nothing depends on the
  // placement.)  With the plain (query block, slice) grid 256k x 256k x 512 was bound by those re-reads
  // at 0.61 of the SAD peak.
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int s = rest % S;
  const int qb = (rest / S) * 8 + xcd;
  if ((long long)qb * (kThreads * Q) >= N) return;  // padding of the query blocks to a multiple of 8
  const int row_begin = s * slice_rows;
  const int row_end = min(M, row_begin + slice_rows);
  int qi[Q];
  const uint4 *yq[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    qi[q] = qb * (kThreads * Q) + q * kThreads + t;
    yq[q] = y + (size_t)min(qi[q], N - 1) * V4;
  }
  uint32_t d1[Q], d2[Q], i1[Q], i2[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    d1[q] = d2[q] = 0xFFFFFFFFu;
    i1[q] = i2[q] = 0xFFFFFFFFu;
  }
  for (int row0 = row_begin; row0 < row_end; row0 += kWideRows) {
    const int nrows = min(kWideRows, row_end - row0);
    __syncthreads();  // the previous tile has been consumed
    for (int e = t; e < kWideRows * V4; e += kThreads) {
      const int r = e / V4;
      wtile[e] = r < nrows ? x[(size_t)(row0 + r) * V4 + (e - r * V4)] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    uint32_t acc[kWideRows][Q];
#pragma unroll
    for (int r = 0; r < kWideRows; ++r)
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[r][q] = 0;
    const int V4full = V4 / (kWideChunk4 / 4) * (kWideChunk4 / 4);
    for (int c4 = 0; c4 < V4full; c4 += kWideChunk4 / 4) {
      uint32_t qreg[Q][kWideChunk4];
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int v = 0; v < kWideChunk4 / 4; ++v) {
          const uint4 w = yq[q][c4 + v];
          qreg[q][4 * v + 0] = w.x;
          qreg[q][4 * v + 1] = w.y;
          qreg[q][4 * v + 2] = w.z;
          qreg[q][4 * v + 3] = w.w;
        }
#pragma unroll
      for (int r = 0; r < kWideRows; ++r) {
        const uint4 *row = wtile + r * V4 + c4;
#pragma unroll
        for (int v = 0; v < kWideChunk4 / 4; ++v) {
          const uint4 xr = row[v];
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[r][q] = __builtin_amdgcn_sad_u8(qreg[q][4 * v + 0], xr.x, acc[r][q]);
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[r][q] = __builtin_amdgcn_sad_u8(qreg[q][4 * v + 1], xr.y, acc[r][q]);
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[r][q] = __builtin_amdgcn_sad_u8(qreg[q][4 * v + 2], xr.z, acc[r][q]);
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[r][q] = __builtin_amdgcn_sad_u8(qreg[q][4 * v + 3], xr.w, acc[r][q]);
        }
      }
    }
    // ragged end of the row (dim is a multiple of 16, not of 128): one 16-byte group at a time, same
    // roles -- round 3; before, such rows were zero padded to whole chunks (dim 272 paid for 384)
    for (int v4 = V4full; v4 < V4; ++v4) {
      uint4 qw[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) qw[q] = yq[q][v4];
#pragma unroll
      for (int r = 0; r < kWideRows; ++r) {
        const uint4 xr = wtile[r * V4 + v4];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          acc[r][q] = __builtin_amdgcn_sad_u8(qw[q].x, xr.x, acc[r][q]);
          acc[r][q] = __builtin_amdgcn_sad_u8(qw[q].y, xr.y, acc[r][q]);
          acc[r][q] = __builtin_amdgcn_sad_u8(qw[q].z, xr.z, acc[r][q]);
          acc[r][q] = __builtin_amdgcn_sad_u8(qw[q].w, xr.w, acc[r][q]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kWideRows; ++r) {
      if (r < nrows) {  // workgroup-uniform
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const uint32_t d = acc[r][q], idx = (uint32_t)(row0 + r);
          const bool lt2 = d < d2[q], lt1 = d < d1[q];
          d2[q] = lt1 ? d1[q] : (lt2 ? d : d2[q]);
          i2[q] = lt1 ? i1[q] : (lt2 ? idx : i2[q]);
          d1[q] = lt1 ? d : d1[q];
          i1[q] = lt1 ? idx : i1[q];
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (qi[q] < N) {
      uint64_t *dst = part + ((size_t)qi[q] * S + s) * 2;
      dst[0] = i1[q] == 0xFFFFFFFFu ? kKey64None : (((uint64_t)d1[q] << 32) | i1[q]);
      dst[1] = i2[q] == 0xFFFFFFFFu ? kKey64None : (((uint64_t)d2[q] << 32) | i2[q]);
    }
  }
}

// ---------------------------------------------------------------------------------
// Merge kernel: LPQ lanes per query stride over the S partial pairs, then an
// argmin-2 butterfly over those lanes (wavefront shuffles).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void merge_pair(uint64_t &a1, uint64_t &a2, uint64_t b1, uint64_t b2) {
  // (a1<=a2), (b1<=b2) -> two smallest of the four
  const uint64_t lo = a1 < b1 ? a1 : b1;
  const uint64_t hi = a1 < b1 ? b1 : a1;
  const uint64_t m2 = a2 < b2 ? a2 : b2;
  a1 = lo;
  a2 = hi < m2 ? hi : m2;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
  const uint32_t lo = __shfl_xor((uint32_t)v, mask, 64);
  const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), mask, 64);
  return ((uint64_t)hi << 32) | lo;
}

template <int LPQ>
__global__ __launch_bounds__(kThreads) void l1k2_merge_kernel(const uint64_t *__restrict__ part,
                                                              int N, int S,
                                                              uint64_t *__restrict__ out_idx,
                                                              int32_t *__restrict__ out_dist) {
  const long long gt = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int query = (int)(gt / LPQ);
  const int sub = (int)(gt % LPQ);
  uint64_t a1 = kKey64None, a2 = kKey64None;
  if (query < N) {
    const uint64_t *p = part + (size_t)query * S * 2;
    for (int s = sub; s < S; s += LPQ) merge_pair(a1, a2, p[2 * s], p[2 * s + 1]);
  }
#pragma unroll
  for (int m = 1; m < LPQ; m <<= 1) {
    const uint64_t b1 = shfl_xor_u64(a1, m);
    const uint64_t b2 = shfl_xor_u64(a2, m);
    merge_pair(a1, a2, b1, b2);
  }
  if (query < N && sub == 0) {
    const bool n1 = a1 == kKey64None, n2 = a2 == kKey64None;
    out_idx[2 * (size_t)query + 0] = n1 ? ~0ull : (a1 & 0xFFFFFFFFull);
    out_idx[2 * (size_t)query + 1] = n2 ? ~0ull : (a2 & 0xFFFFFFFFull);
    out_dist[2 * (size_t)query + 0] = n1 ? 0x7FFFFFFF : (int32_t)(a1 >> 32);
    out_dist[2 * (size_t)query + 1] = n2 ? 0x7FFFFFFF : (int32_t)(a2 >> 32);
  }
}

// Zero-pad rows from `dim` to `dim_pad` bytes (L1 distances are unchanged).
__global__ __launch_bounds__(kThreads) void pad_rows_kernel(const uint8_t *__restrict__ src,
                                                            uint8_t *__restrict__ dst, size_t rows,
                                                            int dim, int dim_pad) {
  const size_t total = rows * (size_t)(dim_pad / 4);
  for (size_t e = blockIdx.x * (size_t)kThreads + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kThreads) {
    const size_t r = e / (dim_pad / 4);
    const int c = (int)(e % (dim_pad / 4)) * 4;
    uint32_t v = 0;
    if (c < dim) v = *reinterpret_cast<const uint32_t *>(src + r * dim + c);  // dim % 16 == 0
    reinterpret_cast<uint32_t *>(dst)[e] = v;
  }
}

// Tuning knobs (read once): SPECTAVI_L1K2_Q / SPECTAVI_L1K2_BLOCKS override the plan
// (tools/l1k2_sweep.py).
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

template <int D4, int Q>
void launch_tile(const uint8_t *x, const uint8_t *y, int M, int N, const L1K2Plan &p,
                 uint64_t *part, hipStream_t stream) {
  dim3 grid(p.qblocks, p.slices);
  hipLaunchKernelGGL((l1k2_tile_kernel<D4, Q>), grid, dim3(kThreads), 0, stream,
                     reinterpret_cast<const uint4 *>(x), reinterpret_cast<const uint4 *>(y), M, N,
                     p.slice_rows, p.slices, part);
}

template <int D4>
void launch_tile_q(const uint8_t *x, const uint8_t *y, int M, int N, const L1K2Plan &p,
                   uint64_t *part, hipStream_t stream) {
  constexpr int QMAX = D4 <= 16 ? 4 : 2;
  if (p.q >= 4 && QMAX >= 4)
    launch_tile<D4, (QMAX >= 4 ? 4 : 1)>(x, y, M, N, p, part, stream);
  else if (p.q >= 2 && QMAX >= 2)
    launch_tile<D4, (QMAX >= 2 ? 2 : 1)>(x, y, M, N, p, part, stream);
  else
    launch_tile<D4, 1>(x, y, M, N, p, part, stream);
}

// Queries per lane.  Measured on MI355X at 256k x 256k, D=128 (tools/l1k2_sweep.py): Q=2
// (154 VGPRs, 3 waves/SIMD) beats Q=4 (224 VGPRs, 2 waves/SIMD) by ~3 % and Q=1 by ~15 %.
// Wide rows (192 / 256) also take Q=2: with one query per lane the broadcast LDS reads, not
// the SADs, bound the kernel (0.66 of the SAD peak measured at Q=1).
int max_q_for(int dim_pad) { return dim_pad <= 64 ? 4 : 2; }

}  // namespace

// Kernel row widths that are instantiated; other dims are zero-padded up.
static int pick_dim_pad(int dim) {
  static const int kDims[] = {32, 48, 64, 80, 96, 112, 128, 144, 160, 192, 256};
  // (widths below 256 that are not instantiated are better off padded to the next one than in the wide
  // kernel unpadded: measured 0.55-0.80 of the SAD peak on the true width against 0.51-0.64, dims 48..240,
  // when only {64, 128, 144, 192, 256} existed)
  for (int d : kDims)
    if (dim <= d) return d;
  return dim <= kMaxGenericDim ? dim : -1;  // wide-row kernel: any multiple of 16 bytes (whole 128-byte chunks + a ragged end)
}

L1K2Plan l1k2_plan(int xrows, int yrows, int dim) {
  L1K2Plan p{};
  p.dim_pad = pick_dim_pad(dim);
  if (p.dim_pad < 0 || xrows < 0 || yrows < 0) return p;
  const bool wide = p.dim_pad > 256;
  const int qmax = wide ? 2 : max_q_for(p.dim_pad);
  const int qlanes = kThreads;  // queries per workgroup per unit of q
  // queries per lane: as many as registers allow once there are enough queries
  // to keep >= 512 workgroups of 256 lanes busy without it
  // as many queries per lane as registers allow, unless that leaves too few workgroups
  // (query blocks x possible database slices) to fill the chip
  int q = qmax;
  while (q > 1) {
    const long long qb = ((long long)yrows + qlanes * q - 1) / (qlanes * q);
    const long long smax = std::max<long long>(1, xrows / (wide ? kWideRows : kTileRows));
    if (qb * smax >= 1024) break;
    q /= 2;
  }
  static const int q_env = env_int("SPECTAVI_L1K2_Q", 0);
  if (q_env == 1 || q_env == 2 || q_env == 4) q = std::min(q_env, qmax);
  p.q = q;
  p.qblocks = std::max(1, (yrows + qlanes * q - 1) / (qlanes * q));
  // database slices: enough workgroups to fill 256 CUs x 2 several times over,
  // each slice a multiple of the tile and <= 65536 rows (16-bit local index)
  static const int want_blocks = std::max(1, env_int("SPECTAVI_L1K2_BLOCKS", 16384));
  int s_target = std::max(1, (want_blocks + p.qblocks - 1) / p.qblocks);
  long long rows = (xrows + s_target - 1) / s_target;
  rows = (rows + kTileRows - 1) / kTileRows * kTileRows;
  rows = std::min<long long>(std::max<long long>(rows, kTileRows), 65536);
  p.slice_rows = (int)rows;
  p.slices = std::max(1, (int)((xrows + rows - 1) / rows));
  if (p.dim_pad != dim) {
    p.pad_x_bytes = round_up((size_t)xrows * p.dim_pad, 256);
    p.pad_y_bytes = round_up((size_t)yrows * p.dim_pad, 256);
  }
  p.part_bytes = round_up((size_t)std::max(yrows, 1) * p.slices * 2 * sizeof(uint64_t), 256);
  p.total_bytes = p.pad_x_bytes + p.pad_y_bytes + p.part_bytes;
  return p;
}

int l1k2_run(const uint8_t *d_x, const uint8_t *d_y, int xrows, int yrows, int dim,
             uint64_t *d_idx, int32_t *d_dist, void *d_ws, size_t ws_bytes, hipStream_t stream) {
  if (xrows < 0 || yrows < 0) return set_error(SPV_ERR_INVALID, "negative row count");
  if (dim <= 0 || dim % 16 != 0)
    return set_error(SPV_ERR_INVALID,
                     "Input matrix inner dimensions must be 16-byte aligned (dim=%d).", dim);
  if (dim > kMaxGenericDim)
    return set_error(SPV_ERR_INVALID, "dim=%d > %d is not supported by the gfx950 L1 kernels", dim,
                     kMaxGenericDim);
  if (yrows == 0) return SPV_OK;
  if (!d_y || !d_idx || !d_dist || (xrows > 0 && !d_x))
    return set_error(SPV_ERR_INVALID, "null device pointer");
  // the kernels read rows as 16-byte vectors (the reference's sad_16 needs the same alignment:
  // _mm_load_si128, src/BruteForceNnL1K2.h:43-48) and write 8- / 4-byte results
  if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y) |
       reinterpret_cast<uintptr_t>(d_ws)) & 15)
    return set_error(SPV_ERR_INVALID, "device pointers must be 16-byte aligned (x %p, y %p, ws %p)",
                     (const void *)d_x, (const void *)d_y, d_ws);
  if ((reinterpret_cast<uintptr_t>(d_idx) & 7) || (reinterpret_cast<uintptr_t>(d_dist) & 3))
    return set_error(SPV_ERR_INVALID, "output pointers must be aligned to their element size");
  const L1K2Plan p = l1k2_plan(xrows, yrows, dim);
  if (ws_bytes < p.total_bytes || !d_ws)
    return set_error(SPV_ERR_INVALID, "workspace too small: %zu < %zu", ws_bytes, p.total_bytes);

  uint8_t *ws = static_cast<uint8_t *>(d_ws);
  const uint8_t *kx = d_x, *ky = d_y;
  if (p.dim_pad != dim) {
    uint8_t *px = ws;
    uint8_t *py = ws + p.pad_x_bytes;
    if (xrows > 0)
      hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(kThreads), 0, stream, d_x, px,
                         (size_t)xrows, dim, p.dim_pad);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(kThreads), 0, stream, d_y, py,
                       (size_t)yrows, dim, p.dim_pad);
    kx = px;
    ky = py;
  }
  uint64_t *part = reinterpret_cast<uint64_t *>(ws + p.pad_x_bytes + p.pad_y_bytes);

  {
    ProfScope prof("l1k2_tile", stream);
  switch (p.dim_pad) {
    case 32: launch_tile_q<8>(kx, ky, xrows, yrows, p, part, stream); break;
    case 48: launch_tile_q<12>(kx, ky, xrows, yrows, p, part, stream); break;
    case 64: launch_tile_q<16>(kx, ky, xrows, yrows, p, part, stream); break;
    case 80: launch_tile_q<20>(kx, ky, xrows, yrows, p, part, stream); break;
    case 96: launch_tile_q<24>(kx, ky, xrows, yrows, p, part, stream); break;
    case 112: launch_tile_q<28>(kx, ky, xrows, yrows, p, part, stream); break;
    case 128: launch_tile_q<32>(kx, ky, xrows, yrows, p, part, stream); break;
    case 144: launch_tile_q<36>(kx, ky, xrows, yrows, p, part, stream); break;
    case 160: launch_tile_q<40>(kx, ky, xrows, yrows, p, part, stream); break;
    case 192: launch_tile_q<48>(kx, ky, xrows, yrows, p, part, stream); break;
    case 256: launch_tile_q<64>(kx, ky, xrows, yrows, p, part, stream); break;
    default: {
      if (p.dim_pad <= 256 || p.dim_pad > kMaxGenericDim || p.dim_pad % 16)
        return set_error(SPV_ERR_INVALID, "internal: bad dim_pad %d", p.dim_pad);
      const size_t lds = (size_t)kWideRows * p.dim_pad;
      const dim3 grid((unsigned)((p.qblocks + 7) / 8 * 8 * p.slices));  // decoded XCD-aware in the kernel
      if (p.q >= 2)
        hipLaunchKernelGGL((l1k2_wide_kernel<2>), grid, dim3(kThreads), lds, stream,
                           reinterpret_cast<const uint4 *>(kx), reinterpret_cast<const uint4 *>(ky), xrows, yrows,
                           p.dim_pad / 4, p.slice_rows, p.slices, part);
      else
        hipLaunchKernelGGL((l1k2_wide_kernel<1>), grid, dim3(kThreads), lds, stream,
                           reinterpret_cast<const uint4 *>(kx), reinterpret_cast<const uint4 *>(ky), xrows, yrows,
                           p.dim_pad / 4, p.slice_rows, p.slices, part);
    }
  }
  }
  SPV_HIP_CHECK(hipGetLastError());

  ProfScope prof_merge("l1k2_merge", stream);

  if (p.slices <= 4) {
    const int blocks = (yrows + kThreads - 1) / kThreads;
    hipLaunchKernelGGL((l1k2_merge_kernel<1>), dim3(blocks), dim3(kThreads), 0, stream, part, yrows,
                       p.slices, d_idx, d_dist);
  } else if (p.slices <= 32) {
    const int blocks = (int)(((long long)yrows * 8 + kThreads - 1) / kThreads);
    hipLaunchKernelGGL((l1k2_merge_kernel<8>), dim3(blocks), dim3(kThreads), 0, stream, part, yrows,
                       p.slices, d_idx, d_dist);
  } else {
    const int blocks = (int)(((long long)yrows * 64 + kThreads - 1) / kThreads);
    hipLaunchKernelGGL((l1k2_merge_kernel<64>), dim3(blocks), dim3(kThreads), 0, stream, part,
                       yrows, p.slices, d_idx, d_dist);
  }
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv
