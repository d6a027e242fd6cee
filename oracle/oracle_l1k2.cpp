// oracle_l1k2.cpp -- CPU restatement of the reference's L1 2-NN hot loop.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under spectavi_amd/ may import, link or
// call this; it is the checker for tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py.
//
// Follows (read as text, restated here, not copied):
//   reference src/BruteForceNnL1K2.h:43-48   sad_16: _mm_sad_epu8 on 16-byte groups
//   reference src/BruteForceNnL1K2.h:84-145  find_neighbours: OpenMP over query
//       rows (:92), sentinels INT_MAX / size_t(-1) (:100-103), sequential scan of
//       candidate rows (:109), early-exit prune against the current second best
//       (:118-121), strict-< streaming top-2 update (:129-139), prune threshold
//       armed once two neighbours exist (:140-142)
//   reference src/BruteForceNnL1K2.h:71-82   ctor checks (equal dims, dim % 16 == 0)
//
// The reference itself cannot be compiled here (Eigen3 / NdArray.h absent, see
// DESIGN.md), so this restatement is pinned by the reference's own test
// property (test/test_feature.py:102-121: distances equal the numpy brute-force
// L1 exactly on 200x144 uniform uint8) in tests/test_oracle.py, and by hand-checked
// tie cases for the index rule.  It doubles as the timed CPU baseline: same loop
// nest, same SSE2 instruction, same prune, OpenMP over queries; build it the
// way the reference is built (-O3 -DNDEBUG -fopenmp, no -march:
// reference CMakeLists.txt:12,55-57).

#include <emmintrin.h>
#include <limits.h>
#include <stddef.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

inline int sad16(const uint8_t *a, const uint8_t *b) {
  const __m128i va = _mm_loadu_si128(reinterpret_cast<const __m128i *>(a));
  const __m128i vb = _mm_loadu_si128(reinterpret_cast<const __m128i *>(b));
  const __m128i s = _mm_sad_epu8(va, vb);  // two 64-bit lanes, each the SAD of 8 bytes
  return _mm_cvtsi128_si32(s) + _mm_extract_epi16(s, 4);
}

// Streaming top-2 state for one query, exactly the reference's update rule.
struct Top2 {
  int32_t d0 = INT_MAX, d1 = INT_MAX;
  uint64_t i0 = ~0ull, i1 = ~0ull;
  int32_t worst = -1;  // prune threshold, armed once a second neighbour exists

  inline void offer(const uint8_t *xr, const uint8_t *yr, int groups, uint64_t row) {
    int32_t acc = 0;
    for (int gidx = 0; gidx < groups; ++gidx) {
      acc += sad16(xr + 16 * gidx, yr + 16 * gidx);
      if (worst >= 0 && acc > worst) return;  // cannot enter the top two any more
    }
    if (acc < d0) {
      d1 = d0;
      i1 = i0;
      d0 = acc;
      i0 = row;
    } else if (acc < d1) {
      d1 = acc;
      i1 = row;
    }
    if (i1 != ~0ull) worst = d1;
  }
};

}  // namespace

extern "C" {

// All-pairs form (IdentityFilter, reference src/BruteForceNnL1K2.h:19-38).
// idx: uint64[yrows,2], dist: int32[yrows,2].  Returns 0, or 1 on bad dims.
int oracle_nn_bruteforcel1k2(const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim,
                             int nthreads, uint64_t *idx, int32_t *dist) {
  if (dim <= 0 || dim % 16 != 0) return 1;
  const int groups = dim / 16;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int q = 0; q < yrows; ++q) {
    const uint8_t *yr = y + (size_t)q * dim;
    Top2 t;
    for (int r = 0; r < xrows; ++r) t.offer(x + (size_t)r * dim, yr, groups, (uint64_t)r);
    idx[2 * (size_t)q] = t.i0;
    idx[2 * (size_t)q + 1] = t.i1;
    dist[2 * (size_t)q] = t.d0;
    dist[2 * (size_t)q + 1] = t.d1;
  }
  return 0;
}

// Candidate-list form (the SetFilter role, reference src/CascadingHashNn.h:22-50):
// query q visits cand[off[q] .. off[q+1]) in the given order.
int oracle_l1k2_candidates(const uint8_t *x, const uint8_t *y, int yrows, int dim,
                           const int64_t *off, const int32_t *cand, int nthreads, uint64_t *idx,
                           int32_t *dist) {
  if (dim <= 0 || dim % 16 != 0) return 1;
  const int groups = dim / 16;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
  for (int q = 0; q < yrows; ++q) {
    const uint8_t *yr = y + (size_t)q * dim;
    Top2 t;
    for (int64_t c = off[q]; c < off[q + 1]; ++c)
      t.offer(x + (size_t)cand[c] * dim, yr, groups, (uint64_t)cand[c]);
    idx[2 * (size_t)q] = t.i0;
    idx[2 * (size_t)q + 1] = t.i1;
    dist[2 * (size_t)q] = t.d0;
    dist[2 * (size_t)q + 1] = t.d1;
  }
  return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// OpenMP team size of the parallel regions that take no explicit count (oracle_cascade.cpp): the
// CPU-baseline table runs the cascade at the reference's hard-wired 8 threads and at all of them.
void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

}  // extern "C"
