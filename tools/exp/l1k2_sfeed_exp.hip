// l1k2_sfeed_exp.hip -- EXPERIMENT, not part of libspectavi.so.
//
// Scalar-feed variant of the L1 tile kernel: the database row is fetched with wave-uniform scalar
// loads (s_load_dwordx8/x16 through the scalar cache) and enters v_sad_hi_u8 as an SGPR operand;
// no LDS, no barriers.  Round 1 measured it at the same rate as the LDS-fed kernel shipped in
// spectavi_amd/csrc/l1k2.hip (profiles/r01_l1k2_sweep.jsonl), which is how the v_sad issue rate --
// not the operand feed -- was identified as the bound.  It used to live in the product library
// behind SPECTAVI_L1K2_FEED=sgpr; it was moved here so that the library ships only what it runs.
// To re-measure: paste the kernel back next to l1k2_tile_kernel (it uses that file's sad_hi,
// top2_insert, widen_key, kThreads, kKeyNone) and launch it with the same grid.

// ---------------------------------------------------------------------------------
// Experimental scalar-feed variant (SPECTAVI_L1K2_FEED=sgpr): the database row is
// fetched with wave-uniform scalar loads (s_load_dwordx8/x16 through the scalar
// cache) and enters v_sad_hi_u8 as an SGPR operand; no LDS, no barriers.
// ---------------------------------------------------------------------------------
template <int D4, int Q>
__device__ __forceinline__ void row_update_s(const uint32_t (&qreg)[Q][D4], const uint32_t (&xs)[D4],
                                             uint32_t j, uint32_t (&k1)[Q], uint32_t (&k2)[Q]) {
  uint32_t acc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = j;
#pragma unroll
  for (int i = 0; i < D4; ++i) {
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = sad_hi(qreg[q][i], xs[i], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) top2_insert(k1[q], k2[q], acc[q]);
}

template <int D4, int Q>
__global__ __launch_bounds__(kThreads, 2) void l1k2_tile_kernel_sfeed(
    const uint32_t *__restrict__ x, const uint4 *__restrict__ y, int M, int N, int slice_rows,
    int S, uint64_t *__restrict__ part) {
  constexpr int V4 = D4 / 4;
  const int t = threadIdx.x;
  const int qb = blockIdx.x;
  const int s = blockIdx.y;
  const int row_begin = s * slice_rows;
  const int row_end = min(M, row_begin + slice_rows);

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

  const int nrows = row_end - row_begin;
  if (nrows > 0) {
    const uint32_t *xr = x + (size_t)row_begin * D4;
    const uint32_t *xlast = x + (size_t)(row_end - 1) * D4;
    uint32_t xa[D4], xb[D4];
#pragma unroll
    for (int i = 0; i < D4; ++i) xa[i] = xr[i];
    for (int r = 0; r < nrows; r += 2) {
      const uint32_t *x1 = (r + 1 < nrows) ? xr + D4 : xlast;
#pragma unroll
      for (int i = 0; i < D4; ++i) xb[i] = x1[i];
      row_update_s<D4, Q>(qreg, xa, (uint32_t)r, k1, k2);
      const uint32_t *x2 = (r + 2 < nrows) ? xr + 2 * D4 : xlast;
#pragma unroll
      for (int i = 0; i < D4; ++i) xa[i] = x2[i];
      if (r + 1 < nrows) row_update_s<D4, Q>(qreg, xb, (uint32_t)(r + 1), k1, k2);
      xr += 2 * D4;
    }
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

