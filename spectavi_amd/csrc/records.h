// records.h -- the 16-byte result record that travels between GPUs.
//
// One record per query: (idx0, idx1, d0, d1) as four 32-bit words -- what north_star's "RCCL gather
// of (idx0, idx1, d0, d1)" moves.  Database indices fit 31 bits (the C-ABI passes xrows as int);
// the no-neighbour sentinel (size_t)-1 of the reference (src/BruteForceNnL1K2.h:100-103) travels as
// -1.  Distances travel as their 32 bits: int32 for nn_bruteforcel1k2, the float32 bit pattern for
// nn_cascading_hash.  The same arithmetic runs on the device (gather.hip) and on the host
// (spv_records_pack / spv_records_unpack, used by callers that run their own collective and by the
// CPU unit tests); spectavi_amd/sharded.py states it a third time with torch.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define SPV_HD __host__ __device__ __forceinline__
#else
#define SPV_HD inline
#endif

namespace spv {

struct Record {
  int32_t idx0, idx1;
  uint32_t d0, d1;
};
static_assert(sizeof(Record) == 16, "record must be 16 bytes");

SPV_HD Record record_pack(uint64_t i0, uint64_t i1, uint32_t d0, uint32_t d1) {
  Record r;
  r.idx0 = (int32_t)(uint32_t)i0;  // (size_t)-1 -> -1; real indices < 2^31 unchanged
  r.idx1 = (int32_t)(uint32_t)i1;
  r.d0 = d0;
  r.d1 = d1;
  return r;
}

SPV_HD uint64_t record_widen_idx(int32_t v) { return v < 0 ? ~0ull : (uint64_t)(uint32_t)v; }

// Contiguous balanced shards of [0, total) over G ranks, the first (total % G) one row longer:
// which rank owns row q, and where inside that rank's shard.
SPV_HD void shard_locate(long long q, long long total, int G, int *rank, long long *local) {
  const long long base = total / G, extra = total % G;
  const long long big = extra * (base + 1);
  if (q < big) {
    *rank = (int)(q / (base + 1));
    *local = q - (long long)*rank * (base + 1);
  } else {
    const long long r = (q - big) / (base > 0 ? base : 1);
    *rank = (int)(extra + r);
    *local = (q - big) - r * base;
  }
}

SPV_HD long long shard_lo(long long total, int G, int r) {
  const long long base = total / G, extra = total % G;
  return r * base + (r < extra ? r : extra);
}

}  // namespace spv
