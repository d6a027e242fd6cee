// common.h -- shared host-side helpers for libspectavi.so (gfx950 build).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>

#include "../../include/spectavi_amd.h"

namespace spv {

// Records status + message for the calling thread; returns `status`.
int set_error(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

#define SPV_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      return ::spv::set_error(_e == hipErrorOutOfMemory ? SPV_ERR_NOMEM : SPV_ERR_HIP,   \
                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                              __FILE__, __LINE__);                                       \
    }                                                                                    \
  } while (0)

#define SPV_TRY(expr)            \
  do {                           \
    int _s = (expr);             \
    if (_s != SPV_OK) return _s; \
  } while (0)

// Selects the process-wide device for host-pointer entry points on this thread.
int ensure_device();

// Compute units of the calling thread's current device (cached per device; 256 on MI355X).
int device_cu_count();

// Brackets a kernel launch with hipEvents on `stream` while profiling is enabled.
struct ProfScope {
  ProfScope(const char *name, hipStream_t stream);
  ~ProfScope();
  const char *name_;
  hipStream_t stream_;
  hipEvent_t start_ = nullptr;
};

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- L1 2-NN (l1k2.hip) -------------------------------------------------------------
struct L1K2Plan {
  int dim_pad;     // kernel row width in bytes (>= dim, zero padded)
  int q;           // queries per lane
  int slice_rows;  // database rows per slice (<= 65536)
  int slices;      // number of database slices
  int qblocks;     // query blocks
  size_t pad_x_bytes, pad_y_bytes;  // padded copies (0 when dim == dim_pad)
  size_t part_bytes;                // partial top-2 keys
  size_t total_bytes;
};
L1K2Plan l1k2_plan(int xrows, int yrows, int dim);
int l1k2_run(const uint8_t *d_x, const uint8_t *d_y, int xrows, int yrows, int dim,
             uint64_t *d_idx, int32_t *d_dist, void *d_ws, size_t ws_bytes, hipStream_t stream);

// ---- cascade hash (cascade.hip) -----------------------------------------------------
size_t cascade_workspace_bytes(int xrows, int yrows, int dim, int m, int n, int g);
int cascade_run(const float *d_x, const float *d_y, int xrows, int yrows, int dim, int m, int n,
                int g, const float *d_dict, uint64_t *d_idx, float *d_dist, int32_t *d_ncand,
                void *d_ws, size_t ws_bytes, hipStream_t stream);

// ---- DLT (dlt.hip) ------------------------------------------------------------------
int dlt_run(const double *P0, const double *P1, long long npt, const double *d_x,
            const double *d_xp, double *d_dst, bool want_error, hipStream_t stream);

// ---- ratio test + compaction (match.hip) ---------------------------------------------
size_t ratio_workspace_bytes(int yrows);
int ratio_run(const uint64_t *d_idx, const void *d_dist, int dist_is_float, int yrows,
              double min_ratio, int *d_matches, int *d_count, void *d_ws, size_t ws_bytes,
              hipStream_t stream);

// ---- format adapters (adapter.hip) ---------------------------------------------------
int sift_split_run(const float *d_table, int rows, float *d_geom, uint8_t *d_desc, hipStream_t stream);
int gather_match_coords_run(const float *d_geom_x, const float *d_geom_y, const int *d_matches,
                            const int *d_count, int capacity, double *d_x0, double *d_x1,
                            hipStream_t stream);

size_t normalize_workspace_bytes(int dim);
// the larger workspace with which normalize_run folds the column sums of a big table instead of
// walking them (same results; with the smaller one it walks)
size_t normalize_workspace_bytes_rows(int rows, int dim);
int normalize_run(const float *d_x, int rows, int dim, float *d_out_f32, unsigned char *d_out_u8,
                  void *d_ws, size_t ws_bytes, hipStream_t stream);

// ---- multi-device result gather over RCCL (gather.hip) -------------------------------
struct GatherCtx;                    // communicator clique + one stream per rank, cached per device list
std::mutex &gather_mutex();          // held by the caller around every use of a clique
int gather_ctx_get(const std::vector<int> &devs, bool use_rccl, GatherCtx **out);
hipStream_t gather_stream(GatherCtx *ctx, int rank);
// (idx uint64[cnt,2], 32-bit dist[cnt,2]) -> cnt 16-byte records (records.h)
int gather_pack_run(const uint64_t *d_idx, const void *d_d32, long long cnt, void *d_rec, hipStream_t stream);
// bytes_per_rank bytes from every rank's d_send[r] into slot r of d_recv_root on rank 0: ncclGather, or
// (peer-copy transport) one hipMemcpyPeerAsync per rank
int gather_bytes_run(GatherCtx *ctx, const std::vector<const void *> &d_send, void *d_recv_root,
                     size_t bytes_per_rank);
// [G][max_cnt] records -> the ABI layout over all `total` rows
int gather_widen_run(const void *d_recv, long long total, int G, long long max_cnt, uint64_t *d_idx, void *d_d32,
                     hipStream_t stream);

size_t dlt_score_workspace_bytes(int nhyp, long long npt);
// d_ws (may be NULL / short: the scorer then runs in one pass, same results) holds the work list of
// the solves that are deferred to the second pass
// d_live / d_nlive (may be NULL): score only the hypotheses 4 f .. 4 f + 3 of the *d_nlive candidates f
// listed in d_live (device memory); the counts of the others stay 0, their mask rows unwritten
int dlt_score_run(const double *P0, const double *d_p1s, int nhyp, long long npt, const double *d_x,
                  const double *d_xp, double max_error, int *d_counts, unsigned char *d_mask, void *d_ws,
                  size_t ws_bytes, hipStream_t stream, const int *d_live = nullptr, const int *d_nlive = nullptr,
                  int rows_cap = 65535);


// ---- RANSAC candidate processing (dlt.hip): gate, E, four cameras, scoring, best camera ----
size_t ransac_workspace_bytes(int nF, long long npt, bool want_mask);
int ransac_process_run(const double *d_Fs, int nF, long long npt, const double *d_x0, const double *d_x1,
                       double ratio_allowed, double required_percent, double max_error, int find_best,
                       int *d_success, int *d_inlier_count, int *d_best_cam, double *d_best_P, double *d_ratio,
                       double *d_E, int *d_counts4, unsigned char *d_mask, void *d_ws, size_t ws_bytes,
                       hipStream_t stream, int score_rows_cap = 65535);  // grid rows of the scorer: a RANSAC batch, whose
                                                                          // candidates are nearly all gated, passes 2048

// ---- seven-point solver and the RANSAC loop around the candidate processing (ransac.hip) ----
// d_x, d_xp double[n,7,2] euclidean; d_Fs double[n,3,9] (NaN in the slots of missing roots);
// d_nroot int[n] and d_basis double[n,2,9] may be NULL
int seven_point_run(const double *d_x, const double *d_xp, int n, double *d_Fs, int *d_nroot, double *d_basis,
                    hipStream_t stream);
int ransac_fit_batch_limit(long long npt);  // tries per batch for this many correspondences
size_t ransac_fit_workspace_bytes(int batch, long long npt);
int ransac_fit_run(const double *d_x0, const double *d_x1, long long npt, double required_percent,
                   double max_error, int max_tries, int find_best, double ratio_allowed,
                   const std::function<void(int, int, int *)> &next_samples, int *success, double *essential,
                   double *camera, int *n_inliers, unsigned char *inlier_mask, int *best_try, int *best_root,
                   int *tries_run, void *d_ws, size_t ws_bytes, int batch, hipStream_t stream);

}  // namespace spv
