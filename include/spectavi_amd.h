/*
 * spectavi_amd.h -- C-ABI of libspectavi.so (MI355X / gfx950 build).
 *
 * Drop-in boundary for the descriptor-matching + DLT hot path of
 * vvhitedog/spectavi.  Section 1 re-exports, symbol for symbol, what the
 * reference's ctypes front-end binds (reference src/Spectavi.cpp, declared to
 * ctypes in spectavi/feature.py and spectavi/mvg.py).  Sections 2 and 3 add
 * status-returning variants with caller-allocated outputs (host pointers) and
 * device-pointer variants (inputs/outputs resident in HBM, asynchronous on a
 * caller stream) used by the benchmark, the tests and multi-GPU sharding.
 *
 * Conventions: row-major C-contiguous arrays, plain pointers and ints, no C++
 * or torch types.  Nothing throws across this boundary (the reference lets
 * std::runtime_error escape extern "C": src/BruteForceNnL1K2.h:75,79).
 * All compute happens in hand-written HIP kernels; there is NO CPU fallback:
 * without a usable gfx950 device every entry point fails with SPV_ERR_HIP.
 */
#ifndef SPECTAVI_AMD_H
#define SPECTAVI_AMD_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

/* NdArray: the in-repo definition, or -- when libspectavi.so is built with
 * `make NDARRAY_INC=/path/to/ctypes_ndarray/src` -- the reference's own <NdArray.h>
 * (reference src/EigenDefinitions.h:22, CMakeLists.txt:26-27). */
#ifdef SPECTAVI_EXTERNAL_NDARRAY
#include <NdArray.h>
#else
#include "NdArray.h"
#endif

/* libspectavi.so is built with -fvisibility=hidden: only what is declared between this push and
 * the matching pop is exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* status                                                                    */
/* ------------------------------------------------------------------------ */
#define SPV_OK 0
#define SPV_ERR_INVALID 1 /* argument rejected (dim % 16, m > 31, k != 2, NULL ...) */
#define SPV_ERR_HIP 2     /* HIP runtime / no device / launch failure */
#define SPV_ERR_NOMEM 3   /* host or device allocation failed */
#define SPV_ERR_INTERNAL 4 /* a C++ exception was caught at the boundary (never propagated) */

/* Status of the last call made by this thread through any entry point below
 * (the void reference-compatible symbols report errors only this way). */
int spv_last_status(void);
/* Human-readable message for spv_last_status(); valid until the thread's next call. */
const char *spv_last_error(void);
/* Number of visible HIP devices (0 if none / no driver). */
int spv_device_count(void);
/* Device used by the host-pointer entry points of this process (default:
 * environment SPECTAVI_DEVICE, else 0). */
int spv_set_device(int device);
/* Shard the host-pointer entry points over several devices of this node (default:
 * environment SPECTAVI_DEVICES = "0,1,..." or "all", else the single device above).
 * Queries / points are split into contiguous balanced shards, the database is
 * replicated, and each shard is written straight into its slice of the caller's
 * output by a host thread per device; a device may be listed more than once. */
int spv_set_devices(const int *devices, int count);
/* How the shards of a multi-device host-pointer call reach the caller's arrays.
 *   SPV_GATHER_RCCL   every device leaves its (idx0, idx1, d0, d1) records (16 bytes per query;
 *                     DLT: its output rows) in HBM, one ncclCommInitAll clique (cached per device
 *                     list) gathers them on the first listed device with ncclGather, one kernel
 *                     widens them to the ABI layout, one copy brings them to the host -- the
 *                     exchange step of SURVEY 8(e) for the loop the reference shards over OpenMP
 *                     threads (src/BruteForceNnL1K2.h:92-93).  Works with a single device too
 *                     (a clique of one).  Kernel names for spv_profile_read: "gather",
 *                     "gather_widen".  librccl is opened on first use (SPECTAVI_RCCL_LIB overrides
 *                     the name).
 *   SPV_GATHER_PEERCOPY the same records, the same root layout and widening kernel, moved with one
 *                     hipMemcpyPeerAsync per device instead of RCCL (no librccl in the process;
 *                     accepts a device listed more than once).
 *   SPV_GATHER_DIRECT every shard is copied straight into its slice of the caller's arrays; no
 *                     collective.
 *   SPV_GATHER_AUTO   (default) environment SPECTAVI_GATHER = "rccl" | "copy" | "direct" if set,
 *                     else RCCL exactly when more than one distinct device is configured. */
#define SPV_GATHER_AUTO (-1)
#define SPV_GATHER_DIRECT 0
#define SPV_GATHER_RCCL 1
#define SPV_GATHER_PEERCOPY 2
int spv_set_gather_mode(int mode);
/* Host statement of the 16-byte record format (no GPU involved), for callers that run their own
 * collective on raw records.  pack: idx uint64[n,2] ((size_t)-1 = no neighbour), dist32 = int32 or
 * float32 [n,2] -> rec int32[n,4] = (idx0, idx1, d0 bits, d1 bits), -1 = no neighbour.
 * unpack: rec [G][max_cnt][4] in rank order, ragged shards (contiguous balanced split of `total`
 * rows, the first total % G shards one row longer) padded to max_cnt -> idx uint64[total,2],
 * dist32 [total,2]. */
int spv_records_pack(const uint64_t *idx, const void *dist32, long long n, int32_t *rec);
int spv_records_unpack(const int32_t *rec, long long total, int G, long long max_cnt, uint64_t *idx,
                       void *dist32);
/* The host-pointer entry points keep freed device buffers in a per-device cache (up to
 * 4 GiB) for reuse by later calls; this releases them. */
void spv_release_cached_memory(void);
/* Library version string. */
const char *spv_version(void);

/* Optional in-library kernel timing: when enabled, the hot kernels (names:
 * "l1k2_tile", "l1k2_merge", "cascade_project", "cascade_buckets",
 * "cascade_probe_refine", "dlt") are bracketed by hipEvents recorded on the
 * stream they are launched on.  spv_profile_read synchronises with the
 * recorded events and returns launch count and summed milliseconds since the
 * last reset. */
void spv_profile_enable(int on);
void spv_profile_reset(void);
int spv_profile_read(const char *kernel, long long *launches, double *total_ms);
/* Diagnostic: sustained issue rate of one VALU instruction with register
 * operands only (op: 0 v_sad_hi_u8, 1 v_sad_u8, 2 v_sad_u16, 3 v_xor+v_add,
 * 4 v_fma_f32, 5 v_dot4_u32_u8, 6 v_med3_u32) and the shader clock held. */
int spv_microbench_valu(int op, int blocks, int iters, double *lane_ops_per_s, double *clock_ghz);
/* Diagnostic: memory ceilings.  mode 0 = 16-byte-per-lane streaming copy of table_bytes
 * (bytes read + written per second); mode 1 = random 128-byte row gathers, 8 lanes per row,
 * from a table of table_bytes (bytes gathered per second). */
int spv_microbench_memory(int mode, size_t table_bytes, double *bytes_per_s);

/* ------------------------------------------------------------------------ */
/* 1. Reference-compatible symbols (same names, argument order and meaning)  */
/* ------------------------------------------------------------------------ */

/* Exact L1 (sum |x-y|) 2-nearest-neighbour of every query row y against every
 * database row x.  Replaces reference src/Spectavi.cpp:284-298
 * (BruteForceNnL1K2::find_neighbours<IdentityFilter>, src/BruteForceNnL1K2.h:84-145).
 *   x: uint8[xrows, dim] database, y: uint8[yrows, dim] queries, dim % 16 == 0
 *   (dim <= 2048 on gfx950; wider rows are rejected with SPV_ERR_INVALID).
 *   outidx : callee-allocated size_t[yrows,2]  (col 0 = nearest)
 *   outdist: callee-allocated int  [yrows,2]
 * Result per query = the two smallest (dist, idx) pairs in lexicographic order;
 * missing neighbours (xrows < 2) are (INT_MAX, (size_t)-1) as in the reference
 * (src/BruteForceNnL1K2.h:100-103).  `nthreads` is accepted and ignored (the
 * reference uses it as the OpenMP team size, src/BruteForceNnL1K2.h:92). */
void nn_bruteforcel1k2(const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim,
                       int nthreads, NdArray *outidx, NdArray *outdist);

/* Cascade-hash candidate prefilter + L1 refine.  Replaces reference
 * src/Spectavi.cpp:321-336 (CascadingHashNn, src/CascadingHashNn.h:86-245).
 *   x,y: float32[rows, dim], integer-valued in [-128,127]; dim % 16 == 0, dim <= 2048.
 *   k must be 2 (the reference sizes the buffers by k but writes two columns,
 *   src/Spectavi.cpp:329-335); hash_bit_rate m in [1,31]; num_hash_tables n >= 1;
 *   num_candidate_neighbours g in [0, m].
 *   outidx size_t[yrows,k], outdist float[yrows,k]; queries with fewer than two
 *   candidates carry ((size_t)-1, 2147483648.0f) (src/CascadingHashNn.h:244).
 * Hyperplanes: n matrices float32[dim, m] of N(0,1) drawn from std::mt19937
 * filled dim-major (src/CascadingHashNn.h:86-100); seeded from
 * std::random_device like the reference unless spv_set_hash_seed() /
 * SPECTAVI_HASH_SEED fixed a seed. */
void nn_cascading_hash(const float *x, const float *y, int xrows, int yrows, int dim, int k,
                       int hash_bit_rate, int num_hash_tables, int num_candidate_neighbours,
                       NdArray *outidx, NdArray *outdist);

/* Two-view DLT triangulation of npt points.  Replaces reference
 * src/Spectavi.cpp:38-52 (DltTriangulator::solve, src/DltTriangulator.h:36-65).
 *   P0,P1: double[3,4]; x,xp: double[npt,3] homogeneous; dst: caller double[npt,4].
 * dst row = unit-norm right singular vector of the smallest singular value of
 * the 4x4 DLT matrix; its sign (arbitrary in the reference: Eigen JacobiSVD)
 * is canonicalised to dst[3] >= 0 (first nonzero component > 0 if dst[3]==0). */
void dlt_triangulate(const double *P0, const double *P1, int npt, const double *x,
                     const double *xp, double *dst);

/* Reprojection error of the triangulated point, ||hn(P0 X)-hn(x)|| + ||hn(P1 X)-hn(xp)||.
 * Replaces reference src/Spectavi.cpp:54-68 (src/DltTriangulator.h:67-74).
 * dst: caller double[npt]. */
void dlt_reprojection_error(const double *P0, const double *P1, int npt, const double *x,
                            const double *xp, double *dst);

/* The seven-point algorithm: the up to three fundamental matrices through seven correspondences.
 * Replaces reference src/Spectavi.cpp:14-36 (FundamentalMatrixFitter::solve,
 * src/FundamentalMatrixFitter.h:108-246).
 *   x, xp: double[7,2] euclidean image points; *nroot: number of solutions (0..3);
 *   dst: caller double[3,3,3], the first *nroot matrices written (row-major), F = z F0 + (1-z) F1
 *   unnormalised as in the reference, with xp^T F x = 0 for the seven pairs. */
void seven_point_algorithm(const double *x, const double *xp, int *nroot, double *dst);

/* RANSAC fit of the two-view geometry.  Replaces reference src/Spectavi.cpp:70-87
 * (RansacFitter::fit_essential, src/RansacFitter.h:152-272): per try a 7-subset of the
 * correspondences, the seven-point solutions, every solution through
 * process_fundamental_matrix, the best model kept; stops at the first model whose inlier share
 * exceeds required_percent_inliers.
 *   x0, x1: double[npt,3] homogeneous (npt >= 10, as the reference's constructor demands).
 *   *success; essential: callee-allocated double[3,3] = the winning seven-point solution (what the
 *   reference stores, :205); camera: double[3,4]; *inlier_percent; inlier_idx: int32[n,1].
 *   No model kept: essential and inlier_idx are 0 x 0, camera is [I | 0], as in the reference.
 * Tries run in batches on the device and are ranked in try order, i.e. the result is the
 * reference's for nthread = 1 given the same subsets (with OpenMP the reference's own result
 * depends on thread timing).  The subsets come from one std::mt19937 per call, seeded from
 * std::random_device like the reference unless SPECTAVI_RANSAC_SEED is set.  `progressbar` is
 * accepted and ignored. */
void ransac_fitter(const double *x0, const double *x1, int npt, double required_percent_inliers,
                   double reprojection_error_allowed, int maximum_tries, bool find_best_even_in_failure,
                   double singular_value_ratio_allowed, bool progressbar, bool *success, NdArray *essential,
                   NdArray *camera, double *inlier_percent, NdArray *inlier_idx);

/* ------------------------------------------------------------------------ */
/* 2. Host-pointer variants: caller-allocated outputs, int status            */
/* ------------------------------------------------------------------------ */

/* idx: uint64[yrows,2], dist: int32[yrows,2]. */
int spv_nn_bruteforcel1k2(const uint8_t *x, const uint8_t *y, int xrows, int yrows, int dim,
                          uint64_t *idx, int32_t *dist);

/* As nn_cascading_hash but with explicit hyperplanes: dict is
 * float32[n, dim, m] (table-major, then dim, then bit: the fill order of
 * src/CascadingHashNn.h:92-98).  idx: uint64[yrows,2], dist: float32[yrows,2].
 * ncand (may be NULL): int32[yrows] number of candidate rows examined (bucket
 * entries visited; a row reached through several tables counts once per table). */
int spv_nn_cascading_hash(const float *x, const float *y, int xrows, int yrows, int dim, int m,
                          int n, int g, const float *dict, uint64_t *idx, float *dist,
                          int32_t *ncand);

/* Fill dict[n*dim*m] exactly as the reference's generate_hash_dict would from
 * std::mt19937(seed) + std::normal_distribution<float>(0,1). */
int spv_generate_hash_dict(uint32_t seed, int dim, int m, int n, float *dict);
/* Fix (use_fixed != 0) or release the seed used by nn_cascading_hash. */
void spv_set_hash_seed(uint32_t seed, int use_fixed);

/* RANSAC hypothesis scoring (the inner loops of reference src/RansacFitter.h:59-95):
 * for each of nhyp candidate second cameras P1s[h] (double[nhyp,3,4]) triangulate all npt
 * correspondences against P0 and count the inliers, i.e. points with
 * reprojection_error() <= max_error that are in front of both cameras
 * (src/DltTriangulator.h:67-86).  counts: int32[nhyp]; mask (may be NULL):
 * uint8[nhyp, npt], 1 = inlier. */
int spv_dlt_score_hypotheses(const double *P0, const double *P1s, int nhyp, int npt,
                             const double *x, const double *xp, double max_error,
                             int32_t *counts, uint8_t *mask);

/* What the reference's RANSAC does with each candidate fundamental matrix,
 * RansacFitter::process_fundamental_matrix (reference src/RansacFitter.h:42-95), batched over nF
 * candidates Fs double[nF,3,3] and all npt correspondences x0, x1 double[npt,3]:
 *   - JacobiSVD of F; gate_ratio = |s0 - s1| / (|s0 + s1| / 2); candidates with
 *     gate_ratio > singular_value_ratio_allowed are rejected (:49-53);
 *   - E = U diag(1,1,0) V^T (:54-56) and its four candidate second cameras, Essential2Cameras
 *     (src/Camera.h:31-46), against the first camera [I | 0];
 *   - every camera scored over all correspondences (spv_dlt_score_hypotheses' rule, :59-73);
 *   - in camera order, a camera becomes the best when its inlier fraction reaches
 *     required_percent_inliers (or find_best_even_in_failure) and exceeds the best so far (:74-84).
 * Outputs per candidate: success int32[nF] (0/1), inlier_count int32[nF], best_camera int32[nF]
 * (0..3 in the order (Ra,t) (Ra,-t) (Rb,t) (Rb,-t), -1 if none); optional (NULL to skip): best_P
 * double[nF,3,4] (zeros if none), gate_ratio double[nF], E double[nF,3,3] (NaN if gated), counts4
 * int32[nF,4] (-1 if gated), inlier_mask uint8[nF,npt] = inliers of the best camera (the
 * reference's inlier_idx, :86-94, as a mask).  The two SVDs run the same two-sided Jacobi iteration
 * as Eigen's JacobiSVD (which of the four cameras comes first depends on its column signs).
 * A candidate with a NaN entry is rejected like a gated one (gate_ratio NaN, counts4 -1). */
int spv_ransac_process_candidates(const double *Fs, int nF, const double *x0, const double *x1, int npt,
                                  double singular_value_ratio_allowed, double required_percent_inliers,
                                  double reprojection_error_allowed, int find_best_even_in_failure,
                                  int32_t *success, int32_t *inlier_count, int32_t *best_camera,
                                  double *best_P, double *gate_ratio, double *E, int32_t *counts4,
                                  uint8_t *inlier_mask);
/* Device form: everything resident in HBM, nF <= 16383 per call, d_ws >=
 * spv_ransac_workspace_bytes(nF, npt, d_inlier_mask != NULL). */
size_t spv_ransac_workspace_bytes(int nF, long long npt, int want_mask);
int spv_ransac_process_candidates_device(const double *d_Fs, int nF, long long npt, const double *d_x0,
                                         const double *d_x1, double singular_value_ratio_allowed,
                                         double required_percent_inliers, double reprojection_error_allowed,
                                         int find_best_even_in_failure, int32_t *d_success,
                                         int32_t *d_inlier_count, int32_t *d_best_camera, double *d_best_P,
                                         double *d_gate_ratio, double *d_E, int32_t *d_counts4,
                                         uint8_t *d_inlier_mask, void *d_ws, size_t ws_bytes, void *stream);

/* Batched seven-point algorithm: x, xp double[n,7,2]; nroot int32[n]; Fs double[n,3,3,3] (slots of
 * missing roots are NaN); basis (may be NULL) double[n,2,3,3], the null-space pair (F0, F1). */
int spv_seven_point(const double *x, const double *xp, int n, int32_t *nroot, double *Fs, double *basis);
int spv_seven_point_device(const double *d_x, const double *d_xp, int n, double *d_Fs, int32_t *d_nroot,
                           double *d_basis, void *stream);

/* ransac_fitter with plain outputs.  seed != 0 fixes the subsets (0: SPECTAVI_RANSAC_SEED if set,
 * else std::random_device); spv_ransac_sample(seed, npt, ntries, samples) writes the very subsets
 * (int32[ntries,7], drawn as the reference's floyd_sample draws them, src/RansacFitter.h:120-132)
 * that spv_ransac_fit(seed) uses, and spv_ransac_fit_samples takes them from the caller.
 *   essential double[9], camera double[12] (untouched if no model was kept), inlier_idx int32[npt]
 *   (first *n_inliers entries); optional (NULL to skip): *best_try / *best_root (-1 if none),
 *   *tries_run (tries actually evaluated: batches end early at the first success). */
int spv_ransac_sample(unsigned long long seed, int npt, int ntries, int32_t *samples);
int spv_ransac_fit(const double *x0, const double *x1, int npt, double required_percent_inliers,
                   double reprojection_error_allowed, int maximum_tries, int find_best_even_in_failure,
                   double singular_value_ratio_allowed, unsigned long long seed, int32_t *success,
                   double *essential, double *camera, double *inlier_percent, int32_t *inlier_idx,
                   int32_t *n_inliers, int32_t *best_try, int32_t *best_root, int32_t *tries_run);
int spv_ransac_fit_samples(const double *x0, const double *x1, int npt, double required_percent_inliers,
                           double reprojection_error_allowed, const int32_t *samples, int ntries,
                           int find_best_even_in_failure, double singular_value_ratio_allowed, int32_t *success,
                           double *essential, double *camera, double *inlier_percent, int32_t *inlier_idx,
                           int32_t *n_inliers, int32_t *best_try, int32_t *best_root, int32_t *tries_run);
/* The correspondences already in HBM (d_x0, d_x1 double[npt,3] on the caller's current device, e.g.
 * what spv_gather_match_coords_device left there): samples (host int32[maximum_tries,7]) or, if
 * NULL, the seed choose the subsets; the small results come back to host memory.  Synchronises
 * `stream` after every batch of tries (the loop stops at the first success). */
int spv_ransac_fit_device(const double *d_x0, const double *d_x1, int npt, double required_percent_inliers,
                          double reprojection_error_allowed, int maximum_tries, int find_best_even_in_failure,
                          double singular_value_ratio_allowed, unsigned long long seed, const int32_t *samples,
                          int32_t *success, double *essential, double *camera, double *inlier_percent,
                          int32_t *inlier_idx, int32_t *n_inliers, int32_t *best_try, int32_t *best_root,
                          int32_t *tries_run, void *stream);

int spv_dlt_triangulate(const double *P0, const double *P1, int npt, const double *x,
                        const double *xp, double *dst);
int spv_dlt_reprojection_error(const double *P0, const double *P1, int npt, const double *x,
                               const double *xp, double *dst);

/* ------------------------------------------------------------------------ */
/* 3. Device-pointer variants: everything resident in HBM, async on `stream` */
/*    (`stream` is a hipStream_t passed as void*; NULL = the null stream).    */
/*    Buffers must belong to the calling thread's current HIP device.         */
/* ------------------------------------------------------------------------ */

/* Scratch bytes needed by spv_l1k2_device for this shape. */
size_t spv_l1k2_workspace_bytes(int xrows, int yrows, int dim);
/* d_x uint8[xrows,dim], d_y uint8[yrows,dim] (16-byte aligned bases),
 * d_idx uint64[yrows,2], d_dist int32[yrows,2], d_ws >= workspace bytes. */
int spv_l1k2_device(const uint8_t *d_x, const uint8_t *d_y, int xrows, int yrows, int dim,
                    uint64_t *d_idx, int32_t *d_dist, void *d_ws, size_t ws_bytes,
                    void *stream);

/* The query loop sharded over the GPUs of one node with everything resident (the loop
 * the reference shards over OpenMP threads, src/BruteForceNnL1K2.h:92-93): one process, rank r =
 * devices[r].  d_x[r] is that device's replica of the database uint8[xrows,dim]; d_y[r] its query
 * shard uint8[spv_shard_lo(total, ndev, r + 1) - spv_shard_lo(total, ndev, r), dim] (contiguous
 * balanced shards of the yrows_total queries, the first total % ndev one row longer).  Every device
 * runs the L1 kernels on its shard and packs (idx0, idx1, d0, d1) into 16-byte records; the records
 * are gathered on devices[0] -- transport SPV_GATHER_RCCL: ncclGather on a cached ncclCommInitAll
 * clique (distinct devices), SPV_GATHER_PEERCOPY: one hipMemcpyPeerAsync per rank -- and widened
 * there into d_idx uint64[yrows_total,2], d_dist int32[yrows_total,2] (memory of devices[0]).
 * Synchronous: returns after all ranks' streams have drained; the caller's buffers must be ready
 * (its own streams synchronised) on entry.  Scratch comes from the library's per-device cache.
 * spv_profile_read("gather") / ("gather_widen") time the exchange. */
int spv_l1k2_gathered_device(int ndev, const int *devices, const uint8_t *const *d_x,
                             const uint8_t *const *d_y, int xrows, long long yrows_total, int dim,
                             uint64_t *d_idx, int32_t *d_dist, int transport);
/* The same for the cascade hash (every device also holds a replica of the hyperplanes d_dict[r],
 * float32[n,dim,m], and rebuilds identical codes and bucket tables; d_ncand int32[yrows_total] on
 * devices[0], may be NULL) and for the DLT (point shards d_x[r], d_xp[r] double[cnt_r,3]; d_dst on
 * devices[0]: double[npt_total,4], or double[npt_total] with want_error != 0; 32 or 8 bytes per point
 * travel, already in the ABI layout).  Loops sharded: src/CascadingHashNn.h:229-245 (the query loop of
 * its BruteForceNnL1K2 with SetFilter), src/Spectavi.cpp:48-51 / :64-67. */
int spv_cascade_gathered_device(int ndev, const int *devices, const float *const *d_x,
                                const float *const *d_y, int xrows, long long yrows_total, int dim, int m,
                                int n, int g, const float *const *d_dict, uint64_t *d_idx, float *d_dist,
                                int32_t *d_ncand, int transport);
int spv_dlt_gathered_device(int ndev, const int *devices, const double *P0, const double *P1,
                            long long npt_total, const double *const *d_x, const double *const *d_xp,
                            double *d_dst, int want_error, int transport);
/* First row of shard r of `total` rows over `shards` contiguous balanced shards (r = shards: total). */
long long spv_shard_lo(long long total, int shards, int r);

/* Scratch bytes needed by spv_cascade_device. */
size_t spv_cascade_workspace_bytes(int xrows, int yrows, int dim, int m, int n, int g);
/* d_x,d_y float32[rows,dim]; d_dict float32[n,dim,m]; outputs as in section 2
 * (d_ncand may be NULL). */
int spv_cascade_device(const float *d_x, const float *d_y, int xrows, int yrows, int dim,
                       int m, int n, int g, const float *d_dict, uint64_t *d_idx,
                       float *d_dist, int32_t *d_ncand, void *d_ws, size_t ws_bytes,
                       void *stream);

/* P0,P1 are HOST pointers (24 doubles, passed by value to the kernel);
 * d_x,d_xp double[npt,3]; d_dst double[npt,4] (triangulate) / double[npt] (error). */
int spv_dlt_triangulate_device(const double *P0, const double *P1, long long npt,
                               const double *d_x, const double *d_xp, double *d_dst,
                               void *stream);
int spv_dlt_reprojection_error_device(const double *P0, const double *P1, long long npt,
                                      const double *d_x, const double *d_xp, double *d_dst,
                                      void *stream);

/* Ratio test + ordered match compaction, the step right after the NN path in the
 * reference's pipeline (example/ex01_essential_estimation.py:102-106): query q passes iff
 * it has a nearest neighbour and (double)dist[q,1] / (double)dist[q,0] >= min_ratio under
 * IEEE division (x/0 = inf passes, 0/0 = NaN fails).  Passing (query, database index)
 * pairs are written in ascending query order.  dist_is_float: 0 = int32 distances
 * (nn_bruteforcel1k2), 1 = float32 (nn_cascading_hash).
 * Host form: matches int32[yrows,2] capacity, returns status; *count receives the number
 * of matches. */
int spv_ratio_test(const uint64_t *idx, const void *dist, int dist_is_float, int yrows,
                   double min_ratio, int32_t *matches, int32_t *count);
size_t spv_ratio_test_workspace_bytes(int yrows);
/* Device form: d_matches int32[yrows,2] capacity, d_count int32[1]. */
int spv_ratio_test_device(const uint64_t *d_idx, const void *d_dist, int dist_is_float, int yrows,
                          double min_ratio, int32_t *d_matches, int32_t *d_count, void *d_ws,
                          size_t ws_bytes, void *stream);

/* SIFT table adapter: rows of 132 float32 = x, y, sigma, angle + 128 descriptor values that
 * are uint8(512*d) stored as float (reference src/Sift.h:13,115-123) are split into
 * geom float32[rows,4] and desc uint8[rows,128] (truncating cast).  Host form. */
int spv_sift_split(const float *table, int rows, float *geom, uint8_t *desc);
int spv_sift_split_device(const float *d_table, int rows, float *d_geom, uint8_t *d_desc,
                          void *stream);
/* (x, y, 1) float64 coordinates of both keypoints of every match (query row, database row)
 * written by spv_ratio_test_device: d_x0[i] from d_geom_x[database row], d_x1[i] from
 * d_geom_y[query row]; rows [0, *d_count) of the [capacity,3] outputs are written
 * (reference example/ex01_essential_estimation.py:104-106). */
int spv_gather_match_coords_device(const float *d_geom_x, const float *d_geom_y,
                                   const int32_t *d_matches, const int32_t *d_count, int capacity,
                                   double *d_x0, double *d_x1, void *stream);

/* normalize_to_ubyte_and_multiple_16_dim (reference spectavi/feature.py:384-407) for a float32
 * input, bit for bit what numpy computes: per-column de-mean (numpy's row-ordered float32 sum),
 * divide by the per-column max-abs, x128, round half-to-even, clip to [-128,127], zero-pad the
 * columns to dim16 = roundup(dim,16).  out_f32 float32[rows,dim16] and/or out_u8 =
 * uint8(out + 128)[rows,dim16] (either may be NULL, not both). */
int spv_normalize(const float *x, int rows, int dim, float *out_f32, uint8_t *out_u8);
size_t spv_normalize_workspace_bytes(int dim);
/* A larger workspace (it grows with rows) with which spv_normalize_device computes the column sums
 * of a table of 65536 rows or more by folding 1024-row chunks instead of walking every row: the same
 * bits, 1.5x (full SIFT tables) to 3.7x (integer-valued tables) faster at a million rows.  With only spv_normalize_workspace_bytes(dim) it walks. */
size_t spv_normalize_workspace_bytes_rows(int rows, int dim);
int spv_normalize_device(const float *d_x, int rows, int dim, float *d_out_f32, uint8_t *d_out_u8,
                         void *d_ws, size_t ws_bytes, void *stream);

/* Device form of spv_dlt_score_hypotheses: P0 is a HOST pointer (12 doubles), d_P1s
 * double[nhyp,12], d_counts int32[nhyp] (zeroed by the call), d_mask uint8[nhyp,npt] or NULL. */
int spv_dlt_score_hypotheses_device(const double *P0, const double *d_P1s, int nhyp,
                                    long long npt, const double *d_x, const double *d_xp,
                                    double max_error, int32_t *d_counts, uint8_t *d_mask,
                                    void *stream);
/* The same with a workspace (spv_dlt_score_workspace_bytes): the solves that do not converge on the
 * fast path are collected in a work list and finished by a second kernel in full waves instead of
 * in place -- same results, about three times the throughput on RANSAC-like hypotheses.  A NULL or
 * short workspace falls back to the one-pass form. */
size_t spv_dlt_score_workspace_bytes(int nhyp, long long npt);
int spv_dlt_score_hypotheses_device_ws(const double *P0, const double *d_P1s, int nhyp, long long npt,
                                       const double *d_x, const double *d_xp, double max_error,
                                       int32_t *d_counts, uint8_t *d_mask, void *d_ws, size_t ws_bytes,
                                       void *stream);

#ifdef __cplusplus
}
#endif

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#endif /* SPECTAVI_AMD_H */
