/*
 * NdArray.h -- C-allocated n-d array out-parameter used by the Spectavi C-ABI.
 *
 * The reference includes <NdArray.h> from its `ctypes_ndarray` git submodule
 * (reference .gitmodules:1-3, src/EigenDefinitions.h:22), which is EMPTY in the
 * reference tree, so its field layout is unknowable offline.  What the reference
 * call sites pin (and this header honours):
 *   - a struct type `NdArray` with a data member `m_data`
 *     (src/Spectavi.cpp:107,136,237,265,292-293,333-334),
 *   - `ndarray_set_size(NdArray*, d0, d1)` and a 3-dim form (src/Spectavi.cpp:103,134,288),
 *   - `ndarray_alloc(NdArray*)` allocating d0*d1*...*itemsize bytes, the item
 *     size having been fixed by the Python side (`NdArray(dtype='uint64')`,
 *     spectavi/feature.py:301-302).
 * This header and spectavi_amd/ndarray.py are a matched pair (see INTEGRATION.md
 * for what to change when linking against the upstream cndarray package).
 */
#ifndef SPECTAVI_AMD_NDARRAY_H
#define SPECTAVI_AMD_NDARRAY_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define NDARRAY_MAX_DIMS 4

typedef struct NdArray {
  void *m_data;                      /* malloc'ed by ndarray_alloc, freed by ndarray_free */
  size_t m_shape[NDARRAY_MAX_DIMS];  /* row-major extents */
  int32_t m_ndim;                    /* number of valid extents */
  int32_t m_itemsize;                /* bytes per element, set by the Python constructor
                                        (byte offset 44 of this 48-byte struct on LP64) */
} NdArray;

/* C has no overloading: the 2-/3-extent forms used by the reference are
 * provided as ndarray_set_size (2-d, the only form the hot path uses) and
 * ndarray_set_size3. */
void ndarray_set_size(NdArray *arr, size_t d0, size_t d1);
void ndarray_set_size3(NdArray *arr, size_t d0, size_t d1, size_t d2);
/* Allocates prod(shape)*itemsize bytes (at least 1) with malloc; returns 0 on
 * success, nonzero on failure (m_data is then NULL). */
int ndarray_alloc(NdArray *arr);
/* Releases m_data (called by the Python object's finaliser). */
void ndarray_free(NdArray *arr);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#endif /* SPECTAVI_AMD_NDARRAY_H */
