/* oracle/stmmqr_binding.c -- TEST INFRASTRUCTURE: the binding stub of INTEGRATION.md 2, verbatim, compiled with the
 * reference's headers into the relinked test artefact (oracle/_ref/refapi_relinked). */
#include <stddef.h>
#include "SparseQR.h"          /* the reference's header */
typedef struct { size_t status, malloc_count, memory_usage, memory_inuse, blas_ok,
                        SPQR_grain, SPQR_small, SPQR_shrink, SPQR_flopcount, SPQR_flopcount_bound; } stm_common_layout;
extern void stmmqr_set_common_layout(const stm_common_layout *);
__attribute__((constructor)) static void stmmqr_bind(void)
{
    stm_common_layout L = {
        offsetof(sparse_common, status),        offsetof(sparse_common, malloc_count),
        offsetof(sparse_common, memory_usage),  offsetof(sparse_common, memory_inuse),
        offsetof(sparse_common, blas_ok),       offsetof(sparse_common, SPQR_grain),
        offsetof(sparse_common, SPQR_small),    offsetof(sparse_common, SPQR_shrink),
        offsetof(sparse_common, SPQR_flopcount),offsetof(sparse_common, SPQR_flopcount_bound) };
    stmmqr_set_common_layout(&L);
}
