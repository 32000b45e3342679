// stmmqr_internal.h -- helpers shared by the host-side translation units of libstmmqr_hip.so (not exported: the
// library's version script keeps everything but the C ABI of include/stmmqr_hip.h local).
#pragma once
#include "../../include/stmmqr_hip.h"

extern "C" {
// record the thread's last error (stmmqr_last_error) and return `code`
int stm_fail(int code, const char *msg);
// cc->status = code through the configured sparse_common layout (no-op for cc == NULL)
void stm_cc_set_status(stm_sparse_common *cc, int code);
// SparseCore_malloc / SparseCore_free semantics (src/core/SparseCore_common.c:603-655): counted in cc->malloc_count / memory_inuse
void *stm_cc_malloc(size_t n, size_t size, stm_sparse_common *cc);
void stm_cc_free(size_t n, size_t size, void *p, stm_sparse_common *cc);
}
