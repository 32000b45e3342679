// stmmqr_internal.h -- helpers shared by the host-side translation units of libstmmqr_hip.so (not exported: the
// library's version script keeps everything but the C ABI of include/stmmqr_hip.h local).
#pragma once
#include "../../include/stmmqr_hip.h"

extern "C" {
// record the thread's last error (stmmqr_last_error) and return `code`
int stm_fail(int code, const char *msg);
// cc->status = code through the configured sparse_common layout (no-op for cc == NULL)
void stm_cc_set_status(stm_sparse_common *cc, int code);
}
