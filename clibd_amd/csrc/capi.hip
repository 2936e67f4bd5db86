// C-ABI housekeeping for libclibd_hip.so: error text, ABI version.
#include "host_util.h"
#include "../../include/clibd_hip.h"

namespace clibd {
char* last_error_buf() {
    static thread_local char buf[kErrBufLen] = {0};
    return buf;
}
}  // namespace clibd

extern "C" const char* clibd_last_error(void) { return clibd::last_error_buf(); }
// 2 (round 4): since 1 the act enum grew (CLIBD_ACT_ADD_AUX, *_U8), entry points were added and clibd_topk_ip's workspace contract
// changed (per-split candidate lists instead of the Q x Nk score matrix): a caller built against 1 must fail loudly, not mis-size.
// 3 (round 5): uint8 patch gather, the LayerNorm -> Linear fold fields of clibd_gemm_epilogue, the adapters' partials workspace.
// 4 (round 5): the 8-bit dgrad entry points (clibd_gemm_fp8_dgrad_nt, clibd_layernorm_bwd_fp8, clibd_quantize_rows_fp8_bf16).
// 5 (round 6): the loss path's fixed-order sums (clibd_softce_workspace_bytes grew), clibd_transpose_colsum_bf16_ws,
//              clibd_layernorm_bwd_fp8_pg (8-bit dgrad with trainable base weights).
extern "C" int clibd_abi_version(void) { return 5; }

// sha256/16 of clibd_amd/csrc/*.{hip,h} + include/clibd_hip.h at build time (clibd_amd/build.py passes it; this unit is rebuilt
// whenever it changes): the Python binding refuses a library that was not built from the sources beside it.
#ifndef CLIBD_CSRC_HASH
#define CLIBD_CSRC_HASH "unknown"
#endif
extern "C" const char* clibd_build_hash(void) { return CLIBD_CSRC_HASH; }
