#include "odw_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void odw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

ODW_EXPORT const char* odw_last_error(void) { return g_err; }
ODW_EXPORT int odw_version(void) { return 1; }

// A HIP stream whose kernels run on a SUBSET of the compute units (hipExtStreamCreateWithCUMask): bit i of mask word i / 32
// = CU i may be used.  The step runs the dense losses' early backward on such a stream while the contrastive branch's chain of
// ~70 tiny dependent launches runs beside it on another: with every CU taken by a large GEMM's workgroups each of those
// launches waits for a tile to retire (loss_fused.py).  Caller owns the stream (odw_stream_destroy).
ODW_EXPORT int odw_stream_create_cu_mask(int n_words, const uint32_t* mask, void** stream_out) {
    ODW_REQUIRE(n_words >= 1 && mask && stream_out, "stream_create_cu_mask: bad arguments");
    hipStream_t s = nullptr;
    ODW_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask), "hipExtStreamCreateWithCUMask");
    *stream_out = (void*)s;
    return ODW_OK;
}

ODW_EXPORT int odw_stream_destroy(void* stream) {
    if (!stream) return ODW_OK;
    ODW_CHECK_HIP(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy");
    return ODW_OK;
}
