// Private glue between the translation units of libemx (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/emx.h"

struct EmxChainView {
    double* chain;       // (stored, N, D) device-resident chain, or nullptr
    double* chain_lp;    // (stored, N)
    int64_t N;
    int32_t D;
    int64_t stored;
    hipStream_t stream;
    int device;
};

// implemented in emx.hip
int emx_internal_chain_view(emx_ctx* c, EmxChainView* v);
int emx_internal_state_view(emx_ctx* c, const double** X, int64_t* N, int32_t* D, int* device);      // settles and synchronises the context first
int emx_internal_fail(emx_ctx* c, int code, const char* msg);      // records the message for emx_last_error, returns code
