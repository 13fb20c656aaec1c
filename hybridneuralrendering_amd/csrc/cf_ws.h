// Weight-stationary form of the colour-feature MLP launch (csrc/cf_ws.hip): color_feature_branch 280 -> 128 -> 128 -> 128 + the 128 -> 64 tail
// (models/aggregators/point_aggregators.py:1028-1037, :1199), the (18, 8, 8, 8) k-step instance of hnr_mlp3_forward.
#pragma once
#include "hnr_h2.h"

namespace hnr {

struct CfWsArgs {
    const float *A; int lda;           // [M, lda] input rows (X5: 256 K-summed features + 24 view-direction encodings), K0 columns used
    const char *wimg; int wbase[4];    // packed weights of hnr_mlp3_pack (k steps 18, 8, 8, 8) + meta
    int K0;
    float slope;
    const unsigned long long *counts; int count_index; long long M_cap;     // M = min(M_cap, counts[count_index]) (counts may be NULL)
    float *C; int ldc;                 // [M, ldc] colour feature (128 columns, LeakyReLU applied)
    float *C2; int ldc2;               // [M, ldc2] tail (64 columns, no activation)
};

// Returns HNR_OK after queueing the kernel; the caller checked the configuration (n_layers = 4, k steps (18, 8, 8, 8), act = {1, 1, 1, 0}, no addend,
// no kept activations, no segments, M_cap * ld < 2^29 elements).
int launch_cf_ws(const CfWsArgs &a, hipStream_t st);

}  // namespace hnr
