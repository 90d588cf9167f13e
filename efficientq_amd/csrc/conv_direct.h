// Internal interface between conv3d.hip (the C-ABI entry conv3d_quant_calib_step) and conv3d_direct.hip.
#pragma once
#include "common.h"

namespace effq {

struct DirectParams {
  const float* x;       // NDHWC
  const float* G;       // reference weight layout [C2][C1][KD*KH*KW]
  const float* bias;
  const float* y;       // NDHWC target
  int N, C1, C2, D, H, W, OD, OH, OW, SD, SH, SW, PD, PH, PW;
  long long V;
  int ntiles;
  double* partials;
  unsigned int* ticket;
  double* sqerr;
};

// 0: not served; 1: 4-channel 3x3x3 conv onto 32 channels; 2: 1x1x1 conv onto <= 4 channels; 3: 1-channel 3x3x3 conv onto 32
int conv_direct_kind(const effq_geom* g);
int conv_direct_launch(int kind, DirectParams& p, size_t max_blocks, hipStream_t st);

}  // namespace effq
