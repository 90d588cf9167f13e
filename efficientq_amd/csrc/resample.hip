// effq_upsample_trilinear: the x2 trilinear up-sampling between the decoder levels of the 3D-UNets (model_blk.py /
// factory_blk.py:70-93, nn.Upsample(scale_factor, mode='trilinear'), align_corners = False) on NDHWC tensors.
// Glue between quantised convs, but 12 calls of the library kernel took 18 ms of a 0.92 s calibration
// (1.5 ms each: its NCDHW indexing strides through channels-last memory); here one thread produces four channels of an
// output voxel from 16-byte loads: HBM-bound (0.54 GB written for the largest level).
// Arithmetic as in the reference's framework: src = 0.5 (dst + 0.5) - 0.5 clamped at 0 per scaled axis, the eight
// neighbours combined as  l0d (l0h (l0w v000 + l1w v001) + l1h (...)) + l1d (...)  in fp32.
#include "common.h"

namespace effq {

struct UpParams {
  const float* x;
  float* y;
  int N, D, H, W, C, sd, sh, sw;
  int OD, OH, OW;
};

__device__ __forceinline__ void up_axis(int o, int scale, int in, int& i0, int& i1, float& l0, float& l1) {
  if (scale == 1) {
    i0 = i1 = o;
    l0 = 1.0f;
    l1 = 0.0f;
    return;
  }
  float src = (1.0f / (float)scale) * ((float)o + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_upsample_trilinear(UpParams p) {
  const int cv = p.C / VEC;
  const size_t total = (size_t)p.N * p.OD * p.OH * p.OW * cv;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % cv) * VEC;
    size_t r = e / cv;
    const int ow = (int)(r % p.OW);
    r /= p.OW;
    const int oh = (int)(r % p.OH);
    r /= p.OH;
    const int od = (int)(r % p.OD);
    const int n = (int)(r / p.OD);
    int d0, d1, h0, h1, w0, w1;
    float ld0, ld1, lh0, lh1, lw0, lw1;
    up_axis(od, p.sd, p.D, d0, d1, ld0, ld1);
    up_axis(oh, p.sh, p.H, h0, h1, lh0, lh1);
    up_axis(ow, p.sw, p.W, w0, w1, lw0, lw1);
    auto at = [&](int d, int h, int w) { return p.x + ((((size_t)n * p.D + d) * p.H + h) * p.W + w) * p.C + c; };
    const float* q000 = at(d0, h0, w0); const float* q001 = at(d0, h0, w1);
    const float* q010 = at(d0, h1, w0); const float* q011 = at(d0, h1, w1);
    const float* q100 = at(d1, h0, w0); const float* q101 = at(d1, h0, w1);
    const float* q110 = at(d1, h1, w0); const float* q111 = at(d1, h1, w1);
    float out[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float a = ld0 * (lh0 * (lw0 * q000[k] + lw1 * q001[k]) + lh1 * (lw0 * q010[k] + lw1 * q011[k]));
      const float b = ld1 * (lh0 * (lw0 * q100[k] + lw1 * q101[k]) + lh1 * (lw0 * q110[k] + lw1 * q111[k]));
      out[k] = a + b;
    }
    float* dst = p.y + ((((size_t)n * p.OD + od) * p.OH + oh) * p.OW + ow) * p.C + c;
#pragma unroll
    for (int k = 0; k < VEC; ++k) dst[k] = out[k];
  }
}

}  // namespace effq
using namespace effq;

extern "C" {

int effq_upsample_trilinear(const float* x_ndhwc, int N, int D, int H, int W, int C, int sd, int sh, int sw,
                            float* y_ndhwc, void* stream) {
  EFFQ_CHECK_ARG(x_ndhwc && y_ndhwc && N > 0 && D > 0 && H > 0 && W > 0 && C > 0);
  EFFQ_CHECK_ARG((sd == 1 || sd == 2) && (sh == 1 || sh == 2) && (sw == 1 || sw == 2));
  UpParams p;
  p.x = x_ndhwc; p.y = y_ndhwc; p.N = N; p.D = D; p.H = H; p.W = W; p.C = C; p.sd = sd; p.sh = sh; p.sw = sw;
  p.OD = D * sd; p.OH = H * sh; p.OW = W * sw;
  const bool v4 = (C % 4) == 0 && ((reinterpret_cast<uintptr_t>(x_ndhwc) | reinterpret_cast<uintptr_t>(y_ndhwc)) & 15) == 0;
  const size_t total = (size_t)N * p.OD * p.OH * p.OW * (v4 ? C / 4 : C);
  size_t nb = (total + 255) / 256;
  if (nb > 65536) nb = 65536;
  if (v4)
    hipLaunchKernelGGL(k_upsample_trilinear<4>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), p);
  else
    hipLaunchKernelGGL(k_upsample_trilinear<1>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), p);
  EFFQ_LAUNCH_CHECK();
  return EFFQ_OK;
}

}  // extern "C"
