// Separable table-driven 2-D resampler: the generator's bilinear 2x upsample and the critic's nearest / bilinear 0.5x
// pooling (reference: gan_lab/utils/custom_layers.py:59-75, nn.Upsample(mode='bilinear') built at
// gan_lab/resnetgan/learner.py:147-170), forward and adjoint through ONE kernel:
//
//   y[p, oy, ox] = sum_{a < Ty} sum_{b < Tx}  wy[oy, a] * wx[ox, b] * x[p, iy[oy, a], ix[ox, b]]
//
// The host builds (iy, wy) / (ix, wx) once per (mode, align_corners, size): T = 1 or 2 taps per output for the forward
// interpolation matrix M, and the rows of M^T (<= 6 taps; zero-weight padding) for the adjoint, so the backward of the
// layer - and its double backward, which is the forward again - are gathers with a fixed summation order: no atomics.
// The nearest 2x upsample and the 2x2 average keep their own streaming kernels (pointwise.hip) and their folds into the
// stride-2 conv kernels; these variants are off the benchmark configurations, a plain streaming pass per call.
#include "common.h"

namespace {

constexpr int RS_MAXT = 6;

// one thread per 4 consecutive outputs of a row: the x taps of neighbouring outputs overlap, so the row gathers hit L1
__global__ __launch_bounds__(256) void resample2d_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const int* __restrict__ iy, const float* __restrict__ wy,
                                                         const int* __restrict__ ix, const float* __restrict__ wx,
                                                         long long planes, int Hi, int Wi, int Ho, int Wo, int Ty,
                                                         int Tx) {
  const int wq = (Wo + 3) >> 2;
  const long long total = planes * Ho * wq;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % wq);
    const long long t = i / wq;
    const int oy = (int)(t % Ho);
    const long long pl = t / Ho;
    const float* xp = x + pl * Hi * Wi;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < Ty; ++a) {
      const float wa = wy[oy * Ty + a];
      const float* xr = xp + (long long)iy[oy * Ty + a] * Wi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = 4 * q + j;
        if (ox < Wo) {
          float r = 0.f;
          for (int b = 0; b < Tx; ++b) r = fmaf(wx[ox * Tx + b], xr[ix[ox * Tx + b]], r);
          acc[j] = fmaf(wa, r, acc[j]);
        }
      }
    }
    float* yo = y + (pl * Ho + oy) * Wo + 4 * q;
    if ((Wo & 3) == 0) {
      *reinterpret_cast<float4*>(yo) = float4{acc[0], acc[1], acc[2], acc[3]};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * q + j < Wo) yo[j] = acc[j];
    }
  }
}

}  // namespace

extern "C" {

int ganlab_resample2d_f32(const float* x, float* y, const int* iy, const float* wy, const int* ix, const float* wx,
                          long long planes, int Hi, int Wi, int Ho, int Wo, int Ty, int Tx, void* stream) {
  if (!x || !y || !iy || !wy || !ix || !wx || planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || Ty <= 0 ||
      Tx <= 0 || Ty > RS_MAXT || Tx > RS_MAXT)
    return GANLAB_EINVAL;
  const long long items = planes * Ho * ((Wo + 3) >> 2);
  const long long blocks = (items + 255) / 256;
  GL_LAUNCH(resample2d_kernel, dim3((unsigned)(blocks < 65536 * 8 ? blocks : 65536 * 8)), dim3(256), 0,
            gl_stream(stream), x, y, iy, wy, ix, wx, planes, Hi, Wi, Ho, Wo, Ty, Tx);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
