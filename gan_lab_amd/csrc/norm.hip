// LayerNorm([C, R, R]) of the ResNet-GAN critics (resnetgan/resblocks.py:15-121 via NormalizeLayer('LayerNorm'),
// utils/custom_layers.py:100-107 = nn.LayerNorm, elementwise affine), first AND second order: the WGAN-GP penalty
// (resnetgan/learner.py:780-827) differentiates the critic's input gradient, so the backward of the backward is
// needed.  A sample is one row of M = C*R*R elements; statistics come from ganlab_instnorm_stats_f32(planes = N,
// HW = M) and the affine-free operator P_x(g) = rstd * (g - mean(g) - xhat * mean(g * xhat)) from
// ganlab_instnorm_style_bwd_{reduce,apply}_f32 with a NULL style.  With ghat = gy * w:
//   forward          y  = xhat * w + b
//   backward         gx = P_x(ghat),  gw[m] = sum_n gy * xhat,  gb[m] = sum_n gy
//   backward^2 (cotangent u of gx; P_x is self-adjoint):
//       d/d gy = w * P_x(u)
//       d/d w  = sum_n gy * P_x(u)
//       d/d x  = -rstd^2 * mean(u * t) * xhat - rstd * beta * P_x(u) - rstd * p * gx
//                with t = ghat - a - xhat * beta, a = mean(ghat), beta = mean(ghat * xhat), p = mean(u * xhat)
// The kernels here are the pieces the instance-norm kernels do not provide.  All HBM-bound, one pass each.
#include "common.h"

namespace {

constexpr int EW_MAX_BLOCKS = 256 * 8;
inline unsigned ew_blocks(long long n) {
  const long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > EW_MAX_BLOCKS ? EW_MAX_BLOCKS : b));
}

// y[n,m] = (x[n,m] - mean[n]) * rstd[n] * w[m] + b[m]     (w / b nullable: 1 / 0)
__global__ void ln_affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ w,
                                     const float* __restrict__ b, float* __restrict__ y, long long total, long long M) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / M, m = i - n * M;
    float v = (x[i] - mean[n]) * rstd[n];
    if (w != nullptr) v *= w[m];
    if (b != nullptr) v += b[m];
    y[i] = v;
  }
}

// out[n,m] = a[n,m] * w[m]
__global__ void colscale_kernel(const float* __restrict__ a, const float* __restrict__ w, float* __restrict__ out,
                                long long total, long long M) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = a[i] * w[i % M];
}

// column reductions over the N rows (thread per column: coalesced across m, fixed summation order):
//   o1[m] = sum_n a[n,m] * f[n,m]   with f = (x - mean[n]) * rstd[n] when mean != NULL, else f = x
//   o2[m] = sum_n a[n,m]            (o2 nullable)
__global__ void coldot_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ mean,
                              const float* __restrict__ rstd, float* __restrict__ o1, float* __restrict__ o2, int N,
                              long long M) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  float s1 = 0.f, s2 = 0.f;
  for (int n = 0; n < N; ++n) {
    const float av = a[(long long)n * M + m];
    float f = x[(long long)n * M + m];
    if (mean != nullptr) f = (f - mean[n]) * rstd[n];
    s1 += av * f;
    s2 += av;
  }
  o1[m] = s1;
  if (o2 != nullptr) o2[m] = s2;
}

// out[n] = sum_m a[n,m] * b[n,m] * (w ? w[m] : 1): one 256-thread block per row, fp64 partials (rows of up to 2^18
// elements whose terms cancel)
__global__ void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                              float* __restrict__ out, long long M) {
  __shared__ double red[4];
  const long long n = blockIdx.x;
  const float* ar = a + n * M;
  const float* br = b + n * M;
  double s = 0.0;
  for (long long m = threadIdx.x; m < M; m += 256) {
    float t = ar[m] * br[m];
    if (w != nullptr) t *= w[m];
    s += (double)t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[n] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// out[n,m] = c1[n] * xhat[n,m] + c2[n] * pu[n,m] + c3[n] * gx[n,m]
__global__ void ln_bwdbwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ pu,
                                       const float* __restrict__ gx, const float* __restrict__ c1,
                                       const float* __restrict__ c2, const float* __restrict__ c3,
                                       float* __restrict__ out, long long total, long long M) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / M;
    const float xh = (x[i] - mean[n]) * rstd[n];
    out[i] = c1[n] * xh + c2[n] * pu[i] + c3[n] * gx[i];
  }
}

}  // namespace

#define ST gl_stream(stream)

extern "C" {

int ganlab_ln_affine_fwd_f32(const float* x, const float* mean, const float* rstd, const float* w, const float* b,
                             float* y, int N, long long M, void* stream) {
  if (!x || !mean || !rstd || !y || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(ln_affine_fwd_kernel, dim3(ew_blocks((long long)N * M)), dim3(256), 0, ST, x, mean, rstd, w, b, y,
            (long long)N * M, M);
  return GL_CHECK_LAUNCH();
}

int ganlab_colscale_f32(const float* a, const float* w, float* out, int N, long long M, void* stream) {
  if (!a || !w || !out || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(colscale_kernel, dim3(ew_blocks((long long)N * M)), dim3(256), 0, ST, a, w, out, (long long)N * M, M);
  return GL_CHECK_LAUNCH();
}

int ganlab_coldot_f32(const float* a, const float* x, const float* mean, const float* rstd, float* o1, float* o2,
                      int N, long long M, void* stream) {
  if (!a || !x || !o1 || N <= 0 || M <= 0 || ((mean == nullptr) != (rstd == nullptr))) return GANLAB_EINVAL;
  GL_LAUNCH(coldot_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ST, a, x, mean, rstd, o1, o2, N, M);
  return GL_CHECK_LAUNCH();
}

int ganlab_rowdot_f32(const float* a, const float* b, const float* w, float* out, int N, long long M, void* stream) {
  if (!a || !b || !out || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(rowdot_kernel, dim3((unsigned)N), dim3(256), 0, ST, a, b, w, out, M);
  return GL_CHECK_LAUNCH();
}

int ganlab_ln_bwdbwd_apply_f32(const float* x, const float* mean, const float* rstd, const float* pu, const float* gx,
                               const float* c1, const float* c2, const float* c3, float* out, int N, long long M,
                               void* stream) {
  if (!x || !mean || !rstd || !pu || !gx || !c1 || !c2 || !c3 || !out || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(ln_bwdbwd_apply_kernel, dim3(ew_blocks((long long)N * M)), dim3(256), 0, ST, x, mean, rstd, pu, gx, c1,
            c2, c3, out, (long long)N * M, M);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
