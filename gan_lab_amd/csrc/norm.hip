// LayerNorm([C, R, R]) of the ResNet-GAN critics (resnetgan/resblocks.py:15-121 via NormalizeLayer('LayerNorm'),
// utils/custom_layers.py:100-107 = nn.LayerNorm, elementwise affine), first AND second order: the WGAN-GP penalty
// (resnetgan/learner.py:780-827) differentiates the critic's input gradient, so the backward of the backward is
// needed.  A sample is one row of M = C*R*R elements; statistics come from ganlab_instnorm_stats_f32(planes = N,
// HW = M); the affine-free operator P_x(g) = rstd * (g - mean(g) - xhat * mean(g * xhat)) is ln_rowsums (several
// blocks per row: a critic has only `batch` rows) + ln_project, both with the elementwise weight folded in.
// With ghat = gy * w:
//   forward          y  = xhat * w + b
//   backward         gx = P_x(ghat),  gw[m] = sum_n gy * xhat,  gb[m] = sum_n gy
//   backward^2 (cotangent u of gx; P_x is self-adjoint):
//       d/d gy = w * P_x(u)
//       d/d w  = sum_n gy * P_x(u)
//       d/d x  = -rstd^2 * mean(u * t) * xhat - rstd * beta * P_x(u) - rstd * p * gx
//                with t = ghat - a - xhat * beta, a = mean(ghat), beta = mean(ghat * xhat), p = mean(u * xhat)
// All HBM-bound, one pass each; reductions are deterministic (fixed-order partial sums, fp64 partials).
#include "common.h"

namespace {

constexpr int EW_MAX_BLOCKS = 256 * 8;
inline unsigned ew_blocks(long long n) {
  const long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > EW_MAX_BLOCKS ? EW_MAX_BLOCKS : b));
}

// y[n,m] = act((x[n,m] - mean[n]) * rstd[n] * w[m] + b[m])     (w / b nullable: 1 / 0; act: the LeakyReLU / ReLU that
// follows the normalisation in every residual block, resnetgan/resblocks.py:48-49 - one pass instead of two)
__global__ void ln_affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ w,
                                     const float* __restrict__ b, float* __restrict__ y, long long total, long long M,
                                     int act, float slope) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / M, m = i - n * M;
    float v = (x[i] - mean[n]) * rstd[n];
    if (w != nullptr) v *= w[m];
    if (b != nullptr) v += b[m];
    if (act == GANLAB_ACT_LRELU) v = gl_lrelu(v, slope);
    y[i] = v;
  }
}

// BatchNorm apply in the CENTRED form y = (x - mean[c]) * scale[c] + shift[c]: the folded x*scale + (shift - mean*scale)
// has an absolute rounding error of eps*|x*scale| instead of eps*|y|, i.e. ~10x more activations land on the wrong
// side of the ReLU that follows (seen in the float64-judged ResNet step test)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ scale, const float* __restrict__ shift,
                                float* __restrict__ y, long long total4, int C, long long hw4, int act, float slope) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / hw4) % C);
    const float mu = mean[c], sc = scale[c], sh = shift != nullptr ? shift[c] : 0.f;
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = (v.x - mu) * sc + sh;
    v.y = (v.y - mu) * sc + sh;
    v.z = (v.z - mu) * sc + sh;
    v.w = (v.w - mu) * sc + sh;
    if (act == GANLAB_ACT_LRELU) {
      v.x = gl_lrelu(v.x, slope); v.y = gl_lrelu(v.y, slope);
      v.z = gl_lrelu(v.z, slope); v.w = gl_lrelu(v.w, slope);
    }
    reinterpret_cast<float4*>(y)[i] = v;
  }
}
__global__ void bn_apply1_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                 float* __restrict__ y, long long total, int C, long long HW, int act, float slope) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    const float v = (x[i] - mean[c]) * scale[c] + (shift != nullptr ? shift[c] : 0.f);
    y[i] = act == GANLAB_ACT_LRELU ? gl_lrelu(v, slope) : v;
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

// LayerNorm backward's projection and its parameter gradients in ONE pass over (a, x) - thread per column, rows in order:
//   gx[n,m] = rstd[n] * (a[n,m] * w[m] - s0[n]/M - xhat[n,m] * s1[n]/M)       (= ln_project with wa = w)
//   gw[m] = sum_n a[n,m] * xhat[n,m],   gb[m] = sum_n a[n,m]                    (= coldot, same summation order)
__global__ void ln_bwd_cols_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ x,
                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ sums, float* __restrict__ gx, float* __restrict__ gw,
                                   float* __restrict__ gb, int N, long long M, float inv_len) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float wm = w[m];
  float s1 = 0.f, s2 = 0.f;
  for (int n = 0; n < N; ++n) {
    const long long i = (long long)n * M + m;
    const float av = a[i], rs = rstd[n];
    const float xh = (x[i] - mean[n]) * rs;
    gx[i] = rs * (av * wm - sums[n * 3] * inv_len - xh * sums[n * 3 + 1] * inv_len);
    s1 += av * xh;
    s2 += av;
  }
  gw[m] = s1;
  gb[m] = s2;
}

// Row sums with several blocks per row (a critic LayerNorm has only N = batch rows of up to 2^18 elements):
//   t = a[n,m] * (wa ? wa[m] : 1);  s0 = sum t;  s1 = sum t * xhat;  s2 = sum a * b2 * (w2 ? w2[m] : 1)  (b2 nullable)
// grid = (S, N): block (s, n) reduces the slice [s*len, (s+1)*len) of row n into part[(n*S + s)*3 ..] (fp64);
// rowsums_finish adds the S partials of a row in a fixed order.
constexpr int ROWSUM_MAX_SPLIT = 64;
// A "row" is LayerNorm's sample (one contiguous segment of M elements) or BatchNorm's channel (N segments of HW
// elements, C*HW apart): element j of row r lives at r*row_stride + (j / seg_len)*seg_stride + (j % seg_len).
struct RowGeom {
  long long L, seg_len, seg_stride, row_stride;   // L = elements per row
};
__device__ __forceinline__ long long row_addr(const RowGeom& g, int r, long long j) {
  const long long seg = j / g.seg_len;
  return (long long)r * g.row_stride + seg * g.seg_stride + (j - seg * g.seg_len);
}

__global__ void ln_rowsums_kernel(const float* __restrict__ a, const float* __restrict__ wa,
                                  const float* __restrict__ x, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const float* __restrict__ b2,
                                  const float* __restrict__ w2, double* __restrict__ part, RowGeom g, long long len,
                                  const float* __restrict__ yact, float* __restrict__ gz, float slope) {
  __shared__ double red[3][4];
  const int n = blockIdx.y, sidx = blockIdx.x, S = gridDim.x;
  const long long lo = sidx * len, hi = (lo + len < g.L) ? lo + len : g.L;
  const float mu = mean != nullptr ? mean[n] : 0.f, rs = rstd != nullptr ? rstd[n] : 1.f;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (long long m = lo + threadIdx.x; m < hi; m += 256) {
    const long long i = row_addr(g, n, m);
    float av = a[i];
    if (yact != nullptr) {      // a = gradient at the OUTPUT of the LeakyReLU behind the normalisation: gz = a * lrelu'(y),
      if (!(yact[i] > 0.f)) av *= slope;      // stored for the projection / parameter-gradient passes that follow
      gz[i] = av;
    }
    const float t = wa != nullptr ? av * wa[m] : av;
    s0 += (double)t;
    s1 += (double)t * (double)((x[i] - mu) * rs);
    if (b2 != nullptr) s2 += (double)(av * b2[i] * (w2 != nullptr ? w2[m] : 1.f));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
    red[2][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    part[((long long)n * S + sidx) * 3 + k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}

// moments = 0: out[n][k] = sum_k.  moments = 1 (called with a = x, mean = rstd = NULL, so s0 = sum x, s1 = sum x^2):
// out[n] = {mean, biased variance, 0} evaluated in fp64.
__global__ void ln_rowsums_finish_kernel(const double* __restrict__ part, float* __restrict__ out, int N, int S,
                                         int moments, double inv_len) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s[3] = {0.0, 0.0, 0.0};
  for (int j = 0; j < S; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] += part[((long long)n * S + j) * 3 + k];
  if (moments) {
    const double mu = s[0] * inv_len;
    double var = s[1] * inv_len - mu * mu;
    out[n * 3] = (float)mu;
    out[n * 3 + 1] = (float)(var > 0.0 ? var : 0.0);
    out[n * 3 + 2] = 0.f;
  } else {
    out[n * 3] = (float)s[0];
    out[n * 3 + 1] = (float)s[1];
    out[n * 3 + 2] = (float)s[2];
  }
}

// out = (wo ? wo[m] : 1) * (pre ? pre[n] : rstd[n]) * (a*(wa ? wa[m] : 1) - s0[n]/L - xhat * s1[n]/L),  sums = [N][3]
__global__ void ln_project_kernel(const float* __restrict__ a, const float* __restrict__ wa,
                                  const float* __restrict__ x, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const float* __restrict__ sums,
                                  const float* __restrict__ wo, const float* __restrict__ pre, float* __restrict__ out,
                                  long long total, RowGeom g, float inv_len) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long n = e / g.L, m = e - n * g.L;
    const long long i = row_addr(g, (int)n, m);
    const float rs = rstd[n];
    const float xh = (x[i] - mean[n]) * rs;
    const float t = wa != nullptr ? a[i] * wa[m] : a[i];
    float v = (pre != nullptr ? pre[n] : rs) * (t - sums[n * 3] * inv_len - xh * sums[n * 3 + 1] * inv_len);
    if (wo != nullptr) v *= wo[m];
    out[i] = v;
  }
}

// out[n,m] = c1[n] * xhat[n,m] + c2[n] * pu[n,m] + c3[n] * gx[n,m] with the per-row coefficients of the LayerNorm double
// backward formed here from the two sets of row sums (sums = {sum ghat, sum ghat*xhat, .}, usums = {sum u, sum u*xhat,
// sum u*ghat}; a, beta, ubar, pbar, r = those / M):  c1 = -rstd^2 (r - a ubar - beta pbar), c2 = -rstd beta, c3 = -rstd pbar
__global__ void ln_bwdbwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ pu,
                                       const float* __restrict__ gx, const float* __restrict__ sums,
                                       const float* __restrict__ usums, float* __restrict__ out, long long total,
                                       long long M, float inv) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / M;
    const float rs = rstd[n];
    const float a = sums[n * 3] * inv, beta = sums[n * 3 + 1] * inv;
    const float ubar = usums[n * 3] * inv, pbar = usums[n * 3 + 1] * inv, r = usums[n * 3 + 2] * inv;
    const float mut = r - a * ubar - beta * pbar;
    const float c1 = -(rs * rs) * mut, c2 = -rs * beta, c3 = -rs * pbar;
    const float xh = (x[i] - mean[n]) * rs;
    out[i] = c1 * xh + c2 * pu[i] + c3 * gx[i];
  }
}

// BatchNorm bookkeeping between the statistics and the apply pass in ONE launch (was ~11 C-element ATen launches per
// layer and forward): out = {mean[C], biased var[C], rstd[C], scale[C] = rstd * weight}, the running estimates move by
// ``momentum`` (unbiased variance, nn.BatchNorm2d) and the batch counter by one.
__global__ void bn_finalize_kernel(const float* __restrict__ mom, const float* __restrict__ weight,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ batches, float* __restrict__ out, int C, float eps,
                                   float keep, float momentum, float unbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && batches) batches[0] += 1;
  if (c >= C) return;
  const float mean = mom[c * 3], var = mom[c * 3 + 1];
  const float rstd = rsqrtf(var + eps);
  out[c] = mean;
  out[C + c] = var;
  out[2 * C + c] = rstd;
  out[3 * C + c] = weight ? rstd * weight[c] : rstd;
  if (running_mean) running_mean[c] = running_mean[c] * keep + momentum * mean;
  if (running_var) running_var[c] = running_var[c] * keep + momentum * (var * unbias);
}

}  // namespace

#define ST gl_stream(stream)

extern "C" {

int ganlab_ln_affine_fwd_f32(const float* x, const float* mean, const float* rstd, const float* w, const float* b,
                             float* y, int N, long long M, int act, float slope, void* stream) {
  if (!x || !mean || !rstd || !y || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(ln_affine_fwd_kernel, dim3(ew_blocks((long long)N * M)), dim3(256), 0, ST, x, mean, rstd, w, b, y,
            (long long)N * M, M, act, slope);
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

int ganlab_ln_bwd_cols_f32(const float* a, const float* w, const float* x, const float* mean, const float* rstd,
                           const float* sums, float* gx, float* gw, float* gb, int N, long long M, void* stream) {
  if (!a || !w || !x || !mean || !rstd || !sums || !gx || !gw || !gb || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(ln_bwd_cols_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ST, a, w, x, mean, rstd, sums, gx, gw,
            gb, N, M, 1.0f / (float)M);
  return GL_CHECK_LAUNCH();
}

size_t ganlab_ln_rowsums_workspace(int N, long long M) {
  if (N <= 0 || M <= 0) return 0;
  return (size_t)N * ROWSUM_MAX_SPLIT * 3 * sizeof(double);
}

static int rowsums_launch(const float* a, const float* wa, const float* x, const float* mean, const float* rstd,
                          const float* b2, const float* w2, float* out, int rows, RowGeom g, int moments,
                          void* workspace, size_t workspace_bytes, void* stream, const float* yact = nullptr,
                          float* gz = nullptr, float slope = 1.f) {
  if (!a || !x || !out || rows <= 0 || g.L <= 0 || (yact != nullptr) != (gz != nullptr)) return GANLAB_EINVAL;
  if (!workspace || workspace_bytes < ganlab_ln_rowsums_workspace(rows, g.L)) return GANLAB_EWORKSPACE;
  int S = (int)((g.L + 4095) / 4096);
  if (S > ROWSUM_MAX_SPLIT) S = ROWSUM_MAX_SPLIT;
  if (S < 1) S = 1;
  const long long len = ((g.L + S - 1) / S + 255) / 256 * 256;
  S = (int)((g.L + len - 1) / len);
  double* part = reinterpret_cast<double*>(workspace);
  GL_LAUNCH(ln_rowsums_kernel, dim3((unsigned)S, (unsigned)rows), dim3(256), 0, ST, a, wa, x, mean, rstd, b2, w2, part,
            g, len, yact, gz, slope);
  GL_LAUNCH(ln_rowsums_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ST, part, out, rows, S,
            moments, 1.0 / (double)g.L);
  return GL_CHECK_LAUNCH();
}

int ganlab_ln_rowsums_f32(const float* a, const float* wa, const float* x, const float* mean, const float* rstd,
                          const float* b2, const float* w2, float* out, int N, long long M, void* workspace,
                          size_t workspace_bytes, const float* yact, float* gz, float slope, void* stream) {
  if (!mean || !rstd) return GANLAB_EINVAL;
  return rowsums_launch(a, wa, x, mean, rstd, b2, w2, out, N, RowGeom{M, M, 0, M}, 0, workspace, workspace_bytes,
                        stream, yact, gz, slope);
}

int ganlab_ln_project_f32(const float* a, const float* wa, const float* x, const float* mean, const float* rstd,
                          const float* sums, const float* wo, float* out, int N, long long M, void* stream) {
  if (!a || !x || !mean || !rstd || !sums || !out || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(ln_project_kernel, dim3(ew_blocks((long long)N * M)), dim3(256), 0, ST, a, wa, x, mean, rstd, sums, wo,
            (const float*)nullptr, out, (long long)N * M, RowGeom{M, M, 0, M}, 1.0f / (float)M);
  return GL_CHECK_LAUNCH();
}

// ---- BatchNorm2d (training mode, first order) of the ResNet generators on the same kernels: a row is a channel ---
// (N segments of HW elements).  stats: out[c] = {batch mean, biased batch variance, 0};  bwd_sums: out[c] = {sum gy =
// d/d bias, sum gy * xhat = d/d weight, 0};  bwd_apply: gx = pre[c] * (gy - s0/L - xhat * s1/L) with pre = rstd * w.
int ganlab_bn_stats_f32(const float* x, float* out, int N, int C, long long HW, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  return rowsums_launch(x, nullptr, x, nullptr, nullptr, nullptr, nullptr, out, C, RowGeom{(long long)N * HW, HW, C * HW, HW},
                        1, workspace, workspace_bytes, stream);
}

int ganlab_bn_finalize_f32(const float* mom, const float* weight, float* running_mean, float* running_var,
                           long long* batches, float* out, int C, float eps, float momentum, float unbias,
                           void* stream) {
  if (!mom || !out || C <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(bn_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, ST, mom, weight, running_mean,
            running_var, batches, out, C, eps, (float)(1.0 - (double)momentum), momentum, unbias);
  return GL_CHECK_LAUNCH();
}

int ganlab_bn_apply_f32(const float* x, const float* mean, const float* scale, const float* shift, float* y, int N,
                        int C, long long HW, int act, float slope, void* stream) {
  if (!x || !mean || !scale || !y || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  const long long total = (long long)N * C * HW;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
    GL_LAUNCH(bn_apply_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, ST, x, mean, scale, shift, y, total / 4, C,
              HW / 4, act, slope);
  else
    GL_LAUNCH(bn_apply1_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST, x, mean, scale, shift, y, total, C, HW, act,
              slope);
  return GL_CHECK_LAUNCH();
}

int ganlab_bn_bwd_sums_f32(const float* gy, const float* x, const float* mean, const float* rstd, float* out, int N,
                           int C, long long HW, void* workspace, size_t workspace_bytes, const float* yact, float* gz,
                           float slope, void* stream) {
  if (!mean || !rstd || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  return rowsums_launch(gy, nullptr, x, mean, rstd, nullptr, nullptr, out, C, RowGeom{(long long)N * HW, HW, C * HW, HW},
                        0, workspace, workspace_bytes, stream, yact, gz, slope);
}

int ganlab_bn_bwd_apply_f32(const float* gy, const float* x, const float* mean, const float* rstd, const float* sums,
                            const float* pre, float* gx, int N, int C, long long HW, void* stream) {
  if (!gy || !x || !mean || !rstd || !sums || !pre || !gx || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  const long long L = (long long)N * HW;
  GL_LAUNCH(ln_project_kernel, dim3(ew_blocks(L * C)), dim3(256), 0, ST, gy, (const float*)nullptr, x, mean, rstd, sums,
            (const float*)nullptr, pre, gx, L * C, RowGeom{L, HW, C * HW, HW}, 1.0f / (float)L);
  return GL_CHECK_LAUNCH();
}

int ganlab_ln_bwdbwd_apply_f32(const float* x, const float* mean, const float* rstd, const float* pu, const float* gx,
                               const float* sums, const float* usums, float* out, int N, long long M, void* stream) {
  if (!x || !mean || !rstd || !pu || !gx || !sums || !usums || !out || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(ln_bwdbwd_apply_kernel, dim3(ew_blocks((long long)N * M)), dim3(256), 0, ST, x, mean, rstd, pu, gx, sums,
            usums, out, (long long)N * M, M, 1.0f / (float)M);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
