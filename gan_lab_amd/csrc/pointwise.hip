// HBM-bound elementwise / resampling / reduction kernels of the G+D step (fp32, NCHW).
// Every kernel is a single pass over its tensors with 16-byte accesses where the row length allows
// (cdna_hip_programming.md G13) and a grid capped at ~8 blocks/CU with a grid-stride loop (G11).
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int kMaxBlocks = 256 * 8;

inline unsigned ew_blocks(long long n_items) {
  long long b = (n_items + 255) / 256;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (unsigned)b;
}

// Chunk c of `chunks` over a plane of hw4 float4: CONTIGUOUS ranges of whole 256-float4 rows (a block streams one
// 4 KB .. 64 KB run instead of 4 KB pieces 256 KB apart: +5..12 % on the 1R1W / 2R1W passes, tools/pw_probe.py)
struct ChunkRange { long long begin, end, stride; };
__device__ __forceinline__ ChunkRange chunk_range(long long hw4, int chunks, int c, int contig) {
  ChunkRange r;
  if (contig) {
    const long long rows = (hw4 + 255) / 256, per = (rows + chunks - 1) / chunks;
    r.begin = (long long)c * per * 256;
    r.end = r.begin + per * 256;
    if (r.end > hw4) r.end = hw4;
    r.stride = 256;
  } else {      // legacy interleaved form (A/B: GANLAB_PW_CONTIG=0)
    r.begin = (long long)c * 256;
    r.end = hw4;
    r.stride = (long long)chunks * 256;
  }
  return r;
}
inline int pw_contig() {
  static const int v = [] { const char* e = getenv("GANLAB_PW_CONTIG"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}

#define GRID_STRIDE(i, n) \
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------------------------- //
// blur: depthwise [1 2 1]x[1 2 1]/16, zero padding (custom_layers.py:41-51)
// ---------------------------------------------------------------------------------------------- //
__global__ void blur3x3_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W) {
  const long long total = planes * H * W;
  GRID_STRIDE(i, total) {
    const int w = (int)(i % W);
    const long long t = i / W;
    const int h = (int)(t % H);
    const float* px = x + (t - h) * W;  // plane base
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = h + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      const float* row = px + (long long)yy * W;
      const float l = (w > 0) ? row[w - 1] : 0.f;
      const float c = row[w];
      const float r = (w + 1 < W) ? row[w + 1] : 0.f;
      const float rs = l + 2.f * c + r;
      acc += (dy == 0) ? 2.f * rs : rs;
    }
    y[i] = acc * (1.f / 16.f);
  }
}

// One row segment (4 columns at x0) of the blur input with its two horizontal neighbours, [1 2 1]-filtered.
// The neighbours come from the adjacent lanes (lane-1 holds columns x0-4..x0-1 of the same row whenever
// q > 0) - only the first / last lane of a wave touches memory for them.  MASKED: the operand is
// in * lrelu'(m) (the masked centre values are returned in cen).
// LeakyReLU masks as BITS (1 = the activation was positive): bit e of the NCHW-linear element index e, 32 per word.  A
// pass that only needs sign(y) reads 1/32 of the bytes of y (the critic's conv -> LeakyReLU -> blur keeps no other use for
// y in its backward: progan/architectures.py:261-284).  Rows must be whole words: W % 32 == 0.
__device__ __forceinline__ unsigned mask_nibble(const unsigned* __restrict__ bits, long long e) {   // e % 4 == 0
  return (bits[e >> 5] >> (unsigned)(e & 31)) & 0xFu;
}
__device__ __forceinline__ unsigned mask_bit(const unsigned* __restrict__ bits, long long e) {
  return (bits[e >> 5] >> (unsigned)(e & 31)) & 1u;
}

template <bool MASKED, bool BITS = false>
__device__ __forceinline__ void blur_row(const float* __restrict__ in, const float* __restrict__ m, long long ro,
                                         int x0, int q, int w4, bool valid, float slope, float (&hrow)[4],
                                         float (&cen)[4], const unsigned* __restrict__ mb = nullptr,
                                         long long mplane = 0) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) {
    v = *reinterpret_cast<const float4*>(in + ro + x0);
    if (MASKED) {
      if (BITS) {
        const unsigned nb = mask_nibble(mb, mplane + ro + x0);
        v.x = (nb & 1u) ? v.x : v.x * slope;
        v.y = (nb & 2u) ? v.y : v.y * slope;
        v.z = (nb & 4u) ? v.z : v.z * slope;
        v.w = (nb & 8u) ? v.w : v.w * slope;
      } else {
        const float4 mm = *reinterpret_cast<const float4*>(m + ro + x0);
        v.x = mm.x > 0.f ? v.x : v.x * slope;
        v.y = mm.y > 0.f ? v.y : v.y * slope;
        v.z = mm.z > 0.f ? v.z : v.z * slope;
        v.w = mm.w > 0.f ? v.w : v.w * slope;
      }
    }
  }
  const int lane = threadIdx.x & 63;
  float l = __shfl_up(v.w, 1, 64);
  float r = __shfl_down(v.x, 1, 64);
  // the wave's two outer neighbours: lane 0 fetches the element left of its segment, lane 63 the one right of it -
  // ONE divergent load (two when masked) serves both
  const bool need_l = lane == 0 && q != 0, need_r = lane == 63 && q != w4 - 1;
  float e = 0.f;
  if (valid && (need_l || need_r)) {
    const long long eo = ro + x0 + (need_l ? -1 : 4);
    e = in[eo];
    if (MASKED) {
      if (BITS) e = mask_bit(mb, mplane + eo) ? e : e * slope;
      else e = m[eo] > 0.f ? e : e * slope;
    }
  }
  if (q == 0) l = 0.f;
  else if (lane == 0) l = e;
  if (q == w4 - 1) r = 0.f;
  else if (lane == 63) r = e;
  hrow[0] = l + 2.f * v.x + v.y;
  hrow[1] = v.x + 2.f * v.y + v.z;
  hrow[2] = v.y + 2.f * v.z + v.w;
  hrow[3] = v.z + 2.f * v.w + r;
  cen[0] = v.x; cen[1] = v.y; cen[2] = v.z; cen[3] = v.w;
}

// vectorised blur: one thread = 4 columns x R rows (W % 4 == 0, H % R == 0): R+2 row segments are read once
// as float4 (edges by lane shuffle) and combined separably
template <int R, bool BITS = false>
__global__ __launch_bounds__(256) void blur3x3_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long long planes, int H, int W,
                                                          unsigned* __restrict__ bits = nullptr) {
  const int w4 = W >> 2, hr = H / R;
  const long long total = planes * hr * w4;
  // all lanes of a wave run the same number of iterations (the shuffles need their neighbours alive)
  const long long span = (long long)gridDim.x * blockDim.x;
  for (long long base = blockIdx.x * (long long)blockDim.x; base < total; base += span) {
    const long long i = base + threadIdx.x;
    const bool live = i < total;
    const long long ii = live ? i : total - 1;
    const int q = (int)(ii % w4);
    const long long t = ii / w4;
    const int rr = (int)(t % hr);
    const long long pl = t / hr;
    const float* px = x + pl * H * W;
    const int y0 = R * rr, x0 = 4 * q;
    float h[R + 2][4], cen[4];
    [[maybe_unused]] unsigned nib[R];
#pragma unroll
    for (int k = 0; k < R + 2; ++k) {
      const int yy = y0 - 1 + k;
      blur_row<false>(px, nullptr, (long long)yy * W, x0, q, w4, (unsigned)yy < (unsigned)H, 0.f, h[k], cen);
      if constexpr (BITS) {
        if (k >= 1 && k <= R)
          nib[k - 1] = (cen[0] > 0.f ? 1u : 0u) | (cen[1] > 0.f ? 2u : 0u) | (cen[2] > 0.f ? 4u : 0u) | (cen[3] > 0.f ? 8u : 0u);
      }
    }
    if constexpr (BITS) {
      // BITS: the sign bits of the INPUT's own R rows, for the LeakyReLU backward of whoever produced it.  W % 32 == 0, so
      // 8 consecutive lanes hold 32 consecutive pixels of one row and a group of 8 is live or dead as a whole
      const int sh = 4 * (threadIdx.x & 7);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        unsigned wv = nib[k] << sh;
        wv |= __shfl_xor(wv, 1, 64);
        wv |= __shfl_xor(wv, 2, 64);
        wv |= __shfl_xor(wv, 4, 64);
        if (live && (threadIdx.x & 7) == 0) bits[(pl * H * W + (long long)(y0 + k) * W + x0) >> 5] = wv;
      }
    }
    if (live) {
      float* py = y + pl * H * W + (long long)y0 * W + x0;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        float4 o;
        o.x = (h[k][0] + 2.f * h[k + 1][0] + h[k + 2][0]) * 0.0625f;
        o.y = (h[k][1] + 2.f * h[k + 1][1] + h[k + 2][1]) * 0.0625f;
        o.z = (h[k][2] + 2.f * h[k + 1][2] + h[k + 2][2]) * 0.0625f;
        o.w = (h[k][3] + 2.f * h[k + 1][3] + h[k + 2][3]) * 0.0625f;
        *reinterpret_cast<float4*>(py + (long long)k * W) = o;
      }
    }
  }
}

__global__ void up2_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W,
                           float scale) {
  const int Wo = 2 * W, Ho = 2 * H;
  const long long total = planes * Ho * Wo;
  GRID_STRIDE(i, total) {
    const int w = (int)(i % Wo);
    const long long t = i / Wo;
    const int h = (int)(t % Ho);
    const long long pl = t / Ho;
    y[i] = scale * x[(pl * H + (h >> 1)) * W + (w >> 1)];
  }
}

__global__ void pool2_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int Ho, int Wo,
                             float scale) {
  const int W = 2 * Wo;
  const long long total = planes * Ho * Wo;
  GRID_STRIDE(i, total) {
    const int w = (int)(i % Wo);
    const long long t = i / Wo;  // = pl*Ho + h
    const float2 a = *reinterpret_cast<const float2*>(x + (2 * t) * W + 2 * w);
    const float2 b = *reinterpret_cast<const float2*>(x + (2 * t + 1) * W + 2 * w);
    y[i] = scale * ((a.x + a.y) + (b.x + b.y));
  }
}

// ---------------------------------------------------------------------------------------------- //
// blur fused with its pointwise neighbours (same 4xR-outputs-per-thread scheme; grid (chunks, C) so the
// per-channel sums of the backward come out of the same pass):
//   BF_FWD: out = act(blur(in) + noise_w[c]*noise[n,hw] + bias[c]*bias_scale)        G layer forward
//   BF_A  : out = lrelu'(y) * blur(in);           sum0[c] = sum out                   D backward (blur^T, then act')
//   BF_AT : out = blur(lrelu'(y) * in);           sum0[c] = sum lrelu'(y)*in, sum1[c] = sum lrelu'(y)*in*noise
//           (adjoint of BF_A; the backward of BF_FWD)
// ---------------------------------------------------------------------------------------------- //
enum { BF_FWD = 0, BF_A = 1, BF_AT = 2 };

// STATS (BF_FWD only): grid (chunks, C, N) - a block stays inside ONE (n, c) plane and also accumulates sum / sum of
// squares of its outputs in fp64 (the InstanceNorm statistics of the next op: stylegan/architectures.py:524-526 reads
// the tensor this kernel writes); block partials go to spart[((n*C + c)*chunks + chunk)*2 + {0,1}].
template <int MODE, int R, bool STATS = false, bool BITS = false>   // BITS: y is a bit mask (mask_nibble), modes BF_A / BF_AT
__global__ __launch_bounds__(256) void blur_fused_kernel(const float* __restrict__ in, const float* __restrict__ y,
                                                         const float* __restrict__ noise,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ noise_w, float* __restrict__ out,
                                                         double* __restrict__ part, int N, int C, int H, int W,
                                                         int chunks, float bias_scale, int act, float slope,
                                                         int want_sums, double* __restrict__ spart = nullptr) {
  __shared__ double red[4];
  __shared__ double dred[2][4];
  const int c = blockIdx.y, chunk = blockIdx.x;
  const int w4 = W >> 2, hr = H / R;
  const long long per_n = (long long)hr * w4, total = STATS ? per_n : (long long)N * per_n, HW = (long long)H * W;
  double ds = 0.0, dss = 0.0;
  const float b = (MODE == BF_FWD && bias) ? bias[c] * bias_scale : 0.f;
  const float nw = (MODE == BF_FWD && noise) ? noise_w[c] : 0.f;
  double s0 = 0.0, s1 = 0.0;        // bias / noise-weight gradient partials in fp64
  for (long long base = chunk * 256LL; base < total; base += (long long)chunks * 256) {
    const long long i = base + threadIdx.x;
    const bool live = i < total;
    const long long ii = live ? i : total - 1;
    const long long n = STATS ? (long long)blockIdx.z : ii / per_n, rem = STATS ? ii : ii - n * per_n;
    const int rr = (int)(rem / w4), q = (int)(rem - (long long)rr * w4);
    const long long plane = (n * C + c) * HW;
    const int y0 = R * rr, x0 = 4 * q;
    float h[R + 2][4], cen[R + 2][4];
#pragma unroll
    for (int k = 0; k < R + 2; ++k) {
      const int yy = y0 - 1 + k;
      blur_row<MODE == BF_AT, BITS>(in + plane, (MODE == BF_AT && !BITS) ? y + plane : nullptr, (long long)yy * W, x0, q,
                                    w4, (unsigned)yy < (unsigned)H, slope, h[k], cen[k],
                                    reinterpret_cast<const unsigned*>(y), plane);
    }
    if (!live) continue;
    const long long co = (long long)y0 * W + x0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (h[k][j] + 2.f * h[k + 1][j] + h[k + 2][j]) * 0.0625f;
      if (MODE == BF_FWD) {
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (noise) *reinterpret_cast<float4*>(nz) = *reinterpret_cast<const float4*>(noise + n * HW + co + k * W);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = o[j] + b + nw * nz[j];
          if (act == GANLAB_ACT_LRELU) t = gl_lrelu(t, slope);
          o[j] = t;
          if (STATS) {
            const double d = (double)t;
            ds += d;
            dss += d * d;
          }
        }
      } else if (MODE == BF_A) {
        if constexpr (BITS) {
          const unsigned nb = mask_nibble(reinterpret_cast<const unsigned*>(y), plane + co + (long long)k * W);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = ((nb >> j) & 1u) ? o[j] : o[j] * slope;
            s0 += (double)o[j];
          }
        } else {
          float m[4];
          *reinterpret_cast<float4*>(m) = *reinterpret_cast<const float4*>(y + plane + co + k * W);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = m[j] > 0.f ? o[j] : o[j] * slope;
            s0 += (double)o[j];
          }
        }
      } else {
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (noise && want_sums)
          *reinterpret_cast<float4*>(nz) = *reinterpret_cast<const float4*>(noise + n * HW + co + k * W);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0 += (double)cen[k + 1][j];
          s1 += (double)cen[k + 1][j] * nz[j];
        }
      }
      *reinterpret_cast<float4*>(out + plane + co + (long long)k * W) = *reinterpret_cast<float4*>(o);
    }
  }
  if (STATS) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ds += __shfl_xor(ds, o, 64);
      dss += __shfl_xor(dss, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      dred[0][threadIdx.x >> 6] = ds;
      dred[1][threadIdx.x >> 6] = dss;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      const int k = threadIdx.x;
      spart[(((long long)blockIdx.z * C + c) * chunks + chunk) * 2 + k] = (dred[k][0] + dred[k][1]) + (dred[k][2] + dred[k][3]);
    }
  }
  if (MODE != BF_FWD && want_sums) {
    s0 = gl_block_sum_256d(s0, red);
    if (threadIdx.x == 0) part[(long long)c * chunks + chunk] = s0;
    if (MODE == BF_AT) {
      __syncthreads();
      s1 = gl_block_sum_256d(s1, red);
      if (threadIdx.x == 0) part[((long long)C + c) * chunks + chunk] = s1;
    }
  }
}

// InstanceNorm + style backward, LeakyReLU' and blur^T in ONE pass (the backward of a BLURRED generator layer's tail:
// stylegan/architectures.py:497-526 behind Upsample -> conv -> blur).  Round 2 ran instnorm_bwd_apply_act_kernel (read gy, x;
// write gz) and then the blur (read gz, write out); here the blur's operand is evaluated on the fly,
//   z = k * (gy - a1 - xhat * a2) * lrelu'(x),   out = blur(z),   sum0[c] += z,  sum1[c] += z * noise,
// with the 4-columns x R-rows-per-thread scheme of blur_fused_kernel (row halos re-read from L1 / L2): 2R + 1W instead of
// 3R + 2W.  grid (chunks, C, N): a block stays inside one (n, c) plane, whose five scalars it reads once.  z is evaluated in
// fp64 and rounded once (see instnorm_bwd_apply_act_kernel); the sums add the unrounded values of the thread's OWN rows.
// part[(c*N + n)*chunks + chunk] (bias), + C*N*chunks (noise weight).
template <int R>
__global__ __launch_bounds__(256) void instnorm_bwd_act_blur_kernel(
    const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ style, const float* __restrict__ s1,
    const float* __restrict__ s2, const float* __restrict__ noise, float* __restrict__ out, double* __restrict__ part,
    int N, int C, int H, int W, int chunks, int act, float slope, int want_b, int want_nw) {
  __shared__ double red[4];
  const int c = blockIdx.y, n = blockIdx.z, chunk = blockIdx.x;
  const long long pl = (long long)n * C + c, HW = (long long)H * W;
  const int w4 = W >> 2, hr = H / R;
  const long long per_n = (long long)hr * w4;
  const double md = (double)mean[pl], rd = (double)rstd[pl];
  const double ys = style ? (double)style[((long long)n * 2 + 0) * C + c] + 1.0 : 1.0;
  const double inv = 1.0 / (double)HW;
  const double k = rd * ys, a1 = (double)s1[pl] * inv, a2 = (double)s2[pl] * inv;
  const float* pg = gy + pl * HW;
  const float* px = x + pl * HW;
  const float* pn = (want_nw && noise) ? noise + (long long)n * HW : nullptr;
  float* po = out + pl * HW;
  const int lane = threadIdx.x & 63;
  double sb = 0.0, snw = 0.0;
  auto zval = [&](float g, float v) -> double {
    double t = k * ((double)g - a1 - ((double)v - md) * rd * a2);
    if (act == GANLAB_ACT_LRELU && !(v > 0.f)) t *= (double)slope;
    return t;
  };
  for (long long base = chunk * 256LL; base < per_n; base += (long long)chunks * 256) {
    const long long i = base + threadIdx.x;
    const bool live = i < per_n;
    const long long ii = live ? i : per_n - 1;
    const int rr = (int)(ii / w4), q = (int)(ii - (long long)rr * w4);
    const int y0 = R * rr, x0 = 4 * q;
    float h[R + 2][4];
#pragma unroll
    for (int kk = 0; kk < R + 2; ++kk) {
      const int yy = y0 - 1 + kk;
      const bool valid = (unsigned)yy < (unsigned)H;
      const long long ro = (long long)yy * W;
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        float g[4], v[4];
        *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(pg + ro + x0);
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(px + ro + x0);
        const bool own = live && kk >= 1 && kk <= R;
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (own && pn) *reinterpret_cast<float4*>(nz) = *reinterpret_cast<const float4*>(pn + ro + x0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double t = zval(g[j], v[j]);
          z[j] = (float)t;
          if (own) {
            sb += t;
            snw += t * (double)nz[j];
          }
        }
      }
      // horizontal neighbours from the adjacent lanes; the wave's two outer ones evaluate the element beyond their segment
      float l = __shfl_up(z[3], 1, 64);
      float r = __shfl_down(z[0], 1, 64);
      const bool need_l = lane == 0 && q != 0, need_r = lane == 63 && q != w4 - 1;
      float e = 0.f;
      if (valid && (need_l || need_r)) {
        const long long eo = ro + x0 + (need_l ? -1 : 4);
        e = (float)zval(pg[eo], px[eo]);
      }
      if (q == 0) l = 0.f;
      else if (lane == 0) l = e;
      if (q == w4 - 1) r = 0.f;
      else if (lane == 63) r = e;
      h[kk][0] = l + 2.f * z[0] + z[1];
      h[kk][1] = z[0] + 2.f * z[1] + z[2];
      h[kk][2] = z[1] + 2.f * z[2] + z[3];
      h[kk][3] = z[2] + 2.f * z[3] + r;
    }
    if (!live) continue;
#pragma unroll
    for (int kk = 0; kk < R; ++kk) {
      float4 o;
      o.x = (h[kk][0] + 2.f * h[kk + 1][0] + h[kk + 2][0]) * 0.0625f;
      o.y = (h[kk][1] + 2.f * h[kk + 1][1] + h[kk + 2][1]) * 0.0625f;
      o.z = (h[kk][2] + 2.f * h[kk + 1][2] + h[kk + 2][2]) * 0.0625f;
      o.w = (h[kk][3] + 2.f * h[kk + 1][3] + h[kk + 2][3]) * 0.0625f;
      *reinterpret_cast<float4*>(po + (long long)(y0 + kk) * W + x0) = o;
    }
  }
  const long long slot = ((long long)c * N + n) * chunks + chunk;
  if (want_b) {
    sb = gl_block_sum_256d(sb, red);
    if (threadIdx.x == 0) part[slot] = sb;
  }
  if (want_nw) {
    if (want_b) __syncthreads();
    snw = gl_block_sum_256d(snw, red);
    if (threadIdx.x == 0) part[(long long)C * N * chunks + slot] = snw;
  }
}

// rows per thread: R + 2 input rows are read for R output rows (the halo rows come from L1 / L2), so taller strips
// mean less cache traffic: 8 rows for the large planes, 4 / 2 for the small ones
// (measured, 32x16x1024^2 / 32x64x256^2: forward 1.13 -> 1.02 ms / 0.27 -> 0.28, blur-then-act' 1.41 -> 1.20 / 0.36 ->
// 0.33, act'-then-blur 1.91 -> 2.26 / 0.36 -> 0.40 - its 134 registers cost occupancy - so 8 rows only where they won)
inline int blur_rows(int H) { return (H & 3) == 0 ? 4 : 2; }
inline int blur_rows_mode(int H, int mode) {
  const int min8 = mode == BF_FWD ? 512 : (mode == BF_A ? 256 : (1 << 30));
  return ((H & 7) == 0 && H >= min8) ? 8 : blur_rows(H);
}
#define BLUR_FUSED_LAUNCH(MODE, ...)                                                                   \
  do {                                                                                                 \
    const int rows_ = blur_rows_mode(H, MODE);                                                         \
    if (rows_ == 8) GL_LAUNCH((blur_fused_kernel<MODE, 8>), dim3(chunks, C), dim3(256), 0, ST, __VA_ARGS__);      \
    else if (rows_ == 4) GL_LAUNCH((blur_fused_kernel<MODE, 4>), dim3(chunks, C), dim3(256), 0, ST, __VA_ARGS__); \
    else GL_LAUNCH((blur_fused_kernel<MODE, 2>), dim3(chunks, C), dim3(256), 0, ST, __VA_ARGS__);                \
  } while (0)

#define BLUR_FUSED_LAUNCH_BITS(MODE, ...)                                                                            \
  do {                                                                                                               \
    const int rows_ = blur_rows_mode(H, MODE);                                                                       \
    if (rows_ == 8) GL_LAUNCH((blur_fused_kernel<MODE, 8, false, true>), dim3(chunks, C), dim3(256), 0, ST, __VA_ARGS__);      \
    else if (rows_ == 4) GL_LAUNCH((blur_fused_kernel<MODE, 4, false, true>), dim3(chunks, C), dim3(256), 0, ST, __VA_ARGS__); \
    else GL_LAUNCH((blur_fused_kernel<MODE, 2, false, true>), dim3(chunks, C), dim3(256), 0, ST, __VA_ARGS__);                \
  } while (0)

inline int blur_fused_chunks(int N, int H, int W) {
  long long c = ((long long)N * (H / blur_rows(H)) * (W / 4) + 256 * 4 - 1) / (256 * 4);
  if (c < 1) c = 1;
  if (c > 128) c = 128;
  return (int)c;
}

// ---------------------------------------------------------------------------------------------- //
// bias / noise / activation that also accumulates the InstanceNorm statistics of its output: grid (chunks, N*C), a block
// stays inside one plane; spart[(plane*chunks + chunk)*2 + {0,1}] = sum y, sum y^2 (fp64).  HW % 4 == 0.
// ---------------------------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void bias_act_stats_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                             const float* __restrict__ noise,
                                                             const float* __restrict__ noise_w, float* __restrict__ y,
                                                             double* __restrict__ spart, int C, long long hw4,
                                                             int chunks, float bias_scale, int act, float slope,
                                                             int contig) {
  __shared__ double dred[2][4];
  const long long pl = blockIdx.y;
  const int c = (int)(pl % C);
  const long long n = pl / C;
  const float b = bias ? bias[c] * bias_scale : 0.f;
  const float nw = noise ? noise_w[c] : 0.f;
  const float4* xp = reinterpret_cast<const float4*>(x) + pl * hw4;
  const float4* np = noise ? reinterpret_cast<const float4*>(noise) + n * hw4 : nullptr;
  float4* yp = reinterpret_cast<float4*>(y) + pl * hw4;
  double ds = 0.0, dss = 0.0;
  const ChunkRange cr = chunk_range(hw4, chunks, blockIdx.x, contig);
  for (long long i = cr.begin + threadIdx.x; i < cr.end; i += cr.stride) {
    float v[4], nz[4] = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<float4*>(v) = xp[i];
    if (np) *reinterpret_cast<float4*>(nz) = np[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = v[k] + b + nw * nz[k];
      if (act == GANLAB_ACT_LRELU) t = gl_lrelu(t, slope);
      v[k] = t;
      const double d = (double)t;
      ds += d;
      dss += d * d;
    }
    yp[i] = *reinterpret_cast<float4*>(v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ds += __shfl_xor(ds, o, 64);
    dss += __shfl_xor(dss, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    dred[0][threadIdx.x >> 6] = ds;
    dred[1][threadIdx.x >> 6] = dss;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = threadIdx.x;
    spart[(pl * chunks + blockIdx.x) * 2 + k] = (dred[k][0] + dred[k][1]) + (dred[k][2] + dred[k][3]);
  }
}

// mean / rstd of every plane from the chunk partials (fixed order, fp64)
__global__ void act_stats_finish_kernel(const double* __restrict__ spart, float* __restrict__ mean,
                                        float* __restrict__ rstd, long long planes, int chunks, double inv_hw, float eps) {
  const long long pl = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (pl >= planes) return;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s += spart[(pl * chunks + k) * 2];
    ss += spart[(pl * chunks + k) * 2 + 1];
  }
  const double m = s * inv_hw;
  double var = ss * inv_hw - m * m;
  if (var < 0.0) var = 0.0;
  mean[pl] = (float)m;
  rstd[pl] = (float)(1.0 / sqrt(var + (double)eps));
}

// partial sums of one chunk of one row: block (row, chunk), fp64 sums of exact float squares, [row][chunk][2]
__global__ __launch_bounds__(256) void row_stats_partial_kernel(const float* __restrict__ x, double* __restrict__ spart,
                                                                int chunks, long long m4) {
  __shared__ double red[2][4];
  const long long row = blockIdx.x / chunks;
  const int k = blockIdx.x % chunks;
  const long long per = (m4 + chunks - 1) / chunks, lo = k * per, hi = lo + per < m4 ? lo + per : m4;
  const float4* p4 = reinterpret_cast<const float4*>(x) + row * m4;
  double s = 0.0, ss = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 v = p4[i];
    const double a = (double)v.x, b = (double)v.y, c = (double)v.z, d = (double)v.w;
    s += (a + b) + (c + d);
    ss += (a * a + b * b) + (c * c + d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    spart[((long long)blockIdx.x) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    spart[((long long)blockIdx.x) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

inline int act_stats_chunks(long long hw4) {
  long long c = (hw4 + 256 * 8 - 1) / (256 * 8);      // ~8 float4 per thread
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  return (int)c;
}

// ---------------------------------------------------------------------------------------------- //
// y = act(x + noise_w[c]*noise[n,hw] + bias[c]*bias_scale)
// ---------------------------------------------------------------------------------------------- //
template <int VEC>
__global__ void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                const float* __restrict__ noise, const float* __restrict__ noise_w,
                                float* __restrict__ y, int N, int C, long long HW, float bias_scale, int act,
                                float slope) {
  const long long hwv = HW / VEC, total = (long long)N * C * hwv;
  GRID_STRIDE(i, total) {
    const long long pl = i / hwv, hv = i - pl * hwv;
    const int c = (int)(pl % C);
    const long long n = pl / C;
    const float b = bias ? bias[c] * bias_scale : 0.f;
    const float nw = noise ? noise_w[c] : 0.f;
    float v[VEC], nz[VEC];
    if (VEC == 4) {
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(x + i * 4);
      if (noise) *reinterpret_cast<float4*>(nz) = *reinterpret_cast<const float4*>(noise + (n * hwv + hv) * 4);
    } else {
      v[0] = x[i];
      if (noise) nz[0] = noise[n * HW + hv];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float t = v[k] + b;
      if (noise) t += nw * nz[k];
      if (act == GANLAB_ACT_LRELU) t = gl_lrelu(t, slope);
      v[k] = t;
    }
    if (VEC == 4) *reinterpret_cast<float4*>(y + i * 4) = *reinterpret_cast<float4*>(v);
    else y[i] = v[0];
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gz,
                               long long n, float slope) {
  const long long n4 = n >> 2;
  GRID_STRIDE(i, n4) {
    const float4 g = reinterpret_cast<const float4*>(gy)[i];
    const float4 o = reinterpret_cast<const float4*>(y)[i];
    float4 r;
    r.x = o.x > 0.f ? g.x : g.x * slope;
    r.y = o.y > 0.f ? g.y : g.y * slope;
    r.z = o.z > 0.f ? g.z : g.z * slope;
    r.w = o.w > 0.f ? g.w : g.w * slope;
    reinterpret_cast<float4*>(gz)[i] = r;
  }
  // tail
  const long long base = n4 << 2;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (tid < n - base) {
    const long long j = base + tid;
    gz[j] = y[j] > 0.f ? gy[j] : gy[j] * slope;
  }
}

// ---------------------------------------------------------------------------------------------- //
// per-channel sums: out[c] = scale * sum_{n,hw} a[n,c,hw] * (b ? b[n,hw] : 1)
// stage 1: grid (chunks, C); stage 2: one wave per channel
// ---------------------------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void channel_sum_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                          double* __restrict__ part, int N, int C, long long HW,
                                                          int chunks) {
  __shared__ double red[4];
  const int c = blockIdx.y, chunk = blockIdx.x;
  const long long total = (long long)N * HW;
  double s = 0.0;       // fp64 from the first addition on: these sums run over 1e5..1e7 terms that mostly cancel
  if ((HW & 3) == 0 && (total >> 2) < 0x7fffffffLL) {
    // 16-byte loads and 32-bit index arithmetic (the scalar form with a 64-bit division per element ran at 2.8 TB/s)
    const unsigned hw4 = (unsigned)(HW >> 2), total4 = (unsigned)(total >> 2);
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (unsigned i = chunk * 256u + threadIdx.x; i < total4; i += (unsigned)chunks * 256u) {
      const unsigned n = i / hw4, q = i - n * hw4;
      const float4 v = a4[((long long)n * C + c) * hw4 + q];
      if (b) {
        const float4 w = b4[i];
        s += ((double)v.x * w.x + (double)v.y * w.y) + ((double)v.z * w.z + (double)v.w * w.w);
      } else {
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      }
    }
  } else {
    for (long long i = chunk * 256LL + threadIdx.x; i < total; i += (long long)chunks * 256) {
      const long long n = i / HW, hw = i - n * HW;
      const float v = a[(n * C + c) * HW + hw];
      s += b ? (double)v * b[i] : (double)v;
    }
  }
  s = gl_block_sum_256d(s, red);
  if (threadIdx.x == 0) part[(long long)c * chunks + chunk] = s;
}

__global__ void channel_sum_stage2(const double* __restrict__ part, float* __restrict__ out, int C, int chunks,
                                   float scale) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < chunks; i += 64) s += part[(long long)c * chunks + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[c] = (float)(s * (double)scale);
}

// gz = gy * lrelu'(y) AND the per-channel sums of gz (bias gradient) in one pass:
// grid (chunks, C); partial sums go to part[c][chunk], channel_sum_stage2 finishes them.
__global__ __launch_bounds__(256) void act_bwd_bias_stage1(const float* __restrict__ gy, const float* __restrict__ y,
                                                           float* __restrict__ gz, double* __restrict__ part, int N,
                                                           int C, long long HW, int chunks, float slope) {
  __shared__ double red[4];
  const int c = blockIdx.y, chunk = blockIdx.x;
  double s = 0.0;
  if ((HW & 3) == 0) {
    const long long hw4 = HW >> 2, total = (long long)N * hw4;
    for (long long i = chunk * 256LL + threadIdx.x; i < total; i += (long long)chunks * 256) {
      const long long n = i / hw4, q = i - n * hw4;
      const long long off = ((n * C + c) * hw4 + q);
      const float4 g = reinterpret_cast<const float4*>(gy)[off];
      const float4 o = reinterpret_cast<const float4*>(y)[off];
      float4 r;
      r.x = o.x > 0.f ? g.x : g.x * slope;
      r.y = o.y > 0.f ? g.y : g.y * slope;
      r.z = o.z > 0.f ? g.z : g.z * slope;
      r.w = o.w > 0.f ? g.w : g.w * slope;
      reinterpret_cast<float4*>(gz)[off] = r;
      s += ((double)r.x + (double)r.y) + ((double)r.z + (double)r.w);
    }
  } else {
    const long long total = (long long)N * HW;
    for (long long i = chunk * 256LL + threadIdx.x; i < total; i += (long long)chunks * 256) {
      const long long n = i / HW, hw = i - n * HW;
      const long long off = (n * C + c) * HW + hw;
      const float r = y[off] > 0.f ? gy[off] : gy[off] * slope;
      gz[off] = r;
      s += (double)r;
    }
  }
  s = gl_block_sum_256d(s, red);
  if (threadIdx.x == 0) part[(long long)c * chunks + chunk] = s;
}

inline int channel_chunks(int N, long long HW) {
  long long c = ((long long)N * HW + 256 * 32 - 1) / (256 * 32);
  if (c < 1) c = 1;
  if (c > 128) c = 128;
  return (int)c;
}

// ---------------------------------------------------------------------------------------------- //
// InstanceNorm (biased var, eps) + AdaIN affine
// ---------------------------------------------------------------------------------------------- //
template <int T>  // threads cooperating on one plane: 64 or 256
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const float* __restrict__ x, float* __restrict__ mean,
                                                             float* __restrict__ rstd, long long planes,
                                                             long long HW, float eps) {
  __shared__ double red[2][4];
  const int sub = (T == 64) ? (threadIdx.x >> 6) : 0;
  const int t = (T == 64) ? (threadIdx.x & 63) : threadIdx.x;
  const long long pl = (T == 64) ? (blockIdx.x * 4LL + sub) : blockIdx.x;
  double s = 0.0, ss = 0.0;
  if (pl < planes) {
    const float* p = x + pl * HW;
    if ((HW & 3) == 0) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      for (long long i = t; i < (HW >> 2); i += T) {
        const float4 v = p4[i];
        // squares and sums in fp64: a float product is exact in double, so var = E[x^2] - mean^2 does
        // not suffer the fp32 cancellation (planes with |mean| >> std: e.g. the constant 4x4 input)
        const double a = (double)v.x, b = (double)v.y, c = (double)v.z, d = (double)v.w;
        s += (a + b) + (c + d);
        ss += (a * a + b * b) + (c * c + d * d);
      }
    } else {
      for (long long i = t; i < HW; i += T) {
        const double v = (double)p[i];
        s += v;
        ss += v * v;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  if (T == 256) {
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = s;
      red[1][threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
  if (t == 0 && pl < planes) {
    const double m = s / (double)HW;
    double var = ss / (double)HW - m * m;
    if (var < 0.0) var = 0.0;
    mean[pl] = (float)m;
    rstd[pl] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <int VEC>
__global__ void instnorm_style_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, const float* __restrict__ style,
                                          float* __restrict__ y, int N, int C, long long HW) {
  const long long hwv = HW / VEC, total = (long long)N * C * hwv;
  GRID_STRIDE(i, total) {
    const long long pl = i / hwv;
    const int c = (int)(pl % C);
    const long long n = pl / C;
    const float m = mean[pl], r = rstd[pl];
    const float ys = style ? style[(n * 2 + 0) * C + c] + 1.f : 1.f;
    const float yb = style ? style[(n * 2 + 1) * C + c] : 0.f;
    const float a = r * ys;
    // subtract the mean FIRST: a constant plane must normalise to exactly 0 like the reference's
    // (x - mean) / sqrt(var + eps), not to rounding noise amplified by rstd = 1e4
    if (VEC == 4) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      v.x = (v.x - m) * a + yb; v.y = (v.y - m) * a + yb; v.z = (v.z - m) * a + yb; v.w = (v.w - m) * a + yb;
      reinterpret_cast<float4*>(y)[i] = v;
    } else {
      y[i] = (x[i] - m) * a + yb;
    }
  }
}

// Large planes (HW a multiple of 1024 * U): one block per (plane, chunk of 256 * U float4) - the plane's four scalars are
// read once per block, the index math is 32-bit, and every thread keeps U independent 16-byte loads in flight (the
// grid-stride form above spends a 64-bit division per float4 and has one load in flight per thread).
template <int U>
__global__ __launch_bounds__(256) void instnorm_style_fwd_chunk_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd,
                                                                       const float* __restrict__ style,
                                                                       float* __restrict__ y, int C, int chunks,
                                                                       long long HW) {
  const unsigned pl = blockIdx.x / (unsigned)chunks, ch = blockIdx.x % (unsigned)chunks;
  const int c = (int)(pl % (unsigned)C);
  const long long n = pl / (unsigned)C;
  const float m = mean[pl], r = rstd[pl];
  const float ys = style ? style[(n * 2 + 0) * C + c] + 1.f : 1.f;
  const float yb = style ? style[(n * 2 + 1) * C + c] : 0.f;
  const float a = r * ys;
  const long long base = (long long)pl * (HW / 4) + (long long)ch * (256 * U) + threadIdx.x;
  const float4* xs = reinterpret_cast<const float4*>(x) + base;
  float4* yd = reinterpret_cast<float4*>(y) + base;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = xs[u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float4 w = v[u];
    w.x = (w.x - m) * a + yb; w.y = (w.y - m) * a + yb; w.z = (w.z - m) * a + yb; w.w = (w.w - m) * a + yb;
    yd[u * 256] = w;
  }
}

// ---------------------------------------------------------------------------------------------- //
// The same two passes for the generator's LAST layer, whose only reader is toRGB (stylegan/architectures.py torgb, a 1x1
// conv to <= 4 channels): its incoming gradient  gy[n, c] = sum_k wp[k][c] * grgb[n, k]  is cheaper to recompute from the
// image gradient (3 planes per image) than to write as C planes and read back twice.  A workgroup owns a pixel range of ONE
// image and walks the C <= 16 channels itself, so the image gradient is read once (a plane-per-workgroup grid re-reads it
// through eight L2s: measured slower than reading gy).  gy's bits are pw_few_to_many_kernel's (gl_few_dot).
// ---------------------------------------------------------------------------------------------- //
constexpr int RGB_MAXC = 16;
struct RgbSrc {
  const float* grgb;   // (N, crgb, HW)
  const float* wp;     // toRGB's input-gradient pack: wp[k * cout_p + c]
  int crgb, cout_p;
};
__device__ __forceinline__ void rgb_src_load(const RgbSrc& r, long long n, long long hw4, long long i, float4 (&xv)[4]) {
  const float4* gb = reinterpret_cast<const float4*>(r.grgb) + n * r.crgb * hw4 + i;
#pragma unroll
  for (int k = 0; k < 4; ++k) xv[k] = k < r.crgb ? gb[(long long)k * hw4] : float4{0.f, 0.f, 0.f, 0.f};
}

// grid (chunks, N); part[((n*C + c)*chunks + chunk)*2 + {0, 1}] = this chunk's share of sum gy, sum gy * xhat
// EXACT: C == 16, crgb == 3 (the shape toRGB has on the timed path) as compile-time constants - with run-time bounds every
// channel and every image plane is a branch
template <bool EXACT>
__global__ __launch_bounds__(256) void instnorm_bwd_reduce_rgb_kernel(RgbSrc rgb, const float* __restrict__ x,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ rstd,
                                                                      double* __restrict__ part, int C_, long long hw4,
                                                                      int chunks, int contig) {
  __shared__ double red[4];
  const int C = EXACT ? RGB_MAXC : C_;
  if (EXACT) rgb.crgb = 3;
  const long long n = blockIdx.y;
  const float4* xb = reinterpret_cast<const float4*>(x) + n * C * hw4;      // (offsets inside one image fit 32 bits)
  double a[RGB_MAXC], b[RGB_MAXC];
#pragma unroll
  for (int c = 0; c < RGB_MAXC; ++c) a[c] = b[c] = 0.0;
  const ChunkRange cr = chunk_range(hw4, chunks, blockIdx.x, contig);
  for (long long i = cr.begin + threadIdx.x; i < cr.end; i += cr.stride) {
    float4 xv[4];
    rgb_src_load(rgb, n, hw4, i, xv);
#pragma unroll
    for (int c = 0; c < RGB_MAXC; ++c) {
      if (c < C) {
        const long long pl = n * C + c;
        const float m = mean[pl], r = rstd[pl];
        const float4 g = gl_few_dot(xv, rgb.wp, rgb.crgb, rgb.cout_p, c, 0.f);
        const float4 v = xb[(unsigned)(c * hw4 + i)];
        a[c] += ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
        b[c] += ((double)g.x * ((v.x - m) * r) + (double)g.y * ((v.y - m) * r)) +
                ((double)g.z * ((v.z - m) * r) + (double)g.w * ((v.w - m) * r));
      }
    }
  }
#pragma unroll
  for (int c = 0; c < RGB_MAXC; ++c) {
    if (c < C) {
      const double sa = gl_block_sum_256d(a[c], red);
      __syncthreads();
      const double sb = gl_block_sum_256d(b[c], red);
      __syncthreads();
      if (threadIdx.x == 0) {
        double* dst = part + (((n * C + c) * chunks) + blockIdx.x) * 2;
        dst[0] = sa;
        dst[1] = sb;
      }
    }
  }
}

__global__ void instnorm_bwd_reduce_rgb_finish_kernel(const double* __restrict__ part, float* __restrict__ s1,
                                                      float* __restrict__ s2, long long planes, int chunks) {
  const long long pl = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (pl >= planes) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < chunks; ++k) {       // fixed order
    a += part[(pl * chunks + k) * 2];
    b += part[(pl * chunks + k) * 2 + 1];
  }
  s1[pl] = (float)a;
  s2[pl] = (float)b;
}

// instnorm_bwd_apply_act_kernel with the channel loop inside; grid (chunks, N), the same partial-sum slots
template <bool EXACT>
__global__ __launch_bounds__(256) void instnorm_bwd_apply_act_rgb_kernel(
    RgbSrc rgb, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ style, const float* __restrict__ s1, const float* __restrict__ s2,
    const float* __restrict__ noise, float* __restrict__ gz, double* __restrict__ part, int N, int C_, long long hw4,
    int chunks, int act_, float slope, int want_b, int want_nw, int contig) {
  __shared__ double red[4];
  __shared__ double cst[RGB_MAXC][5];          // k, a1, a2, mean, rstd of this image's planes
  const int C = EXACT ? RGB_MAXC : C_;
  const int act = EXACT ? GANLAB_ACT_LRELU : act_;
  if (EXACT) rgb.crgb = 3;
  const long long n = blockIdx.y;
  const float4* xb = reinterpret_cast<const float4*>(x) + n * C * hw4;
  float4* zb = reinterpret_cast<float4*>(gz) + n * C * hw4;
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    const long long pl = n * C + c;
    const float r = rstd[pl];
    const float ys = style ? style[(n * 2 + 0) * C + c] + 1.f : 1.f;
    const double inv = 1.0 / (double)(hw4 * 4);
    cst[c][0] = (double)r * (double)ys;
    cst[c][1] = (double)s1[pl] * inv;
    cst[c][2] = (double)s2[pl] * inv;
    cst[c][3] = (double)mean[pl];
    cst[c][4] = (double)r;
  }
  __syncthreads();
  const float4* pn = (want_nw && noise) ? reinterpret_cast<const float4*>(noise) + n * hw4 : nullptr;
  double sb[RGB_MAXC], snw[RGB_MAXC];
#pragma unroll
  for (int c = 0; c < RGB_MAXC; ++c) sb[c] = snw[c] = 0.0;
  const ChunkRange cr = chunk_range(hw4, chunks, blockIdx.x, contig);
  for (long long i = cr.begin + threadIdx.x; i < cr.end; i += cr.stride) {
    float4 xv[4];
    rgb_src_load(rgb, n, hw4, i, xv);
    float4 z = float4{0.f, 0.f, 0.f, 0.f};
    if (pn) z = pn[i];
    // (an offset the compiler cannot see through: hoisted out of this loop the 80 constants would sit in 160 registers)
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const double* cp = &cst[0][0] + opaque;
#pragma unroll
    for (int c = 0; c < RGB_MAXC; ++c) {
      if (c < C) {
        const double k = cp[c * 5], a1 = cp[c * 5 + 1], a2 = cp[c * 5 + 2], md = cp[c * 5 + 3], rd = cp[c * 5 + 4];
        float g[4], v[4], o[4];
        double td[4];
        *reinterpret_cast<float4*>(g) = gl_few_dot(xv, rgb.wp, rgb.crgb, rgb.cout_p, c, 0.f);
        *reinterpret_cast<float4*>(v) = xb[(unsigned)(c * hw4 + i)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double t = k * ((double)g[j] - a1 - ((double)v[j] - md) * rd * a2);
          if (act == GANLAB_ACT_LRELU && !(v[j] > 0.f)) t *= (double)slope;
          td[j] = t;
          o[j] = (float)t;
        }
        zb[(unsigned)(c * hw4 + i)] = *reinterpret_cast<float4*>(o);
        sb[c] += (td[0] + td[1]) + (td[2] + td[3]);
        if (pn) snw[c] += (td[0] * z.x + td[1] * z.y) + (td[2] * z.z + td[3] * z.w);
      }
      // four channels' loads in flight at a time: unbounded, the scheduler hoists all 16 and the kernel needs 360 registers
      if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int c = 0; c < RGB_MAXC; ++c) {
    if (c < C) {
      const long long slot = ((long long)c * N + n) * chunks + blockIdx.x;
      if (want_b) {
        const double v = gl_block_sum_256d(sb[c], red);
        __syncthreads();
        if (threadIdx.x == 0) part[slot] = v;
      }
      if (want_nw) {
        const double v = gl_block_sum_256d(snw[c], red);
        __syncthreads();
        if (threadIdx.x == 0) part[(long long)C * N * chunks + slot] = v;
      }
    }
  }
}

template <int T>
__global__ __launch_bounds__(256) void instnorm_bwd_reduce_kernel(const float* __restrict__ gy,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd,
                                                                  float* __restrict__ s1, float* __restrict__ s2,
                                                                  long long planes, long long HW) {
  __shared__ double red[2][4];
  const int sub = (T == 64) ? (threadIdx.x >> 6) : 0;
  const int t = (T == 64) ? (threadIdx.x & 63) : threadIdx.x;
  const long long pl = (T == 64) ? (blockIdx.x * 4LL + sub) : blockIdx.x;
  double a = 0.0, b = 0.0;
  if (pl < planes) {
    const float m = mean[pl], r = rstd[pl];
    const float* pg = gy + pl * HW;
    const float* px = x + pl * HW;
    if ((HW & 3) == 0) {
      for (long long i = t; i < (HW >> 2); i += T) {
        const float4 g = reinterpret_cast<const float4*>(pg)[i];
        const float4 v = reinterpret_cast<const float4*>(px)[i];
        a += ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
        b += ((double)g.x * ((v.x - m) * r) + (double)g.y * ((v.y - m) * r)) +
             ((double)g.z * ((v.z - m) * r) + (double)g.w * ((v.w - m) * r));
      }
    } else {
      for (long long i = t; i < HW; i += T) {
        a += (double)pg[i];
        b += (double)pg[i] * ((px[i] - m) * r);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (T == 256) {
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = a;
      red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
  if (t == 0 && pl < planes) {
    s1[pl] = (float)a;
    s2[pl] = (float)b;
  }
}

template <int VEC>
__global__ void instnorm_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                          const float* __restrict__ style, const float* __restrict__ s1,
                                          const float* __restrict__ s2, float* __restrict__ gx, int N, int C,
                                          long long HW) {
  const long long hwv = HW / VEC, total = (long long)N * C * hwv;
  const float inv = 1.f / (float)HW;
  GRID_STRIDE(i, total) {
    const long long pl = i / hwv;
    const int c = (int)(pl % C);
    const long long n = pl / C;
    const float m = mean[pl], r = rstd[pl];
    const float ys = style ? style[(n * 2 + 0) * C + c] + 1.f : 1.f;
    const float k = r * ys, a1 = s1[pl] * inv, a2 = s2[pl] * inv;
    if (VEC == 4) {
      const float4 g = reinterpret_cast<const float4*>(gy)[i];
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 o;
      o.x = k * (g.x - a1 - (v.x - m) * r * a2);
      o.y = k * (g.y - a1 - (v.y - m) * r * a2);
      o.z = k * (g.z - a1 - (v.z - m) * r * a2);
      o.w = k * (g.w - a1 - (v.w - m) * r * a2);
      reinterpret_cast<float4*>(gx)[i] = o;
    } else {
      gx[i] = k * (gy[i] - a1 - (x[i] - m) * r * a2);
    }
  }
}

// InstanceNorm+style backward fused with the backward of the LeakyReLU / bias / noise in front of it (the generator
// layer tail, stylegan/architectures.py:497-526): x is at once the InstanceNorm input and the LeakyReLU output, so
//   gz = [k*(gy - a1 - xhat*a2)] * lrelu'(x)      and its channel sums  sum gz (bias),  sum gz*noise (noise weight)
// come out of ONE pass.  grid (chunks, N*C); part[(c*N + n)*chunks + chunk] (+ C*N*chunks for the noise sums).
__global__ __launch_bounds__(256) void instnorm_bwd_apply_act_kernel(
    const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ style, const float* __restrict__ s1,
    const float* __restrict__ s2, const float* __restrict__ noise, float* __restrict__ gz, double* __restrict__ part,
    int N, int C, long long hw4, int chunks, int act, float slope, int want_b, int want_nw, int contig) {
  __shared__ double red[4];
  const long long pl = blockIdx.y;
  const int c = (int)(pl % C);
  const long long n = pl / C;
  const float m = mean[pl], r = rstd[pl];
  const float ys = style ? style[(n * 2 + 0) * C + c] + 1.f : 1.f;
  // The element formula runs in fp64 and is rounded ONCE, to the float that is stored; the bias / noise-weight sums add
  // the unrounded values.  A bias in front of an InstanceNorm has an analytically ZERO gradient (sum over the plane of
  // g - mean g - xhat * mean(g xhat)): summing float-rounded elements leaves sqrt(HW) * eps of noise per plane where the
  // reference's own fp32 path leaves the same kind - in fp64 the cancellation is carried out.  ~10 fp64 operations per
  // element of an HBM-bound pass (78 TFLOP/s of fp64 vector rate against 0.5 G elements per 2 GiB tensor).
  const double inv = 1.0 / (double)(hw4 * 4);
  const double k = (double)r * (double)ys, a1 = (double)s1[pl] * inv, a2 = (double)s2[pl] * inv, md = (double)m, rd = (double)r;
  const float4* pg = reinterpret_cast<const float4*>(gy) + pl * hw4;
  const float4* px = reinterpret_cast<const float4*>(x) + pl * hw4;
  const float4* pn = (want_nw && noise) ? reinterpret_cast<const float4*>(noise) + n * hw4 : nullptr;
  float4* po = reinterpret_cast<float4*>(gz) + pl * hw4;
  double sb = 0.0, snw = 0.0;      // bias / noise-weight gradients: fp64 partials (see gl_block_sum_256d)
  const ChunkRange cr = chunk_range(hw4, chunks, blockIdx.x, contig);
  for (long long i = cr.begin + threadIdx.x; i < cr.end; i += cr.stride) {
    float g[4], v[4], o[4];
    double td[4];
    *reinterpret_cast<float4*>(g) = pg[i];
    *reinterpret_cast<float4*>(v) = px[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double t = k * ((double)g[j] - a1 - ((double)v[j] - md) * rd * a2);
      if (act == GANLAB_ACT_LRELU && !(v[j] > 0.f)) t *= (double)slope;
      td[j] = t;
      o[j] = (float)t;
    }
    po[i] = *reinterpret_cast<float4*>(o);
    sb += (td[0] + td[1]) + (td[2] + td[3]);
    if (pn) {
      const float4 z = pn[i];
      snw += (td[0] * z.x + td[1] * z.y) + (td[2] * z.z + td[3] * z.w);
    }
  }
  const long long slot = ((long long)c * N + n) * chunks + blockIdx.x;
  if (want_b) {
    sb = gl_block_sum_256d(sb, red);
    if (threadIdx.x == 0) part[slot] = sb;
  }
  if (want_nw) {
    if (want_b) __syncthreads();
    snw = gl_block_sum_256d(snw, red);
    if (threadIdx.x == 0) part[(long long)C * N * chunks + slot] = snw;
  }
}

// ---------------------------------------------------------------------------------------------- //
// PixelNorm over channels (custom_layers.py:85-86); one thread per pixel, coalesced across pixels
// ---------------------------------------------------------------------------------------------- //
__global__ void pixelnorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, long long HW,
                                     float eps) {
  const long long total = (long long)N * HW;
  GRID_STRIDE(i, total) {
    const long long n = i / HW, hw = i - n * HW;
    const float* p = x + n * C * HW + hw;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = p[c * HW];
      s += v * v;
    }
    const float r = rsqrtf(s / (float)C + eps);
    float* q = y + n * C * HW + hw;
    for (int c = 0; c < C; ++c) q[c * HW] = p[c * HW] * r;
  }
}

__global__ void pixelnorm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                     float* __restrict__ gx, int N, int C, long long HW, float eps) {
  const long long total = (long long)N * HW;
  GRID_STRIDE(i, total) {
    const long long n = i / HW, hw = i - n * HW;
    const float* p = x + n * C * HW + hw;
    const float* g = gy + n * C * HW + hw;
    float s = 0.f, d = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = p[c * HW];
      s += v * v;
      d += v * g[c * HW];
    }
    const float r = rsqrtf(s / (float)C + eps);
    const float k = r * r * r * d / (float)C;
    float* q = gx + n * C * HW + hw;
    for (int c = 0; c < C; ++c) q[c * HW] = g[c * HW] * r - p[c * HW] * k;
  }
}

// The same two kernels with a pixel's channels held in registers (C <= 128): the generic ones walk the channels twice and
// the second walk misses the caches on the large maps (256 pixels x C x 4 bytes per workgroup in between), so the
// forward moved 3 and the backward 5 tensor passes through HBM instead of 2 and 3.  Same operation order, same bits.
template <int C>
__global__ __launch_bounds__(256) void pixelnorm_fwd_reg_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                                long long HW, float eps) {
  const long long total = (long long)N * HW;
  GRID_STRIDE(i, total) {
    const long long n = i / HW, hw = i - n * HW;
    const float* p = x + n * C * HW + hw;
    float v[C];          // (a running pointer: c * HW as 64-bit offsets would cost two address registers per channel)
#pragma unroll
    for (int c = 0; c < C; ++c, p += HW) v[c] = *p;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s += v[c] * v[c];
    const float r = rsqrtf(s / (float)C + eps);
    float* q = y + n * C * HW + hw;
#pragma unroll
    for (int c = 0; c < C; ++c, q += HW) *q = v[c] * r;
  }
}

template <int C>
__global__ __launch_bounds__(256) void pixelnorm_bwd_reg_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                float* __restrict__ gx, int N, long long HW, float eps) {
  const long long total = (long long)N * HW;
  GRID_STRIDE(i, total) {
    const long long n = i / HW, hw = i - n * HW;
    const float* p = x + n * C * HW + hw;
    const float* g = gy + n * C * HW + hw;
    float v[C], w[C];
#pragma unroll
    for (int c = 0; c < C; ++c, p += HW, g += HW) {
      v[c] = *p;
      w[c] = *g;
    }
    float s = 0.f, d = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      s += v[c] * v[c];
      d += v[c] * w[c];
    }
    const float r = rsqrtf(s / (float)C + eps);
    const float k = r * r * r * d / (float)C;
    float* q = gx + n * C * HW + hw;
#pragma unroll
    for (int c = 0; c < C; ++c, q += HW) *q = w[c] * r - v[c] * k;
  }
}

// Wide layers on small maps (C >= 256 at 64^2 and below): one thread per pixel leaves a 32 x 512 x 8 x 8 tensor with 2048
// threads walking 512 dependent-stride loads each (2.3 TB/s measured at 256 x 64^2).  Here a 256-thread block takes
// 256 / S pixels and S channel slices per pixel, the partial sums meet in LDS (summed in slice order: deterministic), and
// the second walk re-reads the block's own 256 / S x C x 4 bytes from the caches.
template <int S>
__global__ __launch_bounds__(256) void pixelnorm_fwd_split_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int N, int C, long long HW, float eps) {
  constexpr int PX = 256 / S;
  __shared__ float red[256];
  const int tl = threadIdx.x % PX, sp = threadIdx.x / PX;
  const long long total = (long long)N * HW;
  for (long long i0 = (long long)blockIdx.x * PX; i0 < total; i0 += (long long)gridDim.x * PX) {
    const long long i = i0 + tl;
    const bool ok = i < total;
    const long long n = ok ? i / HW : 0, hw = ok ? i - n * HW : 0;
    const float* p = x + n * C * HW + hw;
    float s = 0.f;
    if (ok)
      for (int c = sp; c < C; c += S) {
        const float v = p[c * HW];
        s += v * v;
      }
    red[threadIdx.x] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < S; ++k) tot += red[k * PX + tl];
    __syncthreads();
    const float r = rsqrtf(tot / (float)C + eps);
    float* q = y + n * C * HW + hw;
    if (ok)
      for (int c = sp; c < C; c += S) q[c * HW] = p[c * HW] * r;
  }
}

template <int S>
__global__ __launch_bounds__(256) void pixelnorm_bwd_split_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                  float* __restrict__ gx, int N, int C, long long HW,
                                                                  float eps) {
  constexpr int PX = 256 / S;
  __shared__ float red[512];
  const int tl = threadIdx.x % PX, sp = threadIdx.x / PX;
  const long long total = (long long)N * HW;
  for (long long i0 = (long long)blockIdx.x * PX; i0 < total; i0 += (long long)gridDim.x * PX) {
    const long long i = i0 + tl;
    const bool ok = i < total;
    const long long n = ok ? i / HW : 0, hw = ok ? i - n * HW : 0;
    const float* p = x + n * C * HW + hw;
    const float* g = gy + n * C * HW + hw;
    float s = 0.f, d = 0.f;
    if (ok)
      for (int c = sp; c < C; c += S) {
        const float v = p[c * HW];
        s += v * v;
        d += v * g[c * HW];
      }
    red[threadIdx.x] = s;
    red[256 + threadIdx.x] = d;
    __syncthreads();
    float ts = 0.f, td = 0.f;
#pragma unroll
    for (int k = 0; k < S; ++k) {
      ts += red[k * PX + tl];
      td += red[256 + k * PX + tl];
    }
    __syncthreads();
    const float r = rsqrtf(ts / (float)C + eps);
    const float kk = r * r * r * td / (float)C;
    float* q = gx + n * C * HW + hw;
    if (ok)
      for (int c = sp; c < C; c += S) q[c * HW] = g[c * HW] * r - p[c * HW] * kk;
  }
}

// slices per pixel for the split kernels: enough threads for ~8 waves per SIMD, at least 16 pixels (64 bytes) per row
static bool pixelnorm_generic() {       // GANLAB_PIXELNORM_GENERIC=1: the one-thread-per-pixel kernels for every shape (A/B)
  static const bool v = [] { const char* e = getenv("GANLAB_PIXELNORM_GENERIC"); return e != nullptr && atoi(e) != 0; }();
  return v;
}

static int pixelnorm_splits(int N, int C, long long HW) {
  // maps only: a block's 16+ pixels are contiguous within a row of HW; the latent vectors of the mapping network
  // (HW == 1, N x 512 values) stay on the one-thread-per-pixel kernel
  if (C < 256 || HW < 16 || pixelnorm_generic()) return 0;
  const long long px = (long long)N * HW;
  return px >= 262144 ? 0 : (px >= 65536 ? 4 : 16);
}

// ---------------------------------------------------------------------------------------------- //
// minibatch-stddev statistic (custom_layers.py:117-140): one workgroup per group, wave-shuffle reduce
// ---------------------------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void mbstd_fwd_kernel(const float* __restrict__ x, float* __restrict__ stat,
                                                        int gs, long long F, float eps) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  const float* base = x + (long long)g * gs * F;
  float acc = 0.f;
  for (long long f = threadIdx.x; f < F; f += 256) {
    float mu = 0.f;
    for (int i = 0; i < gs; ++i) mu += base[i * F + f];
    mu /= (float)gs;
    float v = 0.f;
    for (int i = 0; i < gs; ++i) {
      const float d = base[i * F + f] - mu;
      v += d * d;
    }
    acc += sqrtf(v / (float)(gs - 1) + eps);
  }
  acc = gl_block_sum_256(acc, red);
  if (threadIdx.x == 0) stat[g] = acc / (float)F;
}

__global__ void mbstd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gstat,
                                 float* __restrict__ gx, int G, int gs, long long F, float eps) {
  const long long total = (long long)G * F;
  GRID_STRIDE(t, total) {
    const long long g = t / F, f = t - g * F;
    const float* base = x + g * gs * F + f;
    float mu = 0.f;
    for (int i = 0; i < gs; ++i) mu += base[i * F];
    mu /= (float)gs;
    float v = 0.f;
    for (int i = 0; i < gs; ++i) {
      const float d = base[i * F] - mu;
      v += d * d;
    }
    const float s = sqrtf(v / (float)(gs - 1) + eps);
    const float k = gstat[g] / ((float)F * (float)(gs - 1) * s);
    float* o = gx + g * gs * F + f;
    for (int i = 0; i < gs; ++i) o[i * F] = k * (base[i * F] - mu);
  }
}

// backward of mbstd_bwd w.r.t. (gstat, x) given the cotangent ggx of gx.
__global__ __launch_bounds__(256) void mbstd_bwdbwd_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ gstat,
                                                           const float* __restrict__ ggx,
                                                           float* __restrict__ g_gstat, float* __restrict__ g_x,
                                                           int gs, long long F, float eps) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  const float* bx = x + (long long)g * gs * F;
  const float* bg = ggx + (long long)g * gs * F;
  float* bo = g_x + (long long)g * gs * F;
  const float c = 1.f / ((float)F * (float)(gs - 1));
  const float G_ = gstat[g];
  float acc = 0.f;
  for (long long f = threadIdx.x; f < F; f += 256) {
    float mu = 0.f, mg = 0.f;
    for (int i = 0; i < gs; ++i) {
      mu += bx[i * F + f];
      mg += bg[i * F + f];
    }
    mu /= (float)gs;
    mg /= (float)gs;
    float v = 0.f, dot = 0.f;
    for (int i = 0; i < gs; ++i) {
      const float d = bx[i * F + f] - mu;
      v += d * d;
      dot += bg[i * F + f] * d;
    }
    const float s = sqrtf(v / (float)(gs - 1) + eps);
    const float is = 1.f / s;
    acc += dot * is;
    const float k2 = dot * is * is * is / (float)(gs - 1);
    for (int i = 0; i < gs; ++i) {
      const float d = bx[i * F + f] - mu;
      bo[i * F + f] = c * G_ * ((bg[i * F + f] - mg) * is - d * k2);
    }
  }
  acc = gl_block_sum_256(acc, red);
  if (threadIdx.x == 0) g_gstat[g] = c * acc;
}

// ---------------------------------------------------------------------------------------------- //
// per-channel affine y = x*scale[c] + shift[c] (BatchNorm / LayerNorm building block), product, tanh
// ---------------------------------------------------------------------------------------------- //
template <int VEC>
__global__ void chan_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float* __restrict__ y, int N, int C,
                                   long long HW) {
  const long long hwv = HW / VEC, total = (long long)N * C * hwv;
  GRID_STRIDE(i, total) {
    const int c = (int)((i / hwv) % C);
    const float a = scale ? scale[c] : 1.f, b = shift ? shift[c] : 0.f;
    if (VEC == 4) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      v.x = v.x * a + b; v.y = v.y * a + b; v.z = v.z * a + b; v.w = v.w * a + b;
      reinterpret_cast<float4*>(y)[i] = v;
    } else {
      y[i] = x[i] * a + b;
    }
  }
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           long long n) {
  GRID_STRIDE(i, n) out[i] = a[i] * b[i];
}

__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  GRID_STRIDE(i, n) y[i] = tanhf(x[i]);
}

__global__ void tanh_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx,
                                long long n) {
  GRID_STRIDE(i, n) gx[i] = gy[i] * (1.f - y[i] * y[i]);
}

// ---------------------------------------------------------------------------------------------- //
// elementwise axpby, reductions, losses
// ---------------------------------------------------------------------------------------------- //
// (out may be y itself: every element is read before it is written, by the same thread)
__global__ void axpby_kernel(const float* __restrict__ x, const float* y, float* out, long long n, float a, float b) {
  const long long n4 = n >> 2;
  GRID_STRIDE(i, n4) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x *= a; v.y *= a; v.z *= a; v.w *= a;
    if (y) {
      const float4 w = reinterpret_cast<const float4*>(y)[i];
      v.x += b * w.x; v.y += b * w.y; v.z += b * w.z; v.w += b * w.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
  const long long base = n4 << 2;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (tid < n - base) {
    const long long j = base + tid;
    out[j] = a * x[j] + (y ? b * y[j] : 0.f);
  }
}

// out = a * gout[0] * (x ? x[i] : 1): backward of the scalar reductions without a host sync
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                 float* __restrict__ out, long long n, float a) {
  const float k = a * gout[0];
  GRID_STRIDE(i, n) out[i] = x ? k * x[i] : k;
}

// out[n,m] = t[n]*a[n,m] + (1-t[n])*b[n,m]   (WGAN-GP interpolates, resnetgan/learner.py:793-796)
__global__ void lerp_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                 const float* __restrict__ t, float* __restrict__ out, long long N, long long M) {
  const long long total = N * M;
  GRID_STRIDE(i, total) {
    const float w = t[i / M];
    out[i] = w * a[i] + (1.f - w) * b[i];
  }
}

__global__ __launch_bounds__(256) void sum_stage1(const float* __restrict__ x, float* __restrict__ part, long long n,
                                                  int squared) {
  __shared__ float red[4];
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float v = x[i];
    s += squared ? v * v : v;
  }
  s = gl_block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_stage2(const float* __restrict__ part, float* __restrict__ out, int nb,
                                                  float scale) {
  __shared__ float red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  float f = gl_block_sum_256((float)s, red);
  if (threadIdx.x == 0) out[0] = f * scale;
}

inline int sum_blocks(long long n) {
  long long b = (n + 256 * 16 - 1) / (256 * 16);
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}

__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int n,
                                                      float t) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = x[i];
    s += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
  }
  s = gl_block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, float* __restrict__ gx,
                               int n, float t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float v = x[i];
    const float sg = 1.f / (1.f + expf(-v));
    gx[i] = gout[0] * (sg - t) / (float)n;
  }
}

__global__ __launch_bounds__(256) void chnorm_pen_stage1(const float* __restrict__ g, float* __restrict__ part,
                                                         int N, int C, long long HW, float gamma) {
  __shared__ float red[4];
  const long long total = (long long)N * HW;
  float s = 0.f;
  GRID_STRIDE(i, total) {
    const long long n = i / HW, hw = i - n * HW;
    const float* p = g + n * C * HW + hw;
    float q = 0.f;
    for (int c = 0; c < C; ++c) q += p[c * HW] * p[c * HW];
    const float d = sqrtf(q) - gamma;
    s += d * d;
  }
  s = gl_block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void chnorm_pen_bwd_kernel(const float* __restrict__ g, const float* __restrict__ gout,
                                      float* __restrict__ gg, int N, int C, long long HW, float gamma, float scale) {
  const long long total = (long long)N * HW;
  const float go = gout[0] * scale * 2.f;
  GRID_STRIDE(i, total) {
    const long long n = i / HW, hw = i - n * HW;
    const float* p = g + n * C * HW + hw;
    float q = 0.f;
    for (int c = 0; c < C; ++c) q += p[c * HW] * p[c * HW];
    const float s = sqrtf(q);
    const float k = s > 0.f ? go * (s - gamma) / s : 0.f;
    float* o = gg + n * C * HW + hw;
    for (int c = 0; c < C; ++c) o[c * HW] = k * p[c * HW];
  }
}

// ---------------------------------------------------------------------------------------------- //
// optimiser + EWMA + RNG
// ---------------------------------------------------------------------------------------------- //
// ``dev``: (lr, bc1, bc2) of THIS step read from device memory - the step-graph form (graphs.py GraphedStep): a captured
// launch replays with its arguments frozen, so what changes from step to step lives in a small device block that one
// ordinary launch (set_scalars_kernel) rewrites before each replay
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2, const float* __restrict__ dev) {
  if (dev != nullptr) {
    lr = dev[0];
    bc1 = dev[1];
    bc2 = dev[2];
  }
  const float step = lr / bc1, isq = rsqrtf(bc2);
  GRID_STRIDE(i, n) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step * mi / (sqrtf(vi) * isq + eps);
  }
}

__global__ void ewma_kernel(float* __restrict__ lag, const float* __restrict__ p, long long n, float beta) {
  GRID_STRIDE(i, n) lag[i] = p[i] * (1.f - beta) + lag[i] * beta;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// ``base``: device-resident stream position added to ``offset`` (step graphs, see adam_kernel)
__global__ void randn_kernel(float* __restrict__ out, long long n, uint64_t seed, uint64_t offset,
                             const uint64_t* __restrict__ base) {
  if (base != nullptr) offset += *base;
  const long long n4 = (n + 3) >> 2;
  GRID_STRIDE(i, n4) {
    const uint64_t ctr = offset + (uint64_t)i;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float r[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float u1 = ((float)c[2 * k] + 1.0f) * (1.0f / 4294967296.0f);  // (0,1]
      const float u2 = (float)c[2 * k + 1] * (1.0f / 4294967296.0f);
      const float rad = sqrtf(-2.f * logf(u1));
      float sn, cs;
      sincosf(6.28318530717958647692f * u2, &sn, &cs);
      r[2 * k] = rad * cs;
      r[2 * k + 1] = rad * sn;
    }
    for (int k = 0; k < 4; ++k)
      if (i * 4 + k < n) out[i * 4 + k] = r[k];
  }
}

__global__ void set_scalars_kernel(uint32_t* __restrict__ block, uint64_t rng_base, float f0, float f1, float f2,
                                   float f3, float f4, float f5) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *reinterpret_cast<uint64_t*>(block) = rng_base;
    float* f = reinterpret_cast<float*>(block) + 4;
    f[0] = f0; f[1] = f1; f[2] = f2; f[3] = f3; f[4] = f4; f[5] = f5;
  }
}

}  // namespace

#define ST gl_stream(stream)

extern "C" {

int ganlab_abi_version(void) { return 1; }

int ganlab_blur3x3_f32(const float* x, float* y, long long planes, int H, int W, void* stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0) return GANLAB_EINVAL;
  if ((W & 3) == 0 && (H & 1) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
    if ((H & 7) == 0 && H >= 256)      // taller strips on the big maps: 10 row reads per 8 outputs instead of 6 per 4
      GL_LAUNCH(blur3x3_vec_kernel<8>, dim3(ew_blocks(planes * (H / 8) * (W / 4))), dim3(256), 0, ST, x, y, planes, H,
                W);
    else if ((H & 3) == 0)
      GL_LAUNCH(blur3x3_vec_kernel<4>, dim3(ew_blocks(planes * (H / 4) * (W / 4))), dim3(256), 0, ST, x, y, planes, H,
                W);
    else
      GL_LAUNCH(blur3x3_vec_kernel<2>, dim3(ew_blocks(planes * (H / 2) * (W / 4))), dim3(256), 0, ST, x, y, planes, H,
                W);
  else
    GL_LAUNCH(blur3x3_kernel, dim3(ew_blocks(planes * H * W)), dim3(256), 0, ST, x, y, planes, H, W);
  return GL_CHECK_LAUNCH();
}

/* sign-bit masks (bit e of the NCHW-linear element index e set iff x[e] > 0): whole words per row */
int ganlab_mask_bits_supported(int H, int W) { return (H >= 2 && (H & 1) == 0 && W >= 32 && (W & 31) == 0) ? 1 : 0; }

/* y = blur(x) AND bits = sign bits of x (planes*H*W/32 words) in the same pass: the critic's conv -> LeakyReLU -> blur
 * (progan/architectures.py:261-284) keeps only the SIGN of the LeakyReLU output for its backward */
int ganlab_blur3x3_bits_f32(const float* x, float* y, unsigned* bits, long long planes, int H, int W, void* stream) {
  if (!x || !y || !bits || planes <= 0) return GANLAB_EINVAL;
  if (!ganlab_mask_bits_supported(H, W) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0)
    return GANLAB_EUNSUPPORTED;
  if ((H & 7) == 0 && H >= 256)
    GL_LAUNCH((blur3x3_vec_kernel<8, true>), dim3(ew_blocks(planes * (H / 8) * (W / 4))), dim3(256), 0, ST, x, y, planes, H,
              W, bits);
  else if ((H & 3) == 0)
    GL_LAUNCH((blur3x3_vec_kernel<4, true>), dim3(ew_blocks(planes * (H / 4) * (W / 4))), dim3(256), 0, ST, x, y, planes, H,
              W, bits);
  else
    GL_LAUNCH((blur3x3_vec_kernel<2, true>), dim3(ew_blocks(planes * (H / 2) * (W / 4))), dim3(256), 0, ST, x, y, planes, H,
              W, bits);
  return GL_CHECK_LAUNCH();
}

int ganlab_up2_f32(const float* x, float* y, long long planes, int H, int W, float scale, void* stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(up2_kernel, dim3(ew_blocks(planes * H * W * 4)), dim3(256), 0, ST, x, y, planes, H, W, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_pool2_f32(const float* x, float* y, long long planes, int Hout, int Wout, float scale, void* stream) {
  if (!x || !y || planes <= 0 || Hout <= 0 || Wout <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(pool2_kernel, dim3(ew_blocks(planes * Hout * Wout)), dim3(256), 0, ST, x, y, planes, Hout,
                     Wout, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_bias_act_f32(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                        int N, int C, long long HW, float bias_scale, int act, float slope, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || HW <= 0 || (noise && !noise_w)) return GANLAB_EINVAL;
  if ((HW & 3) == 0)
    GL_LAUNCH(bias_act_kernel<4>, dim3(ew_blocks((long long)N * C * HW / 4)), dim3(256), 0, ST, x, bias,
                       noise, noise_w, y, N, C, HW, bias_scale, act, slope);
  else
    GL_LAUNCH(bias_act_kernel<1>, dim3(ew_blocks((long long)N * C * HW)), dim3(256), 0, ST, x, bias,
                       noise, noise_w, y, N, C, HW, bias_scale, act, slope);
  return GL_CHECK_LAUNCH();
}

int ganlab_act_bwd_f32(const float* gy, const float* y, float* gz, long long n, float slope, void* stream) {
  if (!gy || !y || !gz || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(act_bwd_kernel, dim3(ew_blocks((n >> 2) + 4)), dim3(256), 0, ST, gy, y, gz, n, slope);
  return GL_CHECK_LAUNCH();
}

size_t ganlab_channel_sum_workspace(int N, int C, long long HW) {
  return (size_t)C * channel_chunks(N, HW) * sizeof(double);
}

int ganlab_channel_sum_f32(const float* a, const float* b, float* out, int N, int C, long long HW, float scale,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !out || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  const int chunks = channel_chunks(N, HW);
  if (!workspace || workspace_bytes < (size_t)C * chunks * sizeof(double)) return GANLAB_EWORKSPACE;
  GL_LAUNCH(channel_sum_stage1, dim3(chunks, C), dim3(256), 0, ST, a, b, (double*)workspace, N, C, HW,
                     chunks);
  GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace, out, C, chunks, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_act_bwd_bias_f32(const float* gy, const float* y, float* gz, float* gb, int N, int C, long long HW,
                            float slope, float scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (!gy || !y || !gz || !gb || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  const int chunks = channel_chunks(N, HW);
  if (!workspace || workspace_bytes < (size_t)C * chunks * sizeof(double)) return GANLAB_EWORKSPACE;
  GL_LAUNCH(act_bwd_bias_stage1, dim3(chunks, C), dim3(256), 0, ST, gy, y, gz, (double*)workspace, N, C, HW, chunks,
            slope);
  GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace, gb, C, chunks, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_blur_fused_supported(int H, int W) { return (H >= 2 && W >= 4 && (H & 1) == 0 && (W & 3) == 0) ? 1 : 0; }

size_t ganlab_blur_fused_workspace(int N, int C, int H, int W) {
  if (!ganlab_blur_fused_supported(H, W)) return 0;
  return (size_t)2 * C * blur_fused_chunks(N, H, W) * sizeof(double);
}

int ganlab_blur_bias_act_f32(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                             int N, int C, int H, int W, float bias_scale, int act, float slope, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || (noise && !noise_w)) return GANLAB_EINVAL;
  if (!ganlab_blur_fused_supported(H, W)) return GANLAB_EUNSUPPORTED;
  const int chunks = blur_fused_chunks(N, H, W);
  BLUR_FUSED_LAUNCH(BF_FWD, x, (const float*)nullptr, noise, bias, noise_w, y, (double*)nullptr, N, C, H, W, chunks,
                    bias_scale, act, slope, 0);
  return GL_CHECK_LAUNCH();
}

size_t ganlab_act_stats_workspace(int N, int C, long long HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return (size_t)N * C * 128 * 2 * sizeof(double);       // up to 128 chunks per plane (blur: blur_fused_chunks)
}

size_t ganlab_instnorm_bwd_act_workspace(int N, int C, long long HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return (size_t)2 * N * C * 64 * sizeof(double);
}

/* backward of  out = InstanceNorm(x)*(ys+1)+yb  with  x = lrelu(z + noise_w*noise + bias*bias_scale):
 * gz = dL/dz, gb = bias_scale * sum gz (or NULL), gnw = sum gz*noise (or NULL).  s1, s2 from
 * ganlab_instnorm_style_bwd_reduce_f32.  HW % 4 == 0. */
int ganlab_instnorm_style_bwd_act_f32(const float* gy, const float* x, const float* mean, const float* rstd,
                                      const float* style, const float* s1, const float* s2, const float* noise,
                                      float* gz, float* gb, float* gnw, int N, int C, long long HW, int act,
                                      float slope, float bias_scale, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (!gy || !x || !mean || !rstd || !s1 || !s2 || !gz || N <= 0 || C <= 0 || HW <= 0 || (gnw && !noise))
    return GANLAB_EINVAL;
  if ((HW & 3) != 0) return GANLAB_EUNSUPPORTED;
  if ((gb || gnw) && (!workspace || workspace_bytes < ganlab_instnorm_bwd_act_workspace(N, C, HW)))
    return GANLAB_EWORKSPACE;
  const long long hw4 = HW / 4, planes = (long long)N * C;
  const int chunks = act_stats_chunks(hw4);
  double* part = reinterpret_cast<double*>(workspace);
  GL_LAUNCH(instnorm_bwd_apply_act_kernel, dim3((unsigned)chunks, (unsigned)planes), dim3(256), 0, ST, gy, x, mean,
            rstd, style, s1, s2, noise, gz, part, N, C, hw4, chunks, act, slope, gb ? 1 : 0, gnw ? 1 : 0, pw_contig());
  if (gb) GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)part, gb, C, N * chunks, bias_scale);
  if (gnw)
    GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)part + (size_t)C * N * chunks, gnw, C,
              N * chunks, 1.f);
  return GL_CHECK_LAUNCH();
}

/* The two InstanceNorm backward passes of a generator layer whose ONLY reader is toRGB, fed with the image gradient instead
 * of the layer's incoming gradient  gy[n, c] = sum_k wp[k * round_up(C, 64) + c] * grgb[n, k]  (wp: toRGB's input-gradient
 * pack, scale included) - toRGB's backward then writes nothing and these passes read crgb planes per image instead of C.
 * gz / gb / gnw: the bits of ganlab_conv_dgrad_f32 followed by the plain entry points given the same s1 / s2; s1 / s2 themselves
 * are fp64 sums in another order (equal after rounding to fp32 up to ties).  C <= 16, crgb <= 4, HW % 4 == 0. */
int ganlab_instnorm_bwd_rgb_supported(int N, int C, int crgb, long long HW) {
  return (N > 0 && N <= 65535 && C > 0 && C <= 16 && crgb > 0 && crgb <= 4 && HW >= 1024 && (HW & 3) == 0) ? 1 : 0;
}

static RgbSrc make_rgb_src(const float* grgb, const float* wp, int crgb, int C) {
  RgbSrc r;
  r.grgb = grgb; r.wp = wp; r.crgb = crgb; r.cout_p = (C + 63) / 64 * 64;
  return r;
}

size_t ganlab_instnorm_bwd_reduce_rgb_workspace(int N, int C, long long HW) {
  if (!ganlab_instnorm_bwd_rgb_supported(N, C, 3, HW)) return 0;
  return (size_t)N * C * act_stats_chunks(HW / 4) * 2 * sizeof(double);
}

int ganlab_instnorm_style_bwd_reduce_rgb_f32(const float* grgb, const float* wp, int crgb, const float* x,
                                             const float* mean, const float* rstd, float* s1, float* s2, int N, int C,
                                             long long HW, void* workspace, size_t workspace_bytes, void* stream) {
  if (!grgb || !wp || !x || !mean || !rstd || !s1 || !s2) return GANLAB_EINVAL;
  if (!ganlab_instnorm_bwd_rgb_supported(N, C, crgb, HW)) return GANLAB_EUNSUPPORTED;
  if (!workspace || workspace_bytes < ganlab_instnorm_bwd_reduce_rgb_workspace(N, C, HW)) return GANLAB_EWORKSPACE;
  const long long hw4 = HW / 4, planes = (long long)N * C;
  const int chunks = act_stats_chunks(hw4);
  double* part = reinterpret_cast<double*>(workspace);
  if (C == RGB_MAXC && crgb == 3)
    GL_LAUNCH(instnorm_bwd_reduce_rgb_kernel<true>, dim3((unsigned)chunks, (unsigned)N), dim3(256), 0, ST,
              make_rgb_src(grgb, wp, crgb, C), x, mean, rstd, part, C, hw4, chunks, pw_contig());
  else
    GL_LAUNCH(instnorm_bwd_reduce_rgb_kernel<false>, dim3((unsigned)chunks, (unsigned)N), dim3(256), 0, ST,
              make_rgb_src(grgb, wp, crgb, C), x, mean, rstd, part, C, hw4, chunks, pw_contig());
  GL_LAUNCH(instnorm_bwd_reduce_rgb_finish_kernel, dim3((unsigned)((planes + 63) / 64)), dim3(64), 0, ST,
            (const double*)part, s1, s2, planes, chunks);
  return GL_CHECK_LAUNCH();
}

int ganlab_instnorm_style_bwd_act_rgb_f32(const float* grgb, const float* wp, int crgb, const float* x, const float* mean,
                                          const float* rstd, const float* style, const float* s1, const float* s2,
                                          const float* noise, float* gz, float* gb, float* gnw, int N, int C, long long HW,
                                          int act, float slope, float bias_scale, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  if (!grgb || !wp || !x || !mean || !rstd || !s1 || !s2 || !gz || (gnw && !noise)) return GANLAB_EINVAL;
  if (!ganlab_instnorm_bwd_rgb_supported(N, C, crgb, HW)) return GANLAB_EUNSUPPORTED;
  if ((gb || gnw) && (!workspace || workspace_bytes < ganlab_instnorm_bwd_act_workspace(N, C, HW)))
    return GANLAB_EWORKSPACE;
  const long long hw4 = HW / 4;
  const int chunks = act_stats_chunks(hw4);
  double* part = reinterpret_cast<double*>(workspace);
  if (C == RGB_MAXC && crgb == 3 && act == GANLAB_ACT_LRELU)
    GL_LAUNCH(instnorm_bwd_apply_act_rgb_kernel<true>, dim3((unsigned)chunks, (unsigned)N), dim3(256), 0, ST,
              make_rgb_src(grgb, wp, crgb, C), x, mean, rstd, style, s1, s2, noise, gz, part, N, C, hw4, chunks, act, slope,
              gb ? 1 : 0, gnw ? 1 : 0, pw_contig());
  else
    GL_LAUNCH(instnorm_bwd_apply_act_rgb_kernel<false>, dim3((unsigned)chunks, (unsigned)N), dim3(256), 0, ST,
              make_rgb_src(grgb, wp, crgb, C), x, mean, rstd, style, s1, s2, noise, gz, part, N, C, hw4, chunks, act, slope,
              gb ? 1 : 0, gnw ? 1 : 0, pw_contig());
  if (gb) GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)part, gb, C, N * chunks, bias_scale);
  if (gnw)
    GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)part + (size_t)C * N * chunks, gnw, C,
              N * chunks, 1.f);
  return GL_CHECK_LAUNCH();
}

/* ganlab_instnorm_style_bwd_act_f32 followed by the blur (self-adjoint) in one pass: out = blur(gz) for a generator layer whose
 * tail sits behind Upsample -> conv -> blur; gz itself is not written.  workspace: ganlab_instnorm_bwd_act_workspace. */
int ganlab_instnorm_style_bwd_act_blur_f32(const float* gy, const float* x, const float* mean, const float* rstd,
                                           const float* style, const float* s1, const float* s2, const float* noise,
                                           float* out, float* gb, float* gnw, int N, int C, int H, int W, int act,
                                           float slope, float bias_scale, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  if (!gy || !x || !mean || !rstd || !s1 || !s2 || !out || N <= 0 || C <= 0 || (gnw && !noise)) return GANLAB_EINVAL;
  if (!ganlab_blur_fused_supported(H, W) || N > 65535 || C > 65535) return GANLAB_EUNSUPPORTED;
  const long long HW = (long long)H * W;
  if ((gb || gnw) && (!workspace || workspace_bytes < ganlab_instnorm_bwd_act_workspace(N, C, HW))) return GANLAB_EWORKSPACE;
  const int rows = blur_rows(H);
  const long long per_n = (long long)(H / rows) * (W / 4);
  int chunks = (int)((per_n + 256 * 4 - 1) / (256 * 4));
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;          // the workspace holds 64 chunks per plane
  double* part = reinterpret_cast<double*>(workspace);
  const dim3 grid((unsigned)chunks, (unsigned)C, (unsigned)N);
  if (rows == 4)
    GL_LAUNCH(instnorm_bwd_act_blur_kernel<4>, grid, dim3(256), 0, ST, gy, x, mean, rstd, style, s1, s2, noise, out, part, N, C,
              H, W, chunks, act, slope, gb ? 1 : 0, gnw ? 1 : 0);
  else
    GL_LAUNCH(instnorm_bwd_act_blur_kernel<2>, grid, dim3(256), 0, ST, gy, x, mean, rstd, style, s1, s2, noise, out, part, N, C,
              H, W, chunks, act, slope, gb ? 1 : 0, gnw ? 1 : 0);
  if (gb) GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)part, gb, C, N * chunks, bias_scale);
  if (gnw)
    GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)part + (size_t)C * N * chunks, gnw, C,
              N * chunks, 1.f);
  return GL_CHECK_LAUNCH();
}

/* y = act(x + noise + bias) and the InstanceNorm statistics (mean, rstd with eps) of y in the same pass */
int ganlab_bias_act_stats_f32(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                              float* mean, float* rstd, int N, int C, long long HW, float bias_scale, int act,
                              float slope, float eps, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !y || !mean || !rstd || N <= 0 || C <= 0 || HW <= 0 || (noise && !noise_w)) return GANLAB_EINVAL;
  if ((HW & 3) != 0) return GANLAB_EUNSUPPORTED;
  if (!workspace || workspace_bytes < ganlab_act_stats_workspace(N, C, HW)) return GANLAB_EWORKSPACE;
  const long long hw4 = HW / 4, planes = (long long)N * C;
  const int chunks = act_stats_chunks(hw4);
  double* sp = reinterpret_cast<double*>(workspace);
  GL_LAUNCH(bias_act_stats_kernel, dim3((unsigned)chunks, (unsigned)planes), dim3(256), 0, ST, x, bias, noise, noise_w,
            y, sp, C, hw4, chunks, bias_scale, act, slope, pw_contig());
  GL_LAUNCH(act_stats_finish_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, ST, (const double*)sp, mean,
            rstd, planes, chunks, 1.0 / (double)HW, eps);
  return GL_CHECK_LAUNCH();
}

/* y = act(blur(x) + noise + bias) and the InstanceNorm statistics of y in the same pass */
int ganlab_blur_bias_act_stats_f32(const float* x, const float* bias, const float* noise, const float* noise_w,
                                   float* y, float* mean, float* rstd, int N, int C, int H, int W, float bias_scale,
                                   int act, float slope, float eps, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (!x || !y || !mean || !rstd || N <= 0 || C <= 0 || (noise && !noise_w)) return GANLAB_EINVAL;
  if (!ganlab_blur_fused_supported(H, W)) return GANLAB_EUNSUPPORTED;
  if (!workspace || workspace_bytes < ganlab_act_stats_workspace(N, C, (long long)H * W)) return GANLAB_EWORKSPACE;
  const int rows = blur_rows_mode(H, BF_FWD);
  long long per_n = (long long)(H / rows) * (W / 4);
  int chunks = (int)((per_n + 256 * 4 - 1) / (256 * 4));
  if (chunks < 1) chunks = 1;
  if (chunks > 128) chunks = 128;
  double* sp = reinterpret_cast<double*>(workspace);
  const dim3 grid((unsigned)chunks, (unsigned)C, (unsigned)N);
  if (rows == 8)
    GL_LAUNCH((blur_fused_kernel<BF_FWD, 8, true>), grid, dim3(256), 0, ST, x, (const float*)nullptr, noise, bias,
              noise_w, y, (double*)nullptr, N, C, H, W, chunks, bias_scale, act, slope, 0, sp);
  else if (rows == 4)
    GL_LAUNCH((blur_fused_kernel<BF_FWD, 4, true>), grid, dim3(256), 0, ST, x, (const float*)nullptr, noise, bias,
              noise_w, y, (double*)nullptr, N, C, H, W, chunks, bias_scale, act, slope, 0, sp);
  else
    GL_LAUNCH((blur_fused_kernel<BF_FWD, 2, true>), grid, dim3(256), 0, ST, x, (const float*)nullptr, noise, bias,
              noise_w, y, (double*)nullptr, N, C, H, W, chunks, bias_scale, act, slope, 0, sp);
  const long long planes = (long long)N * C;
  GL_LAUNCH(act_stats_finish_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, ST, (const double*)sp, mean,
            rstd, planes, chunks, 1.0 / ((double)H * W), eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_blur_act_bwd_f32(const float* g, const float* y, float* out, float* gb, int N, int C, int H, int W,
                            float slope, float bias_scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !y || !out || N <= 0 || C <= 0) return GANLAB_EINVAL;
  if (!ganlab_blur_fused_supported(H, W)) return GANLAB_EUNSUPPORTED;
  const int chunks = blur_fused_chunks(N, H, W);
  if (gb && (!workspace || workspace_bytes < (size_t)C * chunks * sizeof(double))) return GANLAB_EWORKSPACE;
  BLUR_FUSED_LAUNCH(BF_A, g, y, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, out,
                    (double*)workspace, N, C, H, W, chunks, 1.f, 0, slope, gb ? 1 : 0);
  if (gb) GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace, gb, C, chunks, bias_scale);
  return GL_CHECK_LAUNCH();
}

/* ganlab_blur_act_bwd_f32 with the activation's sign bits (ganlab_blur3x3_bits_f32) in place of y */
int ganlab_blur_act_bwd_bits_f32(const float* g, const unsigned* ybits, float* out, float* gb, int N, int C, int H, int W,
                                 float slope, float bias_scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !ybits || !out || N <= 0 || C <= 0) return GANLAB_EINVAL;
  if (!ganlab_blur_fused_supported(H, W) || !ganlab_mask_bits_supported(H, W)) return GANLAB_EUNSUPPORTED;
  const int chunks = blur_fused_chunks(N, H, W);
  if (gb && (!workspace || workspace_bytes < (size_t)C * chunks * sizeof(double))) return GANLAB_EWORKSPACE;
  BLUR_FUSED_LAUNCH_BITS(BF_A, g, reinterpret_cast<const float*>(ybits), (const float*)nullptr, (const float*)nullptr,
                         (const float*)nullptr, out, (double*)workspace, N, C, H, W, chunks, 1.f, 0, slope, gb ? 1 : 0);
  if (gb) GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace, gb, C, chunks, bias_scale);
  return GL_CHECK_LAUNCH();
}

/* ganlab_act_bwd_blur_f32 (its adjoint) with sign bits in place of y */
int ganlab_act_bwd_blur_bits_f32(const float* g, const unsigned* ybits, const float* noise, float* out, float* gb,
                                 float* gnw, int N, int C, int H, int W, float slope, float bias_scale, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!g || !ybits || !out || N <= 0 || C <= 0 || (gnw && !noise)) return GANLAB_EINVAL;
  if (!ganlab_blur_fused_supported(H, W) || !ganlab_mask_bits_supported(H, W)) return GANLAB_EUNSUPPORTED;
  const int chunks = blur_fused_chunks(N, H, W);
  const int sums = (gb || gnw) ? 1 : 0;
  if (sums && (!workspace || workspace_bytes < (size_t)2 * C * chunks * sizeof(double))) return GANLAB_EWORKSPACE;
  BLUR_FUSED_LAUNCH_BITS(BF_AT, g, reinterpret_cast<const float*>(ybits), gnw ? noise : (const float*)nullptr,
                         (const float*)nullptr, (const float*)nullptr, out, (double*)workspace, N, C, H, W, chunks, 1.f, 0,
                         slope, sums);
  if (gb) GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace, gb, C, chunks, bias_scale);
  if (gnw)
    GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace + (size_t)C * chunks, gnw, C,
              chunks, 1.f);
  return GL_CHECK_LAUNCH();
}

int ganlab_act_bwd_blur_f32(const float* g, const float* y, const float* noise, float* out, float* gb, float* gnw,
                            int N, int C, int H, int W, float slope, float bias_scale, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!g || !y || !out || N <= 0 || C <= 0 || (gnw && !noise)) return GANLAB_EINVAL;
  if (!ganlab_blur_fused_supported(H, W)) return GANLAB_EUNSUPPORTED;
  const int chunks = blur_fused_chunks(N, H, W);
  const int sums = (gb || gnw) ? 1 : 0;
  if (sums && (!workspace || workspace_bytes < (size_t)2 * C * chunks * sizeof(double))) return GANLAB_EWORKSPACE;
  BLUR_FUSED_LAUNCH(BF_AT, g, y, gnw ? noise : (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, out,
                    (double*)workspace, N, C, H, W, chunks, 1.f, 0, slope, sums);
  if (gb) GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace, gb, C, chunks, bias_scale);
  if (gnw)
    GL_LAUNCH(channel_sum_stage2, dim3(C), dim3(64), 0, ST, (const double*)workspace + (size_t)C * chunks, gnw, C,
              chunks, 1.f);
  return GL_CHECK_LAUNCH();
}

int ganlab_instnorm_stats_f32(const float* x, float* mean, float* rstd, long long planes, long long HW, float eps,
                              void* stream) {
  if (!x || !mean || !rstd || planes <= 0 || HW <= 0) return GANLAB_EINVAL;
  if (HW >= 1024)
    GL_LAUNCH(instnorm_stats_kernel<256>, dim3((unsigned)planes), dim3(256), 0, ST, x, mean, rstd, planes,
                       HW, eps);
  else
    GL_LAUNCH(instnorm_stats_kernel<64>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, ST, x, mean,
                       rstd, planes, HW, eps);
  return GL_CHECK_LAUNCH();
}

/* the same statistics for FEW, LONG rows (LayerNorm over C*H*W: batch-many rows of 10^5..10^6 elements - one block per
 * row leaves 3/4 of the CUs idle): each row is cut into chunks, fp64 partial sums per chunk, fixed-order finish */
size_t ganlab_row_stats_workspace(long long rows, long long M) {
  if (rows <= 0 || M <= 0) return 0;
  return (size_t)rows * act_stats_chunks(M / 4) * 2 * sizeof(double);
}

int ganlab_row_stats_f32(const float* x, float* mean, float* rstd, long long rows, long long M, float eps,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !mean || !rstd || rows <= 0 || M <= 0) return GANLAB_EINVAL;
  const int chunks = act_stats_chunks(M / 4);
  if ((M & 3) != 0 || chunks < 2 || rows * chunks > 0x7fffffffLL)      // short or ragged rows: one block per row
    return ganlab_instnorm_stats_f32(x, mean, rstd, rows, M, eps, stream);
  if (!workspace || workspace_bytes < ganlab_row_stats_workspace(rows, M)) return GANLAB_EWORKSPACE;
  double* sp = reinterpret_cast<double*>(workspace);
  GL_LAUNCH(row_stats_partial_kernel, dim3((unsigned)(rows * chunks)), dim3(256), 0, ST, x, sp, chunks, M >> 2);
  GL_LAUNCH(act_stats_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ST, (const double*)sp, mean,
            rstd, rows, chunks, 1.0 / (double)M, eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_instnorm_style_fwd_f32(const float* x, const float* mean, const float* rstd, const float* style,
                                  float* y, int N, int C, long long HW, void* stream) {
  if (!x || !mean || !rstd || !y || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  // U = 2 float4 per thread measured best (tools/pw_probe.py: 5.8 vs 5.1 TB/s for the grid-stride form at 32x16x1024^2;
  // U = 4 equal, U = 8 5.4); GANLAB_PW_CHUNK=0 selects the grid-stride kernel (A/B)
  // (the grid and the kernel's per-block span must agree: only U in {2, 4, 8} exists, anything else means 2)
  static const int chunk_u = [] {
    const char* e = getenv("GANLAB_PW_CHUNK");
    const int v = e ? atoi(e) : 2;
    return (v == 0 || v == 2 || v == 4 || v == 8) ? v : 2;
  }();
  if (chunk_u > 0 && HW % (1024 * 8) == 0 && (long long)N * C * (HW / (1024 * chunk_u)) < 0x7fffffffLL) {
    const int chunks = (int)(HW / (1024 * chunk_u));
    const unsigned grid = (unsigned)((long long)N * C * chunks);
    if (chunk_u == 8) GL_LAUNCH(instnorm_style_fwd_chunk_kernel<8>, dim3(grid), dim3(256), 0, ST, x, mean, rstd, style, y, C, chunks, HW);
    else if (chunk_u == 2) GL_LAUNCH(instnorm_style_fwd_chunk_kernel<2>, dim3(grid), dim3(256), 0, ST, x, mean, rstd, style, y, C, chunks, HW);
    else GL_LAUNCH(instnorm_style_fwd_chunk_kernel<4>, dim3(grid), dim3(256), 0, ST, x, mean, rstd, style, y, C, chunks, HW);
  } else if ((HW & 3) == 0)
    GL_LAUNCH(instnorm_style_fwd_kernel<4>, dim3(ew_blocks((long long)N * C * HW / 4)), dim3(256), 0, ST,
                       x, mean, rstd, style, y, N, C, HW);
  else
    GL_LAUNCH(instnorm_style_fwd_kernel<1>, dim3(ew_blocks((long long)N * C * HW)), dim3(256), 0, ST, x,
                       mean, rstd, style, y, N, C, HW);
  return GL_CHECK_LAUNCH();
}

int ganlab_instnorm_style_bwd_reduce_f32(const float* gy, const float* x, const float* mean, const float* rstd,
                                         float* s1, float* s2, long long planes, long long HW, void* stream) {
  if (!gy || !x || !mean || !rstd || !s1 || !s2 || planes <= 0 || HW <= 0) return GANLAB_EINVAL;
  if (HW >= 1024)
    GL_LAUNCH(instnorm_bwd_reduce_kernel<256>, dim3((unsigned)planes), dim3(256), 0, ST, gy, x, mean,
                       rstd, s1, s2, planes, HW);
  else
    GL_LAUNCH(instnorm_bwd_reduce_kernel<64>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, ST, gy, x,
                       mean, rstd, s1, s2, planes, HW);
  return GL_CHECK_LAUNCH();
}

int ganlab_instnorm_style_bwd_apply_f32(const float* gy, const float* x, const float* mean, const float* rstd,
                                        const float* style, const float* s1, const float* s2, float* gx, int N,
                                        int C, long long HW, void* stream) {
  if (!gy || !x || !mean || !rstd || !s1 || !s2 || !gx || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  if ((HW & 3) == 0)
    GL_LAUNCH(instnorm_bwd_apply_kernel<4>, dim3(ew_blocks((long long)N * C * HW / 4)), dim3(256), 0, ST,
                       gy, x, mean, rstd, style, s1, s2, gx, N, C, HW);
  else
    GL_LAUNCH(instnorm_bwd_apply_kernel<1>, dim3(ew_blocks((long long)N * C * HW)), dim3(256), 0, ST, gy,
                       x, mean, rstd, style, s1, s2, gx, N, C, HW);
  return GL_CHECK_LAUNCH();
}

int ganlab_pixelnorm_fwd_f32(const float* x, float* y, int N, int C, long long HW, float eps, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  const dim3 gd(ew_blocks((long long)N * HW)), bd(256);
  if (pixelnorm_generic()) GL_LAUNCH(pixelnorm_fwd_kernel, gd, bd, 0, ST, x, y, N, C, HW, eps);
  else if (C == 16) GL_LAUNCH(pixelnorm_fwd_reg_kernel<16>, gd, bd, 0, ST, x, y, N, HW, eps);
  else if (C == 32) GL_LAUNCH(pixelnorm_fwd_reg_kernel<32>, gd, bd, 0, ST, x, y, N, HW, eps);
  else if (C == 64) GL_LAUNCH(pixelnorm_fwd_reg_kernel<64>, gd, bd, 0, ST, x, y, N, HW, eps);
  else if (C == 128) GL_LAUNCH(pixelnorm_fwd_reg_kernel<128>, gd, bd, 0, ST, x, y, N, HW, eps);
  else if (pixelnorm_splits(N, C, HW) == 4)
    GL_LAUNCH(pixelnorm_fwd_split_kernel<4>, dim3(ew_blocks((long long)N * HW * 4)), bd, 0, ST, x, y, N, C, HW, eps);
  else if (pixelnorm_splits(N, C, HW) == 16)
    GL_LAUNCH(pixelnorm_fwd_split_kernel<16>, dim3(ew_blocks((long long)N * HW * 16)), bd, 0, ST, x, y, N, C, HW, eps);
  else GL_LAUNCH(pixelnorm_fwd_kernel, gd, bd, 0, ST, x, y, N, C, HW, eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_pixelnorm_bwd_f32(const float* gy, const float* x, float* gx, int N, int C, long long HW, float eps,
                             void* stream) {
  if (!gy || !x || !gx || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  const dim3 gd(ew_blocks((long long)N * HW)), bd(256);
  if (pixelnorm_generic()) GL_LAUNCH(pixelnorm_bwd_kernel, gd, bd, 0, ST, gy, x, gx, N, C, HW, eps);
  else if (C == 16) GL_LAUNCH(pixelnorm_bwd_reg_kernel<16>, gd, bd, 0, ST, gy, x, gx, N, HW, eps);
  else if (C == 32) GL_LAUNCH(pixelnorm_bwd_reg_kernel<32>, gd, bd, 0, ST, gy, x, gx, N, HW, eps);
  else if (C == 64) GL_LAUNCH(pixelnorm_bwd_reg_kernel<64>, gd, bd, 0, ST, gy, x, gx, N, HW, eps);
  else if (C == 128) GL_LAUNCH(pixelnorm_bwd_reg_kernel<128>, gd, bd, 0, ST, gy, x, gx, N, HW, eps);
  else if (pixelnorm_splits(N, C, HW) == 4)
    GL_LAUNCH(pixelnorm_bwd_split_kernel<4>, dim3(ew_blocks((long long)N * HW * 4)), bd, 0, ST, gy, x, gx, N, C, HW, eps);
  else if (pixelnorm_splits(N, C, HW) == 16)
    GL_LAUNCH(pixelnorm_bwd_split_kernel<16>, dim3(ew_blocks((long long)N * HW * 16)), bd, 0, ST, gy, x, gx, N, C, HW,
              eps);
  else GL_LAUNCH(pixelnorm_bwd_kernel, gd, bd, 0, ST, gy, x, gx, N, C, HW, eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_mbstd_fwd_f32(const float* x, float* stat, int G, int gs, long long F, float eps, void* stream) {
  if (!x || !stat || G <= 0 || gs <= 1 || F <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(mbstd_fwd_kernel, dim3(G), dim3(256), 0, ST, x, stat, gs, F, eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_mbstd_bwd_f32(const float* x, const float* gstat, float* gx, int G, int gs, long long F, float eps,
                         void* stream) {
  if (!x || !gstat || !gx || G <= 0 || gs <= 1 || F <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(mbstd_bwd_kernel, dim3(ew_blocks((long long)G * F)), dim3(256), 0, ST, x, gstat, gx, G, gs, F,
                     eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_mbstd_bwdbwd_f32(const float* x, const float* gstat, const float* ggx, float* g_gstat, float* g_x,
                            int G, int gs, long long F, float eps, void* stream) {
  if (!x || !gstat || !ggx || !g_gstat || !g_x || G <= 0 || gs <= 1 || F <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(mbstd_bwdbwd_kernel, dim3(G), dim3(256), 0, ST, x, gstat, ggx, g_gstat, g_x, gs, F, eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_chan_affine_f32(const float* x, const float* scale, const float* shift, float* y, int N, int C,
                           long long HW, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  if ((HW & 3) == 0)
    GL_LAUNCH(chan_affine_kernel<4>, dim3(ew_blocks((long long)N * C * HW / 4)), dim3(256), 0, ST, x, scale, shift, y,
              N, C, HW);
  else
    GL_LAUNCH(chan_affine_kernel<1>, dim3(ew_blocks((long long)N * C * HW)), dim3(256), 0, ST, x, scale, shift, y, N,
              C, HW);
  return GL_CHECK_LAUNCH();
}

int ganlab_mul_f32(const float* a, const float* b, float* out, long long n, void* stream) {
  if (!a || !b || !out || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(mul_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, a, b, out, n);
  return GL_CHECK_LAUNCH();
}

int ganlab_tanh_fwd_f32(const float* x, float* y, long long n, void* stream) {
  if (!x || !y || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(tanh_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, x, y, n);
  return GL_CHECK_LAUNCH();
}

int ganlab_tanh_bwd_f32(const float* gy, const float* y, float* gx, long long n, void* stream) {
  if (!gy || !y || !gx || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(tanh_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, gy, y, gx, n);
  return GL_CHECK_LAUNCH();
}

int ganlab_axpby_f32(const float* x, const float* y, float* out, long long n, float a, float b, void* stream) {
  if (!x || !out || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(axpby_kernel, dim3(ew_blocks((n >> 2) + 4)), dim3(256), 0, ST, x, y, out, n, a, b);
  return GL_CHECK_LAUNCH();
}

int ganlab_scale_dev_f32(const float* x, const float* gout, float* out, long long n, float a, void* stream) {
  if (!gout || !out || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(scale_dev_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, x, gout, out, n, a);
  return GL_CHECK_LAUNCH();
}

int ganlab_lerp_rows_f32(const float* a, const float* b, const float* t, float* out, long long N, long long M,
                         void* stream) {
  if (!a || !b || !t || !out || N <= 0 || M <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(lerp_rows_kernel, dim3(ew_blocks(N * M)), dim3(256), 0, ST, a, b, t, out, N, M);
  return GL_CHECK_LAUNCH();
}

size_t ganlab_sum_workspace(long long n) { return (size_t)sum_blocks(n) * sizeof(float); }

int ganlab_sum_f32(const float* x, float* out, long long n, float scale, int squared, void* workspace,
                   size_t workspace_bytes, void* stream) {
  if (!x || !out || n <= 0) return GANLAB_EINVAL;
  const int nb = sum_blocks(n);
  if (!workspace || workspace_bytes < (size_t)nb * sizeof(float)) return GANLAB_EWORKSPACE;
  GL_LAUNCH(sum_stage1, dim3(nb), dim3(256), 0, ST, x, (float*)workspace, n, squared);
  GL_LAUNCH(sum_stage2, dim3(1), dim3(256), 0, ST, (const float*)workspace, out, nb, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_bce_logits_fwd_f32(const float* x, float* out, int n, float target, void* stream) {
  if (!x || !out || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(bce_fwd_kernel, dim3(1), dim3(256), 0, ST, x, out, n, target);
  return GL_CHECK_LAUNCH();
}

int ganlab_bce_logits_bwd_f32(const float* x, const float* gout, float* gx, int n, float target, void* stream) {
  if (!x || !gout || !gx || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(bce_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, x, gout, gx, n, target);
  return GL_CHECK_LAUNCH();
}

int ganlab_chnorm_penalty_fwd_f32(const float* g, float* out, int N, int C, long long HW, float gamma, float scale,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !out || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  const int nb = sum_blocks((long long)N * HW);
  if (!workspace || workspace_bytes < (size_t)nb * sizeof(float)) return GANLAB_EWORKSPACE;
  GL_LAUNCH(chnorm_pen_stage1, dim3(nb), dim3(256), 0, ST, g, (float*)workspace, N, C, HW, gamma);
  GL_LAUNCH(sum_stage2, dim3(1), dim3(256), 0, ST, (const float*)workspace, out, nb, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_chnorm_penalty_bwd_f32(const float* g, const float* gout, float* gg, int N, int C, long long HW,
                                  float gamma, float scale, void* stream) {
  if (!g || !gout || !gg || N <= 0 || C <= 0 || HW <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(chnorm_pen_bwd_kernel, dim3(ew_blocks((long long)N * HW)), dim3(256), 0, ST, g, gout, gg, N,
                     C, HW, gamma, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_adam_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, float wd, float bc1, float bc2, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || bc1 <= 0.f || bc2 <= 0.f) return GANLAB_EINVAL;
  GL_LAUNCH(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps, wd,
                     bc1, bc2, (const float*)nullptr);
  return GL_CHECK_LAUNCH();
}

int ganlab_adam_dev_f32(float* p, const float* g, float* m, float* v, long long n, const float* lr_bc1_bc2, float beta1,
                        float beta2, float eps, float wd, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || !lr_bc1_bc2) return GANLAB_EINVAL;
  GL_LAUNCH(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, p, g, m, v, n, 0.f, beta1, beta2, eps, wd, 1.f, 1.f,
            lr_bc1_bc2);
  return GL_CHECK_LAUNCH();
}

int ganlab_step_scalars_size(void) { return GANLAB_STEP_SCALARS_BYTES; }

int ganlab_set_step_scalars(void* block, uint64_t rng_base, float f0, float f1, float f2, float f3, float f4, float f5,
                            void* stream) {
  if (!block) return GANLAB_EINVAL;
  GL_LAUNCH(set_scalars_kernel, dim3(1), dim3(64), 0, ST, reinterpret_cast<uint32_t*>(block), rng_base, f0, f1, f2, f3,
            f4, f5);
  return GL_CHECK_LAUNCH();
}

int ganlab_ewma_f32(float* lagged, const float* p, long long n, float beta, void* stream) {
  if (!lagged || !p || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(ewma_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, lagged, p, n, beta);
  return GL_CHECK_LAUNCH();
}

int ganlab_randn_f32(float* out, long long n, uint64_t seed, uint64_t offset, void* stream) {
  if (!out || n <= 0) return GANLAB_EINVAL;
  GL_LAUNCH(randn_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, ST, out, n, seed, offset,
            (const uint64_t*)nullptr);
  return GL_CHECK_LAUNCH();
}

int ganlab_randn_dev_f32(float* out, long long n, uint64_t seed, const void* base, uint64_t delta, void* stream) {
  if (!out || n <= 0 || !base) return GANLAB_EINVAL;
  GL_LAUNCH(randn_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, ST, out, n, seed, delta,
            reinterpret_cast<const uint64_t*>(base));
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
