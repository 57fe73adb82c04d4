// "Deferred InstanceNorm": the generator layer chain of stylegan/architectures.py:497-526 without its normalisation
// passes.  A layer ends in   b = (a - mean[n,c]) * rstd[n,c] * (ys[n,c] + 1) + yb[n,c] = a * s[n,c] + t[n,c]
// (InstanceNorm + AdaIN of the activated tensor a).  Round 1 ran that as its own HBM round trip (read a, write b) and the
// next convolution read b again.  Here the CONSUMER applies it: every kernel that stages a patch of b through registers
// into LDS computes  a * s + t  on the way (one fma per element, with the per-image (s, t) table in LDS) and leaves the
// elements that fall into the zero padding of b at zero - "affine on load" (conv.hip / conv_s2.hip / conv_s2_roll.hip /
// wgrad_roll.hip, template flag AFF).  Shared packed weights, no border special case, and the weight gradient contracts
// the same on-the-fly b, so nothing is accumulated per image.  (Round 2 folded s into PER-SAMPLE packed weights and t into a
// border-class bias table, with per-image weight gradients recombined on the host: more launches, ATen glue, and only the
// thinnest layers.)
// This file: the thin rolling-window layer (Cin, Cout <= 16, W % 64 == 0, H % 4 == 0 - the 1024^2 layers) that also FINISHES
// the layer it computes (conv -> + noise_w * noise + bias -> LeakyReLU) and accumulates the InstanceNorm statistics of ITS
// output in the epilogue (per-lane fp32 sums of a step's 16 values, fp64 across steps, fixed-order finish), so a plain 3x3
// layer of the generator is ONE pass from a_in to a_out; and toRGB (1x1, linear): per-sample weights w * s and bias
// b + w.t built by a small kernel.
// Backward: the input gradient is the plain kernel on the shared weights (it IS d/db, which the InstanceNorm backward
// kernels of round 1 take); the weight gradient is the rolling-window kernel with AFF on its x operand.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int RM_TW = 64, RM_ROWS = 4, RM_SLOTS = 6, RM_RP = 80, RM_SLOT = 16 * RM_RP, RM_Q = 18;
constexpr int RM_ITEMS = RM_ROWS * 16 * RM_Q;            // float4 items of one 4-row prefetch: 1152
constexpr int RM_PT = (RM_ITEMS + 255) / 256;            // 5
constexpr int RM_LOAD_AT = 6;
constexpr int RM_OOB = (int)0x80000000;

struct RMArgs {
  const float* x;        // (N, Cin, H, W): the producer's activated tensor a
  const float* wp;       // [9][Cin_p][Cout_p] packed weights (ganlab_conv_pack_f32, scale folded in), shared by the images
  const float* aff_s;    // [N][Cin] deferred InstanceNorm + style of the INPUT: b = a * s + t inside the image; or null
  const float* aff_t;
  const float* bias;     // [Cout] (x bias_scale), or null
  const float* noise;    // (N, 1, H, W) or null
  const float* noise_w;  // [Cout]
  float* y;
  double* spart;         // [(n*Cout + co)*chunks + chunk][2] statistics partials, or null
  int N, Cin, Cout, H, W;
  int tiles_x, tiles_y, strips_x, strip;
  int Cin_p, Cout_p;
  float bias_scale, slope;
  int act;
};

__global__ __launch_bounds__(256, 3) void conv_fwd_rollmod_kernel(RMArgs p) {
  __shared__ __attribute__((aligned(16))) float ring[RM_SLOTS * RM_SLOT];
  __shared__ double sred[4][16][2];
  __shared__ float afftab[32];           // s | t of this image's (<= 16) input channels
  constexpr int C4N = 4, NSTEP = 36, PD = 3, NB = 4;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int txi = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int syi = bid % p.strips_x;
  const int n0 = bid / p.strips_x;
  const int step0 = syi * p.strip;
  const int nsteps = min(p.strip, p.tiles_y - step0);
  const int ox0 = txi * RM_TW, oy_first = step0 * RM_ROWS;
  const int plane = p.H * p.W;
  const float* xb = p.x + (long long)n0 * p.Cin * plane;

  int gbase[RM_PT], lo[RM_PT];
#pragma unroll
  for (int i = 0; i < RM_PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % RM_Q;
    const int t = e / RM_Q;
    const int ci = t & 15, k = t >> 4;
    const int vx = ox0 - 4 + 4 * q;
    gbase[i] = (e < RM_ITEMS && ci < p.Cin && (unsigned)vx < (unsigned)p.W) ? (ci * plane + vx) * 4 : RM_OOB;
    lo[i] = (ci * RM_RP + 4 * q) | (k << 20);
  }
  const bool aff = p.aff_s != nullptr;
  if (aff && tid < 32) {
    const int c = tid & 15;
    afftab[tid] = c < p.Cin ? (tid < 16 ? p.aff_s : p.aff_t)[(long long)n0 * p.Cin + c] : 0.f;
  }
  float wreg[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st)
    wreg[st] = p.wp[(long long)((st / C4N) * p.Cin_p + (st % C4N) * 4 + (lane >> 4)) * p.Cout_p + (lane & 15)];
  const int co_lane = lane & 15;
  const bool co_ok = co_lane < p.Cout;
  const float bv = (p.bias != nullptr && co_ok) ? p.bias[co_lane] * p.bias_scale : 0.f;
  const float nwv = (p.noise != nullptr && co_ok) ? p.noise_w[co_lane] : 0.f;
  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) boff[nb] = (lane >> 4) * RM_RP + nb * 16 + (lane & 15) + 3;

  const long long out_plane = (long long)p.H * p.W;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n0 * p.Cout * out_plane, 0, (unsigned)(p.Cout * out_plane * 4), 0x00020000);
  int vo_lane[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    vo_lane[nb] = co_ok ? (int)(((long long)co_lane * out_plane + ox0 + nb * 16 + (lane >> 4) * 4) * 4) : RM_OOB;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  double ds = 0.0, dss = 0.0;

  float4 xr[RM_PT];
  auto load_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < RM_PT; ++i) {
      const int k = lo[i] >> 20;
      const int vy = oy_first - 1 + rel0 + k;
      const bool ok = gbase[i] != RM_OOB && k < nrows && (unsigned)vy < (unsigned)p.H;
      const int off = ok ? gbase[i] + (int)((unsigned)vy * (unsigned)(p.W * 4)) : RM_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto store_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < RM_PT; ++i) {
      const int k = lo[i] >> 20;
      if (tid + i * 256 < RM_ITEMS && k < nrows) {
        float4 v = xr[i];
        if (aff) {            // the load's validity test again: rows / columns / channels outside the image stay zero
          const int vy = oy_first - 1 + rel0 + k, ci = (lo[i] & 0xfffff) / RM_RP;
          const bool ok = gbase[i] != RM_OOB && (unsigned)vy < (unsigned)p.H;
          const float sv = ok ? afftab[ci] : 0.f, tv = ok ? afftab[16 + ci] : 0.f;
          v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
        }
        *reinterpret_cast<float4*>(ring + ((rel0 + k) % RM_SLOTS) * RM_SLOT + (lo[i] & 0xfffff)) = v;
      }
    }
  };

  __syncthreads();           // the (s, t) table
  load_rows(0, 4);
  store_rows(0, 4);
  load_rows(4, 2);
  store_rows(4, 2);
  __syncthreads();
  for (int t = 0; t < nsteps; ++t) {
    {
      int sbase[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) sbase[ky] = ((4 * t + wn + ky) % RM_SLOTS) * RM_SLOT;
      float rb[PD + 1][NB];
      auto fetch = [&](int st, int slot) {
        const int ky = st / (3 * C4N), kx = (st / C4N) % 3, c4 = st % C4N;
        const float* xrow = ring + sbase[ky] + c4 * 4 * RM_RP + kx;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) rb[slot][nb] = xrow[boff[nb]];
      };
#pragma unroll
      for (int st = 0; st < PD; ++st) fetch(st, st % (PD + 1));
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + PD < NSTEP) fetch(st + PD, (st + PD) % (PD + 1));
        const int slot = st % (PD + 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][nb], wreg[st], acc[nb], 0, 0, 0);
        if (st == RM_LOAD_AT) load_rows(4 * t + 6, t + 1 < nsteps ? 4 : 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue of output row oy: noise, bias, LeakyReLU, statistics, 16-byte stores ----
    const int oy = oy_first + 4 * t + wn;
    const int orow = oy * p.W * 4;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int x0 = ox0 + nb * 16 + (lane >> 4) * 4;
      float nz[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.noise != nullptr)
        *reinterpret_cast<float4*>(nz) = *reinterpret_cast<const float4*>(p.noise + (long long)n0 * plane + (long long)oy * p.W + x0);
      u32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nb][r] + bv + nwv * nz[r];
        if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
        ssum += v;
        ssq += v * v;
        o[r] = __float_as_uint(v);
      }
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo_lane[nb] == RM_OOB ? vo_lane[nb] : vo_lane[nb] + orow, 0, 0);
      acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (p.spart != nullptr) {
      ds += (double)ssum;
      dss += (double)ssq;
    }
    __syncthreads();
    store_rows((4 * t + 6), 4);          // (rows past the strip were loaded as zeros and are never read)
    __syncthreads();
  }
  if (p.spart != nullptr) {
    // the four k-groups of a wave hold the same channel: add them, then the four waves through LDS (fixed order)
    ds += __shfl_xor(ds, 16, 64);
    dss += __shfl_xor(dss, 16, 64);
    ds += __shfl_xor(ds, 32, 64);
    dss += __shfl_xor(dss, 32, 64);
    if (lane < 16) {
      sred[wn][lane][0] = ds;
      sred[wn][lane][1] = dss;
    }
    __syncthreads();
    if (tid < 32) {
      const int c = tid >> 1, k = tid & 1;
      if (c < p.Cout) {
        const int chunks = p.tiles_x * p.strips_x, chunk = syi * p.tiles_x + txi;
        p.spart[(((long long)n0 * p.Cout + c) * chunks + chunk) * 2 + k] =
            (sred[0][c][k] + sred[1][c][k]) + (sred[2][c][k] + sred[3][c][k]);
      }
    }
  }
}

// mean / rstd of every (n, c) plane from the chunk partials (fixed order, fp64)
__global__ void rm_stats_finish_kernel(const double* __restrict__ spart, float* __restrict__ mean,
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

// per-sample toRGB operands from the shared weight and the deferred (s, t):  weff[n][ci][4] = scale * w[co][ci] * s[n,ci],
// beff[n][4] = bias[co] * bias_scale + scale * sum_ci w[co][ci] * t[n,ci]      (Cout <= 4, Cin <= 16; one block per image)
__global__ void rm_torgb_prep_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                     const float* __restrict__ s, const float* __restrict__ t, float* __restrict__ weff,
                                     float* __restrict__ beff, int Cin, int Cout, float scale, float bias_scale) {
  const int n = blockIdx.x, i = threadIdx.x;          // 64 threads: (ci = i >> 2, co = i & 3)
  const int ci = i >> 2, co = i & 3;
  if (ci < Cin) weff[((long long)n * Cin + ci) * 4 + co] = co < Cout ? scale * w[co * Cin + ci] * s[(long long)n * Cin + ci] : 0.f;
  if (i < 4) {
    float b = 0.f;
    if (i < Cout) {
      for (int c = 0; c < Cin; ++c) b += w[i * Cin + c] * t[(long long)n * Cin + c];
      b = b * scale + (bias ? bias[i] * bias_scale : 0.f);
    }
    beff[(long long)n * 4 + i] = b;
  }
}

// weight / bias gradient of the modulated toRGB from the per-image cross sums (rm_torgb_cross_kernel: out[n][68]):
//   gw[co][ci] = scale * sum_n ( s[n,ci] * cross[n][ci][co] + t[n,ci] * gsum[n][co] ),  gb[co] = bias_scale * sum_n gsum[n][co]
__global__ void rm_torgb_wgrad_kernel(const float* __restrict__ out, const float* __restrict__ s,
                                      const float* __restrict__ t, float* __restrict__ gw, float* __restrict__ gb, int N,
                                      int Cin, int Cout, float scale, float bias_scale) {
  const int i = threadIdx.x;                           // 64 threads: (ci = i >> 2, co = i & 3)
  const int ci = i >> 2, co = i & 3;
  if (ci < Cin && co < Cout && gw != nullptr) {
    float a = 0.f;
    for (int n = 0; n < N; ++n)
      a += s[(long long)n * Cin + ci] * out[(long long)n * 68 + ci * 4 + co] + t[(long long)n * Cin + ci] * out[(long long)n * 68 + 64 + co];
    gw[co * Cin + ci] = a * scale;
  }
  if (i < Cout && gb != nullptr) {
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += out[(long long)n * 68 + 64 + i];
    gb[i] = a * bias_scale;
  }
}

// s = rstd * (ys + 1), t = yb - mean * s  per (n, c) from the statistics and the style (N, 2C) = [ys | yb] (or null)
__global__ void in_affine_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const float* __restrict__ style, float* __restrict__ s, float* __restrict__ t, int N, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  const float ys = style ? style[(long long)(n * 2 + 0) * C + c] + 1.f : 1.f;
  const float yb = style ? style[(long long)(n * 2 + 1) * C + c] : 0.f;
  const float sv = rstd[i] * ys;
  s[i] = sv;
  t[i] = yb - mean[i] * sv;
}

// ---- toRGB with per-sample weights / bias -----------------------------------------------------------------------
// y[n,co,px] = sum_ci weff[n][ci][co] * x[n,ci,px] + beff[n][co]      (Cout <= 4, weff rows padded to 4)
__global__ __launch_bounds__(256) void rm_torgb_fwd_kernel(const float* __restrict__ x, const float* __restrict__ weff,
                                                           const float* __restrict__ beff, float* __restrict__ y, int N,
                                                           int Cin, int Cout, long long hw4) {
  const unsigned n = blockIdx.y;
  const float* wn = weff + (long long)n * Cin * 4;
  float b[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) b[co] = beff[(long long)n * 4 + co];
  const float4* xb = reinterpret_cast<const float4*>(x) + (long long)n * Cin * hw4;
  float4* yb = reinterpret_cast<float4*>(y) + (long long)n * Cout * hw4;
  for (long long q = blockIdx.x * 256LL + threadIdx.x; q < hw4; q += (long long)gridDim.x * 256) {
    float4 a[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) a[co] = float4{b[co], b[co], b[co], b[co]};
    for (int ci = 0; ci < Cin; ++ci) {
      const float4 v = xb[(long long)ci * hw4 + q];
      const float4 wv = *reinterpret_cast<const float4*>(wn + ci * 4);
      a[0].x += wv.x * v.x; a[0].y += wv.x * v.y; a[0].z += wv.x * v.z; a[0].w += wv.x * v.w;
      a[1].x += wv.y * v.x; a[1].y += wv.y * v.y; a[1].z += wv.y * v.z; a[1].w += wv.y * v.w;
      a[2].x += wv.z * v.x; a[2].y += wv.z * v.y; a[2].z += wv.z * v.z; a[2].w += wv.z * v.w;
      a[3].x += wv.w * v.x; a[3].y += wv.w * v.y; a[3].z += wv.w * v.z; a[3].w += wv.w * v.w;
    }
#pragma unroll
    for (int co = 0; co < 4; ++co)
      if (co < Cout) yb[(long long)co * hw4 + q] = a[co];
  }
}

// per-image cross sums: part[(n*blocks + blk)][ci 16][5] = sum_px x[n,ci,px] * {gy[n,0..3,px], 1}   (Cin <= 16, Cout <= 4)
__global__ __launch_bounds__(256) void rm_torgb_cross_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             float* __restrict__ part, int Cin, int Cout, long long hw4) {
  __shared__ float red[4][80];
  const unsigned n = blockIdx.y;
  const float4* xb = reinterpret_cast<const float4*>(x) + (long long)n * Cin * hw4;
  const float4* gb = reinterpret_cast<const float4*>(gy) + (long long)n * Cout * hw4;
  float acc[16][4], gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[c][s] = 0.f;
  for (long long q = blockIdx.x * 256LL + threadIdx.x; q < hw4; q += (long long)gridDim.x * 256) {
    float4 g[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) g[s] = s < Cout ? gb[(long long)s * hw4 + q] : float4{0.f, 0.f, 0.f, 0.f};
    float4 xv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) xv[c] = c < Cin ? xb[(long long)c * hw4 + q] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) gs[s] += (g[s].x + g[s].y) + (g[s].z + g[s].w);
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[c][s] += (xv[c].x * g[s].x + xv[c].y * g[s].y) + (xv[c].z * g[s].z + xv[c].w * g[s].w);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float v = gl_wave_sum(acc[c][s]);
      if (lane == 0) red[wv][c * 4 + s] = v;
    }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float v = gl_wave_sum(gs[s]);
    if (lane == 0) red[wv][64 + s] = v;
  }
  __syncthreads();
  if (threadIdx.x < 68)
    part[((long long)n * gridDim.x + blockIdx.x) * 68 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[n][68] = sum_blk part[n][blk][68]  (fixed order)
__global__ void rm_torgb_cross_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int blocks) {
  const int n = blockIdx.x, t = threadIdx.x;
  if (t >= 68) return;
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 1 < blocks; k += 2) {
    s0 += part[((long long)n * blocks + k) * 68 + t];
    s1 += part[((long long)n * blocks + k + 1) * 68 + t];
  }
  if (k < blocks) s0 += part[((long long)n * blocks + k) * 68 + t];
  out[(long long)n * 68 + t] = s0 + s1;
}

inline bool rm_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool rm_geom_ok(const ganlab_conv_geom* g) {
  return g && g->N > 0 && g->Cin > 0 && g->Cout > 0 && g->Cin <= 16 && g->Cout <= 16 && g->ks == 3 && g->pad == 1 &&
         g->up == 0 && g->pool == 0 && g->Win % RM_TW == 0 && g->Hin % RM_ROWS == 0 && g->Hin >= 8 &&
         (long long)16 * g->Hin * g->Win * 4 < 0x7fffffffLL;
}

void rm_grid(const ganlab_conv_geom* g, RMArgs& a) {
  a.tiles_x = g->Win / RM_TW;
  a.tiles_y = g->Hin / RM_ROWS;
  const long long cols = (long long)a.tiles_x * g->N;
  int kk = 1;     // strips per column: >= ~4096 workgroups, >= 8 steps each (conv.hip's rolling-window rule)
  while (kk < a.tiles_y && cols * kk < 4096 && (a.tiles_y + kk) / (kk + 1) >= 8) ++kk;
  a.strip = (a.tiles_y + kk - 1) / kk;
  a.strips_x = (a.tiles_y + a.strip - 1) / a.strip;
}

}  // namespace

extern "C" {

int ganlab_mod_conv_supported(const ganlab_conv_geom* g) { return rm_geom_ok(g) ? 1 : 0; }

/* number of statistics chunks per (n, co) plane the forward writes (for the workspace: N*Cout*chunks*2 doubles) */
int ganlab_mod_conv_stat_chunks(const ganlab_conv_geom* g) {
  if (!rm_geom_ok(g)) return 0;
  RMArgs a{};
  rm_grid(g, a);
  return a.tiles_x * a.strips_x;
}

int ganlab_mod_conv_fwd_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                            const float* noise, const float* noise_w, float* y, float* mean, float* rstd,
                            const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!rm_geom_ok(g) || !x || !wp || !y || !rm_aligned16(x) || !rm_aligned16(y) || (noise && (!noise_w || !rm_aligned16(noise))) ||
      ((aff_s == nullptr) != (aff_t == nullptr)))
    return GANLAB_EINVAL;
  RMArgs a{};
  a.x = x; a.wp = wp; a.aff_s = aff_s; a.aff_t = aff_t; a.bias = bias; a.noise = noise; a.noise_w = noise_w; a.y = y;
  a.N = g->N; a.Cin = g->Cin; a.Cout = g->Cout; a.H = g->Hin; a.W = g->Win;
  a.Cin_p = 16; a.Cout_p = 64;      // ganlab_conv_pack_f32's padding of a (<= 16) x (<= 16) 3x3 weight
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  rm_grid(g, a);
  const int chunks = a.tiles_x * a.strips_x;
  const long long planes = (long long)g->N * g->Cout;
  if (mean != nullptr) {
    if (!rstd || !workspace || workspace_bytes < (size_t)planes * chunks * 2 * sizeof(double)) return GANLAB_EWORKSPACE;
    a.spart = reinterpret_cast<double*>(workspace);
  }
  const long long grid = (long long)a.tiles_x * a.strips_x * g->N;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  hipStream_t st = gl_stream(stream);
  GL_LAUNCH(conv_fwd_rollmod_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
  if (mean != nullptr)
    GL_LAUNCH(rm_stats_finish_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, st, (const double*)a.spart,
              mean, rstd, planes, chunks, 1.0 / ((double)g->Hin * g->Win), eps);
  return GL_CHECK_LAUNCH();
}

int ganlab_mod_torgb_fwd_f32(const float* x, const float* weff, const float* beff, float* y, int N, int Cin, int Cout,
                             long long HW, void* stream) {
  if (!x || !weff || !beff || !y || N <= 0 || Cin <= 0 || Cout <= 0 || Cout > 4 || (HW & 3) || !rm_aligned16(x) ||
      !rm_aligned16(y) || !rm_aligned16(weff))
    return GANLAB_EINVAL;
  const long long hw4 = HW / 4;
  long long blocks = (hw4 + 255) / 256;
  if (blocks > 128) blocks = 128;
  GL_LAUNCH(rm_torgb_fwd_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, gl_stream(stream), x, weff, beff, y, N,
            Cin, Cout, hw4);
  return GL_CHECK_LAUNCH();
}

/* per-sample toRGB operands: weff [N][Cin][4], beff [N][4] (see rm_torgb_prep_kernel); w is the (Cout, Cin, 1, 1) parameter */
int ganlab_mod_torgb_prep_f32(const float* w, const float* bias, const float* s, const float* t, float* weff, float* beff,
                              int N, int Cin, int Cout, float scale, float bias_scale, void* stream) {
  if (!w || !s || !t || !weff || !beff || N <= 0 || Cin <= 0 || Cin > 16 || Cout <= 0 || Cout > 4) return GANLAB_EINVAL;
  GL_LAUNCH(rm_torgb_prep_kernel, dim3((unsigned)N), dim3(64), 0, gl_stream(stream), w, bias, s, t, weff, beff, Cin, Cout,
            scale, bias_scale);
  return GL_CHECK_LAUNCH();
}

/* gw (Cout, Cin) and gb (Cout) of the modulated toRGB from ganlab_mod_torgb_cross_f32's N x 68 output; either may be null */
int ganlab_mod_torgb_wgrad_f32(const float* cross, const float* s, const float* t, float* gw, float* gb, int N, int Cin,
                               int Cout, float scale, float bias_scale, void* stream) {
  if (!cross || !s || !t || N <= 0 || Cin <= 0 || Cin > 16 || Cout <= 0 || Cout > 4) return GANLAB_EINVAL;
  GL_LAUNCH(rm_torgb_wgrad_kernel, dim3(1), dim3(64), 0, gl_stream(stream), cross, s, t, gw, gb, N, Cin, Cout, scale,
            bias_scale);
  return GL_CHECK_LAUNCH();
}

/* s = rstd * (ys + 1), t = yb - mean * s per (n, c): the deferred InstanceNorm + style as a per-plane affine */
int ganlab_in_affine_f32(const float* mean, const float* rstd, const float* style, float* s, float* t, int N, int C,
                         void* stream) {
  if (!mean || !rstd || !s || !t || N <= 0 || C <= 0) return GANLAB_EINVAL;
  const long long total = (long long)N * C;
  GL_LAUNCH(in_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, gl_stream(stream), mean, rstd, style, s, t,
            N, C);
  return GL_CHECK_LAUNCH();
}

size_t ganlab_mod_torgb_cross_workspace(int N) { return (size_t)N * 64 * 68 * sizeof(float); }

/* out[n][ci 16][4] (cross sums sum_px x[n,ci] * gy[n,co]) followed by out[n][64 + co] = sum_px gy[n,co]: N x 68 floats */
int ganlab_mod_torgb_cross_f32(const float* x, const float* gy, float* out, int N, int Cin, int Cout, long long HW,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !gy || !out || N <= 0 || Cin <= 0 || Cin > 16 || Cout <= 0 || Cout > 4 || (HW & 3) || !rm_aligned16(x) ||
      !rm_aligned16(gy))
    return GANLAB_EINVAL;
  if (!workspace || workspace_bytes < ganlab_mod_torgb_cross_workspace(N)) return GANLAB_EWORKSPACE;
  const long long hw4 = HW / 4;
  int blocks = (int)((hw4 + 256 * 16 - 1) / (256 * 16));
  if (blocks > 64) blocks = 64;
  if (blocks < 1) blocks = 1;
  hipStream_t st = gl_stream(stream);
  GL_LAUNCH(rm_torgb_cross_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, st, x, gy, (float*)workspace, Cin,
            Cout, hw4);
  GL_LAUNCH(rm_torgb_cross_finish_kernel, dim3((unsigned)N), dim3(128), 0, st, (const float*)workspace, out, blocks);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
