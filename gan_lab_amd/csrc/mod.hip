// "Deferred InstanceNorm": the generator layer chain of stylegan/architectures.py:497-526 without its normalisation
// passes.  A layer ends in   b = (a - mean[n,c]) * rstd[n,c] * (ys[n,c] + 1) + yb[n,c] = a * s[n,c] + t[n,c]
// (InstanceNorm + AdaIN of the activated tensor a).  Round 1 ran that as its own HBM round trip (read a, write b) and the
// next convolution read b again.  Here the CONSUMER applies it - without touching a single activation:
//   * the scale folds into PER-SAMPLE weights  w_eff[n][tap][ci][co] = w[tap][ci][co] * s[n,ci]  (these kernels process
//     one image per workgroup, so it is a weight pointer per image; StyleGAN2's weight modulation, applied to AdaIN);
//   * the shift is a per-sample bias  sum_ci t[n,ci] * sum_tap w[tap][ci][co]  - except on the one-pixel border, where the
//     taps that fall into the ZERO padding of b must not contribute: a 3x3 table of (row class, column class) per (n, co),
//     `btab`, built on the host from the weight's partial tap sums, added in the epilogue;
//   * toRGB is 1x1 (no padding): per-sample weights and bias only.
// The same forward kernel can also finish the layer it computes (conv -> + noise_w * noise + bias -> LeakyReLU) and
// accumulate the InstanceNorm statistics of ITS output in the epilogue (per-lane fp32 sums of a step's 16 values, fp64
// across steps, fixed-order finish), so a plain 3x3 layer of the generator is ONE pass from a_in to a_out.
// Backward: the input gradient is the plain kernel on the shared weights (it IS d/db, which the InstanceNorm backward
// kernels of round 1 take); the weight gradient is accumulated PER IMAGE (wgrad_roll.hip's kernel, slots grouped by image)
// and recombined with s / t on [N, Cout, Cin, 9] tensors by the caller (gan_lab_amd/ops.py::_ConvMod).
// Geometry: the thin rolling-window layers (Cin, Cout <= 16, W % 64 == 0, H % 4 == 0) - at StyleGAN-1024 the two 1024^2
// layers, half of the generator's activation bytes.
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
  const float* wmod;     // [N][9][16][16] per-sample packed weights (scale and s folded in), or [1][...] with wstride 0
  const float* btab;     // [N][3][3][16] border-class bias, or null
  const float* bias;     // [Cout] (x bias_scale), or null
  const float* noise;    // (N, 1, H, W) or null
  const float* noise_w;  // [Cout]
  float* y;
  double* spart;         // [(n*Cout + co)*chunks + chunk][2] statistics partials, or null
  int N, Cin, Cout, H, W;
  int tiles_x, tiles_y, strips_x, strip;
  long long wstride;
  float bias_scale, slope;
  int act;
};

__global__ __launch_bounds__(256, 3) void conv_fwd_rollmod_kernel(RMArgs p) {
  __shared__ __attribute__((aligned(16))) float ring[RM_SLOTS * RM_SLOT];
  __shared__ double sred[4][16][2];
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
  const float* wn_ = p.wmod + (long long)n0 * p.wstride;
  float wreg[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) wreg[st] = wn_[((st / C4N) * 16 + (st % C4N) * 4 + (lane >> 4)) * 16 + (lane & 15)];
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
      if (tid + i * 256 < RM_ITEMS && k < nrows)
        *reinterpret_cast<float4*>(ring + ((rel0 + k) % RM_SLOTS) * RM_SLOT + (lo[i] & 0xfffff)) = xr[i];
    }
  };

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
    // ---- epilogue of output row oy: border-class bias, noise, bias, LeakyReLU, statistics, 16-byte stores ----
    const int oy = oy_first + 4 * t + wn;
    float bt[3] = {0.f, 0.f, 0.f};
    if (p.btab != nullptr && co_ok) {
      const int ry = oy == 0 ? 0 : (oy == p.H - 1 ? 2 : 1);
      const float* bp = p.btab + ((long long)n0 * 9 + ry * 3) * 16 + co_lane;
      bt[0] = bp[0];
      bt[1] = bp[16];
      bt[2] = bp[32];
    }
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
        const int xx = x0 + r;
        float v = acc[nb][r] + bv + (xx == 0 ? bt[0] : (xx == p.W - 1 ? bt[2] : bt[1])) + nwv * nz[r];
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

// wmod[n][tap][ci16][co16] = scale * w[co][ci][tap] * (s ? s[n][ci] : 1); channel padding is zero
__global__ void rm_pack_kernel(const float* __restrict__ w, const float* __restrict__ s, float* __restrict__ out, int N,
                               int Cout, int Cin, float scale) {
  const long long total = (long long)N * 9 * 256;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int co = (int)(i & 15), ci = (int)((i >> 4) & 15);
    const long long t2 = i >> 8;
    const int tap = (int)(t2 % 9);
    const long long n = t2 / 9;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = scale * w[((long long)co * Cin + ci) * 9 + tap] * (s ? s[n * Cin + ci] : 1.f);
    out[i] = v;
  }
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

int ganlab_mod_conv_pack_f32(const float* w, const float* s, float* out, int N, int Cout, int Cin, float scale,
                             void* stream) {
  if (!w || !out || N <= 0 || Cout <= 0 || Cin <= 0 || Cout > 16 || Cin > 16) return GANLAB_EINVAL;
  const long long total = (long long)N * 9 * 256;
  GL_LAUNCH(rm_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, gl_stream(stream), w, s, out, N, Cout,
            Cin, scale);
  return GL_CHECK_LAUNCH();
}

int ganlab_mod_conv_fwd_f32(const float* x, const float* wmod, int per_sample, const float* btab, const float* bias,
                            const float* noise, const float* noise_w, float* y, float* mean, float* rstd,
                            const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!rm_geom_ok(g) || !x || !wmod || !y || !rm_aligned16(x) || !rm_aligned16(y) || (noise && (!noise_w || !rm_aligned16(noise))))
    return GANLAB_EINVAL;
  RMArgs a{};
  a.x = x; a.wmod = wmod; a.btab = btab; a.bias = bias; a.noise = noise; a.noise_w = noise_w; a.y = y;
  a.N = g->N; a.Cin = g->Cin; a.Cout = g->Cout; a.H = g->Hin; a.W = g->Win;
  a.wstride = per_sample ? 9 * 256 : 0;
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
