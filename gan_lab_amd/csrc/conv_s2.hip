// Stride-2 fused convolutions on the fp32 MFMA (v_mfma_f32_16x16x4_f32), NCHW.
//
// Two layer patterns of the reference carry 2/3 of all conv FLOPs of the G+D step:
//   D "down" layer : AvgPool2d(2)( conv3x3(x, W, pad 1) )            progan/architectures.py:261-284
//   G "up"   layer : conv3x3( Upsample(2x nearest)(x), W, pad 1 )    stylegan/architectures.py:292-334
// Both are linear maps that collapse EXACTLY to a 4x4 stride-2 kernel K4 = M (x) M applied to W
// (M = 4x3 combination matrix, below):
//   down:  y[Y,X]          = sum_{a,b<4} K4d[a][b] * xpad[2Y+a-1, 2X+b-1]          ("S": strided conv)
//   up  :  y[2Y+py,2X+px]  = sum_{Y',X'} K4u[2(Y-Y')+py+1][2(X-X')+px+1] * x[Y',X'] ("T": its transpose)
// which needs 16 MACs per (ci,co) per low-res pixel instead of 36 (9 taps at 4 high-res pixels):
// 2.25x fewer matrix-core FLOPs, no full-resolution intermediate, and the pool / upsample kernels
// disappear.  The three derivatives of each pattern are again S / T / a 16-tap weight gradient "W":
//   down: fwd = S(K4d), dgrad = T(K4d), wgrad = W(low = gy, high = x)
//   up  : fwd = T(K4u), dgrad = S(K4u), wgrad = W(low = x, high = gy)   (then gw = M^T gK4 M)
// Results differ from the reference only by fp32 re-association (weights are pre-summed).
//
// S stages the high-res patch into LDS split by pixel parity (space-to-depth on the fly) so that the
// MFMA B-operand reads are stride-1; T keeps 4 output-phase accumulators per low-res pixel and stores
// float2 {px=0, px=1} pairs (128-byte coalesced rows); all staging is 16-byte vectorised with the
// next K-chunk prefetched into registers during the MFMA phase (same scheme as conv.hip).
#include "common.h"

#include <stdlib.h>

namespace {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int round_up_c(int v, int m) { return (v + m - 1) / m * m; }
constexpr int ceil_div_c(int a, int b) { return (a + b - 1) / b; }
constexpr int pad_mod32(int v, int r) { return v + ((r - (v % 32)) + 32) % 32; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// tap a in 0..3 of the stride-2 kernel reads high row 2Y + a - 1 = 2(Y + DY[a]) + PY[a]
__device__ __constant__ const int kDY[4] = {-1, 0, 0, 1};
__device__ __constant__ const int kPY[4] = {1, 0, 1, 0};
constexpr int cDY(int a) { return a == 0 ? -1 : (a == 3 ? 1 : 0); }
constexpr int cPY(int a) { return (a == 0 || a == 2) ? 1 : 0; }

// ------------------------------------------------------------------------------------------------
// pack: OIHW 3x3 -> K4 packed [16 taps][rows_p][cols_p];  K4[a][b] = scale * sum M[a][ky] M[b][kx] w[ky][kx]
//   down: M = 0.25-normalised {D0={0}, D1={0,1}, D2={1,2}, D3={2}}   (0.25 overall -> 0.5 per dim)
//   up  : M = {U0={2}, U1={1,2}, U2={0,1}, U3={0}}
//   transpose = 1: rows = co, cols = ci (the operator consumes gy)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float comb(int up, int a, int k) {
  if (up) {
    const int lo = (a == 0) ? 2 : (a == 1 ? 1 : 0), hi = (a == 0) ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 0));
    return (k >= lo && k <= hi) ? 1.f : 0.f;
  }
  const int lo = (a <= 1) ? 0 : (a == 2 ? 1 : 2), hi = (a == 0) ? 0 : (a == 1 ? 1 : 2);
  return (k >= lo && k <= hi) ? 0.5f : 0.f;
}

__global__ void pack_s2_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int rows,
                               int cols, int rows_p, int cols_p, int up, int transpose, float scale) {
  const long long total = 16LL * rows_p * cols_p;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e % cols_p);
    const long long t = e / cols_p;
    const int row = (int)(t % rows_p), tap = (int)(t / rows_p);
    float v = 0.f;
    if (row < rows && col < cols) {
      const int co = transpose ? row : col, ci = transpose ? col : row;
      const int a = tap >> 2, b = tap & 3;
      const float* ws = w + ((long long)co * Cin + ci) * 9;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v += comb(up, a, ky) * comb(up, b, kx) * ws[ky * 3 + kx];
      v *= scale;
    }
    out[e] = v;
  }
}

// ================================================================================================
// S : strided 4x4 conv.  in: high-res (N, Cin, H, W) ; out: low-res (N, Cout, H/2, W/2)
//   tile = 32x8 low-res pixels x CO_T channels; LDS X = [4 parities][CI_T][PL] (parity planes of the
//   high-res patch, (TH+2) x (TW+4) low-res positions each), W = [16 taps][CI_T][COP]
// ================================================================================================
struct S2Args {
  const float* x;     // S: high-res input ; T: low-res input
  const float* wp;    // [16][Cin_p][Cout_p]
  const float* bias;
  float* y;
  int N, Cin, Cout;
  int Hl, Wl;         // LOW resolution (high = 2x)
  int Cin_p, Cout_p;
  int tiles_x, tiles_y, tiles_co;
  float bias_scale, slope;
  int act;
  // T only - deferred InstanceNorm (ops.Deferred): x is a generator layer's activated output a, the operand is
  // b = a * s[n,ci] + t[n,ci] inside the image and 0 in the padding; applied between the prefetch registers and LDS
  const float* aff_s;   // [N][Cin] or null
  const float* aff_t;
};

// TWL_ = 5: 32x8 low-res tiles; TWL_ = 4: 16x16 tiles for 16-pixel-wide outputs (a 32-wide tile would be half empty:
// the 512->512 down layer at 32^2 ran at 63 TFLOP/s executed, its neighbours at 120+)
#ifndef GL_S2_FRAG_PREFETCH
#define GL_S2_FRAG_PREFETCH 1
#endif
#ifndef GL_S2_UP_NBL1
#define GL_S2_UP_NBL1 0
#endif
template <int MB_, int TWL_ = 5, int NB_ = 4>
struct SCfg {
  static constexpr int MB = MB_, NB = NB_, TWL = TWL_, TW = 1 << TWL_, TH = 64 * NB_ / TW;
  static constexpr int CO_T = 16 * MB_;
  static constexpr int CI_T = 4;
  static constexpr int RPL = TW + 4, RL = TH + 2;             // parity-plane geometry (low-res units)
  static constexpr int PL = pad_mod32(RL * RPL, 16);
  static constexpr int HROWS = 2 * RL, HROW4 = (2 * TW + 8) / 4;   // high-res rows / float4 per row
  static constexpr int COP = pad_mod32(CO_T, 16);
  static constexpr int XS = 4 * CI_T * PL, WS = 16 * CI_T * COP;
  static constexpr int NXI = CI_T * HROWS * HROW4, XPT = ceil_div_c(NXI, 256);
  static constexpr int NWI = 16 * CI_T * CO_T / 4, WPT = ceil_div_c(NWI, 256);
};

template <class Cfg>
__global__ __launch_bounds__(256, Cfg::NB >= 4 ? 2 : 3) void conv_s2_down_kernel(S2Args p) {
  constexpr int MB = Cfg::MB, NB = Cfg::NB, CI_T = Cfg::CI_T, PL = Cfg::PL, RPL = Cfg::RPL, COP = Cfg::COP;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, CO_T = Cfg::CO_T, XPT = Cfg::XPT, WPT = Cfg::WPT;
  // SPIPE (the 64-channel x 128-pixel tile): TWO images of the patch, one of the weights (47 KB: three workgroups per CU stay) - see
  // the chunk loop
#ifndef GL_S2_DOWN_PIPE
#define GL_S2_DOWN_PIPE 1
#endif
  constexpr bool SPIPE = GL_S2_DOWN_PIPE && GL_ACC_DUMP && GL_S2_FRAG_PREFETCH && MB == 4 && NB == 2 && CI_T == 4;
  __shared__ __attribute__((aligned(16))) float smem[(SPIPE ? 2 : 1) * Cfg::XS + Cfg::WS];
  float* Xs = smem;
  float* Ws = smem + (SPIPE ? 2 : 1) * Cfg::XS;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int txi = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tyi = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = co_t * CO_T, ox0 = txi * TW, oy0 = tyi * TH;
  const int H = 2 * p.Hl, W = 2 * p.Wl, plane = H * W;
  const float* xb = p.x + (long long)n * p.Cin * plane;

  // staging descriptors: item = one high-res float4
  int xg[XPT], xl[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % Cfg::HROW4;
    const int t = e / Cfg::HROW4;
    const int hr = t % Cfg::HROWS, ci = t / Cfg::HROWS;
    const int gy_ = 2 * oy0 - 2 + hr, gx_ = 2 * ox0 - 4 + 4 * q;
    const int ly = hr >> 1, py = hr & 1;
    xl[i] = ((py * 2) * CI_T + ci) * PL + ly * RPL + 2 * q;  // px = 0 plane; px = 1 plane is CI_T*PL further
    xl[i] |= ci << 20;
    xg[i] = (e < Cfg::NXI && (unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W)
                ? ci * plane + gy_ * W + gx_ : -1;
    if (e >= Cfg::NXI) xl[i] = -1;
  }
  int wg[WPT], wl[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int e = tid + i * 256;
    const int c4 = e % (CO_T / 4);
    const int t = e / (CO_T / 4);
    const int ci = t % CI_T, tap = t / CI_T;
    wl[i] = (tap * CI_T + ci) * COP + 4 * c4;
    wg[i] = e < Cfg::NWI ? (tap * p.Cin_p + ci) * p.Cout_p + co0 + 4 * c4 : -1;
  }
  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane & 15);
    const int ty = j >> Cfg::TWL, tx = j & (TW - 1);
    boff[nb] = (ty + 1) * RPL + tx + 2 + (lane >> 4) * PL;
  }
  const int aoff = (lane >> 4) * COP + (lane & 15);

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // second-level sums of the accumulation chains (conv.hip, GL_ACC_DUMP): a 16-tap output contracts 16 * Cin terms
  // (three 64-term chunks per dump: chains of 192 + 16 Cin / 192 second-level terms stay within 1.25x of a 144-term chain's
  // rounding error for every Cin <= 512, with a third fewer passes over the accumulators)
  constexpr int DUMP = GL_ACC_DUMP ? (GL_ACC_DUMP_TERMS + 48) / (16 * CI_T) : 0;
  [[maybe_unused]] f32x4 acc2[DUMP ? MB : 1][DUMP ? NB : 1];
  [[maybe_unused]] int since_dump = 0;
  if constexpr (DUMP > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc2[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  float4 xr[XPT], wr[WPT];
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wp), 0, (unsigned)(16 * p.Cin_p * p.Cout_p * 4), 0x00020000);
  auto load = [&](int ci0) {
    // buffer descriptors (this image's Cin planes; the packed weights): items outside the image / past Cin get the
    // offset 0x80000000 and read as zeros from the bounds check - straight-line code instead of exec-mask branches
    const int xs = ci0 * plane * 4, wsoff = ci0 * p.Cout_p * 4;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int ci = (xl[i] >> 20) & 0x3ff;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
          rs_x, (xg[i] >= 0 && ci0 + ci < p.Cin) ? xg[i] * 4 : (int)0x80000000, xs, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wg[i] >= 0 ? wg[i] * 4 : (int)0x80000000, wsoff, 0);
      wr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  // ---- SPIPE: the chunk in two tap halves (tap rows a = 0, 1 | 2, 3; 64 MFMAs per wave each) with a barrier behind each, as
  // before - but nothing is stored BETWEEN barriers any more.  During half A of chunk k: this chunk's weight rows a = 2, 3 (read
  // from half B on; their LDS rows died at the previous barrier) go to LDS and are reloaded for chunk k+1.  During half B: the patch of
  // chunk k+1 goes to the OTHER patch image, chunk k+1's weight rows a = 0, 1 to the rows half A just finished with, and both
  // register sets are refilled for chunk k+2.  Every store / load is a piece behind a K-step's 8 MFMAs; every load is consumed two
  // halves (128 MFMAs per wave) later.  (conv.hip, conv_fwd_kernel PIPE, for the measurements that led here.)
  if constexpr (SPIPE) {
    static_assert(!SPIPE || (XPT == 4 && WPT == 4 && Cfg::NWI == 4 * 256), "piece schedule");
    constexpr int NOITEM = (int)0x80000000, XSZ = Cfg::XS;
    const int nch = p.Cin_p / CI_T;
    auto load_x1 = [&](int k, int i) {                    // chunk k's patch item i
      const int ci = (xl[i] >> 20) & 0x3ff, c0 = k * CI_T;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
          rs_x, (k < nch && xg[i] >= 0 && c0 + ci < p.Cin) ? xg[i] * 4 : NOITEM, c0 * plane * 4, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto load_w1 = [&](int k, int i) {                    // chunk k's weight item i (i < 2: tap rows 0, 1)
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, k < nch ? wg[i] * 4 : NOITEM, k * CI_T * p.Cout_p * 4, 0);
      wr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto store_x1 = [&](int img, int i) {
      if (xl[i] != -1) {
        const int l = img * XSZ + (xl[i] & 0xfffff);
        *reinterpret_cast<float2*>(Xs + l) = float2{xr[i].x, xr[i].z};              // px = 0
        *reinterpret_cast<float2*>(Xs + l + CI_T * PL) = float2{xr[i].y, xr[i].w};  // px = 1
      }
    };
    auto store_w1 = [&](int i) { *reinterpret_cast<float4*>(Ws + wl[i]) = wr[i]; };
    // prologue: chunk 0 complete except its weight rows 2, 3 (registers); chunk 1's patch and weight rows 0, 1 in registers
#pragma unroll
    for (int i = 0; i < XPT; ++i) load_x1(0, i);
#pragma unroll
    for (int i = 0; i < WPT; ++i) load_w1(0, i);
#pragma unroll
    for (int i = 0; i < XPT; ++i) store_x1(0, i);
    store_w1(0);
    store_w1(1);
#pragma unroll
    for (int i = 0; i < XPT; ++i) load_x1(1, i);
    load_w1(1, 0);
    load_w1(1, 1);
    __syncthreads();
    constexpr int NKH = 8 * (CI_T / 4);                   // K-steps of a tap half
    for (int k = 0; k < nch; ++k) {
      const float* xs = Xs + (k & 1) * XSZ;
      float av[2][MB], bv[2][NB];
      auto fetch = [&](int kk, int st) {
        const int tap = kk / (CI_T / 4), c4 = kk % (CI_T / 4), a = tap >> 2, b = tap & 3;
        const int dy = (a == 0) ? -1 : (a == 3 ? 1 : 0), py = (a == 0 || a == 2) ? 1 : 0;
        const int dx = cDY(b), px = cPY(b);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[st][mb] = Ws[a * (4 * CI_T * COP) + aoff + (b * CI_T + c4 * 4) * COP + mb * 16];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          bv[st][nb] = xs[(py * 2) * CI_T * PL + dy * RPL + (px * CI_T + c4 * 4) * PL + boff[nb] + dx];
      };
      auto piece_a = [&](int j) {          // half A: weight rows 2, 3 of THIS chunk to LDS, then those of the next into the registers
        if (j < 2) store_w1(2 + j);
        else if (j < 4) load_w1(k + 1, j);
      };
      auto piece_b = [&](int j) {          // half B: chunk k+1's patch and weight rows 0, 1 to LDS; chunk k+2's into the registers
        if (j < 4) store_x1((k + 1) & 1, j);
        if (j < 2) store_w1(j);
        if (j >= 4) load_x1(k + 2, j - 4);
        if (j >= 4 && j < 6) load_w1(k + 2, j - 4);
      };
      fetch(0, 0);
#pragma unroll
      for (int kk = 0; kk < 2 * NKH; ++kk) {
        if (kk + 1 < 2 * NKH && kk + 1 != NKH) fetch(kk + 1, (kk + 1) & 1);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk & 1][mb], bv[kk & 1][nb], acc[mb][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kk < NKH) piece_a(kk); else piece_b(kk - NKH);
        __builtin_amdgcn_sched_barrier(0);
        if (kk == NKH - 1) {
          __syncthreads();                 // weight rows 2, 3 of this chunk are in; rows 0, 1 are free
          fetch(NKH, NKH & 1);
        }
      }
      if (++since_dump == DUMP && k + 1 < nch) {
        since_dump = 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc2[mb][nb] += acc[mb][nb];
            acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      __syncthreads();                     // chunk k+1's patch image and weight rows 0, 1 are in; this chunk's are free
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] += acc2[mb][nb];
  } else {
  load(0);
  for (int ci0 = 0; ci0 < p.Cin_p; ci0 += CI_T) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      if (xl[i] != -1) {
        const int l = xl[i] & 0xfffff;
        *reinterpret_cast<float2*>(Xs + l) = float2{xr[i].x, xr[i].z};              // px = 0
        *reinterpret_cast<float2*>(Xs + l + CI_T * PL) = float2{xr[i].y, xr[i].w};  // px = 1
      }
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      if (tid + i * 256 < Cfg::NWI) *reinterpret_cast<float4*>(Ws + wl[i]) = wr[i];
    __syncthreads();
    if (ci0 + CI_T < p.Cin_p) load(ci0 + CI_T);   // (issuing it inside the tap loop costs 20 VGPRs = the third workgroup)
    if constexpr (GL_S2_FRAG_PREFETCH) {
      // operand fragments of K-step k+1 requested before the MFMAs of step k (conv.hip, mfma_chunk): two waves per SIMD do
      // not cover an exposed LDS latency per group of MFMAs
      constexpr int NK = 16 * (CI_T / 4);
      float av[2][MB], bv[2][NB];
      auto fetch = [&](int k, int st) {
        const int tap = k / (CI_T / 4), c4 = k % (CI_T / 4), a = tap >> 2, b = tap & 3;
        const int dy = (a == 0) ? -1 : (a == 3 ? 1 : 0), py = (a == 0 || a == 2) ? 1 : 0;
        const int dx = cDY(b), px = cPY(b);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[st][mb] = Ws[a * (4 * CI_T * COP) + aoff + (b * CI_T + c4 * 4) * COP + mb * 16];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          bv[st][nb] = Xs[(py * 2) * CI_T * PL + dy * RPL + (px * CI_T + c4 * 4) * PL + boff[nb] + dx];
      };
      fetch(0, 0);
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (k + 1 < NK) fetch(k + 1, (k + 1) & 1);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k & 1][mb], bv[k & 1][nb], acc[mb][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 1
      for (int a = 0; a < 4; ++a) {
        const int dy = (a == 0) ? -1 : (a == 3 ? 1 : 0), py = (a == 0 || a == 2) ? 1 : 0;
        const float* wrow = Ws + a * (4 * CI_T * COP) + aoff;
        const float* xrow = Xs + (py * 2) * CI_T * PL + dy * RPL;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int dx = cDY(b), px = cPY(b);
#pragma unroll
          for (int c4 = 0; c4 < CI_T / 4; ++c4) {
            float av[MB], bv[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = wrow[(b * CI_T + c4 * 4) * COP + mb * 16];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = xrow[(px * CI_T + c4 * 4) * PL + boff[nb] + dx];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (DUMP > 0) {
      if (++since_dump == DUMP && ci0 + CI_T < p.Cin_p) {
        since_dump = 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc2[mb][nb] += acc[mb][nb];
            acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
  }
  if constexpr (DUMP > 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] += acc2[mb][nb];
  }
  }      // (!SPIPE)
  // epilogue (low-res): + bias, activation
  const long long oplane = (long long)p.Hl * p.Wl;
  float bvv[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + mb * 16 + (lane >> 4) * 4 + r;
      bvv[mb][r] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] * p.bias_scale : 0.f;
    }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane & 15);
    const int oy = oy0 + (j >> Cfg::TWL), ox = ox0 + (j & (TW - 1));
    if (oy >= p.Hl || ox >= p.Wl) continue;
    float* dst = p.y + ((long long)n * p.Cout + co0 + (lane >> 4) * 4) * oplane + (long long)oy * p.Wl + ox;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + mb * 16 + (lane >> 4) * 4 + r;
        if (co < p.Cout) {
          float v = acc[mb][nb][r] + bvv[mb][r];
          if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
          dst[(long long)(mb * 16 + r) * oplane] = v;
        }
      }
  }
}

// ================================================================================================
// T : transposed stride-2 4x4 conv.  in: low-res (N, Cin, Hl, Wl) ; out: high-res (N, Cout, 2Hl, 2Wl)
//   out[2Y+py, 2X+px] = sum_{(dy,a) in taps(py)} sum_{(dx,b) in taps(px)} K4[a][b] * x[Y+dy, X+dx]
//   taps(0) = {(-1,3), (0,1)}   taps(1) = {(0,2), (+1,0)}
//   tile = TWl x THl low-res pixels (NBL blocks of 16 per wave), 4 phase accumulators per block
// ================================================================================================
template <int MB_, int NBL_>
struct TCfg {
  static constexpr int MB = MB_, NBL = NBL_;
  static constexpr int TW = NBL_ == 4 ? 32 : 16, TH = NBL_ == 1 ? 4 : 8, TWL = NBL_ == 4 ? 5 : 4;
  static constexpr int CO_T = 16 * MB_;
  static constexpr int CI_T = 8;
  static constexpr int RP = TW + 8, R = TH + 2;
  static constexpr int PLANE = pad_mod32(R * RP, 16);
  static constexpr int ROW4 = RP / 4;
  static constexpr int COP = pad_mod32(CO_T, 16);
  static constexpr int XS = CI_T * PLANE, WS = 16 * CI_T * COP;
  static constexpr int NXI = CI_T * R * ROW4, XPT = ceil_div_c(NXI, 256);
  static constexpr int NWI = 16 * CI_T * CO_T / 4, WPT = ceil_div_c(NWI, 256);
};

constexpr int S2_AFF_MAXC = 512;
template <class Cfg, bool AFF = false>
__global__ __launch_bounds__(256, (Cfg::MB <= 2 && !GL_ACC_DUMP ? 3 : 2)) void conv_s2_up_kernel(S2Args p) {
  constexpr int MB = Cfg::MB, NBL = Cfg::NBL, CI_T = Cfg::CI_T, PLANE = Cfg::PLANE, RP = Cfg::RP, COP = Cfg::COP;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, CO_T = Cfg::CO_T, XPT = Cfg::XPT, WPT = Cfg::WPT;
  __shared__ __attribute__((aligned(16))) float smem[Cfg::XS + Cfg::WS + (AFF ? 2 * S2_AFF_MAXC : 0)];
  float* Xs = smem;
  float* Ws = smem + Cfg::XS;
  [[maybe_unused]] float* afftab = smem + Cfg::XS + Cfg::WS;      // AFF: s[0 .. Cin) | t at + S2_AFF_MAXC of THIS image
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int txi = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tyi = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = co_t * CO_T, ox0 = txi * TW, oy0 = tyi * TH;
  const int plane = p.Hl * p.Wl;
  const float* xb = p.x + (long long)n * p.Cin * plane;

  int xg[XPT], xl[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % Cfg::ROW4;
    const int t = e / Cfg::ROW4;
    const int r = t % Cfg::R, ci = t / Cfg::R;
    const int gy_ = oy0 - 1 + r, gx_ = ox0 - 4 + 4 * q;
    xl[i] = (ci * PLANE + r * RP + 4 * q) | (ci << 20);
    xg[i] = (e < Cfg::NXI && (unsigned)gy_ < (unsigned)p.Hl && (unsigned)gx_ < (unsigned)p.Wl)
                ? ci * plane + gy_ * p.Wl + gx_ : -1;
    if (e >= Cfg::NXI) xl[i] = -1;
  }
  int wg[WPT], wl[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int e = tid + i * 256;
    const int c4 = e % (CO_T / 4);
    const int t = e / (CO_T / 4);
    const int ci = t % CI_T, tap = t / CI_T;
    wl[i] = (tap * CI_T + ci) * COP + 4 * c4;
    wg[i] = e < Cfg::NWI ? (tap * p.Cin_p + ci) * p.Cout_p + co0 + 4 * c4 : -1;
  }
  int boff[NBL];
#pragma unroll
  for (int nb = 0; nb < NBL; ++nb) {
    const int j = wn * (16 * NBL) + nb * 16 + (lane & 15);
    const int ty = j >> Cfg::TWL, tx = j & (TW - 1);
    boff[nb] = (ty + 1) * RP + tx + 4 + (lane >> 4) * PLANE;
  }
  const int aoff = (lane >> 4) * COP + (lane & 15);

  f32x4 acc[4][MB][NBL];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBL; ++nb) acc[ph][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // second-level sums (conv.hip, GL_ACC_DUMP): an output of one parity contracts 4 taps * Cin terms
  constexpr int DUMP = GL_ACC_DUMP ? GL_ACC_DUMP_TERMS / (4 * CI_T) : 0;
  [[maybe_unused]] f32x4 acc2[DUMP ? 4 : 1][DUMP ? MB : 1][DUMP ? NBL : 1];
  [[maybe_unused]] int since_dump = 0;
  if constexpr (DUMP > 0) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBL; ++nb) acc2[ph][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- TPIPE: by half chunks (conv.hip, conv_fwd_kernel PIPE): the two 4-channel groups of the 8-channel chunk's LDS space are
  // two images; the next half chunk is stored into the one that died at the previous barrier, in pieces behind the (phase, tap)
  // combinations of the current one; loads two half chunks ahead in two register sets; one barrier per half chunk ----
#ifndef GL_S2_UP_PIPE
#define GL_S2_UP_PIPE 1
#endif
  constexpr bool TPIPE = GL_S2_UP_PIPE && GL_ACC_DUMP && GL_S2_FRAG_PREFETCH && CI_T == 8 && NBL == 2;
  if constexpr (TPIPE) {
    constexpr int HC = 4;
    constexpr int XI = HC * Cfg::R * Cfg::ROW4, WI = 16 * HC * (CO_T / 4);
    constexpr int XPH = (XI + 255) / 256, WPH = (WI + 255) / 256;
    static_assert(!TPIPE || 2 * (XPH + WPH) <= 16, "staging pieces of a half chunk");
    constexpr int NOITEM = (int)0x80000000;
    int hxg[XPH], hxl[XPH];          // byte offset inside the image (or NOITEM) ; LDS offset inside the half | ci << 20 (or -1)
#pragma unroll
    for (int i = 0; i < XPH; ++i) {
      const int e = tid + i * 256;
      const int q = e % Cfg::ROW4, t = e / Cfg::ROW4;
      const int r = t % Cfg::R, ci = t / Cfg::R;
      const int gy_ = oy0 - 1 + r, gx_ = ox0 - 4 + 4 * q;
      hxl[i] = e < XI ? ((ci * PLANE + r * RP + 4 * q) | (ci << 20)) : -1;
      hxg[i] = (e < XI && (unsigned)gy_ < (unsigned)p.Hl && (unsigned)gx_ < (unsigned)p.Wl) ? (ci * plane + gy_ * p.Wl + gx_) * 4 : NOITEM;
    }
    int hwg[WPH];
#pragma unroll
    for (int i = 0; i < WPH; ++i) {
      const int e = tid + i * 256;
      const int c4 = e % (CO_T / 4), t = e / (CO_T / 4);
      const int ci = t % HC, tap = t / HC;
      hwg[i] = e < WI ? ((tap * p.Cin_p + ci) * p.Cout_p + co0 + 4 * c4) * 4 : NOITEM;
    }
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.Cin * plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wp), 0, (unsigned)(16 * p.Cin_p * p.Cout_p * 4), 0x00020000);
    const int nh = p.Cin_p / HC;               // even
    float4 xr2[2][XPH], wr2[2][WPH];
    auto load_x2 = [&](int h, int set, int i) {
      const int c0 = h * HC, ci = (hxl[i] >> 20) & 0x3ff;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (h < nh && c0 + ci < p.Cin) ? hxg[i] : NOITEM, c0 * plane * 4, 0);
      xr2[set][i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto load_w2 = [&](int h, int set, int i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, h < nh ? hwg[i] : NOITEM, h * HC * p.Cout_p * 4, 0);
      wr2[set][i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto store_x2 = [&](int h, int set, int i) {        // half chunk h -> channels 4 (h & 1) .. of the chunk's planes
      float4 v = xr2[set][i];
      if constexpr (AFF) {
        const int c = h * HC + ((hxl[i] >> 20) & 0x3ff);
        const bool ok = hxl[i] != -1 && hxg[i] != NOITEM && h < nh && c < p.Cin;
        const float sv = ok ? afftab[ok ? c : 0] : 0.f, tv = ok ? afftab[ok ? S2_AFF_MAXC + c : 0] : 0.f;
        v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
      }
      if (hxl[i] != -1) *reinterpret_cast<float4*>(Xs + (h & 1) * HC * PLANE + (hxl[i] & 0xfffff)) = v;
    };
    auto store_w2 = [&](int h, int set, int i) {        // rows (tap, 4 (h & 1) + ci) of the chunk's [tap][8][COP] slab
      const int e = tid + i * 256;
      const int c4 = e % (CO_T / 4), t = e / (CO_T / 4);
      if (e < WI) *reinterpret_cast<float4*>(Ws + ((t / HC) * CI_T + (h & 1) * HC + (t % HC)) * COP + 4 * c4) = wr2[set][i];
    };
    auto load_half = [&](int h, int set) {
#pragma unroll
      for (int i = 0; i < XPH; ++i) load_x2(h, set, i);
#pragma unroll
      for (int i = 0; i < WPH; ++i) load_w2(h, set, i);
    };
    load_half(0, 0);
    load_half(1, 1);
    if constexpr (AFF) {
      for (int c = tid; c < p.Cin; c += 256) {
        afftab[c] = p.aff_s[(long long)n * p.Cin + c];
        afftab[S2_AFF_MAXC + c] = p.aff_t[(long long)n * p.Cin + c];
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < XPH; ++i) store_x2(0, 0, i);
#pragma unroll
    for (int i = 0; i < WPH; ++i) store_w2(0, 0, i);
    load_half(2, 0);
    __syncthreads();
    constexpr int DUMPH = 2 * DUMP;
    auto half_chunk = [&](int h, int par) {
      constexpr int PD = 2;
      float av[PD + 1][MB], bv[PD + 1][NBL];
      auto fetch = [&](int i, int st) {          // (phase, tap) combination i of the 4-channel group `par`
        const int py = (i >> 3) & 1, px = (i >> 2) & 1, iy = (i >> 1) & 1, ix = i & 1;
        const int dy = py == 0 ? (iy == 0 ? -1 : 0) : (iy == 0 ? 0 : 1);
        const int a = py == 0 ? (iy == 0 ? 3 : 1) : (iy == 0 ? 2 : 0);
        const int dx = px == 0 ? (ix == 0 ? -1 : 0) : (ix == 0 ? 0 : 1);
        const int b = px == 0 ? (ix == 0 ? 3 : 1) : (ix == 0 ? 2 : 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[st][mb] = Ws[par * 4 * COP + aoff + ((a * 4 + b) * CI_T) * COP + mb * 16];
#pragma unroll
        for (int nb = 0; nb < NBL; ++nb) bv[st][nb] = Xs[par * 4 * PLANE + boff[nb] + dy * RP + dx];
      };
      auto piece = [&](int k) {
        if (k < XPH) store_x2(h + 1, par ^ 1, k);
        else if (k < XPH + WPH) store_w2(h + 1, par ^ 1, k - XPH);
        else if (k < 2 * XPH + WPH) load_x2(h + 3, par ^ 1, k - XPH - WPH);
        else if (k < 2 * (XPH + WPH)) load_w2(h + 3, par ^ 1, k - 2 * XPH - WPH);
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) fetch(i, i % (PD + 1));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + PD < 16) fetch(i + PD, (i + PD) % (PD + 1));
        const int st = i % (PD + 1), ph = (i >> 2) & 3;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NBL; ++nb)
            acc[ph][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][mb], bv[st][nb], acc[ph][mb][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        piece(i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (++since_dump == DUMPH && h + 1 < nh) {
        since_dump = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBL; ++nb) {
              acc2[ph][mb][nb] += acc[ph][mb][nb];
              acc[ph][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
      }
      __syncthreads();
    };
    for (int h = 0; h < nh; h += 2) {
      half_chunk(h, 0);
      half_chunk(h + 1, 1);
    }
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBL; ++nb) acc[ph][mb][nb] += acc2[ph][mb][nb];
  } else {
  float4 xr[XPT], wr[WPT];
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wp), 0, (unsigned)(16 * p.Cin_p * p.Cout_p * 4), 0x00020000);
  auto load = [&](int ci0) {
    // buffer descriptors (this image's Cin planes; the packed weights): items outside the image / past Cin get the
    // offset 0x80000000 and read as zeros from the bounds check - straight-line code instead of exec-mask branches
    const int xs = ci0 * plane * 4, wsoff = ci0 * p.Cout_p * 4;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int ci = (xl[i] >> 20) & 0x3ff;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
          rs_x, (xg[i] >= 0 && ci0 + ci < p.Cin) ? xg[i] * 4 : (int)0x80000000, xs, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wg[i] >= 0 ? wg[i] * 4 : (int)0x80000000, wsoff, 0);
      wr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  load(0);
  if constexpr (AFF) {      // (visible to every wave after the first barrier of the chunk loop)
    for (int c = tid; c < p.Cin; c += 256) {
      afftab[c] = p.aff_s[(long long)n * p.Cin + c];
      afftab[S2_AFF_MAXC + c] = p.aff_t[(long long)n * p.Cin + c];
    }
  }
  for (int ci0 = 0; ci0 < p.Cin_p; ci0 += CI_T) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XPT; ++i)
      if (xl[i] != -1) {
        float4 v = xr[i];
        if constexpr (AFF) {   // elements inside the image only (the loads returned zeros for the padding: it stays zero)
          const int ci = (xl[i] >> 20) & 0x3ff;
          const bool ok = xg[i] >= 0 && ci0 + ci < p.Cin;
          const float sv = ok ? afftab[ci0 + ci] : 0.f, tv = ok ? afftab[S2_AFF_MAXC + ci0 + ci] : 0.f;
          v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
        }
        *reinterpret_cast<float4*>(Xs + (xl[i] & 0xfffff)) = v;
      }
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      if (tid + i * 256 < Cfg::NWI) *reinterpret_cast<float4*>(Ws + wl[i]) = wr[i];
    __syncthreads();
    const bool has_next = ci0 + CI_T < p.Cin_p;
    if constexpr (GL_S2_FRAG_PREFETCH) {
      // the operand fragments of (K-step, phase, tap) combination i + 2 are requested before the MFMAs of combination i
      // (conv.hip, mfma_chunk); the next chunk's loads go out behind the first K-step
      constexpr int NC = 16 * (CI_T / 4), PD = 2;
      float av[PD + 1][MB], bv[PD + 1][NBL];
      auto fetch = [&](int i, int st) {
        const int c4 = i >> 4, py = (i >> 3) & 1, px = (i >> 2) & 1, iy = (i >> 1) & 1, ix = i & 1;
        const int dy = py == 0 ? (iy == 0 ? -1 : 0) : (iy == 0 ? 0 : 1);
        const int a = py == 0 ? (iy == 0 ? 3 : 1) : (iy == 0 ? 2 : 0);
        const int dx = px == 0 ? (ix == 0 ? -1 : 0) : (ix == 0 ? 0 : 1);
        const int b = px == 0 ? (ix == 0 ? 3 : 1) : (ix == 0 ? 2 : 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[st][mb] = Ws[c4 * 4 * COP + aoff + ((a * 4 + b) * CI_T) * COP + mb * 16];
#pragma unroll
        for (int nb = 0; nb < NBL; ++nb) bv[st][nb] = Xs[c4 * 4 * PLANE + boff[nb] + dy * RP + dx];
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) fetch(i, i % (PD + 1));
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        if (i + PD < NC) fetch(i + PD, (i + PD) % (PD + 1));
        if (i == 16 && has_next) load(ci0 + CI_T);
        const int st = i % (PD + 1), ph = (i >> 2) & 3;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NBL; ++nb)
            acc[ph][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][mb], bv[st][nb], acc[ph][mb][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // K-steps of 4 channels are a real loop (bounds register pressure); the 16 (phase, tap) combinations
      // inside are unrolled with immediate LDS offsets and static accumulator indices
#pragma unroll 1
      for (int c4 = 0; c4 < CI_T / 4; ++c4) {
        // the next chunk's loads go out after the first K-step: nothing but LDS reads between the barrier and the
        // first MFMA, and the address arithmetic runs under queued matrix work (2 workgroups per CU either way)
        if (c4 == 1 && has_next) load(ci0 + CI_T);
        const float* wc = Ws + c4 * 4 * COP + aoff;
        const float* xc = Xs + c4 * 4 * PLANE;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
#pragma unroll
          for (int px = 0; px < 2; ++px) {
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
              // taps(0) = {(-1,3), (0,1)} ; taps(1) = {(0,2), (+1,0)}
              const int dy = py == 0 ? (iy == 0 ? -1 : 0) : (iy == 0 ? 0 : 1);
              const int a = py == 0 ? (iy == 0 ? 3 : 1) : (iy == 0 ? 2 : 0);
#pragma unroll
              for (int ix = 0; ix < 2; ++ix) {
                const int dx = px == 0 ? (ix == 0 ? -1 : 0) : (ix == 0 ? 0 : 1);
                const int b = px == 0 ? (ix == 0 ? 3 : 1) : (ix == 0 ? 2 : 0);
                float av[MB], bv[NBL];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[mb] = wc[((a * 4 + b) * CI_T) * COP + mb * 16];
#pragma unroll
                for (int nb = 0; nb < NBL; ++nb) bv[nb] = xc[boff[nb] + dy * RP + dx];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                  for (int nb = 0; nb < NBL; ++nb)
                    acc[py * 2 + px][mb][nb] =
                        __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[py * 2 + px][mb][nb], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    if constexpr (DUMP > 0) {
      if (++since_dump == DUMP && has_next) {
        since_dump = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBL; ++nb) {
              acc2[ph][mb][nb] += acc[ph][mb][nb];
              acc[ph][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
      }
    }
  }
  if constexpr (DUMP > 0) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBL; ++nb) acc[ph][mb][nb] += acc2[ph][mb][nb];
  }
  }      // (!TPIPE)
  // epilogue (high-res): lane holds px = 0 and px = 1 of its low-res pixel -> float2 stores
  const int Wo = 2 * p.Wl;
  const long long oplane = 4LL * p.Hl * p.Wl;
  float bvv[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + mb * 16 + (lane >> 4) * 4 + r;
      bvv[mb][r] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] * p.bias_scale : 0.f;
    }
#pragma unroll
  for (int nb = 0; nb < NBL; ++nb) {
    const int j = wn * (16 * NBL) + nb * 16 + (lane & 15);
    const int Y = oy0 + (j >> Cfg::TWL), X = ox0 + (j & (TW - 1));
    if (Y >= p.Hl || X >= p.Wl) continue;
    float* dst = p.y + ((long long)n * p.Cout + co0 + (lane >> 4) * 4) * oplane + (long long)(2 * Y) * Wo + 2 * X;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + mb * 16 + (lane >> 4) * 4 + r;
        if (co < p.Cout) {
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            float v0 = acc[py * 2 + 0][mb][nb][r] + bvv[mb][r], v1 = acc[py * 2 + 1][mb][nb][r] + bvv[mb][r];
            if (p.act == GANLAB_ACT_LRELU) { v0 = gl_lrelu(v0, p.slope); v1 = gl_lrelu(v1, p.slope); }
            *reinterpret_cast<float2*>(dst + (long long)(mb * 16 + r) * oplane + py * Wo) = float2{v0, v1};
          }
        }
      }
  }
}

// ================================================================================================
// W : 16-tap weight gradient.  gK4[tap][cl][ch] = sum_{n,Y,X} low[n,cl,Y,X] * high_pad[n,ch,2Y+a-1,2X+b-1]
//   A = low tile [cl][px] (16 channels per workgroup), B = parity planes of the high patch
//   [4 parities][16 ch][PL]; wave w owns the tap row a = w (4 taps x NBA accumulators = 32 registers) and walks ALL
//   of the tile's 4-pixel K-steps, so the kernel needs ~130 VGPRs and 40 KB of LDS: 3-4 workgroups per CU cover each
//   other's staging and barriers (with the K-steps split over the waves instead - 128 accumulator registers, 2
//   workgroups per CU - the kernel ran at 55% of the MFMA peak), and a workgroup dumps ONE slot instead of four.
//   Each workgroup walks tiles split, split+S, ... and dumps to its slot; reduce_s2_kernel sums the
//   slots and folds K4 back to 3x3:  gw[ky][kx] = scale * sum_{a,b} M[a][ky] M[b][kx] gK4[a][b].
// ================================================================================================
struct W2Args {
  const float* low;
  const float* high;
  float* part;  // [slots][16][Cl][Ch]
  int N, Cl, Ch, Hl, Wl;
  int tiles_x, tiles_y, tiles_cl, tiles_ch, S;
};

template <int NBA_>
struct WCfg {
  static constexpr int NBA = NBA_, CL_T = 16 * NBA_;
  static constexpr int TW = 16, TH = 4, TWL = 4, PX_T = 64;
  static constexpr int RPL = TW + 4, RL = TH + 2;
  // plane pitch = 26 (mod 32): 16 channels land on 16 distinct even banks, the next pixel on the odd ones, and
  // the workgroup stays under 40 KB of LDS (registers, 159 VGPRs, allow 3 per CU; forcing 128 VGPRs spills and is slower)
  static constexpr int PL = pad_mod32(RL * RPL, 26);
  static constexpr int HROWS = 2 * RL, HROW4 = (2 * TW + 8) / 4;
  static constexpr int GP = pad_mod32(PX_T, 2);
  static constexpr int GS = CL_T * GP, XS = 4 * 16 * PL;
  static constexpr int NXI = 16 * HROWS * HROW4, XPT = ceil_div_c(NXI, 256);
  static constexpr int NGI = CL_T * PX_T / 4, GPT = ceil_div_c(NGI, 256);
};

template <class Cfg>
__global__ __launch_bounds__(256, 3) void conv_s2_wgrad_kernel(W2Args p) {
  constexpr int NBA = Cfg::NBA;
  constexpr int PL = Cfg::PL, RPL = Cfg::RPL, GP = Cfg::GP, TW = Cfg::TW, TH = Cfg::TH, XPT = Cfg::XPT,
                GPT = Cfg::GPT, PX_T = Cfg::PX_T;
  __shared__ __attribute__((aligned(16))) float smem[Cfg::GS + Cfg::XS];
  float* Gs = smem;
  float* Xs = smem + Cfg::GS;
  const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
  int bid = blockIdx.x;
  const int split = bid % p.S;
  bid /= p.S;
  const int ch_t = bid % p.tiles_ch;
  const int cl_t = bid / p.tiles_ch;
  const int cl0 = cl_t * Cfg::CL_T, ch0 = ch_t * 16;
  const int H = 2 * p.Hl, W = 2 * p.Wl, hplane = H * W, lplane = p.Hl * p.Wl;

  f32x4 acc[4][NBA];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < NBA; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this wave's tap row a = wk: parity plane and row offset of the high-resolution operand
  const int a_py = (wk == 0 || wk == 2) ? 1 : 0, a_dy = wk == 0 ? -1 : (wk == 3 ? 1 : 0);
  const int tap_base = (a_py * 2 * 16) * PL + a_dy * RPL;

  // Tile-independent parts of the staging descriptors; loads go through buffer descriptors over this image's planes (no 64-bit
  // addresses, no exec-mask branches) and the tile coordinates advance by S's own digits with carries: the kernel is bound by
  // what a wave issues per tile next to its MFMAs (conv.hip, conv_wgrad_kernel DESCW).
  constexpr int NOITEM = (int)0x80000000;
  int xrel[XPT], xyx[XPT], xl[XPT];   // ch*hplane + dy*W + dx (or NOITEM) ; dy | dx << 16 ; LDS offset (or -1)
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % Cfg::HROW4;
    const int t = e / Cfg::HROW4;
    const int hr = t % Cfg::HROWS, ch = t / Cfg::HROWS;
    const int ly = hr >> 1, py = hr & 1;
    xl[i] = e < Cfg::NXI ? (((py * 2) * 16 + ch) * PL + ly * RPL + 2 * q) : -1;
    const int dy = hr - 2, dx = 4 * q - 4;
    xrel[i] = (e < Cfg::NXI && ch0 + ch < p.Ch) ? (ch0 + ch) * hplane + dy * W + dx : NOITEM;
    xyx[i] = (dy & 0xffff) | (dx << 16);
  }
  int gl_[GPT], grel[GPT], gyx[GPT];
#pragma unroll
  for (int i = 0; i < GPT; ++i) {
    const int e = tid + i * 256;
    const int j = (e % (PX_T / 4)) * 4, c = e / (PX_T / 4);
    const int ty = j >> Cfg::TWL, tx = j & (TW - 1);
    gl_[i] = e < Cfg::NGI ? c * GP + j : -1;
    grel[i] = (e < Cfg::NGI && cl0 + c < p.Cl) ? (cl0 + c) * lplane + ty * p.Wl + tx : NOITEM;
    gyx[i] = ty | (tx << 16);
  }
  float4 xr[XPT], gr[GPT];
  const int tiles_img = p.tiles_x * p.tiles_y, n_tiles = p.N * tiles_img;
  auto load_tile_at = [&](int txi, int tyi, int n) {
    const int ox0 = txi * TW, oy0 = tyi * TH;
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.high + (long long)n * p.Ch * hplane), 0, (unsigned)(p.Ch * hplane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.low + (long long)n * p.Cl * lplane), 0, (unsigned)(p.Cl * lplane * 4), 0x00020000);
    const int hbase = 2 * oy0 * W + 2 * ox0, lbase = oy0 * p.Wl + ox0;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int vy = 2 * oy0 + (int)(short)(xyx[i] & 0xffff), vx = 2 * ox0 + (xyx[i] >> 16);
      const bool ok = xrel[i] != NOITEM && (unsigned)vy < (unsigned)H && (unsigned)vx < (unsigned)W;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_h, ok ? (xrel[i] + hbase) * 4 : NOITEM, 0, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const bool ok = grel[i] != NOITEM && oy0 + (gyx[i] & 0xffff) < p.Hl && ox0 + (gyx[i] >> 16) < p.Wl;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_l, ok ? (grel[i] + lbase) * 4 : NOITEM, 0, 0);
      gr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  int tile = split;
  int txi = tile % p.tiles_x, tyi = (tile / p.tiles_x) % p.tiles_y, tni = tile / tiles_img;
  const int sdx = p.S % p.tiles_x, sdy = (p.S / p.tiles_x) % p.tiles_y, sdn = p.S / tiles_img;
  if (tile < n_tiles) load_tile_at(txi, tyi, tni);
  while (tile < n_tiles) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      if (xl[i] != -1) {
        const int l = xl[i];
        *reinterpret_cast<float2*>(Xs + l) = float2{xr[i].x, xr[i].z};
        *reinterpret_cast<float2*>(Xs + l + 16 * PL) = float2{xr[i].y, xr[i].w};
      }
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      if (gl_[i] >= 0) {
        *reinterpret_cast<float2*>(Gs + gl_[i]) = float2{gr[i].x, gr[i].y};
        *reinterpret_cast<float2*>(Gs + gl_[i] + 2) = float2{gr[i].z, gr[i].w};
      }
    }
    __syncthreads();
    const int next = tile + p.S;
    {
      txi += sdx;
      int carry = txi >= p.tiles_x ? 1 : 0;
      txi -= carry * p.tiles_x;
      tyi += sdy + carry;
      carry = tyi >= p.tiles_y ? 1 : 0;
      tyi -= carry * p.tiles_y;
      tni += sdn + carry;
    }
    if (next < n_tiles) load_tile_at(txi, tyi, tni);   // (issued inside the K loop instead: same-box A/B 288.4 vs 287.6 ms/step - no gain)
#pragma unroll 4
    for (int q = 0; q < PX_T / 4; ++q) {
      const int j = 4 * q + (lane >> 4);
      const int ty = j >> Cfg::TWL, tx = j & (TW - 1);
      const int poff = (ty + 1) * RPL + tx + 2 + (lane & 15) * PL + tap_base;
      float av[NBA];
#pragma unroll
      for (int m = 0; m < NBA; ++m) av[m] = Gs[(m * 16 + (lane & 15)) * GP + j];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float bv = Xs[(cPY(b) * 16) * PL + poff + cDY(b)];
#pragma unroll
        for (int m = 0; m < NBA; ++m)
          acc[b][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv, acc[b][m], 0, 0, 0);
      }
    }
    tile = next;
  }
  // dump: taps 4 wk + b of this workgroup's slot; D rows = cl (lane>>4)*4+r, col = ch (lane&15)
  float* dst = p.part + (long long)split * 16 * p.Cl * p.Ch;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int m = 0; m < NBA; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = cl0 + m * 16 + (lane >> 4) * 4 + r, ch = ch0 + (lane & 15);
        if (cl < p.Cl && ch < p.Ch) dst[((long long)(wk * 4 + b) * p.Cl + cl) * p.Ch + ch] = acc[b][m][r];
      }
}

// stage 1: sum slots (grouped) ; stage 2 (fold = 1): K4 -> 3x3 and write OIHW.
__global__ void reduce_s2_slots_kernel(const float* __restrict__ part, float* __restrict__ out, long long n,
                                       int slots, int groups) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = blockIdx.y;
  float s0 = 0.f, s1 = 0.f;
  int k = g;
  for (; k + groups < slots; k += 2 * groups) {
    s0 += part[(long long)k * n + i];
    s1 += part[(long long)(k + groups) * n + i];
  }
  for (; k < slots; k += groups) s0 += part[(long long)k * n + i];
  out[(long long)g * n + i] = s0 + s1;
}

// gw[co][ci][ky][kx] = scale * sum_{a,b} M[a][ky] M[b][kx] * sum_g part[g][a*4+b][cl][ch]
//   low_is_co = 1: (cl, ch) = (co, ci)  (down layer) ; 0: (cl, ch) = (ci, co)  (up layer)
__global__ void fold_s2_kernel(const float* __restrict__ part, float* __restrict__ gw, int Cout, int Cin, int Cl,
                               int Ch, int groups, int up, int low_is_co, float scale) {
  const long long n = (long long)Cout * Cin;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int co = (int)(i / Cin), ci = (int)(i % Cin);
  const int cl = low_is_co ? co : ci, ch = low_is_co ? ci : co;
  float k4[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += part[((long long)g * 16 + t) * Cl * Ch + (long long)cl * Ch + ch];
    k4[t] = s;
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s += comb(up, a, ky) * comb(up, b, kx) * k4[a * 4 + b];
      gw[i * 9 + ky * 3 + kx] = s * scale;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool s2_ok(const ganlab_conv_geom* g, int* Hl, int* Wl) {
  if (!g || g->N <= 0 || g->Cin <= 0 || g->Cout <= 0 || g->ks != 3 || g->pad != 1) return false;
  if ((g->up != 0) == (g->pool != 0)) return false;  // exactly one of them
  int hl, wl;
  if (g->up) { hl = g->Hin; wl = g->Win; } else {
    if ((g->Hin & 1) || (g->Win & 1)) return false;
    hl = g->Hin / 2; wl = g->Win / 2;
  }
  // vector staging: rows of the low tensor must be float4-aligned, and the tiles are 16/32 wide
  if (hl < 4 || wl < 16 || (wl & 3)) return false;
  // (byte offsets inside one image's HIGH-resolution planes - 4 hl wl floats per channel - are 32-bit)
  if ((long long)(g->Cin > g->Cout ? g->Cin : g->Cout) * hl * wl * 16 >= 0x7fffffffLL) return false;
  *Hl = hl; *Wl = wl;
  return true;
}

template <class Cfg, class K>
int launch_s2(K kernel, S2Args a, hipStream_t st) {
  a.tiles_x = ceil_div(a.Wl, Cfg::TW);
  a.tiles_y = ceil_div(a.Hl, Cfg::TH);
  a.tiles_co = ceil_div(a.Cout, Cfg::CO_T);
  const long long grid = (long long)a.tiles_x * a.tiles_y * a.tiles_co * a.N;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  GL_LAUNCH(kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
  return GL_CHECK_LAUNCH();
}

int run_S(S2Args a, hipStream_t st) {
  // the thin top-resolution pair (16 high-resolution channels -> 32 low-resolution ones) on 32-aligned planes: the
  // rolling-window kernel (conv_s2_roll.hip); GANLAB_S2_ROLL=0 keeps the tile kernel (same-box A/B measurements)
  if (gl_s2_roll_supported(0, a.N, a.Cin, a.Cout, a.Hl, a.Wl, a.x, a.y))
    return gl_s2_roll_launch(0, a.x, a.wp, a.bias, a.y, a.N, a.Cin, a.Cout, a.Hl, a.Wl, a.Cin_p, a.Cout_p, a.bias_scale,
                             a.act, a.slope, st);
  if (a.Wl <= 16 && a.Cout > 32) {
    // 16-wide outputs: one 16x16 tile per image.  With 64-channel tiles 512 -> 512 at batch 32 is 256 workgroups - one per
    // CU, nothing to cover its stalls (0.67 of the peak); 32-channel tiles make it 512, two per CU (GL_S2_DOWN_W16_MB2=0: A/B)
#ifndef GL_S2_DOWN_W16_MB2
#define GL_S2_DOWN_W16_MB2 1
#endif
    const long long wg64 = (long long)ceil_div(a.Wl, 16) * ceil_div(a.Hl, 16) * a.N * ceil_div(a.Cout, 64);
#ifndef GL_S2_DOWN_W16      // 1: the pipelined 64-channel x (16 x 8)-pixel tile here as well (512 workgroups too): 512 -> 512 pool @32^2 x32
#define GL_S2_DOWN_W16 1    // 0.569 -> 0.521 ms (before that kernel was pipelined the same routing measured equal); 0 / 2: A/B builds
#endif
    if (GL_S2_DOWN_W16 == 1) return launch_s2<SCfg<4, 4, 2>>(conv_s2_down_kernel<SCfg<4, 4, 2>>, a, st);
    if (GL_S2_DOWN_W16 == 2) return launch_s2<SCfg<2, 4, 2>>(conv_s2_down_kernel<SCfg<2, 4, 2>>, a, st);
    if (GL_S2_DOWN_W16_MB2 && wg64 < 512) return launch_s2<SCfg<2, 4>>(conv_s2_down_kernel<SCfg<2, 4>>, a, st);
    return launch_s2<SCfg<4, 4>>(conv_s2_down_kernel<SCfg<4, 4>>, a, st);
  }
  if (a.Cout <= 16) return launch_s2<SCfg<1>>(conv_s2_down_kernel<SCfg<1>>, a, st);
#ifndef GL_S2_DOWN_MB2_CIN
#define GL_S2_DOWN_MB2_CIN 0
#endif
  if (a.Cout <= 32 || (GL_ACC_DUMP && a.Cin <= GL_S2_DOWN_MB2_CIN)) return launch_s2<SCfg<2>>(conv_s2_down_kernel<SCfg<2>>, a, st);
#ifndef GL_S2_DOWN_NB2      // the 64-channel tile over 128 low-resolution pixels (16 x 8): half the accumulators, see conv.hip ThickCfg
#define GL_S2_DOWN_NB2 1
#endif
  if (GL_S2_DOWN_NB2 && GL_ACC_DUMP) return launch_s2<SCfg<4, 4, 2>>(conv_s2_down_kernel<SCfg<4, 4, 2>>, a, st);
  return launch_s2<SCfg<4>>(conv_s2_down_kernel<SCfg<4>>, a, st);
}

#ifndef GL_S2_UP_MB1_CIN
#define GL_S2_UP_MB1_CIN 128
#endif
int run_T(S2Args a, hipStream_t st) {
  if (gl_s2_roll_supported(1, a.N, a.Cin, a.Cout, a.Hl, a.Wl, a.x, a.y))
    return gl_s2_roll_launch(1, a.x, a.wp, a.bias, a.y, a.N, a.Cin, a.Cout, a.Hl, a.Wl, a.Cin_p, a.Cout_p, a.bias_scale,
                             a.act, a.slope, st, a.aff_s, a.aff_t);
  if (a.aff_s != nullptr) {       // deferred-InstanceNorm input: the 32-channel-tile kernel only (ganlab_conv_s2_aff_supported)
    if (a.Cout <= 16 || a.Cin > S2_AFF_MAXC) return GANLAB_EUNSUPPORTED;
    if (GL_ACC_DUMP && a.Cin >= GL_S2_UP_MB1_CIN) return launch_s2<TCfg<1, 2>>(conv_s2_up_kernel<TCfg<1, 2>, true>, a, st);
    return launch_s2<TCfg<2, 2>>(conv_s2_up_kernel<TCfg<2, 2>, true>, a, st);
  }
  if (a.Cout <= 16) return launch_s2<TCfg<1, 4>>(conv_s2_up_kernel<TCfg<1, 4>>, a, st);
#if GL_S2_UP_NBL1
  return launch_s2<TCfg<2, 1>>(conv_s2_up_kernel<TCfg<2, 1>>, a, st);
#endif
  // 16 output channels per workgroup where the contraction is long (>= 128 channels): 113 registers with the second
  // accumulator set against 208 for the 32-channel tile, four workgroups per CU against two - measured in round 4 at batch 32,
  // input gradient of 128 -> 256 pool @128^2 1.035 -> 0.998 ms, 256 -> 512 @64^2 1.027 -> 0.990, forward of 256 -> 128 up @64^2
  // 1.035 -> 0.998, 512 -> 256 @32^2 1.025 -> 0.989; with 64 contracted channels the extra patch staging loses 1 %
  if (GL_ACC_DUMP && a.Cin >= GL_S2_UP_MB1_CIN) return launch_s2<TCfg<1, 2>>(conv_s2_up_kernel<TCfg<1, 2>>, a, st);
  // (short contractions, measured in round 4 on 32 -> 64 pool @512^2 input gradient / 64 -> 32 up @256^2 forward: the 16-channel
  // tile 1.131 -> 1.113..1.137 / 1.131 -> 1.121..1.137 ms = noise; 32 channels x 64 low-res pixels 1.170 / 1.169: slower)
  if (a.Cout <= 32) return launch_s2<TCfg<2, 2>>(conv_s2_up_kernel<TCfg<2, 2>>, a, st);
  // 32 output channels per workgroup for the thick layers as well: the 64-channel tile keeps 128 accumulator registers
  // (219 VGPRs, two workgroups per CU) and measured 3-4 % slower on every layer than this one (123 VGPRs, four per CU)
  return launch_s2<TCfg<2, 2>>(conv_s2_up_kernel<TCfg<2, 2>>, a, st);
}

struct W2Plan { int nba, tiles_x, tiles_y, tiles_cl, tiles_ch, S, slots; };
W2Plan plan_w2(int N, int Cl, int Ch, int Hl, int Wl) {
  W2Plan pl{};
  pl.nba = Cl > 16 ? 2 : 1;
  pl.tiles_x = ceil_div(Wl, WCfg<1>::TW); pl.tiles_y = ceil_div(Hl, WCfg<1>::TH);
  pl.tiles_cl = ceil_div(Cl, 16 * pl.nba); pl.tiles_ch = ceil_div(Ch, 16);
  const long long n_tiles = (long long)pl.tiles_x * pl.tiles_y * N, base = (long long)pl.tiles_cl * pl.tiles_ch;
  long long S = (768 + base - 1) / base;    // three workgroups per CU fit (LDS 40 KB, 150 VGPRs): one full round
  if (S > n_tiles) S = n_tiles;
  if (S < 1) S = 1;
  pl.S = (int)S;
  pl.slots = pl.S;          // one slot per workgroup (its four waves own four different tap rows)
  return pl;
}

// slot reduction + fold of the 16-tap gradient back onto the 3x3 parameter (shared by the plain and the AFF entry point)
static int s2_finish_wgrad(float* ws, int slots, long long nk, float* gw, const ganlab_conv_geom* g, int Cl, int Ch,
                           float scale, hipStream_t st) {
  float* stage2 = ws + (long long)slots * nk;
  // slots -> 32 groups -> 1 with the wide reduce kernel (16*Cl*Ch threads); the fold kernel has only Cl*Ch threads
  // (512 for the 16 -> 32 layer), so every slot it still had to add up cost it 16 serial strided reads per thread
  const float* folded_src = ws;
  int fold_groups = slots;
  if (fold_groups >= 64) {
    GL_LAUNCH(reduce_s2_slots_kernel, dim3((unsigned)((nk + 255) / 256), 32), dim3(256), 0, st, (const float*)ws,
              stage2, nk, slots, 32);
    folded_src = stage2;
    fold_groups = 32;
  }
  if (fold_groups > 2) {
    float* dst = folded_src == ws ? stage2 : ws;
    GL_LAUNCH(reduce_s2_slots_kernel, dim3((unsigned)((nk + 255) / 256), 1), dim3(256), 0, st, folded_src, dst, nk,
              fold_groups, 1);
    folded_src = dst;
    fold_groups = 1;
  }
  const long long nw = (long long)g->Cout * g->Cin;
  GL_LAUNCH(fold_s2_kernel, dim3((unsigned)((nw + 127) / 128)), dim3(128), 0, st, folded_src, gw, g->Cout, g->Cin, Cl,
            Ch, fold_groups, g->up ? 1 : 0, g->pool ? 1 : 0, scale);
  return GL_CHECK_LAUNCH();
}

}  // namespace

extern "C" {

int ganlab_conv_s2_supported(const ganlab_conv_geom* g) {
  int hl, wl;
  return s2_ok(g, &hl, &wl) ? 1 : 0;
}

long long ganlab_conv_s2_pack_f32(const float* w, float* out, int Cout, int Cin, int up, int transpose, float scale,
                                  void* stream) {
  if (Cout <= 0 || Cin <= 0) return GANLAB_EINVAL;
  const int rows = transpose ? Cout : Cin, cols = transpose ? Cin : Cout;
  const int rows_p = round_up_c(rows, 16), cols_p = round_up_c(cols, 64);
  const long long total = 16LL * rows_p * cols_p;
  if (!out) return total;
  if (!w) return GANLAB_EINVAL;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GL_LAUNCH(pack_s2_kernel, dim3((unsigned)blocks), dim3(256), 0, gl_stream(stream), w, out, Cout, Cin, rows, cols,
            rows_p, cols_p, up ? 1 : 0, transpose ? 1 : 0, scale);
  return GL_CHECK_LAUNCH() == GANLAB_OK ? total : GANLAB_ELAUNCH;
}

int ganlab_conv_s2_fwd_f32(const float* x, const float* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                           float bias_scale, int act, float slope, void* stream) {
  int hl, wl;
  if (!s2_ok(g, &hl, &wl) || !x || !wp || !y || !aligned16(x) || !aligned16(wp) || !aligned16(y)) return GANLAB_EINVAL;
  S2Args a{};
  a.x = x; a.wp = wp; a.bias = bias; a.y = y;
  a.N = g->N; a.Cin = g->Cin; a.Cout = g->Cout; a.Hl = hl; a.Wl = wl;
  a.Cin_p = round_up_c(g->Cin, 16); a.Cout_p = round_up_c(g->Cout, 64);
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  return g->pool ? run_S(a, gl_stream(stream)) : run_T(a, gl_stream(stream));
}

int ganlab_conv_s2_dgrad_f32(const float* gy, const float* wp, float* gx, const ganlab_conv_geom* g, void* stream) {
  int hl, wl;
  if (!s2_ok(g, &hl, &wl) || !gy || !wp || !gx || !aligned16(gy) || !aligned16(wp) || !aligned16(gx))
    return GANLAB_EINVAL;
  S2Args a{};
  a.x = gy; a.wp = wp; a.bias = nullptr; a.y = gx;
  a.N = g->N; a.Cin = g->Cout; a.Cout = g->Cin; a.Hl = hl; a.Wl = wl;   // roles swap: the operator consumes gy
  a.Cin_p = round_up_c(g->Cout, 16); a.Cout_p = round_up_c(g->Cin, 64);
  a.bias_scale = 0.f; a.slope = 0.f; a.act = GANLAB_ACT_NONE;
  // down layer: gy is low-res -> T ; up layer: gy is high-res -> S
  return g->pool ? run_T(a, gl_stream(stream)) : run_S(a, gl_stream(stream));
}

// ---- deferred InstanceNorm on the low-resolution input of an up layer (S2Args::aff_s) --------------------------------
int ganlab_conv_s2_aff_supported(const ganlab_conv_geom* g) {
  int hl, wl;
  if (!s2_ok(g, &hl, &wl) || !g->up) return 0;
  int bits = 0;
  if (gl_s2_roll_supported(1, g->N, g->Cin, g->Cout, hl, wl, nullptr, nullptr) || (g->Cout > 16 && g->Cin <= S2_AFF_MAXC))
    bits |= 1;
  const char* roll_env = GL_ENV_ONCE("GANLAB_WGRAD_ROLL");
  if (!(roll_env && roll_env[0] == '0') && gl_wgrad_s2_roll_supported(g->N, g->Cin, g->Cout, hl, wl, nullptr, nullptr))
    bits |= 2;
  return bits;
}

int ganlab_conv_s2_fwd_aff_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t, const float* bias,
                               float* y, const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream) {
  int hl, wl;
  if (!(ganlab_conv_s2_aff_supported(g) & 1) || !s2_ok(g, &hl, &wl)) return GANLAB_EUNSUPPORTED;
  if (!x || !wp || !y || !aff_s || !aff_t || !aligned16(x) || !aligned16(wp) || !aligned16(y)) return GANLAB_EINVAL;
  S2Args a{};
  a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.aff_s = aff_s; a.aff_t = aff_t;
  a.N = g->N; a.Cin = g->Cin; a.Cout = g->Cout; a.Hl = hl; a.Wl = wl;
  a.Cin_p = round_up_c(g->Cin, 16); a.Cout_p = round_up_c(g->Cout, 64);
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  return run_T(a, gl_stream(stream));
}

int ganlab_conv_s2_wgrad_aff_f32(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                                 const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  int hl, wl;
  if (!(ganlab_conv_s2_aff_supported(g) & 2) || !s2_ok(g, &hl, &wl)) return GANLAB_EUNSUPPORTED;
  if (!gy || !x || !gw || !aff_s || !aff_t || !aligned16(gy) || !aligned16(x)) return GANLAB_EINVAL;
  const int Cl = g->Cin, Ch = g->Cout;            // up layer: low = x (deferred), high = gy
  const long long nk = 16LL * Cl * Ch;
  const int slots = gl_wgrad_s2_roll_slots(g->N, Cl, Ch, hl, wl);
  if (!workspace || workspace_bytes < (size_t)(slots + 32) * nk * sizeof(float)) return GANLAB_EWORKSPACE;
  hipStream_t st = gl_stream(stream);
  const int rc = gl_wgrad_s2_roll_launch(x, gy, (float*)workspace, g->N, Cl, Ch, hl, wl, st, aff_s, aff_t);
  if (rc != GANLAB_OK) return rc;
  return s2_finish_wgrad((float*)workspace, slots, nk, gw, g, Cl, Ch, scale, st);
}

size_t ganlab_conv_s2_wgrad_workspace(const ganlab_conv_geom* g) {
  int hl, wl;
  if (!s2_ok(g, &hl, &wl)) return 0;
  const int Cl = g->pool ? g->Cout : g->Cin, Ch = g->pool ? g->Cin : g->Cout;
  const W2Plan pl = plan_w2(g->N, Cl, Ch, hl, wl);
  return (size_t)(pl.slots + 32) * 16 * Cl * Ch * sizeof(float);
}

int ganlab_conv_s2_wgrad_f32(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g, float scale,
                             void* workspace, size_t workspace_bytes, void* stream) {
  int hl, wl;
  if (!s2_ok(g, &hl, &wl) || !gy || !x || !gw || !aligned16(gy) || !aligned16(x)) return GANLAB_EINVAL;
  const int Cl = g->pool ? g->Cout : g->Cin, Ch = g->pool ? g->Cin : g->Cout;
  const W2Plan pl = plan_w2(g->N, Cl, Ch, hl, wl);
  const long long nk = 16LL * Cl * Ch;
  if (!workspace || workspace_bytes < (size_t)(pl.slots + 32) * nk * sizeof(float)) return GANLAB_EWORKSPACE;
  W2Args a{};
  a.low = g->pool ? gy : x; a.high = g->pool ? x : gy; a.part = (float*)workspace;
  a.N = g->N; a.Cl = Cl; a.Ch = Ch; a.Hl = hl; a.Wl = wl;
  a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.tiles_cl = pl.tiles_cl; a.tiles_ch = pl.tiles_ch; a.S = pl.S;
  hipStream_t st = gl_stream(stream);
  int slots = pl.slots;
  // planes whose low-resolution width is a multiple of 32: the rolling-window kernel (wgrad_roll.hip), same slot
  // layout; GANLAB_WGRAD_ROLL=0 keeps the tile kernel (same-box A/B measurements)
  const char* roll_env = GL_ENV_ONCE("GANLAB_WGRAD_ROLL");
  bool rolled = false;
  if (!(roll_env && roll_env[0] == '0') && gl_wgrad_s2_roll_supported(g->N, Cl, Ch, hl, wl, a.low, a.high)) {
    const int rs = gl_wgrad_s2_roll_slots(g->N, Cl, Ch, hl, wl);
    if (workspace_bytes >= (size_t)(rs + 32) * nk * sizeof(float)) {
      const int rc = gl_wgrad_s2_roll_launch(a.low, a.high, a.part, g->N, Cl, Ch, hl, wl, st);
      if (rc != GANLAB_OK) return rc;
      slots = rs;
      rolled = true;
    }
  }
  if (!rolled) {
    const unsigned wgrid = (unsigned)((long long)pl.tiles_cl * pl.tiles_ch * pl.S);
    if (pl.nba == 2) GL_LAUNCH(conv_s2_wgrad_kernel<WCfg<2>>, dim3(wgrid), dim3(256), 0, st, a);
    else GL_LAUNCH(conv_s2_wgrad_kernel<WCfg<1>>, dim3(wgrid), dim3(256), 0, st, a);
  }
  return s2_finish_wgrad((float*)workspace, slots, nk, gw, g, Cl, Ch, scale, st);
}

}  // extern "C"
