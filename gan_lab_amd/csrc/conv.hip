// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32), NCHW.
//
// One kernel family serves Conv2dEx / LinearEx forward, their input gradient (same kernel, flipped
// + transposed packed weights) and their weight gradient (separate kernel, split over pixels):
// reference math = F.conv2d(x * wscale, W, padding) at utils/custom_layers.py:202-211 and
// F.linear at :282-291, plus the nearest 2x upsample in front of generator convs
// (stylegan/architectures.py:292-334) which is folded into the tile staging (never materialised).
//
// GEMM view (forward):  D[co][p] = sum_{tap,ci} Wp[tap][ci][co] * Xv[n(p), ci, oy(p)+ky-pad, ox(p)+kx-pad]
//   MFMA 16x16x4 f32:  A[i=co][k=ci]  (lane l holds A[l&15][l>>4])
//                      B[k=ci][j=px]  (lane l holds B[l>>4][l&15])
//                      D[i][j]        (lane l holds rows (l>>4)*4+r, r=0..3, column l&15)
//   so every lane ends up with 4 consecutive output channels of ONE pixel and the 16 lanes of a
//   quarter-wave cover 16 consecutive pixels -> 64-byte coalesced NCHW stores.
// A workgroup (256 threads = 4 waves, one per SIMD) owns a tile of CO_T channels x PX_T pixels,
// where the pixels are an NI x TH x TW patch; per K-chunk of CI_T input channels the (halo'd)
// activation patch and the [tap][ci][co] weight slab are staged in LDS, then each wave runs
// KK * CI_T/4 * MB * NB MFMAs straight out of LDS (one ds_read_b32 per operand fragment).
// LDS strides are padded so the two ci-halves of a 32-lane read group land on disjoint banks.
#include "common.h"

namespace {

constexpr int round_up_c(int v, int m) { return (v + m - 1) / m * m; }
// smallest s >= v with s % 32 == r
constexpr int pad_mod32(int v, int r) { return v + ((r - (v % 32)) + 32) % 32; }

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int N, Cin, Hi, Wi;  // physical input
  int Hv, Wv;          // virtual input (after optional up2)
  int Cout, Ho, Wo;
  int pad, up;
  int Cin_p, Cout_p;   // packed-weight dims: wp[tap][Cin_p][Cout_p]
  int tiles_x, tiles_y, tiles_n, tiles_co;
  float bias_scale, slope;
  int act;
};

template <int KS_, int MB_, int TWL_, int THL_, int NIL_>
struct FwdCfg {
  static constexpr int KS = KS_, KK = KS_ * KS_, MB = MB_;
  static constexpr int WN = 4;  // 4 waves side by side along the pixel dim
  static constexpr int TWL = TWL_, THL = THL_, NIL = NIL_;
  static constexpr int TW = 1 << TWL_, TH = 1 << THL_, NI = 1 << NIL_;
  static constexpr int PX_T = TW * TH * NI;
  static constexpr int NB = PX_T / (16 * WN);
  static constexpr int CO_T = 16 * MB_;
  static constexpr int CI_T = (KS_ == 1) ? 32 : 8;
  static constexpr int R = TH + KS_ - 1, C = TW + KS_ - 1;
  static constexpr int IMG = R * C;
  static constexpr int PLANE = pad_mod32(NI * IMG, 16);
  static constexpr int COP = pad_mod32(CO_T, 16);
  static constexpr int XS = CI_T * PLANE, WS = KK * CI_T * COP;
  static_assert(NB >= 1, "pixel tile too small");
};

template <class Cfg>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs p) {
  constexpr int KS = Cfg::KS, KK = Cfg::KK, MB = Cfg::MB, NB = Cfg::NB, CI_T = Cfg::CI_T;
  constexpr int C = Cfg::C, R = Cfg::R, IMG = Cfg::IMG, PLANE = Cfg::PLANE, COP = Cfg::COP;
  constexpr int NI = Cfg::NI, TW = Cfg::TW, TH = Cfg::TH, CO_T = Cfg::CO_T;
  __shared__ float smem[Cfg::XS + Cfg::WS];
  float* Xs = smem;
  float* Ws = smem + Cfg::XS;

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int txi = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tyi = bid % p.tiles_y;
  const int tni = bid / p.tiles_y;
  const int co0 = co_t * CO_T, ox0 = txi * TW, oy0 = tyi * TH, n0 = tni * NI;

  int boff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane & 15);
    const int ni = j >> (Cfg::TWL + Cfg::THL), ty = (j >> Cfg::TWL) & (TH - 1), tx = j & (TW - 1);
    boff[nb] = ni * IMG + ty * C + tx + (lane >> 4) * PLANE;
  }
  const int aoff = (lane >> 4) * COP + (lane & 15);

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long in_plane = (long long)p.Hi * p.Wi;
  for (int ci0 = 0; ci0 < p.Cin_p; ci0 += CI_T) {
    __syncthreads();
    // ---- stage the activation patch (zero padding, optional nearest-2x upsample) ----
    for (int e = tid; e < CI_T * NI * IMG; e += 256) {
      const int c = e % C;
      int t = e / C;
      const int r = t % R;
      t /= R;
      const int ni = t % NI;
      const int ci = t / NI;
      const int vy = oy0 + r - p.pad, vx = ox0 + c - p.pad, n = n0 + ni, cig = ci0 + ci;
      float v = 0.f;
      if (cig < p.Cin && n < p.N && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv) {
        const int iy = p.up ? (vy >> 1) : vy, ix = p.up ? (vx >> 1) : vx;
        v = p.x[((long long)n * p.Cin + cig) * in_plane + (long long)iy * p.Wi + ix];
      }
      Xs[ci * PLANE + ni * IMG + r * C + c] = v;
    }
    // ---- stage the weight slab: contiguous CO_T runs of wp[tap][ci][co] ----
    for (int e = tid; e < KK * CI_T * CO_T; e += 256) {
      const int co = e % CO_T;
      const int t = e / CO_T;
      const int ci = t % CI_T, tap = t / CI_T;
      Ws[(tap * CI_T + ci) * COP + co] =
          p.wp[((long long)tap * p.Cin_p + ci0 + ci) * p.Cout_p + co0 + co];
    }
    __syncthreads();
    // ---- MFMA ----
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      const int toff = (tap / KS) * C + (tap % KS);
#pragma unroll
      for (int c4 = 0; c4 < CI_T / 4; ++c4) {
        float a[MB], b[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = Ws[(tap * CI_T + c4 * 4) * COP + aoff + mb * 16];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = Xs[c4 * 4 * PLANE + boff[nb] + toff];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: + bias, activation, NCHW store ----
  const long long out_plane = (long long)p.Ho * p.Wo;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = wn * (16 * NB) + nb * 16 + (lane & 15);
    const int ni = j >> (Cfg::TWL + Cfg::THL), ty = (j >> Cfg::TWL) & (TH - 1), tx = j & (TW - 1);
    const int n = n0 + ni, oy = oy0 + ty, ox = ox0 + tx;
    if (n >= p.N || oy >= p.Ho || ox >= p.Wo) continue;
    const long long pix = (long long)oy * p.Wo + ox;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + mb * 16 + (lane >> 4) * 4 + r;
        if (co < p.Cout) {
          float v = acc[mb][nb][r];
          if (p.bias) v += p.bias[co] * p.bias_scale;
          if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
          p.y[((long long)n * p.Cout + co) * out_plane + pix] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight re-layout: OIHW -> [tap][rows_p][cols_p] (rows = GEMM-K channel, cols = GEMM-M channel)
// ------------------------------------------------------------------------------------------------
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                            int KK, int rows, int cols, int rows_p, int cols_p, int dgrad, float scale) {
  const long long total = (long long)KK * rows_p * cols_p;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e % cols_p);
    const long long t = e / cols_p;
    const int row = (int)(t % rows_p), tap = (int)(t / rows_p);
    float v = 0.f;
    if (row < rows && col < cols) {
      // fwd:   row = ci, col = co, src tap = tap
      // dgrad: row = co, col = ci, src tap = KK-1-tap
      const int co = dgrad ? row : col, ci = dgrad ? col : row, st = dgrad ? (KK - 1 - tap) : tap;
      v = scale * w[((long long)co * Cin + ci) * KK + st];
    }
    out[e] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: D[co][ci] (per tap) = sum_px gy[co][px] * Xv[ci][px shifted by tap]
//   MFMA 16x16x4:  A[i=co][k=px] from a [co][px] LDS tile of gy, B[k=px][j=ci] from the halo'd patch.
//   Waves are laid out WM (over co blocks) x WK (over the tile's 4-pixel K-steps); each workgroup
//   walks pixel tiles `split, split+S, ...` and finally dumps its accumulators to a private slot of
//   the workspace; wgrad_reduce_kernel sums the slots in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* gy;
  const float* x;
  float* part;  // [slots][Cout][Cin][KK]
  int N, Cin, Hi, Wi, Hv, Wv, Cout, Ho, Wo, pad, up;
  int tiles_x, tiles_y, tiles_n, tiles_co, tiles_ci, S;
};

template <int KS_, int WM_, int WK_, int TWL_, int THL_, int NIL_>
struct WgCfg {
  static constexpr int KS = KS_, KK = KS_ * KS_, WM = WM_, WK = WK_;
  static constexpr int NBC = 2;  // 32 input channels per workgroup
  static constexpr int TWL = TWL_, THL = THL_, NIL = NIL_;
  static constexpr int TW = 1 << TWL_, TH = 1 << THL_, NI = 1 << NIL_;
  static constexpr int PX_T = TW * TH * NI;
  static constexpr int CO_T = 16 * WM_, CI_T = 16 * NBC;
  static constexpr int R = TH + KS_ - 1, C = TW + KS_ - 1, IMG = R * C;
  static constexpr int PLANE = pad_mod32(NI * IMG, 2);
  static constexpr int GP = pad_mod32(PX_T, 2);
  static constexpr int GS = CO_T * GP, XS = CI_T * PLANE;
  static_assert(WM_ * WK_ == 4, "4 waves");
  static_assert(PX_T % (4 * WK_) == 0, "K-steps must split evenly over waves");
};

template <class Cfg>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int KS = Cfg::KS, KK = Cfg::KK, NBC = Cfg::NBC, WK = Cfg::WK;
  constexpr int C = Cfg::C, R = Cfg::R, IMG = Cfg::IMG, PLANE = Cfg::PLANE, GP = Cfg::GP;
  constexpr int NI = Cfg::NI, TW = Cfg::TW, TH = Cfg::TH, CO_T = Cfg::CO_T, CI_T = Cfg::CI_T;
  constexpr int PX_T = Cfg::PX_T;
  __shared__ float smem[Cfg::GS + Cfg::XS];
  float* Gs = smem;
  float* Xs = smem + Cfg::GS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WK, wk = wave % WK;
  int bid = blockIdx.x;
  const int split = bid % p.S;
  bid /= p.S;
  const int ci_t = bid % p.tiles_ci;
  const int co_t = bid / p.tiles_ci;
  const int co0 = co_t * CO_T, ci0 = ci_t * CI_T;

  f32x4 acc[KK][NBC];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb) acc[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long in_plane = (long long)p.Hi * p.Wi, out_plane = (long long)p.Ho * p.Wo;
  const int n_tiles = p.tiles_n * p.tiles_y * p.tiles_x;
  for (int tile = split; tile < n_tiles; tile += p.S) {
    const int txi = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int tyi = t2 % p.tiles_y, tni = t2 / p.tiles_y;
    const int ox0 = txi * TW, oy0 = tyi * TH, n0 = tni * NI;
    __syncthreads();
    for (int e = tid; e < CO_T * PX_T; e += 256) {
      const int j = e % PX_T, co = e / PX_T;
      const int ni = j >> (Cfg::TWL + Cfg::THL), ty = (j >> Cfg::TWL) & (TH - 1), tx = j & (TW - 1);
      const int n = n0 + ni, oy = oy0 + ty, ox = ox0 + tx, cog = co0 + co;
      float v = 0.f;
      if (cog < p.Cout && n < p.N && oy < p.Ho && ox < p.Wo)
        v = p.gy[((long long)n * p.Cout + cog) * out_plane + (long long)oy * p.Wo + ox];
      Gs[co * GP + j] = v;
    }
    for (int e = tid; e < CI_T * NI * IMG; e += 256) {
      const int c = e % C;
      int t = e / C;
      const int r = t % R;
      t /= R;
      const int ni = t % NI;
      const int ci = t / NI;
      const int vy = oy0 + r - p.pad, vx = ox0 + c - p.pad, n = n0 + ni, cig = ci0 + ci;
      float v = 0.f;
      if (cig < p.Cin && n < p.N && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv) {
        const int iy = p.up ? (vy >> 1) : vy, ix = p.up ? (vx >> 1) : vx;
        v = p.x[((long long)n * p.Cin + cig) * in_plane + (long long)iy * p.Wi + ix];
      }
      Xs[ci * PLANE + ni * IMG + r * C + c] = v;
    }
    __syncthreads();
    for (int q = wk; q < PX_T / 4; q += WK) {
      const int j = 4 * q + (lane >> 4);
      const int ni = j >> (Cfg::TWL + Cfg::THL), ty = (j >> Cfg::TWL) & (TH - 1), tx = j & (TW - 1);
      const int poff = ni * IMG + ty * C + tx + (lane & 15) * PLANE;
      const float a = Gs[(wm * 16 + (lane & 15)) * GP + j];
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) {
        const int toff = (tap / KS) * C + (tap % KS);
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb) {
          const float b = Xs[nb * 16 * PLANE + poff + toff];
          acc[tap][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tap][nb], 0, 0, 0);
        }
      }
    }
  }
  // dump: slot = split*WK + wk ; D rows = co (lane>>4)*4+r, col = ci (lane&15)
  const int slot = split * WK + wk;
  float* dst = p.part + (long long)slot * p.Cout * p.Cin * KK;
#pragma unroll
  for (int tap = 0; tap < KK; ++tap)
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * 16 + (lane >> 4) * 4 + r;
        const int ci = ci0 + nb * 16 + (lane & 15);
        if (co < p.Cout && ci < p.Cin) dst[((long long)co * p.Cin + ci) * KK + tap] = acc[tap][nb][r];
      }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, long long n,
                                    int slots, float scale) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < slots; ++k) s += part[(long long)k * n + i];
  gw[i] = s * scale;
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
inline int cin_pad(int ks) { return ks == 1 ? 32 : 8; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <class Cfg>
int launch_fwd(ConvArgs a, hipStream_t st) {
  a.tiles_x = ceil_div(a.Wo, Cfg::TW);
  a.tiles_y = ceil_div(a.Ho, Cfg::TH);
  a.tiles_n = ceil_div(a.N, Cfg::NI);
  a.tiles_co = ceil_div(a.Cout, Cfg::CO_T);
  const long long grid = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.tiles_co;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  GL_LAUNCH(conv_fwd_kernel<Cfg>, dim3((unsigned)grid), dim3(256), 0, st, a);
  return GL_CHECK_LAUNCH();
}

template <int KS, int MB>
int dispatch_geom(const ConvArgs& a, hipStream_t st) {
  if (a.Ho == 1 && a.Wo == 1) return launch_fwd<FwdCfg<KS, MB, 0, 0, 6>>(a, st);
  if (a.Wo >= 32) return launch_fwd<FwdCfg<KS, MB, 5, 3, 0>>(a, st);
  if (a.Wo >= 16) return launch_fwd<FwdCfg<KS, MB, 4, 4, 0>>(a, st);
  if (a.Wo >= 8) return launch_fwd<FwdCfg<KS, MB, 3, 3, 2>>(a, st);
  return launch_fwd<FwdCfg<KS, MB, 2, 2, 4>>(a, st);
}

template <int KS>
int dispatch_co(const ConvArgs& a, hipStream_t st) {
  if (a.Cout <= 16) return dispatch_geom<KS, 1>(a, st);
  if (a.Cout <= 32) return dispatch_geom<KS, 2>(a, st);
  return dispatch_geom<KS, 4>(a, st);
}

int run_conv(const float* x, const float* wp, const float* bias, float* y, int N, int Cin, int Hi, int Wi,
             int Cout, int ks, int pad, int up, float bias_scale, int act, float slope, hipStream_t st) {
  if (!x || !wp || !y || N <= 0 || Cin <= 0 || Cout <= 0 || Hi <= 0 || Wi <= 0) return GANLAB_EINVAL;
  ConvArgs a{};
  a.x = x; a.wp = wp; a.bias = bias; a.y = y;
  a.N = N; a.Cin = Cin; a.Hi = Hi; a.Wi = Wi;
  a.Hv = up ? 2 * Hi : Hi; a.Wv = up ? 2 * Wi : Wi;
  a.Cout = Cout; a.pad = pad; a.up = up;
  a.Ho = a.Hv + 2 * pad - ks + 1; a.Wo = a.Wv + 2 * pad - ks + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return GANLAB_EINVAL;
  a.Cin_p = round_up_c(Cin, cin_pad(ks)); a.Cout_p = round_up_c(Cout, 64);
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  if (ks == 1) return dispatch_co<1>(a, st);
  if (ks == 3) return dispatch_co<3>(a, st);
  return GANLAB_EINVAL;
}

struct WgPlan { int thin; int tiles_x, tiles_y, tiles_n, tiles_co, tiles_ci, S, slots; };

template <class Cfg>
WgPlan plan_wgrad_cfg(int N, int Cin, int Cout, int Ho, int Wo, int thin) {
  WgPlan pl{};
  pl.thin = thin;
  pl.tiles_x = ceil_div(Wo, Cfg::TW); pl.tiles_y = ceil_div(Ho, Cfg::TH); pl.tiles_n = ceil_div(N, Cfg::NI);
  pl.tiles_co = ceil_div(Cout, Cfg::CO_T); pl.tiles_ci = ceil_div(Cin, Cfg::CI_T);
  const long long n_tiles = (long long)pl.tiles_x * pl.tiles_y * pl.tiles_n;
  const long long base = (long long)pl.tiles_co * pl.tiles_ci;
  long long S = (1024 + base - 1) / base;  // aim for ~4 workgroups per CU
  if (S > n_tiles) S = n_tiles;
  if (S < 1) S = 1;
  pl.S = (int)S;
  pl.slots = pl.S * Cfg::WK;
  return pl;
}

// thin : CO_T = 16, waves split the pixel K-steps (WM=1, WK=4), 256-pixel tiles
// thick: CO_T = 64, one co block per wave (WM=4, WK=1), 64-pixel tiles
template <int KS> using WgThinA = WgCfg<KS, 1, 4, 5, 3, 0>;   // 32x8
template <int KS> using WgThinB = WgCfg<KS, 1, 4, 4, 4, 0>;   // 16x16
template <int KS> using WgThinC = WgCfg<KS, 1, 4, 3, 3, 2>;   // 8x8 x4 images
template <int KS> using WgThinD = WgCfg<KS, 1, 4, 2, 2, 4>;   // 4x4 x16 images
template <int KS> using WgThinE = WgCfg<KS, 1, 4, 0, 0, 6>;   // 1x1 x64 samples (linear)
template <int KS> using WgThickA = WgCfg<KS, 4, 1, 3, 3, 0>;  // 8x8
template <int KS> using WgThickD = WgCfg<KS, 4, 1, 2, 2, 2>;  // 4x4 x4 images
template <int KS> using WgThickE = WgCfg<KS, 4, 1, 0, 0, 6>;  // 1x1 x64 samples (linear)

enum WgGeom { WG_A, WG_B, WG_C, WG_D, WG_E };
inline WgGeom wg_geom(int Ho, int Wo, int thin) {
  if (Ho == 1 && Wo == 1) return WG_E;
  if (!thin) return (Wo >= 8) ? WG_A : WG_D;
  if (Wo >= 32) return WG_A;
  if (Wo >= 16) return WG_B;
  if (Wo >= 8) return WG_C;
  return WG_D;
}

template <int KS>
WgPlan plan_wgrad(int N, int Cin, int Cout, int Ho, int Wo) {
  const int thin = Cout <= 32;
  switch (wg_geom(Ho, Wo, thin)) {
    case WG_A: return thin ? plan_wgrad_cfg<WgThinA<KS>>(N, Cin, Cout, Ho, Wo, 1)
                           : plan_wgrad_cfg<WgThickA<KS>>(N, Cin, Cout, Ho, Wo, 0);
    case WG_B: return plan_wgrad_cfg<WgThinB<KS>>(N, Cin, Cout, Ho, Wo, 1);
    case WG_C: return plan_wgrad_cfg<WgThinC<KS>>(N, Cin, Cout, Ho, Wo, 1);
    case WG_D: return thin ? plan_wgrad_cfg<WgThinD<KS>>(N, Cin, Cout, Ho, Wo, 1)
                           : plan_wgrad_cfg<WgThickD<KS>>(N, Cin, Cout, Ho, Wo, 0);
    default: return thin ? plan_wgrad_cfg<WgThinE<KS>>(N, Cin, Cout, Ho, Wo, 1)
                         : plan_wgrad_cfg<WgThickE<KS>>(N, Cin, Cout, Ho, Wo, 0);
  }
}

template <class Cfg>
int launch_wgrad(WgradArgs a, const WgPlan& pl, hipStream_t st) {
  a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.tiles_n = pl.tiles_n;
  a.tiles_co = pl.tiles_co; a.tiles_ci = pl.tiles_ci; a.S = pl.S;
  const long long grid = (long long)pl.tiles_co * pl.tiles_ci * pl.S;
  GL_LAUNCH(conv_wgrad_kernel<Cfg>, dim3((unsigned)grid), dim3(256), 0, st, a);
  return GL_CHECK_LAUNCH();
}

template <int KS>
int run_wgrad_ks(const WgradArgs& a, const WgPlan& pl, hipStream_t st) {
  switch (wg_geom(a.Ho, a.Wo, pl.thin)) {
    case WG_A: return pl.thin ? launch_wgrad<WgThinA<KS>>(a, pl, st) : launch_wgrad<WgThickA<KS>>(a, pl, st);
    case WG_B: return launch_wgrad<WgThinB<KS>>(a, pl, st);
    case WG_C: return launch_wgrad<WgThinC<KS>>(a, pl, st);
    case WG_D: return pl.thin ? launch_wgrad<WgThinD<KS>>(a, pl, st) : launch_wgrad<WgThickD<KS>>(a, pl, st);
    default: return pl.thin ? launch_wgrad<WgThinE<KS>>(a, pl, st) : launch_wgrad<WgThickE<KS>>(a, pl, st);
  }
}

bool geom_ok(const ganlab_conv_geom* g) {
  return g && g->N > 0 && g->Cin > 0 && g->Hin > 0 && g->Win > 0 && g->Cout > 0 &&
         (g->ks == 1 || g->ks == 3) && g->pad >= 0 && g->pad < g->ks && (g->up == 0 || g->up == 1);
}

}  // namespace

extern "C" {

int ganlab_conv_out_hw(const ganlab_conv_geom* g, int* Hout, int* Wout) {
  if (!geom_ok(g)) return GANLAB_EINVAL;
  const int hv = g->up ? 2 * g->Hin : g->Hin, wv = g->up ? 2 * g->Win : g->Win;
  const int ho = hv + 2 * g->pad - g->ks + 1, wo = wv + 2 * g->pad - g->ks + 1;
  if (ho <= 0 || wo <= 0) return GANLAB_EINVAL;
  if (Hout) *Hout = ho;
  if (Wout) *Wout = wo;
  return GANLAB_OK;
}

long long ganlab_conv_pack_f32(const float* w, float* out, int Cout, int Cin, int ks, int mode, float scale,
                               void* stream) {
  if (Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3) || (mode != GANLAB_PACK_FWD && mode != GANLAB_PACK_DGRAD))
    return GANLAB_EINVAL;
  const int rows = mode == GANLAB_PACK_DGRAD ? Cout : Cin, cols = mode == GANLAB_PACK_DGRAD ? Cin : Cout;
  const int rows_p = round_up_c(rows, cin_pad(ks)), cols_p = round_up_c(cols, 64);
  const long long total = (long long)ks * ks * rows_p * cols_p;
  if (!out) return total;
  if (!w) return GANLAB_EINVAL;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GL_LAUNCH(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, gl_stream(stream), w, out, Cout, Cin,
                     ks * ks, rows, cols, rows_p, cols_p, mode == GANLAB_PACK_DGRAD ? 1 : 0, scale);
  return GL_CHECK_LAUNCH() == GANLAB_OK ? total : GANLAB_ELAUNCH;
}

int ganlab_conv_fwd_f32(const float* x, const float* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                        float bias_scale, int act, float slope, void* stream) {
  if (!geom_ok(g)) return GANLAB_EINVAL;
  return run_conv(x, wp, bias, y, g->N, g->Cin, g->Hin, g->Win, g->Cout, g->ks, g->pad, g->up, bias_scale, act,
                  slope, gl_stream(stream));
}

int ganlab_conv_dgrad_f32(const float* gy, const float* wp, float* gx_virtual, const ganlab_conv_geom* g,
                          void* stream) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK) return GANLAB_EINVAL;
  // the transposed conv is a plain conv of gy (N, Cout, Ho, Wo) with flipped/transposed weights and
  // padding ks-1-pad, producing the gradient w.r.t. the virtual (possibly upsampled) input
  return run_conv(gy, wp, nullptr, gx_virtual, g->N, g->Cout, ho, wo, g->Cin, g->ks, g->ks - 1 - g->pad, 0, 0.f,
                  GANLAB_ACT_NONE, 0.f, gl_stream(stream));
}

size_t ganlab_conv_wgrad_workspace(const ganlab_conv_geom* g) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK) return 0;
  const WgPlan pl = g->ks == 1 ? plan_wgrad<1>(g->N, g->Cin, g->Cout, ho, wo)
                               : plan_wgrad<3>(g->N, g->Cin, g->Cout, ho, wo);
  return (size_t)pl.slots * g->Cout * g->Cin * g->ks * g->ks * sizeof(float);
}

int ganlab_conv_wgrad_f32(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g, float scale,
                          void* workspace, size_t workspace_bytes, void* stream) {
  int ho, wo;
  if (ganlab_conv_out_hw(g, &ho, &wo) != GANLAB_OK || !gy || !x || !gw) return GANLAB_EINVAL;
  const WgPlan pl = g->ks == 1 ? plan_wgrad<1>(g->N, g->Cin, g->Cout, ho, wo)
                               : plan_wgrad<3>(g->N, g->Cin, g->Cout, ho, wo);
  const long long nw = (long long)g->Cout * g->Cin * g->ks * g->ks;
  if (!workspace || workspace_bytes < (size_t)pl.slots * nw * sizeof(float)) return GANLAB_EWORKSPACE;
  WgradArgs a{};
  a.gy = gy; a.x = x; a.part = (float*)workspace;
  a.N = g->N; a.Cin = g->Cin; a.Hi = g->Hin; a.Wi = g->Win;
  a.Hv = g->up ? 2 * g->Hin : g->Hin; a.Wv = g->up ? 2 * g->Win : g->Win;
  a.Cout = g->Cout; a.Ho = ho; a.Wo = wo; a.pad = g->pad; a.up = g->up;
  hipStream_t st = gl_stream(stream);
  const int rc = g->ks == 1 ? run_wgrad_ks<1>(a, pl, st) : run_wgrad_ks<3>(a, pl, st);
  if (rc != GANLAB_OK) return rc;
  GL_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st,
                     (const float*)workspace, gw, nw, pl.slots, scale);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
