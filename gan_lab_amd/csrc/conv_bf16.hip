// bf16-compute implicit-GEMM 3x3 convolution on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16), NCHW.
//
// BASELINE config #2 (StyleGAN 128^2, "bf16 compute / fp32 master"): activations, weights and
// gradients stay fp32 in HBM (the reference's storage type; parameters are the fp32 masters of
// utils/custom_layers.py:147-200), operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on
// their way into LDS and products accumulate in fp32.  Same math as csrc/conv.hip:
// F.conv2d(x * wscale, W, padding=1) at utils/custom_layers.py:202-211 and its two autograd rules.
//
// GEMM view (forward / input gradient):  D[co][p] = sum_{tap,ci} Wp[tap][ci][co] * X[n, ci, oy+ky-1, ox+kx-1]
//   MFMA 16x16x32 bf16:  A[i=co][k]  lane l holds A[l&15][8(l>>4) .. 8(l>>4)+7]   (8 input channels, 16 bytes)
//                        B[k][j=px]  lane l holds B[8(l>>4) .. +7][l&15]
//                        D[i][j]     lane l holds rows 4(l>>4)+r, r = 0..3, column l&15  (as the f32 16x16x4 form)
// LDS images are "8 channels per 16-byte unit": Xs[kg][row][col][8 ci], Ws[tap][kg][co][8 ci]; the 16 lanes of a
// k-group read 256 contiguous bytes and the k-group planes are a multiple of 256 bytes apart -> conflict-free
// ds_read_b128 for both operands.  A workgroup (256 threads, 4 waves) owns 64 output channels x an 8x32 pixel
// patch of one image; K runs over chunks of 32 input channels x 9 taps (36 MFMAs per accumulator tile and chunk).
// Staging: each thread gathers float4 rows of 8 channels (buffer loads, hardware zero fill outside the image),
// transposes them in registers and writes four 16-byte pixel units; the next chunk's loads are issued before the
// current chunk's MFMA loop (register prefetch).
//
// Weight gradient: K = pixels.  A = gy[co][8 consecutive px], B = x[8 consecutive px (shifted by the tap)][ci];
// the three horizontal tap shifts are materialised as three LDS copies of the activation rows so every operand
// read stays a 16-byte aligned ds_read_b128.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 32;          // pixel tile of one workgroup
constexpr int PR = TH + 2, PC = TW + 8; // staged patch: rows oy0-1 .. oy0+8, columns ox0-4 .. ox0+35
constexpr int CK = 32;                  // input channels per K chunk (= one MFMA k-step per tap)
constexpr int COT = 64;                 // output channels per workgroup

__device__ __forceinline__ u32x4 pack8(float a, float b, float c, float d, float e, float f, float g, float h) {
  bf16x8 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
  v[4] = (__bf16)e; v[5] = (__bf16)f; v[6] = (__bf16)g; v[7] = (__bf16)h;
  return __builtin_bit_cast(u32x4, v);
}

// ---- weight packing: OIHW fp32 -> [chunk = ci/32][tap][kg = (ci%32)/8][CO][ci%8] bf16 -------------------------
__global__ void pack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int mode,
                                 float scale) {
  // GEMM roles: forward CO = Cout, CI = Cin; dgrad CO = Cin, CI = Cout with flipped taps
  const int CO = mode == GANLAB_PACK_DGRAD ? Cin : Cout;
  const int CI = mode == GANLAB_PACK_DGRAD ? Cout : Cin;
  const long long total = 9LL * CO * CI;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7);
    long long t = e >> 3;
    const int co = (int)(t % CO);
    t /= CO;
    const int kg = (int)(t & 3);
    t >>= 2;
    const int tap = (int)(t % 9);
    const int chunk = (int)(t / 9);
    const int ci = chunk * CK + kg * 8 + j;
    float v;
    if (mode == GANLAB_PACK_DGRAD)
      v = w[((long long)ci * Cin + co) * 9 + (8 - tap)];   // w[o = ci_gemm][i = co_gemm][flipped tap]
    else
      v = w[((long long)co * Cin + ci) * 9 + tap];
    out[e] = (__bf16)(v * scale);
  }
}

struct BfArgs {
  const float* x;
  const __bf16* wp;
  const float* bias;
  float* y;
  int N, CI, CO, H, W;
  int tiles_x, tiles_y, tiles_co;
  float bias_scale, slope;
  int act;
};

// ---- forward / input-gradient kernel ---------------------------------------------------------------------------
constexpr int X_UNITS = 4 * PR * PC;          // 16-byte units of one activation chunk (kg, row, col)
constexpr int W_UNITS = 9 * 4 * COT;          // 16-byte units of one weight chunk (tap, kg, co)
constexpr int X_ITEMS = 4 * PR * (PC / 4);    // staging items: (kg, row, 4-column group)
constexpr int X_PT = (X_ITEMS + 255) / 256;   // 2
constexpr int W_PT = W_UNITS / 256;           // 9

__global__ __launch_bounds__(256, 2) void conv_fwd_bf16_kernel(BfArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 Xs[X_UNITS];
  __shared__ __attribute__((aligned(16))) u32x4 Ws[W_UNITS];

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int l16 = lane & 15, kgl = lane >> 4;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int txi = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tyi = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = co_t * COT, oy0 = tyi * TH, ox0 = txi * TW;
  const int plane = p.H * p.W;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (long long)n * p.CI * plane), 0, (unsigned)((long long)p.CI * plane * 4), 0x00020000);

  // staging items of this thread: byte offset of channel kg*8 (or out-of-range marker) and LDS unit index
  int goff[X_PT], lunit[X_PT];
#pragma unroll
  for (int i = 0; i < X_PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % (PC / 4);
    int t = e / (PC / 4);
    const int r = t % PR, kg = t / PR;
    const int vy = oy0 - 1 + r, vx = ox0 - 4 + 4 * q;
    const bool ok = e < X_ITEMS && (unsigned)vy < (unsigned)p.H && (unsigned)vx < (unsigned)p.W;
    goff[i] = ok ? ((kg * 8) * plane + vy * p.W + vx) * 4 : (int)0x80000000;
    lunit[i] = e < X_ITEMS ? (kg * PR + r) * PC + 4 * q : -1;
  }
  const int cstride = plane * 4;   // bytes between channels

  float4 xr[X_PT][8];
  u32x4 wr[W_PT];
  auto load_chunk = [&](int c) {
    const int soff = c * CK * plane * 4;
#pragma unroll
    for (int i = 0; i < X_PT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // an out-of-range base stays out of range after adding j*cstride < 2^31 only if it cannot wrap: keep the
        // marker by selecting per load
        const int off = goff[i] == (int)0x80000000 ? (int)0x80000000 : goff[i] + j * cstride;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, soff, 0);
        xr[i][j] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.wp) + (long long)c * 9 * 4 * p.CO;
#pragma unroll
    for (int i = 0; i < W_PT; ++i) {
      const int u = tid + i * 256;           // (tap*4 + kg) * 64 + co
      wr[i] = wsrc[(long long)(u >> 6) * p.CO + co0 + (u & 63)];
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < X_PT; ++i) {
      if (lunit[i] < 0) continue;
      u32x4* dst = Xs + lunit[i];
      dst[0] = pack8(xr[i][0].x, xr[i][1].x, xr[i][2].x, xr[i][3].x, xr[i][4].x, xr[i][5].x, xr[i][6].x, xr[i][7].x);
      dst[1] = pack8(xr[i][0].y, xr[i][1].y, xr[i][2].y, xr[i][3].y, xr[i][4].y, xr[i][5].y, xr[i][6].y, xr[i][7].y);
      dst[2] = pack8(xr[i][0].z, xr[i][1].z, xr[i][2].z, xr[i][3].z, xr[i][4].z, xr[i][5].z, xr[i][6].z, xr[i][7].z);
      dst[3] = pack8(xr[i][0].w, xr[i][1].w, xr[i][2].w, xr[i][3].w, xr[i][4].w, xr[i][5].w, xr[i][6].w, xr[i][7].w);
    }
#pragma unroll
    for (int i = 0; i < W_PT; ++i) Ws[tid + i * 256] = wr[i];
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand base units: B: pixel (row 2wn + nb/2, col 16(nb&1) + l16) of k-group kgl; column 3 = LP(4) - pad(1)
  int bbase[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bbase[nb] = (kgl * PR + 2 * wn + (nb >> 1)) * PC + 16 * (nb & 1) + l16 + 3;
  const int abase = kgl * COT + l16;

  const int nchunks = p.CI / CK;
  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    store_chunk();
    __syncthreads();
    if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      bf16x8 a[4], b[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) a[mb] = __builtin_bit_cast(bf16x8, Ws[tap * 4 * COT + abase + mb * 16]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) b[nb] = __builtin_bit_cast(bf16x8, Xs[bbase[nb] + ky * PC + kx]);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: lane holds channels co0 + 16mb + 4kgl + r of pixel (row, col); 16 lanes -> 64 contiguous bytes
  float* yb = p.y + (long long)n * p.CO * plane;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + mb * 16 + kgl * 4 + r;
      const float bv = p.bias != nullptr ? p.bias[co] * p.bias_scale : 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int oy = oy0 + 2 * wn + (nb >> 1), ox = ox0 + 16 * (nb & 1) + l16;
        float v = acc[mb][nb][r] + bv;
        if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
        yb[(long long)co * plane + oy * p.W + ox] = v;
      }
    }
}

// ---- weight-gradient kernel --------------------------------------------------------------------------------------
// Workgroup: 64 output channels x 32 input channels x 9 taps, summed over a slice of the pixel tiles (8 rows x 32
// columns each); wave w owns output-channel block w (16 channels) -> 2 (ci blocks) x 9 (taps) accumulator tiles.
// LDS: Gs[co 64][row 8][32 px] bf16 (row pitch 64 B, channel pitch padded) and Xc[kx 3][ci 32][row 10][32 px] bf16
// where copy kx holds x[.., col + kx - 1] at position col.
constexpr int G_CP = TH * TW * 2 + 16;      // bytes per gy channel (8 rows x 64 B, +16 B pad: conflict-free A reads)
constexpr int X_CP = PR * TW * 2 + 16;      // bytes per x channel of one shifted copy (10 rows x 64 B, +16 B pad)
constexpr int WG_CI = 32;
struct __attribute__((packed, aligned(4))) F4u {
  float x, y, z, w;
};

struct BfWgArgs {
  const float* gy;
  const float* x;
  float* ws;          // [slot][co][ci][9] partial sums
  int N, CI, CO, H, W;
  int tiles_x, tiles_y, tiles_co, tiles_ci, slots;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(BfWgArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char Gs[COT * G_CP];
  __shared__ __attribute__((aligned(16))) unsigned char Xc[3 * WG_CI * X_CP];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l16 = lane & 15, kgl = lane >> 4;
  int bid = blockIdx.x;
  const int slot = bid % p.slots;
  bid /= p.slots;
  const int ci_t = bid % p.tiles_ci;
  const int co_t = bid / p.tiles_ci;
  const int co0 = co_t * COT, ci0 = ci_t * WG_CI;
  const int plane = p.H * p.W;
  const int ntiles = p.N * p.tiles_y * p.tiles_x;

  f32x4 acc[2][9];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int tile = slot; tile < ntiles; tile += p.slots) {
    const int txi = tile % p.tiles_x;
    int t2 = tile / p.tiles_x;
    const int tyi = t2 % p.tiles_y;
    const int n = t2 / p.tiles_y;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const float* gyb = p.gy + ((long long)n * p.CO + co0) * plane;
    const float* xb = p.x + ((long long)n * p.CI + ci0) * plane;
    __syncthreads();   // previous tile's operand reads are done
    // gy tile: 64 co x 8 rows x 8 float4 = 4096 items, 16 per thread
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int e = tid + i * 256;
      const int q = e & 7, r = (e >> 3) & 7, co = e >> 6;
      const float4 v = *reinterpret_cast<const float4*>(gyb + (long long)co * plane + (oy0 + r) * p.W + ox0 + 4 * q);
      bf16x4 h;
      h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
      *reinterpret_cast<u32x2*>(Gs + co * G_CP + r * (TW * 2) + q * 8) = __builtin_bit_cast(u32x2, h);
    }
    // x copies: 3 shifts x 32 ci x 10 rows x 8 groups of 4 px = 7680 items, 30 per thread
#pragma unroll 5
    for (int i = 0; i < 30; ++i) {
      const int e = tid + i * 256;
      const int q = e & 7;
      int t = e >> 3;
      const int r = t % PR;
      t /= PR;
      const int ci = t & 31, kx = t >> 5;
      const int vy = oy0 - 1 + r, vx = ox0 + 4 * q + kx - 1;
      float4 v = float4{0.f, 0.f, 0.f, 0.f};
      if ((unsigned)vy < (unsigned)p.H) {
        const float* row = xb + (long long)ci * plane + vy * p.W;
        if (vx >= 0 && vx + 3 < p.W) {
          const F4u u = *reinterpret_cast<const F4u*>(row + vx);   // 4-byte aligned 16-byte load
          v = float4{u.x, u.y, u.z, u.w};
        } else {
          if ((unsigned)vx < (unsigned)p.W) v.x = row[vx];
          if ((unsigned)(vx + 1) < (unsigned)p.W) v.y = row[vx + 1];
          if ((unsigned)(vx + 2) < (unsigned)p.W) v.z = row[vx + 2];
          if ((unsigned)(vx + 3) < (unsigned)p.W) v.w = row[vx + 3];
        }
      }
      bf16x4 h;
      h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
      *reinterpret_cast<u32x2*>(Xc + (kx * WG_CI + ci) * X_CP + r * (TW * 2) + q * 8) = __builtin_bit_cast(u32x2, h);
    }
    __syncthreads();
    // K loop: 8 rows x one 32-pixel k-step
#pragma unroll 2
    for (int r = 0; r < TH; ++r) {
      const bf16x8 a = __builtin_bit_cast(
          bf16x8, *reinterpret_cast<const u32x4*>(Gs + (wv * 16 + l16) * G_CP + r * (TW * 2) + kgl * 16));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          const bf16x8 b = __builtin_bit_cast(
              bf16x8, *reinterpret_cast<const u32x4*>(Xc + (kx * WG_CI + nb * 16 + l16) * X_CP + (r + ky) * (TW * 2) +
                                                      kgl * 16));
          acc[nb][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[nb][tap], 0, 0, 0);
        }
    }
  }
  // D[i = co][j = ci]: lane holds co = co0 + 16wv + 4kgl + r, ci = ci0 + 16nb + l16
  float* wsb = p.ws + (long long)slot * p.CO * p.CI * 9;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wv * 16 + kgl * 4 + r, ci = ci0 + nb * 16 + l16;
        wsb[((long long)co * p.CI + ci) * 9 + tap] = acc[nb][tap][r];
      }
}

__global__ void wgrad_bf16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, long long n, int slots,
                                         float scale) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < slots; ++k) s += ws[(long long)k * n + i];   // fixed order: deterministic
  gw[i] = s * scale;
}

bool bf16_ok(const ganlab_conv_geom* g) {
  return g != nullptr && g->ks == 3 && g->pad == 1 && g->up == 0 && g->pool == 0 && g->N > 0 && g->Cin > 0 &&
         g->Cout > 0 && g->Cin % 64 == 0 && g->Cout % 64 == 0 && g->Hin % TH == 0 && g->Win % TW == 0 &&
         (long long)g->Cin * g->Hin * g->Win * 4 < (1LL << 31) && (long long)g->Cout * g->Hin * g->Win * 4 < (1LL << 31);
}

int wgrad_slots(const ganlab_conv_geom* g) {
  const int groups = (g->Cout / COT) * (g->Cin / WG_CI);
  const int ntiles = g->N * (g->Hin / TH) * (g->Win / TW);
  int s = (2 * 256 + groups - 1) / groups;   // ~2 workgroups per CU
  if (s > ntiles) s = ntiles;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" {

int ganlab_conv_bf16_supported(const ganlab_conv_geom* g) { return bf16_ok(g) ? 1 : 0; }

long long ganlab_conv_pack_bf16(const float* w, void* out, int Cout, int Cin, int mode, float scale, void* stream) {
  if (Cout <= 0 || Cin <= 0 || Cout % 64 != 0 || Cin % 64 != 0 || (mode != GANLAB_PACK_FWD && mode != GANLAB_PACK_DGRAD))
    return GANLAB_EINVAL;
  const long long n = 9LL * Cout * Cin;
  if (out == nullptr) return n;
  if (w == nullptr) return GANLAB_EINVAL;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  GL_LAUNCH(pack_bf16_kernel, dim3(blocks), dim3(256), 0, gl_stream(stream), w, reinterpret_cast<__bf16*>(out), Cout,
            Cin, mode, scale);
  const int st = GL_CHECK_LAUNCH();
  return st != GANLAB_OK ? st : n;
}

static int launch_fwd(const float* x, const void* wp, const float* bias, float* y, int N, int CI, int CO, int H, int W,
                      float bias_scale, int act, float slope, void* stream) {
  BfArgs a;
  a.x = x; a.wp = reinterpret_cast<const __bf16*>(wp); a.bias = bias; a.y = y;
  a.N = N; a.CI = CI; a.CO = CO; a.H = H; a.W = W;
  a.tiles_x = W / TW; a.tiles_y = H / TH; a.tiles_co = CO / COT;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  const long long grid = (long long)N * a.tiles_x * a.tiles_y * a.tiles_co;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  GL_LAUNCH(conv_fwd_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, gl_stream(stream), a);
  return GL_CHECK_LAUNCH();
}

int ganlab_conv_fwd_bf16(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                         float bias_scale, int act, float slope, void* stream) {
  if (x == nullptr || wp == nullptr || y == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_ok(g)) return GANLAB_EUNSUPPORTED;
  return launch_fwd(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin, g->Win, bias_scale, act, slope, stream);
}

int ganlab_conv_dgrad_bf16(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream) {
  if (gy == nullptr || wp == nullptr || gx == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_ok(g)) return GANLAB_EUNSUPPORTED;
  return launch_fwd(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin, g->Win, 0.f, GANLAB_ACT_NONE, 0.f, stream);
}

size_t ganlab_conv_wgrad_bf16_workspace(const ganlab_conv_geom* g) {
  if (!bf16_ok(g)) return 0;
  return (size_t)wgrad_slots(g) * g->Cout * g->Cin * 9 * sizeof(float);
}

int ganlab_conv_wgrad_bf16(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g, float scale,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (gy == nullptr || x == nullptr || gw == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_ok(g)) return GANLAB_EUNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < ganlab_conv_wgrad_bf16_workspace(g)) return GANLAB_EWORKSPACE;
  BfWgArgs a;
  a.gy = gy; a.x = x; a.ws = reinterpret_cast<float*>(workspace);
  a.N = g->N; a.CI = g->Cin; a.CO = g->Cout; a.H = g->Hin; a.W = g->Win;
  a.tiles_x = a.W / TW; a.tiles_y = a.H / TH; a.tiles_co = a.CO / COT; a.tiles_ci = a.CI / WG_CI;
  a.slots = wgrad_slots(g);
  const long long grid = (long long)a.tiles_co * a.tiles_ci * a.slots;
  GL_LAUNCH(conv_wgrad_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, gl_stream(stream), a);
  const long long n = 9LL * a.CO * a.CI;
  GL_LAUNCH(wgrad_bf16_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gl_stream(stream), a.ws, gw, n,
            a.slots, scale);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
