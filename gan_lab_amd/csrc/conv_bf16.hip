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
// the three horizontal tap shifts are materialised as three LDS copies of the activation rows (built from ONE aligned
// load plus a lane shuffle) so every operand read stays a 16-byte aligned ds_read_b128.
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
// Workgroup: 64 output channels x 32 input channels x 9 taps, summed over a slice of the pixel tiles (4 rows x 32
// columns each); wave w owns output-channel block w (16 channels) -> 2 (ci blocks) x 9 (taps) accumulator tiles.
// LDS: Gs[co 64][row 4][32 px] bf16 (row pitch 64 B, channel pitch padded) and Xc[kx 3][ci 32][row 6][32 px] bf16
// where copy kx holds x[.., col + kx - 1] at position col.
// Pixel tile of the weight gradient: 4 rows x 32 columns.  56 KB of LDS per workgroup -> TWO workgroups per CU, so one
// stages (global -> bf16 -> LDS) while the other multiplies; with 8-row tiles (97 KB, one workgroup per CU) staging
// and MFMA alternated and the kernel ran at 145 TFLOP/s.
constexpr int WTH = 4, WPR = WTH + 2;
constexpr int G_CP = WTH * TW * 2 + 16;     // bytes per gy channel (4 rows x 64 B, +16 B pad: conflict-free A reads)
constexpr int X_CP = WPR * TW * 2 + 16;     // bytes per x channel of one shifted copy (6 rows x 64 B, +16 B pad)
constexpr int WG_CI = 32;

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

  // register prefetch: the next tile's gy / x values (fp32) are loaded while the MFMA loop of the current tile runs and
  // are converted, shuffled and written to LDS after the barrier
  float4 gr[8], xr[6];
  float xe[6];           // edge pixel of the lane's group: x[ox0 - 1] for q == 0, x[ox0 + 32] for q == 7
  auto load_tile = [&](int tile) {
    const int txi = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int tyi = t2 % p.tiles_y, n = t2 / p.tiles_y;
    const int oy0 = tyi * WTH, ox0 = txi * TW;
    const float* gyb = p.gy + ((long long)n * p.CO + co0) * plane + (long long)oy0 * p.W + ox0;
    const float* xb = p.x + ((long long)n * p.CI + ci0) * plane;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + i * 256;
      const int q = e & 7, r = (e >> 3) & 3, co = e >> 5;
      gr[i] = *reinterpret_cast<const float4*>(gyb + (long long)co * plane + r * p.W + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int e = tid + i * 256;
      const int q = e & 7;
      const int t = e >> 3;
      const int r = t % WPR, ci = t / WPR;
      const int vy = oy0 - 1 + r;
      const bool rok = (unsigned)vy < (unsigned)p.H;
      const float* row = xb + (long long)ci * plane + (long long)(rok ? vy : 0) * p.W + ox0;
      xr[i] = rok ? *reinterpret_cast<const float4*>(row + 4 * q) : float4{0.f, 0.f, 0.f, 0.f};
      float ev = 0.f;
      if (q == 0 && rok && ox0 > 0) ev = row[-1];
      if (q == 7 && rok && ox0 + TW < p.W) ev = row[TW];
      xe[i] = ev;
    }
  };
  auto store_tile = [&]() {
    // gy tile: 64 co x 4 rows x 8 float4 = 2048 items, 8 per thread
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + i * 256;
      const int q = e & 7, r = (e >> 3) & 3, co = e >> 5;
      bf16x4 h;
      h[0] = (__bf16)gr[i].x; h[1] = (__bf16)gr[i].y; h[2] = (__bf16)gr[i].z; h[3] = (__bf16)gr[i].w;
      *reinterpret_cast<u32x2*>(Gs + co * G_CP + r * (TW * 2) + q * 8) = __builtin_bit_cast(u32x2, h);
    }
    // x: ONE aligned float4 per (ci, row, 4-pixel group) = 1536 items, 6 per thread; the two shifted copies take their
    // missing pixel from the neighbouring lane (groups are lane-consecutive) or, at the tile's left / right edge,
    // from the extra scalar load
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int e = tid + i * 256;
      const int q = e & 7;
      const int t = e >> 3;
      const int r = t % WPR, ci = t / WPR;
      const float4 v = xr[i];
      float left = __shfl_up(v.w, 1, 8), right = __shfl_down(v.x, 1, 8);
      if (q == 0) left = xe[i];
      if (q == 7) right = xe[i];
      unsigned char* dst = Xc + ci * X_CP + r * (TW * 2) + q * 8;
      bf16x4 h;
      h[0] = (__bf16)left; h[1] = (__bf16)v.x; h[2] = (__bf16)v.y; h[3] = (__bf16)v.z;
      *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);                          // kx = 0: x[col - 1]
      h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
      *reinterpret_cast<u32x2*>(dst + WG_CI * X_CP) = __builtin_bit_cast(u32x2, h);           // kx = 1
      h[0] = (__bf16)v.y; h[1] = (__bf16)v.z; h[2] = (__bf16)v.w; h[3] = (__bf16)right;
      *reinterpret_cast<u32x2*>(dst + 2 * WG_CI * X_CP) = __builtin_bit_cast(u32x2, h);       // kx = 2: x[col + 1]
    }
  };

  int tile = slot;
  if (tile < ntiles) load_tile(tile);
  while (tile < ntiles) {
    __syncthreads();   // previous tile's operand reads are done
    store_tile();
    __syncthreads();
    const int next = tile + p.slots;
    if (next < ntiles) load_tile(next);   // in flight during the MFMA loop
    // K loop: 4 rows x one 32-pixel k-step
#pragma unroll 2
    for (int r = 0; r < WTH; ++r) {
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
    tile = next;
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
  const int ntiles = g->N * (g->Hin / WTH) * (g->Win / TW);
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
  a.tiles_x = a.W / TW; a.tiles_y = a.H / WTH; a.tiles_co = a.CO / COT; a.tiles_ci = a.CI / WG_CI;
  a.slots = wgrad_slots(g);
  const long long grid = (long long)a.tiles_co * a.tiles_ci * a.slots;
  GL_LAUNCH(conv_wgrad_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, gl_stream(stream), a);
  const long long n = 9LL * a.CO * a.CI;
  GL_LAUNCH(wgrad_bf16_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gl_stream(stream), a.ws, gw, n,
            a.slots, scale);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
