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

#include <cstdlib>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TW = 32;                  // strip width of the weight-gradient kernels
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
  int tiles_x, tiles_y, tiles_co;   // H, W: the resolution the 3x3 taps run at (2x the input's with UP, 2x the output's with POOL)
  float bias_scale, slope, oscale;  // y = act(oscale * (pooled) conv + bias * bias_scale)
  int act;
  // split-K (few output tiles, long contraction: the 512-channel 16x16 layers at batch 8 are 64 workgroups): workgroup
  // (tile, ks) contracts chunks [ks * ks_chunks, ..) and writes its raw accumulators at the taps' resolution to
  // part[ks][n][co][H][W]; bf16_splitk_finish_kernel adds them in a fixed order, pools, scales, adds the bias, activates
  int ksplit, ks_chunks;
  float* part;
};

// ---- forward / input-gradient kernel ---------------------------------------------------------------------------
// Pixel tile: 256 pixels = 16 MFMA column blocks, 4 per wave.  TH x TW = 8 x 32 for maps at least 32 wide, 16 x 16 for
// the 16-pixel-wide maps (one whole 16^2 image per workgroup).
constexpr int W_UNITS = 9 * 4 * COT;          // 16-byte units of one weight chunk (tap, kg, co)
constexpr int W_PT = W_UNITS / 256;           // 9

// UP: the input is the nearest 2x upsample of x (H/2 x W/2 in memory) - the staging reads each source pixel pair once and
// writes it to two columns, so the upsampled tensor is never materialised.  POOL: the epilogue adds the 2 x 2 output
// pixels (rows: two accumulator tiles of the same wave; columns: the neighbouring lane) and stores H/2 x W/2.  The four
// stride-2 passes are these two: conv(up2 x) and the input gradient of pool2(conv x) take UP (oscale 1 / 0.25),
// pool2(conv x) and the input gradient of conv(up2 x) take POOL (oscale 0.25 / 1).
template <int TH, int TW, bool UP = false, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv_fwd_bf16_kernel(BfArgs p) {
  constexpr int PR = TH + 2, PC = TW + 8;       // staged patch: rows oy0-1 .. oy0+TH, columns ox0-4 .. ox0+TW+3
  constexpr int X_UNITS = 4 * PR * PC;          // 16-byte units of one activation chunk (kg, row, col)
  constexpr int X_ITEMS = 4 * PR * (PC / 4);    // staging items: (kg, row, 4-column group)
  constexpr int X_PT = (X_ITEMS + 255) / 256;   // 2
  constexpr int COLB = TW / 16;                 // MFMA column blocks per tile row
  static_assert(TH * TW == 256 && (PR * PC) % 16 == 0, "k-group planes must stay a multiple of 256 bytes apart");
  __shared__ __attribute__((aligned(16))) u32x4 Xs[X_UNITS];
  __shared__ __attribute__((aligned(16))) u32x4 Ws[W_UNITS];

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int l16 = lane & 15, kgl = lane >> 4;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int ks = p.ksplit > 1 ? bid % p.ksplit : 0;
  if (p.ksplit > 1) bid /= p.ksplit;
  const int co_t = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int txi = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int tyi = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = co_t * COT, oy0 = tyi * TH, ox0 = txi * TW;
  const int iW = UP ? p.W >> 1 : p.W;
  const int plane = UP ? (p.H >> 1) * iW : p.H * p.W;        // input plane (elements)

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
    goff[i] = !ok ? (int)0x80000000
                  : UP ? ((kg * 8) * plane + (vy >> 1) * iW + (vx >> 1)) * 4 : ((kg * 8) * plane + vy * p.W + vx) * 4;
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
        if constexpr (UP) {   // columns vx .. vx+3 of the upsampled row = source pixels vx/2, vx/2 + 1, each twice
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_in, off, soff, 0);
          xr[i][j] = float4{__uint_as_float(v.x), __uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.y)};
        } else {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, soff, 0);
          xr[i][j] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
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

  // operand base units: B: column block 4wn + nb = pixel (row (4wn + nb) / COLB, col 16 ((4wn + nb) % COLB) + l16) of
  // k-group kgl; column 3 = LP(4) - pad(1)
  int bbase[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
    bbase[nb] = (kgl * PR + (4 * wn + nb) / COLB) * PC + 16 * ((4 * wn + nb) % COLB) + l16 + 3;
  const int abase = kgl * COT + l16;

  const int c_first = ks * p.ks_chunks;
  const int nchunks = p.ksplit > 1 ? min(p.CI / CK, c_first + p.ks_chunks) : p.CI / CK;
  load_chunk(c_first);
  for (int c = c_first; c < nchunks; ++c) {
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

  if (p.ksplit > 1) {      // raw partial sums at the taps' resolution; the finish kernel does the rest
    const int oplane = p.H * p.W;
    float* yb = p.part + ((long long)ks * p.N + n) * p.CO * oplane;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + mb * 16 + kgl * 4 + r;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int oy = oy0 + (4 * wn + nb) / COLB, ox = ox0 + 16 * ((4 * wn + nb) % COLB) + l16;
          yb[(long long)co * oplane + oy * p.W + ox] = acc[mb][nb][r];
        }
      }
    return;
  }
  // epilogue: lane holds channels co0 + 16mb + 4kgl + r of pixel (row, col); 16 lanes -> 64 contiguous bytes
  if constexpr (!POOL) {
    const int oplane = p.H * p.W;
    float* yb = p.y + (long long)n * p.CO * oplane;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + mb * 16 + kgl * 4 + r;
        const float bv = p.bias != nullptr ? p.bias[co] * p.bias_scale : 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int oy = oy0 + (4 * wn + nb) / COLB, ox = ox0 + 16 * ((4 * wn + nb) % COLB) + l16;
          float v = acc[mb][nb][r] * p.oscale + bv;
          if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
          yb[(long long)co * oplane + oy * p.W + ox] = v;
        }
      }
  } else {
    // the wave's four column blocks are two tile rows x COLB... : blocks (b, b + COLB) of the same column range are
    // vertical neighbours when COLB == 2 (rows 2wn, 2wn+1), blocks (2k, 2k+1) when COLB == 1 (rows 4wn+2k, 4wn+2k+1)
    const int oW = p.W >> 1, oplane = (p.H >> 1) * oW;
    float* yb = p.y + (long long)n * p.CO * oplane;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + mb * 16 + kgl * 4 + r;
        const float bv = p.bias != nullptr ? p.bias[co] * p.bias_scale : 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int b0 = COLB == 2 ? k : 2 * k, b1 = COLB == 2 ? k + 2 : 2 * k + 1;
          float sum = acc[mb][b0][r] + acc[mb][b1][r];
          sum += __shfl_xor(sum, 1, 64);                       // the horizontal neighbour (l16 ^ 1)
          const int oy = (oy0 + (4 * wn + b0) / COLB) >> 1;
          const int ox = (ox0 + 16 * ((4 * wn + b0) % COLB) + l16) >> 1;
          float v = sum * p.oscale + bv;
          if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
          if ((l16 & 1) == 0) yb[(long long)co * oplane + oy * oW + ox] = v;
        }
      }
  }
}

// y = act(oscale * pool?(sum_s part[s]) + bias * bias_scale): finishes a split-K launch (fixed summation order)
__global__ void bf16_splitk_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                          int S, int N, int CO, int H, int W, int pool, float oscale, float bias_scale,
                                          int act, float slope) {
  const int oH = pool ? H >> 1 : H, oW = pool ? W >> 1 : W;
  const long long total = (long long)N * CO * oH * oW, splane = (long long)N * CO * H * W;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ox = (int)(i % oW);
  long long t = i / oW;
  const int oy = (int)(t % oH);
  t /= oH;                                   // n * CO + co
  const int co = (int)(t % CO);
  float v = 0.f;
  for (int s = 0; s < S; ++s) {
    const float* ps = part + s * splane + t * H * W;
    if (pool) {
      const float* q = ps + (2 * oy) * W + 2 * ox;
      v += (q[0] + q[1]) + (q[W] + q[W + 1]);
    } else {
      v += ps[oy * W + ox];
    }
  }
  v = v * oscale + (bias != nullptr ? bias[co] * bias_scale : 0.f);
  if (act == GANLAB_ACT_LRELU) v = gl_lrelu(v, slope);
  y[i] = v;
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

// ---- weight gradient, rolling rows fed by LDS-DMA (the default) ----------------------------------------------------
// The tile kernel above writes every activation row to LDS three times (one bf16 copy per horizontal tap) and is bound by
// those ds_write_b64, not by the matrix cores; a first rolling kernel with register staging ran one global-load latency
// per row.  Here the fp32 rows go HBM -> LDS with global_load_lds_dwordx4 (no staging registers, three rows in flight
// per pipeline) and become bf16 in REGISTERS, where the tap shift is free - it is the choice of which fp32 pairs
// v_cvt_pk_bf16_f32 packs:
//   lane (k-group g, l16 = ci): f[4..11] = x[ci][8g .. 8g+7] (2 ds_read_b128), f[3] / f[12] = the neighbouring k-group's
//                               last / first pixel (ds_bpermute) or the strip's halo pixel; B_kx[j] = f[3 + j + kx]
//   lane (g, l16 = co):         A[j] = gy[co][8g + j] (2 ds_read_b128)
// A pipeline (4 waves) owns 64 co x 64 ci x 9 taps (wave = 32 x 32: 2 x 2 x 9 accumulator tiles) and walks segments of RS
// image rows x 32 columns.  One step = one activation row R: it meets the gradient rows R+1, R, R-1 (ky = 0, 1, 2),
// whose fragments roll through three register slots (the loop is unrolled by 3 so the slot <-> ky map is static), so
// each step brings ONE x row and ONE gy row: 16.5 KB into a ring of four slots, one barrier per step, 36 MFMAs per wave.
// Every step runs all 36 MFMAs: a gradient row outside the segment, an activation row outside the image and a halo
// pixel outside the image are read from a slot of zeros (an address offset), which keeps the accumulators out of
// branches.
// LDS image of a row slice: [channel][8 units of 4 pixels], unit u of channel c at position u ^ swz(c) - an LDS-DMA
// writes 64 x 16 bytes in lane order, so the swizzle is applied to the SOURCE address (8 lanes still fetch one whole
// 128-byte line) and again by the reader; swz() makes the 16 lanes of every ds_read_b128 group hit 16 different slots.
// A 512-thread workgroup runs TWO pipelines on different segments of the same 64 x 64 tile and adds the second one's
// accumulators to the first through LDS before the flush, which halves the partial-sum traffic.
// Measured per row and pipeline pair (128 -> 128 @128^2 x8): 0.27 us of MFMA, 0.17 us of LDS-DMA (the address unit
// takes 64 B per clock: 16.5 KB + two 64-line halo gathers) and 0.07 us of everything else - and they ADD UP: with the
// DMAs issued but never waited for the time is the same, so the cost is the address unit's, not the latency's.
constexpr int RW_T = 64;
constexpr int RW_D = 4;                                  // ring slots: compute on t, DMAs of t+1 .. t+3 in flight
constexpr int RW_ROW = 8 * RW_T;                         // 16-byte units of a 32-pixel row slice of 64 channels
constexpr int RW_SLOT = 2 * RW_ROW + 32;                 // gy slice, x slice, 64 + 64 halo pixels
constexpr int RW_PIPE = RW_D * RW_SLOT;                  // 66 KB per pipeline

struct BfWrArgs {
  const float* gy;
  const float* x;
  float* ws;          // [flush slot][pair][wave][tile][lane] float4
  int N, CI, CO, H, W;
  int tiles_ci, npairs, RS, segs_y, strips, nseg, streams;
};

__device__ __forceinline__ int rw_swz(int ch) { return (((ch >> 1) & 3) << 1) | ((((ch & 15) + 4) >> 3) & 1); }

// half-resolution slices ([channel][4 units]): the 16 lanes of a ds_read_b128 group are 8 channels of one k-group and
// the 8 complementary channels of its neighbour; P = (0, 3, 2, 1) over channel / 4 keeps their 16 slots distinct
__device__ __forceinline__ int rw_swz_up(int ch) { return (0x6C >> (((ch >> 2) & 3) * 2)) & 3; }

typedef __attribute__((address_space(1))) const void* rw_gptr;
typedef __attribute__((address_space(3))) void* rw_lptr;

// XUP / GUP: the activation / the output gradient is stored at HALF the resolution of the taps and the kernel sees its
// nearest 2x upsample (the weight gradient of conv(up2 x), resp. of pool2(conv x) whose gy is up2(gy_pooled) / 4 - the 1/4
// goes into the reduce pass's scale).  Row R of the taps is source row R >> 1, a strip's 32 columns are 16 source pixels:
// four 16-byte units per channel instead of eight (position u ^ swz_up(c)), half the DMAs, and the fragment builder
// writes every source pixel twice before the tap shift picks its pairs.
template <bool XUP, bool GUP>
__global__ __launch_bounds__(512, 2) void conv_wgrad_bf16_roll_kernel(BfWrArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 smem[2 * RW_PIPE + RW_SLOT];   // two rings + one slot of zeros

  // wave-uniform quantities are made SGPRs explicitly: the row / segment arithmetic then costs no vector registers
  const int kgp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int wv = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
  const int lane = threadIdx.x & 63, l16 = lane & 15, kg = lane >> 4;
  const int wa = wv >> 1, wc = wv & 1;
  const int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % p.npairs, slot = bid / p.npairs;
  const int ci0 = (pair % p.tiles_ci) * RW_T, co0 = (pair / p.tiles_ci) * RW_T;
  const int xW = XUP ? p.W >> 1 : p.W, gW = GUP ? p.W >> 1 : p.W;                 // row lengths in memory
  const int xplane = XUP ? (p.H >> 1) * xW : p.H * p.W, gplane = GUP ? (p.H >> 1) * gW : p.H * p.W;
  u32x4* const ring = smem + kgp * RW_PIPE;
  for (int i = threadIdx.x; i < RW_SLOT; i += 512) smem[2 * RW_PIPE + i] = u32x4{0u, 0u, 0u, 0u};
  const unsigned zslot = (unsigned)((2 - kgp) * RW_PIPE) * 16u;       // byte offset of the zero slot from `ring`
  // LDS byte addresses of this lane's fragment reads inside a ring slot (the reads are inline asm: hipcc would put
  // s_waitcnt vmcnt(0) in front of every ds_read it emits itself while an LDS-DMA is in flight)
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(rw_lptr)ring;
  const int sw = rw_swz(l16), swu = rw_swz_up(l16);
  unsigned aaddr[2][2], baddr[2][2], eaddr[2];
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    const int ca = wa * 32 + nn * 16 + l16, cb = wc * 32 + nn * 16 + l16;
#pragma unroll
    for (int o = 0; o < 2; ++o) {     // half-resolution slices: ONE unit (4 source pixels) per lane, index o unused
      aaddr[nn][o] = lds0 + (unsigned)(GUP ? ca * 4 + (kg ^ swu) : ca * 8 + ((2 * kg + o) ^ sw)) * 16u;
      baddr[nn][o] = lds0 + (unsigned)(RW_ROW + (XUP ? cb * 4 + (kg ^ swu) : cb * 8 + ((2 * kg + o) ^ sw))) * 16u;
    }
    eaddr[nn] = lds0 + (unsigned)(2 * RW_ROW) * 16u + (unsigned)((kg >> 1) * 64 + cb) * 4u;
  }

  // DMA work of this wave per step: LDS-DMA instructions 2wv, 2wv + 1 of the gy slice and of the x slice (8 channels x
  // 8 units each; the same per-lane byte offset serves both tensors) and, for waves 0 / 1, one 4-byte gather of the 64
  // left / right halo pixels
  // (a half-resolution slice is 4 instructions of 16 channels x 4 units: one per wave)
  unsigned dgoff[2], dxoff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ch = 8 * (2 * wv + k) + (lane >> 3), chu = 16 * wv + (lane >> 2);
    dgoff[k] = GUP ? (unsigned)(chu * gplane + 4 * ((lane & 3) ^ rw_swz_up(chu))) * 4u
                   : (unsigned)(ch * gplane + 4 * ((lane & 7) ^ rw_swz(ch))) * 4u;
    dxoff[k] = XUP ? (unsigned)(chu * xplane + 4 * ((lane & 3) ^ rw_swz_up(chu))) * 4u
                   : (unsigned)(ch * xplane + 4 * ((lane & 7) ^ rw_swz(ch))) * 4u;
  }
  const unsigned hoff = (unsigned)(lane * xplane) * 4u;

  f32x4 acc[2][2][9];
#pragma unroll
  for (int na = 0; na < 2; ++na)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[na][nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int RS = p.RS, T = (RS + 4) / 3 * 3;                   // RS + 2 steps, rounded up to the unroll
  const int nloop = (p.nseg + p.streams - 1) / p.streams;      // the same for both pipelines: barriers are shared
  for (int it = 0; it < nloop; ++it) {
    const int seg = (slot * 2 + kgp) + it * p.streams;
    const bool live = seg < p.nseg && slot * 2 + kgp < p.streams;
    int sx = 0, sy = 0, n = 0;
    if (live) {
      sx = seg % p.strips;
      const int t2 = seg / p.strips;
      sy = t2 % p.segs_y;
      n = t2 / p.segs_y;
    }
    const int ox0 = sx * 32, r0 = sy * RS;
    const char* gyb = reinterpret_cast<const char*>(p.gy + ((long long)n * p.CO + co0) * gplane);
    const char* xb = reinterpret_cast<const char*>(p.x + ((long long)n * p.CI + ci0) * xplane);
    const bool lok = ox0 > 0, rok = ox0 + 32 < p.W;
    // halo gather: wave 0 fetches x[.., ox0 - 1], wave 1 x[.., ox0 + 32] (clamped into the row; lanes whose halo
    // pixel lies outside the image read the zero slot instead); in source columns when x is stored at half resolution
    const int xc0 = XUP ? ox0 >> 1 : ox0, xcn = XUP ? 16 : 32, gc0 = GUP ? ox0 >> 1 : ox0;
    const int hcol = (wv & 1) ? (rok ? xc0 + xcn : xc0 + xcn - 1) : (lok ? xc0 - 1 : xc0);
    const bool ezero = (kg == 0 && !lok) || (kg == 3 && !rok);

    auto issue = [&](int t) {
      u32x4* q = ring + (t & (RW_D - 1)) * RW_SLOT;
      int gr = r0 + t, xr = r0 - 1 + t;                       // rows; out of range -> any valid row, never read
      gr = gr < p.H ? gr : p.H - 1;
      xr = xr < 0 ? 0 : (xr < p.H ? xr : p.H - 1);
      if (GUP) gr >>= 1;
      if (XUP) xr >>= 1;
      const unsigned grow = (unsigned)(gr * gW + gc0) * 4u, xrow = (unsigned)(xr * xW + xc0) * 4u;
#pragma unroll
      for (int k = 0; k < (GUP ? 1 : 2); ++k)
        __builtin_amdgcn_global_load_lds((rw_gptr)(gyb + (dgoff[k] + grow)),
                                         (rw_lptr)(q + (GUP ? wv : 2 * wv + k) * 64), 16, 0, 0);
#pragma unroll
      for (int k = 0; k < (XUP ? 1 : 2); ++k)
        __builtin_amdgcn_global_load_lds((rw_gptr)(xb + (dxoff[k] + xrow)),
                                         (rw_lptr)(q + RW_ROW + (XUP ? wv : 2 * wv + k) * 64), 16, 0, 0);
      if (wv < 2)
        __builtin_amdgcn_global_load_lds((rw_gptr)(xb + (hoff + (unsigned)(xr * xW + hcol) * 4u)),
                                         (rw_lptr)(reinterpret_cast<float*>(q + 2 * RW_ROW) + wv * 64), 4, 0, 0);
    };
    // outstanding DMAs of two steps: per step 4 (3 with a half-resolution operand), +1 for waves 0 and 1 (the halo)
    auto wait_two_steps = [&]() {
      if (XUP || GUP) {
        if (wv < 2)
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        if (wv < 2)
          asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
    };

    bf16x8 a[3][2];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
      for (int na = 0; na < 2; ++na) a[s_][na] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});

    auto compute = [&](int t, auto phase) {
      constexpr int P = decltype(phase)::value;
      const unsigned so = (unsigned)((t & (RW_D - 1)) * RW_SLOT) * 16u;
      const int R = r0 - 1 + t;
      const bool gok = live && t < RS, xok = live && t < RS + 2 && (unsigned)R < (unsigned)p.H;
      const unsigned sog = gok ? so : zslot, sox = xok ? so : zslot;
      u32x4 fa[2][2], fm[2][2];
      float fe[2];
#pragma unroll
      for (int na = 0; na < 2; ++na) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[na][0]) : "v"(aaddr[na][0] + sog));
        if (!GUP) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[na][1]) : "v"(aaddr[na][1] + sog));
        else fa[na][1] = u32x4{0u, 0u, 0u, 0u};      // unused (never a copy of fa[na][0]: that register is still in flight)
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fm[nb][0]) : "v"(baddr[nb][0] + sox));
        if (!XUP) asm volatile("ds_read_b128 %0, %1" : "=v"(fm[nb][1]) : "v"(baddr[nb][1] + sox));
        else fm[nb][1] = u32x4{0u, 0u, 0u, 0u};
        asm volatile("ds_read_b32 %0, %1" : "=v"(fe[nb]) : "v"(ezero ? eaddr[nb] + zslot : eaddr[nb] + sox));
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fm[0][0]), "+v"(fm[0][1]),
                     "+v"(fe[0]), "+v"(fm[1][0]), "+v"(fm[1][1]), "+v"(fe[1]));
      // from here to the barrier the compiler is free to interleave the conversions with the MFMAs
#pragma unroll
      for (int na = 0; na < 2; ++na) {
        const u32x4 f0 = fa[na][0], f1 = fa[na][1];
        if (GUP)    // 4 source pixels, each twice
          a[P][na] = __builtin_bit_cast(
              bf16x8, pack8(__uint_as_float(f0.x), __uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.y),
                            __uint_as_float(f0.z), __uint_as_float(f0.z), __uint_as_float(f0.w), __uint_as_float(f0.w)));
        else
          a[P][na] = __builtin_bit_cast(
              bf16x8, pack8(__uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.z), __uint_as_float(f0.w),
                            __uint_as_float(f1.x), __uint_as_float(f1.y), __uint_as_float(f1.z), __uint_as_float(f1.w)));
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const u32x4 m0 = fm[nb][0], m1 = XUP ? fm[nb][0] : fm[nb][1];
        const float e = fe[nb];
        const float pl = __int_as_float(__builtin_amdgcn_ds_bpermute((lane - 16) << 2, (int)m1.w));
        const float pr = __int_as_float(__builtin_amdgcn_ds_bpermute((lane + 16) << 2, (int)m0.x));
        const float f3 = kg == 0 ? e : pl, f12 = kg == 3 ? e : pr;
        // (XUP: m1 == m0 holds source pixels 4g .. 4g+3, so the neighbours' m1.w / m0.x are source pixels 4g-1 / 4g+4)
        const float f4 = __uint_as_float(m0.x), f5 = __uint_as_float(XUP ? m0.x : m0.y),
                    f6 = __uint_as_float(XUP ? m0.y : m0.z), f7 = __uint_as_float(XUP ? m0.y : m0.w),
                    f8 = __uint_as_float(XUP ? m0.z : m1.x), f9 = __uint_as_float(XUP ? m0.z : m1.y),
                    f10 = __uint_as_float(XUP ? m0.w : m1.z), f11 = __uint_as_float(XUP ? m0.w : m1.w);
        bf16x8 b[3];
        b[0] = __builtin_bit_cast(bf16x8, pack8(f3, f4, f5, f6, f7, f8, f9, f10));     // x[8g - 1 .. 8g + 6]
        b[1] = __builtin_bit_cast(bf16x8, pack8(f4, f5, f6, f7, f8, f9, f10, f11));    // x[8g .. 8g + 7]
        b[2] = __builtin_bit_cast(bf16x8, pack8(f5, f6, f7, f8, f9, f10, f11, f12));   // x[8g + 1 .. 8g + 8]
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int sl = (P + 3 - ky) % 3;      // gradient row r0 + t - ky sits in slot (P - ky) mod 3
#pragma unroll
          for (int na = 0; na < 2; ++na)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
              acc[na][nb][ky * 3 + kx] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[sl][na], b[kx], acc[na][nb][ky * 3 + kx], 0, 0, 0);
        }
      }
    };
    // DMA(t+3) goes into the slot read in step t-1 (all waves are past that step's barrier); before this step's
    // barrier each wave waits until only its DMAs of t+2 and t+3 are outstanding, i.e. its part of t+1 has landed
    auto step = [&](int t, auto phase) {
      compute(t, phase);
      issue(t + 3);          // after the MFMAs are queued (issued first it costs 5 - 10 % more)
      wait_two_steps();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };

    issue(0);
    issue(1);
    issue(2);
    wait_two_steps();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // first pass: the zero slot's ds_writes
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < T; t += 3) {      // T is a multiple of 3: steps past RS + 1 multiply zeros
      step(t, std::integral_constant<int, 0>{});
      step(t + 1, std::integral_constant<int, 1>{});
      step(t + 2, std::integral_constant<int, 2>{});
    }
    // the DMAs issued for steps T .. T+2 still target the ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // the second pipeline hands its accumulators over through LDS, one tap row (12 tiles) per round
  f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    __syncthreads();
    if (kgp == 1) {
#pragma unroll
      for (int na = 0; na < 2; ++na)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            red[((wv * 12) + (na * 2 + nb) * 3 + kx) * 64 + lane] = acc[na][nb][ky * 3 + kx];
    }
    __syncthreads();
    if (kgp == 0) {
#pragma unroll
      for (int na = 0; na < 2; ++na)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            acc[na][nb][ky * 3 + kx] += red[((wv * 12) + (na * 2 + nb) * 3 + kx) * 64 + lane];
    }
  }
  if (kgp == 1) return;
  // flush in register order (whole 1 KB wave stores): ws[slot][pair][wave][tile = (na, nb, tap)][lane] float4; the
  // reduce kernel sums the slots and scatters to [co][ci][9]
  f32x4* wsb = reinterpret_cast<f32x4*>(p.ws) + (((long long)slot * p.npairs + pair) * 4 + wv) * (36 * 64) + lane;
#pragma unroll
  for (int na = 0; na < 2; ++na)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wsb[((na * 2 + nb) * 9 + tap) * 64] = acc[na][nb][tap];
}

// D[i = co][j = ci] of tile (na, nb, tap) of wave (wa, wc): lane holds co = 32wa + 16na + 4kg + r, ci = 32wc + 16nb + l16
__global__ void wgrad_bf16_roll_reduce_kernel(const f32x4* __restrict__ ws, float* __restrict__ gw, int npairs,
                                              int tiles_ci, int CI, int slots, float scale) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;        // (pair, wave, tile, lane)
  if (e >= npairs * 4 * 36 * 64) return;
  const int lane = e & 63, tile = (e >> 6) % 36, wv = (e / (64 * 36)) & 3, pair = e / (64 * 36 * 4);
  f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < slots; ++k) sum += ws[(long long)k * npairs * (4 * 36 * 64) + e];   // fixed order: deterministic
  const int tap = tile % 9, nb = (tile / 9) & 1, na = tile / 18;
  const int co = (pair / tiles_ci) * RW_T + (wv >> 1) * 32 + na * 16 + (lane >> 4) * 4;
  const int ci = (pair % tiles_ci) * RW_T + (wv & 1) * 32 + nb * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) gw[((long long)(co + r) * CI + ci) * 9 + tap] = sum[r] * scale;
}

struct WrPlan {
  int RS, segs_y, strips, nseg, streams, flush_slots;
};

bool wr_enabled() {
  static const bool v = [] {
    const char* e = getenv("GANLAB_BF16_WGRAD_ROLL");
    return e == nullptr || atoi(e) != 0;
  }();
  return v;
}

WrPlan wr_plan(const ganlab_conv_geom* g) {
  WrPlan q;
  const int H = g->Hin * (g->up ? 2 : 1), W = g->Win * (g->up ? 2 : 1);   // the resolution of the taps
  q.RS = 4;
  for (int d = 32; d >= 4; --d)
    if (H % d == 0) {
      q.RS = d;
      break;
    }
  q.segs_y = H / q.RS;
  q.strips = W / 32;
  q.nseg = g->N * q.strips * q.segs_y;
  const int npairs = (g->Cout / RW_T) * (g->Cin / RW_T);
  int want = (512 + npairs - 1) / npairs;          // two 4-wave pipelines (one 512-thread workgroup) per CU
  if (want > q.nseg) want = q.nseg;
  if (want < 1) want = 1;
  const int per = (q.nseg + want - 1) / want;      // segments per pipeline
  q.streams = (q.nseg + per - 1) / per;
  q.flush_slots = (q.streams + 1) / 2;
  return q;
}

__global__ void wgrad_bf16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, long long n, int slots,
                                         float scale) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < slots; ++k) s += ws[(long long)k * n + i];   // fixed order: deterministic
  gw[i] = s * scale;
}

// forward / input gradient: 8 x 32 pixel tiles, or 16 x 16 tiles for maps whose width is a multiple of 16 only; the
// nearest upsample in front (up) or the 2 x 2 average pool behind (pool) fold into the kernel (not both)
bool bf16_ok(const ganlab_conv_geom* g) {
  if (g == nullptr || g->ks != 3 || g->pad != 1 || (g->up && g->pool) || g->N <= 0 || g->Cin <= 0 || g->Cout <= 0 ||
      g->Cin % 64 != 0 || g->Cout % 64 != 0 || g->Hin <= 0 || g->Win <= 0)
    return false;
  const long long H = (long long)g->Hin * (g->up ? 2 : 1), W = (long long)g->Win * (g->up ? 2 : 1);   // where the taps run
  if (!((H % 8 == 0 && W % 32 == 0) || (H % 16 == 0 && W % 16 == 0))) return false;
  return (long long)g->Cin * H * W * 4 < (1LL << 31) && (long long)g->Cout * H * W * 4 < (1LL << 31);
}

// the weight-gradient kernels walk 32-pixel strips
// (with up / pool only the rolling kernel: it reads the half-resolution operand in place)
bool bf16_wgrad_ok(const ganlab_conv_geom* g) {
  if (!bf16_ok(g)) return false;
  const int m = g->up ? 2 : 1;
  return (g->Hin * m) % 8 == 0 && (g->Win * m) % 32 == 0 && (!(g->up || g->pool) || wr_enabled());
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

// mode 0: plain; 1: UP (x is H/2 x W/2); 2: POOL (y is H/2 x W/2).  H, W: the resolution of the 3x3 taps.
// Split factor of a forward / input-gradient launch at the taps' resolution H x W: only where the plain launch leaves most
// of the 256 CUs idle, at least four 32-channel chunks per workgroup.
static int bf16_splitk_plan(int N, int CI, int CO, int H, int W) {
  const bool wide = W % 32 == 0 && H % 8 == 0;
  const long long tiles = (long long)N * (wide ? (W / 32) * (H / 8) : (W / 16) * (H / 16)) * (CO / COT);
  const int chunks = CI / CK;
  if (tiles >= 128 || chunks < 8) return 1;
  int S = (int)((256 + tiles - 1) / tiles);
  if (S > 4) S = 4;
  while (S > 1 && chunks / S < 4) --S;
  return S < 2 ? 1 : S;
}

static int launch_fwd(const float* x, const void* wp, const float* bias, float* y, int N, int CI, int CO, int H, int W,
                      float bias_scale, int act, float slope, int mode, float oscale, void* stream, float* part = nullptr,
                      int ksplit = 1) {
  BfArgs a;
  a.ksplit = ksplit; a.part = part;
  a.ks_chunks = ksplit > 1 ? (CI / CK + ksplit - 1) / ksplit : 0;
  a.x = x; a.wp = reinterpret_cast<const __bf16*>(wp); a.bias = bias; a.y = y;
  a.N = N; a.CI = CI; a.CO = CO; a.H = H; a.W = W;
  const bool wide = W % 32 == 0 && H % 8 == 0;
  a.tiles_x = wide ? W / 32 : W / 16; a.tiles_y = wide ? H / 8 : H / 16; a.tiles_co = CO / COT;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act; a.oscale = oscale;
  const long long grid = (long long)N * a.tiles_x * a.tiles_y * a.tiles_co * (ksplit > 1 ? ksplit : 1);
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  const dim3 gd((unsigned)grid), bd(256);
  hipStream_t st = gl_stream(stream);
  if (wide) {
    if (mode == 1) GL_LAUNCH((conv_fwd_bf16_kernel<8, 32, true, false>), gd, bd, 0, st, a);
    else if (mode == 2) GL_LAUNCH((conv_fwd_bf16_kernel<8, 32, false, true>), gd, bd, 0, st, a);
    else GL_LAUNCH((conv_fwd_bf16_kernel<8, 32>), gd, bd, 0, st, a);
  } else {
    if (mode == 1) GL_LAUNCH((conv_fwd_bf16_kernel<16, 16, true, false>), gd, bd, 0, st, a);
    else if (mode == 2) GL_LAUNCH((conv_fwd_bf16_kernel<16, 16, false, true>), gd, bd, 0, st, a);
    else GL_LAUNCH((conv_fwd_bf16_kernel<16, 16>), gd, bd, 0, st, a);
  }
  if (ksplit > 1) {
    const int pool = mode == 2;
    const long long total = (long long)N * CO * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    GL_LAUNCH(bf16_splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, y,
              ksplit, N, CO, H, W, pool, oscale, bias_scale, act, slope);
  }
  return GL_CHECK_LAUNCH();
}

int ganlab_conv_fwd_bf16(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                         float bias_scale, int act, float slope, void* stream) {
  if (x == nullptr || wp == nullptr || y == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_ok(g)) return GANLAB_EUNSUPPORTED;
  const int m = g->up ? 2 : 1;        // conv(up2 x): UP; pool2(conv x): POOL with the 1/4 of the average
  return launch_fwd(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin * m, g->Win * m, bias_scale, act, slope,
                    g->up ? 1 : (g->pool ? 2 : 0), g->pool ? 0.25f : 1.f, stream);
}

/* split-K forms for layers with few output tiles (the 512-channel 16x16 layers at small batch): plan = number of K-splits
 * (1: use the plain entry point), workspace = plan * N * Cout_of_the_operator * H * W floats at the taps' resolution */
int ganlab_conv_bf16_splitk_plan(const ganlab_conv_geom* g, int dgrad) {
  if (g == nullptr || !bf16_ok(g)) return 1;
  const int m = g->up ? 2 : 1;
  const char* e = GL_ENV_ONCE("GANLAB_BF16_SPLITK");
  if (e != nullptr && e[0] == '0') return 1;
  return dgrad ? bf16_splitk_plan(g->N, g->Cout, g->Cin, g->Hin * m, g->Win * m)
               : bf16_splitk_plan(g->N, g->Cin, g->Cout, g->Hin * m, g->Win * m);
}

int ganlab_conv_fwd_bf16_splitk(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                                float bias_scale, int act, float slope, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (x == nullptr || wp == nullptr || y == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_ok(g)) return GANLAB_EUNSUPPORTED;
  const int S = ganlab_conv_bf16_splitk_plan(g, 0);
  const int m = g->up ? 2 : 1;
  if (S < 2) return ganlab_conv_fwd_bf16(x, wp, bias, y, g, bias_scale, act, slope, stream);
  if (workspace == nullptr || workspace_bytes < (size_t)S * g->N * g->Cout * g->Hin * m * g->Win * m * sizeof(float))
    return GANLAB_EWORKSPACE;
  return launch_fwd(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin * m, g->Win * m, bias_scale, act, slope,
                    g->up ? 1 : (g->pool ? 2 : 0), g->pool ? 0.25f : 1.f, stream, reinterpret_cast<float*>(workspace), S);
}

int ganlab_conv_dgrad_bf16_splitk(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (gy == nullptr || wp == nullptr || gx == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_ok(g)) return GANLAB_EUNSUPPORTED;
  const int S = ganlab_conv_bf16_splitk_plan(g, 1);
  const int m = g->up ? 2 : 1;
  if (S < 2) return ganlab_conv_dgrad_bf16(gy, wp, gx, g, stream);
  if (workspace == nullptr || workspace_bytes < (size_t)S * g->N * g->Cin * g->Hin * m * g->Win * m * sizeof(float))
    return GANLAB_EWORKSPACE;
  return launch_fwd(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin * m, g->Win * m, 0.f, GANLAB_ACT_NONE, 0.f,
                    g->up ? 2 : (g->pool ? 1 : 0), g->pool ? 0.25f : 1.f, stream, reinterpret_cast<float*>(workspace), S);
}

int ganlab_conv_dgrad_bf16(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream) {
  if (gy == nullptr || wp == nullptr || gx == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_ok(g)) return GANLAB_EUNSUPPORTED;
  // adjoints: of the nearest upsample the 2 x 2 sum (POOL, factor 1), of the average pool the upsample / 4 (UP)
  const int m = g->up ? 2 : 1;
  return launch_fwd(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin * m, g->Win * m, 0.f, GANLAB_ACT_NONE, 0.f,
                    g->up ? 2 : (g->pool ? 1 : 0), g->pool ? 0.25f : 1.f, stream);
}

size_t ganlab_conv_wgrad_bf16_workspace(const ganlab_conv_geom* g) {
  if (!bf16_wgrad_ok(g)) return 0;
  const int slots = wr_enabled() ? wr_plan(g).flush_slots : wgrad_slots(g);
  return (size_t)slots * g->Cout * g->Cin * 9 * sizeof(float);
}

int ganlab_conv_wgrad_bf16(const float* gy, const float* x, float* gw, const ganlab_conv_geom* g, float scale,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (gy == nullptr || x == nullptr || gw == nullptr || g == nullptr) return GANLAB_EINVAL;
  if (!bf16_wgrad_ok(g)) return GANLAB_EUNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < ganlab_conv_wgrad_bf16_workspace(g)) return GANLAB_EWORKSPACE;
  if (wr_enabled()) {
    const WrPlan q = wr_plan(g);
    BfWrArgs r;
    r.gy = gy; r.x = x; r.ws = reinterpret_cast<float*>(workspace);
    const int m = g->up ? 2 : 1;
    r.N = g->N; r.CI = g->Cin; r.CO = g->Cout; r.H = g->Hin * m; r.W = g->Win * m;
    r.tiles_ci = r.CI / RW_T; r.npairs = r.tiles_ci * (r.CO / RW_T);
    r.RS = q.RS; r.segs_y = q.segs_y; r.strips = q.strips; r.nseg = q.nseg; r.streams = q.streams;
    const dim3 gd((unsigned)((long long)r.npairs * q.flush_slots)), bd(512);
    // conv(up2 x): x is stored at half the taps' resolution; pool2(conv x): gy is, and its adjoint carries 1/4
    if (g->up) GL_LAUNCH((conv_wgrad_bf16_roll_kernel<true, false>), gd, bd, 0, gl_stream(stream), r);
    else if (g->pool) GL_LAUNCH((conv_wgrad_bf16_roll_kernel<false, true>), gd, bd, 0, gl_stream(stream), r);
    else GL_LAUNCH((conv_wgrad_bf16_roll_kernel<false, false>), gd, bd, 0, gl_stream(stream), r);
    const int n4 = r.npairs * 4 * 36 * 64;
    GL_LAUNCH(wgrad_bf16_roll_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, gl_stream(stream),
              reinterpret_cast<const f32x4*>(r.ws), gw, r.npairs, r.tiles_ci, r.CI, q.flush_slots,
              g->pool ? scale * 0.25f : scale);
    return GL_CHECK_LAUNCH();
  }
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
