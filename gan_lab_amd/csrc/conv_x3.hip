// fp32 3x3 convolution on the bf16 matrix cores: every operand is split into THREE bf16 planes, x = h + m + l exactly
// (round-to-nearest at each cut: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); 8 + 8 + 8 significand bits with the
// signs carrying the 25th), and a product a*b is the SIX bf16 products  ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh);
// the three dropped ones (am*bl, al*bm, al*bl) are <= 2^-26 |a b|.  v_mfma_f32_16x16x32_bf16 multiplies bf16 pairs exactly
// and accumulates in fp32 at 16x the rate of v_mfma_f32_16x16x4_f32, so six of them per fp32 product leave 2.67x.
//
// Same math as csrc/conv.hip: F.conv2d(x * wscale, W, padding=1) of utils/custom_layers.py:202-211 and its input gradient.
// What decides the accuracy is not the products but the ACCUMULATION (tools/x3_probe.hip, profiles/r05_x3_probe.txt):
// adding the five small products into the accumulator that holds the hi*hi sum rounds it six times per k-step instead of
// once (2.6x the error), so they go to their own accumulator set S (magnitude 2^-8 of the result: its roundings do not
// count) and only hi*hi goes to H.  H is ONE rounding per 32 terms; a chain of 72 of them (256 input channels) is 1.3x
// further from float64 than ATen's blocked sums, so H is a chain over ONE 32-channel chunk (9 k-steps) that is then added
// to a third set T: 96 accumulator registers for a 64 x 32 tile per wave.
//
// GEMM view: D[px][co] += sum_k A[px][k] * B[k][co]; A = activation patch (MFMA A operand: lane l holds 8 consecutive
// k of pixel l & 15), B = packed weights, D: lane holds 4 consecutive pixels of one output channel -> 16-byte NCHW stores.
// Workgroup: 512 threads (8 waves, 4 along the pixels x 2 along the channels), tile = 16 x 16 pixels of one image x 64
// output channels, 64 x 32 per wave.
// k-step = one MFMA depth (32): lane groups 0,1 take (tap a, 16 input channels), groups 2,3 (tap b, 16 channels), so the
// activation patch is staged in HALVES of 16 channels (31 KB for the three planes of an 18 x 18 halo patch) and three
// half slots ring through LDS: 9 k-steps per 32 channels,
//   steps 0-3: taps (0,1) (2,3) (4,5) (6,7) of half 0 | step 4: tap 8 of half 0 and of half 1 | steps 5-8: half 1.
// The weights of TWO k-steps (a "stage", 24 KB) are double buffered; one barrier per stage (96 MFMAs per wave).
// Staging is register-staged (buffer loads -> split -> ds_write), loads issued one to two stages ahead of their ds_write.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int X3_NT = 64;                      // output channels per workgroup
constexpr int X3_PL = 336;                     // units (16 B) per (channel group, plane) of a half patch: 18 x 18 = 324, padded
                                               // to a multiple of 16 units so the k-groups stay 256 bytes apart (no conflicts)
constexpr int X3_HALF = 6 * X3_PL;             // 2 channel groups x 3 planes
constexpr int X3_WSTEP = 3 * 4 * X3_NT;        // units of one k-step of weights: [plane][k-group][co]
constexpr int X3_AOFF = 0, X3_WOFF = 3 * X3_HALF;
constexpr int X3_WSTAGE = 2 * X3_WSTEP;         // two k-steps per barrier
constexpr int X3_LDS = 3 * X3_HALF + 2 * X3_WSTAGE;  // 9120 units = 145,920 bytes

__device__ __forceinline__ float x3_up(__bf16 b) { return (float)b; }

// (tap, half) of lane groups 0,1 and 2,3 at step s of a 32-channel chunk
__host__ __device__ constexpr int x3_tap_lo(int s) { return s < 4 ? 2 * s : s == 4 ? 8 : 2 * (s - 5); }
__host__ __device__ constexpr int x3_tap_hi(int s) { return s < 4 ? 2 * s + 1 : s == 4 ? 8 : 2 * (s - 5) + 1; }
__host__ __device__ constexpr int x3_half_lo(int s) { return s <= 4 ? 0 : 1; }
__host__ __device__ constexpr int x3_half_hi(int s) { return s < 4 ? 0 : 1; }

// ---- weight packing: OIHW fp32 -> [co tile][k-step][plane][k-group][co 64][8] bf16 ------------------------------------
__global__ void x3_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int mode, float scale) {
  const int CO = mode == GANLAB_PACK_DGRAD ? Cin : Cout;     // GEMM roles
  const int CI = mode == GANLAB_PACK_DGRAD ? Cout : Cin;
  const long long total = 9LL * CO * CI;                      // elements per plane
  const int steps = CI / 32 * 9;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7);
    long long t = e >> 3;
    const int col = (int)(t % X3_NT); t /= X3_NT;
    const int kg = (int)(t & 3); t >>= 2;
    const int ks = (int)(t % steps);
    const int ct = (int)(t / steps);
    const int c = ks / 9, s = ks % 9;
    const int tap = (kg >> 1) ? x3_tap_hi(s) : x3_tap_lo(s);
    const int half = (kg >> 1) ? x3_half_hi(s) : x3_half_lo(s);
    const int ci = c * 32 + half * 16 + (kg & 1) * 8 + j;
    const int co = ct * X3_NT + col;
    float v = mode == GANLAB_PACK_DGRAD ? w[((long long)ci * Cin + co) * 9 + (8 - tap)] : w[((long long)co * Cin + ci) * 9 + tap];
    v *= scale;
    const __bf16 h = (__bf16)v;
    const float r1 = v - x3_up(h);
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - x3_up(m));
    // unit index within the k-step image: (plane * 4 + kg) * 64 + col
    __bf16* base = out + ((long long)ct * steps + ks) * X3_WSTEP * 8;
    base[((0 * 4 + kg) * X3_NT + col) * 8 + j] = h;
    base[((1 * 4 + kg) * X3_NT + col) * 8 + j] = m;
    base[((2 * 4 + kg) * X3_NT + col) * 8 + j] = l;
  }
}

struct X3Args {
  const float* x;
  const u32x4* wp;
  const float* bias;
  float* y;
  int N, CI, CO, H, W;
  int tiles_x, tiles_y, tiles_co;
  float bias_scale, slope;
  int act;
};

// tap offset (units) inside the halo patch
__host__ __device__ constexpr int x3_toff(int tap) { return (tap / 3) * 18 + tap % 3; }

__global__ __launch_bounds__(512) void conv_x3_fwd_kernel(X3Args p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[X3_LDS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;          // pixel rows 4wm .. 4wm+3 of the tile, channels 32wn .. 32wn+31
  const int l16 = lane & 15, kg = lane >> 4;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.tiles_co; bid /= p.tiles_co;
  const int txi = bid % p.tiles_x; bid /= p.tiles_x;
  const int tyi = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int oy0 = tyi * 16, ox0 = txi * 16, co0 = co_t * X3_NT;
  const int plane = p.H * p.W;
  const int nstages = p.CI / 64 * 9, nhalves = p.CI / 16;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (long long)n * p.CI * plane), 0, (unsigned)((long long)p.CI * plane * 4), 0x00020000);
  const u32x4* wsrc = p.wp + (long long)co_t * nstages * X3_WSTAGE;

  // ---- activation staging item of this thread: (channel group g, row r, 4-column group cg, channel quad cq) ----------
  // columns ox0 - 4 + 4cg .. + 3; halo column of element i = 4cg - 3 + i (valid 0 .. 17: cg 0 keeps i = 3, cg 5 keeps i = 0)
  const bool a_item = tid < 432;
  int a_goff, a_unit, a_i0, a_i1;
  {
    const int e = a_item ? tid : 0;
    const int cq = e & 1;
    int t = e >> 1;
    const int cg = t % 6; t /= 6;
    const int r = t % 18, g = t / 18;
    const int vy = oy0 - 1 + r, vx = ox0 - 4 + 4 * cg;
    const bool ok = a_item && (unsigned)vy < (unsigned)p.H && (unsigned)vx < (unsigned)p.W;
    a_goff = ok ? ((g * 8 + cq * 4) * plane + vy * p.W + vx) * 4 : (int)0x80000000;
    a_unit = ((g * 3) * X3_PL + r * 18 + 4 * cg - 3) * 16 + cq * 8;     // byte offset of element 0, plane 0
    a_i0 = cg == 0 ? 3 : 0;
    a_i1 = cg == 5 ? 1 : 4;
  }
  const int cstride = plane * 4;
  float4 ar[4];
  auto a_load = [&](int half) {     // channels 16 half + 8g + 4cq + j
    const int soff = half * 16 * plane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_goff + j * cstride, soff, 0);
      ar[j] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto a_store = [&](int slot) {
    if (!a_item) return;
    unsigned char* dst = reinterpret_cast<unsigned char*>(lds + X3_AOFF + slot * X3_HALF) + a_unit;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < a_i0 || i >= a_i1) continue;
      bf16x4 h, m, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = i == 0 ? ar[j].x : i == 1 ? ar[j].y : i == 2 ? ar[j].z : ar[j].w;
        h[j] = (__bf16)v;
        const float r1 = v - x3_up(h[j]);
        m[j] = (__bf16)r1;
        l[j] = (__bf16)(r1 - x3_up(m[j]));
      }
      *reinterpret_cast<u32x2*>(dst + i * 16) = __builtin_bit_cast(u32x2, h);
      *reinterpret_cast<u32x2*>(dst + i * 16 + X3_PL * 16) = __builtin_bit_cast(u32x2, m);
      *reinterpret_cast<u32x2*>(dst + i * 16 + 2 * X3_PL * 16) = __builtin_bit_cast(u32x2, l);
    }
  };

  // ---- weight staging: 1536 units per stage (two k-steps), 3 per thread ---------------------------------------------
  u32x4 wr[3];
  auto w_load = [&](int st) {
    const u32x4* s = wsrc + (long long)st * X3_WSTAGE;
#pragma unroll
    for (int i = 0; i < 3; ++i) wr[i] = s[tid + i * 512];
  };
  auto w_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) lds[X3_WOFF + buf * X3_WSTAGE + tid + i * 512] = wr[i];
  };

  f32x4 accS[4][2], accH[4][2], accT[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
      accS[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; accT[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  const int laneA = X3_AOFF + (kg & 1) * 3 * X3_PL + (4 * wm) * 18 + l16;
  const int laneB = X3_WOFF + kg * X3_NT + wn * 32 + l16;
  const bool khi = kg >= 2;

  bf16x8 aF[3][4], bF[2][3];
  auto a_frags = [&](int off, int m) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) aF[pl][m] = __builtin_bit_cast(bf16x8, lds[off + pl * X3_PL + m * 18]);
  };
  auto b_frags = [&](int buf, int step1, int nn, int set) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      bF[set][pl] = __builtin_bit_cast(bf16x8, lds[laneB + buf * X3_WSTAGE + step1 * X3_WSTEP + pl * 4 * X3_NT + nn * 16]);
  };

  // ---- prologue: halves 0, 1, weights of stage 0; loads of stage 1 in flight -----------------------------------------
  a_load(0); a_store(0);
  a_load(1); a_store(1);
  w_load(0); w_store(0);
  if (nstages > 1) w_load(1);
  __syncthreads();
  {
    const int off0 = laneA + (khi ? x3_toff(x3_tap_hi(0)) : x3_toff(x3_tap_lo(0)));
#pragma unroll
    for (int m = 0; m < 4; ++m) a_frags(off0, m);
    b_frags(0, 0, 0, 0);
  }

  int st = 0;                       // global stage index
  const int ndc = p.CI / 64;
  int r0 = 0;                       // ring slot of this double chunk's first half: (4 dc) % 3
  for (int dc = 0; dc < ndc; ++dc) {
    // ring slots of the halves 4dc + 0 .. 5
    const int sl0 = r0, sl1 = r0 == 2 ? 0 : r0 + 1, sl2 = sl1 == 2 ? 0 : sl1 + 1;
    const bool more = dc + 1 < ndc;
#pragma unroll
    for (int j = 0; j < 9; ++j, ++st) {
      const int buf = st & 1;
      // stores at the top of the stage (their loads were issued one or two stages ago)
      if (st + 1 < nstages) w_store(buf ^ 1);
      if (j == 0 && dc > 0) a_store(sl1);            // half 4dc + 1
      if (j == 2) a_store(sl2);                      // half 4dc + 2
      if (j == 4) a_store(sl0);                      // half 4dc + 3
      if (j == 7 && more) a_store(sl1);              // half 4dc + 4
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int s18 = 2 * j + (sub >> 1);          // k-step within the double chunk
        const int nn = sub & 1;
        // operand prefetch for the next sub-block
        if (sub == 0) b_frags(buf, 0, 1, 1);
        if (sub == 1) b_frags(buf, 1, 0, 0);
        if (sub == 2) b_frags(buf, 1, 1, 1);
        if (sub == 3) {
          __syncthreads();           // the next stage's weights (and any half patch stored at the top) are visible
          if (st + 2 < nstages) w_load(st + 2);
          if (j == 0) a_load(4 * dc + 2);
          if (j == 2) a_load(4 * dc + 3);
          if (j == 5 && more) a_load(4 * dc + 4);
          if (j == 7 && more) a_load(4 * dc + 5);
          if (st + 1 < nstages) b_frags(buf ^ 1, 0, 0, 0);
        }
        const int set = nn;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          f32x4 sacc = accS[m][nn];
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[2][m], bF[set][0], sacc, 0, 0, 0);
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[0][m], bF[set][2], sacc, 0, 0, 0);
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[1][m], bF[set][1], sacc, 0, 0, 0);
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[1][m], bF[set][0], sacc, 0, 0, 0);
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[0][m], bF[set][1], sacc, 0, 0, 0);
          accS[m][nn] = sacc;
          accH[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aF[0][m], bF[set][0], accH[m][nn], 0, 0, 0);
          if (nn == 1 && (s18 < 17 || more)) {        // this tile row's fragments of the next k-step
            const int s1 = s18 == 17 ? 0 : s18 + 1;   // next k-step within its double chunk
            const int c1 = s1 / 9, s9 = s1 % 9;       // chunk of the pair, step within the chunk
            const int hl = 2 * c1 + x3_half_lo(s9), hh = 2 * c1 + x3_half_hi(s9);     // halves 0 .. 3 of that double chunk
            // ring slots: within this double chunk half k sits in slot (r0 + k) % 3; the next one starts at (r0 + 4) % 3 = sl1
            const int b0 = s18 == 17 ? 1 : 0;
            const int kl = (hl + b0) % 3, kh = (hh + b0) % 3;
            const int slot_l = kl == 0 ? sl0 : kl == 1 ? sl1 : sl2, slot_h = kh == 0 ? sl0 : kh == 1 ? sl1 : sl2;
            const int offn = laneA + (khi ? slot_h * X3_HALF + x3_toff(x3_tap_hi(s9)) : slot_l * X3_HALF + x3_toff(x3_tap_lo(s9)));
            a_frags(offn, m);
          }
        }
        if (nn == 1 && (s18 == 8 || s18 == 17)) {     // a 32-channel chunk is done: close its hi*hi chain
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) { accT[m][q] += accH[m][q]; accH[m][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    r0 = sl1;      // (r0 + 4) % 3
  }

  // ---- epilogue: T + S + bias, activation; lane = 4 consecutive pixels of one channel ---------------------------------
  float* yb = p.y + (long long)n * p.CO * plane;
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    const int co = co0 + wn * 32 + nn * 16 + l16;
    const float bv = p.bias != nullptr ? p.bias[co] * p.bias_scale : 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long long o = (long long)co * plane + (oy0 + 4 * wm + m) * p.W + ox0 + 4 * kg;
      f32x4 v = accT[m][nn] + accS[m][nn];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float f = v[r] + bv;
        if (p.act == GANLAB_ACT_LRELU) f = gl_lrelu(f, p.slope);
        v[r] = f;
      }
      *reinterpret_cast<f32x4*>(yb + o) = v;
    }
  }
}

bool x3_ok(const ganlab_conv_geom* g, int dgrad) {
  if (g == nullptr || g->ks != 3 || g->pad != 1 || g->up || g->pool) return false;
  const int CI = dgrad ? g->Cout : g->Cin, CO = dgrad ? g->Cin : g->Cout;
  return CI % 64 == 0 && CO % X3_NT == 0 && g->Hin % 16 == 0 && g->Win % 16 == 0 && g->N > 0;
}

int x3_launch(const float* x, const void* wp, const float* bias, float* y, int N, int CI, int CO, int H, int W, float bias_scale,
              int act, float slope, hipStream_t st) {
  X3Args a;
  a.x = x; a.wp = reinterpret_cast<const u32x4*>(wp); a.bias = bias; a.y = y;
  a.N = N; a.CI = CI; a.CO = CO; a.H = H; a.W = W;
  a.tiles_x = W / 16; a.tiles_y = H / 16; a.tiles_co = CO / X3_NT;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  const long long grid = (long long)N * a.tiles_x * a.tiles_y * a.tiles_co;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  GL_LAUNCH(conv_x3_fwd_kernel, dim3((unsigned)grid), dim3(512), 0, st, a);
  return GL_CHECK_LAUNCH();
}

}  // namespace

extern "C" {

int ganlab_conv_x3_supported(const ganlab_conv_geom* g, int dgrad) { return x3_ok(g, dgrad) ? 1 : 0; }

long long ganlab_conv_x3_pack(const float* w, void* out, int Cout, int Cin, int mode, float scale, void* stream) {
  if (Cout <= 0 || Cin <= 0 || (mode != GANLAB_PACK_FWD && mode != GANLAB_PACK_DGRAD)) return GANLAB_EINVAL;
  const int CO = mode == GANLAB_PACK_DGRAD ? Cin : Cout, CI = mode == GANLAB_PACK_DGRAD ? Cout : Cin;
  if (CO % X3_NT != 0 || CI % 64 != 0) return GANLAB_EINVAL;
  const long long n = 3LL * 9 * Cout * Cin;       // bf16 elements
  if (out == nullptr) return n;
  if (w == nullptr) return GANLAB_EINVAL;
  const long long per_plane = n / 3;
  const int blocks = (int)((per_plane + 255) / 256 < 4096 ? (per_plane + 255) / 256 : 4096);
  GL_LAUNCH(x3_pack_kernel, dim3(blocks), dim3(256), 0, gl_stream(stream), w, reinterpret_cast<__bf16*>(out), Cout, Cin, mode,
            scale);
  const int st = GL_CHECK_LAUNCH();
  return st != GANLAB_OK ? st : n;
}

int ganlab_conv_fwd_x3(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g, float bias_scale,
                       int act, float slope, void* stream) {
  if (!x3_ok(g, 0) || x == nullptr || wp == nullptr || y == nullptr) return GANLAB_EINVAL;
  return x3_launch(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin, g->Win, bias_scale, act, slope, gl_stream(stream));
}

int ganlab_conv_dgrad_x3(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream) {
  if (!x3_ok(g, 1) || gy == nullptr || wp == nullptr || gx == nullptr) return GANLAB_EINVAL;
  return x3_launch(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin, g->Win, 1.f, GANLAB_ACT_NONE, 0.f, gl_stream(stream));
}

}  // extern "C"
