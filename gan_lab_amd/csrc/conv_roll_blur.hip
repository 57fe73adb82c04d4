// conv 3x3 -> +bias -> LeakyReLU -> binomial blur in ONE kernel, for the thinnest layer of the critic
// (conv -> LeakyReLU -> blur -> pooled conv, gan_lab/progan/architectures.py:254-284 at the top resolution: 16 -> 16
// channels at 1024^2): the blur pass of the composed form (csrc/pointwise.hip blur3x3_vec_kernel: 1R + 1W of the
// widest tensor of the network) is folded into the rolling-window convolution, which also emits the sign bits of the
// activation (the only thing the backward needs of it: ops._ConvBiasAct).
//
// Same skeleton as conv.hip's conv_fwd_roll_kernel - a workgroup owns a 64-pixel column strip and walks DOWN it four
// rows per step over a six-slot LDS ring of input rows, weights in registers - with the wave-to-output map turned by
// 90 degrees: wave w owns the 16-pixel column BLOCK w for all four rows of the step (not one row across four blocks):
//   * the vertical half of the blur stays in registers: a lane holds the same four pixels of its channel in rows
//     4s .. 4s+3, plus the two (horizontally blurred) rows carried over from the previous step; the output lags the
//     convolution by one row, and a row strip computes ONE extra step for the rows its neighbours own (v rows Y0-1 and
//     Y0+4n: +1/32 of the MFMA work at the benchmark size);
//   * the horizontal half needs one pixel from the left / right neighbour: the next lane group (ds_bpermute), the next
//     wave (2 KB exchange through LDS at the step's first barrier), or - at the strip's own edges - the conv output of
//     the two columns just outside the strip.  Those 8 pixels per step are a fifth MFMA block whose 36 K-steps are split
//     among the four waves by channel group (9 MFMAs each, +6 %); the partial sums meet in the same LDS exchange;
//   * an input row feeds up to three (output row, ky) pairs of the SAME wave: 72 LDS operand reads per step and wave
//     instead of 144.
// Zero padding of the blur: activation rows / columns outside the image are zero (not "conv of the padding").
#include "common.h"

#include <stdlib.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int RB_TW = 64, RB_SLOTS = 6, RB_RP = 80, RB_SLOT = 16 * RB_RP, RB_Q = 18;
constexpr int RB_ITEMS = 4 * 16 * RB_Q;             // float4 items of one 4-row prefetch: 1152
constexpr int RB_PT = (RB_ITEMS + 255) / 256;       // 5
constexpr int RB_OOB = (int)0x80000000;
// cache policy of the output stores / input loads (aux 2 = nt): measured in round 4 on 16 -> 16 at 1024^2 x 32 - nt stores
// 1.318 against 1.31 ms, nt loads 1.44 ms (the column-halo re-reads then miss the L2): both stay at the default policy
#ifndef RB_STORE_AUX
#define RB_STORE_AUX 0
#endif
#ifndef RB_LOAD_AUX
#define RB_LOAD_AUX 0
#endif
#ifndef RB_SETPRIO
#define RB_SETPRIO 0
#endif

struct RBArgs {
  const float* x;
  const float* wp;          // PACK_FWD: [9 taps][Cin_p][Cout_p]
  const float* bias;
  float* y;
  unsigned short* bits;     // sign bits of the activation, 16 pixels per halfword (bit e of the NCHW-linear index); or null
  int N, Cin, Cout, H, W;
  int Cin_p, Cout_p;
  int cols, strips, spu;    // 64-pixel columns per row, row strips per column, steps (4 rows) per strip
  float bias_scale, slope;
  int act;                  // plain form only (the blurred form is LeakyReLU by definition)
  // MODE RB_RGB (the kernel as the INPUT GRADIENT of the critic's first 3x3 conv): nothing is stored; the gradient meets
  // the layer in front - fromRGB, a 1x1 conv of the <= 4-channel image + LeakyReLU (progan/architectures.py:232-237) - right
  // here: gz = result * lrelu'(mbits), part[(co*4 + c)][wg] = sum gz * img[c] (c < 3: fromRGB's weight gradient), c = 3: sum gz
  const float* img;          // (N, img_c, H, W)
  const unsigned char* mbits;   // sign bits of fromRGB's output, bit e of the NCHW-linear index
  double* part;              // [Cout][4][grid]
  int img_c;
};
enum { RB_PLAIN = 0, RB_BLUR = 1, RB_RGB = 2 };

__device__ __forceinline__ float rb_act(float v, float slope) { return v > 0.f ? v : v * slope; }

#ifdef GL_PHASES   // tools/phase_probe_rb.py (debug builds only): accumulated time of the step's phases, wave 0 of each workgroup
__device__ unsigned long long* rb_phase_buf;
#define RB_PH_DECL unsigned long long rb_t = wall_clock64(), rb_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RB_PH(i) { const unsigned long long rb_n = wall_clock64(); rb_ph[i] += rb_n - rb_t; rb_t = rb_n; }
#define RB_PH_FLUSH(nsteps) if (threadIdx.x == 0) { for (int i_ = 0; i_ < 10; ++i_) rb_phase_buf[(long long)blockIdx.x * 12 + i_] = rb_ph[i_]; \
    rb_phase_buf[(long long)blockIdx.x * 12 + 10] = (nsteps); rb_phase_buf[(long long)blockIdx.x * 12 + 11] = wall_clock64(); }
#else
#define RB_PH_DECL
#define RB_PH(i)
#define RB_PH_FLUSH(nsteps)
#endif

// BLUR = false: the same wave-owns-a-column-block MFMA phase with a plain epilogue (+bias, activation, store) - the rolling
// 3x3 kernel with half the LDS operand reads of conv.hip's conv_fwd_roll_kernel (wave owns a row).
template <int MODE>
__global__ __launch_bounds__(256, MODE == RB_PLAIN ? 4 : 3) void conv_fwd_roll_blur_kernel(RBArgs p) {
  constexpr bool BLUR = MODE == RB_BLUR, RGB = MODE == RB_RGB;
  constexpr int RO = BLUR ? 1 : 0;       // the blurred form computes activation rows Y0 - 1 .. (one extra step)
  __shared__ __attribute__((aligned(16))) float img_s[RGB ? 4 * 4 * 64 : 4];   // [c][row of the step][64 pixels]
  __shared__ double red_s[RGB ? 4 * 16 * 4 : 1];
  __shared__ __attribute__((aligned(16))) float ring[RB_SLOTS * RB_SLOT];       // 30720 B
  __shared__ __attribute__((aligned(16))) float xn[2 * 4 * 16 * 4];             // [side][wave][co][row]: own edge pixels
  __shared__ __attribute__((aligned(16))) float xe[2 * 4 * 16 * 4];             // [side][wave][co][row]: outside columns, partial
  // sign bits of a step, [row of the step][co][8 bytes = the strip's 64 pixels]: the waves drop their bytes here and wave 0
  // writes the step's 64 x 8 bytes with ONE store at the top of the next step (four 2-byte stores per wave and step before:
  // 2.1 of the 10.5 us of a step, tools/phase_probe_rb.py)
  __shared__ __attribute__((aligned(16))) unsigned bits_s[BLUR ? 16 * 8 : 4];      // [co][8 byte positions of the strip] x 4 row bytes
  __shared__ __attribute__((aligned(16))) float scratch_s[BLUR ? 256 * 4 : 4];     // one 16-byte slot per thread for writes that do not apply
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int co = lane & 15, kk = lane >> 4;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int txi = bid % p.cols;
  bid /= p.cols;
  const int syi = bid % p.strips;
  const int n0 = bid / p.strips;
  const int ns = min(p.spu, p.H / 4 - syi * p.spu);       // steps that OWN output rows; the blurred form runs ns + 1
  const int nrun = BLUR ? ns + 1 : ns;
  const int ox0 = txi * RB_TW, Y0 = syi * p.spu * 4;
  const int plane = p.H * p.W;
  const float* xb = p.x + (long long)n0 * p.Cin * plane;

  // staging items of a 4-row group: (k = row in group, ci, q = float4 column)
  int gbase[RB_PT], lo_k[RB_PT], lo_off[RB_PT];
#pragma unroll
  for (int i = 0; i < RB_PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % RB_Q, t = e / RB_Q;
    const int ci = t & 15, k = t >> 4;
    const int vx = ox0 - 4 + 4 * q;
    gbase[i] = (e < RB_ITEMS && ci < p.Cin && (unsigned)vx < (unsigned)p.W) ? (ci * plane + vx) * 4 : RB_OOB;
    lo_k[i] = k;
    lo_off[i] = ci * RB_RP + 4 * q;
  }
  // weights -> registers (k-step st = tap * 4 + c4)
  float wreg[36];
#pragma unroll
  for (int st = 0; st < 36; ++st)
    wreg[st] = p.wp[(long long)((st >> 2) * p.Cin_p + (st & 3) * 4 + kk) * p.Cout_p + co];
  // the outside columns' K-steps of this wave: all nine taps of channel group w
  float wedge[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) wedge[tap] = p.wp[(long long)(tap * p.Cin_p + w * 4 + kk) * p.Cout_p + co];
  const bool co_ok = co < p.Cout;
  const float bv = (p.bias != nullptr && co_ok) ? p.bias[co] * p.bias_scale : 0.f;

  const long long oplane = (long long)p.H * p.W;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n0 * p.Cout * oplane, 0, (unsigned)(p.Cout * oplane * 4), 0x00020000);
  const int vo_lane = co_ok ? (int)(((long long)co * oplane + ox0 + w * 16 + kk * 4) * 4) : RB_OOB;

  float4 xr[RB_PT];
  // rel row r of the strip = input row Y0 - RO - 1 + r; rows outside the image and k >= nrows read as zeros
  auto load_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < RB_PT; ++i) {
      const int vy = Y0 - RO - 1 + rel0 + lo_k[i];
      const bool ok = gbase[i] != RB_OOB && lo_k[i] < nrows && (unsigned)vy < (unsigned)p.H;
      const int off = ok ? gbase[i] + (int)((unsigned)vy * (unsigned)(p.W * 4)) : RB_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, RB_LOAD_AUX);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto store_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < RB_PT; ++i)
      if (tid + i * 256 < RB_ITEMS && lo_k[i] < nrows)
        *reinterpret_cast<float4*>(ring + ((rel0 + lo_k[i]) % RB_SLOTS) * RB_SLOT + lo_off[i]) = xr[i];
  };

  f32x4 acc[4], acce;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) acc[r4] = f32x4{0.f, 0.f, 0.f, 0.f};
  acce = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;       // horizontally blurred rows carried from the previous step
  // RGB: this thread's share of the image rows of a step (tid < 64 * img_c: channel tid / 64, row (tid / 16) % 4, float4
  // tid % 16), the per-channel sums of this lane's output channel `co`
  [[maybe_unused]] float4 imgr = float4{0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] double rsum[4] = {0.0, 0.0, 0.0, 0.0};
  [[maybe_unused]] const int img_item = (tid < 64 * p.img_c) ? ((tid >> 6) * plane + ((tid >> 4) & 3) * p.W + ox0 + 4 * (tid & 15)) : -1;

  // A-operand addressing.  Main blocks: pixel = lane & 15 of column block w, channel 4*c4 + kk.
  const int a_lane = kk * RB_RP + w * 16 + co + 3;     // + LP(4) - pad(1); `co` doubles as the pixel index here
  // Outside columns (fifth block): pixel e = lane & 7: row e & 3 of the step, column -1 (e < 4) or 64; this wave
  // contracts channel group c4 = w of it
  const int e_row = lane & 3, e_side = (lane >> 2) & 1;
  const int e_lane = (w * 4 + kk) * RB_RP + (e_side ? 64 : -1) + 3;

  [[maybe_unused]] auto flush_bits = [&](int sp) {       // wave 0: lane = (row of step sp) * 16 + co
    if constexpr (BLUR) {
      if (p.bits != nullptr && w == 0) {
        const int r4 = lane >> 4, row = Y0 - 1 + 4 * sp + r4;
        const uint4 lo = *reinterpret_cast<const uint4*>(bits_s + co * 8), hi = *reinterpret_cast<const uint4*>(bits_s + co * 8 + 4);
        const int sh = 8 * r4;
        uint2 o;
        o.x = ((lo.x >> sh) & 0xffu) | (((lo.y >> sh) & 0xffu) << 8) | (((lo.z >> sh) & 0xffu) << 16) | (((lo.w >> sh) & 0xffu) << 24);
        o.y = ((hi.x >> sh) & 0xffu) | (((hi.y >> sh) & 0xffu) << 8) | (((hi.z >> sh) & 0xffu) << 16) | (((hi.w >> sh) & 0xffu) << 24);
        if (co_ok && row >= Y0 && row < Y0 + 4 * ns)
          *reinterpret_cast<uint2*>(p.bits + ((((long long)(n0 * p.Cout + co) * p.H + row) * p.W + ox0) >> 4)) = o;
      }
    }
  };
  RB_PH_DECL
  load_rows(0, 4);
  store_rows(0, 4);
  load_rows(4, 2);
  store_rows(4, 2);
  __syncthreads();
  RB_PH(0)

  for (int s = 0; s < nrun; ++s) {
    if (s > 0) flush_bits(s - 1);
    [[maybe_unused]] unsigned mb[4] = {0xffu, 0xffu, 0xffu, 0xffu};
    if constexpr (RGB) {      // the tail's memory operands, requested before the MFMA phase
      if (img_item >= 0)
        imgr = *reinterpret_cast<const float4*>(p.img + (long long)n0 * p.img_c * plane + img_item + (Y0 + 4 * s) * p.W);
      if (co_ok) {
        const unsigned char* bp = p.mbits + ((((long long)n0 * p.Cout + co) * p.H + Y0 + 4 * s) * p.W + ox0 + w * 16 + kk * 4) / 8;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) mb[r4] = bp[r4 * (p.W >> 3)] >> (4 * (kk & 1));
      }
    }
    // ---- MFMA phase: v rows Y0 - RO + 4s + r4 (r4 < 4) of column block w from rel input rows 4s .. 4s+5 ----
    int sb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) sb[j] = ((4 * s + j) % RB_SLOTS) * RB_SLOT;
    int eb[3];
    {
      const int s6 = (4 * s) % RB_SLOTS;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int e = s6 + e_row + ky;
        e = e >= RB_SLOTS ? e - RB_SLOTS : e;
        e = e >= RB_SLOTS ? e - RB_SLOTS : e;
        eb[ky] = e * RB_SLOT + e_lane;
      }
    }
    // fetch order: input rows (0, 5) interleaved (one accumulator each), then 1, 4, 2, 3; 12 (kx, c4) pairs per row
    constexpr int JORD[6] = {0, 5, 1, 4, 2, 3};
    constexpr int PD = 3;
    float rb[PD + 1], ae = 0.f;
    auto fetch = [&](int f, int slot) {
      // f < 24: rows 0 / 5 alternate; else rows in JORD order, 12 fetches each
      const int j = f < 24 ? JORD[f & 1] : JORD[2 + (f - 24) / 12];
      const int i12 = f < 24 ? (f >> 1) : (f - 24) % 12;
      const int kx = i12 >> 2, c4 = i12 & 3;
      rb[slot] = ring[sb[j] + c4 * 4 * RB_RP + kx + a_lane];
    };
#if RB_SETPRIO
    __builtin_amdgcn_s_setprio(RB_SETPRIO);     // the MFMA phase issues ahead of the other workgroups' epilogues on this SIMD
#endif
#pragma unroll
    for (int f = 0; f < PD; ++f) fetch(f, f % (PD + 1));
#pragma unroll
    for (int f = 0; f < 72; ++f) {
      if (f + PD < 72) fetch(f + PD, (f + PD) % (PD + 1));
      const int slot = f % (PD + 1);
      const int j = f < 24 ? JORD[f & 1] : JORD[2 + (f - 24) / 12];
      const int i12 = f < 24 ? (f >> 1) : (f - 24) % 12;
      const int kx = i12 >> 2, c4 = i12 & 3;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int ky = j - r4;
        if (ky >= 0 && ky < 3)
          acc[r4] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot], wreg[(ky * 3 + kx) * 4 + c4], acc[r4], 0, 0, 0);
      }
      if constexpr (BLUR) {
        if ((f & 7) == 3) ae = ring[eb[(f >> 3) / 3] + (f >> 3) % 3];     // operand of the outside columns' K-step f / 8
        if ((f & 7) == 7)          // this wave's share of the fifth block: tap f / 8 of channel group w
          acce = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, wedge[f >> 3], acce, 0, 0, 0);
      }
      if (f == 8) load_rows(4 * s + 6, s + 1 < nrun ? 4 : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if RB_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    RB_PH(1)
    if constexpr (RGB) {
      __syncthreads();          // rows 4s .. 4s+3 of the ring are no longer read; the previous step's image rows neither
      store_rows(4 * s + 6, s + 1 < nrun ? 4 : 0);
      if (img_item >= 0) *reinterpret_cast<float4*>(img_s + (tid >> 4) * 64 + 4 * (tid & 15)) = imgr;
      __syncthreads();
      // gz = gradient * lrelu'(fromRGB output); per-channel sums of gz * image and of gz: four pixels in fp32, then fp64
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        float gz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gz[r] = ((mb[r4] >> r) & 1u) ? acc[r4][r] : acc[r4][r] * p.slope;
        acc[r4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 iv = *reinterpret_cast<const float4*>(img_s + (c * 4 + r4) * 64 + w * 16 + kk * 4);
          rsum[c] += (double)(fmaf(gz[0], iv.x, gz[1] * iv.y) + fmaf(gz[2], iv.z, gz[3] * iv.w));
        }
        rsum[3] += (double)((gz[0] + gz[1]) + (gz[2] + gz[3]));
      }
      continue;
    }
    if constexpr (!BLUR) {
      // ---- plain epilogue: +bias, activation, 16-byte stores of the four rows ----
      // (tried in round 4: the rows through an LDS tile, leaving as 256-byte runs from inside the next step's MFMA loop -
      // +1 %; a build WITHOUT the stores runs 1.13 instead of 1.35 ms: what the kernel pays is the 2 GiB of write traffic
      // next to its 2 GiB of reads, not the shape of the store instructions)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = Y0 + 4 * s + r4;
        u32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[r4][r] + bv;
          if (p.act == GANLAB_ACT_LRELU) t = gl_lrelu(t, p.slope);
          o[r] = __float_as_uint(t);
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo_lane == RB_OOB ? vo_lane : vo_lane + row * p.W * 4, 0, RB_STORE_AUX);
        acc[r4] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      RB_PH(2)
      __syncthreads();          // rows 4s .. 4s+3 of the ring are no longer read
      RB_PH(3)
      store_rows(4 * s + 6, s + 1 < nrun ? 4 : 0);
      RB_PH(4)
      __syncthreads();
      RB_PH(5)
      continue;
    }
    // ---- activation; publish what the neighbours need ----
    f32x4 v[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int row = Y0 - 1 + 4 * s + r4;
      const bool in = (unsigned)row < (unsigned)p.H;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r4][r] = in ? rb_act(acc[r4][r] + bv, p.slope) : 0.f;
      acc[r4] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The epilogue is written WITHOUT lane-dependent branches (round 4: the first version's 19 exec-mask regions, four
    // serialised shuffle -> wait -> byte-store rounds and per-item guards made it a ~600-instruction dependent chain of ~5 us per
    // step, tools/phase_probe_rb.py): lanes that have nothing to publish write to a scratch slot of their own, every lane
    // reads the neighbour values it may need, selections are v_cndmask.
    {
      const bool k0 = kk == 0, k3 = kk == 3;
      const float4 e1 = k0 ? float4{v[0][0], v[1][0], v[2][0], v[3][0]} : float4{v[0][3], v[1][3], v[2][3], v[3][3]};
      float* d1 = k0 ? xn + ((0 * 4 + w) * 16 + co) * 4 : (k3 ? xn + ((1 * 4 + w) * 16 + co) * 4 : scratch_s + tid * 4);
      *reinterpret_cast<float4*>(d1) = e1;
      float* d2 = kk < 2 ? xe + ((kk * 4 + w) * 16 + co) * 4 : scratch_s + tid * 4;   // kk = 0: column -1, kk = 1: column 64
      *reinterpret_cast<float4*>(d2) = float4{acce[0], acce[1], acce[2], acce[3]};
    }
    acce = f32x4{0.f, 0.f, 0.f, 0.f};
    RB_PH(2)
    __syncthreads();          // rows 4s .. 4s+3 of the ring are no longer read; the exchange buffers are complete
    RB_PH(3)
    store_rows(4 * s + 6, s + 1 < nrun ? 4 : 0);       // into the slots of rows 4s .. 4s+3 (an unguarded form of this
                                                       // staging spilled 48 registers: it keeps its per-item guards)
    RB_PH(6)
    // ---- horizontal blur (unnormalised [1 2 1]); w and txi are wave-uniform: scalar branches, every lane reads ----
    float4 nl = float4{0.f, 0.f, 0.f, 0.f}, nr = nl;      // left neighbour of pixel 0 / right neighbour of pixel 15 (by row)
    if (w > 0) {
      nl = *reinterpret_cast<const float4*>(xn + ((1 * 4 + (w - 1)) * 16 + co) * 4);
    } else if (txi > 0) {
      float4 t = *reinterpret_cast<const float4*>(xe + ((0 * 4 + 0) * 16 + co) * 4);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const float4 u = *reinterpret_cast<const float4*>(xe + ((0 * 4 + ww) * 16 + co) * 4);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      nl = float4{rb_act(t.x + bv, p.slope), rb_act(t.y + bv, p.slope), rb_act(t.z + bv, p.slope), rb_act(t.w + bv, p.slope)};
    }
    if (w < 3) {
      nr = *reinterpret_cast<const float4*>(xn + ((0 * 4 + (w + 1)) * 16 + co) * 4);
    } else if (txi + 1 < p.cols) {
      float4 t = *reinterpret_cast<const float4*>(xe + ((1 * 4 + 0) * 16 + co) * 4);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const float4 u = *reinterpret_cast<const float4*>(xe + ((1 * 4 + ww) * 16 + co) * 4);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      nr = float4{rb_act(t.x + bv, p.slope), rb_act(t.y + bv, p.slope), rb_act(t.z + bv, p.slope), rb_act(t.w + bv, p.slope)};
    }
    f32x4 hb[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int row = Y0 - 1 + 4 * s + r4;
      const bool in = (unsigned)row < (unsigned)p.H;
      float l = __shfl_up(v[r4][3], 16, 64);      // lane - 16: pixel 4kk - 1
      float r = __shfl_down(v[r4][0], 16, 64);    // lane + 16: pixel 4kk + 4
      const float nlv = r4 == 0 ? nl.x : r4 == 1 ? nl.y : r4 == 2 ? nl.z : nl.w;
      const float nrv = r4 == 0 ? nr.x : r4 == 1 ? nr.y : r4 == 2 ? nr.z : nr.w;
      l = kk == 0 ? (in ? nlv : 0.f) : l;
      r = kk == 3 ? (in ? nrv : 0.f) : r;
      hb[r4][0] = l + 2.f * v[r4][0] + v[r4][1];
      hb[r4][1] = v[r4][0] + 2.f * v[r4][1] + v[r4][2];
      hb[r4][2] = v[r4][1] + 2.f * v[r4][2] + v[r4][3];
      hb[r4][3] = v[r4][2] + 2.f * v[r4][3] + r;
    }
    RB_PH(7)
    // ---- vertical blur, one row behind: output rows Y0 - 2 + 4s + i from (c0, c1, hb[0..3]); rows this strip does not own
    //      get an out-of-range offset (dropped by the descriptor's bounds check) ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = Y0 - 2 + 4 * s + i;
      const f32x4 a = i == 0 ? c0 : i == 1 ? c1 : hb[i - 2];
      const f32x4 b = i == 0 ? c1 : hb[i - 1];
      const f32x4 c = hb[i];
      u32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = __float_as_uint((a[r] + 2.f * b[r] + c[r]) * 0.0625f);
      const bool own = row >= Y0 && row < Y0 + 4 * ns;
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, (own && vo_lane != RB_OOB) ? vo_lane + row * p.W * 4 : RB_OOB, 0, RB_STORE_AUX);
    }
    c0 = hb[2];
    c1 = hb[3];
    RB_PH(8)
    // ---- sign bits of the step's four activation rows: ONE shuffle and ONE 4-byte LDS store per lane pair (8 pixels x 4
    //      rows), bits_s[co][w * 2 + kk / 2] = {row 0, row 1, row 2, row 3} bytes ----
    if (p.bits != nullptr) {
      unsigned n16 = 0;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        n16 |= ((v[r4][0] > 0.f ? 1u : 0u) | (v[r4][1] > 0.f ? 2u : 0u) | (v[r4][2] > 0.f ? 4u : 0u) | (v[r4][3] > 0.f ? 8u : 0u))
               << (4 * r4);
      const unsigned h16 = __shfl_down(n16, 16, 64);           // lane group kk + 1: the next four pixels of every row
      auto spread = [](unsigned x) {                            // nibble r -> low nibble of byte r
        x = (x | (x << 8)) & 0x00ff00ffu;
        return (x | (x << 4)) & 0x0f0f0f0fu;
      };
      unsigned* dst = (kk & 1) == 0 ? bits_s + co * 8 + w * 2 + (kk >> 1) : reinterpret_cast<unsigned*>(scratch_s) + tid * 4;
      *dst = spread(n16) | (spread(h16) << 4);
    }
    RB_PH(9)
    __syncthreads();   // the ring holds rows 4s+4 .. 4s+9; the exchange buffers may be rewritten
    RB_PH(5)
  }
  if (nrun > 0) flush_bits(nrun - 1);
  RB_PH_FLUSH(nrun)
  if constexpr (RGB) {        // this workgroup's partial sums: lane groups, then waves (fixed order)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      rsum[c] += __shfl_xor(rsum[c], 16, 64);
      rsum[c] += __shfl_xor(rsum[c], 32, 64);
    }
    __syncthreads();
    if (kk == 0)
#pragma unroll
      for (int c = 0; c < 4; ++c) red_s[(w * 16 + co) * 4 + c] = rsum[c];
    __syncthreads();
    if (tid < 64 && (tid >> 2) < p.Cout) {
      const int c = tid & 3, ch = tid >> 2;
      const double t = (red_s[(0 * 16 + ch) * 4 + c] + red_s[(1 * 16 + ch) * 4 + c]) +
                       (red_s[(2 * 16 + ch) * 4 + c] + red_s[(3 * 16 + ch) * 4 + c]);
      p.part[(long long)(ch * 4 + c) * gridDim.x + blockIdx.x] = t;
    }
  }
}

// gw[co][c] (c < img_c) = scale * sum_k part[co*4 + c][k], gb[co] = bias_scale * sum_k part[co*4 + 3][k]; `accumulate`: onto
// what the outputs hold (a parameter's second gradient contribution of a step)
__global__ void rb_rgb_finish_kernel(const double* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb, int chunks,
                                     int img_c, float scale, float bias_scale, int accumulate) {
  const int ch = blockIdx.x >> 2, c = blockIdx.x & 3;
  double t = 0.0;
  for (int i = threadIdx.x; i < chunks; i += 64) t += part[(long long)blockIdx.x * chunks + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if (threadIdx.x != 0) return;
  if (c < img_c && gw != nullptr) {
    float* d = gw + ch * img_c + c;
    *d = ((accumulate & 1) ? *d : 0.f) + (float)(t * (double)scale);
  } else if (c == 3 && gb != nullptr) {
    gb[ch] = ((accumulate & 2) ? gb[ch] : 0.f) + (float)(t * (double)bias_scale);
  }
}

}  // namespace

// ---- host side (called from conv.hip's entry points) ----
// row strips per column: >= ~4096 workgroups, >= 16 (blurred form: every strip pays one extra step) / 8 steps each
// ------------------------------------------------------------------------------------------------------------------------------
// The plain form (conv + bias + activation, no blur) by HALF steps: two output rows per barrier.  The four-row step above is
// [144 MFMAs per wave][row stores][barrier][prefetched rows -> ring][barrier] with the ring stores 13.5 % of a step
// (profiles/r04_phase_probe_rb.txt) - they overwrite rows the step still reads.  A half step h reads ring rows 2h .. 2h+3 (rel row r =
// input row Y0 - 1 + r) for output rows Y0 + 2h, Y0 + 2h + 1; the rows the next half step adds (2h+4, 2h+5) go to the slots of rows
// 2h-2, 2h-1, dead since the previous barrier: their LDS stores - and the loads of rows 2h+8, 2h+9 into the same register set, two
// half steps ahead - are pieces between the MFMAs, and one barrier per half step publishes them (conv_s2_wgrad_roll2_kernel,
// wgrad_roll.hip, for the measurements behind this form).  Same column-block map: wave w owns the 16-pixel block w of both rows.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void conv_fwd_roll2_kernel(RBArgs p) {
  __shared__ __attribute__((aligned(16))) float ring[RB_SLOTS * RB_SLOT];
  constexpr int HITEMS = 2 * 16 * RB_Q, HPT = (HITEMS + 255) / 256;       // two rows: 576 float4, 3 per thread
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int co = lane & 15, kk = lane >> 4;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int txi = bid % p.cols;
  bid /= p.cols;
  const int syi = bid % p.strips;
  const int n0 = bid / p.strips;
  const int nh = 2 * min(p.spu, p.H / 4 - syi * p.spu);      // half steps = pairs of output rows
  const int ox0 = txi * RB_TW, Y0 = syi * p.spu * 4;
  const int plane = p.H * p.W;
  const float* xb = p.x + (long long)n0 * p.Cin * plane;

  int gb[HPT], lo[HPT];         // byte offset of (ci, column) inside the image (or OOB) ; LDS offset | row of the pair << 16 (or -1)
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % RB_Q, t = e / RB_Q;
    const int ci = t & 15, k = t >> 4;
    const int vx = ox0 - 4 + 4 * q;
    gb[i] = (e < HITEMS && ci < p.Cin && (unsigned)vx < (unsigned)p.W) ? (ci * plane + vx) * 4 : RB_OOB;
    lo[i] = e < HITEMS ? ((ci * RB_RP + 4 * q) | (k << 16)) : -1;
  }
  float wreg[36];
#pragma unroll
  for (int st = 0; st < 36; ++st)
    wreg[st] = p.wp[(long long)((st >> 2) * p.Cin_p + (st & 3) * 4 + kk) * p.Cout_p + co];
  const bool co_ok = co < p.Cout;
  const float bv = (p.bias != nullptr && co_ok) ? p.bias[co] * p.bias_scale : 0.f;
  const long long oplane = (long long)p.H * p.W;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (unsigned)(p.Cin * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n0 * p.Cout * oplane, 0, (unsigned)(p.Cout * oplane * 4), 0x00020000);
  const int vo_lane = co_ok ? (int)(((long long)co * oplane + ox0 + w * 16 + kk * 4) * 4) : RB_OOB;
  const int W4 = p.W * 4;

  float4 xr[2][HPT];
  auto load_item = [&](int r0, bool on, int set, int i) {        // rel rows r0, r0 + 1
    const int vy = Y0 - 1 + r0 + ((lo[i] >> 16) & 1);
    const int off = (on && gb[i] != RB_OOB && (unsigned)vy < (unsigned)p.H) ? gb[i] + vy * W4 : RB_OOB;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
    xr[set][i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  };
  auto store_item = [&](int r0, int set, int i) {                // r0 even: slots r0 % 6, r0 % 6 + 1
    if (lo[i] != -1)
      *reinterpret_cast<float4*>(ring + ((r0 % RB_SLOTS) + ((lo[i] >> 16) & 1)) * RB_SLOT + (lo[i] & 0xffff)) = xr[set][i];
  };
#pragma unroll
  for (int i = 0; i < HPT; ++i) load_item(0, true, 0, i);
#pragma unroll
  for (int i = 0; i < HPT; ++i) load_item(2, true, 1, i);
#pragma unroll
  for (int i = 0; i < HPT; ++i) store_item(0, 0, i);
#pragma unroll
  for (int i = 0; i < HPT; ++i) store_item(2, 1, i);
#pragma unroll
  for (int i = 0; i < HPT; ++i) load_item(4, true, 0, i);        // stored in half step 0
#pragma unroll
  for (int i = 0; i < HPT; ++i) load_item(6, nh > 2, 1, i);      // stored in half step 1
  __syncthreads();

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int a_lane = kk * RB_RP + w * 16 + co + 3;               // + LP(4) - pad(1); `co` doubles as the pixel index here
  auto half_step = [&](int h, int par) {                         // `par` = h & 1, a literal at both call sites
    int sb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sb[j] = ((2 * h + j) % RB_SLOTS) * RB_SLOT;
    // fetch order: input rows (0, 3) interleaved - one MFMA each, on different accumulators - then rows 1 and 2 (two MFMAs each)
    constexpr int PD = 3, NF = 48;
    float rb[PD + 1];
    auto fj = [](int f) { return f < 24 ? (f & 1 ? 3 : 0) : (f < 36 ? 1 : 2); };
    auto fi = [](int f) { return f < 24 ? (f >> 1) : (f - 24) % 12; };
    auto fetch = [&](int f, int slot) {
      const int i12 = fi(f), kx = i12 >> 2, c4 = i12 & 3;
      rb[slot] = ring[sb[fj(f)] + c4 * 4 * RB_RP + kx + a_lane];
    };
    auto piece = [&](int k) {          // k < 3: rows 2h+4, 2h+5 -> ring; then the same register set refilled with rows 2h+8, 2h+9
      if (k < HPT) store_item(2 * h + 4, par, k);
      else load_item(2 * h + 8, h + 3 < nh, par, k - HPT);
    };
#pragma unroll
    for (int f = 0; f < PD; ++f) fetch(f, f % (PD + 1));
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (f + PD < NF) fetch(f + PD, (f + PD) % (PD + 1));
      const int slot = f % (PD + 1), j = fj(f), i12 = fi(f), kx = i12 >> 2, c4 = i12 & 3;
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int ky = j - r2;
        if (ky >= 0 && ky < 3)
          acc[r2] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot], wreg[(ky * 3 + kx) * 4 + c4], acc[r2], 0, 0, 0);
      }
      if ((f & 7) == 3) {
        __builtin_amdgcn_sched_barrier(0);
        piece(f >> 3);                 // six pieces at f = 3, 11, 19, 27, 35, 43
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int row = Y0 + 2 * h + r2;
      u32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[r2][r] + bv;
        if (p.act == GANLAB_ACT_LRELU) t = gl_lrelu(t, p.slope);
        o[r] = __float_as_uint(t);
      }
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo_lane == RB_OOB ? vo_lane : vo_lane + row * W4, 0, 0);
      acc[r2] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
  };
  for (int h = 0; h < nh; h += 2) {
    half_step(h, 0);
    half_step(h + 1, 1);
  }
}

static long long rb_plan(RBArgs& a, int N, int H, int W, int blur) {
  a.cols = W / RB_TW;
  const int steps = H / 4;
  const long long cols = (long long)a.cols * N;
  int k = 1;
  while (k < steps && cols * k < 4096 && (steps + k) / (k + 1) >= (blur ? 16 : 8)) ++k;
  a.spu = (steps + k - 1) / k;
  a.strips = (steps + a.spu - 1) / a.spu;
  return cols * a.strips;
}

bool gl_roll_blur_supported(int N, int Cin, int Cout, int H, int W, const void* x, const void* y) {
  // (the A/B switch GANLAB_ROLL_BLUR=0 - conv kernel + blur pass - is the CALLER's: gan_lab_amd/ops.py k_conv_fwd_blur_bits)
  return N > 0 && Cin >= 1 && Cin <= 16 && Cout >= 1 && Cout <= 16 && W % RB_TW == 0 && H % 4 == 0 &&
         (long long)Cin * H * W * 4 < 0x7fffffffLL && (long long)Cout * H * W * 4 < 0x7fffffffLL &&
         (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

int gl_roll_blur_launch(const float* x, const float* wp, const float* bias, float* y, void* bits, int N, int Cin, int Cout,
                        int H, int W, int Cin_p, int Cout_p, float bias_scale, float slope, hipStream_t st, int blur,
                        int act) {
  RBArgs a{};
  a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.bits = reinterpret_cast<unsigned short*>(bits);
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.Cin_p = Cin_p; a.Cout_p = Cout_p;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  const long long grid = rb_plan(a, N, H, W, blur);
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  static const int half = [] { const char* e = getenv("GANLAB_ROLL_HALF"); return (e && e[0] == '0') ? 0 : 1; }();
  if (blur) GL_LAUNCH(conv_fwd_roll_blur_kernel<RB_BLUR>, dim3((unsigned)grid), dim3(256), 0, st, a);
  else if (half) GL_LAUNCH(conv_fwd_roll2_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);     // (GANLAB_ROLL_HALF=0: the 4-row steps)
  else GL_LAUNCH(conv_fwd_roll_blur_kernel<RB_PLAIN>, dim3((unsigned)grid), dim3(256), 0, st, a);
  return GL_CHECK_LAUNCH();
}

#ifdef GL_PHASES
extern "C" int ganlab_dbg_set_phase_buf_rb(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(rb_phase_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

int ganlab_conv_fwd_blur_supported(const ganlab_conv_geom* g, const void* x, const void* y) {
  if (!g || g->ks != 3 || g->pad != 1 || g->up || g->pool) return 0;
  return gl_roll_blur_supported(g->N, g->Cin, g->Cout, g->Hin, g->Win, x, y) ? 1 : 0;
}

int ganlab_conv_fwd_blur_bits_f32(const float* x, const float* wp, const float* bias, float* y, unsigned* ybits,
                                  const ganlab_conv_geom* g, float bias_scale, float slope, void* stream) {
  if (!x || !wp || !y || !g) return GANLAB_EINVAL;
  if (!ganlab_conv_fwd_blur_supported(g, x, y)) return GANLAB_EUNSUPPORTED;
  const int cin_p = (g->Cin + 15) / 16 * 16, cout_p = (g->Cout + 63) / 64 * 64;
  return gl_roll_blur_launch(x, wp, bias, y, ybits, g->N, g->Cin, g->Cout, g->Hin, g->Win, cin_p, cout_p, bias_scale,
                             slope, gl_stream(stream), 1, GANLAB_ACT_LRELU);
}

int ganlab_conv_dgrad_rgb_sums_supported(const ganlab_conv_geom* g, int img_c) {
  if (!g || g->ks != 3 || g->pad != 1 || g->up || g->pool || img_c < 1 || img_c > 3) return 0;
  return gl_roll_blur_supported(g->N, g->Cout, g->Cin, g->Hin, g->Win, nullptr, nullptr) ? 1 : 0;
}

size_t ganlab_conv_dgrad_rgb_sums_workspace(const ganlab_conv_geom* g) {
  if (!ganlab_conv_dgrad_rgb_sums_supported(g, 3)) return 0;
  RBArgs a{};
  const long long grid = rb_plan(a, g->N, g->Hin, g->Win, 0);
  return (size_t)g->Cin * 4 * (size_t)grid * sizeof(double);
}

/* The input gradient of the critic's first 3x3 conv (geometry g, weights wp as ganlab_conv_dgrad_f32 takes them) met by the
 * backward of the fromRGB layer in front of it (1x1 conv of the img_c-channel image + LeakyReLU, progan/architectures.py:
 * 232-237) inside the kernel: gz = dgrad(gy, w) * lrelu'(ybits); gw_rgb[co][c] = scale * sum gz[co] * img[c],
 * gb_rgb[co] = bias_scale * sum gz[co] - the gradient tensor itself is not written.  acc_w / acc_b: add to what gw_rgb /
 * gb_rgb hold.  For callers that do not need fromRGB's input gradient. */
int ganlab_conv_dgrad_rgb_sums_f32(const float* gy, const float* wp, const unsigned* ybits, const float* img, float* gw_rgb,
                                   float* gb_rgb, const ganlab_conv_geom* g, int img_c, float scale, float bias_scale,
                                   float slope, int acc_w, int acc_b, void* workspace, size_t workspace_bytes, void* stream) {
  if (!gy || !wp || !ybits || !img || !g || (!gw_rgb && !gb_rgb)) return GANLAB_EINVAL;
  if (!ganlab_conv_dgrad_rgb_sums_supported(g, img_c) || (reinterpret_cast<uintptr_t>(gy) & 15) ||
      (reinterpret_cast<uintptr_t>(img) & 15))
    return GANLAB_EUNSUPPORTED;
  if (!workspace || workspace_bytes < ganlab_conv_dgrad_rgb_sums_workspace(g)) return GANLAB_EWORKSPACE;
  RBArgs a{};
  a.x = gy; a.wp = wp; a.bias = nullptr; a.y = nullptr; a.bits = nullptr;
  a.N = g->N; a.Cin = g->Cout; a.Cout = g->Cin; a.H = g->Hin; a.W = g->Win;     // roles swap: the operator consumes gy
  a.Cin_p = (g->Cout + 15) / 16 * 16; a.Cout_p = (g->Cin + 63) / 64 * 64;
  a.bias_scale = 0.f; a.slope = slope; a.act = GANLAB_ACT_NONE;
  a.img = img; a.mbits = reinterpret_cast<const unsigned char*>(ybits); a.part = reinterpret_cast<double*>(workspace);
  a.img_c = img_c;
  const long long grid = rb_plan(a, g->N, g->Hin, g->Win, 0);
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  hipStream_t st = gl_stream(stream);
  GL_LAUNCH(conv_fwd_roll_blur_kernel<RB_RGB>, dim3((unsigned)grid), dim3(256), 0, st, a);
  GL_LAUNCH(rb_rgb_finish_kernel, dim3((unsigned)(g->Cin * 4)), dim3(64), 0, st, (const double*)a.part, gw_rgb, gb_rgb,
            (int)grid, img_c, scale, bias_scale, (acc_w ? 1 : 0) | (acc_b ? 2 : 0));
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
