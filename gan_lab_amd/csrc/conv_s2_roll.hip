// Rolling-window forms of the THIN stride-2 fused convolutions (conv_s2.hip's S and T operators) for the one channel
// pair where the tile kernels are furthest from the matrix-core peak: 16 channels at the high resolution, 32 at the low
// one - the top resolution block of both networks (D: conv3x3 16->32 + AvgPool2d, progan/architectures.py:261-284; G:
// Upsample + conv3x3 32->16, stylegan/architectures.py:292-334) and their input gradients:
//   S  high (N,16,2Hl,2Wl) -> low (N,32,Hl,Wl):  forward of the pooled conv, input gradient of the up-conv
//   T  low  (N,32,Hl,Wl)  -> high (N,16,2Hl,2Wl): forward of the up-conv,    input gradient of the pooled conv
// The tile kernels (conv_s2_down_kernel<SCfg<2>>, conv_s2_up_kernel<TCfg<1,4>>) stage a 4- / 8-channel K-chunk of a
// halo'd 32x8 tile per 128 MFMAs per wave, with two barriers per chunk and the weight slab re-staged every chunk:
// 0.63-0.69 of the fp32 MFMA peak.  Here, as in conv.hip's conv_fwd_roll_kernel and wgrad_roll.hip:
//   * a workgroup owns a column strip of 32 low-resolution pixels and walks DOWN it two low rows (four high rows) per
//     step; the input rows live in an LDS ring (S: 10 slots of [16 ch][80], T: 6 slots of [32 ch][48]); every input
//     element is fetched once per workgroup (plus the column halo), the next rows are prefetched into registers
//     during the MFMA loop and written into the slots nobody reads: ONE barrier per step of 128 MFMAs per wave;
//   * ALL input channels are contracted in one go and the 16-tap weight set lives in REGISTERS: each of the four
//     waves owns one (output row of the step, output-channel block / output row parity) pair, so it needs 64 of the
//     8192 weights per lane - no weight traffic after the prologue, LDS carries activations only;
//   * S stores a high row split by column parity (E | O'), so that the four horizontal taps of a low pixel are two
//     pairs of ADJACENT floats (ds_read2_b32: two MFMA operands per LDS instruction); T reads x[X-1], x[X], x[X+1] once
//     for the four (output parity, tap) products that use them;
//   * the activations are the MFMA's A operand (M = pixel), the weights its B operand (N = channel): a lane ends up
//     with 4 consecutive pixels of one output channel - 16-byte stores (T: 8 consecutive high-resolution pixels, two).
// Channel pitches 80 / 48 floats are = 16 mod 32: the two k-groups of a 32-lane ds_read_b32 group hit disjoint banks.
#include "common.h"

#include <stdlib.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int SR_OOB = (int)0x80000000;

struct SRArgs {
  const float* x;       // S: high-res input ; T: low-res input
  const float* wp;      // [16 taps][Cin_p][Cout_p] (ganlab_conv_s2_pack_f32)
  const float* bias;
  float* y;
  int N, Cin, Cout;
  int Hl, Wl;           // LOW resolution
  int Cin_p, Cout_p;
  int cols, strips, spu;   // column strips per row, row strips per column, steps (2 low rows) per strip
  float bias_scale, slope;
  int act;
  const float* aff_s;      // T only: deferred InstanceNorm of the input (conv_s2.hip, S2Args), [N][Cin] or null
  const float* aff_t;
};

// ===================================================================================================================
// S: y[n,co,Y,X] = sum_{a,b<4} sum_ci K4[a][b][ci][co] * xpad[n,ci,2Y+a-1,2X+b-1]
// ===================================================================================================================
constexpr int SR_TW = 32;                 // low pixels per strip (64 high)
constexpr int SR_CP = 80;                 // floats per channel row: E[e] = high[2(X0+e)] at e, O'[o] = high[2(X0+o)-1] at 41+o
constexpr int SR_SLOT = 16 * SR_CP;       // one high row, 16 channels
#ifndef SR_NSLOTS
#define SR_NSLOTS 10
#endif
constexpr int SR_SLOTS = SR_NSLOTS;       // 10: rows 4t .. 4t+5 are read while 4t+6 .. 4t+9 are written (one barrier per step); 6: they
                                          // overwrite rows 4t .. 4t+3 behind a second barrier (31 KB instead of 51)
constexpr int SR_Q = 18;                  // float4 per (row, channel): high columns 2X0-4 .. 2X0+67
constexpr int SR_ITEMS = 4 * 16 * SR_Q;   // 1152 float4 per 4-row prefetch
constexpr int SR_PT = (SR_ITEMS + 255) / 256;   // 5

__global__ __launch_bounds__(256, (SR_NSLOTS == 6 ? 4 : 3)) void conv_s2_down_roll_kernel(SRArgs p) {
  __shared__ __attribute__((aligned(16))) float ring[SR_SLOTS * SR_SLOT];      // 51200 B: three workgroups per CU
  constexpr int NG = 16, PD = 2;            // 16 operand groups of 8 MFMAs per step; LDS prefetch distance
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mco = wv & 1, jrow = wv >> 1;   // this wave: output channels 16*mco .., low row 2t + jrow of every step
  const int px = lane & 15, kk = lane >> 4;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int col = bid % p.cols;             // x fastest: neighbouring workgroups share image rows
  bid /= p.cols;
  const int strip = bid % p.strips;
  const int n = bid / p.strips;
  const int X0 = col * SR_TW, Ys = strip * p.spu * 2;
  const int nsteps = min(p.spu, p.Hl / 2 - strip * p.spu);
  const int H = 2 * p.Hl, W = 2 * p.Wl, hplane = H * W;
  const long long lplane = (long long)p.Hl * p.Wl;

  int gbase[SR_PT], lo[SR_PT];
#pragma unroll
  for (int i = 0; i < SR_PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % SR_Q, t = e / SR_Q;
    const int ci = t & 15, k = t >> 4;
    const int vx = 2 * X0 - 4 + 4 * q;
    gbase[i] = (e < SR_ITEMS && ci < p.Cin && (unsigned)vx < (unsigned)W) ? (ci * hplane + vx) * 4 : SR_OOB;
    lo[i] = (ci * SR_CP + 2 * q) | (k << 20) | ((q > 0 ? 1 : 0) << 24);
  }
  // weights -> registers: wreg[(a*4 + b)*4 + c4] = K4[a][b][4*c4 + kk][16*mco + px]
  float wreg[64];
#pragma unroll
  for (int i = 0; i < 64; ++i)
    wreg[i] = p.wp[(long long)((i >> 2) * p.Cin_p + (i & 3) * 4 + kk) * p.Cout_p + mco * 16 + px];
  const int co = mco * 16 + px;
  const bool co_ok = co < p.Cout;
  const float bv = (p.bias != nullptr && co_ok) ? p.bias[co] * p.bias_scale : 0.f;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (long long)n * p.Cin * hplane), 0, (unsigned)(p.Cin * hplane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n * p.Cout * lplane, 0, (unsigned)(p.Cout * lplane * 4), 0x00020000);
  int vo[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
    vo[blk] = co_ok ? (int)(((long long)co * lplane + X0 + blk * 16 + 4 * kk) * 4) : SR_OOB;

  float4 xr[SR_PT];
  // high rows rel0 .. rel0 + nrows - 1 of the strip (rel row r = high row 2*Ys - 1 + r) -> registers; rows outside the
  // image, columns outside it and k >= nrows read as zeros (out-of-range offsets)
  auto load_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < SR_PT; ++i) {
      const int k = (lo[i] >> 20) & 3;
      const int vy = 2 * Ys - 1 + rel0 + k;
      const bool ok = gbase[i] != SR_OOB && k < nrows && (unsigned)vy < (unsigned)H;
      const int off = ok ? gbase[i] + (int)((unsigned)vy * (unsigned)(W * 4)) : SR_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto store_rows = [&](int rel0, int nrows) {
#pragma unroll
    for (int i = 0; i < SR_PT; ++i) {
      const int k = (lo[i] >> 20) & 3;
      if (tid + i * 256 < SR_ITEMS && k < nrows) {
        float* d = ring + ((rel0 + k) % SR_SLOTS) * SR_SLOT + (lo[i] & 0xfffff);
        if (lo[i] & (1 << 24)) *reinterpret_cast<float2*>(d - 2) = float2{xr[i].x, xr[i].z};    // E[2q-2], E[2q-1]
        *reinterpret_cast<float2*>(d + 40) = float2{xr[i].y, xr[i].w};                          // O'[2q-1], O'[2q]
      }
    }
  };

  // two accumulation chains of 128 products per output (tap rows 0-1, tap rows 2-3), added at the end: a single chain
  // of 256 rounds 1.3x worse than ATen's blocked sums (tools/op_error_probe.py; common.h GL_ACC_DUMP)
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  [[maybe_unused]] f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  load_rows(0, 4);
  store_rows(0, 4);
  load_rows(4, 2);
  store_rows(4, 2);
  __syncthreads();
  const int lane_off = kk * SR_CP + px;
  for (int t = 0; t < nsteps; ++t) {
    int sb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) sb[a] = ((4 * t + 2 * jrow + a) % SR_SLOTS) * SR_SLOT + lane_off;
    // group g = a*4 + c4: E[x], E[x+1], O'[x], O'[x+1] of channel 4*c4 + kk, high row 2Y + a - 1, for both 16-pixel
    // blocks - the eight MFMAs of a group alternate between the two accumulators (a dependent 16x16x4 MFMA issues 40
    // cycles after its producer, an independent one 32)
    float rb[PD + 1][2][4];
    auto fetch = [&](int g, int s) {
      const int a = g >> 2, c4 = g & 3;
      const float* src = ring + sb[a] + c4 * 4 * SR_CP;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        rb[s][blk][0] = src[blk * 16];
        rb[s][blk][1] = src[blk * 16 + 1];
        rb[s][blk][2] = src[blk * 16 + 41];
        rb[s][blk][3] = src[blk * 16 + 42];
      }
    };
#pragma unroll
    for (int g = 0; g < PD; ++g) fetch(g, g % (PD + 1));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PD < NG) fetch(g + PD, (g + PD) % (PD + 1));
      const int s = g % (PD + 1), a = g >> 2, c4 = g & 3;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int v = b == 0 ? 2 : (b == 1 ? 0 : (b == 2 ? 3 : 1));     // b=0: O'[x], 1: E[x], 2: O'[x+1], 3: E[x+1]
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          if (GL_ACC_DUMP && g >= NG / 2)
            accb[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[s][blk][v], wreg[(a * 4 + b) * 4 + c4], accb[blk], 0, 0, 0);
          else
            acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[s][blk][v], wreg[(a * 4 + b) * 4 + c4], acc[blk], 0, 0, 0);
        }
      }
      // the next step's four high rows: issued a few groups INTO the loop (behind queued MFMAs, see conv_fwd_roll_kernel)
      if (g == 1) load_rows(4 * t + 6, t + 1 < nsteps ? 4 : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SR_SLOTS == 10) store_rows(4 * t + 6, 4);          // the four slots no wave reads in this step
    const int Y = Ys + 2 * t + jrow;
    const int orow = Y * p.Wl * 4;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      u32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (GL_ACC_DUMP ? acc[blk][r] + accb[blk][r] : acc[blk][r]) + bv;
        if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
        o[r] = __float_as_uint(v);
      }
      // offset in the VGPR, soffset 0 (store-data hazard with an SGPR soffset: conv.hip, conv_fwd_strip2_kernel)
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, vo[blk] == SR_OOB ? vo[blk] : vo[blk] + orow, 0, 0);
      acc[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
      accb[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                   // rows 4t .. 4t+3 are free for the next step's prefetch; 4t+6 .. 4t+9 complete
    if constexpr (SR_SLOTS == 6) {
      store_rows(4 * t + 6, 4);        // = the slots of rows 4t .. 4t+3
      __syncthreads();
    }
  }
}

// ===================================================================================================================
// T: y[n,co,2Y+py,2X+px] = sum_{(dy,a) in taps(py)} sum_{(dx,b) in taps(px)} sum_ci K4[a][b][ci][co] * x[n,ci,Y+dy,X+dx]
//    taps(0) = {(-1,3), (0,1)}   taps(1) = {(0,2), (+1,0)}
// ===================================================================================================================
constexpr int TR_TW = 32;
constexpr int TR_CP = 48;                 // floats per channel row: low column X at X - X0 + 4
constexpr int TR_SLOT = 32 * TR_CP;       // one low row, 32 channels
constexpr int TR_SLOTS = 6;               // rows 2t .. 2t+3 are read while 2t+4, 2t+5 are written
constexpr int TR_Q = 10;                  // float4 per (row, channel): low columns X0-4 .. X0+35
constexpr int TR_ITEMS = 2 * 32 * TR_Q;   // 640 float4 per 2-row prefetch
constexpr int TR_PT = (TR_ITEMS + 255) / 256;   // 3

template <bool AFF>
__global__ __launch_bounds__(256, 3) void conv_s2_up_roll_kernel(SRArgs p) {
  __shared__ __attribute__((aligned(16))) float ring[TR_SLOTS * TR_SLOT];      // 36864 B
  __shared__ float afftab[AFF ? 64 : 1];    // s | t of this image's (<= 32) input channels
  constexpr int NG = 32, PD = 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int py = wv & 1, jrow = wv >> 1;    // this wave: output row parity, low row 2t + jrow of every step
  const int px = lane & 15, kk = lane >> 4;
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int col = bid % p.cols;
  bid /= p.cols;
  const int strip = bid % p.strips;
  const int n = bid / p.strips;
  const int X0 = col * TR_TW, Ys = strip * p.spu * 2;
  const int nsteps = min(p.spu, p.Hl / 2 - strip * p.spu);
  const int lplane = p.Hl * p.Wl, Wh = 2 * p.Wl;
  const long long hplane = 4LL * p.Hl * p.Wl;

  int gbase[TR_PT], lo[TR_PT];
#pragma unroll
  for (int i = 0; i < TR_PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % TR_Q, t = e / TR_Q;
    const int ci = t & 31, k = t >> 5;
    const int vx = X0 - 4 + 4 * q;
    gbase[i] = (e < TR_ITEMS && ci < p.Cin && (unsigned)vx < (unsigned)p.Wl) ? (ci * lplane + vx) * 4 : SR_OOB;
    lo[i] = (ci * TR_CP + 4 * q) | (k << 20);
  }
  // weights -> registers: wreg[(iy*4 + b)*8 + c4] = K4[a(py, iy)][b][4*c4 + kk][px]
  float wreg[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int iy = i >> 5, b = (i >> 3) & 3, c4 = i & 7;
    const int a = py == 0 ? (iy == 0 ? 3 : 1) : (iy == 0 ? 2 : 0);
    wreg[i] = p.wp[(long long)((a * 4 + b) * p.Cin_p + c4 * 4 + kk) * p.Cout_p + px];
  }
  const bool co_ok = px < p.Cout;
  const float bv = (p.bias != nullptr && co_ok) ? p.bias[px] * p.bias_scale : 0.f;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (long long)n * p.Cin * lplane), 0, (unsigned)(p.Cin * lplane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n * p.Cout * hplane, 0, (unsigned)(p.Cout * hplane * 4), 0x00020000);
  int vo[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
    vo[blk] = co_ok ? (int)(((long long)px * hplane + 2 * (X0 + blk * 16 + 4 * kk)) * 4) : SR_OOB;

  float4 xr[TR_PT];
  // low rows rel0, rel0 + 1 of the strip (rel row r = low row Ys - 1 + r); `on` false: nothing (all offsets out of range)
  auto load_rows = [&](int rel0, bool on) {
#pragma unroll
    for (int i = 0; i < TR_PT; ++i) {
      const int k = (lo[i] >> 20) & 1;
      const int vy = Ys - 1 + rel0 + k;
      const bool ok = on && gbase[i] != SR_OOB && (unsigned)vy < (unsigned)p.Hl;
      const int off = ok ? gbase[i] + (int)((unsigned)vy * (unsigned)(p.Wl * 4)) : SR_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto store_rows = [&](int rel0, bool on) {
#pragma unroll
    for (int i = 0; i < TR_PT; ++i) {
      const int k = (lo[i] >> 20) & 1;
      if (tid + i * 256 < TR_ITEMS) {
        float4 v = xr[i];
        if constexpr (AFF) {    // the load's validity test again: rows / columns / channels outside stay zero
          const int vy = Ys - 1 + rel0 + k, ci = (lo[i] & 0xfffff) / TR_CP;
          const bool ok = on && gbase[i] != SR_OOB && (unsigned)vy < (unsigned)p.Hl;
          const float sv = ok ? afftab[ci] : 0.f, tv = ok ? afftab[32 + ci] : 0.f;
          v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
        }
        *reinterpret_cast<float4*>(ring + ((rel0 + k) % TR_SLOTS) * TR_SLOT + (lo[i] & 0xfffff)) = v;
      }
    }
  };
  if constexpr (AFF) {
    if (tid < 64) {
      const int c = tid & 31;
      afftab[tid] = c < p.Cin ? (tid < 32 ? p.aff_s : p.aff_t)[(long long)n * p.Cin + c] : 0.f;
    }
    __syncthreads();
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) acc[ph][blk] = f32x4{0.f, 0.f, 0.f, 0.f};
  load_rows(0, true);
  store_rows(0, true);
  load_rows(2, true);
  store_rows(2, true);
  __syncthreads();
  const int lane_off = kk * TR_CP + px + 4;
  for (int t = 0; t < nsteps; ++t) {
    int sb[2];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      const int dy = py == 0 ? (iy == 0 ? -1 : 0) : (iy == 0 ? 0 : 1);
      sb[iy] = ((2 * t + jrow + dy + 1) % TR_SLOTS) * TR_SLOT + lane_off;
    }
    // group g = (iy*8 + c4)*2 + blk: x[X-1], x[X], x[X+1] of channel 4*c4 + kk, low row Y + dy(iy)
    float rb[PD + 1][3];
    auto fetch = [&](int g, int s) {
      const int iy = g >> 4, c4 = (g >> 1) & 7, blk = g & 1;
      const float* src = ring + sb[iy] + c4 * 4 * TR_CP + blk * 16;
      rb[s][0] = src[-1];
      rb[s][1] = src[0];
      rb[s][2] = src[1];
    };
#pragma unroll
    for (int g = 0; g < PD; ++g) fetch(g, g % (PD + 1));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PD < NG) fetch(g + PD, (g + PD) % (PD + 1));
      const int s = g % (PD + 1), iy = g >> 4, c4 = (g >> 1) & 7, blk = g & 1;
      const float* w = wreg + iy * 32 + c4;
      acc[0][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[s][0], w[3 * 8], acc[0][blk], 0, 0, 0);   // px 0: (dx -1, b 3)
      acc[1][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[s][1], w[2 * 8], acc[1][blk], 0, 0, 0);   // px 1: (dx  0, b 2)
      acc[0][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[s][1], w[1 * 8], acc[0][blk], 0, 0, 0);   // px 0: (dx  0, b 1)
      acc[1][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[s][2], w[0 * 8], acc[1][blk], 0, 0, 0);   // px 1: (dx +1, b 0)
      if (g == 2) load_rows(2 * t + 4, t + 1 < nsteps);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_rows(2 * t + 4, t + 1 < nsteps);             // the two slots no wave reads in this step
    const int Y = Ys + 2 * t + jrow;
    const int orow = (2 * Y + py) * Wh * 4;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      float v0[4], v1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0[r] = acc[0][blk][r] + bv;
        v1[r] = acc[1][blk][r] + bv;
        if (p.act == GANLAB_ACT_LRELU) {
          v0[r] = gl_lrelu(v0[r], p.slope);
          v1[r] = gl_lrelu(v1[r], p.slope);
        }
      }
      const u32x4 oa = {__float_as_uint(v0[0]), __float_as_uint(v1[0]), __float_as_uint(v0[1]), __float_as_uint(v1[1])};
      const u32x4 ob = {__float_as_uint(v0[2]), __float_as_uint(v1[2]), __float_as_uint(v0[3]), __float_as_uint(v1[3])};
      const int off = vo[blk] == SR_OOB ? SR_OOB : vo[blk] + orow;
      __builtin_amdgcn_raw_buffer_store_b128(oa, rs_out, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(ob, rs_out, off == SR_OOB ? SR_OOB : off + 16, 0, 0);
      acc[0][blk] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1][blk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
  }
}

inline bool sr_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// Strips per column: many more workgroups than the chip holds at once (3 per CU for S, 4 for T), at least 8 steps each (a
// strip's six-row prologue is not overlapped).  Measured (tools/s2_roll_bench.py, 16 <-> 32 channels at 1024^2, batch 32):
// 2560 workgroups of 64 steps 0.723 / 0.748 / 0.749 / 0.740 of the fp32 MFMA peak (S fwd, T dgrad, T fwd, S dgrad); a
// "balanced" split - exactly one or two full rounds, 1536 / 1024 workgroups of 86 / 128 steps - 0.711 / 0.735 / 0.740 /
// 0.734: the workgroups of a round run in lockstep, and more, shorter ones drift apart and cover each other's barriers.
void sr_plan(SRArgs& a, int target) {
  a.cols = a.Wl / 32;
  const int steps = a.Hl / 2;
  const long long columns = (long long)a.cols * a.N;
  int kk = 1;
  while (kk < steps && columns * kk < target && (steps + kk) / (kk + 1) >= 8) ++kk;
  if (const char* e = GL_ENV_ONCE("GANLAB_S2_ROLL_STRIPS")) { if (atoi(e) > 0) kk = atoi(e); }    // tuning knob
  a.spu = (steps + kk - 1) / kk;
  a.strips = (steps + a.spu - 1) / a.spu;
}

bool sr_enabled() {
  static const bool on = [] { const char* e = getenv("GANLAB_S2_ROLL"); return !(e && e[0] == '0'); }();
  return on;
}

}  // namespace

// S: high channels (Cin) <= 16 -> low channels (Cout) in (16, 32]; T: low channels (Cin) in (16, 32] -> high (Cout) <= 16.
bool gl_s2_roll_supported(int is_T, int N, int Cin, int Cout, int Hl, int Wl, const void* x, const void* y) {
  if (!sr_enabled() || N <= 0 || Hl < 4 || (Hl & 1) || Wl % 32 != 0) return false;
  if (is_T ? !(Cin > 16 && Cin <= 32 && Cout <= 16) : !(Cin <= 16 && Cout > 16 && Cout <= 32)) return false;
  if ((long long)32 * Hl * Wl * 4 * 4 >= 0x7fffffffLL) return false;      // byte offsets of either tensor fit an int
  if ((x && !sr_aligned16(x)) || (y && !sr_aligned16(y))) return false;
  return true;
}

int gl_s2_roll_launch(int is_T, const float* x, const float* wp, const float* bias, float* y, int N, int Cin, int Cout,
                      int Hl, int Wl, int Cin_p, int Cout_p, float bias_scale, int act, float slope, hipStream_t st,
                      const float* aff_s, const float* aff_t) {
  if (aff_s != nullptr && !is_T) return GANLAB_EUNSUPPORTED;
  SRArgs a{};
  a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.aff_s = aff_s; a.aff_t = aff_t;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.Hl = Hl; a.Wl = Wl; a.Cin_p = Cin_p; a.Cout_p = Cout_p;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  sr_plan(a, 4096);
  const long long grid = (long long)a.cols * a.strips * N;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  if (is_T && aff_s != nullptr) GL_LAUNCH(conv_s2_up_roll_kernel<true>, dim3((unsigned)grid), dim3(256), 0, st, a);
  else if (is_T) GL_LAUNCH(conv_s2_up_roll_kernel<false>, dim3((unsigned)grid), dim3(256), 0, st, a);
  else GL_LAUNCH(conv_s2_down_roll_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
  return GL_CHECK_LAUNCH();
}
