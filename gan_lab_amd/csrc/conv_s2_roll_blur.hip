// The thin transposed stride-2 convolution (conv_s2_roll.hip's T: 32 low-resolution channels -> 16 high-resolution ones, the
// top resolution block of both networks) with the binomial blur that FOLLOWS it - and what follows the blur - in ONE kernel:
//   TB_TAIL  generator forward, Upsample -> conv3x3 -> blur -> +noise -> +bias -> LeakyReLU (stylegan/architectures.py:
//            292-334, 497-526):  a = act(blur(T x) + noise_w[c] * noise[n,hw] + bias[c] * bias_scale)  plus the fp64 partial
//            sums of a and a^2 per (n, c) plane, from which the InstanceNorm statistics of the layer are finished - the
//            blur + tail pass (pointwise.hip blur_fused_kernel<BF_FWD, 8, stats>: 1R + 1W of a 2 GiB tensor) is gone;
//   TB_MASK  critic backward, conv -> LeakyReLU -> blur -> pooled conv (progan/architectures.py:254-284): the pooled
//            conv's input gradient T(gy), the blur's adjoint (itself) and the LeakyReLU derivative of the layer in front,
//            gz = lrelu'(y) * blur(T gy)  from y's sign bits, plus the fp64 partial sums of gz per channel (that layer's
//            bias gradient) - the blur^T + act' pass (blur_fused_kernel<BF_A, 8>: 2R + 1W) is gone.
// The MFMA phase is conv_s2_up_roll_kernel's: a workgroup owns a strip of 32 low-resolution columns and walks down it two
// low rows (four high rows) per step, wave (py, jrow) computes high row 2*(Y + jrow) + py of 64 pixels x 16 channels with
// its 64 of the 8192 weights in registers.  Then
//   * horizontal blur inside the wave that owns the row (a lane holds 2 x 8 consecutive pixels of its channel; the pixel
//     beyond an 8-group comes from the next lane group: ds_bpermute).  The two columns just outside the strip are
//     contracted on the VALU from the wave's own weight registers (each lane group holds a quarter of the 32 input
//     channels: 64 FMAs, two shuffle reductions) - no MFMA block of 2 useful pixels in 16;
//   * the horizontally blurred rows go through a six-row LDS ring (the four of this step + the last two of the previous
//     one); after the step's first barrier wave q blurs row 4s - 1 + q vertically out of it, applies the tail and stores:
//     the output lags the convolution by one row, and a row strip runs ONE extra step for the conv rows its neighbours
//     own (rows 4*Y0 - 1 and 4*Y0 + 4n);
//   * the input ring shrinks to four low rows (two barriers per step, like conv_fwd_roll_kernel's six-slot form) to pay for
//     the output ring: 24.6 + 26.1 KB, three workgroups per CU.
// Zero padding of the blur: conv rows / columns outside the image are zero.
#include "common.h"

#include <stdlib.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TB_OOB = (int)0x80000000;
constexpr int TB_TW = 32, TB_CP = 48, TB_SLOT = 32 * TB_CP, TB_SLOTS = 4, TB_Q = 10;
constexpr int TB_ITEMS = 2 * 32 * TB_Q, TB_PT = (TB_ITEMS + 255) / 256;      // 640 float4 per 2-row prefetch, 3 per thread
constexpr int TB_OP = 68, TB_OROW = 16 * TB_OP, TB_OSLOTS = 6;               // blurred rows: [16 ch][68] floats
enum { TB_TAIL = 0, TB_MASK = 1 };

struct TBArgs {
  const float* x;           // low-resolution input (TAIL: the deferred activation a; MASK: the pooled conv's output gradient)
  const float* wp;          // [16 taps][Cin_p][Cout_p] (ganlab_conv_s2_pack_f32)
  float* y;                 // high-resolution output
  const float* aff_s;       // TAIL with a deferred-InstanceNorm input: [N][Cin] scale / shift applied on load, or null
  const float* aff_t;
  const float* bias;        // TAIL
  const float* noise;       // TAIL: (N, 1, 2Hl, 2Wl) or null
  const float* noise_w;     // TAIL: [Cout]
  const unsigned char* bits;   // MASK: sign bits of the activation in front of the blur, bit e of the NCHW-linear index
  double* part;             // TAIL: [N * Cout][chunks][2] (sum a, sum a^2); MASK: [Cout][grid] (sum gz) or null
  int N, Cin, Cout, Hl, Wl, Cin_p, Cout_p;
  int cols, strips, spu;    // column strips per row, row strips per column, steps (2 low rows) per strip
  float bias_scale, slope;
  int act;
};

template <int MODE, bool AFF>
__global__ __launch_bounds__(256, 3) void conv_s2_up_roll_blur_kernel(TBArgs p) {
  __shared__ __attribute__((aligned(16))) float ring[TB_SLOTS * TB_SLOT];       // 24576 B
  __shared__ __attribute__((aligned(16))) float oring[TB_OSLOTS * TB_OROW];     // 26112 B
  __shared__ double red[4 * 16 * 2];
  __shared__ float afftab[AFF ? 64 : 1];
  constexpr int NG = 32, PD = 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int py = wv & 1, jrow = wv >> 1;    // conv row of the step: 2 * jrow + py = wv
  const int co = lane & 15, kk = lane >> 4; // MFMA: lane & 15 = pixel (A) / output channel (B, D); kk = channel in the K-group
  int bid = gl_xcd_remap(blockIdx.x, gridDim.x);
  const int col = bid % p.cols;
  bid /= p.cols;
  const int strip = bid % p.strips;
  const int n = bid / p.strips;
  const int X0 = col * TB_TW, Ys = strip * p.spu * 2;
  const int ns = min(p.spu, p.Hl / 2 - strip * p.spu);      // steps that own output rows; the kernel runs ns + 1
  const int lplane = p.Hl * p.Wl, Hh = 2 * p.Hl, Wh = 2 * p.Wl, HY0 = 2 * Ys;
  const long long hplane = 4LL * p.Hl * p.Wl;

  int gbase[TB_PT], lo_k[TB_PT], lo_off[TB_PT];
#pragma unroll
  for (int i = 0; i < TB_PT; ++i) {
    const int e = tid + i * 256;
    const int q = e % TB_Q, t = e / TB_Q;
    const int ci = t & 31, k = t >> 5;
    const int vx = X0 - 4 + 4 * q;
    gbase[i] = (e < TB_ITEMS && ci < p.Cin && (unsigned)vx < (unsigned)p.Wl) ? (ci * lplane + vx) * 4 : TB_OOB;
    lo_k[i] = k & 1;
    lo_off[i] = ci * TB_CP + 4 * q;
  }
  // weights -> registers: wreg[(iy*4 + b)*8 + c4] = K4[a(py, iy)][b][4*c4 + kk][co]
  float wreg[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int iy = i >> 5, b = (i >> 3) & 3, c4 = i & 7;
    const int a = py == 0 ? (iy == 0 ? 3 : 1) : (iy == 0 ? 2 : 0);
    wreg[i] = p.wp[(long long)((a * 4 + b) * p.Cin_p + c4 * 4 + kk) * p.Cout_p + co];
  }
  const bool co_ok = co < p.Cout;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (long long)n * p.Cin * lplane), 0, (unsigned)(p.Cin * lplane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (long long)n * p.Cout * hplane, 0, (unsigned)(p.Cout * hplane * 4), 0x00020000);
  // this lane's 2 x 8 output pixels of a row: high columns 2*X0 + 32*blk + 8*kk ..
  const int vo = co_ok ? (int)(((long long)co * hplane + 2 * X0 + 8 * kk) * 4) : TB_OOB;

  float4 xr[TB_PT];
  // rel low row r of the strip = low row Ys - 2 + r; `on` false: nothing (all offsets out of range)
  auto load_rows = [&](int rel0, bool on) {
#pragma unroll
    for (int i = 0; i < TB_PT; ++i) {
      const int vy = Ys - 2 + rel0 + lo_k[i];
      const bool ok = on && gbase[i] != TB_OOB && (unsigned)vy < (unsigned)p.Hl;
      const int off = ok ? gbase[i] + (int)((unsigned)vy * (unsigned)(p.Wl * 4)) : TB_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
      xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
  };
  auto store_rows = [&](int rel0, bool on) {
#pragma unroll
    for (int i = 0; i < TB_PT; ++i) {
      if (tid + i * 256 < TB_ITEMS) {
        float4 v = xr[i];
        if constexpr (AFF) {    // the load's validity test again: rows / columns / channels outside stay zero
          const int vy = Ys - 2 + rel0 + lo_k[i], ci = lo_off[i] / TB_CP;
          const bool ok = on && gbase[i] != TB_OOB && (unsigned)vy < (unsigned)p.Hl;
          const float sv = ok ? afftab[ci] : 0.f, tv = ok ? afftab[32 + ci] : 0.f;
          v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
        }
        *reinterpret_cast<float4*>(ring + ((rel0 + lo_k[i]) % TB_SLOTS) * TB_SLOT + lo_off[i]) = v;
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
  double sum1 = 0.0, sum2 = 0.0;
  float bv = 0.f, nwv = 0.f;
  if constexpr (MODE == TB_TAIL) {
    bv = (p.bias != nullptr && co_ok) ? p.bias[co] * p.bias_scale : 0.f;
    nwv = (p.noise != nullptr && co_ok) ? p.noise_w[co] : 0.f;
  }
  load_rows(0, true);
  store_rows(0, true);
  load_rows(2, true);
  store_rows(2, true);
  __syncthreads();
  const int lane_off = kk * TB_CP + co + 4;      // A operand: low column X0 + (lane & 15) of channel 4*c4 + kk
  const bool edge_l = X0 > 0, edge_r = X0 + TB_TW < p.Wl;

  for (int s = 0; s <= ns; ++s) {
    // this step's output row (one behind the convolution) and what its tail reads from memory - the noise map resp. the
    // sign bits of its 2 x 8 pixels - requested NOW: the round trip hides behind the MFMA phase
    const int orow_g = HY0 - 3 + 4 * s + wv;
    const bool own = orow_g >= HY0 && orow_g < HY0 + 4 * ns;
    [[maybe_unused]] float4 nzp[4];
    [[maybe_unused]] unsigned mbits[2] = {0xffu, 0xffu};
    if constexpr (MODE == TB_TAIL) {
#pragma unroll
      for (int i = 0; i < 4; ++i) nzp[i] = float4{0.f, 0.f, 0.f, 0.f};
      if (own && p.noise != nullptr) {
        const float* nrow = p.noise + ((long long)n * Hh + orow_g) * Wh + 2 * X0 + 8 * kk;
#pragma unroll
        for (int i = 0; i < 4; ++i) nzp[i] = *reinterpret_cast<const float4*>(nrow + 32 * (i >> 1) + 4 * (i & 1));
      }
    } else {
      if (own && co_ok) {
        const unsigned char* brow = p.bits + (((((long long)n * p.Cout + co) * Hh + orow_g) * Wh + 2 * X0 + 8 * kk) >> 3);
        mbits[0] = brow[0];
        mbits[1] = brow[4];
      }
    }
    // ---- MFMA phase: conv row HY0 - 2 + 4s + wv from rel low rows 2s .. 2s+3 ----
    int sb[2], se[2];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      const int dy = py == 0 ? (iy == 0 ? -1 : 0) : (iy == 0 ? 0 : 1);
      const int slot = (2 * s + 1 + jrow + dy) % TB_SLOTS;
      sb[iy] = slot * TB_SLOT + lane_off;
      se[iy] = slot * TB_SLOT + kk * TB_CP;
    }
    float rb[PD + 1][3];
    auto fetch = [&](int g, int sl) {
      const int iy = g >> 4, c4 = (g >> 1) & 7, blk = g & 1;
      const float* src = ring + sb[iy] + c4 * 4 * TB_CP + blk * 16;
      rb[sl][0] = src[-1];
      rb[sl][1] = src[0];
      rb[sl][2] = src[1];
    };
#pragma unroll
    for (int g = 0; g < PD; ++g) fetch(g, g % (PD + 1));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PD < NG) fetch(g + PD, (g + PD) % (PD + 1));
      const int sl = g % (PD + 1), iy = g >> 4, c4 = (g >> 1) & 7, blk = g & 1;
      const float* w = wreg + iy * 32 + c4;
      acc[0][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[sl][0], w[3 * 8], acc[0][blk], 0, 0, 0);   // px 0: (dx -1, b 3)
      acc[1][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[sl][1], w[2 * 8], acc[1][blk], 0, 0, 0);   // px 1: (dx  0, b 2)
      acc[0][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[sl][1], w[1 * 8], acc[0][blk], 0, 0, 0);   // px 0: (dx  0, b 1)
      acc[1][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[sl][2], w[0 * 8], acc[1][blk], 0, 0, 0);   // px 1: (dx +1, b 0)
      if (g == 2) load_rows(2 * s + 4, s < ns);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the two high columns just outside the strip, on the VALU: 2*X0 - 1 (odd pixel of low column X0 - 1: taps
    //      (dx 0, b 2) and (dx +1, b 0)) and 2*X0 + 64 (even pixel of low column X0 + 32: (dx -1, b 3), (dx 0, b 1)) ----
    float el = 0.f, er = 0.f;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float* src = ring + se[iy] + c4 * 4 * TB_CP;
        const float* w = wreg + iy * 32 + c4;
        el = fmaf(src[3], w[2 * 8], el);      // low column X0 - 1
        el = fmaf(src[4], w[0 * 8], el);      // X0
        er = fmaf(src[35], w[3 * 8], er);     // X0 + 31
        er = fmaf(src[36], w[1 * 8], er);     // X0 + 32
      }
    el += __shfl_xor(el, 16, 64);
    el += __shfl_xor(el, 32, 64);
    er += __shfl_xor(er, 16, 64);
    er += __shfl_xor(er, 32, 64);
    const int crow = HY0 - 2 + 4 * s + wv;            // this wave's conv row
    const bool rin = (unsigned)crow < (unsigned)Hh;
    if (!edge_l || !rin) el = 0.f;
    if (!edge_r || !rin) er = 0.f;
    // ---- horizontal blur (unnormalised [1 2 1]) of the wave's row; into the output ring ----
    {
      float px8[2][8];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          px8[blk][2 * r] = rin ? acc[0][blk][r] : 0.f;
          px8[blk][2 * r + 1] = rin ? acc[1][blk][r] : 0.f;
          acc[0][blk][r] = 0.f;
          acc[1][blk][r] = 0.f;
        }
      float lft[2], rgt[2];
      lft[0] = __shfl_up(px8[0][7], 16, 64);
      lft[1] = __shfl_up(px8[1][7], 16, 64);
      rgt[0] = __shfl_down(px8[0][0], 16, 64);
      rgt[1] = __shfl_down(px8[1][0], 16, 64);
      const float wrap_l = __shfl(px8[0][7], (lane + 48) & 63, 64);     // kk == 0 of block 1: block 0's last pixel (lane group 3)
      const float wrap_r = __shfl(px8[1][0], (lane + 16) & 63, 64);     // kk == 3 of block 0: block 1's first pixel (lane group 0)
      if (kk == 0) { lft[0] = el; lft[1] = wrap_l; }
      if (kk == 3) { rgt[0] = wrap_r; rgt[1] = er; }
      float* orow = oring + ((4 * s + wv) % TB_OSLOTS) * TB_OROW + co * TB_OP + 8 * kk;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        float h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = j == 0 ? lft[blk] : px8[blk][j - 1];
          const float c = j == 7 ? rgt[blk] : px8[blk][j + 1];
          h[j] = a + 2.f * px8[blk][j] + c;
        }
        *reinterpret_cast<float4*>(orow + 32 * blk) = float4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<float4*>(orow + 32 * blk + 4) = float4{h[4], h[5], h[6], h[7]};
      }
    }
    __syncthreads();          // rel rows 2s, 2s+1 of the input ring are dead; the blurred rows 4s .. 4s+3 are complete
    store_rows(2 * s + 4, s < ns);
    // ---- vertical blur, one row behind: wave q finishes conv row 4s - 1 + q (high row HY0 - 3 + 4s + q) ----
    if (own) {
      const int r1 = 4 * s - 1 + wv;                 // >= 2 here
      const float* ra = oring + ((r1 - 1) % TB_OSLOTS) * TB_OROW + co * TB_OP + 8 * kk;
      const float* rbm = oring + (r1 % TB_OSLOTS) * TB_OROW + co * TB_OP + 8 * kk;
      const float* rc = oring + ((r1 + 1) % TB_OSLOTS) * TB_OROW + co * TB_OP + 8 * kk;
      const int obase = vo == TB_OOB ? TB_OOB : vo + orow_g * Wh * 4;
      float rs1 = 0.f, rs2 = 0.f;        // this row's 16 pixels: fp32 partial sums, then one fp64 addition each
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const float4 a = *reinterpret_cast<const float4*>(ra + 32 * blk + 4 * hf);
          const float4 b = *reinterpret_cast<const float4*>(rbm + 32 * blk + 4 * hf);
          const float4 c = *reinterpret_cast<const float4*>(rc + 32 * blk + 4 * hf);
          // two pixels per instruction (v_pk_fma_f32 / v_pk_mul_f32): VALU issue is what these kernels pay for the fold
          const f32x2 alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w}, clo = {c.x, c.y}, chi = {c.z, c.w};
          const f32x2 two = {2.f, 2.f}, sc = {0.0625f, 0.0625f};
          const f32x2 olo = (__builtin_elementwise_fma(blo, two, alo) + clo) * sc;
          const f32x2 ohi = (__builtin_elementwise_fma(bhi, two, ahi) + chi) * sc;
          float o[4] = {olo[0], olo[1], ohi[0], ohi[1]};
          if constexpr (MODE == TB_TAIL) {
            const float4 nz = nzp[2 * blk + hf];
            const float nzv[4] = {nz.x, nz.y, nz.z, nz.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = fmaf(nzv[j], nwv, o[j] + bv);
              if (p.act == GANLAB_ACT_LRELU) v = gl_lrelu(v, p.slope);
              o[j] = v;
            }
            // four terms in fp32 (a relative 1e-7 of a 4-term sum), everything beyond in fp64
            rs1 += (o[0] + o[1]) + (o[2] + o[3]);
            rs2 += fmaf(o[0], o[0], o[1] * o[1]) + fmaf(o[2], o[2], o[3] * o[3]);
          } else {
            const unsigned m = mbits[blk] >> (4 * hf);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = ((m >> j) & 1u) ? o[j] : o[j] * p.slope;
            rs1 += (o[0] + o[1]) + (o[2] + o[3]);
          }
          const u32x4 ov = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
          __builtin_amdgcn_raw_buffer_store_b128(ov, rs_out, obase == TB_OOB ? TB_OOB : obase + (32 * blk + 4 * hf) * 4, 0, 0);
        }
      sum1 += (double)rs1;
      if constexpr (MODE == TB_TAIL) sum2 += (double)rs2;
    }
    __syncthreads();          // the input ring holds rel rows 2s+2 .. 2s+5; output-ring rows 4s-2 .. 4s+1 may be rewritten
  }
  // ---- this workgroup's partial sums (fixed order: lane groups, then waves) ----
  if (p.part != nullptr) {
    sum1 += __shfl_xor(sum1, 16, 64);
    sum1 += __shfl_xor(sum1, 32, 64);
    if constexpr (MODE == TB_TAIL) {
      sum2 += __shfl_xor(sum2, 16, 64);
      sum2 += __shfl_xor(sum2, 32, 64);
    }
    if (kk == 0) {
      red[(wv * 16 + co) * 2] = sum1;
      red[(wv * 16 + co) * 2 + 1] = sum2;
    }
    __syncthreads();
    if (tid < 16 && tid < p.Cout) {
      const double a = (red[(0 * 16 + tid) * 2] + red[(1 * 16 + tid) * 2]) + (red[(2 * 16 + tid) * 2] + red[(3 * 16 + tid) * 2]);
      if constexpr (MODE == TB_TAIL) {
        const double b = (red[(0 * 16 + tid) * 2 + 1] + red[(1 * 16 + tid) * 2 + 1]) +
                         (red[(2 * 16 + tid) * 2 + 1] + red[(3 * 16 + tid) * 2 + 1]);
        const long long chunks = (long long)p.cols * p.strips, chunk = (long long)col * p.strips + strip;
        double* dst = p.part + ((((long long)n * p.Cout + tid) * chunks) + chunk) * 2;
        dst[0] = a;
        dst[1] = b;
      } else {
        p.part[(long long)tid * gridDim.x + blockIdx.x] = a;
      }
    }
  }
}

// mean / rstd of every plane from the workgroup partials (fixed order, fp64) - pointwise.hip's act_stats_finish_kernel
__global__ void tb_stats_finish_kernel(const double* __restrict__ spart, float* __restrict__ mean, float* __restrict__ rstd,
                                       long long planes, int chunks, double inv_hw, float eps) {
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

// out[c] = scale * sum_k part[c][k]: one wave per channel, lane-strided then a shuffle tree (channel_sum_stage2's order)
__global__ void tb_sum_finish_kernel(const double* __restrict__ part, float* __restrict__ out, int chunks, float scale) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < chunks; i += 64) s += part[(long long)c * chunks + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[c] = (float)(s * (double)scale);
}

inline bool tb_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool tb_enabled() { return true; }      // (the A/B switch GANLAB_S2_ROLL_BLUR=0 is the caller's: gan_lab_amd/ops.py conv_s2_blur_ok)

void tb_plan(TBArgs& a) {
  a.cols = a.Wl / TB_TW;
  const int steps = a.Hl / 2;
  const long long columns = (long long)a.cols * a.N;
  int k = 1;               // >= ~4096 workgroups, >= 16 steps each (every strip pays one extra step)
  while (k < steps && columns * k < 4096 && (steps + k) / (k + 1) >= 16) ++k;
  a.spu = (steps + k - 1) / k;
  a.strips = (steps + a.spu - 1) / a.spu;
}

bool tb_geometry_ok(int N, int Cin, int Cout, int Hl, int Wl) {
  return tb_enabled() && N > 0 && Hl >= 2 && (Hl & 1) == 0 && Wl % TB_TW == 0 && Cin > 16 && Cin <= 32 && Cout >= 1 &&
         Cout <= 16 && (long long)32 * Hl * Wl * 4 * 4 < 0x7fffffffLL;
}

}  // namespace

extern "C" {

// geometry g: the UP layer (g->up == 1: Cin low channels -> Cout high ones) for the TAIL form; the POOLED layer
// (g->pool == 1: its input gradient maps Cout low channels -> Cin high ones) for the MASK form
int ganlab_conv_s2_blur_supported(const ganlab_conv_geom* g) {
  if (!g || g->ks != 3 || g->pad != 1 || (g->up != 0) == (g->pool != 0)) return 0;
  if (g->up) return tb_geometry_ok(g->N, g->Cin, g->Cout, g->Hin, g->Win) ? 1 : 0;
  if ((g->Hin & 1) || (g->Win & 1)) return 0;
  return tb_geometry_ok(g->N, g->Cout, g->Cin, g->Hin / 2, g->Win / 2) ? 1 : 0;
}

size_t ganlab_conv_s2_blur_workspace(const ganlab_conv_geom* g) {
  if (!ganlab_conv_s2_blur_supported(g)) return 0;
  TBArgs a{};
  a.N = g->N;
  a.Hl = g->up ? g->Hin : g->Hin / 2;
  a.Wl = g->up ? g->Win : g->Win / 2;
  tb_plan(a);
  const long long grid = (long long)a.cols * a.strips * a.N;
  return (size_t)(g->up ? (long long)a.N * g->Cout * a.cols * a.strips * 2 : (long long)g->Cin * grid) * sizeof(double);
}

/* a = act(blur(conv(up2(x * s + t), w)) + noise_w * noise + bias * bias_scale); mean / rstd: InstanceNorm statistics of a.
 * aff_s / aff_t may be NULL (plain input).  wp: ganlab_conv_s2_pack_f32(up = 1, transposed = 0). */
int ganlab_conv_s2_fwd_blur_tail_f32(const float* x, const float* wp, const float* aff_s, const float* aff_t,
                                     const float* bias, const float* noise, const float* noise_w, float* y, float* mean,
                                     float* rstd, const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !wp || !y || !mean || !rstd || !g || (noise && !noise_w) || ((aff_s == nullptr) != (aff_t == nullptr)))
    return GANLAB_EINVAL;
  if (!g->up || !ganlab_conv_s2_blur_supported(g) || !tb_aligned16(x) || !tb_aligned16(y) || (noise && !tb_aligned16(noise)))
    return GANLAB_EUNSUPPORTED;
  if (!workspace || workspace_bytes < ganlab_conv_s2_blur_workspace(g)) return GANLAB_EWORKSPACE;
  TBArgs a{};
  a.x = x; a.wp = wp; a.y = y; a.aff_s = aff_s; a.aff_t = aff_t; a.bias = bias; a.noise = noise; a.noise_w = noise_w;
  a.part = reinterpret_cast<double*>(workspace);
  a.N = g->N; a.Cin = g->Cin; a.Cout = g->Cout; a.Hl = g->Hin; a.Wl = g->Win;
  a.Cin_p = (g->Cin + 15) / 16 * 16; a.Cout_p = (g->Cout + 63) / 64 * 64;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  tb_plan(a);
  const long long grid = (long long)a.cols * a.strips * a.N;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  hipStream_t st = gl_stream(stream);
  if (aff_s != nullptr) GL_LAUNCH((conv_s2_up_roll_blur_kernel<TB_TAIL, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
  else GL_LAUNCH((conv_s2_up_roll_blur_kernel<TB_TAIL, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
  const long long planes = (long long)g->N * g->Cout;
  GL_LAUNCH(tb_stats_finish_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, st, (const double*)a.part, mean, rstd,
            planes, a.cols * a.strips, 1.0 / (4.0 * g->Hin * g->Win), eps);
  return GL_CHECK_LAUNCH();
}

/* gz = lrelu'(ybits) * blur(dgrad(gy, w)) and gb[c] = bias_scale * sum gz (gb may be NULL): the input gradient of the pooled
 * conv g fused with the backward of the  LeakyReLU -> blur  in front of it.  wp: ganlab_conv_s2_pack_f32(up = 0,
 * transposed = 1) - what ganlab_conv_s2_dgrad_f32 takes. */
int ganlab_conv_s2_dgrad_blur_act_bits_f32(const float* gy, const float* wp, const unsigned* ybits, float* gz, float* gb,
                                           const ganlab_conv_geom* g, float slope, float bias_scale, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (!gy || !wp || !ybits || !gz || !g) return GANLAB_EINVAL;
  if (!g->pool || !ganlab_conv_s2_blur_supported(g) || !tb_aligned16(gy) || !tb_aligned16(gz)) return GANLAB_EUNSUPPORTED;
  if (gb && (!workspace || workspace_bytes < ganlab_conv_s2_blur_workspace(g))) return GANLAB_EWORKSPACE;
  TBArgs a{};
  a.x = gy; a.wp = wp; a.y = gz; a.bits = reinterpret_cast<const unsigned char*>(ybits);
  a.part = gb ? reinterpret_cast<double*>(workspace) : nullptr;
  a.N = g->N; a.Cin = g->Cout; a.Cout = g->Cin; a.Hl = g->Hin / 2; a.Wl = g->Win / 2;    // roles swap: the operator consumes gy
  a.Cin_p = (g->Cout + 15) / 16 * 16; a.Cout_p = (g->Cin + 63) / 64 * 64;
  a.slope = slope; a.act = GANLAB_ACT_NONE;
  tb_plan(a);
  const long long grid = (long long)a.cols * a.strips * a.N;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  hipStream_t st = gl_stream(stream);
  GL_LAUNCH((conv_s2_up_roll_blur_kernel<TB_MASK, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
  if (gb) GL_LAUNCH(tb_sum_finish_kernel, dim3((unsigned)g->Cin), dim3(64), 0, st, (const double*)a.part, gb, (int)grid, bias_scale);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
