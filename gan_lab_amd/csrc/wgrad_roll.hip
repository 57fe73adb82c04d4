// Rolling-window weight gradient for the THIN 3x3 layers (<= 32 channels on either side, W % 64 == 0, H % 4 == 0):
//   gw[co][ci][ky][kx] = sum_{n,y,x} gy[n,co,y,x] * xin[n,ci,y+ky-1,x+kx-1]        (wgrad of custom_layers.py:202-211)
//
// Why a second weight-gradient kernel.  conv_wgrad_kernel (conv.hip) stages one 32x8-pixel tile of BOTH operands per
// 144 MFMAs per wave: the halo'd input patch is 40x10 = 1.56x the tile, so a 16 -> 16 layer moves (64 + 100) bytes per
// pixel for 4608 FLOP - 28 FLOP/byte against a ridge of 25 (157.3 TFLOP/s over ~6.3 TB/s): MFMA pipes and HBM are both
// near saturation and neither hides the other (0.55 of the MFMA peak at 1024^2).  Here a workgroup owns a 64-pixel
// column strip and walks DOWN it four rows per step, like conv_fwd_roll_kernel:
//   * input rows live in a ring of 6 LDS row slots; every input element is fetched ONCE per workgroup (plus the
//     8-of-72 column halo): (64 + 72) bytes per pixel;
//   * wave w owns output-gradient row 4t + w of the step: the contraction index of the MFMA is the PIXEL, and the
//     k -> pixel map is transposed (lane k-group holds pixels 4k .. 4k+3 of a 16-pixel block, MFMA q takes pixel
//     4k + q), so ONE ds_read_b128 feeds four MFMAs' A operands and one ds_read_b128 + one ds_read_b64 feed the B
//     operands of twelve (three horizontal taps x four pixels): 0.2 LDS instructions per MFMA instead of 1.1;
//   * the x rows sit one float to the right in LDS (column c'' = vx - ox0 + 5), which is what makes those six B values
//     an aligned 16-byte + 8-byte pair; row pitches 88 / 72 floats make both reads bank-conflict free
//     (MI355X_MICROARCH.md, LDS table: ds_read_b128 is served in 4 groups of 16 lanes over 64 banks);
//   * nine tap accumulators per wave (36 registers), so consecutive MFMAs never wait on each other's 40-cycle
//     accumulator latency; the four waves' sums are added through LDS at the end: ONE slot per workgroup, and the
//     existing fixed-order slot reduction finishes (deterministic).
#include "common.h"

#include <stdlib.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WR_TW = 64, WR_ROWS = 4, WR_SLOTS = 6;
constexpr int WR_XP = 88, WR_XSLOT = 16 * WR_XP;        // x row slot: [ci 16][88]; column c'' = vx - ox0 + 5 (1 .. 72 used)
constexpr int WR_GP = 72, WR_GROW = 16 * WR_GP;         // gy row: [co 16][72]; column = vx - ox0 (0 .. 63 used)
constexpr int WR_XQ = 18;                               // float4 per (row, ci): vx = ox0 - 4 .. ox0 + 67
constexpr int WR_XITEMS = WR_ROWS * 16 * WR_XQ;         // 1152 float4 per 4-row group
constexpr int WR_XPT = (WR_XITEMS + 255) / 256;         // 5
constexpr int WR_GPT = 4;                               // gy: 4 rows x 16 co x 16 float4 = 1024 items, row k = item i
constexpr int WR_SMEM = WR_SLOTS * WR_XSLOT + WR_ROWS * WR_GROW;   // 13056 floats = 52224 B: three workgroups per CU
constexpr int WR_OOB = (int)0x80000000;

struct WRArgs {
  const float* x;
  const float* gy;
  float* part;            // [slots = gridDim.x][Cout][Cin][9]
  int N, Cin, Cout, H, W;
  int cols, strips, spu;  // 64-pixel columns per image, row strips per column, steps (of 4 rows) per strip
  int tiles_ci, tiles_co, S;
  int units;              // N * cols * strips
  int xcd;                // XCD-aware block mapping (A/B knob)
  int per_image;          // S = N * S_img workgroups per channel pair: workgroup `split` = (image split % N, j = split / N)
  int S_img;              // walks units j, j + S_img, ... of its own image only; slot `split` -> groups of N = per image
  // deferred InstanceNorm (ops.Deferred): x is a generator layer's activated output a; the operand of the contraction is
  // b = a * s[n,ci] + t[n,ci] inside the image, 0 in the padding - applied on the way from the prefetch registers to LDS
  const float* aff_s;     // [N][Cin] or null
  const float* aff_t;
};

template <bool AFF>   // AFF: its own instantiation - the plain kernel sits at the 168-VGPR cap of three workgroups per CU
__global__ __launch_bounds__(256, 3) void conv_wgrad_roll_kernel(WRArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[WR_SMEM];
  __shared__ float afftab[AFF ? 32 : 1];   // s | t of this workgroup's 16 input channels for the current unit's image
  float* ring = smem;
  float* gbuf = smem + WR_SLOTS * WR_XSLOT;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  // optional (see wr_plan): blocks b and b + 8 share an XCD - give each XCD a contiguous range of logical ids, so that
  // workgroups walking neighbouring column strips (consecutive `split`) share one L2
  int bid = p.xcd ? gl_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int split = bid % p.S;
  bid /= p.S;
  const int ci_t = bid % p.tiles_ci, co_t = bid / p.tiles_ci;
  const int ci0 = ci_t * 16, co0 = co_t * 16;
  const int plane = p.H * p.W;

  // staging items of a 4-row x group: (k = row in group, ci, q = float4 column)
  int xch[WR_XPT], xlo[WR_XPT], xk[WR_XPT], xcol[WR_XPT];
#pragma unroll
  for (int i = 0; i < WR_XPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % WR_XQ, t = e / WR_XQ;
    const int ci = t & 15, k = t >> 4;
    xch[i] = (e < WR_XITEMS && ci0 + ci < p.Cin) ? (ci0 + ci) * plane * 4 : WR_OOB;
    xlo[i] = ci * WR_XP + 4 * q + 1;
    xk[i] = k | (ci << 8);
    xcol[i] = 4 * q - 4;
  }
  // gy items: thread = (co = tid >> 4, q = tid & 15), item i = row i of the step
  const int gco = tid >> 4, gq = tid & 15;
  const int gch = (co0 + gco < p.Cout) ? (co0 + gco) * plane * 4 : WR_OOB;
  const int glo = gco * WR_GP + 4 * gq;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_off = wn * WR_GROW + (lane & 15) * WR_GP + 4 * (lane >> 4);
  const int b_off = (lane & 15) * WR_XP + 4 * (lane >> 4);

  float4 xr[WR_XPT], gr[WR_GPT];

  const int units_img = p.cols * p.strips;
  const int u_first = p.per_image ? (split % p.N) * units_img + split / p.N : split;
  const int u_end = p.per_image ? (split % p.N + 1) * units_img : p.units;
  const int u_step = p.per_image ? p.S_img : p.S;
  for (int u = u_first; u < u_end; u += u_step) {
    const int col = u % p.cols;
    const int t2 = u / p.cols;
    const int strip = t2 % p.strips, n = t2 / p.strips;
    const int ox0 = col * WR_TW, y0 = strip * p.spu * WR_ROWS;
    const int nsteps = min(p.spu, p.H / WR_ROWS - strip * p.spu);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (long long)n * p.Cin * plane), 0, (unsigned)(p.Cin * plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.gy + (long long)n * p.Cout * plane), 0, (unsigned)(p.Cout * plane * 4), 0x00020000);

    // rows rel0 .. rel0 + nrows - 1 of the strip (rel row r = image row y0 - 1 + r) -> registers; everything outside the
    // image, past nrows or in the channel padding reads as zero (out-of-range descriptor offsets)
    auto load_x = [&](int rel0, int nrows) {
#pragma unroll
      for (int i = 0; i < WR_XPT; ++i) {
        const int k = xk[i] & 255;
        const int vy = y0 - 1 + rel0 + k, vx = ox0 + xcol[i];
        const bool ok = xch[i] != WR_OOB && k < nrows && (unsigned)vy < (unsigned)p.H && (unsigned)vx < (unsigned)p.W;
        const int off = ok ? xch[i] + (vy * p.W + vx) * 4 : WR_OOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    };
    auto store_x = [&](int rel0, int nrows) {
#pragma unroll
      for (int i = 0; i < WR_XPT; ++i) {
        const int k = xk[i] & 255;
        if (tid + i * 256 < WR_XITEMS && k < nrows) {
          float4 v = xr[i];
          if constexpr (AFF) {          // the same validity test as the load: padding / channel padding stays zero
            const int vy = y0 - 1 + rel0 + k, vx = ox0 + xcol[i], ci = xk[i] >> 8;
            const bool ok = xch[i] != WR_OOB && (unsigned)vy < (unsigned)p.H && (unsigned)vx < (unsigned)p.W;
            const float sv = ok ? afftab[ci] : 0.f, tv = ok ? afftab[16 + ci] : 0.f;
            v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
          }
          float* d = ring + ((rel0 + k) % WR_SLOTS) * WR_XSLOT + xlo[i];
          d[0] = v.x;
          *reinterpret_cast<float2*>(d + 1) = float2{v.y, v.z};
          d[3] = v.w;
        }
      }
    };
    auto load_g = [&](int t, bool on) {
#pragma unroll
      for (int i = 0; i < WR_GPT; ++i) {
        const int off = (on && gch != WR_OOB) ? gch + ((y0 + 4 * t + i) * p.W + ox0 + 4 * gq) * 4 : WR_OOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off, 0, 0);
        gr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    };
    auto store_g = [&]() {
#pragma unroll
      for (int i = 0; i < WR_GPT; ++i) *reinterpret_cast<float4*>(gbuf + i * WR_GROW + glo) = gr[i];
    };

    if constexpr (AFF) {                   // (nobody reads the table between the previous unit's last barrier and this one)
      if (tid < 32) {
        const int c = ci0 + (tid & 15);
        afftab[tid] = c < p.Cin ? (tid < 16 ? p.aff_s : p.aff_t)[(long long)n * p.Cin + c] : 0.f;
      }
    }
    __syncthreads();                       // the previous unit's last step has been read
    load_x(0, 4);
    load_g(0, true);
    store_x(0, 4);
    store_g();
    load_x(4, 2);
    store_x(4, 2);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      const bool more = t + 1 < nsteps;
      int sb[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) sb[ky] = ((4 * t + wn + ky) % WR_SLOTS) * WR_XSLOT + b_off;
      // software pipeline over the 12 (g, ky) sub-steps: operands of sub-step s+1 are read while the 12 MFMAs of s issue
      float4 av[2];
      float4 b4[2];
      float2 b2[2];
      auto fetch_b = [&](int s, int buf) {
        const int g = s / 3, ky = s % 3;
        const float* src = ring + sb[ky] + 16 * g;
        b4[buf] = *reinterpret_cast<const float4*>(src + 4);
        b2[buf] = *reinterpret_cast<const float2*>(src + 8);
      };
      av[0] = *reinterpret_cast<const float4*>(gbuf + a_off);
      fetch_b(0, 0);
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        const int g = s / 3, ky = s % 3, cur = s & 1;
        if (s + 1 < 12) fetch_b(s + 1, cur ^ 1);
        if (ky == 0 && g + 1 < 4) av[(g + 1) & 1] = *reinterpret_cast<const float4*>(gbuf + a_off + 16 * (g + 1));
        // rows 4t+6 .. 4t+9 and the next four gy rows: issued one sub-step into the loop (MFMAs already queued)
        if (s == 1) {
          load_x(4 * t + 6, more ? 4 : 0);
          load_g(t + 1, more);
        }
        // pin the operand reads of sub-step s+1 (and the global prefetch) IN FRONT of this sub-step's MFMAs: left to
        // itself the scheduler sinks them to their first use, i.e. the loads to the barrier (their latency then sits
        // between the barrier and the LDS stores of every step) and the reads to a lgkmcnt(0) in front of every MFMA group
        __builtin_amdgcn_sched_barrier(0);
        const float a[4] = {av[g & 1].x, av[g & 1].y, av[g & 1].z, av[g & 1].w};
        const float e[6] = {b4[cur].x, b4[cur].y, b4[cur].z, b4[cur].w, b2[cur].x, b2[cur].y};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], e[q + kx], acc[ky * 3 + kx], 0, 0, 0);
      }
      __syncthreads();                     // rows 4t .. 4t+5 and the gy rows of this step are no longer read
      if (more) {
        store_x(4 * t + 6, 4);             // = the slots of rows 4t .. 4t+3
        store_g();
      }
      __syncthreads();
    }
  }

  // add the four waves' accumulators through LDS (fixed order) and write this workgroup's slot
  __syncthreads();
  float* red = smem;                       // [wave 4][tap 9][256]
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(red + (wn * 9 + t) * 256 + lane * 4) = acc[t];
  __syncthreads();
  // slot index = split: one slot per workgroup of this (co_t, ci_t) pair; the pairs write disjoint (co, ci) ranges
  float* dst = p.part + (long long)split * p.Cout * p.Cin * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float s = (red[(0 * 9 + t) * 256 + tid] + red[(1 * 9 + t) * 256 + tid]) +
                    (red[(2 * 9 + t) * 256 + tid] + red[(3 * 9 + t) * 256 + tid]);
    const int l = tid >> 2, r = tid & 3;
    const int co = co0 + (l >> 4) * 4 + r, ci = ci0 + (l & 15);
    if (co < p.Cout && ci < p.Cin) dst[((long long)co * p.Cin + ci) * 9 + t] = s;
  }
}

// ================================================================================================================
// The same rolling window for the 16-tap weight gradient of the stride-2 fused layers (conv_s2.hip, "W"):
//   gK4[a][b][cl][ch] = sum_{n,Y,X} low[n,cl,Y,X] * high_pad[n,ch,2Y+a-1,2X+b-1]      a, b in 0..3
// (down layer: low = gy, high = x; up layer: low = x, high = gy; conv_s2.hip folds gK4 back onto the 3x3 parameter).
// The tile kernel there stages a 40x12 high-resolution patch per 32x8... 16x4 low-resolution tile: 1.875x the tile's
// own pixels, every tile - its arithmetic intensity sits on the ridge as well (0.60-0.70 of the MFMA peak).
//   * column strip = 32 low-resolution pixels (64 high-resolution columns + halo = the same 18 float4 per row and
//     channel as above), step = 2 low rows = 4 new high rows: the SAME 6-slot ring bookkeeping (rel rows 4t .. 4t+5);
//   * wave a owns tap row a (4 x NBA accumulators): for low row j of the step it reads high rel row 4t + 2j + a;
//   * a high row is stored split by column parity, E[e] = high[2(X0+e)] at [0, 36) and O'[o] = high[2(X0+o)-1] at
//     [36, 72) of its 72-float channel row: tap b of low pixel X0+x reads E[x] (b=1), E[x+1] (b=3), O'[x] (b=0),
//     O'[x+1] (b=2), so with the transposed k -> pixel map (lane k-group = pixels 4k .. 4k+3) the five values per plane
//     are an aligned ds_read_b128 + ds_read_b64: 6 LDS instructions per 32 MFMAs.
// ================================================================================================================
constexpr int W2_TWL = 32, W2_RL = 2;
constexpr int W2_HP = 72, W2_HSLOT = 16 * W2_HP;        // high row slot [ch 16][E 36 | O' 36]
constexpr int W2_LP = 40;                               // low row: [cl][40], 32 used
constexpr int W2_LQ = 8;                                // float4 per (low row, cl)

struct W2RArgs {
  const float* low;
  const float* high;
  float* part;            // [slots][16 taps][Cl][Ch]
  int N, Cl, Ch, Hl, Wl;
  int cols, strips, spu;
  int tiles_cl, tiles_ch, S;
  int units;
  int xcd;
  const float* aff_s;     // deferred InstanceNorm on the LOW operand (up layer: low = x): [N][Cl] or null
  const float* aff_t;
};

template <int NBA, bool AFF>
__global__ __launch_bounds__(256, 3) void conv_s2_wgrad_roll_kernel(W2RArgs p) {
  constexpr int CL_T = 16 * NBA, LROW = CL_T * W2_LP;
  constexpr int LITEMS = W2_RL * CL_T * W2_LQ, LPT = LITEMS / 256;      // 256 / 512 float4 per step: 1 / 2 per thread
  static_assert(LITEMS % 256 == 0, "low staging items");
  __shared__ __attribute__((aligned(16))) float smem[WR_SLOTS * W2_HSLOT + W2_RL * LROW];
  __shared__ float afftab[AFF ? 2 * CL_T : 1];   // s | t of this workgroup's low channels for the current unit's image
  float* ring = smem;
  float* lbuf = smem + WR_SLOTS * W2_HSLOT;
  const int tid = threadIdx.x, lane = tid & 63, wa = tid >> 6;          // wa = tap row a of this wave
  // optional (see wr_plan): blocks b and b + 8 share an XCD - give each XCD a contiguous range of logical ids, so that
  // workgroups walking neighbouring column strips (consecutive `split`) share one L2
  int bid = p.xcd ? gl_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int split = bid % p.S;
  bid /= p.S;
  const int ch_t = bid % p.tiles_ch, cl_t = bid / p.tiles_ch;
  const int ch0 = ch_t * 16, cl0 = cl_t * CL_T;
  const int H = 2 * p.Hl, W = 2 * p.Wl, hplane = H * W, lplane = p.Hl * p.Wl;

  int xch[WR_XPT], xlo[WR_XPT], xk[WR_XPT], xq[WR_XPT];
#pragma unroll
  for (int i = 0; i < WR_XPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % WR_XQ, t = e / WR_XQ;
    const int ch = t & 15, k = t >> 4;
    xch[i] = (e < WR_XITEMS && ch0 + ch < p.Ch) ? (ch0 + ch) * hplane * 4 : WR_OOB;
    xlo[i] = ch * W2_HP + 2 * q;          // E pair at xlo - 2, O' pair at 36 + xlo - 1, 36 + xlo
    xk[i] = k;
    xq[i] = q;
  }
  int lch[LPT], llo[LPT], lrow[LPT], lq[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % W2_LQ, t = e / W2_LQ;
    const int cl = t % CL_T, j = t / CL_T;
    lch[i] = (cl0 + cl < p.Cl) ? (cl0 + cl) * lplane * 4 : WR_OOB;
    llo[i] = j * LROW + cl * W2_LP + 4 * q;
    lrow[i] = j | (cl << 8);
    lq[i] = q;
  }

  f32x4 acc[4][NBA];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int m = 0; m < NBA; ++m) acc[b][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_off = (lane & 15) * W2_LP + 4 * (lane >> 4);
  const int b_off = (lane & 15) * W2_HP + 4 * (lane >> 4);
  float4 xr[WR_XPT], lr[LPT];

  for (int u = split; u < p.units; u += p.S) {
    const int col = u % p.cols;
    const int t2 = u / p.cols;
    const int strip = t2 % p.strips, n = t2 / p.strips;
    const int X0 = col * W2_TWL, Y0 = strip * p.spu * W2_RL;
    const int nsteps = min(p.spu, p.Hl / W2_RL - strip * p.spu);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.high + (long long)n * p.Ch * hplane), 0, (unsigned)(p.Ch * hplane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.low + (long long)n * p.Cl * lplane), 0, (unsigned)(p.Cl * lplane * 4), 0x00020000);

    // high rel rows rel0 .. rel0 + nrows - 1 (rel row r = high row 2*Y0 - 1 + r)
    auto load_h = [&](int rel0, int nrows) {
#pragma unroll
      for (int i = 0; i < WR_XPT; ++i) {
        const int vy = 2 * Y0 - 1 + rel0 + xk[i], vx = 2 * X0 - 4 + 4 * xq[i];
        const bool ok = xch[i] != WR_OOB && xk[i] < nrows && (unsigned)vy < (unsigned)H && (unsigned)vx < (unsigned)W;
        const int off = ok ? xch[i] + (vy * W + vx) * 4 : WR_OOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_h, off, 0, 0);
        xr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    };
    auto store_h = [&](int rel0, int nrows) {
#pragma unroll
      for (int i = 0; i < WR_XPT; ++i) {
        if (tid + i * 256 < WR_XITEMS && xk[i] < nrows) {
          float* d = ring + ((rel0 + xk[i]) % WR_SLOTS) * W2_HSLOT + xlo[i];
          if (xq[i] > 0) {
            *reinterpret_cast<float2*>(d - 2) = float2{xr[i].x, xr[i].z};      // E[2q-2], E[2q-1]
            d[36 - 1] = xr[i].y;                                               // O'[2q-1]
          }
          d[36] = xr[i].w;                                                     // O'[2q]
        }
      }
    };
    auto load_l = [&](int t, bool on) {
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        const int off = (on && lch[i] != WR_OOB) ? lch[i] + ((Y0 + W2_RL * t + (lrow[i] & 255)) * p.Wl + X0 + 4 * lq[i]) * 4
                                                 : WR_OOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_l, off, 0, 0);
        lr[i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    };
    auto store_l = [&]() {       // (only called for rows that were loaded: every low pixel of the tile is inside the image)
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        float4 v = lr[i];
        if constexpr (AFF) {
          const int cl = lrow[i] >> 8;
          const bool ok = lch[i] != WR_OOB;
          const float sv = ok ? afftab[cl] : 0.f, tv = ok ? afftab[CL_T + cl] : 0.f;
          v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
        }
        *reinterpret_cast<float4*>(lbuf + llo[i]) = v;
      }
    };

    if constexpr (AFF) {
      if (tid < 2 * CL_T) {
        const int c = cl0 + (tid % CL_T);
        afftab[tid] = c < p.Cl ? (tid < CL_T ? p.aff_s : p.aff_t)[(long long)n * p.Cl + c] : 0.f;
      }
    }
    __syncthreads();
    load_h(0, 4);
    load_l(0, true);
    store_h(0, 4);
    store_l();
    load_h(4, 2);
    store_h(4, 2);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      const bool more = t + 1 < nsteps;
      int sb[W2_RL];
#pragma unroll
      for (int j = 0; j < W2_RL; ++j) sb[j] = ((4 * t + 2 * j + wa) % WR_SLOTS) * W2_HSLOT + b_off;
      // 4 sub-steps s = (j, g): 16 low pixels of row j each, 32 (NBA = 2) MFMAs per sub-step
      float4 av[2][NBA], e4[2], o4[2];
      float2 e2[2], o2[2];
      auto fetch = [&](int s, int buf) {
        const int j = s >> 1, g = s & 1;
        const float* src = ring + sb[j] + 16 * g;
        e4[buf] = *reinterpret_cast<const float4*>(src);
        e2[buf] = *reinterpret_cast<const float2*>(src + 4);
        o4[buf] = *reinterpret_cast<const float4*>(src + 36);
        o2[buf] = *reinterpret_cast<const float2*>(src + 40);
#pragma unroll
        for (int m = 0; m < NBA; ++m)
          av[buf][m] = *reinterpret_cast<const float4*>(lbuf + j * LROW + m * 16 * W2_LP + a_off + 16 * g);
      };
      fetch(0, 0);
#pragma unroll
      for (int s = 0; s < 2 * W2_RL; ++s) {
        const int cur = s & 1;
        if (s + 1 < 2 * W2_RL) fetch(s + 1, cur ^ 1);
        if (s == 0) {
          load_h(4 * t + 6, more ? 4 : 0);
          load_l(t + 1, more);
        }
        __builtin_amdgcn_sched_barrier(0);      // see conv_wgrad_roll_kernel: keep the prefetches in front of the MFMAs
        const float eE[5] = {e4[cur].x, e4[cur].y, e4[cur].z, e4[cur].w, e2[cur].x};
        const float eO[5] = {o4[cur].x, o4[cur].y, o4[cur].z, o4[cur].w, o2[cur].x};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float bv[4] = {eO[q], eE[q], eO[q + 1], eE[q + 1]};
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int m = 0; m < NBA; ++m) {
              const float a = q == 0 ? av[cur][m].x : (q == 1 ? av[cur][m].y : (q == 2 ? av[cur][m].z : av[cur][m].w));
              acc[b][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[b], acc[b][m], 0, 0, 0);
            }
        }
      }
      __syncthreads();
      if (more) {
        store_h(4 * t + 6, 4);
        store_l();
      }
      __syncthreads();
    }
  }
  // every wave owns its own tap row: taps 4a + b of this workgroup's slot; D rows = cl (lane>>4)*4+r, col = ch (lane&15)
  float* dst = p.part + (long long)split * 16 * p.Cl * p.Ch;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int m = 0; m < NBA; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = cl0 + m * 16 + (lane >> 4) * 4 + r, ch = ch0 + (lane & 15);
        if (cl < p.Cl && ch < p.Ch) dst[((long long)(wa * 4 + b) * p.Cl + cl) * p.Ch + ch] = acc[b][m][r];
      }
}

// The same kernel stepping by HALF steps (one low row = two new high rows per barrier).  In the kernel above a step is
// [128 MFMAs per wave][barrier][LDS stores of the prefetched rows][barrier]: the stores cannot start earlier because the four new
// high rows overwrite slots the step is still reading, and they cost 10 % of the kernel (a build without them: 1.105 -> 0.946 ms
// on 128 -> 256 pool @128^2 x32; without the loads as well: the loads are 4 %).  A half step h reads high rel rows 2h .. 2h+3
// and low row h; the rows the NEXT half step adds (2h+4, 2h+5; low row h+1) go to the two ring slots / the low slot that died
// at the previous barrier, so their LDS stores are issued at the top of half step h, in front of its 64 MFMAs, and ONE barrier
// per half step publishes them.  Loads are issued one half step before their stores.  Everything that does not depend on the
// half step is hoisted to the unit (column offset and its bounds test) or to the workgroup.
template <int NBA, bool AFF>
__global__ __launch_bounds__(256, 3) void conv_s2_wgrad_roll2_kernel(W2RArgs p) {
  constexpr int CL_T = 16 * NBA, LROW = CL_T * W2_LP;
  constexpr int HITEMS = 2 * 16 * WR_XQ, HPT = (HITEMS + 255) / 256;      // two high rows: 576 float4, 3 per thread
  constexpr int LITEMS = CL_T * W2_LQ;                                    // one low row: 256 / 128 float4
  static_assert(LITEMS <= 256, "one low item per thread");
  __shared__ __attribute__((aligned(16))) float smem[WR_SLOTS * W2_HSLOT + W2_RL * LROW];
  __shared__ float afftab[AFF ? 2 * CL_T : 1];
  // writes that do not apply (the third item of threads >= 64, the left-halo float4's E pair, a step past the unit's end) go
  // to the thread's own 16-byte slot here instead of being branched around: the half step stays ONE basic block, which is
  // what lets the staging instructions be scheduled between its MFMAs
  __shared__ __attribute__((aligned(16))) float sink_s[256 * 4];
  float* ring = smem;
  float* lbuf = smem + WR_SLOTS * W2_HSLOT;
  const int tid = threadIdx.x, lane = tid & 63, wa = tid >> 6;          // wa = tap row a of this wave
  float* sink = sink_s + 4 * tid;
  int bid = p.xcd ? gl_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int split = bid % p.S;
  bid /= p.S;
  const int ch_t = bid % p.tiles_ch, cl_t = bid / p.tiles_ch;
  const int ch0 = ch_t * 16, cl0 = cl_t * CL_T;
  const int H = 2 * p.Hl, W = 2 * p.Wl, hplane = H * W, lplane = p.Hl * p.Wl;

  // workgroup-level item descriptors: channel byte offset (or OOB), LDS offset | row of the pair << 16 | (q > 0) << 17 (or -1)
  int hlo[HPT];
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int e = tid + i * 256;
    const int q = e % WR_XQ, t = e / WR_XQ;
    const int ch = t & 15, k = t >> 4;
    hlo[i] = e < HITEMS ? ((ch * W2_HP + 2 * q) | (k << 16) | ((q > 0 ? 1 : 0) << 17)) : -1;
  }
  const int lq_ = tid % W2_LQ, lcl = tid / W2_LQ;
  const bool lvalid = tid < LITEMS;
  const int lch = (lvalid && cl0 + lcl < p.Cl) ? (cl0 + lcl) * lplane * 4 : WR_OOB;
  const int llo = lcl * W2_LP + 4 * lq_;

  f32x4 acc[4][NBA];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int m = 0; m < NBA; ++m) acc[b][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_off = (lane & 15) * W2_LP + 4 * (lane >> 4);
  const int b_off = (lane & 15) * W2_HP + 4 * (lane >> 4);
  float4 xr[2][HPT], lr[2];

  for (int u = split; u < p.units; u += p.S) {
    const int col = u % p.cols;
    const int t2 = u / p.cols;
    const int strip = t2 % p.strips, n = t2 / p.strips;
    const int X0 = col * W2_TWL, Y0 = strip * p.spu * W2_RL;
    const int nh = W2_RL * min(p.spu, p.Hl / W2_RL - strip * p.spu);     // half steps = low rows of this unit
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.high + (long long)n * p.Ch * hplane), 0, (unsigned)(p.Ch * hplane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.low + (long long)n * p.Cl * lplane), 0, (unsigned)(p.Cl * lplane * 4), 0x00020000);
    // unit-level: column offset folded in, its bounds test done
    int hu[HPT];      // (channel offset recomputed per unit: three registers less across the half-step loop)
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      const int e = tid + i * 256;
      const int ch = (e / WR_XQ) & 15, vx = 2 * X0 - 4 + 4 * (e % WR_XQ);
      hu[i] = (e < HITEMS && ch0 + ch < p.Ch && (unsigned)vx < (unsigned)W) ? (ch0 + ch) * hplane * 4 + vx * 4 : WR_OOB;
    }
    const int lu = lch != WR_OOB ? lch + (Y0 * p.Wl + X0 + 4 * lq_) * 4 : WR_OOB;
    const int vy0 = 2 * Y0 - 1, W4 = W * 4, Wl4 = p.Wl * 4;

    // two register sets: a half step stores set h & 1 (loaded TWO half steps ago: ~4000 MFMA cycles of cover for the loads;
    // with one set and one half step of distance the first store piece waited for HBM) and refills it for half step h + 2
    auto load_h2_item = [&](int r0, bool on, int set, int i) {   // item i of high rel rows r0, r0 + 1 (rel row r = high row 2 Y0 - 1 + r)
      const int vy = vy0 + r0 + ((hlo[i] >> 16) & 1);
      const int off = (on && hu[i] != WR_OOB && (unsigned)vy < (unsigned)H) ? hu[i] + vy * W4 : WR_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_h, off, 0, 0);
      xr[set][i] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto load_h2 = [&](int r0, bool on, int set) {
#pragma unroll
      for (int i = 0; i < HPT; ++i) load_h2_item(r0, on, set, i);
    };
    auto store_h2_item = [&](int r0, int set, int i) {         // r0 even: slots r0 % 6 and r0 % 6 + 1
      const int s0 = (r0 % WR_SLOTS) * W2_HSLOT;
      const bool valid = hlo[i] != -1, inner = valid && (hlo[i] & (1 << 17));
      float* d = ring + s0 + ((hlo[i] >> 16) & 1) * W2_HSLOT + (hlo[i] & 0xffff);
      *reinterpret_cast<float2*>(inner ? d - 2 : sink) = float2{xr[set][i].x, xr[set][i].z};      // E[2q-2], E[2q-1]
      *(inner ? d + 35 : sink + 2) = xr[set][i].y;                                                // O'[2q-1]
      *(valid ? d + 36 : sink + 3) = xr[set][i].w;                                                // O'[2q]
    };
    auto store_h2 = [&](int r0, int set) {
#pragma unroll
      for (int i = 0; i < HPT; ++i) store_h2_item(r0, set, i);
    };
    auto load_l1 = [&](int row, bool on, int set) {     // low row Y0 + row
      const int off = (on && lu != WR_OOB) ? lu + row * Wl4 : WR_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_l, off, 0, 0);
      lr[set] = float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    };
    auto store_l1 = [&](int slot, int set) {
      float4 v = lr[set];
      if constexpr (AFF) {
        const bool ok = lch != WR_OOB;
        const float sv = ok ? afftab[ok ? lcl : 0] : 0.f, tv = ok ? afftab[ok ? CL_T + lcl : 0] : 0.f;
        v.x = fmaf(v.x, sv, tv); v.y = fmaf(v.y, sv, tv); v.z = fmaf(v.z, sv, tv); v.w = fmaf(v.w, sv, tv);
      }
      *reinterpret_cast<float4*>(lvalid ? lbuf + slot * LROW + llo : sink) = v;
    };

    if constexpr (AFF) {
      if (tid < 2 * CL_T) {
        const int c = cl0 + (tid % CL_T);
        afftab[tid] = c < p.Cl ? (tid < CL_T ? p.aff_s : p.aff_t)[(long long)n * p.Cl + c] : 0.f;
      }
      __syncthreads();
    }
    load_h2(0, true, 0);
    load_l1(0, true, 0);
    load_h2(2, true, 1);
    store_h2(0, 0);
    store_l1(0, 0);
    store_h2(2, 1);
    load_h2(4, true, 0);           // stored in half step 0
    load_l1(1, true, 0);
    load_h2(6, nh > 2, 1);         // stored in half step 1
    load_l1(2, nh > 2, 1);
    __syncthreads();
    // one half step; `par` = h & 1 is a literal at both call sites (register-set index, low slot)
    auto half_step = [&](int h, int par) {
      const int sbase = ((2 * h + wa) % WR_SLOTS) * W2_HSLOT + b_off;
      const float* lrow_s = lbuf + par * LROW + a_off;
      float4 av[2][NBA], e4[2], o4[2];
      float2 e2[2], o2[2];
      auto fetch = [&](int g, int buf) {
        const float* src = ring + sbase + 16 * g;
        e4[buf] = *reinterpret_cast<const float4*>(src);
        e2[buf] = *reinterpret_cast<const float2*>(src + 4);
        o4[buf] = *reinterpret_cast<const float4*>(src + 36);
        o2[buf] = *reinterpret_cast<const float2*>(src + 40);
#pragma unroll
        for (int m = 0; m < NBA; ++m) av[buf][m] = *reinterpret_cast<const float4*>(lrow_s + m * 16 * W2_LP + 16 * g);
      };
      fetch(0, 0);
      fetch(1, 1);
      // The staging work of the half step in eight pieces, one behind each group of 4 NBA MFMAs (order pinned): what the next
      // half step adds goes to LDS (past the unit's end: zeros into slots nobody reads again), then the same register set is
      // refilled for half step h + 2.  Issued in front of the MFMAs none of it is covered by this wave's own matrix work.
      static_assert(HPT == 3, "eight staging pieces");
      auto piece = [&](int k) {
        if (k < 3) store_h2_item(2 * h + 4, par, k);
        else if (k == 3) store_l1(par ^ 1, par);
        else if (k < 7) load_h2_item(2 * h + 8, h + 3 < nh, par, k - 4);
        else load_l1(h + 3, h + 3 < nh, par);
      };
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float eE[5] = {e4[g].x, e4[g].y, e4[g].z, e4[g].w, e2[g].x};
        const float eO[5] = {o4[g].x, o4[g].y, o4[g].z, o4[g].w, o2[g].x};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float bv[4] = {eO[q], eE[q], eO[q + 1], eE[q + 1]};
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int m = 0; m < NBA; ++m) {
              const float a = q == 0 ? av[g][m].x : (q == 1 ? av[g][m].y : (q == 2 ? av[g][m].z : av[g][m].w));
              acc[b][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[b], acc[b][m], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
          piece(g * 4 + q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    };
    for (int h = 0; h < nh; h += 2) {      // nh is even (two low rows per step of the plan)
      half_step(h, 0);
      half_step(h + 1, 1);
    }
  }
  float* dst = p.part + (long long)split * 16 * p.Cl * p.Ch;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int m = 0; m < NBA; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = cl0 + m * 16 + (lane >> 4) * 4 + r, ch = ch0 + (lane & 15);
        if (cl < p.Cl && ch < p.Ch) dst[((long long)(wa * 4 + b) * p.Cl + cl) * p.Ch + ch] = acc[b][m][r];
      }
}

inline bool wr_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// Thin 3x3 "same" convolutions on 64-pixel-aligned planes.  Pointers may be null (workspace / plan queries).
bool gl_wgrad_roll_supported(int N, int Cin, int Cout, int H, int W, int ks, int pad, int up, const void* x,
                             const void* gy) {
  if (ks != 3 || pad != 1 || up || N <= 0 || Cin <= 0 || Cout <= 0) return false;
  if (Cin > 32 || Cout > 32) return false;
  if (H % WR_ROWS != 0 || W % WR_TW != 0 || H < 8) return false;
  if ((long long)Cin * H * W * 4 >= 0x7fffffffLL || (long long)Cout * H * W * 4 >= 0x7fffffffLL) return false;
  if ((x && !wr_aligned16(x)) || (gy && !wr_aligned16(gy))) return false;
  return true;
}

static void wr_plan(WRArgs& a) {
  a.cols = a.W / WR_TW;
  const int steps = a.H / WR_ROWS;
  a.tiles_ci = (a.Cin + 15) / 16;
  a.tiles_co = (a.Cout + 15) / 16;
  const int base = a.tiles_ci * a.tiles_co;
  const int target = (768 + base - 1) / base;          // one full round of three workgroups per CU
  // Steps per unit: long strips amortise a unit's un-overlapped prologue (six row loads before the first MFMA), short
  // ones balance the units over the workgroups.  Measured (tools/wgrad_bench.py, MI355X): 64 is best while every
  // workgroup still gets >= 2.5 units (16 -> 16 at 1024^2: 0.746 of peak vs 0.735 at 16), 16 otherwise.
  int spu = 16;
  if (const char* e = GL_ENV_ONCE("GANLAB_WR_SPU")) {       // tuning knob (tools/wgrad_bench.py)
    spu = atoi(e) > 0 ? atoi(e) : 16;
  } else {
    for (int cand = 64; cand > 16; cand >>= 1) {
      const long long units = (long long)a.N * a.cols * ((steps + cand - 1) / cand);
      if (steps >= cand && 2 * units >= 5LL * target) { spu = cand; break; }
    }
  }
  a.spu = steps < spu ? steps : spu;
  a.strips = (steps + a.spu - 1) / a.spu;
  a.units = a.N * a.cols * a.strips;
  int S = target;
  if (S > a.units) S = a.units;
  a.S = S < 1 ? 1 : S;
  // XCD-aware mapping is OFF: it cuts the L2-miss traffic (FETCH_SIZE x2: 1.47-1.63x -> 1.14-1.16x the algorithmic
  // bytes, profiles/r02_wgrad_roll_pmc.txt) but the thin layers ran 8-10 % SLOWER with it in a same-process A/B
  // (tools/wgrad_bench.py; thick layers: no difference) - the halo lines the neighbours miss are served by the
  // Infinity Cache anyway.  GANLAB_WR_XCD=1 turns it on.
  { const char* e = GL_ENV_ONCE("GANLAB_WR_XCD"); a.xcd = (e && e[0] == '1') ? 1 : 0; }
}

// number of partial-sum slots ([Cout][Cin][9] floats each) the launch writes
int gl_wgrad_roll_slots(int N, int Cin, int Cout, int H, int W) {
  WRArgs a{};
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  wr_plan(a);
  return a.S;
}

int gl_wgrad_roll_launch(const float* x, const float* gy, float* part, int N, int Cin, int Cout, int H, int W,
                         hipStream_t st, const float* aff_s, const float* aff_t) {
  WRArgs a{};
  a.x = x; a.gy = gy; a.part = part; a.aff_s = aff_s; a.aff_t = aff_t;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  wr_plan(a);
  const long long grid = (long long)a.tiles_co * a.tiles_ci * a.S;
  if (aff_s != nullptr) GL_LAUNCH(conv_wgrad_roll_kernel<true>, dim3((unsigned)grid), dim3(256), 0, st, a);
  else GL_LAUNCH(conv_wgrad_roll_kernel<false>, dim3((unsigned)grid), dim3(256), 0, st, a);
  return GL_CHECK_LAUNCH();
}

// ---- stride-2 (16-tap) variant --------------------------------------------------------------------------------------
bool gl_wgrad_s2_roll_supported(int N, int Cl, int Ch, int Hl, int Wl, const void* low, const void* high) {
  if (N <= 0 || Cl <= 0 || Ch <= 0) return false;
  if (Wl % W2_TWL != 0 || Hl % W2_RL != 0 || Hl < 4) return false;
  if ((long long)Ch * Hl * Wl * 16 >= 0x7fffffffLL || (long long)Cl * Hl * Wl * 4 >= 0x7fffffffLL) return false;
  if ((low && !wr_aligned16(low)) || (high && !wr_aligned16(high))) return false;
  return true;
}

static void w2r_plan(W2RArgs& a) {
  a.cols = a.Wl / W2_TWL;
  const int steps = a.Hl / W2_RL;
  const int nba = a.Cl > 16 ? 2 : 1;
  a.tiles_cl = (a.Cl + 16 * nba - 1) / (16 * nba);
  a.tiles_ch = (a.Ch + 15) / 16;
  const long long base = (long long)a.tiles_cl * a.tiles_ch;
  const int target = (int)((768 + base - 1) / base);
  int spu = 16;
  if (const char* e = GL_ENV_ONCE("GANLAB_WR_SPU")) {
    spu = atoi(e) > 0 ? atoi(e) : 16;
  } else {
    for (int cand = 64; cand > 16; cand >>= 1) {
      const long long units = (long long)a.N * a.cols * ((steps + cand - 1) / cand);
      if (steps >= cand && 2 * units >= 5LL * target) { spu = cand; break; }
    }
  }
  a.spu = steps < spu ? steps : spu;
  a.strips = (steps + a.spu - 1) / a.spu;
  a.units = a.N * a.cols * a.strips;
  int S = target;
  if (S > a.units) S = a.units;
  a.S = S < 1 ? 1 : S;
  // XCD-aware mapping is OFF: it cuts the L2-miss traffic (FETCH_SIZE x2: 1.47-1.63x -> 1.14-1.16x the algorithmic
  // bytes, profiles/r02_wgrad_roll_pmc.txt) but the thin layers ran 8-10 % SLOWER with it in a same-process A/B
  // (tools/wgrad_bench.py; thick layers: no difference) - the halo lines the neighbours miss are served by the
  // Infinity Cache anyway.  GANLAB_WR_XCD=1 turns it on.
  { const char* e = GL_ENV_ONCE("GANLAB_WR_XCD"); a.xcd = (e && e[0] == '1') ? 1 : 0; }
}

int gl_wgrad_s2_roll_slots(int N, int Cl, int Ch, int Hl, int Wl) {
  W2RArgs a{};
  a.N = N; a.Cl = Cl; a.Ch = Ch; a.Hl = Hl; a.Wl = Wl;
  w2r_plan(a);
  return a.S;
}

int gl_wgrad_s2_roll_launch(const float* low, const float* high, float* part, int N, int Cl, int Ch, int Hl, int Wl,
                            hipStream_t st, const float* aff_s, const float* aff_t) {
  W2RArgs a{};
  a.low = low; a.high = high; a.part = part; a.aff_s = aff_s; a.aff_t = aff_t;
  a.N = N; a.Cl = Cl; a.Ch = Ch; a.Hl = Hl; a.Wl = Wl;
  w2r_plan(a);
  const long long grid = (long long)a.tiles_cl * a.tiles_ch * a.S;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  static const int half = [] { const char* e = getenv("GANLAB_W2R_HALF"); return (e && e[0] == '0') ? 0 : 1; }();
  if (half) {            // half-step kernel (default; GANLAB_W2R_HALF=0: the whole-step kernel, same-process A/B)
    if (aff_s != nullptr) {
      if (a.Cl > 16) GL_LAUNCH((conv_s2_wgrad_roll2_kernel<2, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
      else GL_LAUNCH((conv_s2_wgrad_roll2_kernel<1, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
    } else {
      if (a.Cl > 16) GL_LAUNCH((conv_s2_wgrad_roll2_kernel<2, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
      else GL_LAUNCH((conv_s2_wgrad_roll2_kernel<1, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
    }
    return GL_CHECK_LAUNCH();
  }
  if (aff_s != nullptr) {
    if (a.Cl > 16) GL_LAUNCH((conv_s2_wgrad_roll_kernel<2, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
    else GL_LAUNCH((conv_s2_wgrad_roll_kernel<1, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
  } else {
    if (a.Cl > 16) GL_LAUNCH((conv_s2_wgrad_roll_kernel<2, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
    else GL_LAUNCH((conv_s2_wgrad_roll_kernel<1, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
  }
  return GL_CHECK_LAUNCH();
}
