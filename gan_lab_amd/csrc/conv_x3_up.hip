// The TRANSPOSED 4x4 stride-2 form of the split-product convolution (see conv_x3.hip for the arithmetic): the forward of an up
// layer conv3x3(Upsample2x(x)) (stylegan/architectures.py:292-334; also with the affine of a deferred InstanceNorm applied on
// load) and the input gradient of a pooled layer AvgPool2(conv3x3(x)) (progan/architectures.py:261-284); the exact-fp32 form is
// the T kernel of csrc/conv_s2.hip.  Output pixel (2y + py, 2x + px) is a 2 x 2-tap convolution of the LOW-resolution input:
// rows y - 1 + py + ty, columns x - 1 + px + tx (ty, tx = 0, 1), with the 3x3 weights of the rows / columns that land on a tap
// summed in fp32 at pack time - 16 instead of 36 products per low-resolution pixel.
// Workgroup: 512 threads, virtual tile = 8 x 16 low-resolution pixels x 64 output channels x ONE row parity py x BOTH column
// parities: wave (wm = 0..3, wn = 0,1) owns low-res rows 2wm, 2wm + 1, channels 32wn .. + 31 and the four blocks nn = (px, 16
// channels), so a lane ends up with 4 + 4 horizontally interleaved output pixels: two 16-byte stores (the first version of this
// form kept one parity per tile and stored 4-byte pieces at stride 8: 18 % of its time).  k-step = the 4 taps (lane group kg =
// (ty, tx)) x 8 input channels; the patch (10 x 18 units) is staged in halves of 16 channels = 2 k-steps that ring through three
// slots; the operand fragments of column parity px = 1 are those of px = 0 one unit to the right.  Weights: one k-step image
// [plane][tap][px][co 64] (24 KB) per LDS-DMA stage, double buffered; one barrier per k-step (48 MFMAs per wave).
#include "common.h"

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XU_CO = 64;                       // output channels per workgroup (x 2 column parities = 128 GEMM columns)
constexpr int XU_PL = 192;                      // units (16 B) of one bf16 plane of an 8-channel group: 10 rows x 18 = 180, padded
constexpr int XU_G = 3 * XU_PL;                 // an 8-channel group: three planes
constexpr int XU_HALF = 2 * XU_G;               // a half slot (16 channels): 1152 units
constexpr int XU_WSTEP = 3 * 4 * 128;           // weights of a k-step: [plane][tap kg][px][co 64] = 1536 units
constexpr int XU_WOFF = 3 * XU_HALF;
constexpr int XU_LDS = 3 * XU_HALF + 2 * XU_WSTEP;   // 6528 units = 104,448 bytes

#define XU_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define XU_ACC8(a) "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3])
#define XU_MFMA_DRAIN(a) asm volatile("s_nop 15\n\ts_nop 15" : XU_ACC8(a))
#define XU_VALU_SETTLE(a) asm volatile("s_nop 7\n\ts_nop 7" : XU_ACC8(a))

__device__ __forceinline__ u32x4 xu_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu,
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void xu_ld(f32x4& d, const u32x4& rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff));
}
template <int YOUNGER>
__device__ __forceinline__ void xu_ld_wait(f32x4 (&a)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(YOUNGER));
}
template <int YOUNGER>
__device__ __forceinline__ void xu_ld_wait(f32x4 (&a)[4], f32x4& s_, f32x4& t_) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(s_), "+v"(t_) : "n"(YOUNGER));
}
template <int YOUNGER>
__device__ __forceinline__ void xu_barrier() {
  if constexpr (YOUNGER == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if constexpr (YOUNGER == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  static_assert(YOUNGER == 0 || YOUNGER == 4 || YOUNGER == 6, "a half's staging issues four loads (six with the affine)");
}

struct XUArgs {
  const float* x;           // (N, CI, Hl, Wl)
  const u32x4* wp;
  const float* bias;
  float* y;                 // (N, CO, 2 Hl, 2 Wl)
  const float* aff_s;       // AFF: the conv reads x * aff_s[n][ci] + aff_t[n][ci] inside the image
  const float* aff_t;
  int N, CI, CO, Hl, Wl;
  int tiles_x, tiles_y, tiles_co, ntiles;
  float bias_scale, slope;
  int act;
};
struct XUTile { int n, oy0, ox0, co_t, py; };

// C32: the 32-output-channel form (64 -> 32 up layers, the input gradient of 32 -> 64 pooled layers).  The same 128 GEMM columns
// mean BOTH row parities x both column parities x 32 channels: wave wn owns row parity py = wn (its operand rows are one patch
// row further down, its weight columns the image's [py][co 32] half) - the tile index loses the parity, everything else is the
// 64-channel kernel.
template <bool AFF, bool C32>
__global__ __launch_bounds__(512) void conv_x3_up_kernel(XUArgs p) {
  constexpr int NLOADS = AFF ? 6 : 4;
  __shared__ __attribute__((aligned(16))) u32x4 lds[XU_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int l16 = lane & 15, kg = lane >> 4, lane16 = lane * 16;
  const int plane = p.Hl * p.Wl;                // input (low-resolution) plane
  const int nsteps = p.CI / 8;                  // k-steps per tile; two per half
  const int G = gridDim.x;

  auto decode = [&](int t) {
    XUTile c;
    c.co_t = t % p.tiles_co; t /= p.tiles_co;
    if constexpr (C32) c.py = 0;
    else { c.py = t & 1; t >>= 1; }
    c.ox0 = (t % p.tiles_x) * 16; t /= p.tiles_x;
    c.oy0 = (t % p.tiles_y) * 8;
    c.n = t / p.tiles_y;
    return c;
  };

  // ---- staging item of this thread (240 of them): channel group g, patch row r = 0..9 (input row oy0 - 1 + r), 4-column group
  //      cg = 0..5 (columns ox0 - 4 + 4 cg + i; patch column 4 cg - 3 + i must lie in 0..17), channel quad cq -------------------
  const bool a_item = tid < 240;
  int a_r, a_cg, a_gq;      // a_gq = g * 2 + cq
  {
    const int e = a_item ? tid : 0;
    const int cq = e & 1;
    int t = e >> 1;
    a_cg = t % 6; t /= 6;
    a_r = t % 10;
    a_gq = (t / 10) * 2 + cq;
  }
  const int cstride = plane * 4;
  f32x4 arA[4], arB[4], svA, tvA, svB, tvB;
  auto a_load_to = [&](f32x4 (&ar)[4], f32x4& a_sv, f32x4& a_tv, const XUTile& c, int half) {     // channels 16 half + 8 g + 4 cq + j
    const u32x4 rs = xu_rsrc(p.x + (long long)c.n * p.CI * plane, (unsigned)((long long)p.CI * plane * 4));
    int r = a_r;
    asm volatile("" : "+v"(r));
    const int iy = c.oy0 - 1 + r, ix = c.ox0 - 4 + 4 * a_cg;
    const bool ok = a_item && (unsigned)iy < (unsigned)p.Hl && (unsigned)ix < (unsigned)p.Wl;
    const int off = ok ? ((a_gq * 4) * plane + iy * p.Wl + ix) * 4 : (int)0x80000000;
    const int soff = half * 16 * plane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) xu_ld(ar[j], rs, off + j * cstride, soff);
    if constexpr (AFF) {       // an item outside the image reads zeros for s and t as well: 0 * 0 + 0 keeps the padding zero
      const unsigned tab = (unsigned)((long long)p.N * p.CI * 4);
      const u32x4 rss = xu_rsrc(p.aff_s, tab), rst = xu_rsrc(p.aff_t, tab);
      const int o = ok ? a_gq * 16 : (int)0x80000000;
      const int so = (c.n * p.CI + half * 16) * 4;
      xu_ld(a_sv, rss, o, so);
      xu_ld(a_tv, rst, o, so);
    }
  };
  auto a_store_from = [&](const f32x4 (&ar)[4], const f32x4& a_sv, const f32x4& a_tv, int slot, int i) {
    if (!a_item) return;
    int r = a_r;
    asm volatile("" : "+v"(r));
    const int c = 4 * a_cg - 3 + i;
    if (c < 0 || c > 17) return;
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = ar[j][i];
      if constexpr (AFF) v = fmaf(v, a_sv[j], a_tv[j]);
      h[j] = (__bf16)v;
      const float r1 = v - (float)h[j];
      m[j] = (__bf16)r1;
      l[j] = (__bf16)(r1 - (float)m[j]);
    }
    const int unit = slot * XU_HALF + (a_gq >> 1) * XU_G + r * 18 + c;
    unsigned char* dst = reinterpret_cast<unsigned char*>(lds + unit) + (a_gq & 1) * 8;
    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
    *reinterpret_cast<u32x2*>(dst + XU_PL * 16) = __builtin_bit_cast(u32x2, m);
    *reinterpret_cast<u32x2*>(dst + 2 * XU_PL * 16) = __builtin_bit_cast(u32x2, l);
  };
  auto a_load = [&](int set, const XUTile& c, int half) {
    if (set == 0) a_load_to(arA, svA, tvA, c, half); else a_load_to(arB, svB, tvB, c, half);
  };
  auto a_store_px = [&](int set, int slot, int i) {
    if (set == 0) a_store_from(arA, svA, tvA, slot, i); else a_store_from(arB, svB, tvB, slot, i);
  };
  auto a_wait = [&](int set, auto younger) {
    constexpr int Y = decltype(younger)::value;
    if (set == 0) { if constexpr (AFF) xu_ld_wait<Y>(arA, svA, tvA); else xu_ld_wait<Y>(arA); }
    else { if constexpr (AFF) xu_ld_wait<Y>(arB, svB, tvB); else xu_ld_wait<Y>(arB); }
  };

  // ---- weights: LDS-DMA, one k-step image per stage ----------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(p.wp), 0, (unsigned)((long long)p.tiles_co * (C32 ? 1 : 2) * nsteps * XU_WSTEP * 16), 0x00020000);
  auto w_dma = [&](const XUTile& c, int step, int buf) {
    const int soff = ((C32 ? c.co_t : c.co_t * 2 + c.py) * nsteps + step) * (XU_WSTEP * 16) + wv * 3072;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(lds + XU_WOFF + buf * XU_WSTEP + wv * 192 + i * 64),
                                               16, lane16, soff + i * 1024, 0, 0);
  };

  f32x4 accS[2][4], accH[2][4], accT[2][4];       // [low-res row m][nn = px * 2 + 16-channel block]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      accS[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; accT[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  XU_VALU_SETTLE(accS);
  XU_VALU_SETTLE(accH);

  // lane's patch unit for column parity 0: tap (ty, tx) = (kg >> 1, kg & 1); rows 2 wm + m + py + ty, columns l16 + px + tx
  const int laneA = (2 * wm + (kg >> 1) + (C32 ? wn : 0)) * 18 + l16 + (kg & 1);
  const int laneB = XU_WOFF + kg * 128 + wn * 32 + l16;

  bf16x8 aF[2][2][3];       // [column parity px][row m][plane]
  bf16x8 bF[2][3];          // [set][plane] of one block nn
  auto a_frags = [&](int off, int px) {       // off: slot * XU_HALF + g * XU_G + py * 18
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) aF[px][m][pl] = __builtin_bit_cast(bf16x8, lds[laneA + off + px + pl * XU_PL + m * 18]);
  };
  auto b_frags = [&](int buf, int nn, int set) {     // block nn = (px = nn >> 1, channels 16 (nn & 1) ..)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      bF[set][pl] = __builtin_bit_cast(bf16x8, lds[laneB + buf * XU_WSTEP + pl * 4 * 128 + (nn >> 1) * 64 + (nn & 1) * 16]);
  };

  int tile = gl_xcd_remap(blockIdx.x, G);
  if (tile >= p.ntiles) return;
  XUTile cur = decode(tile);
  int sa = 0, sb = 1, sc = 2;       // ring slots: the half being multiplied, the next one, the one after

  // ---- prologue: half 0 in LDS, half 1 in registers (set B), weight steps 0 (landed) and 1 (in flight) ---------------------------
  w_dma(cur, 0, 0);
  a_load(0, cur, 0);
  a_wait(0, std::integral_constant<int, 0>{});
#pragma unroll
  for (int i = 0; i < 4; ++i) { a_store_px(0, sa, i); __builtin_amdgcn_sched_barrier(0); }
  xu_barrier<0>();
  w_dma(cur, 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  a_load(1, cur, 1);
  a_wait(1, std::integral_constant<int, 0>{});      // (once per workgroup: the loop's counted wait assumes a DMA behind the loads)
  int offA = sa * XU_HALF + cur.py * 18;             // k-step 0: group g = 0
  a_frags(offA, 0);
  b_frags(0, 0, 0);

  int s0 = 0, gs = 0;               // k-step within the tile; k-steps since the kernel started (weight-buffer parity)
  for (;;) {
    const int ntile = tile + G;
    const bool nvalid = ntile < p.ntiles;
    const XUTile nxt = decode(nvalid ? ntile : tile);
    for (; s0 < nsteps; s0 += 8) {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int st = s0 + s8, g = s8 & 1;               // this k-step: half st >> 1, channel group g
        const int h = st >> 1;
        const int buf = (gs + s8) & 1;
        const int setn = ((s8 >> 1) + 1) & 1;              // staging register set of half h + 1 (s0 is a multiple of 8)
        const bool next_h = 2 * (h + 1) < nsteps || nvalid;        // a half follows this one
        const bool has_next = g == 0 || next_h;
        const bool wrap = g == 1 && 2 * (h + 1) >= nsteps;         // the next k-step opens the next tile
        // next k-step's patch offset: group 1 of this half, or group 0 of the next half (row parity of the tile it belongs to)
        const int offN = g == 0 ? sa * XU_HALF + XU_G + cur.py * 18 : sb * XU_HALF + (wrap ? nxt.py : cur.py) * 18;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          if (nn < 3) b_frags(buf, nn + 1, (nn + 1) & 1);
          if (nn == 0) a_frags(offA, 1);                   // column parity 1: one unit to the right
          if (nn == 3) {
            // ---- the step's barrier: the next k-step's weights are visible behind it; this step's buffer is free --------------
            if (g == 1 && (2 * (h + 2) < nsteps || nvalid)) xu_barrier<NLOADS>(); else xu_barrier<0>();   // (half h + 2's loads: step g = 0)
            {
              const int st2 = st + 2;                      // weights two k-steps on
              if (st2 < nsteps) w_dma(cur, st2, buf);
              else if (nvalid) w_dma(nxt, st2 - nsteps, buf);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) {                                  // half h + 2 requested (two k-steps before its first use)
              if (2 * (h + 2) < nsteps) a_load(setn ^ 1, cur, h + 2);
              else if (nvalid) a_load(setn ^ 1, nxt, h + 2 - nsteps / 2);
            }
            if (has_next) { a_frags(offN, 0); b_frags(buf ^ 1, 0, 0); }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            XU_MFMA(accS[m][nn], aF[nn >> 1][m][2], bF[nn & 1][0]);
            XU_MFMA(accS[m][nn], aF[nn >> 1][m][0], bF[nn & 1][2]);
            XU_MFMA(accS[m][nn], aF[nn >> 1][m][1], bF[nn & 1][1]);
            XU_MFMA(accS[m][nn], aF[nn >> 1][m][1], bF[nn & 1][0]);
            XU_MFMA(accS[m][nn], aF[nn >> 1][m][0], bF[nn & 1][1]);
            XU_MFMA(accH[m][nn], aF[nn >> 1][m][0], bF[nn & 1][0]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if ((nn == 1 || nn == 2) && next_h) {
            // half h + 1 goes to LDS, one pixel of every item behind blocks 1 and 2 of its two k-steps (behind its loads: the DMA
            // of the k-step in between)
            if (g == 0 && nn == 1) a_wait(setn, std::integral_constant<int, 3>{});
            a_store_px(setn, sb, g * 2 + nn - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        offA = offN;
        if (g == 1) { const int t_ = sa; sa = sb; sb = sc; sc = t_; }
      }
      gs += 8;
      // 64 channels x 4 taps = 256 terms: close the hi*hi chain
      XU_MFMA_DRAIN(accH);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) { accT[m][nn] += accH[m][nn]; accH[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      XU_VALU_SETTLE(accH);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: output row 2 (oy0 + 2wm + m) + py, columns 2 (ox0 + 4kg) .. + 7: the two column parities interleaved --------------
    {
      XU_MFMA_DRAIN(accS);
      typedef const __attribute__((address_space(4))) XUArgs* XUArgsK;
      unsigned long long kpi = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kpi));
      const XUArgsK kp = (XUArgsK)kpi;
      float* const y = kp->y;
      const float* const bias = kp->bias;
      const float bias_scale = kp->bias_scale, slope = kp->slope;
      const int act = kp->act;
      const int oW = 2 * p.Wl;
      const long long ib = (long long)cur.n * p.CO * plane * 4;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int co = C32 ? cur.co_t * 32 + cb * 16 + l16 : cur.co_t * XU_CO + wn * 32 + cb * 16 + l16;
        const float bv = bias != nullptr ? bias[co] * bias_scale : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          f32x4 v0 = accT[m][cb] + accS[m][cb], v1 = accT[m][2 + cb] + accS[m][2 + cb];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float f0 = v0[r] + bv, f1 = v1[r] + bv;
            if (act == GANLAB_ACT_LRELU) { f0 = gl_lrelu(f0, slope); f1 = gl_lrelu(f1, slope); }
            v0[r] = f0; v1[r] = f1;
          }
          float* dst = y + ib + (long long)co * plane * 4 + (long long)(2 * (cur.oy0 + 2 * wm + m) + (C32 ? wn : cur.py)) * oW + 2 * (cur.ox0 + 4 * kg);
          *reinterpret_cast<f32x4*>(dst) = f32x4{v0[0], v1[0], v0[1], v1[1]};
          *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v0[2], v1[2], v0[3], v1[3]};
          accT[m][cb] = f32x4{0.f, 0.f, 0.f, 0.f}; accS[m][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
          accT[m][2 + cb] = f32x4{0.f, 0.f, 0.f, 0.f}; accS[m][2 + cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      XU_VALU_SETTLE(accS);
    }
    if (!nvalid) break;
    tile = ntile;
    cur = nxt;
    s0 = 0;
  }
}

__global__ void x3_up_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int up, float scale) {
  const int CO = up ? Cout : Cin, CI = up ? Cin : Cout;        // GEMM roles: up layer's forward (up = 1), pooled layer's input gradient
  const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (e >= (long long)CO * CI) return;
  int ci, co;
  if ((CO & 63) == 0) {
    const int col = (int)(e & 63);
    const long long t = e >> 6;
    ci = (int)(t % CI);
    co = (int)(t / CI) * 64 + col;
  } else {
    co = (int)(e % CO);
    ci = (int)(e / CO);
  }
  const float* w9 = up ? w + ((long long)co * Cin + ci) * 9 : w + ((long long)ci * Cin + co) * 9;
  gl_x3_up_pack_position(w9, up, scale, out, CI, CO, ci, co);
}

// the transposed form takes: an up layer's forward (dgrad = 0) or a pooled layer's input gradient (dgrad = 1)
bool xu_ok(const ganlab_conv_geom* g, int dgrad) {
  if (g == nullptr || g->ks != 3 || g->pad != 1 || g->N <= 0) return false;
  if (dgrad ? !(g->pool == 1 && g->up == 0) : !(g->up == 1 && g->pool == 0)) return false;
  const int CI = dgrad ? g->Cout : g->Cin, CO = dgrad ? g->Cin : g->Cout;
  if (dgrad && ((g->Hin | g->Win) & 1)) return false;
  const int Hl = dgrad ? g->Hin / 2 : g->Hin, Wl = dgrad ? g->Win / 2 : g->Win;      // the up layer's input; the pooled layer's OUTPUT
  if ((long long)CI * Hl * Wl * 4 > 0x7fffffffLL || (long long)g->N * CI * 4 > 0x7fffffffLL) return false;
  return CI % 64 == 0 && CO % 32 == 0 && Hl % 8 == 0 && Wl % 16 == 0;
}

int xu_launch(bool aff, XUArgs a, hipStream_t st) {
  const bool c32 = a.CO % XU_CO != 0;      // the 32-channel form (gl_x3_up_pack_position packs by the same rule)
  a.tiles_x = a.Wl / 16; a.tiles_y = a.Hl / 8; a.tiles_co = c32 ? a.CO / 32 : a.CO / XU_CO;
  const int images = a.tiles_co * (c32 ? 1 : 2);
  const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y * images;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL || (long long)images * (a.CI / 8) * XU_WSTEP * 16 > 0xffffffffLL) return GANLAB_EINVAL;
  a.ntiles = (int)ntiles;
  const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
  if (c32) {
    if (aff) GL_LAUNCH((conv_x3_up_kernel<true, true>), dim3(grid), dim3(512), 0, st, a);
    else GL_LAUNCH((conv_x3_up_kernel<false, true>), dim3(grid), dim3(512), 0, st, a);
  } else {
    if (aff) GL_LAUNCH((conv_x3_up_kernel<true, false>), dim3(grid), dim3(512), 0, st, a);
    else GL_LAUNCH((conv_x3_up_kernel<false, false>), dim3(grid), dim3(512), 0, st, a);
  }
  return GL_CHECK_LAUNCH();
}

XUArgs xu_args(const float* x, const void* wp, const float* bias, float* y, int N, int CI, int CO, int Hl, int Wl, float bias_scale,
               int act, float slope) {
  XUArgs a{};
  a.x = x; a.wp = reinterpret_cast<const u32x4*>(wp); a.bias = bias; a.y = y;
  a.N = N; a.CI = CI; a.CO = CO; a.Hl = Hl; a.Wl = Wl;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  return a;
}

}  // namespace

extern "C" {

/* ---- the stride-2 fused layers' transposed form (ganlab_conv_s2_fwd_f32 with up = 1, ganlab_conv_s2_dgrad_f32 with pool = 1) ---- */
int ganlab_conv_s2_x3_supported(const ganlab_conv_geom* g, int dgrad) { return xu_ok(g, dgrad) ? 1 : 0; }

/* `up`: 1 = an up layer's forward weights, 0 = a pooled layer's input-gradient weights; 48*Cout*Cin bf16 elements */
long long ganlab_conv_s2_x3_pack(const float* w, void* out, int Cout, int Cin, int up, float scale, void* stream) {
  if (Cout <= 0 || Cin <= 0 || (up != 0 && up != 1)) return GANLAB_EINVAL;
  if ((up ? Cin : Cout) % 64 != 0 || (up ? Cout : Cin) % 32 != 0) return GANLAB_EINVAL;      // contraction channels, GEMM columns
  const long long n = 48LL * Cout * Cin;       // 16 taps x 3 planes
  if (out == nullptr) return n;
  if (w == nullptr) return GANLAB_EINVAL;
  const long long positions = (long long)Cout * Cin;
  GL_LAUNCH(x3_up_pack_kernel, dim3((unsigned)((positions + 255) / 256)), dim3(256), 0, gl_stream(stream), w,
            reinterpret_cast<__bf16*>(out), Cout, Cin, up, scale);
  const int st = GL_CHECK_LAUNCH();
  return st != GANLAB_OK ? st : n;
}

int ganlab_conv_s2_fwd_x3(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g, float bias_scale,
                          int act, float slope, void* stream) {
  if (!xu_ok(g, 0) || x == nullptr || wp == nullptr || y == nullptr) return GANLAB_EINVAL;
  return xu_launch(false, xu_args(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin, g->Win, bias_scale, act, slope), gl_stream(stream));
}

int ganlab_conv_s2_fwd_aff_x3(const float* x, const void* wp, const float* aff_s, const float* aff_t, const float* bias, float* y,
                              const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream) {
  if (!xu_ok(g, 0) || x == nullptr || wp == nullptr || y == nullptr || aff_s == nullptr || aff_t == nullptr) return GANLAB_EINVAL;
  XUArgs a = xu_args(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin, g->Win, bias_scale, act, slope);
  a.aff_s = aff_s; a.aff_t = aff_t;
  return xu_launch(true, a, gl_stream(stream));
}

/* input gradient of AvgPool2(conv3x3(x)): gy is (N, Cout, Hin / 2, Win / 2), gx (N, Cin, Hin, Win) */
int ganlab_conv_s2_dgrad_x3(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream) {
  if (!xu_ok(g, 1) || gy == nullptr || wp == nullptr || gx == nullptr) return GANLAB_EINVAL;
  return xu_launch(false, xu_args(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin / 2, g->Win / 2, 1.f, GANLAB_ACT_NONE, 0.f),
                   gl_stream(stream));
}

}  // extern "C"
