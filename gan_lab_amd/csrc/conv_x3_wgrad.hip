// Weight gradient of the thick plain 3x3 layers as split products on the bf16 matrix cores (see conv_x3.hip for the arithmetic:
// three bf16 planes per operand that sum to it exactly, six v_mfma_f32_16x16x32_bf16 per fp32 product, the five small
// products in their own accumulator set S, hi*hi in a chain H that is closed into T every 32 k-steps = 1024 terms).
//   gw[co][ci][ky][kx] = scale * sum_{n,y,x} gy[n,co,y,x] * X[n,ci,y+ky-1,x+kx-1]        (custom_layers.py:202-211, autograd)
// GEMM per tap: D[co][ci] += sum_k A[co][k] B[k][ci], k = 32 consecutive pixels px' of ONE input row (a k-step):
//   A = gy[co][y][px' - (kx-1)]  - the three column shifts are built in registers from the aligned 8-pixel unit and the two
//       neighbouring dwords (v_alignbit), so LDS holds gy once;
//   B = X[ci][y+ky-1][px']       - the three rows come from a ring of four row slots (every row is staged once and serves
//       three k-steps), rows outside the image from a zero slot.
// Workgroup: 512 threads, 32 output x 64 input channels x 9 taps; wave (wc = 0,1; wi = 0..3) owns 16 x 16 x 9 = nine
// accumulator tiles x three sets = 108 registers.  A workgroup walks a contiguous range of 32-pixel column strips, row by
// row (the k-loop runs on across strips and images), stages one gy row (with a 4-pixel halo) and one X row per k-step two
// steps ahead of their use (inline-asm loads, hand-counted waits), splits them on the way into LDS, and dumps T + H + S into
// its slot of the workspace at the end; x3w_reduce_kernel adds the slots in a fixed order.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XW_CO = 32, XW_CI = 64;
// LDS images are channel-major with an ODD-ish channel pitch: a fragment read takes 16 channels x 16 bytes (pitch 7 resp. 5 units
// = 112 / 80 bytes: the 16 lanes land on 16 disjoint groups of four banks), a staging store 8 bytes per lane along a channel's
// pixels.  (Pixel-major images - channels 16 bytes apart - made every staging store a 16- to 32-way bank conflict: 56 % of the
// kernel's LDS cycles, profiles/r05_x3_pmc.txt.)
constexpr int XW_GP = 7, XW_XP = 5;                // units per channel: gy 6 (px -8 .. 39) + 1, X 4 + 1
constexpr int XW_GPL = XW_CO * XW_GP, XW_XPL = XW_CI * XW_XP;      // one bf16 plane of a row image
constexpr int XW_GROW = 3 * XW_GPL;                // units (16 B) of a gy row image: [plane][co 32][unit 7]
constexpr int XW_XROW = 3 * XW_XPL;                // units of an X row image: [plane][ci 64][unit 5]
constexpr int XW_XOFF = 2 * XW_GROW;               // two gy buffers, then four ring slots and the zero slot
constexpr int XW_ZERO = 4;
constexpr int XW_LDS = 2 * XW_GROW + 5 * XW_XROW;  // 6144 units = 98,304 bytes
constexpr int XW_DUMP = 32;                        // k-steps per hi*hi chain

#define XW_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define XW_ACC9(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8])
#define XW_MFMA_DRAIN(a) asm volatile("s_nop 15\n\ts_nop 15" : XW_ACC9(a))
#define XW_VALU_SETTLE(a) asm volatile("s_nop 7\n\ts_nop 7" : XW_ACC9(a))

__device__ __forceinline__ u32x4 xw_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu,
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void xw_ld(f32x4& d, const u32x4& rs, int voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(rs));
}
template <int YOUNGER>
__device__ __forceinline__ void xw_ld_wait(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(YOUNGER));
}

struct XWArgs {
  const float* gy;
  const float* x;
  float* part;              // [slot][CO][CI][9]
  const float* aff_s;       // AFF: X = x * aff_s[n][ci] + aff_t[n][ci]
  const float* aff_t;
  int N, CI, CO, H, W;
  int hshift;               // H = 1 << hshift
  int strips_x, nstrips;    // W / 32, N * W / 32
  int tiles_ci, splits, sps;   // input-channel tiles, k-splits per channel-tile pair, strips per split
};

// split four values into planes: 8 bytes each
__device__ __forceinline__ void xw_split4(const f32x4& v, u32x2& h, u32x2& m, u32x2& l) {
  bf16x4 hh, mm, ll;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hh[j] = (__bf16)v[j];
    const float r1 = v[j] - (float)hh[j];
    mm[j] = (__bf16)r1;
    ll[j] = (__bf16)(r1 - (float)mm[j]);
  }
  h = __builtin_bit_cast(u32x2, hh); m = __builtin_bit_cast(u32x2, mm); l = __builtin_bit_cast(u32x2, ll);
}

template <bool AFF>
__global__ __launch_bounds__(512) void conv_x3_wgrad_kernel(XWArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[XW_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wv >> 2, wi = wv & 3;             // output-channel block (16) of the tile's 32, input-channel block of its 64
  const int l16 = lane & 15, kg = lane >> 4;

  int b = blockIdx.x;
  const int split = b % p.splits; b /= p.splits;
  const int ci_t = b % p.tiles_ci;
  const int co_t = b / p.tiles_ci;
  const int co0 = co_t * XW_CO, ci0 = ci_t * XW_CI;
  const int s_first = split * p.sps;
  const int s_count = min(p.sps, p.nstrips - s_first);
  const int F = s_count << p.hshift;               // k-steps of this workgroup (a multiple of 4)
  const int hmask = p.H - 1;
  const long long plane = (long long)p.H * p.W;

  const u32x4 rs_g = xw_rsrc(p.gy, (unsigned)((long long)p.N * p.CO * plane * 4));
  const u32x4 rs_x = xw_rsrc(p.x, (unsigned)((long long)p.N * p.CI * plane * 4));

  // ---- staging items: gy (co = tid / 10, 4 pixels x0 - 4 + 4q, q = tid % 10; threads >= 320 issue an out-of-range load so
  //      every wave's operation count is the same), X (ci = tid / 8, pixels x0 + 4q, q = tid % 8) -----------------------------
  const bool g_item = tid < 320;
  const int g_co = g_item ? tid / 10 : 0, g_q = g_item ? tid % 10 : 0;
  const int g_dst = ((g_co * XW_GP + ((g_q + 1) >> 1)) * 16 + ((g_q + 1) & 1) * 8);          // + plane * XW_GPL * 16
  const int x_ci = tid >> 3, x_q = tid & 7;
  const int x_dst = ((x_ci * XW_XP + (x_q >> 1)) * 16 + (x_q & 1) * 8);                      // + plane * XW_XPL * 16
  float a_s = 1.f, a_t = 0.f;                       // AFF: scale / shift of this thread's input channel in the current image
  int aff_n = -1;

  auto flat_pos = [&](int f, int& n, int& y, int& x0) {
    const int strip = s_first + (f >> p.hshift);
    y = f & hmask;
    n = strip / p.strips_x;
    x0 = (strip - n * p.strips_x) * 32;
  };
  auto g_off = [&](int f) {     // byte offset of this thread's gy item of flat row f (out of range: zeros)
    int n, y, x0;
    flat_pos(f, n, y, x0);
    const int px = x0 - 4 + 4 * g_q;
    const bool ok = g_item && f < F && px >= 0 && px < p.W;
    return ok ? (int)((((long long)n * p.CO + co0 + g_co) * plane + (long long)y * p.W + px) * 4) : (int)0x80000000;
  };
  auto x_off = [&](int f) {
    int n, y, x0;
    flat_pos(f, n, y, x0);
    return f < F ? (int)((((long long)n * p.CI + ci0 + x_ci) * plane + (long long)y * p.W + x0 + 4 * x_q) * 4) : (int)0x80000000;
  };
  auto g_store = [&](const f32x4& v, int buf) {
    if (!g_item) return;
    u32x2 h, m, l;
    xw_split4(v, h, m, l);
    unsigned char* d = reinterpret_cast<unsigned char*>(lds + buf * XW_GROW) + g_dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + XW_GPL * 16) = m;
    *reinterpret_cast<u32x2*>(d + 2 * XW_GPL * 16) = l;
  };
  auto x_store = [&](f32x4 v, int slot, int f) {
    if constexpr (AFF) {
      int n, y, x0;
      flat_pos(f, n, y, x0);
      if (n != aff_n && f < F) {                    // a new image: this channel's affine (once per strip at most)
        aff_n = n;
        a_s = p.aff_s[(long long)n * p.CI + ci0 + x_ci];
        a_t = p.aff_t[(long long)n * p.CI + ci0 + x_ci];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], a_s, a_t);
    }
    u32x2 h, m, l;
    xw_split4(v, h, m, l);
    unsigned char* d = reinterpret_cast<unsigned char*>(lds + XW_XOFF + slot * XW_XROW) + x_dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + XW_XPL * 16) = m;
    *reinterpret_cast<u32x2*>(d + 2 * XW_XPL * 16) = l;
  };

  f32x4 accS[9], accH[9], accT[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { accS[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accT[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  XW_VALU_SETTLE(accS);
  XW_VALU_SETTLE(accH);
  // fragment addresses (units): gy centre unit 1 + kg of row image `buf`; X unit kg of slot
  const int laneG = (wc * 16 + l16) * XW_GP + 1 + kg;
  const int laneX = XW_XOFF + (wi * 16 + l16) * XW_XP + kg;

  u32x4 gc[3];              // raw gy centre units of the NEXT k-step (planes)
  unsigned gp_[3], gn_[3];  // ... and the neighbouring dwords: last of the unit before, first of the unit after
  bf16x8 aS[3][3];          // [kx][plane] shifted gy fragments of this k-step
  bf16x8 bX[2][3];          // [set][plane] X fragments of a row
  auto g_frags = [&](int buf) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const int u = buf * XW_GROW + pl * XW_GPL + laneG;
      gc[pl] = lds[u];
      gp_[pl] = reinterpret_cast<const unsigned*>(lds + u - 1)[3];
      gn_[pl] = reinterpret_cast<const unsigned*>(lds + u + 1)[0];
    }
  };
  auto a_build = [&]() {     // kx = 0: gy[px' + 1]; kx = 1: gy[px']; kx = 2: gy[px' - 1]
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const u32x4 c = gc[pl];
      u32x4 up, dn;
      up[0] = __builtin_amdgcn_alignbit(c[1], c[0], 16); up[1] = __builtin_amdgcn_alignbit(c[2], c[1], 16);
      up[2] = __builtin_amdgcn_alignbit(c[3], c[2], 16); up[3] = __builtin_amdgcn_alignbit(gn_[pl], c[3], 16);
      dn[0] = __builtin_amdgcn_alignbit(c[0], gp_[pl], 16); dn[1] = __builtin_amdgcn_alignbit(c[1], c[0], 16);
      dn[2] = __builtin_amdgcn_alignbit(c[2], c[1], 16); dn[3] = __builtin_amdgcn_alignbit(c[3], c[2], 16);
      aS[0][pl] = __builtin_bit_cast(bf16x8, up);
      aS[1][pl] = __builtin_bit_cast(bf16x8, c);
      aS[2][pl] = __builtin_bit_cast(bf16x8, dn);
    }
  };
  auto x_frags = [&](int slot, int set) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bX[set][pl] = __builtin_bit_cast(bf16x8, lds[laneX + slot * XW_XROW + pl * XW_XPL]);
  };

  // ---- prologue: the zero slot; gy row 0 and X rows 0, 1 in LDS; rows (gy 1, X 2) and (gy 2, X 3) in flight ---------------------
  for (int i = tid; i < XW_XROW; i += 512) lds[XW_XOFF + XW_ZERO * XW_XROW + i] = u32x4{0u, 0u, 0u, 0u};
  f32x4 gA, xA, gB, xB;      // two load sets: (gy f + 1, X f + 2) of even / odd k-steps f
  xw_ld(gA, rs_g, g_off(0)); xw_ld(xA, rs_x, x_off(0));
  xw_ld(gB, rs_g, (int)0x80000000); xw_ld(xB, rs_x, x_off(1));
  xw_ld_wait<0>(gA, xA);
  xw_ld_wait<0>(gB, xB);
  g_store(gA, 0); x_store(xA, 0, 0); x_store(xB, 1, 1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  xw_ld(gA, rs_g, g_off(1)); xw_ld(xA, rs_x, x_off(2));
  xw_ld(gB, rs_g, g_off(2)); xw_ld(xB, rs_x, x_off(3));
  g_frags(0);
  x_frags(XW_ZERO, 0);       // k-step 0: row -1 of the first strip

  int since_dump = 0;
  for (int f0 = 0; f0 < F; f0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = f0 + u;
      const int y = f & hmask;
      // ring slots of X rows y (flat f) and y + 1 (f + 1); rows outside the image: the zero slot
      const int slot1 = u, slot2 = y == hmask ? XW_ZERO : (u + 1) & 3;
      a_build();
      __builtin_amdgcn_sched_barrier(0);
      // ---- ky = 0 (X row y - 1, fragments read at the end of the step before) -----------------------------------------------
      x_frags(slot1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        XW_MFMA(accS[kx], aS[kx][2], bX[0][0]); XW_MFMA(accS[kx], aS[kx][0], bX[0][2]); XW_MFMA(accS[kx], aS[kx][1], bX[0][1]);
        XW_MFMA(accS[kx], aS[kx][1], bX[0][0]); XW_MFMA(accS[kx], aS[kx][0], bX[0][1]); XW_MFMA(accH[kx], aS[kx][0], bX[0][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- staging: rows gy f + 1 and X f + 2 (requested two steps ago) go to LDS; behind them only the loads of the step
      //      before (2) are younger -----------------------------------------------------------------------------------------
      if (u & 1) { xw_ld_wait<2>(gB, xB); g_store(gB, (u + 1) & 1); x_store(xB, (u + 2) & 3, f + 2); }
      else { xw_ld_wait<2>(gA, xA); g_store(gA, (u + 1) & 1); x_store(xA, (u + 2) & 3, f + 2); }
      __builtin_amdgcn_sched_barrier(0);
      x_frags(slot2, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- ky = 1 (X row y) ----------------------------------------------------------------------------------------------------
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        XW_MFMA(accS[3 + kx], aS[kx][2], bX[1][0]); XW_MFMA(accS[3 + kx], aS[kx][0], bX[1][2]); XW_MFMA(accS[3 + kx], aS[kx][1], bX[1][1]);
        XW_MFMA(accS[3 + kx], aS[kx][1], bX[1][0]); XW_MFMA(accS[3 + kx], aS[kx][0], bX[1][1]); XW_MFMA(accH[3 + kx], aS[kx][0], bX[1][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // rows gy f + 3, X f + 4 requested
      if (u & 1) { xw_ld(gB, rs_g, g_off(f + 3)); xw_ld(xB, rs_x, x_off(f + 4)); }
      else { xw_ld(gA, rs_g, g_off(f + 3)); xw_ld(xA, rs_x, x_off(f + 4)); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // the rows stored above are visible behind it
      __builtin_amdgcn_sched_barrier(0);
      // next k-step's gy fragments and its row y' - 1 (= this row y, or the zero slot at the top of a strip)
      g_frags((u + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      // ---- ky = 2 (X row y + 1) ------------------------------------------------------------------------------------------------
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        XW_MFMA(accS[6 + kx], aS[kx][2], bX[0][0]); XW_MFMA(accS[6 + kx], aS[kx][0], bX[0][2]); XW_MFMA(accS[6 + kx], aS[kx][1], bX[0][1]);
        XW_MFMA(accS[6 + kx], aS[kx][1], bX[0][0]); XW_MFMA(accS[6 + kx], aS[kx][0], bX[0][1]); XW_MFMA(accH[6 + kx], aS[kx][0], bX[0][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      x_frags(y == hmask ? XW_ZERO : u, 0);      // row y' - 1 of the next k-step
      if (++since_dump == XW_DUMP) {
        since_dump = 0;
        XW_MFMA_DRAIN(accH);
#pragma unroll
        for (int t = 0; t < 9; ++t) { accT[t] += accH[t]; accH[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        XW_VALU_SETTLE(accH);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- this workgroup's partial sums: part[slot = split][co][ci][tap] -------------------------------------------------------------
  XW_MFMA_DRAIN(accS);
  XW_MFMA_DRAIN(accH);
  float* dst = p.part + (long long)split * p.CO * p.CI * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const f32x4 v = accT[t] + accH[t] + accS[t];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      dst[((long long)(co0 + wc * 16 + 4 * kg + r) * p.CI + ci0 + wi * 16 + l16) * 9 + t] = v[r];
  }
}

// gw = scale * sum over the slots, fixed order
__global__ void x3w_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long n, int slots, float scale) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < slots; k += 4) {
    s0 += part[(long long)k * n + i]; s1 += part[(long long)(k + 1) * n + i];
    s2 += part[(long long)(k + 2) * n + i]; s3 += part[(long long)(k + 3) * n + i];
  }
  for (; k < slots; ++k) s0 += part[(long long)k * n + i];
  out[i] = ((s0 + s1) + (s2 + s3)) * scale;
}

struct XWPlan { int splits, sps; };
bool xw_ok(const ganlab_conv_geom* g) {
  if (g == nullptr || g->ks != 3 || g->pad != 1 || g->up || g->pool || g->N <= 0) return false;
  const int H = g->Hin, W = g->Win;
  if (H < 4 || (H & (H - 1)) != 0 || W % 32 != 0) return false;
  if (g->Cin % XW_CI != 0 || g->Cout % XW_CO != 0) return false;
  const long long bytes = (long long)g->N * (g->Cin > g->Cout ? g->Cin : g->Cout) * H * W * 4;
  return bytes < 0x7fffffffLL;
}
XWPlan xw_plan(const ganlab_conv_geom* g) {
  const int pairs = (g->Cout / XW_CO) * (g->Cin / XW_CI);
  const int nstrips = g->N * (g->Win / 32);
  int splits = (512 + pairs - 1) / pairs;          // ~ two rounds of workgroups on the 256 CUs
  if (splits > nstrips) splits = nstrips;
  if (splits < 1) splits = 1;
  const int sps = (nstrips + splits - 1) / splits;
  splits = (nstrips + sps - 1) / sps;
  return XWPlan{splits, sps};
}

}  // namespace

extern "C" {

int ganlab_conv_wgrad_x3_supported(const ganlab_conv_geom* g) { return xw_ok(g) ? 1 : 0; }

size_t ganlab_conv_wgrad_x3_workspace(const ganlab_conv_geom* g) {
  if (!xw_ok(g)) return 0;
  return (size_t)xw_plan(g).splits * g->Cout * g->Cin * 9 * sizeof(float);
}

/* ganlab_conv_wgrad_f32 / ganlab_conv_wgrad_aff_f32 (aff_s, aff_t non-null: the x operand is x * aff_s[n][ci] + aff_t[n][ci]) */
int ganlab_conv_wgrad_x3(const float* gy, const float* x, const float* aff_s, const float* aff_t, float* gw,
                         const ganlab_conv_geom* g, float scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (!xw_ok(g)) return GANLAB_EUNSUPPORTED;
  if (!gy || !x || !gw || (aff_s == nullptr) != (aff_t == nullptr)) return GANLAB_EINVAL;
  const XWPlan pl = xw_plan(g);
  const long long nw = (long long)g->Cout * g->Cin * 9;
  if (!workspace || workspace_bytes < (size_t)pl.splits * nw * sizeof(float)) return GANLAB_EWORKSPACE;
  XWArgs a{};
  a.gy = gy; a.x = x; a.part = reinterpret_cast<float*>(workspace); a.aff_s = aff_s; a.aff_t = aff_t;
  a.N = g->N; a.CI = g->Cin; a.CO = g->Cout; a.H = g->Hin; a.W = g->Win;
  a.hshift = 0;
  while ((1 << a.hshift) < a.H) ++a.hshift;
  a.strips_x = a.W / 32; a.nstrips = a.N * a.strips_x;
  a.tiles_ci = a.CI / XW_CI; a.splits = pl.splits; a.sps = pl.sps;
  const long long grid = (long long)(a.CO / XW_CO) * a.tiles_ci * pl.splits;
  if (grid <= 0 || grid > 0x7fffffffLL) return GANLAB_EINVAL;
  hipStream_t st = gl_stream(stream);
  if (aff_s != nullptr) GL_LAUNCH(conv_x3_wgrad_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, a);
  else GL_LAUNCH(conv_x3_wgrad_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, a);
  GL_LAUNCH(x3w_reduce_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, (const float*)a.part, gw, nw, pl.splits, scale);
  return GL_CHECK_LAUNCH();
}

}  // extern "C"
