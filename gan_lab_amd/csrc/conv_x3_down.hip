// The STRIDED 4x4 stride-2 form of the split-product convolution (see conv_x3.hip for the arithmetic): the forward of a pooled
// layer AvgPool2(conv3x3(x)) (progan/architectures.py:261-284) and the input gradient of an up layer conv3x3(Upsample2x(x))
// (stylegan/architectures.py:292-334); the exact-fp32 form is the S kernel of csrc/conv_s2.hip.
//   out[co][oy][ox] = sum_ci sum_{a,b = 0..3} K4[co][ci][a][b] * in[ci][2 oy - 1 + a][2 ox - 1 + b],   K4 = M W M^T
// (16 instead of 36 products per low-resolution pixel; K4 is formed in fp32 at pack time exactly as ganlab_conv_s2_pack_f32
// forms it, then cut into three bf16 planes).
// The high-resolution patch is split by PARITY on its way into LDS (space to depth): tap (a, b) of output pixel (oy, ox) is
// element (oy + (a >> 1), ox + (b >> 1)) of parity plane (a & 1, b & 1), so every operand read is unit-stride.
// Workgroup: 512 threads, tile = 8 x 16 output pixels x 128 output channels; wave (wm = 0..3, wn = 0,1): output rows 2wm,
// 2wm + 1 x 64 channels = 8 accumulator tiles x (S, H, T) = 96 registers.  k-step = 4 taps (one tap row a; lane group kg =
// tap column b) x 8 input channels; the patch is staged in QUARTERS of 8 channels (4 parity planes of 9 x 17 units x 3 bf16
// planes = 30 KB) that ring through three slots, four k-steps per quarter; the weights of ONE k-step (24 KB) per LDS-DMA stage,
// double buffered: 141 KB of LDS, one barrier per k-step (48 MFMAs per wave).  Persistent workgroups, k-loop across tiles.
#include "common.h"

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XD_NT = 128;                      // output channels per workgroup
constexpr int XD_PP = 160;                      // units (16 B) of one parity plane of a quarter: 9 rows x 17 = 153, padded to 10 x 16
constexpr int XD_PL = 4 * XD_PP;                // one bf16 plane of a quarter: 4 parity planes
constexpr int XD_Q = 3 * XD_PL;                 // a quarter slot: 1920 units
constexpr int XD_WSTEP = 3 * 4 * XD_NT;         // weights of a k-step: [plane][tap column b][co 128] = 1536 units
constexpr int XD_WOFF = 3 * XD_Q;
constexpr int XD_LDS = 3 * XD_Q + 2 * XD_WSTEP; // 8832 units = 141,312 bytes

#define XD_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define XD_ACC8(a) "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3])
#define XD_MFMA_DRAIN(a) asm volatile("s_nop 15\n\ts_nop 15" : XD_ACC8(a))
#define XD_VALU_SETTLE(a) asm volatile("s_nop 7\n\ts_nop 7" : XD_ACC8(a))

__device__ __forceinline__ u32x4 xd_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu,
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void xd_ld(f32x4& d, const u32x4& rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff));
}
template <int YOUNGER>
__device__ __forceinline__ void xd_ld_wait(f32x4 (&a)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(YOUNGER));
}
template <int YOUNGER>
__device__ __forceinline__ void xd_barrier() {       // this wave's LDS-DMA of the next k-step's weights has landed: all but the
  if constexpr (YOUNGER == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // YOUNGER loads behind it
  else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  static_assert(YOUNGER == 0 || YOUNGER == 4, "a quarter's staging issues four loads");
}

struct XDArgs {
  const float* x;           // (N, CI, 2 Hl, 2 Wl)
  const u32x4* wp;
  const float* bias;
  float* y;                 // (N, CO, Hl, Wl)
  int N, CI, CO, Hl, Wl;
  int tiles_x, tiles_y, tiles_co, ntiles;
  float bias_scale, slope;
  int act;
};
struct XDTile { int n, oy0, ox0, co_t; };

__global__ __launch_bounds__(512) void conv_x3_down_kernel(XDArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[XD_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int l16 = lane & 15, kg = lane >> 4, lane16 = lane * 16;
  const int Hi = 2 * p.Hl, Wi = 2 * p.Wl;
  const int plane = Hi * Wi;                    // input plane
  const int oplane = p.Hl * p.Wl;
  const int nq = p.CI / 8;                      // quarters per tile; 4 k-steps each
  const int G = gridDim.x;

  auto decode = [&](int t) {
    XDTile c;
    c.co_t = t % p.tiles_co; t /= p.tiles_co;
    c.ox0 = (t % p.tiles_x) * 16; t /= p.tiles_x;
    c.oy0 = (t % p.tiles_y) * 8;
    c.n = t / p.tiles_y;
    return c;
  };

  // ---- staging item of this thread (360 of them): patch row r = 0..17 (input row 2 oy0 - 1 + r), 4-column group cg = 0..9
  //      (input columns 2 ox0 - 4 + 4 cg + i), channel quad cq.  Element i goes to row class r & 1 (index r >> 1), column class
  //      (i & 1) ^ 1 ... spelled out below; patch column c = 4 cg - 3 + i must lie in 0 .. 33. ---------------------------------
  const bool a_item = tid < 360;
  int a_r, a_cg, a_cq;
  {
    const int e = a_item ? tid : 0;
    a_cq = e & 1;
    const int t = e >> 1;
    a_cg = t % 10;
    a_r = t / 10;
  }
  const int cstride = plane * 4;
  f32x4 arA[4], arB[4];
  auto a_load_to = [&](f32x4 (&ar)[4], const XDTile& c, int q) {       // channels 8 q + 4 cq + j
    const u32x4 rs = xd_rsrc(p.x + (long long)c.n * p.CI * plane, (unsigned)((long long)p.CI * plane * 4));
    int r = a_r;
    asm volatile("" : "+v"(r));
    const int iy = 2 * c.oy0 - 1 + r, ix = 2 * c.ox0 - 4 + 4 * a_cg;
    const bool ok = a_item && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
    const int off = ok ? ((a_cq * 4) * plane + iy * Wi + ix) * 4 : (int)0x80000000;
    const int soff = q * 8 * plane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) xd_ld(ar[j], rs, off + j * cstride, soff);
  };
  // element i (one input pixel, 4 channels) of the item -> its parity plane: patch column c = 4 cg - 3 + i; row class r & 1,
  // row index r >> 1; column class c & 1, column index c >> 1; unit = ((rowclass * 2 + colclass) * XD_PP + (r >> 1) * 17 + (c >> 1))
  auto a_store_from = [&](const f32x4 (&ar)[4], int slot, int i) {
    if (!a_item) return;
    int r = a_r;
    asm volatile("" : "+v"(r));
    const int c = 4 * a_cg - 3 + i;
    if (c < 0 || c > 33) return;
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = ar[j][i];
      h[j] = (__bf16)v;
      const float r1 = v - (float)h[j];
      m[j] = (__bf16)r1;
      l[j] = (__bf16)(r1 - (float)m[j]);
    }
    const int unit = slot * XD_Q + ((r & 1) * 2 + (c & 1)) * XD_PP + (r >> 1) * 17 + (c >> 1);
    unsigned char* dst = reinterpret_cast<unsigned char*>(lds + unit) + a_cq * 8;
    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
    *reinterpret_cast<u32x2*>(dst + XD_PL * 16) = __builtin_bit_cast(u32x2, m);
    *reinterpret_cast<u32x2*>(dst + 2 * XD_PL * 16) = __builtin_bit_cast(u32x2, l);
  };
  auto a_load = [&](int set, const XDTile& c, int q) { if (set == 0) a_load_to(arA, c, q); else a_load_to(arB, c, q); };
  auto a_store_px = [&](int set, int slot, int i) { if (set == 0) a_store_from(arA, slot, i); else a_store_from(arB, slot, i); };
  auto a_wait = [&](int set, auto younger) {
    constexpr int Y = decltype(younger)::value;
    if (set == 0) xd_ld_wait<Y>(arA); else xd_ld_wait<Y>(arB);
  };

  // ---- weights: LDS-DMA, one k-step image (24 pieces of 1 KB, 3 per wave) ---------------------------------------------------
  const int nsteps = nq * 4;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(p.wp), 0, (unsigned)((long long)p.tiles_co * nsteps * XD_WSTEP * 16), 0x00020000);
  auto w_dma = [&](const XDTile& c, int step, int buf) {
    const int soff = (c.co_t * nsteps + step) * (XD_WSTEP * 16) + wv * 3072;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(lds + XD_WOFF + buf * XD_WSTEP + wv * 192 + i * 64),
                                               16, lane16, soff + i * 1024, 0, 0);
  };

  f32x4 accS[2][4], accH[2][4], accT[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      accS[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; accT[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  XD_VALU_SETTLE(accS);
  XD_VALU_SETTLE(accH);

  // lane's patch unit: column class kg & 1, column offset kg >> 1; rows 2 wm + m
  const int laneA = (kg & 1) * XD_PP + (2 * wm) * 17 + l16 + (kg >> 1);
  const int laneB = XD_WOFF + kg * XD_NT + wn * 64 + l16;

  bf16x8 aF[2][2][3];       // [set][row m][plane]
  bf16x8 bF[2][3];          // [set][plane] of one channel block nn
  auto a_frags = [&](int off, int set) {      // off: slot * XD_Q + rowclass * 2 * XD_PP + rowoffset * 17
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) aF[set][m][pl] = __builtin_bit_cast(bf16x8, lds[laneA + off + pl * XD_PL + m * 17]);
  };
  auto b_frags = [&](int buf, int nn, int set) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bF[set][pl] = __builtin_bit_cast(bf16x8, lds[laneB + buf * XD_WSTEP + pl * 4 * XD_NT + nn * 16]);
  };

  int tile = gl_xcd_remap(blockIdx.x, G);
  if (tile >= p.ntiles) return;
  XDTile cur = decode(tile);
  int sa = 0, sb = 1, sc = 2;       // ring slots: the quarter being multiplied, the next one, the one after

  // ---- prologue: quarter 0 in LDS, quarter 1 in registers (set B), weight steps 0 (landed) and 1 (in flight) ---------------------
  w_dma(cur, 0, 0);
  a_load(0, cur, 0);
  a_wait(0, std::integral_constant<int, 0>{});
#pragma unroll
  for (int i = 0; i < 4; ++i) { a_store_px(0, sa, i); __builtin_amdgcn_sched_barrier(0); }
  xd_barrier<0>();
  w_dma(cur, 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  a_load(1, cur, 1);
  a_wait(1, std::integral_constant<int, 0>{});      // (once per workgroup: the loop's counted wait assumes the DMAs of a whole quarter behind them)
  a_frags(sa * XD_Q, 0);            // k-step 0: tap row a = 0: row class 0, offset 0
  b_frags(0, 0, 0);

  int q = 0, gs = 0;                // quarter within the tile; k-steps since the kernel started (weight-buffer parity)
  for (;;) {
    const int ntile = tile + G;
    const bool nvalid = ntile < p.ntiles;
    const XDTile nxt = decode(nvalid ? ntile : tile);
    for (; q < nq; q += 2) {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int qq = q + (s8 >> 2), a = s8 & 3;         // this k-step: quarter qq of the tile, tap row a
        const int buf = (gs + s8) & 1, cs = s8 & 1;        // weight buffer, A register set
        const int setn = (s8 >> 2) ^ 1;                    // staging register set of quarter qq + 1: (qq + 1) & 1 (q is even)
        const bool has_next = a < 3 || qq + 1 < nq || nvalid;
        const bool next_q = qq + 1 < nq || nvalid;         // a quarter follows this one
        // next k-step's patch offset: tap row a + 1 of this quarter, or row 0 of the next quarter
        const int offN = a < 3 ? sa * XD_Q + ((a + 1) & 1) * 2 * XD_PP + ((a + 1) >> 1) * 17 : sb * XD_Q;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          if (nn < 3) b_frags(buf, nn + 1, (nn + 1) & 1);
          if (nn == 3) {
            // ---- the step's barrier: the next k-step's weights are visible behind it; this step's buffer is free --------------
            if (a == 1 && (qq + 2 < nq || nvalid)) xd_barrier<4>(); else xd_barrier<0>();       // (quarter qq + 2's loads: issued in step a = 0)
            {
              const int st2 = qq * 4 + a + 2;          // weights two k-steps on
              if (st2 < nsteps) w_dma(cur, st2, buf);
              else if (nvalid) w_dma(nxt, st2 - nsteps, buf);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (a == 0) {                              // quarter qq + 2 requested (four k-steps before its first use)
              if (qq + 2 < nq) a_load(setn ^ 1, cur, qq + 2);
              else if (nvalid) a_load(setn ^ 1, nxt, qq + 2 - nq);
            }
            if (has_next) { a_frags(offN, cs ^ 1); b_frags(buf ^ 1, 0, 0); }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            XD_MFMA(accS[m][nn], aF[cs][m][2], bF[nn & 1][0]);
            XD_MFMA(accS[m][nn], aF[cs][m][0], bF[nn & 1][2]);
            XD_MFMA(accS[m][nn], aF[cs][m][1], bF[nn & 1][1]);
            XD_MFMA(accS[m][nn], aF[cs][m][1], bF[nn & 1][0]);
            XD_MFMA(accS[m][nn], aF[cs][m][0], bF[nn & 1][1]);
            XD_MFMA(accH[m][nn], aF[cs][m][0], bF[nn & 1][0]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (nn == 1 && next_q) {
            // quarter qq + 1 goes to LDS, one pixel of every item per k-step (behind its loads: the 3 DMAs of each k-step since
            // - the loads were issued behind the DMA of step a = 0 of the quarter before - and the next quarter's 4 loads)
            if (a == 0) a_wait(setn, std::integral_constant<int, 9>{});      // the DMAs of steps a = 1, 2, 3 of the quarter before
            a_store_px(setn, sb, a);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (a == 3) { const int t_ = sa; sa = sb; sb = sc; sc = t_; }
      }
      gs += 8;
      // 16 channels x 16 taps = 256 terms: close the hi*hi chain
      XD_MFMA_DRAIN(accH);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) { accT[m][nn] += accH[m][nn]; accH[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      XD_VALU_SETTLE(accH);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    {
      XD_MFMA_DRAIN(accS);
      typedef const __attribute__((address_space(4))) XDArgs* XDArgsK;
      unsigned long long kpi = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kpi));
      const XDArgsK kp = (XDArgsK)kpi;
      float* const y = kp->y;
      const float* const bias = kp->bias;
      const float bias_scale = kp->bias_scale, slope = kp->slope;
      const int act = kp->act;
      const long long ib = (long long)cur.n * p.CO * oplane;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        const int co = cur.co_t * XD_NT + wn * 64 + nn * 16 + l16;
        const float bv = bias != nullptr ? bias[co] * bias_scale : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          f32x4 v = accT[m][nn] + accS[m][nn];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float f = v[r] + bv;
            if (act == GANLAB_ACT_LRELU) f = gl_lrelu(f, slope);
            v[r] = f;
          }
          *reinterpret_cast<f32x4*>(y + ib + (long long)co * oplane + (cur.oy0 + 2 * wm + m) * p.Wl + cur.ox0 + 4 * kg) = v;
          accT[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
          accS[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      XD_VALU_SETTLE(accS);
    }
    if (!nvalid) break;
    tile = ntile;
    cur = nxt;
    q = 0;
  }
}

__global__ void x3_down_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int up, float scale) {
  const int CO = up ? Cin : Cout, CI = up ? Cout : Cin;        // GEMM roles: pooled forward (up = 0), up layer's input gradient
  const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (e >= (long long)CO * CI) return;
  const int col = (int)(e & 127);
  const long long t = e >> 7;
  const int ci = (int)(t % CI), ct = (int)(t / CI);
  const int co = ct * 128 + col;
  const float* w9 = up ? w + ((long long)ci * Cin + co) * 9 : w + ((long long)co * Cin + ci) * 9;
  gl_x3_down_pack_position(w9, up, scale, out, CI, ci, co);
}

bool xd_ok(const ganlab_conv_geom* g, int dgrad) {
  if (g == nullptr || g->ks != 3 || g->pad != 1 || g->N <= 0) return false;
  if (dgrad ? !(g->up == 1 && g->pool == 0) : !(g->pool == 1 && g->up == 0)) return false;
  const int CI = dgrad ? g->Cout : g->Cin, CO = dgrad ? g->Cin : g->Cout;
  // low resolution: the pooled layer's output; the up layer's INPUT
  if (!dgrad && ((g->Hin | g->Win) & 1)) return false;
  const int Hl = dgrad ? g->Hin : g->Hin / 2, Wl = dgrad ? g->Win : g->Win / 2;
  if ((long long)CI * Hl * Wl * 16 > 0x7fffffffLL) return false;
  return CI % 16 == 0 && CO % XD_NT == 0 && Hl % 8 == 0 && Wl % 16 == 0;
}

int xd_launch(const float* x, const void* wp, const float* bias, float* y, int N, int CI, int CO, int Hl, int Wl, float bias_scale,
              int act, float slope, hipStream_t st) {
  XDArgs a{};
  a.x = x; a.wp = reinterpret_cast<const u32x4*>(wp); a.bias = bias; a.y = y;
  a.N = N; a.CI = CI; a.CO = CO; a.Hl = Hl; a.Wl = Wl;
  a.tiles_x = Wl / 16; a.tiles_y = Hl / 8; a.tiles_co = CO / XD_NT;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  const long long ntiles = (long long)N * a.tiles_x * a.tiles_y * a.tiles_co;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL || (long long)a.tiles_co * (CI / 8 * 4) * XD_WSTEP * 16 > 0xffffffffLL) return GANLAB_EINVAL;
  a.ntiles = (int)ntiles;
  const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
  GL_LAUNCH(conv_x3_down_kernel, dim3(grid), dim3(512), 0, st, a);
  return GL_CHECK_LAUNCH();
}

}  // namespace

extern "C" {

/* the strided form takes: a pooled layer's forward (dgrad = 0) or an up layer's input gradient (dgrad = 1) */
int ganlab_conv_s2_down_x3_supported(const ganlab_conv_geom* g, int dgrad) { return xd_ok(g, dgrad) ? 1 : 0; }

/* `up`: 0 = a pooled layer's forward weights, 1 = an up layer's input-gradient weights; 48*Cout*Cin bf16 elements */
long long ganlab_conv_s2_down_x3_pack(const float* w, void* out, int Cout, int Cin, int up, float scale, void* stream) {
  if (Cout <= 0 || Cin <= 0 || (up != 0 && up != 1)) return GANLAB_EINVAL;
  const int CO = up ? Cin : Cout, CI = up ? Cout : Cin;
  if (CO % XD_NT != 0 || CI % 16 != 0) return GANLAB_EINVAL;
  const long long n = 48LL * Cout * Cin;
  if (out == nullptr) return n;
  if (w == nullptr) return GANLAB_EINVAL;
  const long long positions = (long long)Cout * Cin;
  GL_LAUNCH(x3_down_pack_kernel, dim3((unsigned)((positions + 255) / 256)), dim3(256), 0, gl_stream(stream), w,
            reinterpret_cast<__bf16*>(out), Cout, Cin, up, scale);
  const int st = GL_CHECK_LAUNCH();
  return st != GANLAB_OK ? st : n;
}

/* ganlab_conv_s2_fwd_f32 of a pooled layer (geom.pool = 1): y = act(AvgPool2(conv3x3(x)) + bias * bias_scale) */
int ganlab_conv_s2_down_fwd_x3(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g,
                               float bias_scale, int act, float slope, void* stream) {
  if (!xd_ok(g, 0) || x == nullptr || wp == nullptr || y == nullptr) return GANLAB_EINVAL;
  return xd_launch(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin / 2, g->Win / 2, bias_scale, act, slope, gl_stream(stream));
}

/* ganlab_conv_s2_dgrad_f32 of an up layer (geom.up = 1): gy is (N, Cout, 2 Hin, 2 Win), gx (N, Cin, Hin, Win) */
int ganlab_conv_s2_down_dgrad_x3(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream) {
  if (!xd_ok(g, 1) || gy == nullptr || wp == nullptr || gx == nullptr) return GANLAB_EINVAL;
  return xd_launch(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin, g->Win, 1.f, GANLAB_ACT_NONE, 0.f, gl_stream(stream));
}

}  // extern "C"
