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

#include <type_traits>

// Ablation builds (make VARIANT=.. DEFS=-DX3_EXP=n; wrong results, timing only): bit 0 no activation split / LDS stores,
// bit 1 no epilogue stores, bit 2 no activation loads
#ifndef X3_EXP
#define X3_EXP 0
#endif

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

// ---- weight packing: OIHW fp32 -> [co tile][k-step][plane][k-group][co 64][8] bf16 (common.h gl_x3_pack_position) ------
// one thread per GEMM position (ci, co); adjacent threads = adjacent output columns
__global__ void x3_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int mode, float scale) {
  const bool dg = mode == GANLAB_PACK_DGRAD;
  const int CO = dg ? Cin : Cout, CI = dg ? Cout : Cin;        // GEMM roles
  const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (e >= (long long)CO * CI) return;
  const int col = (int)(e & 63);
  const long long t = e >> 6;
  const int ci = (int)(t % CI), ct = (int)(t / CI);
  const int co = ct * 64 + col;
  const float* w9 = dg ? w + ((long long)ci * Cin + co) * 9 : w + ((long long)co * Cin + ci) * 9;
  gl_x3_pack_position(w9, dg, scale, out, CI, ci, co);
}

struct X3Args {
  const float* x;
  const u32x4* wp;
  const float* bias;
  float* y;
  int N, CI, CO, H, W;
  int tiles_x, tiles_y, tiles_co, ntiles;
  float bias_scale, slope;
  int act;
  const float* mask;        // MASK: y *= (mask > 0 ? 1 : mslope), mask shaped like y (the input gradient of a conv whose input
  float mslope;             //       is a LeakyReLU output comes out multiplied by that LeakyReLU's derivative)
  const float* aff_s;       // AFF: the conv reads x * aff_s[n][ci] + aff_t[n][ci] inside the image (deferred InstanceNorm)
  const float* aff_t;
  const float* noise;       // TAIL: y = act(conv + noise_w[c] * noise[n][hw] + bias) and this tile's sums of y, y^2 per (n, c):
  const float* noise_w;     //       spart[((n * CO + c) * 4 tiles_per_plane + 4 tile + wave row) * 2]
  double* spart;
};

// Activation loads are inline asm: hipcc neither sees them in its vmcnt bookkeeping (beside an LDS-DMA it waits vmcnt(0) in
// front of the first use of any load it does see: the whole prefetch pipeline drained several times per stage) nor may recycle
// their registers before x3_ld_wait, which is tied to them and counts the younger operations by hand.
__device__ __forceinline__ u32x4 x3_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu,
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void x3_ld(f32x4& d, const u32x4& rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff));
}
template <int YOUNGER>
__device__ __forceinline__ void x3_ld_wait(f32x4 (&a)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(YOUNGER));
}
template <int YOUNGER>
__device__ __forceinline__ void x3_ld_wait(f32x4 (&a)[4], f32x4& s_, f32x4& t_) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(s_), "+v"(t_) : "n"(YOUNGER));
}

// kernel forms
constexpr int X3_PLAIN = 0, X3_MASK = 1, X3_AFF = 2, X3_AFF_TAIL = 3;

// The MFMAs are inline asm: accumulators pinned in the accumulation registers (tied operand, "a" class), issued in exactly
// this order.  hipcc's hazard recogniser does not see through asm: an accumulator read by the vector ALU needs the matrix
// pipe drained first (X3_MFMA_DRAIN before the chain dump and the epilogue).
#define X3_MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// drain / settle: wait states tied to a whole accumulator set (the set counts as rewritten by them, so neither the vector
// ALU code in front of the block nor the MFMAs behind it can be scheduled across)
#define X3_ACC8(a) "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1])
#define X3_MFMA_DRAIN(a) asm volatile("s_nop 15\n\ts_nop 15" : X3_ACC8(a))
#define X3_VALU_SETTLE(a) asm volatile("s_nop 7\n\ts_nop 7" : X3_ACC8(a))

// tap offset (units) inside the halo patch
__host__ __device__ constexpr int x3_toff(int tap) { return (tap / 3) * 18 + tap % 3; }

struct X3Tile { int n, oy0, ox0, co_t; };

// One barrier per stage, in its second k-step: everything a wave must have finished before it (its LDS-DMA of the next stage's
// weights: all but the `YOUNGER` vector-memory operations issued after them; its ds_writes and ds_reads)
template <int YOUNGER>
__device__ __forceinline__ void x3_barrier() {
  if constexpr (YOUNGER == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if constexpr (YOUNGER == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  static_assert(YOUNGER == 0 || YOUNGER == 4 || YOUNGER == 6, "the activation staging issues four loads (six with the affine)");
}

// Persistent workgroups: workgroup b takes the tiles remap(b) + i * gridDim.x; the k-loop runs on across tiles (the next tile's
// first two half patches and first weight stages are staged during the last stages of this one).
template <int FORM>
__global__ __launch_bounds__(512) void conv_x3_fwd_kernel(X3Args p) {
  constexpr bool MASK = FORM == X3_MASK, AFF = FORM == X3_AFF || FORM == X3_AFF_TAIL, TAIL = FORM == X3_AFF_TAIL;
  constexpr int NLOADS = AFF ? 6 : 4;           // vector-memory loads one activation staging issues
  __shared__ __attribute__((aligned(16))) u32x4 lds[X3_LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;          // pixel rows 4wm .. 4wm+3 of the tile, channels 32wn .. 32wn+31
  const int l16 = lane & 15, kg = lane >> 4, lane16 = lane * 16;
  const int plane = p.H * p.W;
  const int nstages = p.CI / 64 * 9, ndc = p.CI / 64;
  const int G = gridDim.x;

  auto decode = [&](int t) {
    X3Tile c;
    c.co_t = t % p.tiles_co; t /= p.tiles_co;
    c.ox0 = (t % p.tiles_x) * 16; t /= p.tiles_x;
    c.oy0 = (t % p.tiles_y) * 16;
    c.n = t / p.tiles_y;
    return c;
  };

  // ---- activation staging item of this thread: (channel group g, row r, 4-column group cg, channel quad cq) ----------
  // columns ox0 - 4 + 4cg .. + 3; halo column of element i = 4cg - 3 + i (valid 0 .. 17: cg 0 keeps i = 3, cg 5 keeps i = 0)
  // One register holds the item: bits 0-14 byte offset of element 0 / plane 0 in a half slot + 48 (>= 0), 16-17 first kept
  // element, 18-20 one past the last, 21-25 patch row, 26-28 column group, 29-30 channel quad index (g * 2 + cq).
  const bool a_item = tid < 432;
  int a_pack;
  {
    const int e = a_item ? tid : 0;
    const int cq = e & 1;
    int t = e >> 1;
    const int cg = t % 6; t /= 6;
    const int r = t % 18, g = t / 18;
    const int unit48 = ((g * 3) * X3_PL + r * 18 + 4 * cg) * 16 + cq * 8;     // + 48: the element-0 unit of cg = 0 is 3 units before
    a_pack = unit48 | ((cg == 0 ? 3 : 0) << 16) | ((cg == 5 ? 1 : 4) << 18) | (r << 21) | (cg << 26) | ((g * 2 + cq) << 29);
  }
  const int cstride = plane * 4;
  f32x4 ar[4];
  f32x4 a_sv, a_tv;                                  // AFF: scale / shift of the item's four channels (zeros outside the image)
  auto a_load = [&](const X3Tile& c, int half) {     // channels 16 half + 8g + 4cq + j of tile c's halo patch
    const u32x4 rs = x3_rsrc(p.x + (long long)c.n * p.CI * plane, (unsigned)((long long)p.CI * plane * 4));
    int pk = a_pack;
    asm volatile("" : "+v"(pk));      // offsets computed HERE, not hoisted out of the k-loop and spilled
    const int ry = ((pk >> 21) & 31) - 1, rx = ((pk >> 26) & 7) * 4 - 4, a_ch = ((pk >> 29) & 3) * 4;
    const bool ok = a_item && (unsigned)(c.oy0 + ry) < (unsigned)p.H && (unsigned)(c.ox0 + rx) < (unsigned)p.W;
    const int off = ok ? (a_ch * plane + (c.oy0 + ry) * p.W + c.ox0 + rx) * 4 : (int)0x80000000;
    const int soff = half * 16 * plane * 4;
    if (X3_EXP & 4) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) x3_ld(ar[j], rs, off + j * cstride, soff);
    if constexpr (AFF) {       // an item outside the image reads zeros for s and t as well: 0 * 0 + 0 keeps the padding zero
      const unsigned tab = (unsigned)((long long)p.N * p.CI * 4);
      const u32x4 rss = x3_rsrc(p.aff_s, tab), rst = x3_rsrc(p.aff_t, tab);
      const int o = ok ? a_ch * 4 : (int)0x80000000;
      const int so = (c.n * p.CI + half * 16) * 4;
      x3_ld(a_sv, rss, o, so);
      x3_ld(a_tv, rst, o, so);
    }
  };
  // the loads' data, once all but the `YOUNGER` vector-memory operations issued after them have completed
  auto a_wait = [&](auto younger) {
    if constexpr (AFF) x3_ld_wait<decltype(younger)::value>(ar, a_sv, a_tv); else x3_ld_wait<decltype(younger)::value>(ar);
  };
  // one pixel (of the item's four) per call: the split is vector-ALU work that belongs BETWEEN the rows' MFMA blocks, not in
  // front of the stage's barrier where every wave would wait for it (ablation: 19 % of the kernel there)
  auto a_store_px = [&](int slot, int i) {
    if (!a_item || (X3_EXP & 1)) return;
    int pk = a_pack;
    asm volatile("" : "+v"(pk));
    const int a_i0 = (pk >> 16) & 3, a_i1 = (pk >> 18) & 7;
    unsigned char* dst = reinterpret_cast<unsigned char*>(lds + X3_AOFF + slot * X3_HALF) + ((pk & 0x7fff) - 48);
    if (i < a_i0 || i >= a_i1) return;
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = i == 0 ? ar[j].x : i == 1 ? ar[j].y : i == 2 ? ar[j].z : ar[j].w;
      if constexpr (AFF) v = fmaf(v, j == 0 ? a_sv.x : j == 1 ? a_sv.y : j == 2 ? a_sv.z : a_sv.w,
                                  j == 0 ? a_tv.x : j == 1 ? a_tv.y : j == 2 ? a_tv.z : a_tv.w);
      h[j] = (__bf16)v;
      const float r1 = v - x3_up(h[j]);
      m[j] = (__bf16)r1;
      l[j] = (__bf16)(r1 - x3_up(m[j]));
    }
    *reinterpret_cast<u32x2*>(dst + i * 16) = __builtin_bit_cast(u32x2, h);
    *reinterpret_cast<u32x2*>(dst + i * 16 + X3_PL * 16) = __builtin_bit_cast(u32x2, m);
    *reinterpret_cast<u32x2*>(dst + i * 16 + 2 * X3_PL * 16) = __builtin_bit_cast(u32x2, l);
  };
  auto a_store = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_store_px(slot, i);
      __builtin_amdgcn_sched_barrier(0);     // one pixel at a time: the split's temporaries stay a dozen registers
    }
  };

  // ---- weight staging: LDS-DMA, the packed stage image is the LDS image; 24 pieces of 1 KB per stage, 3 per wave ----
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(p.wp), 0, (unsigned)((long long)p.tiles_co * nstages * X3_WSTAGE * 16), 0x00020000);
  auto w_dma = [&](const X3Tile& c, int st, int buf) {
    const int soff = (c.co_t * nstages + st) * (X3_WSTAGE * 16) + wv * 3072;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(lds + X3_WOFF + buf * X3_WSTAGE + wv * 192 + i * 64),
                                               16, lane16, soff + i * 1024, 0, 0);
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
  const int khi_i = khi ? 1 : 0;

  bf16x8 aF[2][3], bF[2][2][3];
  auto a_frags = [&](int off, int m) {       // row m of a k-step -> register set m & 1
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) aF[m & 1][pl] = __builtin_bit_cast(bf16x8, lds[off + pl * X3_PL + m * 18]);
  };
  auto b_frags = [&](int buf, int step1, int nn, int set) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      bF[set][nn][pl] = __builtin_bit_cast(bf16x8, lds[laneB + buf * X3_WSTAGE + step1 * X3_WSTEP + pl * 4 * X3_NT + nn * 16]);
  };

  int tile = gl_xcd_remap(blockIdx.x, G);
  if (tile >= p.ntiles) return;          // (grid <= ntiles: never taken)
  X3Tile cur = decode(tile);

  // ---- prologue: halves 0, 1, weight stages 0, 1 of the first tile -----------------------------------------------------
  w_dma(cur, 0, 0);
  a_load(cur, 0); a_wait(std::integral_constant<int, 0>{}); a_store(0);
  a_load(cur, 1); a_wait(std::integral_constant<int, 0>{}); a_store(1);
  x3_barrier<0>();
  if (nstages > 1) w_dma(cur, 1, 1);
  int offA;                          // this lane's patch offset (units) of the k-step whose rows are being read
  {
    offA = laneA + (khi ? x3_toff(x3_tap_hi(0)) : x3_toff(x3_tap_lo(0)));
    a_frags(offA, 0);
    b_frags(0, 0, 0, 0);
    b_frags(0, 0, 1, 0);
  }

  int st = 0;                       // stage within the tile
  int gst = 0;                      // stages since the kernel started
  int r0 = 0;                       // ring slot of this double chunk's first half
  for (;;) {
    const int ntile = tile + G;
    const bool nvalid = ntile < p.ntiles;
    const X3Tile nxt = decode(nvalid ? ntile : tile);
    for (int dc = 0; dc < ndc; ++dc) {
      // ring slots of the halves 4dc + 0 .. 5
      const int sl0 = r0, sl1 = r0 == 2 ? 0 : r0 + 1, sl2 = sl1 == 2 ? 0 : sl1 + 1;
      const bool lastdc = dc + 1 == ndc;
      const bool more = !lastdc || nvalid;       // there is a double chunk after this one
#pragma unroll
      for (int s18 = 0; s18 < 18; ++s18) {
        const int j = s18 >> 1, par = s18 & 1;
        const int buf = (gst + j) & 1;           // weight buffer of this stage (the parity runs on across tiles: 9 stages per 64 channels)
        const int cb = s18 & 1;                  // B register set of this k-step
        // next k-step: patch offset of this lane
        const int s1 = s18 == 17 ? 0 : s18 + 1;
        const int c1 = s1 / 9, s9 = s1 % 9;
        const int hl = 2 * c1 + x3_half_lo(s9), hh = 2 * c1 + x3_half_hi(s9);
        const int b0 = s18 == 17 ? 1 : 0;        // the next double chunk's ring starts one slot on
        const int kl = (hl + b0) % 3, kh = (hh + b0) % 3;
        const int slot_l = kl == 0 ? sl0 : kl == 1 ? sl1 : sl2, slot_h = kh == 0 ? sl0 : kh == 1 ? sl1 : sl2;
        int la = laneA;
        asm volatile("" : "+v"(la));           // keeps the 18 per-step offsets from being hoisted out of the double chunk
        const int u_lo = slot_l * X3_HALF + x3_toff(x3_tap_lo(s9)), u_hi = slot_h * X3_HALF + x3_toff(x3_tap_hi(s9));
        const int offN = la + u_lo + khi_i * (u_hi - u_lo);
        const bool has_next = s18 < 17 || more;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          // ---- operand prefetch -----------------------------------------------------------------------------------
          if (m < 3) a_frags(offA, m + 1);
          else if (has_next) a_frags(offN, 0);
          if (par == 0) {                       // next k-step is in the same weight buffer
            if (m == 0) b_frags(buf, 1, 0, cb ^ 1);
            if (m == 1) b_frags(buf, 1, 1, cb ^ 1);
          } else {                              // next k-step opens the next stage: behind the barrier
            if (m == 1 && has_next) b_frags(buf ^ 1, 0, 0, cb ^ 1);
            if (m == 2 && has_next) b_frags(buf ^ 1, 0, 1, cb ^ 1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nn = 0; nn < 2; ++nn) {
            X3_MFMA(accS[m][nn], aF[m & 1][2], bF[cb][nn][0]);
            X3_MFMA(accS[m][nn], aF[m & 1][0], bF[cb][nn][2]);
            X3_MFMA(accS[m][nn], aF[m & 1][1], bF[cb][nn][1]);
            X3_MFMA(accS[m][nn], aF[m & 1][1], bF[cb][nn][0]);
            X3_MFMA(accS[m][nn], aF[m & 1][0], bF[cb][nn][1]);
            X3_MFMA(accH[m][nn], aF[m & 1][0], bF[cb][nn][0]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (par == 0) {
            // a half patch goes to LDS in the first k-step of its stage, one pixel of every item behind each row's MFMAs
            // (visible behind the stage's barrier in the second k-step)
            const bool first_half1 = j == 0 && (dc > 0 || st > 0 || tile != gl_xcd_remap(blockIdx.x, G));
            // behind the loads of this half: the DMA of the stage in between (3 operations; a tile's epilogue stores on top
            // of them are simply waited for as well)
            if (m == 0 && (first_half1 || j == 2 || j == 4 || (j == 7 && more))) a_wait(std::integral_constant<int, 3>{});
            if (first_half1) a_store_px(sl1, m);                 // half 4dc + 1
            if (j == 2) a_store_px(sl2, m);                      // half 4dc + 2
            if (j == 4) a_store_px(sl0, m);                      // half 4dc + 3
            if (j == 7 && more) a_store_px(sl1, m);              // half 4dc + 4 (the next tile's half 0 behind the last double chunk)
            __builtin_amdgcn_sched_barrier(0);
          }
          if (par == 1 && m == 0) {
            // ---- the stage's barrier: the next stage's weights (and a half patch stored in the first k-step) are visible
            //      behind it ---------------------------------------------------------------------------------------------
            if (j == 1 || j == 3) x3_barrier<NLOADS>();
            else if (j == 6 || j == 8) { if (more) x3_barrier<NLOADS>(); else x3_barrier<0>(); }
            else x3_barrier<0>();
            // weights two stages on -> the buffer this stage has finished reading
            {
              const int s2 = st + j + 2;
              if (s2 < nstages) w_dma(cur, s2, buf);
              else if (nvalid) w_dma(nxt, s2 - nstages, buf);
            }
            __builtin_amdgcn_sched_barrier(0);    // the counted wait of the next barrier assumes: DMA first, then the 4 loads
            if (j == 0) a_load(cur, 4 * dc + 2);
            if (j == 2) a_load(cur, 4 * dc + 3);
            if (j == 5 && more) { if (lastdc) a_load(nxt, 0); else a_load(cur, 4 * dc + 4); }
            if (j == 7 && more) { if (lastdc) a_load(nxt, 1); else a_load(cur, 4 * dc + 5); }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        offA = offN;
        if (s18 == 8 || s18 == 17) {     // a 32-channel chunk is done: close its hi*hi chain
          X3_MFMA_DRAIN(accH);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) { accT[m][q] += accH[m][q]; accH[m][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          X3_VALU_SETTLE(accH);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      st += 9;
      gst += 9;
      r0 = sl1;      // (r0 + 4) % 3
    }

    // ---- epilogue: T + S + bias, activation; lane = 4 consecutive pixels of one channel --------------------------------
    {
      X3_MFMA_DRAIN(accS);
      // the epilogue's arguments are read from the kernel-argument segment HERE: held in scalar registers across the k-loop
      // they cost a dozen spills
      // (a copy in registers, read once per tile by scalar loads: through the laundered pointer every use would be a flat load
      // re-issued behind each store)
      typedef const __attribute__((address_space(4))) X3Args* X3ArgsK;
      unsigned long long kpi = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kpi));
      const X3ArgsK kp = (X3ArgsK)kpi;
      struct { float* y; const float* bias; const float* mask; const float* noise; const float* noise_w; double* spart;
               float bias_scale, slope, mslope; int act; } q;
      q.y = kp->y; q.bias = kp->bias; q.mask = kp->mask; q.noise = kp->noise; q.noise_w = kp->noise_w; q.spart = kp->spart;
      q.bias_scale = kp->bias_scale; q.slope = kp->slope; q.mslope = kp->mslope; q.act = kp->act;
      const long long ib = (long long)cur.n * p.CO * plane;
      float bv[2], nwv[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        const int co = cur.co_t * X3_NT + wn * 32 + nn * 16 + l16;
        bv[nn] = q.bias != nullptr ? q.bias[co] * q.bias_scale : 0.f;
        nwv[nn] = 0.f;
        if constexpr (TAIL) nwv[nn] = q.noise != nullptr ? q.noise_w[co] : 0.f;
      }
      // every load of the epilogue in flight before the first use (one round trip, not one per tile row)
      float4 nzs[TAIL ? 4 : 1];
      f32x4 mks[MASK ? 4 : 1][MASK ? 2 : 1];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int px = (cur.oy0 + 4 * wm + m) * p.W + cur.ox0 + 4 * kg;
        if constexpr (TAIL) nzs[m] = q.noise != nullptr ? *reinterpret_cast<const float4*>(q.noise + (long long)cur.n * plane + px) : float4{0.f, 0.f, 0.f, 0.f};
        if constexpr (MASK) {
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
            mks[m][nn] = *reinterpret_cast<const f32x4*>(q.mask + ib + (long long)(cur.co_t * X3_NT + wn * 32 + nn * 16 + l16) * plane + px);
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int px = (cur.oy0 + 4 * wm + m) * p.W + cur.ox0 + 4 * kg;
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
          const int co = cur.co_t * X3_NT + wn * 32 + nn * 16 + l16;
          const long long o = ib + (long long)co * plane + px;
          f32x4 v = accT[m][nn] + accS[m][nn];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float f;
            if constexpr (TAIL) f = fmaf(r == 0 ? nzs[m].x : r == 1 ? nzs[m].y : r == 2 ? nzs[m].z : nzs[m].w, nwv[nn], v[r] + bv[nn]);
            else f = v[r] + bv[nn];
            if (q.act == GANLAB_ACT_LRELU) f = gl_lrelu(f, q.slope);
            if constexpr (MASK) { if (!(mks[m][nn][r] > 0.f)) f *= q.mslope; }
            v[r] = f;
          }
          if (!(X3_EXP & 2)) *reinterpret_cast<f32x4*>(q.y + o) = v;
          if constexpr (TAIL) {
            s1[nn] += (v[0] + v[1]) + (v[2] + v[3]);
            s2[nn] += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
          }
          accT[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
          accS[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if constexpr (TAIL) {      // this wave's 64 pixels of each channel: fp64 from the lane sums on, one partial per wave row
        const long long chunks = (long long)p.tiles_x * p.tiles_y * 4, tl = ((long long)(cur.oy0 >> 4) * p.tiles_x + (cur.ox0 >> 4)) * 4 + wm;
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
          double a = (double)s1[nn], b = (double)s2[nn];
          a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
          b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
          const int co = cur.co_t * X3_NT + wn * 32 + nn * 16 + l16;
          if (kg == 0) {
            double* dst = q.spart + (((long long)cur.n * p.CO + co) * chunks + tl) * 2;
            dst[0] = a; dst[1] = b;
          }
        }
      }
      X3_VALU_SETTLE(accS);
    }
    if (!nvalid) break;
    tile = ntile;
    cur = nxt;
    st = 0;
  }
}

bool x3_ok(const ganlab_conv_geom* g, int dgrad) {
  if (g == nullptr || g->ks != 3 || g->pad != 1 || g->up || g->pool) return false;
  const int CI = dgrad ? g->Cout : g->Cin, CO = dgrad ? g->Cin : g->Cout;
  return CI % 64 == 0 && CO % X3_NT == 0 && g->Hin % 16 == 0 && g->Win % 16 == 0 && g->N > 0;
}

int x3_launch(int form, X3Args a, hipStream_t st) {
  a.tiles_x = a.W / 16; a.tiles_y = a.H / 16; a.tiles_co = a.CO / X3_NT;
  const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y * a.tiles_co;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL || (long long)a.tiles_co * (a.CI / 64 * 9) * X3_WSTAGE * 16 > 0xffffffffLL ||
      (long long)a.CI * a.H * a.W * 4 > 0x7fffffffLL || (long long)a.N * a.CI * 4 > 0x7fffffffLL)
    return GANLAB_EINVAL;
  a.ntiles = (int)ntiles;
  const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);       // one workgroup per CU (146 KB of LDS), persistent
  switch (form) {
    case X3_PLAIN: GL_LAUNCH(conv_x3_fwd_kernel<X3_PLAIN>, dim3(grid), dim3(512), 0, st, a); break;
    case X3_MASK: GL_LAUNCH(conv_x3_fwd_kernel<X3_MASK>, dim3(grid), dim3(512), 0, st, a); break;
    case X3_AFF: GL_LAUNCH(conv_x3_fwd_kernel<X3_AFF>, dim3(grid), dim3(512), 0, st, a); break;
    default: GL_LAUNCH(conv_x3_fwd_kernel<X3_AFF_TAIL>, dim3(grid), dim3(512), 0, st, a); break;
  }
  return GL_CHECK_LAUNCH();
}

X3Args x3_args(const float* x, const void* wp, const float* bias, float* y, int N, int CI, int CO, int H, int W, float bias_scale,
               int act, float slope) {
  X3Args a{};
  a.x = x; a.wp = reinterpret_cast<const u32x4*>(wp); a.bias = bias; a.y = y;
  a.N = N; a.CI = CI; a.CO = CO; a.H = H; a.W = W;
  a.bias_scale = bias_scale; a.slope = slope; a.act = act;
  return a;
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
  const long long positions = (long long)Cout * Cin;
  GL_LAUNCH(x3_pack_kernel, dim3((unsigned)((positions + 255) / 256)), dim3(256), 0, gl_stream(stream), w,
            reinterpret_cast<__bf16*>(out), Cout, Cin, mode, scale);
  const int st = GL_CHECK_LAUNCH();
  return st != GANLAB_OK ? st : n;
}

int ganlab_conv_fwd_x3(const float* x, const void* wp, const float* bias, float* y, const ganlab_conv_geom* g, float bias_scale,
                       int act, float slope, void* stream) {
  if (!x3_ok(g, 0) || x == nullptr || wp == nullptr || y == nullptr) return GANLAB_EINVAL;
  return x3_launch(X3_PLAIN, x3_args(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin, g->Win, bias_scale, act, slope), gl_stream(stream));
}

int ganlab_conv_dgrad_x3(const float* gy, const void* wp, float* gx, const ganlab_conv_geom* g, void* stream) {
  if (!x3_ok(g, 1) || gy == nullptr || wp == nullptr || gx == nullptr) return GANLAB_EINVAL;
  return x3_launch(X3_PLAIN, x3_args(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin, g->Win, 1.f, GANLAB_ACT_NONE, 0.f),
                   gl_stream(stream));
}

/* ganlab_conv_dgrad_mask_f32: gx = dgrad(gy, w) * lrelu'(x), x shaped like gx */
int ganlab_conv_dgrad_mask_x3(const float* gy, const void* wp, const float* x, float* gx, const ganlab_conv_geom* g, float slope,
                              void* stream) {
  if (!x3_ok(g, 1) || gy == nullptr || wp == nullptr || gx == nullptr || x == nullptr) return GANLAB_EINVAL;
  X3Args a = x3_args(gy, wp, nullptr, gx, g->N, g->Cout, g->Cin, g->Hin, g->Win, 1.f, GANLAB_ACT_NONE, 0.f);
  a.mask = x; a.mslope = slope;
  return x3_launch(X3_MASK, a, gl_stream(stream));
}

/* ganlab_conv_fwd_aff_f32: the forward on b = x * aff_s[n][ci] + aff_t[n][ci] (zero padding of b stays zero) */
int ganlab_conv_fwd_aff_x3(const float* x, const void* wp, const float* aff_s, const float* aff_t, const float* bias, float* y,
                           const ganlab_conv_geom* g, float bias_scale, int act, float slope, void* stream) {
  if (!x3_ok(g, 0) || x == nullptr || wp == nullptr || y == nullptr || aff_s == nullptr || aff_t == nullptr) return GANLAB_EINVAL;
  X3Args a = x3_args(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin, g->Win, bias_scale, act, slope);
  a.aff_s = aff_s; a.aff_t = aff_t;
  return x3_launch(X3_AFF, a, gl_stream(stream));
}

/* ganlab_conv_fwd_aff_tail_f32: y = act(conv(x * aff_s + aff_t, w) + noise_w * noise + bias * bias_scale), mean / rstd = the
 * InstanceNorm statistics of y.  ganlab_conv_fwd_aff_tail_x3_chunks: partial sums per plane (workspace: N * Cout * tiles * 2 doubles) */
int ganlab_conv_fwd_aff_tail_x3_chunks(const ganlab_conv_geom* g) { return x3_ok(g, 0) ? (g->Hin / 16) * (g->Win / 16) * 4 : 0; }
int ganlab_conv_fwd_aff_tail_x3(const float* x, const void* wp, const float* aff_s, const float* aff_t, const float* bias,
                                const float* noise, const float* noise_w, float* y, float* mean, float* rstd,
                                const ganlab_conv_geom* g, float bias_scale, int act, float slope, float eps, void* workspace,
                                size_t workspace_bytes, void* stream) {
  const int chunks = ganlab_conv_fwd_aff_tail_x3_chunks(g);
  if (chunks <= 0) return GANLAB_EUNSUPPORTED;
  if (!x || !wp || !y || !aff_s || !aff_t || !mean || !rstd || (noise && !noise_w)) return GANLAB_EINVAL;
  const long long planes = (long long)g->N * g->Cout;
  if (!workspace || workspace_bytes < (size_t)planes * chunks * 2 * sizeof(double)) return GANLAB_EWORKSPACE;
  X3Args a = x3_args(x, wp, bias, y, g->N, g->Cin, g->Cout, g->Hin, g->Win, bias_scale, act, slope);
  a.aff_s = aff_s; a.aff_t = aff_t; a.noise = noise; a.noise_w = noise_w; a.spart = reinterpret_cast<double*>(workspace);
  const int rc = x3_launch(X3_AFF_TAIL, a, gl_stream(stream));
  if (rc != GANLAB_OK) return rc;
  return gl_tail_stats_finish(a.spart, mean, rstd, planes, chunks, 1.0 / ((double)g->Hin * g->Win), eps, gl_stream(stream));
}

}  // extern "C"
