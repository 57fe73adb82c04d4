// Real-image input path on the device: uint8 NHWC dataset images -> box-filter
// downsample to the current resolution -> fp32 NCHW normalised batch, in one HBM-bound pass.
//
// Replaces, per batch, the host chain  PIL Image.resize(BOX) -> ToTensor -> Normalize
// (gan_lab/data_config.py:307-341; the Resize is swapped on every growth event,
// gan_lab/progan/learner.py:1099-1112).  The uint8 stage is BIT-EXACT with PIL's two-pass resampler for
// power-of-two factors: PIL filters horizontally, rounds half-up to uint8, then filters vertically and
// rounds again (fixed point, 22 fractional bits; 2^22/f is exact for f = 2^k).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void u8_box_decode_kernel(const unsigned char* __restrict__ in,
                                                            float* __restrict__ out, int N, int Hs, int Ws, int C,
                                                            int f, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv,
                                                            const unsigned char* __restrict__ flip) {
  const int Ho = Hs / f, Wo = Ws / f;
  const long long total = (long long)N * Ho * Wo;
  const unsigned coef = (1u << 22) / (unsigned)f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    const long long t = i / Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int ws = (flip && flip[n]) ? (Wo - 1 - wo) : wo;   // RandomHorizontalFlip acts after the Resize
    const unsigned char* base = in + (((long long)n * Hs + (long long)ho * f) * Ws + (long long)ws * f) * C;
    for (int c = 0; c < C; ++c) {
      unsigned v = 0;
      for (int dy = 0; dy < f; ++dy) {
        const unsigned char* row = base + (long long)dy * Ws * C + c;
        unsigned h = 0;
        for (int dx = 0; dx < f; ++dx) h += row[dx * C];
        v += (unsigned)(((unsigned long long)h * coef + (1u << 21)) >> 22);      // horizontal pass, rounded to u8
      }
      v = (unsigned)(((unsigned long long)v * coef + (1u << 21)) >> 22);         // vertical pass, rounded to u8
      const float x = (float)v / 255.f;                                          // ToTensor
      out[(((long long)n * C + c) * Ho + ho) * Wo + wo] = (x - mean[c]) / stdv[c];   // Normalize
    }
  }
}

}  // namespace

extern "C" int ganlab_u8_box_decode_f32(const unsigned char* in_nhwc, float* out_nchw, int N, int Hs, int Ws, int C,
                                        int factor, const float* mean, const float* stdv,
                                        const unsigned char* flip, void* stream) {
  if (!in_nhwc || !out_nchw || !mean || !stdv || N <= 0 || Hs <= 0 || Ws <= 0 || C <= 0) return GANLAB_EINVAL;
  if (factor < 1 || (factor & (factor - 1)) != 0 || Hs % factor != 0 || Ws % factor != 0) return GANLAB_EUNSUPPORTED;
  const long long total = (long long)N * (Hs / factor) * (Ws / factor);
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  GL_LAUNCH(u8_box_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in_nhwc, out_nchw, N, Hs,
            Ws, C, factor, mean, stdv, flip);
  return GL_CHECK_LAUNCH();
}
