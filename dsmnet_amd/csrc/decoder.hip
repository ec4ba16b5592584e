// One decoder level of DispNetC / iResNet in one launch -- models/dispnetcorr.py:89-132 and
// models/iresnet.py:119-161,186-193:
//     myCat2d( relu(deconv(x) + bias),  upsample(pr),  skip )
// where the reference runs ConvTranspose2d (+ bias) -> ReLU -> nn.Upsample(scale_factor=2,
// bilinear) -> three crops to the common (h, w) (util_fun.py:7-15) -> torch.cat: ~7 launches per
// level and 5-6 levels per forward (~65 element-wise launches, 0.4 of DispNetC's 2.3 ms).
// Here the transposed convolution runs WITHOUT its bias (stock kernel) and this kernel does the
// rest: out (B, Cu + Cp + Cs, h, w) NCHW fp32, h = min(Hu, 2 Hp, Hs), w likewise.
//   channels [0, Cu):        relu?(up + bias[c])            (bit-identical to the stock ops)
//   channels [Cu, Cu + Cp):  bilinear x2 of pr, align_corners = False, with torch's
//                            area_pixel_compute_source_index arithmetic (scale 0.5)
//   channels [Cu + Cp, ...): skip
// HBM-bound streaming copy; each thread moves 4 consecutive x.
#include "common.hpp"

namespace {

struct DecCatParams {
  const float* up; const float* bias; const float* pr; const float* skip; float* out;
  int B, Cu, Cp, Cs;
  int Hu, Wu, Hp, Wp, Hs, Ws;
  int h, w, relu;
};

__device__ __forceinline__ float up2_sample(const float* __restrict__ plane, int Hp, int Wp, int y, int x) {
  const float sy = fmaxf(0.5f * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < Hp - 1 ? 1 : 0), x1 = x0 + (x0 < Wp - 1 ? 1 : 0);
  const float ly = sy - y0, lx = sx - x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  return hy * (hx * plane[y0 * Wp + x0] + lx * plane[y0 * Wp + x1]) +
         ly * (hx * plane[y1 * Wp + x0] + lx * plane[y1 * Wp + x1]);
}

__global__ __launch_bounds__(256) void decoder_cat_kernel(DecCatParams p) {
  const int wq = (p.w + 3) >> 2;
  const int C = p.Cu + p.Cp + p.Cs;
  const long n = (long)p.B * C * p.h * wq;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long t = i;
  const int xq = t % wq; t /= wq;
  const int y = t % p.h; t /= p.h;
  const int c = t % C; const int b = t / C;
  const int x0 = 4 * xq;
  float v[4];
  if (c < p.Cu) {
    const float* src = p.up + (((long)b * p.Cu + c) * p.Hu + y) * p.Wu;
    const float bs = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = (x0 + k < p.w) ? src[x0 + k] + bs : 0.f;
      v[k] = p.relu ? fmaxf(a, 0.f) : a;
    }
  } else if (c < p.Cu + p.Cp) {
    const float* plane = p.pr + ((long)b * p.Cp + (c - p.Cu)) * p.Hp * p.Wp;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (x0 + k < p.w) ? up2_sample(plane, p.Hp, p.Wp, y, x0 + k) : 0.f;
  } else {
    const float* src = p.skip + (((long)b * p.Cs + (c - p.Cu - p.Cp)) * p.Hs + y) * p.Ws;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (x0 + k < p.w) ? src[x0 + k] : 0.f;
  }
  float* dst = p.out + (((long)b * C + c) * p.h + y) * p.w + x0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (x0 + k < p.w) dst[k] = v[k];
}

// Image staging of the 2-D towers: the two views (B,C,H,W) NCHW, C <= 16, become ONE batch
// (2B,16,H,W) in NHWC memory with zero channels C..15 -- the 16-channel granularity the MFMA kernel
// stages.  One pass instead of torch.cat + two strided copies + a fill (4-5 launches, 66 us at
// 384 x 1280).  `right` may be NULL (one view: B images).  Thread = one pixel: C coalesced plane
// reads, four 16-byte stores.
__global__ __launch_bounds__(256) void stage_pair_kernel(const float* __restrict__ left,
                                                         const float* __restrict__ right,
                                                         float* __restrict__ out, int B, int C, long hw) {
  const long n = (long)(right ? 2 * B : B) * hw;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long b = i / hw, px = i % hw;
  const float* src = (b < B ? left + b * C * hw : right + (b - B) * C * hw) + px;
  float v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = c < C ? src[c * hw] : 0.f;
  f32x4* dst = reinterpret_cast<f32x4*>(out + i * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

}  // namespace

extern "C" int dsm_stage_images_nhwc16(const void* left, const void* right, void* out, int B, int C,
                                       int H, int W, dsm_stream_t stream) {
  DSM_REQUIRE(left && out && B > 0 && C > 0 && H > 0 && W > 0, DSM_ERR_ARG);
  DSM_REQUIRE(C <= 16, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(dsm_aligned16(out), DSM_ERR_ALIGN);
  const long n = (long)(right ? 2 : 1) * B * H * W;
  DSM_REQUIRE(n / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(stage_pair_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)left, (const float*)right, (float*)out, B, C, (long)H * W);
  return dsm_launch_status();
}

extern "C" int dsm_decoder_cat(const void* up, const void* bias, const void* pr, const void* skip,
                               void* out, int B, int Cu, int Cp, int Cs, int Hu, int Wu, int Hp,
                               int Wp, int Hs, int Ws, int relu, dsm_stream_t stream) {
  DSM_REQUIRE(up && out && B > 0 && Cu > 0 && Hu > 0 && Wu > 0, DSM_ERR_ARG);
  DSM_REQUIRE(Cp >= 0 && Cs >= 0 && (Cp == 0 || (pr && Hp > 0 && Wp > 0)) &&
              (Cs == 0 || (skip && Hs > 0 && Ws > 0)), DSM_ERR_ARG);
  DecCatParams p;
  p.up = (const float*)up; p.bias = (const float*)bias; p.pr = (const float*)pr;
  p.skip = (const float*)skip; p.out = (float*)out;
  p.B = B; p.Cu = Cu; p.Cp = Cp; p.Cs = Cs;
  p.Hu = Hu; p.Wu = Wu; p.Hp = Hp; p.Wp = Wp; p.Hs = Hs; p.Ws = Ws;
  p.h = Hu; p.w = Wu;
  if (Cp) { p.h = p.h < 2 * Hp ? p.h : 2 * Hp; p.w = p.w < 2 * Wp ? p.w : 2 * Wp; }
  if (Cs) { p.h = p.h < Hs ? p.h : Hs; p.w = p.w < Ws ? p.w : Ws; }
  p.relu = relu ? 1 : 0;
  const long n = (long)B * (Cu + Cp + Cs) * p.h * ((p.w + 3) / 4);
  DSM_REQUIRE(n / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(decoder_cat_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p);
  return dsm_launch_status();
}
