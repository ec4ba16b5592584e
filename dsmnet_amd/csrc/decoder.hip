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

// PSMNet's first tower layer on the raw images: Conv2d(3 -> 32, k3, stride 2, pad 1) + folded BN +
// ReLU (models/psmnet/submodule.py:70-72), both views in one launch, NCHW images in, NHWC 32-channel
// map out.  27 x 32 MACs per output pixel on the VALU (v_pk_fma_f32: the tap's 32 weights as SGPR
// pairs, the input value broadcast), instead of staging the images to 16 NHWC channels (63 MB
// written and read back at 384 x 1280) for the MFMA kernel: one launch replaces two.
// wt: [27 taps (c, ky, kx)][32 couts].  Thread = one output pixel.
__global__ __launch_bounds__(256) void conv_first3_kernel(const float* __restrict__ left,
                                                          const float* __restrict__ right,
                                                          const float* __restrict__ wt,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          float* __restrict__ out, int B, int H, int W,
                                                          int Ho, int Wo, int relu) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef const float __attribute__((address_space(4))) cfloat;
  typedef f32x4 __attribute__((address_space(4))) cquad;
  const long n = (long)(right ? 2 * B : B) * Ho * Wo;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), b = (int)(i / ((long)Wo * Ho));
  const float* img = b < B ? left + (long)b * 3 * H * W : right + (long)(b - B) * 3 * H * W;
  cfloat* wc = (cfloat*)wt;
  asm volatile("" : "+s"(wc));                         // keep the weight loads inside the tap loop
  f2 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = f2{0.f, 0.f};
  float in[27];                                        // all 27 inputs first: one round of load latency
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const int c = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    const int yi = 2 * yo - 1 + ky, xi = 2 * xo - 1 + kx;
    in[tap] = (yi >= 0 && yi < H && xi >= 0 && xi < W) ? img[((long)c * H + yi) * W + xi] : 0.f;
  }
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const float a = in[tap];
    f32x4 w4[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w4[q] = *(const volatile cquad*)(wc + tap * 32 + q * 4);
    const f2 a2 = {a, a};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc[2 * q] = __builtin_elementwise_fma(a2, f2{w4[q].x, w4[q].y}, acc[2 * q]);
      acc[2 * q + 1] = __builtin_elementwise_fma(a2, f2{w4[q].z, w4[q].w}, acc[2 * q + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);                 // one tap's 32 SGPRs live at a time
  }
  f32x4* dst = reinterpret_cast<f32x4*>(out + i * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f32x4 v = {acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y};
    if (scale) v = v * *reinterpret_cast<const f32x4*>(scale + 4 * q);
    if (shift) v = v + *reinterpret_cast<const f32x4*>(shift + 4 * q);
    if (relu) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
    dst[q] = v;
  }
}

}  // namespace

extern "C" int dsm_conv2d_first3_fwd(const void* left, const void* right, const void* w_taps,
                                     const float* scale, const float* shift, void* out, int B, int H,
                                     int W, int relu, dsm_stream_t stream) {
  DSM_REQUIRE(left && w_taps && out && B > 0 && H > 0 && W > 0, DSM_ERR_ARG);
  DSM_REQUIRE(dsm_aligned16(out) && dsm_aligned16(w_taps) && dsm_aligned16(scale) && dsm_aligned16(shift),
              DSM_ERR_ALIGN);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long n = (long)(right ? 2 : 1) * B * Ho * Wo;
  DSM_REQUIRE(n / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(conv_first3_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)left, (const float*)right, (const float*)w_taps, scale, shift,
                     (float*)out, B, H, W, Ho, Wo, relu ? 1 : 0);
  return dsm_launch_status();
}

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
