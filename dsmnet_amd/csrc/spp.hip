// PSMNet's spatial-pyramid-pooling head on NHWC maps:
//   models/psmnet/submodule.py:81-99 (branch1..4 = AvgPool2d(64/32/16/8) -> convbn(128,32,1,1,0,1)
//   -> ReLU) and :126-137 (bilinear upsample of the four branches to the 1/4-resolution grid,
//   concat [raw 64 | skip 128 | branch4 | branch3 | branch2 | branch1] -> 320 channels).
// `convbn` pads by its dilation whatever the kernel size (submodule.py:10-13), so the 1x1 branch
// convolution runs with padding 1: its output is (h+2) x (w+2) with a border of BN(0) = shift.
//
// Stock torch needs ~21 launches of tiny kernels for this (0.6 ms per PSMNet forward at
// 384x1280, launch-bound).  Here: three launches, all HBM/L2-bound and small:
//   spp_pool8_kernel     skip (B,H,W,128) -> P8 (B,H/8,W/8,128)             31 MB read
//   spp_branches_kernel  P8 -> four (B,h+2,w+2,32) maps: the 16/32/64 pools are means of
//                        2x2 / 4x4 / 8x8 P8 pixels (equal tiling windows, floor semantics of the
//                        reference's direct pools preserved), 1x1 conv + folded BN + ReLU
//   spp_concat_kernel    raw, skip, branch maps -> (B,H,W,320), bilinear taps on the fly   79 MB written
#include "common.hpp"

namespace {

constexpr int C_RAW = 64, C_SKIP = 128, C_BR = 32, C_OUT = C_RAW + C_SKIP + 4 * C_BR;

// branch bi (0..3) pools P8 by f = 1 << bi; map size (h8/f + 2) x (w8/f + 2)
struct BranchGeo { int h[4], w[4]; long off[4]; long total; };
__host__ __device__ inline BranchGeo branch_geo(int B, int h8, int w8) {
  BranchGeo g;
  long o = 0;
  for (int i = 0; i < 4; ++i) {
    g.h[i] = (h8 >> i) + 2; g.w[i] = (w8 >> i) + 2;
    g.off[i] = o;
    o += (long)B * g.h[i] * g.w[i] * C_BR;
  }
  g.total = o;
  return g;
}

// one workgroup per pooled pixel: 8 rows x 32 channel quads, then a row reduction in LDS
__global__ __launch_bounds__(256) void spp_pool8_kernel(const float* __restrict__ x,
                                                         float* __restrict__ y, int H, int W,
                                                         int h8, int w8) {
  __shared__ f32x4 part[8][32];
  const int q = threadIdx.x & 31, r = threadIdx.x >> 5;
  int id = blockIdx.x;
  const int px = id % w8; id /= w8;
  const int py = id % h8; const int b = id / h8;
  const f32x4* row = reinterpret_cast<const f32x4*>(
      x + (((long)b * H + py * 8 + r) * W + px * 8) * C_SKIP) + q;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; ++c) s += row[c * (C_SKIP / 4)];
  part[r][q] = s;
  __syncthreads();
  if (r == 0) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += part[k][q];
    reinterpret_cast<f32x4*>(y + (((long)b * h8 + py) * w8 + px) * C_SKIP)[q] = s * (1.f / 64.f);
  }
}

// one workgroup (128 threads) per branch-map pixel
__global__ __launch_bounds__(128) void spp_branches_kernel(const float* __restrict__ p8,
                                                           const float* __restrict__ wt,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           float* __restrict__ out, int B, int h8,
                                                           int w8) {
  __shared__ float v[C_SKIP];
  __shared__ float part[4][C_BR];
  // branch bi: map (h8 >> bi) + 2 by (w8 >> bi) + 2, maps back to back (scalar walk: indexing the
  // BranchGeo arrays with a runtime bi would put them in scratch)
  long id = blockIdx.x, off = 0;
  int bi = 0, hb = h8 + 2, wb = w8 + 2;
  while (bi < 3 && id >= (long)B * hb * wb) {
    id -= (long)B * hb * wb; off += (long)B * hb * wb * C_BR;
    ++bi; hb = (h8 >> bi) + 2; wb = (w8 >> bi) + 2;
  }
  const int f = 1 << bi;
  const int xx = (int)(id % wb); id /= wb;
  const int yy = (int)(id % hb); const int b = (int)(id / hb);
  const int t = threadIdx.x;
  float* o = out + off + (((long)b * hb + yy) * wb + xx) * C_BR;
  const bool border = yy == 0 || xx == 0 || yy == hb - 1 || xx == wb - 1;   // block-uniform
  if (border) {                                     // zero padding: conv = 0, BN(0) = shift
    if (t < C_BR) o[t] = fmaxf(shift[bi * C_BR + t], 0.f);
    return;
  }
  const float* src = p8 + (((long)b * h8 + (yy - 1) * f) * w8 + (xx - 1) * f) * C_SKIP + t;
  float s = 0.f;
  for (int dy = 0; dy < f; ++dy)
    for (int dx = 0; dx < f; ++dx) s += src[((long)dy * w8 + dx) * C_SKIP];
  v[t] = s * (1.f / (float)(f * f));
  __syncthreads();
  const int oc = t & 31, pt = t >> 5;
  const float* w = wt + (long)bi * C_SKIP * C_BR + (long)pt * 32 * C_BR + oc;   // [branch][cin][cout]
  float a = 0.f;
#pragma unroll 8
  for (int c = 0; c < 32; ++c) a = fmaf(v[pt * 32 + c], w[c * C_BR], a);
  part[pt][oc] = a;
  __syncthreads();
  if (t < C_BR) {
    const float z = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    o[t] = fmaxf(z * scale[bi * C_BR + t] + shift[bi * C_BR + t], 0.f);
  }
}

struct Tap { int i0, i1; float w0, w1; };
// torch's area_pixel_compute_source_index (align_corners = False) + guard_index_and_lambda
__device__ __forceinline__ Tap tap_at(int o, int in_size, int out_size) {
  Tap r;
  const float scale = (float)in_size / (float)out_size;
  const float src = fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
  const int i0 = min((int)floorf(src), in_size - 1);
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  r.w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  r.w0 = 1.f - r.w1;
  return r;
}

// A workgroup takes 16 pixels at a time: thread = (pixel tid / 16, quad tid % 16 + 16 pass), five
// passes over the pixel's 80 16-byte quads -- 256 contiguous bytes per pixel and pass, the pixel's
// coordinates and bilinear taps computed once, 32-bit index arithmetic (the first form, one flat
// 64-bit index per quad with a division by 80 and by W, H each, spent its time in integer division).
__device__ __forceinline__ float quad_amax(const f32x4 v) {
  return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__global__ __launch_bounds__(256) void spp_concat_kernel(const float* __restrict__ raw,
                                                         const float* __restrict__ skip,
                                                         const float* __restrict__ br,
                                                         float* __restrict__ out, int B, int H,
                                                         int W, int h8, int w8, float* __restrict__ y_amax) {
  constexpr int NQ = C_OUT / 4;
  float am = 0.f;               // max |out| of this thread's quads (the fp16 convolution modes' x_amax of lastconv)
  const BranchGeo g = branch_geo(B, h8, w8);
  const int npix = B * H * W;                                    // < 2^31: checked by the host
  const int lp = threadIdx.x >> 4, lq = threadIdx.x & 15;
  for (int pix = blockIdx.x * 16 + lp; pix < npix; pix += gridDim.x * 16) {
    const int x = pix % W, y = (pix / W) % H, b = pix / (W * H);
    f32x4* dst = reinterpret_cast<f32x4*>(out) + (long)pix * NQ;
    // passes 0 (raw: quads 0..15), 1..2 (skip: quads 16..47)
    const f32x4 c0 = reinterpret_cast<const f32x4*>(raw + (long)pix * C_RAW)[lq];
    const f32x4 c1 = reinterpret_cast<const f32x4*>(skip + (long)pix * C_SKIP)[lq];
    const f32x4 c2 = reinterpret_cast<const f32x4*>(skip + (long)pix * C_SKIP)[16 + lq];
    dst[lq] = c0; dst[16 + lq] = c1; dst[32 + lq] = c2;
    am = fmaxf(fmaxf(fmaxf(am, quad_amax(c0)), quad_amax(c1)), quad_amax(c2));
    // passes 3..4: quads 48..79 = four branches x 8 quads
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int bq = 16 * pass + lq, bi = bq >> 3, cq = bq & 7;
      const int hb = g.h[bi], wb = g.w[bi];
      const Tap ty = tap_at(y, hb, H), tx = tap_at(x, wb, W);
      const f32x4* m = reinterpret_cast<const f32x4*>(br + g.off[bi] + (long)b * hb * wb * C_BR) + cq;
      const f32x4 v00 = m[(ty.i0 * wb + tx.i0) * (C_BR / 4)];
      const f32x4 v01 = m[(ty.i0 * wb + tx.i1) * (C_BR / 4)];
      const f32x4 v10 = m[(ty.i1 * wb + tx.i0) * (C_BR / 4)];
      const f32x4 v11 = m[(ty.i1 * wb + tx.i1) * (C_BR / 4)];
      const f32x4 v = ty.w0 * (tx.w0 * v00 + tx.w1 * v01) + ty.w1 * (tx.w0 * v10 + tx.w1 * v11);
      dst[48 + bq] = v;
      am = fmaxf(am, quad_amax(v));
    }
  }
  if (y_amax) {                 // uniform: one atomic per workgroup, none when the slot already holds as much
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
      am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      if (am > __builtin_nontemporal_load(y_amax))
        atomicMax(reinterpret_cast<unsigned*>(y_amax), __builtin_bit_cast(unsigned, am));
    }
  }
}

}  // namespace

extern "C" size_t dsm_spp_branch_floats(int B, int h8, int w8) {
  if (B <= 0 || h8 < 8 || w8 < 8) return 0;
  return (size_t)branch_geo(B, h8, w8).total;
}

extern "C" int dsm_spp_pool8(const void* skip, void* p8, int B, int H, int W,
                             dsm_stream_t stream) {
  DSM_REQUIRE(skip && p8, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && H >= 64 && W >= 64, DSM_ERR_ARG);   // the reference's AvgPool2d(64) needs a window
  DSM_REQUIRE(dsm_aligned16(skip) && dsm_aligned16(p8), DSM_ERR_ALIGN);
  const int h8 = H / 8, w8 = W / 8;
  const long blocks = (long)B * h8 * w8;
  DSM_REQUIRE(blocks < (1L << 31), DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(spp_pool8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float*)skip, (float*)p8, H, W, h8, w8);
  return dsm_launch_status();
}

extern "C" int dsm_spp_branches(const void* p8, const void* w_t, const void* scale,
                                const void* shift, void* branches, int B, int h8, int w8,
                                dsm_stream_t stream) {
  DSM_REQUIRE(p8 && w_t && scale && shift && branches, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && h8 >= 8 && w8 >= 8, DSM_ERR_ARG);
  const long blocks = branch_geo(B, h8, w8).total / C_BR;
  DSM_REQUIRE(blocks < (1L << 31), DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(spp_branches_kernel, dim3((unsigned)blocks), dim3(128), 0,
                     (hipStream_t)stream, (const float*)p8, (const float*)w_t,
                     (const float*)scale, (const float*)shift, (float*)branches, B, h8, w8);
  return dsm_launch_status();
}

extern "C" int dsm_spp_concat(const void* raw, const void* skip, const void* branches, void* out,
                              int B, int H, int W, float* y_amax, dsm_stream_t stream) {
  DSM_REQUIRE(raw && skip && branches && out, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && H >= 64 && W >= 64, DSM_ERR_ARG);
  DSM_REQUIRE(dsm_aligned16(raw) && dsm_aligned16(skip) && dsm_aligned16(branches) &&
              dsm_aligned16(out), DSM_ERR_ALIGN);
  const long npix = (long)B * H * W;
  DSM_REQUIRE(npix < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  const long blocks = (npix + 15) / 16 < 256 * 32 ? (npix + 15) / 16 : 256 * 32;   // grid-stride beyond 32 per CU
  dsm_clear_stale_error();
  hipLaunchKernelGGL(spp_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float*)raw, (const float*)skip, (const float*)branches, (float*)out,
                     B, H, W, H / 8, W / 8, y_amax);
  return dsm_launch_status();
}
