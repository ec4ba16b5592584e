// Disparity warp fused with the reconstruction error:
//   utils/imwrap.py:37-72  imwrap_BCHW(im_src, disp)  (defaults: fliplr False, LeftTop [0,0], scale 1)
//   models/iresnet.py:169-170  reconerror = |deconv1L2L - imwrap_BCHW(deconv1R2R, -r_pr0)|
// Stock torch spends ~8 launches and 5 passes over the (B,32,H,W) maps on this (linspace grids,
// stack, `im_src + delt`, grid_sample, sub, abs); here one pass: every map is read once and the
// error written once (3 x 63 MB at 384x1280 -> HBM-bound).
//
// Arithmetic follows this container's torch exactly where it matters: torch.linspace's
// two-sided fp32 formula for the base grid, grid_sample(bilinear, zeros, align_corners=False)
// un-normalisation ((g + 1) * size - 1) / 2 and its tap order nw, ne, sw, se; the reference's
// `+ delt` is applied to in-bounds taps only (out-of-bounds taps of `im_src + delt` are zeros).
#include "common.hpp"

namespace {

__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  const float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

__global__ __launch_bounds__(256) void warp_abs_error_kernel(
    const float* __restrict__ L, const float* __restrict__ R, const float* __restrict__ disp,
    float* __restrict__ out, int C, int H, int W, int H0, int W0, float x1, float y1, float delt,
    int ncg) {
  // thread = (pixel, group of 8 channels): the r01 form walked all C channels per pixel, one
  // dependent iteration after the other (80 us for 189 MB); here the eight channels' taps are all
  // in flight together and the grid is ncg times as large
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z / ncg, c0 = (blockIdx.z % ncg) * 8;
  if (x >= W) return;
  const float d = disp[((long)b * H + y) * W + x];
  const float gx = linspace_at(-1.f, x1, W, x) - d * 2.0f / (float)(W0 - 1);
  const float gy = linspace_at(-1.f, y1, H, y);
  const float ix = ((gx + 1.f) * (float)W0 - 1.f) / 2.f;
  const float iy = ((gy + 1.f) * (float)H0 - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  // |ix| can be huge for wild disparities: clamp before the int conversion (both taps then
  // fall outside and contribute zero, as in torch)
  const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W0 + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H0 + 1.f);
  const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  const bool inx0 = x0 >= 0 && x0 < W0, inx1 = x0 + 1 >= 0 && x0 + 1 < W0;
  const bool iny0 = y0 >= 0 && y0 < H0, iny1 = y0 + 1 >= 0 && y0 + 1 < H0;
  const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
  const long plane0 = (long)H0 * W0, plane = (long)H * W;
  const float* r = R + ((long)b * C + c0) * plane0 + (long)y0 * W0 + x0;
  const long o = ((long)b * C + c0) * plane + (long)y * W + x;
  const int nc = min(8, C - c0);
  float t[8][4], lv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {                  // every load of the group issued before the first use
    const float* rc = r + (c < nc ? c : 0) * plane0;
    t[c][0] = (iny0 && inx0) ? rc[0] : 0.f;
    t[c][1] = (iny0 && inx1) ? rc[1] : 0.f;
    t[c][2] = (iny1 && inx0) ? rc[W0] : 0.f;
    t[c][3] = (iny1 && inx1) ? rc[W0 + 1] : 0.f;
    lv[c] = L ? L[o + (c < nc ? c : 0) * plane] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c >= nc) break;
    float v = 0.f;                               // the same sum order as the r01 form (bit-identical)
    if (iny0 && inx0) v += (t[c][0] + delt) * nw;
    if (iny0 && inx1) v += (t[c][1] + delt) * ne;
    if (iny1 && inx0) v += (t[c][2] + delt) * sw;
    if (iny1 && inx1) v += (t[c][3] + delt) * se;
    out[o + c * plane] = L ? fabsf(lv[c] - v) : v;
  }
}

}  // namespace

extern "C" int dsm_warp_abs_error(const void* L, const void* R, const void* disp, void* out, int B,
                                  int C, int H, int W, int H0, int W0, float delt,
                                  dsm_stream_t stream) {
  DSM_REQUIRE(R && disp && out, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && H <= 65535, DSM_ERR_ARG);
  const int ncg = (C + 7) / 8;
  DSM_REQUIRE((long)B * ncg <= 65535, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(H > 1 && W > 1 && H0 > 1 && W0 > 1, DSM_ERR_ARG);      // imwrap.py:48
  const float x1 = (float)(-1.0 + (W - 1) * 2.0 / (W0 - 1));
  const float y1 = (float)(-1.0 + (H - 1) * 2.0 / (H0 - 1));
  dsm_clear_stale_error();
  hipLaunchKernelGGL(warp_abs_error_kernel, dim3(dsm_cdiv(W, 256), H, B * ncg), dim3(256), 0,
                     (hipStream_t)stream, (const float*)L, (const float*)R, (const float*)disp,
                     (float*)out, C, H, W, H0, W0, x1, y1, delt, ncg);
  return dsm_launch_status();
}
