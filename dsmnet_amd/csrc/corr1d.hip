// 1-D left/right feature correlation (DispNetC / iResNet) for gfx950.
//
// Replaces Corr1d.forward, models/util_conv.py:71-86: a python loop that, per
// disparity, materialises a (B,C,H,W-d) product, reduces it over C and slice-assigns
// one plane (~3 launches and ~4*C*H*W*4 bytes per plane).  Here one launch reads
// each feature row once: a workgroup stages fL[b,:,y,x0:x0+64] and the shifted
// window fR[b,:,y,x0-Dpad*s:x0+64] in LDS per 32-channel chunk, and each thread
// owns a 4(x) x 8(d) register block, fed by 16-B LDS reads (1 for fL, 3 or 5 for
// the fR window) -- 32 FMAs per 4 or 6 ds_read_b128, which balances LDS and VALU.
//
// Algorithmic bytes: 2*C*H*W*4 + D*H*W*4; DispNetC 384x1280
// = 36.5 MB, i.e. Infinity-Cache resident and launch/latency scale (about 6 us at
// HBM rate) -- reported as such, not as an HBM fraction.
#include "common.hpp"

namespace {
constexpr int TX = 64;     // x per workgroup
constexpr int DB = 8;      // disparities per thread
constexpr int CC = 32;     // channels per LDS chunk

__device__ __forceinline__ f32x4 load4_guarded(const float* __restrict__ row, int x, int W,
                                               bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (vec && x >= 0 && x + 3 < W) return *reinterpret_cast<const f32x4*>(row + x);
  if (x + 0 >= 0 && x + 0 < W) v.x = row[x + 0];
  if (x + 1 >= 0 && x + 1 < W) v.y = row[x + 1];
  if (x + 2 >= 0 && x + 2 < W) v.z = row[x + 2];
  if (x + 3 >= 0 && x + 3 < W) v.w = row[x + 3];
  return v;
}
}  // namespace

// grid (ceil(W/64), H, B); block = 16 * ceil(D/8) threads rounded up to a wave.
template <int S>
__global__ __launch_bounds__(256) void corr1d_fwd_kernel(
    const float* __restrict__ fL, const float* __restrict__ fR, float* __restrict__ out,
    int C, int H, int W, int D, int vec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ndg = (D + DB - 1) / DB;
  const int padl = ndg * DB * S;            // left halo of the fR window (multiple of 8)
  const int RW = TX + padl;                 // fR columns; column j <-> x = x0 - padl + j
  float* Ls = lds;                          // [CC][TX]
  float* Rs = lds + CC * TX;                // [CC][RW]
  const int x0 = blockIdx.x * TX, y = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int xg = tid & 15, dg = tid >> 4;
  const bool worker = dg < ndg;
  const int d0 = dg * DB;
  constexpr int WIN = 4 + DB * S;           // floats of the fR window per thread
  float acc[DB][4];
#pragma unroll
  for (int i = 0; i < DB; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;

  for (int c0 = 0; c0 < C; c0 += CC) {
    const int cc = min(CC, C - c0);
    __syncthreads();
    for (int i = tid; i < cc * (TX / 4); i += nt) {
      const int c = i / (TX / 4), k = i % (TX / 4);
      const float* row = fL + (((long)b * C + c0 + c) * H + y) * W;
      *reinterpret_cast<f32x4*>(&Ls[c * TX + 4 * k]) = load4_guarded(row, x0 + 4 * k, W, vec);
    }
    const int r4 = RW / 4;
    for (int i = tid; i < cc * r4; i += nt) {
      const int c = i / r4, k = i % r4;
      const float* row = fR + (((long)b * C + c0 + c) * H + y) * W;
      *reinterpret_cast<f32x4*>(&Rs[c * RW + 4 * k]) =
          load4_guarded(row, x0 - padl + 4 * k, W, vec);
    }
    __syncthreads();
    if (worker) {
      const float* lp = Ls + 4 * xg;
      const float* rp = Rs + 4 * xg + padl - (d0 + DB) * S;   // 16-B aligned
      for (int c = 0; c < cc; ++c) {
        const f32x4 l = *reinterpret_cast<const f32x4*>(lp + c * TX);
        float win[WIN];
#pragma unroll
        for (int k = 0; k < WIN / 4; ++k) {
          const f32x4 r = *reinterpret_cast<const f32x4*>(rp + c * RW + 4 * k);
          win[4 * k] = r.x; win[4 * k + 1] = r.y; win[4 * k + 2] = r.z; win[4 * k + 3] = r.w;
        }
#pragma unroll
        for (int i = 0; i < DB; ++i) {
          // output x = x0+4xg+j pairs with fR column (x - (d0+i)S) = window[j + (DB-i)S]
          acc[i][0] = fmaf(l.x, win[0 + (DB - i) * S], acc[i][0]);
          acc[i][1] = fmaf(l.y, win[1 + (DB - i) * S], acc[i][1]);
          acc[i][2] = fmaf(l.z, win[2 + (DB - i) * S], acc[i][2]);
          acc[i][3] = fmaf(l.w, win[3 + (DB - i) * S], acc[i][3]);
        }
      }
    }
  }
  if (!worker) return;
  const int x = x0 + 4 * xg;
  if (x >= W) return;
#pragma unroll
  for (int i = 0; i < DB; ++i) {
    const int d = d0 + i;
    if (d >= D) break;
    float* o = out + (((long)b * D + d) * H + y) * W + x;
    if (vec && x + 3 < W) {
      f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
      *reinterpret_cast<f32x4*>(o) = v;
    } else {
      for (int j = 0; j < 4 && x + j < W; ++j) o[j] = acc[i][j];
    }
  }
}

// Any stride: one thread per output element (kept for strides other than 1, 2).
__global__ __launch_bounds__(256) void corr1d_fwd_generic_kernel(
    const float* __restrict__ fL, const float* __restrict__ fR, float* __restrict__ out,
    int C, int H, int W, int D, int S) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  const int b = blockIdx.z / D, d = blockIdx.z % D;
  if (x >= W) return;
  const int xs = x - d * S;
  float a = 0.f;
  if (xs >= 0) {
    const long plane = (long)H * W;
    const float* l = fL + (long)b * C * plane + (long)y * W + x;
    const float* r = fR + (long)b * C * plane + (long)y * W + xs;
    for (int c = 0; c < C; ++c) a = fmaf(l[c * plane], r[c * plane], a);
  }
  out[(((long)b * D + d) * H + y) * W + x] = a;
}

// AvgPool2d(k, stride 1, pad k/2) with the padding counted in the divisor
// (util_conv.py:82-85); also its own adjoint, so the backward reuses it.
__global__ __launch_bounds__(256) void box_filter_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int H, int W, int k) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= W) return;
  const float* p = src + (long)blockIdx.z * H * W;
  const int r = k / 2;
  float a = 0.f;
  for (int dy = -r; dy <= r; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -r; dx <= r; ++dx) {
      const int xx = x + dx;
      if (xx >= 0 && xx < W) a += p[(long)yy * W + xx];
    }
  }
  dst[(long)blockIdx.z * H * W + (long)y * W + x] = a / (float)(k * k);
}

// Backward:
//   dfL[c,y,x ] = sum_i g[i,y,x]      * fR[c,y,x-i*s]
//   dfR[c,y,x'] = sum_i g[i,y,x'+i*s] * fL[c,y,x'+i*s]
__global__ __launch_bounds__(256) void corr1d_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ fL, const float* __restrict__ fR,
    float* __restrict__ dfL, float* __restrict__ dfR, int C, int H, int W, int D, int S) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  const int bc = blockIdx.z;
  if (x >= W) return;
  const int b = bc / C;
  const long plane = (long)H * W;
  const float* gp = g + (long)b * D * plane + (long)y * W;
  const float* lrow = fL + (long)bc * plane + (long)y * W;
  const float* rrow = fR + (long)bc * plane + (long)y * W;
  const int dlim = min(D, W);
  float aL = 0.f, aR = 0.f;
  for (int i = 0; i < dlim; ++i) {
    const int sh = i * S;
    if (x - sh >= 0) aL = fmaf(gp[i * plane + x], rrow[x - sh], aL);
    if (x + sh < W) aR = fmaf(gp[i * plane + x + sh], lrow[x + sh], aR);
  }
  dfL[(long)bc * plane + (long)y * W + x] = aL;
  dfR[(long)bc * plane + (long)y * W + x] = aR;
}

static int check_corr(const void* a, const void* b, const void* c, int B, int C, int H, int W,
                      int D, int stride, int ksize, int dtype) {
  DSM_REQUIRE(a && b && c, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0 && stride > 0, DSM_ERR_ARG);
  DSM_REQUIRE(ksize >= 1 && (ksize & 1) == 1, DSM_ERR_ARG);       // util_conv.py:83
  DSM_REQUIRE(dtype == DSM_F32, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(H <= 65535 && (long)B * D <= 65535 && (long)B * C <= 65535, DSM_ERR_UNSUPPORTED);
  return DSM_OK;
}

extern "C" int dsm_corr1d_fwd(const void* fL, const void* fR, void* out, void* tmp, int B, int C,
                              int H, int W, int D, int stride, int ksize, int dtype,
                              dsm_stream_t stream) {
  int rc = check_corr(fL, fR, out, B, C, H, W, D, stride, ksize, dtype);
  if (rc != DSM_OK) return rc;
  DSM_REQUIRE(ksize == 1 || tmp, DSM_ERR_ARG);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  float* raw = (float*)(ksize > 1 ? tmp : out);
  const int ndg = (D + DB - 1) / DB;
  const int vec = (W % 4 == 0) && dsm_aligned16(fL) && dsm_aligned16(fR) && dsm_aligned16(raw);
  if ((stride == 1 || stride == 2) && ndg * 16 <= 256) {
    const int threads = ((ndg * 16 + 63) / 64) * 64;
    const size_t lds = (size_t)CC * (TX + TX + ndg * DB * stride) * sizeof(float);
    dim3 grid(dsm_cdiv(W, TX), H, B);
    if (stride == 1)
      hipLaunchKernelGGL(corr1d_fwd_kernel<1>, grid, dim3(threads), lds, s, (const float*)fL,
                         (const float*)fR, raw, C, H, W, D, vec);
    else
      hipLaunchKernelGGL(corr1d_fwd_kernel<2>, grid, dim3(threads), lds, s, (const float*)fL,
                         (const float*)fR, raw, C, H, W, D, vec);
  } else {
    dim3 grid(dsm_cdiv(W, 256), H, B * D);
    hipLaunchKernelGGL(corr1d_fwd_generic_kernel, grid, dim3(256), 0, s, (const float*)fL,
                       (const float*)fR, raw, C, H, W, D, stride);
  }
  if (ksize > 1) {
    dim3 grid(dsm_cdiv(W, 256), H, B * D);
    hipLaunchKernelGGL(box_filter_kernel, grid, dim3(256), 0, s, (const float*)raw, (float*)out,
                       H, W, ksize);
  }
  return dsm_launch_status();
}

extern "C" int dsm_corr1d_bwd(const void* grad_out, const void* fL, const void* fR, void* dfL,
                              void* dfR, void* tmp, int B, int C, int H, int W, int D, int stride,
                              int ksize, int dtype, dsm_stream_t stream) {
  int rc = check_corr(grad_out, fL, fR, B, C, H, W, D, stride, ksize, dtype);
  if (rc != DSM_OK) return rc;
  DSM_REQUIRE(dfL && dfR, DSM_ERR_ARG);
  DSM_REQUIRE(ksize == 1 || tmp, DSM_ERR_ARG);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  const float* g = (const float*)grad_out;
  if (ksize > 1) {
    dim3 grid(dsm_cdiv(W, 256), H, B * D);
    hipLaunchKernelGGL(box_filter_kernel, grid, dim3(256), 0, s, g, (float*)tmp, H, W, ksize);
    g = (const float*)tmp;
  }
  dim3 grid(dsm_cdiv(W, 256), H, B * C);
  hipLaunchKernelGGL(corr1d_bwd_kernel, grid, dim3(256), 0, s, g, (const float*)fL,
                     (const float*)fR, (float*)dfL, (float*)dfR, C, H, W, D, stride);
  return dsm_launch_status();
}
