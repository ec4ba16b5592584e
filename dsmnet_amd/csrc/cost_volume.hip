// Concatenation cost volume for GCNet / PSMNet on gfx950 -- streaming, HBM-write bound.
//
// Replaces the per-disparity slice-copy loops of models/gcnet.py:130-135 and
// models/psmnet/stackhourglass.py:124-133 (zeros() + 2*D strided copies) with ONE
// pass that writes every output byte exactly once (no memset) with 16 B per lane.
//
// Algorithmic bytes per pair: 2*C*H*W*4 read + 2*C*D*H*W*4
// written; PSMNet 384x1280: 385.4 MB.  The two feature maps (7.9 MB) stay L2 /
// Infinity-Cache resident, so HBM traffic is the volume write.
#include "common.hpp"

// ----------------------------------------------------------------------------
// NDHWC (torch.channels_last_3d) build: vol[b][d][y][x][2C].
// One workgroup owns (b, y, x-tile).  The NCHW rows fL[b,:,y,x0:x0+TX] and
// fR[b,:,y,x0-(D-1):x0+TX] are transposed once into LDS ([x][c], c contiguous);
// each thread then keeps one (x, 4-channel) position and walks d, so every
// disparity plane receives TX*2C*4 contiguous bytes from the workgroup.
// ----------------------------------------------------------------------------
template <int TX>
__global__ __launch_bounds__(256) void volume_ndhwc_fwd_kernel(
    const float* __restrict__ fL, const float* __restrict__ fR, float* __restrict__ vol,
    int C, int H, int W, int D, int mask_left) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int pitch = C + 4;                 // floats; keeps 16-B alignment, staggers banks
  const int nR = TX + D - 1;
  float* Ls = lds;                         // [TX][pitch]
  float* Rs = lds + TX * pitch;            // [nR][pitch], column j <-> x = x0-(D-1)+j
  const int x0 = blockIdx.x * TX, y = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  // bit 1 of mask_left: the RIGHT-referenced volume of gcnet_LR (models/gcnet.py:155-164): the
  // caller passes (fR, fL) as (first, second); the second map is then read at x + d (zero for
  // x + d >= W) instead of x - d, i.e. the staged window starts at x0 and the walk goes forward
  const bool rightref = (mask_left & 2) != 0;
  mask_left &= 1;
  const long rowL = ((long)b * C * H + y) * W;      // + c*H*W + x
  const long cstride = (long)H * W;
  // stage + transpose (global reads coalesced along x)
  for (int c = tid / TX; c < C; c += 256 / TX) {
    const int t = tid % TX, x = x0 + t;
    Ls[t * pitch + c] = (x < W) ? fL[rowL + c * cstride + x] : 0.f;
  }
  for (int c = tid >> 6; c < C; c += 4) {
    for (int j = tid & 63; j < nR; j += 64) {
      const int x = rightref ? x0 + j : x0 - (D - 1) + j;
      Rs[j * pitch + c] = (x >= 0 && x < W) ? fR[rowL + c * cstride + x] : 0.f;
    }
  }
  __syncthreads();
  const int q = C / 4;                     // float4 per half voxel
  const int per_plane = TX * 2 * q;        // float4 per disparity plane of this tile
  const long plane = (long)H * W * 2 * C;  // floats between consecutive d
  for (int e = tid; e < per_plane; e += 256) {
    const int t = e / (2 * q), c4 = e % (2 * q);
    const int x = x0 + t;
    if (x >= W) continue;
    float* dst = vol + ((((long)b * D) * H + y) * W + x) * 2 * C + c4 * 4;
    if (c4 < q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Ls[t * pitch + c4 * 4]);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const int dfull = mask_left ? (rightref ? min(D, W - x) : min(D, x + 1)) : D;     // planes with x >= d (x + d < W)
      for (int d = 0; d < dfull; ++d) *reinterpret_cast<f32x4*>(dst + d * plane) = v;
      for (int d = dfull; d < D; ++d) *reinterpret_cast<f32x4*>(dst + d * plane) = z;
    } else {
      const float* src = &Rs[(rightref ? t : t + D - 1) * pitch + (c4 - q) * 4];
      const int step = rightref ? pitch : -pitch;
#pragma unroll 4
      for (int d = 0; d < D; ++d)          // columns outside [0, W) were staged as zeros
        *reinterpret_cast<f32x4*>(dst + d * plane) =
            *reinterpret_cast<const f32x4*>(src + d * step);
    }
  }
}

// ----------------------------------------------------------------------------
// NCDHW (torch contiguous) build: vol[b][2C][d][y][x].  Every (b, c2, d) plane is
// a contiguous H*W block: a masked copy (left half) or an x-shifted copy (right).
// ----------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void volume_ncdhw_fwd_kernel(
    const float* __restrict__ fL, const float* __restrict__ fR, float* __restrict__ vol,
    int C, int H, int W, int D, int mask_left) {
  const int p = blockIdx.x;                // ((b*2C + c2)*D + d)
  const int d = p % D;
  const int c2 = (p / D) % (2 * C);
  const int b = p / (D * 2 * C);
  const bool right = c2 >= C;
  const float* src = (right ? fR : fL) + ((long)b * C + (right ? c2 - C : c2)) * H * W;
  float* dst = vol + (long)p * H * W;
  const int shift = right ? d : 0;
  const int lo = (right || mask_left) ? d : 0;       // first x that carries data
  if (VEC) {
    const int W4 = W >> 2, n4 = H * W4;
    for (int i = blockIdx.y * 1024 + threadIdx.x, k = 0; k < 4 && i < n4; ++k, i += 256) {
      const int yy = i / W4, x = (i - yy * W4) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (x >= lo) {
        v = *reinterpret_cast<const f32x4u*>(src + (long)yy * W + x - shift);
      } else if (x + 3 >= lo) {
        const float* s = src + (long)yy * W - shift;
        v.x = (x + 0 >= lo) ? s[x + 0] : 0.f;
        v.y = (x + 1 >= lo) ? s[x + 1] : 0.f;
        v.z = (x + 2 >= lo) ? s[x + 2] : 0.f;
        v.w = (x + 3 >= lo) ? s[x + 3] : 0.f;
      }
      *reinterpret_cast<f32x4*>(dst + (long)yy * W + x) = v;
    }
  } else {
    const int n = H * W;
    for (int i = blockIdx.y * 1024 + threadIdx.x, k = 0; k < 4 && i < n; ++k, i += 256) {
      const int yy = i / W, x = i - yy * W;
      dst[i] = (x >= lo) ? src[(long)yy * W + x - shift] : 0.f;
    }
  }
}

// ----------------------------------------------------------------------------
// Backward: dfL = sum_d g[:C] (masked as the forward), dfR[x'] = sum_d g[C:, d, x'+d].
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void volume_ncdhw_bwd_kernel(
    const float* __restrict__ g, float* __restrict__ dfL, float* __restrict__ dfR,
    int C, int H, int W, int D, int mask_left) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  const int bc = blockIdx.z;               // b*C + c
  if (x >= W) return;
  const int b = bc / C, c = bc % C;
  const long plane = (long)H * W;
  const float* gl = g + (((long)b * 2 * C + c) * D) * plane + (long)y * W;
  const float* gr = g + (((long)b * 2 * C + C + c) * D) * plane + (long)y * W;
  float aL = 0.f, aR = 0.f;
  const int dl = mask_left ? min(D, x + 1) : D;
  for (int d = 0; d < dl; ++d) aL += gl[d * plane + x];
  const int dr = min(D, W - x);
  for (int d = 0; d < dr; ++d) aR += gr[d * plane + x + d];
  const long o = ((long)bc * H + y) * W + x;
  dfL[o] = aL;
  dfR[o] = aR;
}

template <int TX>
__global__ __launch_bounds__(256) void volume_ndhwc_bwd_kernel(
    const float* __restrict__ g, float* __restrict__ dfL, float* __restrict__ dfR,
    int C, int H, int W, int D, int mask_left) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2C][TX+1]
  const int x0 = blockIdx.x * TX, y = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int q = C / 4;
  const long plane = (long)H * W * 2 * C;
  for (int e = tid; e < TX * 2 * q; e += 256) {
    const int t = e / (2 * q), c4 = e % (2 * q);
    const int x = x0 + t;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (x < W) {
      const float* src = g + ((((long)b * D) * H + y) * W + x) * 2 * C + c4 * 4;
      if (c4 < q) {
        const int dl = mask_left ? min(D, x + 1) : D;
        for (int d = 0; d < dl; ++d) acc += *reinterpret_cast<const f32x4*>(src + d * plane);
      } else {
        const int dr = min(D, W - x);
        for (int d = 0; d < dr; ++d)
          acc += *reinterpret_cast<const f32x4*>(src + d * plane + (long)d * 2 * C);
      }
    }
    float* l = &lds[(c4 * 4) * (TX + 1) + t];
    l[0] = acc.x; l[TX + 1] = acc.y; l[2 * (TX + 1)] = acc.z; l[3 * (TX + 1)] = acc.w;
  }
  __syncthreads();
  for (int c2 = tid / TX; c2 < 2 * C; c2 += 256 / TX) {
    const int t = tid % TX, x = x0 + t;
    if (x >= W) continue;
    float* out = (c2 < C) ? dfL : dfR;
    const int c = (c2 < C) ? c2 : c2 - C;
    out[(((long)b * C + c) * H + y) * W + x] = lds[c2 * (TX + 1) + t];
  }
}

// ----------------------------------------------------------------------------
// NCDHW <-> NDHWC repack: (B, C, S) <-> (B, S, C) tiled transpose through LDS.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relayout_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int C, long S, int to_ndhwc) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const long s0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* sb = src + (long)b * C * S;
  float* db = dst + (long)b * C * S;
  if (to_ndhwc) {
    for (int r = ty; r < 64; r += 4)       // r: channel, tx: spatial (contiguous in src)
      if (c0 + r < C && s0 + tx < S) tile[r][tx] = sb[(long)(c0 + r) * S + s0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4)       // r: spatial, tx: channel (contiguous in dst)
      if (c0 + tx < C && s0 + r < S) db[(s0 + r) * C + c0 + tx] = tile[tx][r];
  } else {
    for (int r = ty; r < 64; r += 4)
      if (c0 + tx < C && s0 + r < S) tile[r][tx] = sb[(s0 + r) * C + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
      if (c0 + r < C && s0 + tx < S) db[(long)(c0 + r) * S + s0 + tx] = tile[tx][r];
  }
}

// ----------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------
static int check_volume_args(const void* a, const void* b, const void* c, int B, int C, int H,
                             int W, int D, int layout, int dtype) {
  DSM_REQUIRE(a && b && c, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0, DSM_ERR_ARG);
  DSM_REQUIRE(layout == DSM_NCDHW || layout == DSM_NDHWC, DSM_ERR_ARG);
  DSM_REQUIRE(dtype == DSM_F32, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(H <= 65535 && B <= 65535, DSM_ERR_UNSUPPORTED);
  return DSM_OK;
}

extern "C" int dsm_concat_volume_fwd(const void* fL, const void* fR, void* vol, int B, int C,
                                     int H, int W, int D, int mask_left, int layout, int dtype,
                                     dsm_stream_t stream) {
  int rc = check_volume_args(fL, fR, vol, B, C, H, W, D, layout, dtype);
  if (rc != DSM_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  const float* l = (const float*)fL;
  const float* r = (const float*)fR;
  float* v = (float*)vol;
  DSM_REQUIRE((mask_left & ~3) == 0, DSM_ERR_ARG);
  if (mask_left & 2) DSM_REQUIRE(layout == DSM_NDHWC, DSM_ERR_UNSUPPORTED);   // right-referenced: NDHWC forward only
  if (layout == DSM_NDHWC) {
    DSM_REQUIRE(C % 4 == 0, DSM_ERR_UNSUPPORTED);
    DSM_REQUIRE(dsm_aligned16(vol), DSM_ERR_ALIGN);
    constexpr int TX = 32;
    const size_t lds = (size_t)(TX + TX + D - 1) * (C + 4) * sizeof(float);
    DSM_REQUIRE(lds <= 160 * 1024, DSM_ERR_UNSUPPORTED);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)volume_ndhwc_fwd_kernel<TX>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return DSM_ERR_LAUNCH;
    dim3 grid(dsm_cdiv(W, TX), H, B);
    hipLaunchKernelGGL(volume_ndhwc_fwd_kernel<TX>, grid, dim3(256), lds, s, l, r, v, C, H, W, D,
                       mask_left);
  } else {
    const bool vec = (W % 4 == 0) && dsm_aligned16(vol);
    const long planes = (long)B * 2 * C * D;
    DSM_REQUIRE(planes < (1L << 31), DSM_ERR_UNSUPPORTED);
    const long per = vec ? (long)H * W / 4 : (long)H * W;
    dim3 grid((unsigned)planes, dsm_cdiv(per, 1024), 1);
    if (vec)
      hipLaunchKernelGGL(volume_ncdhw_fwd_kernel<true>, grid, dim3(256), 0, s, l, r, v, C, H, W,
                         D, mask_left);
    else
      hipLaunchKernelGGL(volume_ncdhw_fwd_kernel<false>, grid, dim3(256), 0, s, l, r, v, C, H, W,
                         D, mask_left);
  }
  return dsm_launch_status();
}

extern "C" int dsm_concat_volume_bwd(const void* gvol, void* dfL, void* dfR, int B, int C, int H,
                                     int W, int D, int mask_left, int layout, int dtype,
                                     dsm_stream_t stream) {
  int rc = check_volume_args(gvol, dfL, dfR, B, C, H, W, D, layout, dtype);
  if (rc != DSM_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  if (layout == DSM_NDHWC) {
    DSM_REQUIRE(C % 4 == 0, DSM_ERR_UNSUPPORTED);
    DSM_REQUIRE(dsm_aligned16(gvol), DSM_ERR_ALIGN);
    constexpr int TX = 32;
    const size_t lds = (size_t)2 * C * (TX + 1) * sizeof(float);
    DSM_REQUIRE(lds <= 64 * 1024, DSM_ERR_UNSUPPORTED);
    dim3 grid(dsm_cdiv(W, TX), H, B);
    hipLaunchKernelGGL(volume_ndhwc_bwd_kernel<TX>, grid, dim3(256), lds, s, (const float*)gvol,
                       (float*)dfL, (float*)dfR, C, H, W, D, mask_left);
  } else {
    DSM_REQUIRE((long)B * C <= 65535, DSM_ERR_UNSUPPORTED);
    dim3 grid(dsm_cdiv(W, 256), H, B * C);
    hipLaunchKernelGGL(volume_ncdhw_bwd_kernel, grid, dim3(256), 0, s, (const float*)gvol,
                       (float*)dfL, (float*)dfR, C, H, W, D, mask_left);
  }
  return dsm_launch_status();
}

extern "C" int dsm_volume_relayout(const void* src, void* dst, int B, int C, int D, int H, int W,
                                   int to_ndhwc, dsm_stream_t stream) {
  DSM_REQUIRE(src && dst && src != dst, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, DSM_ERR_ARG);
  const long S = (long)D * H * W;
  DSM_REQUIRE(B <= 65535 && dsm_cdiv(C, 64) <= 65535, DSM_ERR_UNSUPPORTED);
  dim3 grid((unsigned)dsm_cdiv(S, 64), dsm_cdiv(C, 64), B);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(relayout_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src,
                     (float*)dst, C, S, to_ndhwc);
  return dsm_launch_status();
}
