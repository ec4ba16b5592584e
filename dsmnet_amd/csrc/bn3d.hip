// Train-mode BatchNorm3d + cropped skip addition + ReLU of the 3-D trunk, forward and backward,
// on NDHWC fp32 volumes -- the part of a training step that convbn_3d / conv3d_bn leave to
// nn.BatchNorm3d, myadd_3d and F.relu in the reference (models/psmnet/submodule.py:16-19,
// models/psmnet/stackhourglass.py:10-20,43-62, models/util_conv.py:150-179, util_fun.py:41-50).
// With stock torch ops this is ~6 launches per layer forward and ~8 backward (4 ms of BatchNorm +
// ~3 ms of element-wise kernels in a 32 ms PSMNet step at 256x512); here two launches each way:
//   forward : bn_stats (per-channel sum, sum of squares -> double atomics) -> bn_finalize
//             (mean, 1/std, folded scale/shift, running statistics) -> bn_apply
//             out = relu?( y*scale + shift (+ residual, cropped to the common corner) )
//   backward: bn_bwd_reduce (sum g', sum g' xhat) -> bn_bwd_apply (dy, dresidual); dgamma, dbeta
// All HBM-bound streaming passes: 16 bytes per lane, channels innermost.
#include "common.hpp"

namespace {

struct BnDims {
  int B, C;
  int Dy, Hy, Wy;      // the convolution output y
  int Dr, Hr, Wr;      // the residual (0 when absent)
  int Do, Ho, Wo;      // the block's output: the common corner
};

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}
__device__ __forceinline__ f32x4 mask4(f32x4 g, f32x4 ref) {      // g where ref > 0
  return f32x4{ref.x > 0.f ? g.x : 0.f, ref.y > 0.f ? g.y : 0.f, ref.z > 0.f ? g.z : 0.f, ref.w > 0.f ? g.w : 0.f};
}

// per-channel sum and sum of squares of y (B*Dy*Hy*Wy voxels x C): thread = (voxel lane, channel quad)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, double* __restrict__ stats,
                                                       long nvox, int C) {
  __shared__ float red[256 * 8];
  const int nq = C >> 2, nvl = 256 / nq;
  const int q = threadIdx.x % nq, vl = threadIdx.x / nq;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
  if (vl < nvl) {
    // four voxels per trip: four independent 16-byte loads in flight per lane (one load per trip left
    // this pass at 3 TB/s: 512 workgroups do not cover the memory latency by themselves)
    const long stride = (long)gridDim.x * nvl;
    long v = (long)blockIdx.x * nvl + vl;
    for (; v + 3 * stride < nvox; v += 4 * stride) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(y + v * C + 4 * q);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(y + (v + stride) * C + 4 * q);
      const f32x4 t2 = *reinterpret_cast<const f32x4*>(y + (v + 2 * stride) * C + 4 * q);
      const f32x4 t3 = *reinterpret_cast<const f32x4*>(y + (v + 3 * stride) * C + 4 * q);
      s += (t0 + t1) + (t2 + t3); ss += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
    }
    for (; v < nvox; v += stride) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(y + v * C + 4 * q);
      s += t; ss += t * t;
    }
  }
  float* r = red + threadIdx.x * 8;
  r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = s.w; r[4] = ss.x; r[5] = ss.y; r[6] = ss.z; r[7] = ss.w;
  __syncthreads();
  for (int u = threadIdx.x; u < 2 * C; u += 256) {
    const int which = u / C, c = u % C;
    double a = 0.0;
    for (int l = 0; l < nvl; ++l) a += (double)red[(l * nq + (c >> 2)) * 8 + which * 4 + (c & 3)];
    atomicAdd(stats + u, a);
  }
}

// stats -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; running statistics as
// nn.BatchNorm3d updates them (momentum, unbiased variance)
__global__ void bn_finalize_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean,
                                   float* __restrict__ rvar, float* __restrict__ out /*[4][C]*/, int C,
                                   double n, float momentum, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = stats[c] / n;
  double var = stats[C + c] / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float scale = g * invstd;
  out[c] = scale;
  out[C + c] = b - (float)mean * scale;
  out[2 * C + c] = (float)mean;
  out[3 * C + c] = invstd;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
}

// y, residual and output boxes coincide (no myadd_3d crop): every tensor shares one voxel index
__device__ __forceinline__ bool same_boxes(const BnDims& d, bool with_res) {
  return d.Do == d.Dy && d.Ho == d.Hy && d.Wo == d.Wy &&
         (!with_res || (d.Dr == d.Dy && d.Hr == d.Hy && d.Wr == d.Wy));
}
// quad index of flat element i: a mask when C / 4 is a power of two (every layer of these networks)
__device__ __forceinline__ int quad_of(long i, int nq) {
  return (nq & (nq - 1)) == 0 ? (int)(i & (nq - 1)) : (int)(i % nq);
}

// max |v| of a launch's output for the fp16 convolution modes (include/dsmnet_hip.h: x_amax): a wave
// reduction, then one atomic per wave -- and none once the slot holds as much (the blocks of an
// element-wise pass are many; most of them find the maximum already there)
__device__ __forceinline__ float amax4(float am, const f32x4 v) {
  return fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// (the apply kernels walk their elements with a bounded grid, so a launch issues at most a few
// thousand of these: same-address atomics serialise in L2 at ~5 ns each)
__device__ __forceinline__ void wave_amax(float* slot, float am) {
  if (!slot) return;
#pragma unroll
  for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
  if ((threadIdx.x & 63) == 0 && am > *reinterpret_cast<volatile float*>(slot))
    atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, am));
}
constexpr int BN_APPLY_MAX_BLOCKS = 4096;

// out[b,z,y,x,:] = relu?( y*scale + shift (+ res) ) over the common corner; thread = (out voxel, quad)
// relu: 0 none, 1 after the addition (PSMNet), 2 before it (GCNet)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                                       const float* __restrict__ aff, float* __restrict__ out,
                                                       BnDims d, int relu, float* __restrict__ out_amax) {
  const int nq = d.C >> 2;
  const long n = (long)d.B * d.Do * d.Ho * d.Wo * nq;
  float am = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
  const int q = quad_of(i, nq);
  long vy, vr;
  if (same_boxes(d, res != nullptr)) {            // no crop: one linear index for y, res and out
    vy = vr = i / nq;                              // (no coordinate decomposition: four 64-bit divisions per quad
  } else {                                         //  made these passes VALU-bound instead of HBM-bound)
    long v = i / nq;
    const int x = v % d.Wo; v /= d.Wo;
    const int yy = v % d.Ho; v /= d.Ho;
    const int z = v % d.Do; const int b = v / d.Do;
    vy = (((long)b * d.Dy + z) * d.Hy + yy) * d.Wy + x;
    vr = (((long)b * d.Dr + z) * d.Hr + yy) * d.Wr + x;
  }
  const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + 4 * q);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + d.C + 4 * q);
  f32x4 t = *reinterpret_cast<const f32x4*>(y + vy * d.C + 4 * q) * sc + sh;
  if (relu == 2) t = relu4(t);
  if (res) t += *reinterpret_cast<const f32x4*>(res + vr * d.C + 4 * q);
  if (relu == 1) t = relu4(t);
  *reinterpret_cast<f32x4*>(out + i * 4) = t;
  am = amax4(am, t);
  }
  wave_amax(out_amax, am);
}

// g' = g masked by the ReLU; per channel sum g' and sum g'*xhat over the output corner
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ out,
                                                            const float* __restrict__ g, const float* __restrict__ aff,
                                                            double* __restrict__ sums, BnDims d, int relu) {
  __shared__ float red[256 * 8];
  const int nq = d.C >> 2, nvl = 256 / nq;
  const int q = threadIdx.x % nq, vl = threadIdx.x / nq;
  const long nvox = (long)d.B * d.Do * d.Ho * d.Wo;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, sx = {0.f, 0.f, 0.f, 0.f};
  if (vl < nvl) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + 4 * q);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + d.C + 4 * q);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(aff + 2 * d.C + 4 * q);
    const f32x4 istd = *reinterpret_cast<const f32x4*>(aff + 3 * d.C + 4 * q);
    const bool flat = same_boxes(d, false);
    for (long vo = (long)blockIdx.x * nvl + vl; vo < nvox; vo += (long)gridDim.x * nvl) {
      long vy = vo;
      if (!flat) {
        long v = vo;
        const int x = v % d.Wo; v /= d.Wo;
        const int yy = v % d.Ho; v /= d.Ho;
        const int z = v % d.Do; const int b = v / d.Do;
        vy = (((long)b * d.Dy + z) * d.Hy + yy) * d.Wy + x;
      }
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + vy * d.C + 4 * q);
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + vo * d.C + 4 * q);
      if (relu == 1) gv = mask4(gv, *reinterpret_cast<const f32x4*>(out + vo * d.C + 4 * q));
      else if (relu == 2) gv = mask4(gv, yv * sc + sh);
      s += gv; sx += gv * ((yv - mean) * istd);
    }
  }
  float* r = red + threadIdx.x * 8;
  r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = s.w; r[4] = sx.x; r[5] = sx.y; r[6] = sx.z; r[7] = sx.w;
  __syncthreads();
  for (int u = threadIdx.x; u < 2 * d.C; u += 256) {
    const int which = u / d.C, c = u % d.C;
    double a = 0.0;
    for (int l = 0; l < nvl; ++l) a += (double)red[(l * nq + (c >> 2)) * 8 + which * 4 + (c & 3)];
    atomicAdd(sums + u, a);
  }
}

// dy over ALL of y (g' = 0 outside the corner): dy = scale * (g' - mean(g') - xhat * mean(g' xhat));
// dres over ALL of the residual: g (relu 2) or g' (relu 0, 1) inside the corner, 0 outside.
// thread = (voxel of the bounding box max(y, res), quad)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ out,
                                                           const float* __restrict__ g, const float* __restrict__ aff,
                                                           const double* __restrict__ sums, float* __restrict__ dy,
                                                           float* __restrict__ dres, BnDims d, int relu, double n,
                                                           float* __restrict__ dy_amax, float* __restrict__ dres_amax) {
  const int nq = d.C >> 2;
  const int Dm = max(d.Dy, d.Dr), Hm = max(d.Hy, d.Hr), Wm = max(d.Wy, d.Wr);
  const long total = (long)d.B * Dm * Hm * Wm * nq;
  float amy = 0.f, amr = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
  const int q = quad_of(i, nq);
  const bool flat = same_boxes(d, dres != nullptr || d.Dr > 0);
  bool in_o = true, in_y = true, in_r = dres != nullptr;
  long vy = i / nq, vo_ = vy, vr_ = vy;
  if (!flat) {
    long v = i / nq;
    const int x = v % Wm; v /= Wm;
    const int yy = v % Hm; v /= Hm;
    const int z = v % Dm; const int b = v / Dm;
    in_o = z < d.Do && yy < d.Ho && x < d.Wo;
    in_y = z < d.Dy && yy < d.Hy && x < d.Wy;
    in_r = dres && z < d.Dr && yy < d.Hr && x < d.Wr;
    vy = (((long)b * d.Dy + z) * d.Hy + yy) * d.Wy + x;
    vo_ = (((long)b * d.Do + z) * d.Ho + yy) * d.Wo + x;
    vr_ = (((long)b * d.Dr + z) * d.Hr + yy) * d.Wr + x;
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 gv = zero, graw = zero, yv = zero;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + 4 * q);
  if (in_y) yv = *reinterpret_cast<const f32x4*>(y + vy * d.C + 4 * q);
  if (in_o) {
    const long vo = vo_;
    graw = *reinterpret_cast<const f32x4*>(g + vo * d.C + 4 * q);
    gv = graw;
    if (relu == 1) gv = mask4(gv, *reinterpret_cast<const f32x4*>(out + vo * d.C + 4 * q));
    else if (relu == 2) gv = mask4(gv, yv * sc + *reinterpret_cast<const f32x4*>(aff + d.C + 4 * q));
  }
  if (in_y) {
    const f32x4 mean = *reinterpret_cast<const f32x4*>(aff + 2 * d.C + 4 * q);
    const f32x4 istd = *reinterpret_cast<const f32x4*>(aff + 3 * d.C + 4 * q);
    f32x4 mg, mgx;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mg[k] = (float)(sums[4 * q + k] / n); mgx[k] = (float)(sums[d.C + 4 * q + k] / n); }
    const f32x4 xhat = (yv - mean) * istd;
    const f32x4 dyv = sc * (gv - mg - xhat * mgx);
    *reinterpret_cast<f32x4*>(dy + vy * d.C + 4 * q) = dyv;
    amy = amax4(amy, dyv);
  }
  if (in_r) {
    const long vr = vr_;
    const f32x4 drv = relu == 2 ? graw : gv;
    *reinterpret_cast<f32x4*>(dres + vr * d.C + 4 * q) = drv;
    amr = amax4(amr, drv);
  }
  }
  wave_amax(dy_amax, amy);
  wave_amax(dres_amax, amr);
}

int check_bn(const dsm_bn3d_args* a) {
  DSM_REQUIRE(a && a->y && a->workspace && a->affine, DSM_ERR_ARG);
  DSM_REQUIRE(a->B > 0 && a->C > 0 && a->Dy > 0 && a->Hy > 0 && a->Wy > 0, DSM_ERR_ARG);
  DSM_REQUIRE(a->C % 4 == 0 && a->C <= 256, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(a->relu >= 0 && a->relu <= 2, DSM_ERR_ARG);
  if (a->residual || a->dresidual) DSM_REQUIRE(a->Dr > 0 && a->Hr > 0 && a->Wr > 0, DSM_ERR_ARG);
  DSM_REQUIRE(dsm_aligned16(a->y) && dsm_aligned16(a->residual) && dsm_aligned16(a->out) &&
              dsm_aligned16(a->affine) && dsm_aligned16(a->gout) && dsm_aligned16(a->dy) &&
              dsm_aligned16(a->dresidual), DSM_ERR_ALIGN);
  return DSM_OK;
}

BnDims dims_of(const dsm_bn3d_args* a) {
  BnDims d;
  d.B = a->B; d.C = a->C; d.Dy = a->Dy; d.Hy = a->Hy; d.Wy = a->Wy;
  const bool r = a->residual || a->dresidual;
  d.Dr = r ? a->Dr : 0; d.Hr = r ? a->Hr : 0; d.Wr = r ? a->Wr : 0;
  d.Do = r ? (a->Dy < a->Dr ? a->Dy : a->Dr) : a->Dy;
  d.Ho = r ? (a->Hy < a->Hr ? a->Hy : a->Hr) : a->Hy;
  d.Wo = r ? (a->Wy < a->Wr ? a->Wy : a->Wr) : a->Wy;
  return d;
}

}  // namespace

extern "C" int dsm_bn3d_train_fwd(const dsm_bn3d_args* a, dsm_stream_t stream) {
  int rc = check_bn(a);
  if (rc != DSM_OK) return rc;
  DSM_REQUIRE(a->out, DSM_ERR_ARG);
  const BnDims d = dims_of(a);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  double* stats = (double*)a->workspace;
  if (hipMemsetAsync(stats, 0, (size_t)2 * a->C * sizeof(double), s) != hipSuccess) return DSM_ERR_LAUNCH;
  const long nvox = (long)a->B * a->Dy * a->Hy * a->Wy;
  const int nvl = 256 / (a->C / 4);
  long blocks = (nvox + nvl - 1) / nvl;
  if (blocks > 512) blocks = 512;        // 2 C double atomics per block: keep their serialisation per address short
  hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)a->y, stats, nvox, a->C);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(dsm_cdiv(a->C, 64)), dim3(64), 0, s, (const double*)stats,
                     a->gamma, a->beta, a->running_mean, a->running_var, (float*)a->affine, a->C,
                     (double)nvox, a->momentum, a->eps);
  const long n = (long)d.B * d.Do * d.Ho * d.Wo * (d.C / 4);
  DSM_REQUIRE(n / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  // a bounded grid when a maximum is asked for (one atomic per wave at most)
  const long ablocks = dsm_cdiv(n, 256);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)(a->out_amax && ablocks > BN_APPLY_MAX_BLOCKS ? BN_APPLY_MAX_BLOCKS : ablocks)), dim3(256), 0, s, (const float*)a->y,
                     (const float*)a->residual, (const float*)a->affine, (float*)a->out, d, a->relu, a->out_amax);
  return dsm_launch_status();
}

extern "C" int dsm_bn3d_train_bwd(const dsm_bn3d_args* a, dsm_stream_t stream) {
  int rc = check_bn(a);
  if (rc != DSM_OK) return rc;
  DSM_REQUIRE(a->gout && a->dy && (a->relu != 1 || a->out), DSM_ERR_ARG);
  const BnDims d = dims_of(a);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  double* sums = (double*)a->workspace;
  if (hipMemsetAsync(sums, 0, (size_t)2 * a->C * sizeof(double), s) != hipSuccess) return DSM_ERR_LAUNCH;
  const long nvo = (long)d.B * d.Do * d.Ho * d.Wo;
  const int nvl = 256 / (a->C / 4);
  long blocks = (nvo + nvl - 1) / nvl;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)a->y,
                     (const float*)a->out, (const float*)a->gout, (const float*)a->affine, sums, d, a->relu);
  const int Dm = d.Dy > d.Dr ? d.Dy : d.Dr, Hm = d.Hy > d.Hr ? d.Hy : d.Hr, Wm = d.Wy > d.Wr ? d.Wy : d.Wr;
  const long total = (long)d.B * Dm * Hm * Wm * (d.C / 4);
  DSM_REQUIRE(total / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  const double n = (double)a->B * a->Dy * a->Hy * a->Wy;
  const long bblocks = dsm_cdiv(total, 256);
  const bool want_max = a->dy_amax || a->dres_amax;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)(want_max && bblocks > BN_APPLY_MAX_BLOCKS ? BN_APPLY_MAX_BLOCKS : bblocks)), dim3(256), 0, s, (const float*)a->y,
                     (const float*)a->out, (const float*)a->gout, (const float*)a->affine, (const double*)sums,
                     (float*)a->dy, (float*)a->dresidual, d, a->relu, n, a->dy_amax, a->dres_amax);
  return dsm_launch_status();
}
