// Fused soft-argmin disparity regression for gfx950.
//
// Replaces, per head, F.upsample(trilinear) -> F.softmax -> permute -> matmul(arange)
// (models/psmnet/stackhourglass.py:152-166, submodule.py:56-63) and
// Softmax2d(-x) -> permute -> matmul (models/gcnet.py:104-111): four-plus passes
// over a (B,D,H,W) tensor become one kernel that interpolates on the fly, keeps an
// online softmax in registers and writes only the (B,H,W) disparity.
//
// Lane layout: a wave covers 64/DSPLIT consecutive x of one row times DSPLIT
// disparity segments; the DSPLIT partial (max, sum, weighted sum) triples of a
// pixel are merged with wave shuffles (__shfl_xor), no LDS.  Lanes along x keep
// the (B,D,H,W) reads of the GCNet form coalesced.
//
// Algorithmic bytes: fused PSMNet head 384x1280 = 5.90 MB
// cost read + 1.97 MB disparity written; GCNet 256x512 = 100.7 MB read + 0.5 MB.
#include "common.hpp"

struct Lerp { int i0, i1; float w0, w1; };

// torch's area_pixel_compute_source_index + guard_index_and_lambda
// (aten/src/ATen/native/UpSample.h), which is what F.upsample/F.interpolate run.
__device__ __forceinline__ Lerp lerp_at(int o, float scale, int in_size, int out_size,
                                        int align) {
  Lerp r;
  if (in_size == out_size) { r.i0 = r.i1 = o; r.w0 = 1.f; r.w1 = 0.f; return r; }
  float src = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
  int i0 = min((int)floorf(src), in_size - 1);
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  r.w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  r.w0 = 1.f - r.w1;
  return r;
}

struct Sm { float m, l, s; };   // running max, sum exp, sum d*exp

__device__ __forceinline__ void sm_push(Sm& a, float v, float d) {
  if (v > a.m) {                       // rare after the first few bins
    const float r = __expf(a.m - v);   // exp(-inf) = 0 on the first bin
    a.l *= r; a.s *= r; a.m = v;
  }
  const float e = __expf(v - a.m);
  a.l += e;
  a.s = fmaf(d, e, a.s);
}

__device__ __forceinline__ Sm sm_merge(const Sm& a, const Sm& b) {
  Sm o;
  o.m = fmaxf(a.m, b.m);
  const float ra = (a.m == -INFINITY) ? 0.f : __expf(a.m - o.m);
  const float rb = (b.m == -INFINITY) ? 0.f : __expf(b.m - o.m);
  o.l = a.l * ra + b.l * rb;
  o.s = a.s * ra + b.s * rb;
  return o;
}

struct SaParams {
  const float* cost; float* disp; float* stats;
  int Dc, Hc, Wc, D, H, W;
  float sd, sh, sw;      // source-index scales (torch: in/out, or (in-1)/(out-1))
  float sign;            // +1 (PSMNet) / -1 (GCNet)
  int align;
};

// bilinear sample of coarse plane k at the pixel's (y, x) stencil
struct Stencil { int o00, o01, o10, o11; float wy0, wy1, wx0, wx1; };
__device__ __forceinline__ float plane_at(const float* __restrict__ base, long plane_stride, int k,
                                          const Stencil& st) {
  const float* p = base + (long)k * plane_stride;
  return st.wy0 * (st.wx0 * p[st.o00] + st.wx1 * p[st.o01]) +
         st.wy1 * (st.wx0 * p[st.o10] + st.wx1 * p[st.o11]);
}

template <bool UPSAMPLE, int DSPLIT>
__global__ __launch_bounds__(256) void soft_argmin_fwd_kernel(SaParams p) {
  constexpr int PXW = DSM_WAVE / DSPLIT;          // pixels per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane / PXW;
  const int x = (blockIdx.x * 4 + wave) * PXW + (lane % PXW);
  const int y = blockIdx.y, b = blockIdx.z;
  const bool live = x < p.W;
  const int dper = (p.D + DSPLIT - 1) / DSPLIT;
  const int d_lo = seg * dper, d_hi = min(p.D, d_lo + dper);
  Sm acc = {-INFINITY, 0.f, 0.f};
  if (live) {
    if (UPSAMPLE) {
      const Lerp ly = lerp_at(y, p.sh, p.Hc, p.H, p.align);
      const Lerp lx = lerp_at(x, p.sw, p.Wc, p.W, p.align);
      Stencil st;
      st.o00 = ly.i0 * p.Wc + lx.i0; st.o01 = ly.i0 * p.Wc + lx.i1;
      st.o10 = ly.i1 * p.Wc + lx.i0; st.o11 = ly.i1 * p.Wc + lx.i1;
      st.wy0 = ly.w0; st.wy1 = ly.w1; st.wx0 = lx.w0; st.wx1 = lx.w1;
      const long ps = (long)p.Hc * p.Wc;
      const float* base = p.cost + (long)b * p.Dc * ps;
      int k = -2; float P0 = 0.f, P1 = 0.f;          // planes k and min(k+1, Dc-1)
      for (int d = d_lo; d < d_hi; ++d) {
        const Lerp ld = lerp_at(d, p.sd, p.Dc, p.D, p.align);
        if (ld.i0 != k) {
          if (ld.i0 == k + 1) { P0 = P1; }
          else { P0 = plane_at(base, ps, ld.i0, st); }
          k = ld.i0;
          P1 = plane_at(base, ps, min(k + 1, p.Dc - 1), st);
        }
        const float v = ld.w0 * P0 + ld.w1 * (ld.i1 == k ? P0 : P1);
        sm_push(acc, p.sign * v, (float)d);
      }
    } else {
      const long ps = (long)p.H * p.W;
      const float* src = p.cost + (long)b * p.D * ps + (long)y * p.W + x;
#pragma unroll 4
      for (int d = d_lo; d < d_hi; ++d) sm_push(acc, p.sign * src[d * ps], (float)d);
    }
  }
#pragma unroll
  for (int off = PXW; off < DSM_WAVE; off <<= 1) {
    Sm o;
    o.m = __shfl_xor(acc.m, off); o.l = __shfl_xor(acc.l, off); o.s = __shfl_xor(acc.s, off);
    acc = sm_merge(acc, o);
  }
  if (live && seg == 0) {
    const long o = ((long)b * p.H + y) * p.W + x;
    p.disp[o] = acc.s / acc.l;
    if (p.stats) {
      const long hw = (long)p.H * p.W;
      p.stats[(long)b * 2 * hw + (long)y * p.W + x] = acc.m;
      p.stats[(long)b * 2 * hw + hw + (long)y * p.W + x] = acc.l;
    }
  }
}

// Fast path for the PSMNet head: D == 4*Dc, align_corners = 0.  torch's source index
// (d + 0.5)/4 - 0.5 makes the four fine bins of coarse interval k
//   v(4k+0) = .375 P[k-1] + .625 P[k]     v(4k+1) = .125 P[k-1] + .875 P[k]
//   v(4k+2) = .875 P[k]   + .125 P[k+1]   v(4k+3) = .625 P[k]   + .375 P[k+1]
// with P[-1] := P[0], P[Dc] := P[Dc-1] (the clamps), P[k] the pixel's bilinear sample of
// coarse plane k.  No per-bin index arithmetic; one pass with a per-PLANE running maximum (see
// the loop) replaces both the branchy per-bin online update and r01's exact two-pass form.
template <int DSPLIT>
__global__ __launch_bounds__(256) void soft_argmin_up4_kernel(SaParams p) {
  constexpr int PXW = DSM_WAVE / DSPLIT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane / PXW;
  const int x = (blockIdx.x * 4 + wave) * PXW + (lane % PXW);
  const int y = blockIdx.y, b = blockIdx.z;
  const bool live = x < p.W;
  const int kper = (p.Dc + DSPLIT - 1) / DSPLIT;
  const int k_lo = seg * kper, k_hi = min(p.Dc, k_lo + kper);
  Sm acc = {-INFINITY, 0.f, 0.f};
  if (live && k_lo < k_hi) {
    const Lerp ly = lerp_at(y, p.sh, p.Hc, p.H, 0);
    const Lerp lx = lerp_at(x, p.sw, p.Wc, p.W, 0);
    Stencil st;
    st.o00 = ly.i0 * p.Wc + lx.i0; st.o01 = ly.i0 * p.Wc + lx.i1;
    st.o10 = ly.i1 * p.Wc + lx.i0; st.o11 = ly.i1 * p.Wc + lx.i1;
    st.wy0 = ly.w0; st.wy1 = ly.w1; st.wx0 = lx.w0; st.wx1 = lx.w1;
    const long ps = (long)p.Hc * p.Wc;
    const float* base = p.cost + (long)b * p.Dc * ps;
    const float sg = p.sign;
    // ONE pass (r02; the r01 kernel evaluated every bilinear sample and lerp twice: an exact-max
    // pass, then the sums).  Every fine value of coarse interval k is a convex combination of
    // P[k-1], P[k], P[k+1], so the running maximum of the P's seen so far -- taken BEFORE interval
    // k is summed, P[k+1] included -- bounds every exponent by 0: the sums are rescaled by
    // exp(m_old - m_new) whenever a larger P arrives (one extra v_exp per plane, 5 instead of
    // 4 + a second sampling pass).  Mathematically the softmax is shift-invariant; numerically
    // each rescale costs one rounding of l and sum, and it happens O(log Dc) times on average.
    // The loop works in base-2 units (samples pre-multiplied by log2 e: v_exp_f32 is 2^x) and on
    // PAIRS of fine bins -- (4k, 4k+1) and (4k+2, 4k+3) -- with packed fp32 instructions
    // (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): ~27 vector operations + 5 v_exp per coarse
    // plane instead of ~45 + 5 (the kernel is VALU-bound).  l and sum are kept as (even | odd) bin
    // pairs and folded at the end.
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const float sg2 = sg * LOG2E;
    float Pm = sg2 * plane_at(base, ps, max(k_lo - 1, 0), st);
    float Pc = sg2 * plane_at(base, ps, k_lo, st);
    float m = fmaxf(Pm, Pc);
    f2 l2 = {0.f, 0.f}, s2 = {0.f, 0.f};
    f2 d01 = {(float)(4 * k_lo), (float)(4 * k_lo + 1)}, d23 = {(float)(4 * k_lo + 2), (float)(4 * k_lo + 3)};
    const f2 ca = {0.375f, 0.125f}, cb = {0.625f, 0.875f}, four = {4.f, 4.f};
    for (int k = k_lo; k < k_hi; ++k) {
      const float Pn = sg2 * plane_at(base, ps, min(k + 1, p.Dc - 1), st);
      const float mn = fmaxf(m, Pn);
      const float rs = __builtin_amdgcn_exp2f(m - mn);     // 1 when the maximum stands
      l2 *= f2{rs, rs}; s2 *= f2{rs, rs}; m = mn;
      const f2 nm = {-m, -m};
      // (v0, v1) = (.375, .125) Pm + (.625, .875) Pc - m;  (v3, v2) = (.375, .125) Pn + (.625, .875) Pc - m
      const f2 v01 = __builtin_elementwise_fma(ca, f2{Pm, Pm}, __builtin_elementwise_fma(cb, f2{Pc, Pc}, nm));
      const f2 v32 = __builtin_elementwise_fma(ca, f2{Pn, Pn}, __builtin_elementwise_fma(cb, f2{Pc, Pc}, nm));
      const f2 e01 = {__builtin_amdgcn_exp2f(v01.x), __builtin_amdgcn_exp2f(v01.y)};
      const f2 e23 = {__builtin_amdgcn_exp2f(v32.y), __builtin_amdgcn_exp2f(v32.x)};
      l2 += e01; l2 += e23;
      s2 = __builtin_elementwise_fma(d01, e01, s2);
      s2 = __builtin_elementwise_fma(d23, e23, s2);
      d01 += four; d23 += four;
      Pm = Pc; Pc = Pn;
    }
    float l = l2.x + l2.y, sum = s2.x + s2.y;
    m *= LN2;                                              // back to natural units (stats, merge)
    // Guard: the running maximum of the P's can exceed every FINE value by more than the range
    // of expf (an isolated peak of height h leaves 0.875 h as the largest fine value): all terms
    // then underflow.  Such a segment (l tiny or zero) is redone with the exact maximum of its
    // fine values -- the r01 two-pass form; rare, and only ill-conditioned costs reach it.
    if (!(l > 1e-30f)) {
      Pm = sg * plane_at(base, ps, max(k_lo - 1, 0), st);
      Pc = sg * plane_at(base, ps, k_lo, st);
      m = -INFINITY;
      for (int k = k_lo; k < k_hi; ++k) {
        const float Pn = sg * plane_at(base, ps, min(k + 1, p.Dc - 1), st);
        const float v0 = 0.375f * Pm + 0.625f * Pc, v1 = 0.125f * Pm + 0.875f * Pc;
        const float v2 = 0.875f * Pc + 0.125f * Pn, v3 = 0.625f * Pc + 0.375f * Pn;
        m = fmaxf(m, fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
        Pm = Pc; Pc = Pn;
      }
      Pm = sg * plane_at(base, ps, max(k_lo - 1, 0), st);
      Pc = sg * plane_at(base, ps, k_lo, st);
      l = 0.f; sum = 0.f;
      for (int k = k_lo; k < k_hi; ++k) {
        const float Pn = sg * plane_at(base, ps, min(k + 1, p.Dc - 1), st);
        const float d0 = (float)(4 * k);
        const float e0 = __expf(0.375f * Pm + 0.625f * Pc - m);
        const float e1 = __expf(0.125f * Pm + 0.875f * Pc - m);
        const float e2 = __expf(0.875f * Pc + 0.125f * Pn - m);
        const float e3 = __expf(0.625f * Pc + 0.375f * Pn - m);
        l += (e0 + e1) + (e2 + e3);
        sum = fmaf(d0, e0, sum); sum = fmaf(d0 + 1.f, e1, sum);
        sum = fmaf(d0 + 2.f, e2, sum); sum = fmaf(d0 + 3.f, e3, sum);
        Pm = Pc; Pc = Pn;
      }
    }
    acc.m = m; acc.l = l; acc.s = sum;
  }
#pragma unroll
  for (int off = PXW; off < DSM_WAVE; off <<= 1) {
    Sm o;
    o.m = __shfl_xor(acc.m, off); o.l = __shfl_xor(acc.l, off); o.s = __shfl_xor(acc.s, off);
    acc = sm_merge(acc, o);
  }
  if (live && seg == 0) {
    const long o = ((long)b * p.H + y) * p.W + x;
    p.disp[o] = acc.s / acc.l;
    if (p.stats) {
      const long hw = (long)p.H * p.W;
      p.stats[(long)b * 2 * hw + (long)y * p.W + x] = acc.m;
      p.stats[(long)b * 2 * hw + hw + (long)y * p.W + x] = acc.l;
    }
  }
}

// Backward: dcost_fine[d] = sign * p_d * (d - E) * g, then (PSMNet form) the adjoint
// of the trilinear stencil.  One thread per full-resolution pixel walks d; the
// contributions to a coarse plane are summed in registers and leave as 4 float
// atomics per (pixel, coarse plane).
struct SaBwdParams {
  const float* cost; const float* disp; const float* stats; const float* gdisp; float* dcost;
  int Dc, Hc, Wc, D, H, W;
  float sd, sh, sw, sign;
  int align;
};

__device__ __forceinline__ void scatter4(float* base, long ps, int k, const Stencil& st, float a) {
  float* q = base + (long)k * ps;
  atomicAdd(q + st.o00, a * st.wy0 * st.wx0);
  atomicAdd(q + st.o01, a * st.wy0 * st.wx1);
  atomicAdd(q + st.o10, a * st.wy1 * st.wx0);
  atomicAdd(q + st.o11, a * st.wy1 * st.wx1);
}

template <bool UPSAMPLE>
__global__ __launch_bounds__(256) void soft_argmin_bwd_kernel(SaBwdParams p) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= p.W) return;
  const long hw = (long)p.H * p.W;
  const long o = (long)y * p.W + x;
  const float m = p.stats[(long)b * 2 * hw + o];
  const float inv_l = 1.f / p.stats[(long)b * 2 * hw + hw + o];
  const float E = p.disp[(long)b * hw + o];
  const float g = p.gdisp[(long)b * hw + o] * p.sign;
  if (UPSAMPLE) {
    const Lerp ly = lerp_at(y, p.sh, p.Hc, p.H, p.align);
    const Lerp lx = lerp_at(x, p.sw, p.Wc, p.W, p.align);
    Stencil st;
    st.o00 = ly.i0 * p.Wc + lx.i0; st.o01 = ly.i0 * p.Wc + lx.i1;
    st.o10 = ly.i1 * p.Wc + lx.i0; st.o11 = ly.i1 * p.Wc + lx.i1;
    st.wy0 = ly.w0; st.wy1 = ly.w1; st.wx0 = lx.w0; st.wx1 = lx.w1;
    const long ps = (long)p.Hc * p.Wc;
    const float* base = p.cost + (long)b * p.Dc * ps;
    float* gbase = p.dcost + (long)b * p.Dc * ps;
    int k = -2; float P0 = 0.f, P1 = 0.f, A0 = 0.f, A1 = 0.f;
    for (int d = 0; d < p.D; ++d) {
      const Lerp ld = lerp_at(d, p.sd, p.Dc, p.D, p.align);
      if (ld.i0 != k) {
        if (k >= 0) {
          scatter4(gbase, ps, k, st, A0);
          if (ld.i0 != k + 1 && k + 1 < p.Dc) scatter4(gbase, ps, k + 1, st, A1);
        }
        if (ld.i0 == k + 1) { P0 = P1; A0 = A1; }
        else { P0 = plane_at(base, ps, ld.i0, st); A0 = 0.f; }
        k = ld.i0; A1 = 0.f;
        P1 = plane_at(base, ps, min(k + 1, p.Dc - 1), st);
      }
      const bool same = (ld.i1 == k);
      const float v = p.sign * (ld.w0 * P0 + ld.w1 * (same ? P0 : P1));
      const float gf = __expf(v - m) * inv_l * ((float)d - E) * g;
      A0 = fmaf(gf, same ? 1.f : ld.w0, A0);
      if (!same) A1 = fmaf(gf, ld.w1, A1);
    }
    if (k >= 0) {
      scatter4(gbase, ps, k, st, A0);
      if (k + 1 < p.Dc) scatter4(gbase, ps, k + 1, st, A1);
    }
  } else {
    const float* src = p.cost + (long)b * p.D * hw + o;
    float* dst = p.dcost + (long)b * p.D * hw + o;
#pragma unroll 4
    for (int d = 0; d < p.D; ++d) {
      const float v = p.sign * src[d * hw];
      dst[d * hw] = __expf(v - m) * inv_l * ((float)d - E) * g;
    }
  }
}

// The same adjoint with the scatter kept on chip: a workgroup owns a 32 x 8 tile of full-resolution
// pixels and a segment of `dseg` disparities; the coarse cells that tile touches ((cw x ch) cells x
// nk planes) are accumulated in LDS with ds_add_f32 and leave as ONE global atomic per cell --
// ~50x fewer global atomics than the per-pixel scatter above (which took 0.54 ms per head at
// 256 x 512, D = 192: profiles/r02_train_kernel_stats.csv).
constexpr int SAB_TX = 32, SAB_TY = 8;
__global__ __launch_bounds__(256) void soft_argmin_bwd_tile_kernel(SaBwdParams p, int dseg, int nseg) {
  extern __shared__ float cell[];
  const int tx = threadIdx.x & (SAB_TX - 1), ty = threadIdx.x / SAB_TX;
  const int x0 = blockIdx.x * SAB_TX, y0 = blockIdx.y * SAB_TY;
  const int b = blockIdx.z / nseg, seg = blockIdx.z % nseg;
  const int d0 = seg * dseg, d1 = min(p.D, d0 + dseg);
  // the coarse box of this (tile, segment): the interpolation indices are monotonic
  const int cx0 = lerp_at(x0, p.sw, p.Wc, p.W, p.align).i0;
  const int cx1 = lerp_at(min(x0 + SAB_TX, p.W) - 1, p.sw, p.Wc, p.W, p.align).i1;
  const int cy0 = lerp_at(y0, p.sh, p.Hc, p.H, p.align).i0;
  const int cy1 = lerp_at(min(y0 + SAB_TY, p.H) - 1, p.sh, p.Hc, p.H, p.align).i1;
  const int k0 = lerp_at(d0, p.sd, p.Dc, p.D, p.align).i0;
  const int k1 = lerp_at(d1 - 1, p.sd, p.Dc, p.D, p.align).i1;
  const int cw = cx1 - cx0 + 1, ch = cy1 - cy0 + 1, nk = k1 - k0 + 1;
  const int pc = cw * ch, n = pc * nk;
  for (int i = threadIdx.x; i < n; i += 256) cell[i] = 0.f;
  __syncthreads();
  const int x = x0 + tx, y = y0 + ty;
  if (x < p.W && y < p.H) {
    const long hw = (long)p.H * p.W;
    const long o = (long)y * p.W + x;
    const float m = p.stats[(long)b * 2 * hw + o];
    const float inv_l = 1.f / p.stats[(long)b * 2 * hw + hw + o];
    const float E = p.disp[(long)b * hw + o];
    const float g = p.gdisp[(long)b * hw + o] * p.sign;
    const Lerp ly = lerp_at(y, p.sh, p.Hc, p.H, p.align);
    const Lerp lx = lerp_at(x, p.sw, p.Wc, p.W, p.align);
    Stencil st;
    st.o00 = ly.i0 * p.Wc + lx.i0; st.o01 = ly.i0 * p.Wc + lx.i1;
    st.o10 = ly.i1 * p.Wc + lx.i0; st.o11 = ly.i1 * p.Wc + lx.i1;
    st.wy0 = ly.w0; st.wy1 = ly.w1; st.wx0 = lx.w0; st.wx1 = lx.w1;
    const int l00 = (ly.i0 - cy0) * cw + (lx.i0 - cx0), l01 = (ly.i0 - cy0) * cw + (lx.i1 - cx0);
    const int l10 = (ly.i1 - cy0) * cw + (lx.i0 - cx0), l11 = (ly.i1 - cy0) * cw + (lx.i1 - cx0);
    // Pixels of one row that share the coarse column pair (lx.i0, lx.i1) are a contiguous run of lanes
    // (the indices are monotonic in x; four lanes at the x4 upsampling).  Their contributions are
    // summed across the run first (segmented suffix sums by doubling, 2 x 5 lane shuffles) and only the
    // run's first lane adds to the cell: 4-way fewer LDS atomics on the same address per row, and the
    // two rows of a wave that share a coarse row are all that is left to collide.
    bool same[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int o = 1 << j;
      const int ki = __shfl_down(lx.i0, o, 32), kj = __shfl_down(lx.i1, o, 32);
      same[j] = tx + o < SAB_TX && ki == lx.i0 && kj == lx.i1 && x + o < p.W;
    }
    const int kp0 = __shfl_up(lx.i0, 1, 32), kp1 = __shfl_up(lx.i1, 1, 32);
    const bool leader = tx == 0 || kp0 != lx.i0 || kp1 != lx.i1;
    auto scatter_lds = [&](int k, float a) {
      float r0 = a * st.wx0, r1 = a * st.wx1;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float o0 = __shfl_down(r0, 1 << j, 32), o1 = __shfl_down(r1, 1 << j, 32);
        if (same[j]) { r0 += o0; r1 += o1; }
      }
      if (!leader) return;
      float* q = cell + (k - k0) * pc;
      __hip_atomic_fetch_add(q + l00, r0 * st.wy0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(q + l01, r1 * st.wy0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(q + l10, r0 * st.wy1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(q + l11, r1 * st.wy1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const long ps = (long)p.Hc * p.Wc;
    const float* base = p.cost + (long)b * p.Dc * ps;
    int k = -2; float P0 = 0.f, P1 = 0.f, A0 = 0.f, A1 = 0.f;
    for (int d = d0; d < d1; ++d) {
      const Lerp ld = lerp_at(d, p.sd, p.Dc, p.D, p.align);
      if (ld.i0 != k) {
        if (k >= 0) {
          scatter_lds(k, A0);
          if (ld.i0 != k + 1 && k + 1 < p.Dc) scatter_lds(k + 1, A1);
        }
        if (ld.i0 == k + 1) { P0 = P1; A0 = A1; }
        else { P0 = plane_at(base, ps, ld.i0, st); A0 = 0.f; }
        k = ld.i0; A1 = 0.f;
        P1 = plane_at(base, ps, min(k + 1, p.Dc - 1), st);
      }
      const bool same = (ld.i1 == k);
      const float v = p.sign * (ld.w0 * P0 + ld.w1 * (same ? P0 : P1));
      const float gf = __expf(v - m) * inv_l * ((float)d - E) * g;
      A0 = fmaf(gf, same ? 1.f : ld.w0, A0);
      if (!same) A1 = fmaf(gf, ld.w1, A1);
    }
    if (k >= 0) {
      scatter_lds(k, A0);
      if (k + 1 <= k1) scatter_lds(k + 1, A1);
    }
  }
  __syncthreads();
  float* gbase = p.dcost + (long)b * p.Dc * p.Hc * p.Wc;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float a = cell[i];
    if (a == 0.f) continue;
    const int kk = i / pc, r = i % pc;
    atomicAdd(gbase + ((long)(k0 + kk) * p.Hc + cy0 + r / cw) * p.Wc + cx0 + r % cw, a);
  }
}

static float src_scale(int in, int out, int align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}

static int check_sa(const void* cost, const void* disp, int B, int Dc, int Hc, int Wc, int D,
                    int H, int W, int dtype) {
  DSM_REQUIRE(cost && disp, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && D > 0 && H > 0 && W > 0, DSM_ERR_ARG);
  DSM_REQUIRE(dtype == DSM_F32, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(H <= 65535 && B <= 65535, DSM_ERR_UNSUPPORTED);
  return DSM_OK;
}

// Disparity segments per pixel handled by separate lanes (merged with __shfl_xor): 4, fewer only
// for very short disparity ranges.  (A/B history, r01: 4 beat 2 and 1 at D = 192)
static int pick_dsplit(int D) {
  int ds = 4;
  while (ds > 1 && D < 2 * ds) ds >>= 1;
  return ds;
}

extern "C" int dsm_soft_argmin_fwd(const void* cost, void* disp, void* stats, int B, int Dc,
                                   int Hc, int Wc, int D, int H, int W, int negate,
                                   int align_corners, int dtype, dsm_stream_t stream) {
  int rc = check_sa(cost, disp, B, Dc, Hc, Wc, D, H, W, dtype);
  if (rc != DSM_OK) return rc;
  SaParams p;
  p.cost = (const float*)cost; p.disp = (float*)disp; p.stats = (float*)stats;
  p.Dc = Dc; p.Hc = Hc; p.Wc = Wc; p.D = D; p.H = H; p.W = W;
  p.sd = src_scale(Dc, D, align_corners); p.sh = src_scale(Hc, H, align_corners);
  p.sw = src_scale(Wc, W, align_corners);
  p.sign = negate ? -1.f : 1.f; p.align = align_corners;
  const bool up = !(Dc == D && Hc == H && Wc == W);
  const int ds = pick_dsplit(D);
  const int px_per_block = 4 * (DSM_WAVE / ds);
  dim3 grid(dsm_cdiv(W, px_per_block), H, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
#define SA_LAUNCH(UP, DS) hipLaunchKernelGGL((soft_argmin_fwd_kernel<UP, DS>), grid, block, 0, s, p)
  if (up && D == 4 * Dc && !align_corners && Dc >= ds) {
    if (ds == 4) hipLaunchKernelGGL(soft_argmin_up4_kernel<4>, grid, block, 0, s, p);
    else if (ds == 2) hipLaunchKernelGGL(soft_argmin_up4_kernel<2>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(soft_argmin_up4_kernel<1>, grid, block, 0, s, p);
  } else if (up) { if (ds == 4) SA_LAUNCH(true, 4); else if (ds == 2) SA_LAUNCH(true, 2); else SA_LAUNCH(true, 1); }
  else    { if (ds == 4) SA_LAUNCH(false, 4); else if (ds == 2) SA_LAUNCH(false, 2); else SA_LAUNCH(false, 1); }
#undef SA_LAUNCH
  return dsm_launch_status();
}

extern "C" int dsm_soft_argmin_bwd(const void* cost, const void* disp, const void* stats,
                                   const void* gdisp, void* dcost, int B, int Dc, int Hc, int Wc,
                                   int D, int H, int W, int negate, int align_corners, int dtype,
                                   dsm_stream_t stream) {
  int rc = check_sa(cost, disp, B, Dc, Hc, Wc, D, H, W, dtype);
  if (rc != DSM_OK) return rc;
  DSM_REQUIRE(stats && gdisp && dcost, DSM_ERR_ARG);
  SaBwdParams p;
  p.cost = (const float*)cost; p.disp = (const float*)disp; p.stats = (const float*)stats;
  p.gdisp = (const float*)gdisp; p.dcost = (float*)dcost;
  p.Dc = Dc; p.Hc = Hc; p.Wc = Wc; p.D = D; p.H = H; p.W = W;
  p.sd = src_scale(Dc, D, align_corners); p.sh = src_scale(Hc, H, align_corners);
  p.sw = src_scale(Wc, W, align_corners);
  p.sign = negate ? -1.f : 1.f; p.align = align_corners;
  const bool up = !(Dc == D && Hc == H && Wc == W);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  dim3 grid(dsm_cdiv(W, 256), H, B), block(256);
  if (up) {
    if (hipMemsetAsync(dcost, 0, (size_t)B * Dc * Hc * Wc * sizeof(float), s) != hipSuccess)
      return DSM_ERR_LAUNCH;
    // LDS-tiled adjoint when the tile's coarse box fits (always, when upsampling)
    int nseg = D >= 96 ? 4 : (D >= 16 ? 2 : 1);
    const int dseg = dsm_cdiv(D, nseg);
    nseg = dsm_cdiv(D, dseg);
    const long cells = ((long)((SAB_TX - 1) * p.sw) + 3) * ((long)((SAB_TY - 1) * p.sh) + 3) *
                       ((long)((dseg - 1) * p.sd) + 3);
    if (cells * 4 <= 48 * 1024 && (long)B * nseg <= 65535) {
      dim3 tgrid(dsm_cdiv(W, SAB_TX), dsm_cdiv(H, SAB_TY), B * nseg);
      hipLaunchKernelGGL(soft_argmin_bwd_tile_kernel, tgrid, block, (size_t)cells * 4, s, p, dseg, nseg);
      return dsm_launch_status();
    }
    hipLaunchKernelGGL(soft_argmin_bwd_kernel<true>, grid, block, 0, s, p);
  } else {
    hipLaunchKernelGGL(soft_argmin_bwd_kernel<false>, grid, block, 0, s, p);
  }
  return dsm_launch_status();
}
