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

// ----------------------------------------------------------------------------
// The tile kernel (r03): stride 1 | 2, W % 4 == 0, D <= 4 DB NDH.
// Workgroup = (b, y, 64 x), 4 NDH waves: wave = (channel quarter cw, d half dh).  What the r01 kernel
// above does in four stage -> barrier -> compute -> barrier rounds with <1 wave per SIMD (40 us for
// 36.5 MB that fit the Infinity Cache) happens here without a barrier before the reduction:
//  * only the fR window [x0 - 4 DB NDH S, x0 + 64) goes through LDS (57 KB at C = 128, D <= 48: two
//    workgroups per CU, the whole 384x1280 problem resident at once); every wave stages the rows of
//    ITS OWN channels -- all its 16-byte loads in flight together, no workgroup barrier -- and reads
//    them back wave-locally;
//  * the fL quad of a lane's four columns comes straight from global memory (the four d-group lanes of
//    a column share the address), a four-deep register ring ahead of its use;
//  * lane = (xg 0..15, dg 0..3) owns a 4(x) x DB(d) register block: per channel (4 + DB S)/4
//    ds_read_b128 feed 4 DB FMAs (DB = 12: 48 FMAs, packed by the compiler into 24 v_pk_fma_f32, per
//    4 reads); every lane of every wave is busy (41 planes padded to 48);
//  * the channel quarters' partial sums meet in LDS (the window's space, dead by then) and each wave
//    writes its share of the d planes.
// ----------------------------------------------------------------------------
template <int S, int DB, int NDH>
__global__ __launch_bounds__(256 * NDH, (NDH == 1 ? 2 : 1)) void corr1d_tile_kernel(
    const float* __restrict__ fL, const float* __restrict__ fR, float* __restrict__ out,
    int C, int H, int W, int D) {
  constexpr int DP = 4 * DB * NDH, PADL = DP * S, RW = TX + PADL;
  constexpr int QR = RW / 4;                    // staged quads per channel
  constexpr int WIN = 4 + DB * S;               // floats of the fR window per thread
  static_assert((DB * S) % 4 == 0, "16-byte aligned window reads");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Rs = lds;                        // [C][RW]; column j <-> x = x0 - PADL + j
  const int x0 = blockIdx.x * TX, y = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cw = wave & 3, dh = wave >> 2;      // channel quarter, d half
  const int xg = lane & 15, dg = lane >> 4, d0 = (dh * 4 + dg) * DB;
  const long plane = (long)H * W;
  const float* const baseL = fL + (long)b * C * plane + (long)y * W;
  const float* const baseR = fR + (long)b * C * plane + (long)y * W;
  const int cq = (C + 3) / 4, c_lo = cw * cq, c_hi = min(C, c_lo + cq);

  float acc[DB][4];
#pragma unroll
  for (int i = 0; i < DB; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
  const int x = x0 + 4 * xg;
  const bool live = x < W;
  // branch-free channel walk (the host sends only C % 16 == 0 here: every quarter is whole groups of
  // four channels): a lane past the row's end reads the last quad of the row (its sums are never
  // stored); pointers advance by uniform increments, LDS reads take immediate offsets
  const float* lptr = baseL + (live ? x : W - 4) + (long)c_lo * plane;
  const float* rcur = Rs + 4 * xg + PADL - (d0 + DB) * S + c_lo * RW;      // 16-B aligned
  constexpr int LA = 4;                         // fL quads in flight ahead of their use
  const int nc = c_hi - c_lo;
  f32x4 lq[LA];
  // channels [c_from, c_to) of the quarter, groups of four
  auto compute = [&](int c_from, int c_to) __attribute__((always_inline)) {
    f32x4 wq[2][WIN / 4];                       // the window of the channel in hand and of the next one
#pragma unroll
    for (int k = 0; k < WIN / 4; ++k) wq[0][k] = *reinterpret_cast<const f32x4*>(rcur + 4 * k);
    for (int c4 = c_from; c4 < c_to; c4 += LA) {
      // the next group's fL quads (the last group re-reads itself: nothing past the tensor is touched)
      const float* lnext = lptr + (c4 + LA < nc ? (long)LA * plane : 0l);
#pragma unroll
      for (int u = 0; u < LA; ++u) {
        const f32x4 l = lq[u];
        lq[u] = *reinterpret_cast<const f32x4*>(lnext + (long)u * plane);
#pragma unroll
        for (int k = 0; k < WIN / 4; ++k)       // (the row past the last staged one: read, never used)
          wq[(u + 1) & 1][k] = *reinterpret_cast<const f32x4*>(rcur + (u + 1) * RW + 4 * k);
        float win[WIN];
#pragma unroll
        for (int k = 0; k < WIN / 4; ++k) {
          const f32x4 r = wq[u & 1][k];
          win[4 * k] = r.x; win[4 * k + 1] = r.y; win[4 * k + 2] = r.z; win[4 * k + 3] = r.w;
        }
#pragma unroll
        for (int i = 0; i < DB; ++i) {
          // output x = x0 + 4 xg + j pairs with fR column x - (d0 + i) S = window[j + (DB - i) S]
          acc[i][0] = fmaf(l.x, win[0 + (DB - i) * S], acc[i][0]);
          acc[i][1] = fmaf(l.y, win[1 + (DB - i) * S], acc[i][1]);
          acc[i][2] = fmaf(l.z, win[2 + (DB - i) * S], acc[i][2]);
          acc[i][3] = fmaf(l.w, win[3 + (DB - i) * S], acc[i][3]);
        }
      }
      lptr = lnext;
      rcur += LA * RW;
    }
  };

  // ---- stage the fR rows of this wave's channels (the d halves of a quarter split them) and compute
  // the channels that have arrived while the later loads are still in flight: the whole problem is
  // resident at once, so a load-everything-then-compute kernel would leave the memory system idle
  // while the chip multiplies and the ALUs idle while it loads
  {
    const int n = nc * QR;                      // quads of the quarter
    constexpr int NB = S == 1 ? 16 : 8;         // loads in flight per thread and batch (registers: S = 2 has the wider window)
#pragma unroll
    for (int u = 0; u < LA; ++u) lq[u] = *reinterpret_cast<const f32x4*>(lptr + (long)u * plane);
    int done = 0;                               // channels of the quarter multiplied so far
    for (int i0 = dh * 64 + lane; i0 - (dh * 64 + lane) < n; i0 += 64 * NDH * NB) {
      f32x4 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int i = i0 + 64 * NDH * u;
        const int c = i / QR, k = i - c * QR;
        const int xr = x0 - PADL + 4 * k;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < n && xr >= 0 && xr < W)         // W % 4 == 0 and xr % 4 == 0: a quad is all in or all out
          v[u] = *reinterpret_cast<const f32x4*>(baseR + (c_lo + c) * plane + xr);
      }
#pragma unroll
      for (int g4 = 0; g4 < NB; g4 += 4) {
#pragma unroll
        for (int u = g4; u < g4 + 4; ++u) {
          const int i = i0 + 64 * NDH * u;
          if (i < n) {
            const int c = i / QR, k = i - c * QR;
            *reinterpret_cast<f32x4*>(Rs + (c_lo + c) * RW + 4 * k) = v[u];
          }
        }
        if constexpr (NDH == 1) {
          // wave-private rows: LDS serves one wave's operations in order -- no workgroup barrier
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // quads [0, first quad of the next store group) of the quarter are in LDS now
          const int upto = min(n, (i0 - lane) + 64 * (g4 + 4));
          const int ready = min(nc, (upto / QR) & ~(LA - 1));
          if (ready > done) { compute(done, ready); done = ready; }
        }
      }
    }
    if constexpr (NDH == 2) {
      __syncthreads();                          // the two d halves of a quarter staged it together
      compute(0, nc);
    } else if (done < nc) {
      compute(done, nc);
    }
  }

  // ---- the channel quarters' partial sums meet in LDS: P[cw][d][x]; then every wave finishes its share
  __syncthreads();                              // the window is dead
  constexpr int PW = TX + 4;                    // row pitch (floats)
  float* const P = lds;                         // [4][DP][PW]
#pragma unroll
  for (int i = 0; i < DB; ++i)
    *reinterpret_cast<f32x4*>(P + (cw * DP + d0 + i) * PW + 4 * xg) =
        f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
  __syncthreads();
  if (!live) return;
  // wave (cw, dh) finishes planes (dh 4 + cw) DB + {dg, dg + 4, ...}: the four dg lanes of a column
  // step through the group together
#pragma unroll
  for (int t = 0; t < DB / 4; ++t) {
    const int d = (dh * 4 + cw) * DB + dg + 4 * t;
    if (d >= D) continue;
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) s4 += *reinterpret_cast<const f32x4*>(P + (w * DP + d) * PW + 4 * xg);
    *reinterpret_cast<f32x4*>(out + (((long)b * D + d) * H + y) * W + x) = s4;
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

// The same for k = 3 on rows of W % 4 == 0: a thread owns four outputs of one (plane, row) and reads
// the three source rows as one 16-byte load + the two neighbours each (the source -- a correlation
// volume just written -- is cache-resident; the scalar kernel above spent 39 us on 20 MB).
__global__ __launch_bounds__(256) void box3_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                   long nplanes, int H, int W) {
  const int wq = W >> 2;
  const long n = nplanes * H * wq;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int xq = (int)(i % wq);
  const long row = i / wq;
  const int y = (int)(row % H);
  const float* p = src + row * W + 4 * xq;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    if (y + dy < 0 || y + dy >= H) continue;
    const float* q = p + (long)dy * W;
    const f32x4 v = *reinterpret_cast<const f32x4*>(q);
    const float l = xq > 0 ? q[-1] : 0.f, r = xq + 1 < wq ? q[4] : 0.f;
    a += f32x4{l + v.x + v.y, v.x + v.y + v.z, v.y + v.z + v.w, v.z + v.w + r};
  }
  *reinterpret_cast<f32x4*>(dst + row * W + 4 * xq) = a * (1.f / 9.f);
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
  // the tile kernel: the fR window of all channels in LDS at once; D <= 48 with four waves, D <= 96
  // with eight (two d halves per channel quarter)
  const int NDH_ = D <= 48 ? 1 : (D <= 96 ? 2 : 0);
  const int DP_ = 48 * NDH_;
  const size_t win_lds = (size_t)C * (TX + DP_ * stride) * sizeof(float);
  const size_t red_lds = (size_t)4 * DP_ * (TX + 4) * sizeof(float);
  const size_t tile_lds = win_lds > red_lds ? win_lds : red_lds;
  if (vec && NDH_ && C % 16 == 0 && (stride == 1 || stride == 2) && tile_lds + 4096 <= 150 * 1024) {
    dim3 grid(dsm_cdiv(W, TX), H, B);
#define DSM_CORR_TILE(S_, NDH__)                                                                     \
    do {                                                                                             \
      static thread_local bool configured = false;                                                  \
      if (!configured) {                                                                             \
        if (hipFuncSetAttribute((const void*)corr1d_tile_kernel<S_, 12, NDH__>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) \
          return DSM_ERR_LAUNCH;                                                                     \
        configured = true;                                                                           \
      }                                                                                              \
      hipLaunchKernelGGL((corr1d_tile_kernel<S_, 12, NDH__>), grid, dim3(256 * NDH__), tile_lds + 4096, s, /* + a row of slack */ \
                         (const float*)fL, (const float*)fR, raw, C, H, W, D);                       \
    } while (0)
    if (stride == 1 && NDH_ == 1) DSM_CORR_TILE(1, 1);
    else if (stride == 1) DSM_CORR_TILE(1, 2);
    else if (NDH_ == 1) DSM_CORR_TILE(2, 1);
    else DSM_CORR_TILE(2, 2);
#undef DSM_CORR_TILE
  } else if ((stride == 1 || stride == 2) && ndg * 16 <= 256) {
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
  if (ksize == 3 && vec && dsm_aligned16(out)) {
    const long n = (long)B * D * H * (W / 4);
    hipLaunchKernelGGL(box3_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s, (const float*)raw, (float*)out,
                       (long)B * D, H, W);
  } else if (ksize > 1) {
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
