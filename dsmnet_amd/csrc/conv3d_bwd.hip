// Backward kernels of the k=3 3-D convolution blocks (training through the trunk).
//
// bwd-data needs no kernel of its own: it is a convolution again and runs on
// conv3d_mfma_kernel / deconv3d_mfma_kernel with re-packed weights (host side,
// dsmnet_amd/costvolume.py):
//   conv stride 1  : dX = conv_s1(dY, W^T flipped)
//   conv stride 2  : dX = convT_s2(dY, W)            (cropped to the input size)
//   convT stride 2 : dX = conv_s2(dY, W)
// This file holds bwd-weight (a voxel-reduction GEMM on fp32 MFMA) and the two small
// Cout = 1 kernels (classifier heads).
//
// bwd-weight:  dW[g][c][tap] = sum_v X[v*S + tap - 1][c] * G[v][g]
//   X: the layer input (halo'd, channels c), G: dY (channels g); for a transposed conv the
//   caller swaps the roles (X := dY, G := x) and gets the ConvTranspose3d layout directly.
// GEMM view per tap: M = 32 X-channels, N = 32 G-channels, K = voxels.  v_mfma_f32_32x32x2_f32
// consumes two voxels per instruction: A[i=c][k] = X[voxel_k + tap][c], B[k][j=g] = G[voxel_k][g]
// -- both are one ds_read_b32 per lane with immediate offsets (no VALU: the fp32 MFMA shares
// the vector ALU, see conv3d.hip).  A workgroup owns one (c-tile, g-tile) pair and a set of
// output tiles; wave w owns taps {w, w+4, ...} (<= 7 accumulators); partial sums leave as
// fp32 atomics into a [pair][tap][c][g] workspace (g contiguous: 128-B segments), which a last
// small kernel permutes into torch's (G, C, 3, 3, 3).
#include "common.hpp"
#include <type_traits>

namespace {
constexpr int NT_ = 256;

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

struct WgradParams {
  const float* x; const float* g; float* ws;      // ws: [npair][27][32][32]
  int B, Cx, Cg;                                   // channels of X and G
  int Dx, Hx, Wx;                                  // X volume
  int Dg, Hg, Wg;                                  // G volume (the strided side)
  int ntx, nty, ntiles, npair;
};

// S: stride of X relative to G.  Tile: 1 z x TY rows x 32 columns of G voxels.
// KZ = 3: the 27-tap 3-D layers, wave w owns taps {w, w + 4, ...}.  KZ = 1: the 9-tap 2-D layers
// (maps as D = 1 volumes; DIL = tap spacing), every wave owns all nine taps and a quarter of the
// tile's voxels instead (9 taps do not divide over 4 waves).
template <int S, int TY, int KZ = 3, int DIL = 1>
__global__ __launch_bounds__(NT_, 1) void conv3d_wgrad_kernel(WgradParams p) {
  constexpr int IY = (TY - 1) * S + 2 * DIL + 1, IX = 31 * S + 2 * DIL + 1, IZ = KZ;
  constexpr int NTAP = 9 * KZ, NACC = KZ == 3 ? 7 : 9;
  static_assert(KZ == 3 || KZ == 1, "3-D (27 taps) or 2-D (9 taps)");
  static_assert(KZ == 1 || DIL == 1, "dilation: 2-D layers only");
  constexpr int NXE = IZ * IY * IX * 8;            // X tile: 32 channels = 8 x 16 B per voxel
  constexpr int NGE = TY * 32 * 8;                 // G tile
  constexpr int NPX = (NXE + NT_ - 1) / NT_, NPG = (NGE + NT_ - 1) / NT_;
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  f32x4* xt = lds;                                 // [z][y][x][8]
  f32x4* gt = lds + NXE;                           // [y][x][8]
  const float* xf = reinterpret_cast<const float*>(xt);
  const float* gf = reinterpret_cast<const float*>(gt);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int pair = blockIdx.y;
  const int ct = pair / (p.Cg / 32), gtile = pair % (p.Cg / 32);   // X-channel tile, G-channel tile
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    int id = t;
    const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
    const int ty0 = (id % p.nty) * TY; id /= p.nty;
    const int tz = id % p.Dg; const int b = id / p.Dg;
    __syncthreads();
    // stage X halo tile (zeros outside the volume) and G tile (zeros outside)
    for (int k = 0; k < NPX; ++k) {
      const int e = tid + k * NT_;
      if (e < NXE) {
        const int q = e & 7, v = e >> 3;
        const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
        const int zi = tz * S - KZ / 2 + zz, yi = ty0 * S - DIL + yy, xi = tx0 * S - DIL + xx;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (zi >= 0 && zi < p.Dx && yi >= 0 && yi < p.Hx && xi >= 0 && xi < p.Wx)
          val = *reinterpret_cast<const f32x4*>(
              p.x + ((((long)b * p.Dx + zi) * p.Hx + yi) * p.Wx + xi) * p.Cx + ct * 32 + q * 4);
        xt[e] = val;
      }
    }
    for (int k = 0; k < NPG; ++k) {
      const int e = tid + k * NT_;
      if (e < NGE) {
        const int q = e & 7, v = e >> 3;
        const int xx = v & 31, yy = v >> 5;
        const int yi = ty0 + yy, xi = tx0 + xx;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (yi < p.Hg && xi < p.Wg)
          val = *reinterpret_cast<const f32x4*>(
              p.g + ((((long)b * p.Dg + tz) * p.Hg + yi) * p.Wg + xi) * p.Cg + gtile * 32 + q * 4);
        gt[e] = val;
      }
    }
    __syncthreads();
    // wave w: taps w, w+4, ...; per tap K = TY*32 voxels, two per MFMA (k = h)
    sfor<0, NACC>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      const int tap = KZ == 3 ? wave + 4 * a : a;
      if (tap < NTAP) {
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        const float* xa = xf + ((dz * IY + dy * DIL) * IX + dx * DIL) * 32 + r;   // + voxel*32*S...
        const int k0 = KZ == 3 ? 0 : wave * (TY * 4), k1 = KZ == 3 ? TY * 16 : k0 + TY * 4;
#pragma unroll 4
        for (int kk = k0; kk < k1; ++kk) {
          const int vy = kk >> 4, vx = ((kk & 15) << 1) + h;              // voxel of this lane's k
          const float av = xa[((vy * S) * IX + vx * S) * 32];
          const float bv = gf[(vy * 32 + vx) * 32 + r];
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
        }
      }
    });
  }
  // flush: ws[pair][tap][c][g] += acc  (row = c, column = g on the lanes)
  sfor<0, NACC>([&](auto ac) {
    constexpr int a = decltype(ac)::value;
    const int tap = KZ == 3 ? wave + 4 * a : a;
    if (tap < NTAP) {
      float* dst = p.ws + (((long)pair * NTAP + tap) * 32) * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = (i & 3) + 8 * (i >> 2) + 4 * h;
        atomicAdd(dst + c * 32, acc[a][i]);
      }
    }
  });
}

// ws [pair = ct*(Cg/32)+gt][tap][c][g]  ->  dW[g_abs][c_abs][tap]   (torch (G, C, 3,3,3) / (G, C, 3,3))
__global__ void wgrad_permute_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cx,
                                     int Cg, int ntap) {
  const long n = (long)Cx * Cg * ntap;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int tap = i % ntap;
  const int c = (i / ntap) % Cx;
  const int g = i / ((long)ntap * Cx);
  const int pair = (c / 32) * (Cg / 32) + g / 32;
  dw[i] = ws[(((long)pair * ntap + tap) * 32 + (c & 31)) * 32 + (g & 31)];
}

template <int S, int TY, int KZ, int DIL>
int launch_wgrad(WgradParams p, float* dw, hipStream_t s) {
  constexpr int IY = (TY - 1) * S + 2 * DIL + 1, IX = 31 * S + 2 * DIL + 1;
  constexpr int NTAP = 9 * KZ;
  constexpr size_t lds = (size_t)(KZ * IY * IX * 8 + TY * 32 * 8) * 16;
  static_assert(lds <= 160 * 1024, "LDS");
  const size_t wsbytes = (size_t)p.npair * NTAP * 32 * 32 * sizeof(float);
  if (hipMemsetAsync(p.ws, 0, wsbytes, s) != hipSuccess) return DSM_ERR_LAUNCH;
  p.ntx = dsm_cdiv(p.Wg, 32); p.nty = dsm_cdiv(p.Hg, TY);
  const long nt = (long)p.B * p.Dg * p.nty * p.ntx;
  DSM_REQUIRE(nt < (1L << 30), DSM_ERR_UNSUPPORTED);
  p.ntiles = (int)nt;
  int bx = 256 / p.npair; if (bx < 1) bx = 1;
  if (bx > p.ntiles) bx = p.ntiles;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)conv3d_wgrad_kernel<S, TY, KZ, DIL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return DSM_ERR_LAUNCH;
    configured = true;
  }
  hipLaunchKernelGGL((conv3d_wgrad_kernel<S, TY, KZ, DIL>), dim3(bx, p.npair), dim3(NT_), lds, s, p);
  const long n = (long)p.Cx * p.Cg * NTAP;
  hipLaunchKernelGGL(wgrad_permute_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s,
                     (const float*)p.ws, dw, p.Cx, p.Cg, NTAP);
  return dsm_launch_status();
}

// ---- Cout = 1 (classifier heads): y[v] = sum_tap sum_c x[v+tap-1][c] w[tap][c] ------------
// bwd-data: dx[u][c] = sum_tap g[u - tap + 1] * w[tap][c];  thread = (voxel, 4 channels)
__global__ __launch_bounds__(256) void cout1_bwd_data_kernel(const float* __restrict__ g,
                                                             const float* __restrict__ w,  // [27][C]
                                                             float* __restrict__ dx, int B, int C,
                                                             int D, int H, int W) {
  const long n = (long)B * D * H * W * (C / 4);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int q = i % (C / 4);
  long v = i / (C / 4);
  const int x = v % W; v /= W;
  const int y = v % H; v /= H;
  const int z = v % D; const int b = v / D;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int tap = 0; tap < 27; ++tap) {
    const int zz = z - tap / 9 + 1, yy = y - (tap / 3) % 3 + 1, xx = x - tap % 3 + 1;
    if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    const float gv = g[(((long)b * D + zz) * H + yy) * W + xx];
    a += gv * *reinterpret_cast<const f32x4*>(w + tap * C + q * 4);
  }
  *reinterpret_cast<f32x4*>(dx + i * 4) = a;
}

// bwd-weight: dw[tap][c] = sum_v x[v+tap-1][c] g[v];  block = one (b, z, y) row, lanes over c,
// partial sums per block leave as atomics into dw (27*C floats).
__global__ __launch_bounds__(256) void cout1_bwd_weight_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ g,
                                                               float* __restrict__ dw, int B, int C,
                                                               int D, int H, int W) {
  // thread = (tap-group, channel): 256 threads cover 27 taps x C<=... in passes
  const int row = blockIdx.x;                      // (b*D + z)*H + y
  const int y = row % H; const int z = (row / H) % D; const int b = row / (H * D);
  const float* grow = g + (long)row * W;
  for (int u = threadIdx.x; u < 27 * C; u += 256) {
    const int tap = u / C, c = u % C;
    const int zz = z + tap / 9 - 1, yy = y + (tap / 3) % 3 - 1, dx = tap % 3 - 1;
    if (zz < 0 || zz >= D || yy < 0 || yy >= H) continue;
    const float* xrow = x + ((((long)b * D + zz) * H + yy) * W) * C + c;
    float a = 0.f;
    const int x_lo = dx < 0 ? 1 : 0, x_hi = dx > 0 ? W - 1 : W;
    for (int xx = x_lo; xx < x_hi; ++xx) a = fmaf(xrow[(long)(xx + dx) * C], grow[xx], a);
    atomicAdd(dw + u, a);
  }
}


// The same sums with x read ONCE: a thread owns a channel quad q and a voxel slot, keeps all 27
// taps' partial sums (27 x f32x4) in registers across the tiles of its workgroup, and pairs each x
// value with the 27 neighbouring g values from a small LDS tile (broadcast reads).  NQ = C / 4 quads
// x (256 / NQ) voxels per tile; a few fat workgroups so that the final atomics stay few (27 C per
// workgroup after the in-wave and cross-wave reductions).  The kernel above re-reads x 27 times
// through L2 and issues 27 C atomics per ROW (0.36 ms per head at 48 x 64 x 128).
template <int NQ>
__global__ __launch_bounds__(256) void cout1_bwd_weight_tile_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ g,
                                                                    float* __restrict__ dw, int B,
                                                                    int D, int H, int W) {
  constexpr int VT = 256 / NQ;                     // voxels (along x) per tile
  constexpr int C = 4 * NQ;
  __shared__ float gl[9 * (VT + 2)];
  __shared__ float red[3 * 27 * C];                // waves 1..3's sums
  const int q = threadIdx.x % NQ, vs = threadIdx.x / NQ;
  const int nxc = (W + VT - 1) / VT;
  const long ntile = (long)B * D * H * nxc;
  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](long t) {                      // this thread's x value of tile t (zeros outside)
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t < ntile) {
      const int xu = (int)(t % nxc) * VT + vs;
      if (xu < W) v = *reinterpret_cast<const f32x4*>(x + ((t / nxc) * W + xu) * C + 4 * q);
    }
    return v;
  };
  f32x4 xnext = load_x(blockIdx.x);
  for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int xc = t % nxc; long r = t / nxc;
    const int y = r % H; r /= H;
    const int z = r % D; const int b = r / D;
    const int x0 = xc * VT;
    const f32x4 xv = xnext;
    xnext = load_x(t + gridDim.x);                 // in flight across this tile's work
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * (VT + 2); i += 256) {
      const int xx = x0 - 1 + i % (VT + 2), ry = (i / (VT + 2)) % 3, rz = i / (3 * (VT + 2));
      const int yy = y - 1 + ry, zz = z - 1 + rz;
      gl[i] = (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D)
                  ? g[(((long)b * D + zz) * H + yy) * W + xx] : 0.f;
    }
    __syncthreads();
    // x[u] meets g[u - tap + 1]: rows (2 - dz, 2 - dy), column vs + 2 - dx of the tile
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      acc[tap] += xv * gl[((2 - dz) * 3 + (2 - dy)) * (VT + 2) + vs + 2 - dx];
    }
  }
  // voxel slots of one wave: lanes q + NQ * k
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
#pragma unroll
    for (int off = NQ; off < 64; off <<= 1) {
      acc[tap].x += __shfl_xor(acc[tap].x, off); acc[tap].y += __shfl_xor(acc[tap].y, off);
      acc[tap].z += __shfl_xor(acc[tap].z, off); acc[tap].w += __shfl_xor(acc[tap].w, off);
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (NQ <= 64 && wave > 0 && lane < NQ)
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
      *reinterpret_cast<f32x4*>(red + ((wave - 1) * 27 + tap) * C + 4 * lane) = acc[tap];
  __syncthreads();
  if (wave == 0 && lane < NQ)
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      f32x4 a = acc[tap];
#pragma unroll
      for (int w = 0; w < 3; ++w) a += *reinterpret_cast<const f32x4*>(red + (w * 27 + tap) * C + 4 * lane);
      float* o = dw + tap * C + 4 * lane;
      atomicAdd(o, a.x); atomicAdd(o + 1, a.y); atomicAdd(o + 2, a.z); atomicAdd(o + 3, a.w);
    }
}

// ----------------------------------------------------------------------------
// ConvTranspose3d(C -> 1, k3, s2, p1, op1) backward -- GCNet's head l37 (models/gcnet.py:63,98).
// Output position o = 2 i - 1 + k per axis.  g: (B,Do,Ho,Wo); x: (B,Di,Hi,Wi,C) NDHWC;
// w: torch layout (C,1,27).  VALU kernels: the layer is 2.7 GMAC at GCNet's full size.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deconv_cout1_bwd_data_kernel(
    const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx, int B, int C,
    int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  extern __shared__ float wt[];                       // [27][C]
  for (int u = threadIdx.x; u < 27 * C; u += 256) wt[(u % 27) * C + u / 27] = w[u];
  __syncthreads();
  const long n = (long)B * Di * Hi * Wi * (C / 4);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int q = i % (C / 4);
  long v = i / (C / 4);
  const int x = v % Wi; v /= Wi;
  const int y = v % Hi; v /= Hi;
  const int z = v % Di; const int b = v / Di;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int tap = 0; tap < 27; ++tap) {
    const int oz = 2 * z - 1 + tap / 9, oy = 2 * y - 1 + (tap / 3) % 3, ox = 2 * x - 1 + tap % 3;
    if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
    const float gv = g[(((long)b * Do + oz) * Ho + oy) * Wo + ox];
    a += gv * *reinterpret_cast<const f32x4*>(wt + tap * C + q * 4);
  }
  *reinterpret_cast<f32x4*>(dx + i * 4) = a;
}

// dw[c][tap] = sum_i x[i][c] g[2 i - 1 + tap]; thread = (voxel lane, channel), 27 running sums
// each, voxel lanes folded through LDS, one atomic per (c, tap) per block.
__global__ __launch_bounds__(256) void deconv_cout1_bwd_weight_kernel(
    const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dw, int B, int C,
    int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  __shared__ float red[256 * 27];
  const int nvl = 256 / C;
  const int c = threadIdx.x % C, vl = threadIdx.x / C;
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  const long nvox = (long)B * Di * Hi * Wi;
  if (vl < nvl) {
    for (long v = (long)blockIdx.x * nvl + vl; v < nvox; v += (long)gridDim.x * nvl) {
      long r = v;
      const int xx = r % Wi; r /= Wi;
      const int yy = r % Hi; r /= Hi;
      const int zz = r % Di; const int b = r / Di;
      const float xv = x[v * C + c];
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const int oz = 2 * zz - 1 + t / 9, oy = 2 * yy - 1 + (t / 3) % 3, ox = 2 * xx - 1 + t % 3;
        if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
        acc[t] = fmaf(xv, g[(((long)b * Do + oz) * Ho + oy) * Wo + ox], acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) red[threadIdx.x * 27 + t] = acc[t];
  __syncthreads();
  for (int u = threadIdx.x; u < 27 * C; u += 256) {
    const int cc = u / 27, t = u % 27;
    float s = 0.f;
    for (int l = 0; l < nvl; ++l) s += red[(l * C + cc) * 27 + t];
    atomicAdd(dw + u, s);
  }
}

}  // namespace

// x: (B,Dx,Hx,Wx,Cx) NDHWC; g: (B,Dg,Hg,Wg,Cg); stride: X positions per G position (1 or 2);
// ws: workspace of (Cx/32)*(Cg/32)*27*32*32 floats (zeroed here); dw: (Cg, Cx, 27), overwritten.
extern "C" int dsm_conv3d_wgrad(const void* x, const void* g, void* ws, void* dw, int B, int Cx,
                                int Cg, int Dx, int Hx, int Wx, int Dg, int Hg, int Wg, int stride,
                                dsm_stream_t stream) {
  DSM_REQUIRE(x && g && ws && dw, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && Cx > 0 && Cg > 0 && Dx > 0 && Hx > 0 && Wx > 0 && Dg > 0 && Hg > 0 && Wg > 0,
              DSM_ERR_ARG);
  DSM_REQUIRE(stride == 1 || stride == 2, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(Cx % 32 == 0 && Cg % 32 == 0, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(dsm_aligned16(x) && dsm_aligned16(g), DSM_ERR_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  WgradParams p;
  p.x = (const float*)x; p.g = (const float*)g; p.ws = (float*)ws;
  p.B = B; p.Cx = Cx; p.Cg = Cg; p.Dx = Dx; p.Hx = Hx; p.Wx = Wx; p.Dg = Dg; p.Hg = Hg; p.Wg = Wg;
  p.npair = (Cx / 32) * (Cg / 32);
  if (stride == 1) return launch_wgrad<1, 4, 3, 1>(p, (float*)dw, s);
  return launch_wgrad<2, 2, 3, 1>(p, (float*)dw, s);
}

// The same for the 2-D towers' 3x3 layers (padding = dilation): x: (B,Hx,Wx,Cx) NHWC,
// g: (B,Hg,Wg,Cg); ws: (Cx/32)*(Cg/32)*9*32*32 floats; dw: (Cg, Cx, 3, 3), overwritten.
extern "C" int dsm_conv2d_wgrad(const void* x, const void* g, void* ws, void* dw, int B, int Cx,
                                int Cg, int Hx, int Wx, int Hg, int Wg, int stride, int dilation,
                                dsm_stream_t stream) {
  DSM_REQUIRE(x && g && ws && dw, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && Cx > 0 && Cg > 0 && Hx > 0 && Wx > 0 && Hg > 0 && Wg > 0, DSM_ERR_ARG);
  DSM_REQUIRE((stride == 1 && (dilation == 1 || dilation == 2)) || (stride == 2 && dilation == 1),
              DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(Cx % 32 == 0 && Cg % 32 == 0, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(dsm_aligned16(x) && dsm_aligned16(g), DSM_ERR_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  WgradParams p;
  p.x = (const float*)x; p.g = (const float*)g; p.ws = (float*)ws;
  p.B = B; p.Cx = Cx; p.Cg = Cg; p.Dx = 1; p.Hx = Hx; p.Wx = Wx; p.Dg = 1; p.Hg = Hg; p.Wg = Wg;
  p.npair = (Cx / 32) * (Cg / 32);
  if (stride == 2) return launch_wgrad<2, 4, 1, 1>(p, (float*)dw, s);
  if (dilation == 2) return launch_wgrad<1, 8, 1, 2>(p, (float*)dw, s);
  return launch_wgrad<1, 8, 1, 1>(p, (float*)dw, s);
}

// Cout = 1 convolution (stride 1): g: (B,D,H,W); x: (B,D,H,W,C); w_packed: [27][C] (as packed
// by dsm_conv3d_pack_weights); dx: (B,D,H,W,C) or NULL; dw: (1, C, 27) torch layout or NULL.
extern "C" int dsm_conv3d_cout1_bwd(const void* x, const void* g, const void* w_packed, void* dx,
                                    void* dw_tapmajor, int B, int C, int D, int H, int W,
                                    dsm_stream_t stream) {
  DSM_REQUIRE(g && (dx || dw_tapmajor), DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && D > 0 && H > 0 && W > 0, DSM_ERR_ARG);
  DSM_REQUIRE((long)B * D * H <= 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  if (dx) {
    DSM_REQUIRE(w_packed, DSM_ERR_ARG);
    const long n = (long)B * D * H * W * (C / 4);
    hipLaunchKernelGGL(cout1_bwd_data_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s,
                       (const float*)g, (const float*)w_packed, (float*)dx, B, C, D, H, W);
  }
  if (dw_tapmajor) {
    DSM_REQUIRE(x, DSM_ERR_ARG);
    if (hipMemsetAsync(dw_tapmajor, 0, (size_t)27 * C * sizeof(float), s) != hipSuccess)
      return DSM_ERR_LAUNCH;
    const long ntile32 = (long)B * D * H * dsm_cdiv(W, 32);
    const int blocks = (int)(ntile32 < 768 ? ntile32 : 768);   // three resident per CU
    if (C == 32 && dsm_aligned16(x))
      hipLaunchKernelGGL(cout1_bwd_weight_tile_kernel<8>, dim3(blocks), dim3(256), 0, s, (const float*)x,
                         (const float*)g, (float*)dw_tapmajor, B, D, H, W);
    else if (C == 64 && dsm_aligned16(x))
      hipLaunchKernelGGL(cout1_bwd_weight_tile_kernel<16>, dim3(blocks), dim3(256), 0, s, (const float*)x,
                         (const float*)g, (float*)dw_tapmajor, B, D, H, W);
    else
      hipLaunchKernelGGL(cout1_bwd_weight_kernel, dim3(B * D * H), dim3(256), 0, s, (const float*)x,
                         (const float*)g, (float*)dw_tapmajor, B, C, D, H, W);
  }
  return dsm_launch_status();
}

// ConvTranspose3d(C -> 1, k3, s2, p1, op1) backward: g (B,Do,Ho,Wo); x (B,Di,Hi,Wi,C) NDHWC;
// w: torch layout (C,1,3,3,3); dx (B,Di,Hi,Wi,C) or NULL; dw (C,1,3,3,3) or NULL (overwritten).
extern "C" int dsm_deconv3d_cout1_bwd(const void* x, const void* g, const void* w, void* dx,
                                      void* dw, int B, int C, int Di, int Hi, int Wi, int Do,
                                      int Ho, int Wo, dsm_stream_t stream) {
  DSM_REQUIRE(g && (dx || dw), DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, DSM_ERR_ARG);
  DSM_REQUIRE(C % 4 == 0 && C <= 256, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(Do <= 2 * Di && Ho <= 2 * Hi && Wo <= 2 * Wi, DSM_ERR_ARG);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  if (dx) {
    DSM_REQUIRE(w, DSM_ERR_ARG);
    const long n = (long)B * Di * Hi * Wi * (C / 4);
    DSM_REQUIRE(n / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(deconv_cout1_bwd_data_kernel, dim3(dsm_cdiv(n, 256)), dim3(256),
                       (size_t)27 * C * sizeof(float), s, (const float*)g, (const float*)w,
                       (float*)dx, B, C, Di, Hi, Wi, Do, Ho, Wo);
  }
  if (dw) {
    DSM_REQUIRE(x, DSM_ERR_ARG);
    if (hipMemsetAsync(dw, 0, (size_t)27 * C * sizeof(float), s) != hipSuccess) return DSM_ERR_LAUNCH;
    const long nvox = (long)B * Di * Hi * Wi;
    const int nvl = 256 / C;
    long blocks = (nvox + nvl - 1) / nvl;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(deconv_cout1_bwd_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       (const float*)x, (const float*)g, (float*)dw, B, C, Di, Hi, Wi, Do, Ho, Wo);
  }
  return dsm_launch_status();
}
