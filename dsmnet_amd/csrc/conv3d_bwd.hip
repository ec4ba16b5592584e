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
  const float* x_amax; const float* g_amax;        // fp16 modes: absolute maxima of X and G (device scalars)
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
        const int k0 = KZ == 3 ? 0 : wave * (TY * 4);
#pragma unroll 4
        for (int ki = 0; ki < (KZ == 3 ? TY * 16 : TY * 4); ++ki) {
          const int kk = k0 + ki;
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

// ---- the same reduction on the bf16 pipe (bf16x3: fp32 operands split into three bf16 terms, six
// cross products per MAC, fp32 accumulation -- DESIGN.md 3.2a) --------------------------------
// v_mfma_f32_32x32x16_bf16 wants, per lane, EIGHT CONSECUTIVE k (= voxels) of one channel, while
// NDHWC memory (and any sane staging pass) has a voxel's channels consecutive.  gfx950's transposed
// LDS read closes the gap: the tiles are staged as [plane][voxel][32 channels] bf16 (64-byte rows,
// one ds_write_b64 per plane per staged f32x4) and ds_read_b64_tr_b16 hands lane (channel) four
// voxels per read -- and since every lane supplies its own row address, the tap shift (any dx, any
// stride) is just another row offset: no shifted copies, no alignment cases.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));

// precision modes as in conv_split.hpp: PM = 3 bf16x3 (six MFMAs per product), PM = 2 f16x2 (the
// power-of-two-scaled operand as two fp16 terms, three MFMAs), PM = 1 f16 (one MFMA)
template <int PM> struct WPrec;
template <> struct WPrec<3> { static constexpr int NP = 3; typedef wbf16x8 frag; };
template <> struct WPrec<2> { static constexpr int NP = 2; typedef wf16x8 frag; };
template <> struct WPrec<1> { static constexpr int NP = 1; typedef wf16x8 frag; };

__device__ __forceinline__ int w_amax_exponent(float amax) {      // conv_common.hpp: dsm_amax_exponent
  const int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
  const int e = 12 - ex;
  return e < -60 ? -60 : (e > 60 ? 60 : e);
}
__device__ __forceinline__ float w_pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

// four fp32 values -> their NP planes (two dwords per plane); `sx`: the tensor's power-of-two scale (PM < 3)
template <int PM>
__device__ __forceinline__ void wsplit4(const f32x4 v, float sx, wu32x2 (&pl)[WPrec<PM>::NP]) {
  if constexpr (PM == 3) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      unsigned u[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f2 t = {r[2 * i], r[2 * i + 1]};
        u[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, b2));
        if (k < 2) {
          r[2 * i] -= __builtin_bit_cast(float, u[i] << 16);
          r[2 * i + 1] -= __builtin_bit_cast(float, u[i] & 0xffff0000u);
        }
      }
      pl[k] = wu32x2{u[0], u[1]};
    }
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float r[4] = {v.x, v.y, v.z, v.w};
    unsigned hi[2], lo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi[i]) : "v"(r[2 * i]), "s"(sx));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi[i]) : "v"(r[2 * i + 1]), "s"(sx));
      if constexpr (PM == 2) {
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(r[2 * i]), "s"(sx), "v"(hi[i]));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(r[2 * i + 1]), "s"(sx), "v"(hi[i]));
        const f2 t = {r0, r1};
        lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, h2));
      }
    }
    pl[0] = wu32x2{hi[0], hi[1]};
    if constexpr (PM == 2) pl[1] = wu32x2{lo[0], lo[1]};
  }
}

template <int PM>
__device__ __forceinline__ void wmma(f32x16& c, const typename WPrec<PM>::frag (&x)[WPrec<PM>::NP],
                                     const typename WPrec<PM>::frag (&g)[WPrec<PM>::NP]) {
  if constexpr (PM == 3) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], g[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], g[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], g[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], g[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], g[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], g[0], c, 0, 0, 0);
  } else if constexpr (PM == 2) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], g[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], g[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], g[0], c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], g[0], c, 0, 0, 0);
  }
}

// eight voxels (rows row0 .. row0 + 7*RS, RS rows apart) of this lane's channel: two transposed reads
template <int RS, typename FR>
__device__ __forceinline__ FR tr_frag(const unsigned char* lane_base, int byte_off) {
  typedef __attribute__((address_space(3))) ws16x4 lds_s16x4;
  const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lane_base + byte_off));
  const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lane_base + byte_off + 4 * RS * 64));
  const ws16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(FR, v);
}

template <int S, int TY, int KZ, int DIL, int NP = 3>
struct WgradGeo {
  static constexpr int IY = (TY - 1) * S + 2 * DIL + 1, IX = 31 * S + 2 * DIL + 1, IZ = KZ;
  static constexpr int NXV = IZ * IY * IX, NGV = TY * 32;          // voxels of the X halo tile / G tile
  static constexpr int XPL = NXV * 64, GPL = NGV * 64;             // plane strides, bytes
  static constexpr int NTAP = 9 * KZ, NACC = KZ == 3 ? 7 : 9;
  static constexpr size_t TILES = NP * (size_t)(XPL + GPL), FOLD = KZ == 1 ? 4 * 4 * 16 * 64 * 4 : 0;
  static constexpr size_t LDS = TILES > FOLD ? TILES : FOLD;     // KZ = 1 folds the four waves' sums through LDS
};

template <int PM, int S, int TY, int KZ, int DIL>
__global__ __launch_bounds__(NT_, 1) void wgrad_split_kernel(WgradParams p) {
  constexpr int NP = WPrec<PM>::NP;
  using frag = typename WPrec<PM>::frag;
  using G = WgradGeo<S, TY, KZ, DIL, NP>;
  constexpr int IY = G::IY, IX = G::IX, NGV = G::NGV, XPL = G::XPL, GPL = G::GPL;
  constexpr int NTAP = G::NTAP, NACC = G::NACC;
  extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
  unsigned char* const xt = wl;                    // [plane][voxel (z, y, x)][32 ch] 16-bit
  unsigned char* const gt = wl + NP * XPL;         // [plane][voxel (y, x)][32 ch] 16-bit
  float sxs = 1.f, sgs = 1.f, so = 1.f;            // operand scales and the output factor (PM < 3)
  if constexpr (PM != 3) {
    const int ex = w_amax_exponent(*p.x_amax), eg = w_amax_exponent(*p.g_amax);
    sxs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w_pow2f(ex))));
    sgs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w_pow2f(eg))));
    so = w_pow2f(-(ex + eg));
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int pair = blockIdx.y;
  const int ct = pair / (p.Cg / 32), gtile = pair % (p.Cg / 32);
  // transposed-read lane roles: 16-lane group = (channel block cb, k half kg); lane 4 q + pp of the
  // group addresses row q, channels 4 pp .. 4 pp + 3 of the 4-row x 16-channel block
  const int grp = lane >> 4, cb = grp & 1, kg = grp >> 1, q4 = (lane & 15) >> 2, pp = lane & 3;
  const unsigned char* const xlane = xt + ((8 * kg + q4) * S) * 64 + cb * 32 + pp * 8;
  const unsigned char* const glane = gt + (8 * kg + q4) * 64 + cb * 32 + pp * 8;
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  // Work: the (column, z) units z-fastest, one contiguous range per workgroup, so that a step along z
  // re-stages only the X planes that are new (KZ = 3: one of three at stride 1, two at stride 2).  X
  // plane zi lives in slot zi mod 3 of the tile.  The planes a z-step brings in (the last NSL of the
  // tile) and the G tile are PREFETCHED: their global loads are issued before the previous tile's MFMA
  // loop and split / written to LDS after it; the extra planes of a range's first tile are staged on
  // the spot.
  constexpr int NSL = KZ == 3 ? S : 1;
  constexpr int PE = IY * IX * 8;                  // f32x4 elements of one X plane
  constexpr int NXR = (PE + NT_ - 1) / NT_, NGR = (NGV * 8 + NT_ - 1) / NT_;
  struct Tile { int b, ty0, tx0, tz, zlo, col; };
  auto tile_at = [&](long u) {
    Tile T;
    T.col = (int)(u / p.Dg); T.tz = (int)(u % p.Dg);
    int id = T.col;
    T.tx0 = (id % p.ntx) * 32; id /= p.ntx;
    T.ty0 = (id % p.nty) * TY; T.b = id / p.nty;
    T.zlo = T.tz * S - KZ / 2;
    return T;
  };
  auto load_plane = [&](const Tile& T, int zi, f32x4 (&dst)[NXR]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
      const int e = tid + i * NT_;
      const int q = e & 7, v = e >> 3;
      const int xx = v % IX, yy = v / IX;
      const int yi = T.ty0 * S - DIL + yy, xi = T.tx0 * S - DIL + xx;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (e < PE && zi >= 0 && zi < p.Dx && yi >= 0 && yi < p.Hx && xi >= 0 && xi < p.Wx)
        val = *reinterpret_cast<const f32x4*>(
            p.x + ((((long)T.b * p.Dx + zi) * p.Hx + yi) * p.Wx + xi) * p.Cx + ct * 32 + q * 4);
      dst[i] = val;
    }
  };
  auto store_plane = [&](int zi, const f32x4 (&src)[NXR]) __attribute__((always_inline)) {
    const int slot = KZ == 3 ? (zi + 3) % 3 : 0;
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
      const int e = tid + i * NT_;
      if (e < PE) {
        wu32x2 pl[NP];
        wsplit4<PM>(src[i], sxs, pl);
#pragma unroll
        for (int k = 0; k < NP; ++k)
          *reinterpret_cast<wu32x2*>(xt + k * XPL + (slot * (IY * IX) + (e >> 3)) * 64 + (e & 7) * 8) = pl[k];
      }
    }
  };
  auto load_g = [&](const Tile& T, f32x4 (&dst)[NGR]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NGR; ++i) {
      const int e = tid + i * NT_;
      const int q = e & 7, v = e >> 3;
      const int yi = T.ty0 + (v >> 5), xi = T.tx0 + (v & 31);
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (e < NGV * 8 && yi < p.Hg && xi < p.Wg)
        val = *reinterpret_cast<const f32x4*>(
            p.g + ((((long)T.b * p.Dg + T.tz) * p.Hg + yi) * p.Wg + xi) * p.Cg + gtile * 32 + q * 4);
      dst[i] = val;
    }
  };
  auto store_g = [&](const f32x4 (&src)[NGR]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NGR; ++i) {
      const int e = tid + i * NT_;
      if (e < NGV * 8) {
        wu32x2 pl[NP];
        wsplit4<PM>(src[i], sgs, pl);
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<wu32x2*>(gt + k * GPL + (e >> 3) * 64 + (e & 7) * 8) = pl[k];
      }
    }
  };
  const long nunits = (long)p.ntiles;              // = columns x Dg
  const long u0 = nunits * blockIdx.x / gridDim.x, u1 = nunits * (blockIdx.x + 1) / gridDim.x;
  if (u0 >= u1) return;
  f32x4 xr[NSL][NXR], gr[NGR];
  Tile cur = tile_at(u0);
#pragma unroll
  for (int j = 0; j < NSL; ++j) load_plane(cur, cur.zlo + KZ - NSL + j, xr[j]);
  load_g(cur, gr);
  int prev_col = -1, prev_hi = -0x40000000;        // column and highest X plane staged last
  for (long u = u0; u < u1; ++u) {
    const int zlo = cur.zlo;
    const int znew = (cur.col == prev_col && prev_hi >= zlo) ? prev_hi + 1 : zlo;   // first plane to stage
    prev_col = cur.col; prev_hi = zlo + KZ - 1;
    // byte offset of tap a's first voxel in the X tile (wave-uniform)
    int xtap[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int tap_ = KZ == 3 ? wave + 4 * a : a;
      const int tap = tap_ < NTAP ? tap_ : 0;      // a slot past the last tap repeats tap 0 (never flushed)
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int slot = KZ == 3 ? (zlo + dz + 3) % 3 : 0;
      xtap[a] = __builtin_amdgcn_readfirstlane(((slot * IY + dy * DIL) * IX + dx * DIL) * 64);
    }
    __syncthreads();
    for (int zi = znew; zi < zlo + KZ - NSL; ++zi) {      // a range's first tile: the older planes too
      f32x4 tmp[NXR];
      load_plane(cur, zi, tmp);
      store_plane(zi, tmp);
    }
#pragma unroll
    for (int j = 0; j < NSL; ++j) store_plane(zlo + KZ - NSL + j, xr[j]);
    store_g(gr);
    __syncthreads();
    Tile nxt = cur;
    if (u + 1 < u1) {
      nxt = tile_at(u + 1);
#pragma unroll
      for (int j = 0; j < NSL; ++j) load_plane(nxt, nxt.zlo + KZ - NSL + j, xr[j]);
      load_g(nxt, gr);
    }
    // k-blocks of 16 voxels along x: (row vy, half xb).  KZ = 3: every wave walks all of them for its
    // own taps; KZ = 1: a wave takes a quarter of them for all nine taps.  The fragments of step
    // (kb, a + 1) -- or (kb + 1, 0) with that block's G fragments -- are read while the six MFMAs of
    // step (kb, a) run.  A tap slot past the last tap (wave 3's seventh at KZ = 3) repeats tap 0 into
    // an accumulator nobody flushes: no branch in the loop.
    constexpr int NKB = TY * 2;
    const int kb0 = KZ == 3 ? 0 : wave * (NKB / 4), kb1 = KZ == 3 ? NKB : kb0 + NKB / 4;
    auto xoff_of = [&](int kb) { return (((kb >> 1) * S) * IX + 16 * (kb & 1) * S) * 64; };
    frag xf[2][NP], gf[2][NP];
    auto read_x = [&](frag (&d)[NP], int o) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NP; ++k) d[k] = tr_frag<S, frag>(xlane, o + k * XPL);
    };
    auto read_g = [&](frag (&d)[NP], int kb) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NP; ++k) d[k] = tr_frag<1, frag>(glane, kb * 1024 + k * GPL);
    };
    read_g(gf[0], kb0);
    read_x(xf[0], xoff_of(kb0) + xtap[0]);
    static_assert(NKB % 2 == 0 && (KZ == 3 || (NKB / 4) % 2 == 0), "k-blocks are walked in pairs");
    for (int kb2 = kb0; kb2 < kb1; kb2 += 2) {
      sfor<0, 2>([&](auto kc) {
        constexpr int par = decltype(kc)::value;    // G fragment buffer of this k-block
        const int kb = kb2 + par;
        const int kbn = kb + 1 < kb1 ? kb + 1 : kb; // the last block re-reads itself (discarded)
        const int xoff = xoff_of(kb), xoffn = xoff_of(kbn);
        sfor<0, NACC>([&](auto ac) {
          constexpr int a = decltype(ac)::value;
          constexpr int xp = (a + NACC * par) & 1;  // X fragment buffer of step (kb, a): NACC is odd
          static_assert(NACC % 2 == 1, "");
          if constexpr (a + 1 < NACC) {
            read_x(xf[xp ^ 1], xoff + xtap[a + 1]);
          } else {
            read_g(gf[par ^ 1], kbn);
            read_x(xf[xp ^ 1], xoffn + xtap[0]);
          }
          __builtin_amdgcn_sched_barrier(0);         // the next step's reads stay ahead of these MFMAs
          wmma<PM>(acc[a], xf[xp], gf[par]);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    }
    cur = nxt;
  }
  if constexpr (KZ == 1) {
    // every wave holds partial sums of all nine taps: fold the four waves through LDS (the tiles are
    // dead now), four taps per round, so that each (tap, c, g) leaves as ONE atomic per workgroup
    float* const fold = reinterpret_cast<float*>(wl);          // [wave][4 taps][16][64]
    static_assert(4 * 4 * 16 * 64 * 4 <= G::LDS, "fold buffer");
    sfor<0, 3>([&](auto rc) {
      constexpr int rnd = decltype(rc)::value;
      __syncthreads();
      sfor<0, 4>([&](auto jc) {
        constexpr int j = decltype(jc)::value, a = 4 * rnd + j;
        if constexpr (a < NTAP) {
#pragma unroll
          for (int i = 0; i < 16; ++i) fold[((wave * 4 + j) * 16 + i) * 64 + lane] = acc[a][i];
        }
      });
      __syncthreads();
      const int tap = 4 * rnd + wave;
      if (tap < NTAP) {
        float* dst = p.ws + (((long)pair * NTAP + tap) * 32) * 32 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) v += fold[((w * 4 + wave) * 16 + i) * 64 + lane];
          const int c = (i & 3) + 8 * (i >> 2) + 4 * h;
          atomicAdd(dst + c * 32, v * so);
        }
      }
    });
    return;
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
        atomicAdd(dst + c * 32, acc[a][i] * so);
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

template <int PM, int S, int TY, int KZ, int DIL>
int launch_wgrad_split(WgradParams p, float* dw, hipStream_t s) {
  using G = WgradGeo<S, TY, KZ, DIL, WPrec<PM>::NP>;
  static_assert(G::LDS <= 160 * 1024, "LDS");
  static_assert(KZ == 3 || (TY * 2) % 4 == 0, "KZ = 1 splits the k-blocks over the four waves");
  const size_t wsbytes = (size_t)p.npair * G::NTAP * 32 * 32 * sizeof(float);
  if (hipMemsetAsync(p.ws, 0, wsbytes, s) != hipSuccess) return DSM_ERR_LAUNCH;
  p.ntx = dsm_cdiv(p.Wg, 32); p.nty = dsm_cdiv(p.Hg, TY);
  const long nt = (long)p.B * p.Dg * p.nty * p.ntx;
  DSM_REQUIRE(nt < (1L << 30), DSM_ERR_UNSUPPORTED);
  p.ntiles = (int)nt;
  int bx = 256 / p.npair; if (bx < 1) bx = 1;
  if (bx > p.ntiles) bx = p.ntiles;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)wgrad_split_kernel<PM, S, TY, KZ, DIL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS) != hipSuccess)
      return DSM_ERR_LAUNCH;
    configured = true;
  }
  hipLaunchKernelGGL((wgrad_split_kernel<PM, S, TY, KZ, DIL>), dim3(bx, p.npair), dim3(NT_), G::LDS, s, p);
  const long n = (long)p.Cx * p.Cg * G::NTAP;
  hipLaunchKernelGGL(wgrad_permute_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s,
                     (const float*)p.ws, dw, p.Cx, p.Cg, G::NTAP);
  return dsm_launch_status();
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

// The same for C = 32 with a thread per VOXEL: the 27 neighbouring g values are loaded once for all
// 32 channels (the kernel above loads them once per channel quad), the tap's 32 weights come as scalar
// loads (SGPR pairs) and the 864 MACs issue as 432 v_pk_fma_f32; the voxel's 128 bytes leave as eight
// 16-byte stores.  88 -> 52 us at 48 x 64 x 128.
__global__ __launch_bounds__(256) void cout1_bwd_data32_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ w,  // [27][32]
                                                               float* __restrict__ dx, int B, int D,
                                                               int H, int W) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef const float __attribute__((address_space(4))) cfloat;
  typedef f32x4 __attribute__((address_space(4))) cquad;
  const long n = (long)B * D * H * W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long v = i;
  const int x = v % W; v /= W;
  const int y = v % H; v /= H;
  const int z = v % D; const int b = v / D;
  float gn[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const int zz = z - tap / 9 + 1, yy = y - (tap / 3) % 3 + 1, xx = x - tap % 3 + 1;
    gn[tap] = (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
                  ? g[(((long)b * D + zz) * H + yy) * W + xx] : 0.f;
  }
  cfloat* wc = (cfloat*)w;
  asm volatile("" : "+s"(wc));
  f2 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = f2{0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    f32x4 w4[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w4[q] = *(const volatile cquad*)(wc + tap * 32 + q * 4);
    const f2 g2 = {gn[tap], gn[tap]};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc[2 * q] = __builtin_elementwise_fma(g2, f2{w4[q].x, w4[q].y}, acc[2 * q]);
      acc[2 * q + 1] = __builtin_elementwise_fma(g2, f2{w4[q].z, w4[q].w}, acc[2 * q + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4* dst = reinterpret_cast<f32x4*>(dx + i * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) dst[q] = f32x4{acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y};
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
  // a workgroup walks a CONTIGUOUS range of tiles (x-chunk fastest, then y, z, b): the tile's
  // coordinates advance by increments -- the strided walk spent more on 64-bit divisions than on sums
  struct Tile { int xc, y, z, b; long row; };       // row = (b*D + z)*H + y
  auto tile_at = [&](long t) {
    Tile T; T.xc = (int)(t % nxc); T.row = t / nxc;
    long r = T.row; T.y = (int)(r % H); r /= H; T.z = (int)(r % D); T.b = (int)(r / D);
    return T;
  };
  auto next_tile = [&](Tile T) {
    if (++T.xc == nxc) { T.xc = 0; ++T.row; if (++T.y == H) { T.y = 0; if (++T.z == D) { T.z = 0; ++T.b; } } }
    return T;
  };
  auto load_x = [&](const Tile& T, bool live) {     // this thread's x value of the tile (zeros outside)
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int xu = T.xc * VT + vs;
    if (live && xu < W) v = *reinterpret_cast<const f32x4*>(x + (T.row * W + xu) * C + 4 * q);
    return v;
  };
  constexpr int NGL = (9 * (VT + 2) + 255) / 256;  // g-tile elements per thread
  int gdx[NGL], gry[NGL], grz[NGL];                 // this thread's g-tile elements: column, row, plane offsets
#pragma unroll
  for (int k = 0; k < NGL; ++k) {
    const int i = threadIdx.x + 256 * k;
    gdx[k] = i % (VT + 2) - 1; gry[k] = (i / (VT + 2)) % 3 - 1; grz[k] = i / (3 * (VT + 2)) - 1;
  }
  auto load_g = [&](const Tile& T, bool live, float (&dst)[NGL]) {
#pragma unroll
    for (int k = 0; k < NGL; ++k) {
      const int xx = T.xc * VT + gdx[k], yy = T.y + gry[k], zz = T.z + grz[k];
      float val = 0.f;
      if (live && threadIdx.x + 256 * k < 9 * (VT + 2) && xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D)
        val = g[(((long)T.b * D + zz) * H + yy) * W + xx];
      dst[k] = val;
    }
  };
  const long t0 = ntile * blockIdx.x / gridDim.x, t1 = ntile * (blockIdx.x + 1) / gridDim.x;
  Tile nxt = tile_at(t0);
  f32x4 xnext = load_x(nxt, t0 < t1);
  float gnext[NGL];
  load_g(nxt, t0 < t1, gnext);
  for (long t = t0; t < t1; ++t) {
    const f32x4 xv = xnext;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NGL; ++k)
      if (threadIdx.x + 256 * k < 9 * (VT + 2)) gl[threadIdx.x + 256 * k] = gnext[k];
    __syncthreads();
    nxt = next_tile(nxt);
    xnext = load_x(nxt, t + 1 < t1);               // both prefetches are in flight across this tile's work
    load_g(nxt, t + 1 < t1, gnext);
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
                                int flags, int precision, const float* x_amax, const float* g_amax,
                                dsm_stream_t stream) {
  DSM_REQUIRE(x && g && ws && dw, DSM_ERR_ARG);
  DSM_REQUIRE(precision == DSM_PREC_F32 || ((precision == DSM_PREC_F16 || precision == DSM_PREC_F16X2) &&
                                            ((flags & DSM_CONV_FP32_MFMA) || (x_amax && g_amax))), DSM_ERR_ARG);
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
  p.x_amax = x_amax; p.g_amax = g_amax;
  if (!(flags & DSM_CONV_FP32_MFMA)) {             // fp32 operands on the 16-bit matrix pipe
#define DSM_WG3(PM_) \
    if (stride == 1) return launch_wgrad_split<PM_, 1, 4, 3, 1>(p, (float*)dw, s); \
    return launch_wgrad_split<PM_, 2, 1, 3, 1>(p, (float*)dw, s)
    if (precision == DSM_PREC_F16X2) { DSM_WG3(2); }
    if (precision == DSM_PREC_F16) { DSM_WG3(1); }
    DSM_WG3(3);
#undef DSM_WG3
  }
  if (stride == 1) return launch_wgrad<1, 4, 3, 1>(p, (float*)dw, s);
  return launch_wgrad<2, 2, 3, 1>(p, (float*)dw, s);
}

// The same for the 2-D towers' 3x3 layers (padding = dilation): x: (B,Hx,Wx,Cx) NHWC,
// g: (B,Hg,Wg,Cg); ws: (Cx/32)*(Cg/32)*9*32*32 floats; dw: (Cg, Cx, 3, 3), overwritten.
extern "C" int dsm_conv2d_wgrad(const void* x, const void* g, void* ws, void* dw, int B, int Cx,
                                int Cg, int Hx, int Wx, int Hg, int Wg, int stride, int dilation,
                                int flags, int precision, const float* x_amax, const float* g_amax,
                                dsm_stream_t stream) {
  DSM_REQUIRE(x && g && ws && dw, DSM_ERR_ARG);
  DSM_REQUIRE(precision == DSM_PREC_F32 || ((precision == DSM_PREC_F16 || precision == DSM_PREC_F16X2) &&
                                            ((flags & DSM_CONV_FP32_MFMA) || (x_amax && g_amax))), DSM_ERR_ARG);
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
  p.x_amax = x_amax; p.g_amax = g_amax;
  if (!(flags & DSM_CONV_FP32_MFMA)) {
#define DSM_WG2(PM_) \
    if (stride == 2) return launch_wgrad_split<PM_, 2, 4, 1, 1>(p, (float*)dw, s); \
    if (dilation == 2) return launch_wgrad_split<PM_, 1, 8, 1, 2>(p, (float*)dw, s); \
    return launch_wgrad_split<PM_, 1, 8, 1, 1>(p, (float*)dw, s)
    if (precision == DSM_PREC_F16X2) { DSM_WG2(2); }
    if (precision == DSM_PREC_F16) { DSM_WG2(1); }
    DSM_WG2(3);
#undef DSM_WG2
  }
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
    if (C == 32 && dsm_aligned16(w_packed) && dsm_aligned16(dx))
      hipLaunchKernelGGL(cout1_bwd_data32_kernel, dim3(dsm_cdiv(n / 8, 256)), dim3(256), 0, s,
                         (const float*)g, (const float*)w_packed, (float*)dx, B, D, H, W);
    else
      hipLaunchKernelGGL(cout1_bwd_data_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s,
                         (const float*)g, (const float*)w_packed, (float*)dx, B, C, D, H, W);
  }
  if (dw_tapmajor) {
    DSM_REQUIRE(x, DSM_ERR_ARG);
    if (hipMemsetAsync(dw_tapmajor, 0, (size_t)27 * C * sizeof(float), s) != hipSuccess)
      return DSM_ERR_LAUNCH;
    const long ntile32 = (long)B * D * H * dsm_cdiv(W, 32);
    // one workgroup per CU: the 27 C atomics each workgroup ends with all land on 27 cache lines, and with
    // 768 workgroups they, not the sums, were the kernel's time (120 us)
    const int blocks = (int)(ntile32 < 256 ? ntile32 : 256);
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
