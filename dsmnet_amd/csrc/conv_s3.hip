// "S3" activations and the z-sliding bf16x3 convolution that consumes them.
//
// S3 is fp32 stored pre-split for the bf16 matrix pipe: every value v is kept as the three bf16
// terms of its EXACT split  v = hi + mid + lo  (hi = bf16(v), mid = bf16(v - hi), lo =
// bf16(v - hi - mid), round-to-nearest; lossless for normal fp32 values -- 8 + 8 + 8 significand
// bits), 6 bytes per value, laid out in the fragment order of v_mfma_f32_16x16x32_bf16:
//
//     [b][z][y][cg = c / 32][plane p = 0..2 (hi, mid, lo)][g = 0..3][x][e = 0..7]   (bf16)
//     element e of unit g of channel group cg is channel  32 cg + 16 (e >> 2) + 4 g + (e & 3)
//
// so that (1) a 16-byte unit (plane, g, x) is exactly what one lane of the MFMA's B operand holds
// for voxel x (lane = 16 g + x mod 16: k = 8 g + e), (2) a row of the halo box of one (plane, g)
// is contiguous in HBM (16 B per voxel along x), and (3) the producing kernel's accumulator lane
// (column x, rows 4 g + i of both 16-channel blocks) owns whole units.  The operand split thereby
// leaves the convolution's MFMA stream (it cost 16 % of the r01 kernel, each voxel split 3.6 times)
// and happens once per value, in the epilogue of the layer that produces it.
//
// conv_s3_kernel: Conv3d(k3, s1, p1), Cin % 32 == 0 -> Cout = 32, fp32 accuracy: the same
// six-term bf16x3 product as conv_bf16x3.hpp (wm xm + wl xh + wh xl + wm xh + wh xm + wh xh,
// fp32 accumulate), on 16x16x32 MFMAs.  Replaces convbn_3d + ReLU + myadd_3d of
// models/psmnet/submodule.py:16-19 and stackhourglass.py:10-20,73-98,135-149 for the 32-channel
// full-resolution layers (dres0, dres1, classif*.0), the dominant kernel of the PSMNet forward.
//
// Structure (what changed against conv_bf16x3_kernel and why):
//  * z-sliding: a workgroup owns an (8 y x 32 x) column and walks z.  One input plane is staged
//    ONCE and contributes to three output planes (z-taps 2, 1, 0 -> accumulator sets A0, A1, A2);
//    after a plane, A0 is complete, is written out and the sets rotate.  HBM/L2 traffic for the
//    input falls from 3.6x (three z-taps x 1.2 halo) to 1.33x (the y/x halo only).
//  * the work is the linearised (column, output plane) space cut into gridDim.x EQUAL ranges
//    (one persistent workgroup per CU): no tail round; a range that crosses a column border just
//    starts a new segment.  Partial sums never leave registers: a segment's first and last
//    planes run only the z-taps whose output plane lies inside the segment.
//  * wave (rh, xh) owns rows 4 rh .. 4 rh + 3 and the 16 voxels xh of the tile, all 32 output
//    channels: per (tap position, z-tap) step 6 weight fragments (L2, ring three steps deep)
//    feed 48 MFMAs; the 12 activation fragments of a tap position are read from LDS once and
//    serve its three z-taps.
//  * staging is a plain copy: buffer_load_dwordx4 -> ds_write_b128, no VALU, image
//    [plane][g][voxel] (each (plane, g) row 16-byte units, conflict-free ds_read_b128).
//  * one workgroup per CU, one wave per SIMD, two LDS images (67.6 KB each), one barrier per
//    (plane, 32-channel group).
#include "common.hpp"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {        // low half = a
  const f32x2 t = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// eight fp32 (elements e = 0..7 of one unit) -> the unit's three planes
__device__ __forceinline__ void split_unit(const f32x4 lo4, const f32x4 hi4, u32x4 (&pl)[3]) {
  float r[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    unsigned u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i] = pack_bf16(r[2 * i], r[2 * i + 1]);
      if (k < 2) { r[2 * i] -= bf16_lo(u[i]); r[2 * i + 1] -= bf16_hi(u[i]); }
    }
    pl[k] = u32x4{u[0], u[1], u[2], u[3]};
  }
}
__device__ __forceinline__ void split_half(const f32x4 v4, u32x2 (&pl)[3]) {   // 4 of a unit's 8 elements
  float r[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    unsigned u[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u[i] = pack_bf16(r[2 * i], r[2 * i + 1]);
      if (k < 2) { r[2 * i] -= bf16_lo(u[i]); r[2 * i + 1] -= bf16_hi(u[i]); }
    }
    pl[k] = u32x2{u[0], u[1]};
  }
}
__device__ __forceinline__ void join_unit(const u32x4 (&pl)[3], f32x4& lo4, f32x4& hi4) {
  float r[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[2 * i] = (bf16_lo(pl[0][i]) + bf16_lo(pl[1][i])) + bf16_lo(pl[2][i]);
    r[2 * i + 1] = (bf16_hi(pl[0][i]) + bf16_hi(pl[1][i])) + bf16_hi(pl[2][i]);
  }
  lo4 = f32x4{r[0], r[1], r[2], r[3]};
  hi4 = f32x4{r[4], r[5], r[6], r[7]};
}

__device__ __forceinline__ f32x4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset,
                                               unsigned soffset) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voffset, (int)soffset, 0);
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// ----------------------------------------------------------------------------
// format conversion (tests, interoperation with the fp32 kernels)
// ----------------------------------------------------------------------------
// x: (B,D,H,W,C) fp32 NDHWC -> s3.  thread = (voxel row position, cg, g)
__global__ __launch_bounds__(256) void s3_from_ndhwc_kernel(const float* __restrict__ x,
                                                            unsigned char* __restrict__ s3, long nrows,
                                                            int W, int C) {
  const int ncg = C / 32;
  const long n = nrows * W * ncg * 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long t = i;
  const int xx = t % W; t /= W;
  const int g = t & 3; t >>= 2;
  const int cg = t % ncg; const long row = t / ncg;
  const float* v = x + (row * W + xx) * C + 32 * cg + 4 * g;
  u32x4 pl[3];
  split_unit(*reinterpret_cast<const f32x4*>(v), *reinterpret_cast<const f32x4*>(v + 16), pl);
  unsigned char* o = s3 + (((row * ncg + cg) * 12 + g) * W + xx) * 16;
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(o + (long)p * 4 * W * 16) = pl[p];
}

__global__ __launch_bounds__(256) void s3_to_ndhwc_kernel(const unsigned char* __restrict__ s3,
                                                          float* __restrict__ x, long nrows, int W, int C) {
  const int ncg = C / 32;
  const long n = nrows * W * ncg * 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long t = i;
  const int g = t & 3; t >>= 2;
  const int cg = t % ncg; t /= ncg;
  const int xx = t % W; const long row = t / W;
  const unsigned char* o = s3 + (((row * ncg + cg) * 12 + g) * W + xx) * 16;
  u32x4 pl[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) pl[p] = *reinterpret_cast<const u32x4*>(o + (long)p * 4 * W * 16);
  f32x4 lo4, hi4;
  join_unit(pl, lo4, hi4);
  float* v = x + (row * W + xx) * C + 32 * cg + 4 * g;
  *reinterpret_cast<f32x4*>(v) = lo4;
  *reinterpret_cast<f32x4*>(v + 16) = hi4;
}

// ----------------------------------------------------------------------------
// concatenation cost volume written as S3 (models/gcnet.py:130-135, mask_left = 0;
// models/psmnet/stackhourglass.py:124-133, mask_left = 1).  The volume is a shifted copy of the
// features, so the features are split once (feat_s3_kernel: NCHW fp32 -> [b][y][side][cgf][12][x]
// units) and the volume build is a pure 16-byte copy: output row (b, d, y) = the left row-set with
// x < d zeroed (PSMNet) | the right row-set shifted by d units.  One pass, no memset.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void feat_s3_kernel(const float* __restrict__ fL,
                                                      const float* __restrict__ fR,
                                                      unsigned char* __restrict__ fs, int B, int C,
                                                      int H, int W) {
  const int ncg = C / 32;
  const long n = (long)B * H * 2 * ncg * 4 * W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long t = i;
  const int xx = t % W; t /= W;
  const int g = t & 3; t >>= 2;
  const int cg = t % ncg; t /= ncg;
  const int side = t & 1; t >>= 1;
  const int y = t % H; const int b = t / H;
  const float* f = (side ? fR : fL) + (((long)b * C + 32 * cg + 4 * g) * H + y) * W + xx;
  const long cs = (long)H * W;
  const f32x4 lo4 = {f[0], f[cs], f[2 * cs], f[3 * cs]};
  const f32x4 hi4 = {f[16 * cs], f[17 * cs], f[18 * cs], f[19 * cs]};
  u32x4 pl[3];
  split_unit(lo4, hi4, pl);
  unsigned char* o = fs + ((((((long)b * H + y) * 2 + side) * ncg + cg) * 12 + g) * W + xx) * 16;
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(o + (long)p * 4 * W * 16) = pl[p];
}

// block = one (b, d-group of DG planes, y): the row-set of (b, y) is read once per block
// (L2-resident: the whole split feature buffer is a few MB) and written DG times.
template <int DG>
__global__ __launch_bounds__(256) void volume_s3_kernel(const unsigned char* __restrict__ fs,
                                                        unsigned char* __restrict__ vol, int B, int C,
                                                        int H, int W, int D, int mask_left) {
  const int ncg = C / 32;                     // per side; the volume has 2 ncg groups
  const int rows = ncg * 12;                  // (cg, plane, g) rows of W units per side
  const int y = blockIdx.x, d0 = blockIdx.y * DG, b = blockIdx.z;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  const unsigned char* src = fs + (((long)b * H + y) * 2) * rows * W * 16;
  const int per_side = rows * W;
  for (int i = threadIdx.x; i < 2 * per_side; i += 256) {
    const int side = i >= per_side;
    const int rem = i - side * per_side;
    const int row = rem / W, xx = rem % W;
#pragma unroll
    for (int dd = 0; dd < DG; ++dd) {
      const int d = d0 + dd;
      if (d >= D) break;
      u32x4 v = zero;
      if (side == 0) {
        if (!mask_left || xx >= d) v = *reinterpret_cast<const u32x4*>(src + ((long)row * W + xx) * 16);
      } else if (xx >= d) {
        v = *reinterpret_cast<const u32x4*>(src + ((long)(rows + row) * W + xx - d) * 16);
      }
      unsigned char* o = vol + (((((long)b * D + d) * H + y) * 2 + side) * rows * W + (long)row * W + xx) * 16;
      *reinterpret_cast<u32x4*>(o) = v;
    }
  }
}

// ----------------------------------------------------------------------------
// weights: torch (Cout = 32, Cin, 3,3,3) -> [cg][tp = ky*3+kx][kz][a = cout/16][plane][lane][8 bf16]
// lane = 16 kg + m holds A[row m = cout % 16][k = 8 kg + e], k <-> channel 32 cg + 16(e>>2) + 4 kg + (e&3)
// ----------------------------------------------------------------------------
__global__ void pack_weights_s3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                       int Cin) {
  const long n = (long)Cin * 32 * 27;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  long i = idx;
  const int e = i & 7; i >>= 3;
  const int lane = i & 63; i >>= 6;
  const int a = i & 1; i >>= 1;
  const int kz = i % 3; i /= 3;
  const int tp = i % 9; const int cg = i / 9;
  const int m = lane & 15, kg = lane >> 4;
  const int cout = 16 * a + m, cin = 32 * cg + 16 * (e >> 2) + 4 * kg + (e & 3);
  const int tap = kz * 9 + tp;
  float v = w[((long)cout * Cin + cin) * 27 + tap];
  unsigned short* o = out + ((((((long)cg * 9 + tp) * 3 + kz) * 2 + a) * 3) * 64 + lane) * 8 + e;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u = pack_bf16(v, 0.f);
    o[(long)q * 64 * 8] = (unsigned short)(u & 0xffffu);
    v -= bf16_lo(u);
  }
}

// ----------------------------------------------------------------------------
// the convolution
// ----------------------------------------------------------------------------
struct S3ConvParams {
  const unsigned char* x;       // S3 (B,Di,Hi,Wi,Cin)
  const unsigned char* w;       // packed by pack_weights_s3_kernel
  const float* scale; const float* shift;
  const float* res;             // fp32 NDHWC (B,Dr,Hr,Wr,32) or null
  float* y;                     // fp32 NDHWC (B,Do,Ho,Wo,32) or null
  unsigned char* ys3;           // S3 (B,Do,Ho,Wo,32) or null
  int B, Cin;
  int Di, Hi, Wi, Do, Ho, Wo, Dr, Hr, Wr;
  int relu;
  int vol, vol_mask_left;       // vol: x is the split feature maps of a VIRTUAL cost volume (see the host entry)
  int ntx, nty, ncol;           // columns: (b, ty, tx), 8 x 32 outputs each
  long nunits;                  // ncol * Do
  unsigned wbytes;
};

// Two tilings of the same kernel (dsm_conv3d_s3_args.tiling; S3_DEFAULT_TILING when that is 0):
//   V = 0: 8 x 32 tile, one workgroup per CU, wave = (row half, x half), both 16-channel A blocks;
//   V = 1: 4 x 32 tile, two workgroups per CU (<= 256 registers), wave = (cout half, x half).
#ifndef S3_DEFAULT_TILING
#define S3_DEFAULT_TILING 1   // 8 x 32: the whole-forward A/B (profiles/r02_ablation.md section 4)
#endif
constexpr int S3_IX = 34;
constexpr int S3_NSTEP = 27;
constexpr int S3_WSTEP = 6 * 1024;                // weight bytes per (tap position, z-tap) step
template <int V> struct S3Cfg {
  static constexpr int TY = V == 0 ? 8 : 4, IY = TY + 2;
  static constexpr int NV = IY * S3_IX;                      // voxels of the halo box: 340 | 204
  static constexpr int NVP = (NV + 15) / 16 * 16;            // (plane, g) rows 256-B aligned: 352 | 208
  static constexpr int ROW = NVP * 16;                       // 5,632 | 3,328 B
  static constexpr int IMG = 12 * ROW;                       // 67,584 B = 66 pieces | 39,936 B = 39 pieces
  static constexpr int NPIECE = IMG / 1024;
  static constexpr int NPW = (NPIECE + 3) / 4;               // pieces per wave: 17 | 10
  // V = 0: the image stride leaves two spare pieces for waves 2, 3's 17th store; V = 1 has no room
  // (two workgroups share the CU's 160 KiB) and predicates the last piece instead.
  static constexpr bool SPARE = V == 0;
  static constexpr int IMGP = SPARE ? 4 * NPW * 1024 : IMG;
  static constexpr int LDS = 2 * IMGP + 256;
  static constexpr int NA = V == 0 ? 2 : 1;                  // 16-channel A blocks per wave
  static constexpr int WGS = V == 0 ? 1 : 2;                 // workgroups per CU
  static_assert(IMG % 1024 == 0, "whole 1-KiB pieces");
  static_assert(NPW <= S3_NSTEP - 3 + 4, "staging schedule");
};

template <int V>
__global__ __launch_bounds__(256, S3Cfg<V>::WGS) void conv_s3_kernel(S3ConvParams p) {
  using C = S3Cfg<V>;
  constexpr int S3_TY = C::TY, S3_NV = C::NV, S3_NVP = C::NVP, S3_ROW = C::ROW, S3_IMGP = C::IMGP,
                S3_NPW = C::NPW, NA = C::NA;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int hi = wave >> 1, xh = wave & 1;
  const int row0 = V == 0 ? 4 * hi : 0;       // V = 0: the wave's 4 rows of the 8-row tile
  const int a0 = V == 0 ? 0 : hi;             // V = 1: the wave's 16-channel block
  const int ncg = p.Cin >> 5;

  // this workgroup's range of the linearised (column, output plane) space; workgroups on one XCD
  // (id % 8) take neighbouring ranges (shared halo columns meet in one L2; speed only)
  const int G = gridDim.x, id = blockIdx.x;
  const int logical = (G & 7) == 0 ? (id & 7) * (G >> 3) + (id >> 3) : id;
  const long u_begin = p.nunits * logical / G, u_end = p.nunits * (logical + 1) / G;
  if (u_begin >= u_end) return;

  // folded-BN affine of this lane's 8 channels (16 a + 4 g + i)
  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * S3_IMGP);
  if (tid < 64) aff[tid] = tid < 32 ? (p.scale ? p.scale[tid] : 1.f) : (p.shift ? p.shift[tid - 32] : 0.f);
  __syncthreads();
  f32x4 sc[NA], sh[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    sc[a] = *reinterpret_cast<const f32x4*>(aff + 16 * (a0 + a) + 4 * g);
    sh[a] = *reinterpret_cast<const f32x4*>(aff + 32 + 16 * (a0 + a) + 4 * g);
  }

  // ---- chunk iterator: live (input plane, channel group) pairs of the segments of [u_begin, u_end)
  struct It { long u; int col, z0, z1, zlo, zhi, zi, cg; bool valid; };
  auto open_segment = [&](long u) {
    It q; q.u = u; q.valid = u < u_end;
    if (!q.valid) { q.col = q.z0 = q.z1 = q.zlo = q.zhi = q.zi = q.cg = 0; return q; }
    q.col = (int)(u / p.Do); q.z0 = (int)(u % p.Do);
    const long left = u_end - u;
    q.z1 = (int)min((long)p.Do, (long)q.z0 + left);
    q.zlo = max(q.z0 - 1, 0); q.zhi = min(q.z1, p.Di - 1);
    q.zi = q.zlo; q.cg = 0;
    return q;
  };
  auto advance = [&](It q) {
    if (++q.cg < ncg) return q;
    q.cg = 0;
    if (++q.zi <= q.zhi) return q;
    return open_segment(q.u + (q.z1 - q.z0));
  };

  // ---- staging: piece k = wave + 4 t of the image, this lane's 16-byte slot 64 k + lane
  const unsigned row_units = (unsigned)ncg * 12u * (unsigned)p.Wi;       // units per input row y
  const unsigned plane_bytes_lo = row_units * 16u * (unsigned)p.Hi;      // < 4 GiB: checked by the host
  constexpr unsigned OOBV = 0x80000000u;
  unsigned voff[S3_NPW];
  int vx[S3_NPW];               // the slot's x coordinate (virtual volume: plane d masks x < d)
  auto column_offsets = [&](int col) {
    const int tx = col % p.ntx, ty = (col / p.ntx) % p.nty;
    const int y0 = ty * S3_TY - 1, x0 = tx * 32 - 1;
#pragma unroll
    for (int t = 0; t < S3_NPW; ++t) {
      const int slot = 64 * (wave + 4 * t) + lane;
      const int pg = slot / S3_NVP, v = slot % S3_NVP;
      const int yy = v / S3_IX, xx = v % S3_IX;
      const int y = y0 + yy, x = x0 + xx;
      const bool ok = slot < 12 * S3_NVP && v < S3_NV && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
      voff[t] = ok ? ((unsigned)y * row_units + (unsigned)pg * (unsigned)p.Wi + (unsigned)x) * 16u : OOBV;
      vx[t] = x;
    }
  };
  // Virtual cost volume (p.vol): the "tensor" is the split feature maps, S3 of (B, Cin, 1, H, W) with
  // channel groups [left | right]; plane d of the volume is that one plane with the right half
  // shifted by d voxels (the descriptor base moves back by d units) and x < d zeroed (right half
  // always, left half when mask_left: models/psmnet/stackhourglass.py:128-129 vs gcnet.py:134).
  auto chunk_rsrc = [&](const It& q) {
    const int b = q.col / (p.ntx * p.nty);
    const bool right = p.vol && 2 * q.cg >= ncg;
    const long shift = right ? (long)q.zi * 16 : 0;
    const long off = (p.vol ? (long)b : (long)b * p.Di + q.zi) * (long)plane_bytes_lo +
                     (long)q.cg * 12 * p.Wi * 16 - shift;
    return make_rsrc(p.x + off, q.valid ? plane_bytes_lo - (unsigned)q.cg * 12u * (unsigned)p.Wi * 16u + (unsigned)shift : 0u);
  };
  auto chunk_xmin = [&](const It& q) {            // staged voxels with x below this are zeros
    return (p.vol && (p.vol_mask_left || 2 * q.cg >= ncg)) ? q.zi : -0x40000000;
  };
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  const unsigned lane16 = lane * 16u;
  const int st_off = wave * 1024 + lane * 16;                  // + 4096 t: the slot of piece wave + 4 t
  auto piece_ok = [&](int t) { return C::SPARE || wave + 4 * t < C::NPIECE; };
  // activation fragment of this lane: voxel (row0 + r + ky, 16 xh + j + kx), unit g, plane pl
  const int rd_off = g * S3_ROW + (row0 * S3_IX + 16 * xh + j) * 16;

  f32x4 acc[3][4][NA];
  // [tap-position parity][row][plane]; V = 1 has one set: a row's fragment of the next tap position
  // takes the registers of the row just finished in the tap position's last step
  constexpr int XQ = V == 0 ? 2 : 1;
  bf16x8 xq[XQ][4][3];
  bf16x8 wq[3][NA][3];          // [step % 3][a][plane]
  f32x4 stg[4];

  auto zero_set = [&](auto sc_) {
    constexpr int s = decltype(sc_)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[s][r][a] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto wload = [&](auto sc_, unsigned wb) {                     // weights of step s (of the chunk at wb)
    constexpr int s = decltype(sc_)::value;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        wq[s % 3][a][q] = __builtin_bit_cast(
            bf16x8, buffer_load16(wrsrc, lane16 + (unsigned)a0 * 3072u, wb + s * S3_WSTEP + (a * 3 + q) * 1024));
  };
  auto xload = [&](auto tpc, auto rc, const unsigned char* rd) {
    constexpr int tp = decltype(tpc)::value, r = decltype(rc)::value;
    constexpr int ky = tp / 3, kx = tp % 3;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      xq[tp & (XQ - 1)][r][q] = *reinterpret_cast<const bf16x8*>(rd + q * 4 * S3_ROW + ((r + ky) * S3_IX + kx) * 16);
  };

  // epilogue of accumulator set 0 = output plane zo of the column
  auto emit = [&](const It& q, int zo) {
    const int tx = q.col % p.ntx, ty = (q.col / p.ntx) % p.nty, b = q.col / (p.ntx * p.nty);
    const int xo = tx * 32 + 16 * xh + j;
    if (xo >= p.Wo) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yo = ty * S3_TY + row0 + r;
      if (yo >= p.Ho) continue;
      f32x4 v[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        v[a] = acc[0][r][a] * sc[a] + sh[a];
        if (p.relu == 2) { v[a].x = fmaxf(v[a].x, 0.f); v[a].y = fmaxf(v[a].y, 0.f); v[a].z = fmaxf(v[a].z, 0.f); v[a].w = fmaxf(v[a].w, 0.f); }
      }
      if (p.res) {
        const float* rv = p.res + ((((long)b * p.Dr + zo) * p.Hr + yo) * p.Wr + xo) * 32 + 16 * a0 + 4 * g;
#pragma unroll
        for (int a = 0; a < NA; ++a) v[a] += *reinterpret_cast<const f32x4*>(rv + 16 * a);
      }
      if (p.relu == 1) {
#pragma unroll
        for (int a = 0; a < NA; ++a) { v[a].x = fmaxf(v[a].x, 0.f); v[a].y = fmaxf(v[a].y, 0.f); v[a].z = fmaxf(v[a].z, 0.f); v[a].w = fmaxf(v[a].w, 0.f); }
      }
      const long vox = (((long)b * p.Do + zo) * p.Ho + yo) * p.Wo + xo;
      if (p.y) {
        float* yv = p.y + vox * 32 + 16 * a0 + 4 * g;
#pragma unroll
        for (int a = 0; a < NA; ++a) *reinterpret_cast<f32x4*>(yv + 16 * a) = v[a];
      }
      if (p.ys3) {
        unsigned char* o = p.ys3 + ((((((long)b * p.Do + zo) * p.Ho + yo) * 12) + g) * p.Wo + xo) * 16;
        if constexpr (NA == 2) {
          u32x4 pl[3];
          split_unit(v[0], v[1], pl);
#pragma unroll
          for (int q2 = 0; q2 < 3; ++q2) *reinterpret_cast<u32x4*>(o + (long)q2 * 4 * p.Wo * 16) = pl[q2];
        } else {                        // this wave's half of the unit: elements 4 a0 .. 4 a0 + 3
          u32x2 pl[3];
          split_half(v[0], pl);
#pragma unroll
          for (int q2 = 0; q2 < 3; ++q2) *reinterpret_cast<u32x2*>(o + (long)q2 * 4 * p.Wo * 16 + 8 * a0) = pl[q2];
        }
      }
    }
  };

  // ---- first chunk of the range: staged synchronously
  It cur = open_segment(u_begin);
  column_offsets(cur.col);
  {
    const __amdgpu_buffer_rsrc_t rs0 = chunk_rsrc(cur);
    const int xmin0 = chunk_xmin(cur);
#pragma unroll
    for (int t = 0; t < S3_NPW; ++t)
      if (piece_ok(t))
        *reinterpret_cast<f32x4*>(lds_raw + st_off + 4096 * t) =
            buffer_load16(rs0, vx[t] >= xmin0 ? voff[t] : OOBV, 0);
  }
  static_for<0, 3>([&](auto s) { zero_set(s); });
  const unsigned w0 = (unsigned)cur.cg * (S3_NSTEP * S3_WSTEP);
  wload(std::integral_constant<int, 0>{}, w0);
  wload(std::integral_constant<int, 1>{}, w0);
  int img = 0;

  while (true) {
    __syncthreads();            // image `img` is complete; everyone is done reading image `img ^ 1`
    const unsigned char* const rd = lds_raw + img * S3_IMGP + rd_off;
    unsigned char* const wr = lds_raw + (img ^ 1) * S3_IMGP + st_off;
    const It nxt = advance(cur);
    if (nxt.valid && nxt.col != cur.col) column_offsets(nxt.col);
    const __amdgpu_buffer_rsrc_t nrsrc = chunk_rsrc(nxt);
    const int nxmin = chunk_xmin(nxt);
    const unsigned wcur = (unsigned)cur.cg * (S3_NSTEP * S3_WSTEP);
    const unsigned wnext = nxt.valid ? (unsigned)nxt.cg * (S3_NSTEP * S3_WSTEP) : 0u;
    // z-tap kz of input plane zi feeds output plane zi - kz + 1: only inside [z0, z1)
    unsigned mask = 0;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      const int zo = cur.zi - kz + 1;
      if (zo >= cur.z0 && zo < cur.z1) mask |= 1u << kz;
    }
    static_for<0, 4>([&](auto rc) { xload(std::integral_constant<int, 0>{}, rc, rd); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, S3_NSTEP>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      constexpr int tp = s / 3, kz = s % 3;
      // unconditional part of the step: weight ring, next tap position's fragments, staging
      if constexpr (s + 2 < S3_NSTEP) wload(std::integral_constant<int, s + 2>{}, wcur);
      else wload(std::integral_constant<int, s + 2 - S3_NSTEP>{}, wnext);
      if constexpr (V == 0 && tp + 1 < 9) {
        if constexpr (kz == 0) {
          xload(std::integral_constant<int, tp + 1>{}, std::integral_constant<int, 0>{}, rd);
          xload(std::integral_constant<int, tp + 1>{}, std::integral_constant<int, 1>{}, rd);
        } else if constexpr (kz == 1) {
          xload(std::integral_constant<int, tp + 1>{}, std::integral_constant<int, 2>{}, rd);
          xload(std::integral_constant<int, tp + 1>{}, std::integral_constant<int, 3>{}, rd);
        }
      }
      if constexpr (s < S3_NPW) stg[s % 4] = buffer_load16(nrsrc, vx[s] >= nxmin ? voff[s] : OOBV, 0);
      if constexpr (s >= 3 && s - 3 < S3_NPW)
        if (piece_ok(s - 3)) *reinterpret_cast<f32x4*>(wr + 4096 * (s - 3)) = stg[(s - 3) % 4];
      __builtin_amdgcn_sched_barrier(0);
      constexpr int set = 2 - kz;
      auto row_mfmas = [&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const bf16x8 xh_ = xq[tp & (XQ - 1)][r][0], xm = xq[tp & (XQ - 1)][r][1], xl = xq[tp & (XQ - 1)][r][2];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const bf16x8 wh = wq[s % 3][a][0], wm = wq[s % 3][a][1], wl = wq[s % 3][a][2];
          f32x4 c = acc[set][r][a];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh_, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh_, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh_, c, 0, 0, 0);
          acc[set][r][a] = c;
        }
      };
      if constexpr (V == 1 && kz == 2 && tp + 1 < 9) {
        static_for<0, 4>([&](auto rc) {
          if (mask & (1u << kz)) row_mfmas(rc);
          __builtin_amdgcn_sched_barrier(0);
          xload(std::integral_constant<int, tp + 1>{}, rc, rd);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        if (mask & (1u << kz)) static_for<0, 4>(row_mfmas);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // the last pieces of the next image (steps 24..26 carried pieces 14..16 in flight)
    static_for<S3_NSTEP - 3, S3_NPW>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (piece_ok(t)) *reinterpret_cast<f32x4*>(wr + 4096 * t) = stg[t % 4];
    });
    if (cur.cg == ncg - 1) {                                    // the plane is complete
      const int zo = cur.zi - 1;
      if (zo >= cur.z0 && zo < cur.z1) emit(cur, zo);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < NA; ++a) { acc[0][r][a] = acc[1][r][a]; acc[1][r][a] = acc[2][r][a]; }
      zero_set(std::integral_constant<int, 2>{});
      if (cur.zi == cur.zhi) {                                  // segment ends
        if (cur.zhi >= cur.z0 && cur.zhi < cur.z1) emit(cur, cur.zhi);   // only when z1 = Di: no plane Di follows
        zero_set(std::integral_constant<int, 0>{});
        zero_set(std::integral_constant<int, 1>{});
      }
    }
    cur = nxt; img ^= 1;
    if (!cur.valid) break;
  }
}

template <int V>
int launch_conv_s3(S3ConvParams p, const dsm_conv3d_s3_args* a, hipStream_t s) {
  using C = S3Cfg<V>;
  p.ntx = dsm_cdiv(a->Wo, 32); p.nty = dsm_cdiv(a->Ho, C::TY);
  const long ncol = (long)a->B * p.nty * p.ntx;
  DSM_REQUIRE(ncol < (1L << 30), DSM_ERR_UNSUPPORTED);
  p.ncol = (int)ncol;
  p.nunits = ncol * a->Do;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)conv_s3_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS) != hipSuccess)
      return DSM_ERR_LAUNCH;
    configured = true;
  }
  int blocks = a->grid > 0 ? a->grid : 256 * C::WGS;           // persistent workgroups: WGS per CU
  if ((long)blocks > p.nunits) blocks = (int)p.nunits;
  hipLaunchKernelGGL(conv_s3_kernel<V>, dim3(blocks), dim3(256), C::LDS, s, p);
  return dsm_launch_status();
}

int check_s3_dims(int B, int C, int D, int H, int W) {
  DSM_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, DSM_ERR_ARG);
  DSM_REQUIRE(C % 32 == 0, DSM_ERR_UNSUPPORTED);
  return DSM_OK;
}

}  // namespace

extern "C" size_t dsm_s3_bytes(int B, int C, int D, int H, int W) {
  if (B <= 0 || C <= 0 || C % 32 || D <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * D * H * W * C * 6;
}

extern "C" int dsm_s3_from_ndhwc(const void* x, void* s3, int B, int C, int D, int H, int W,
                                 dsm_stream_t stream) {
  DSM_REQUIRE(x && s3, DSM_ERR_ARG);
  int rc = check_s3_dims(B, C, D, H, W);
  if (rc != DSM_OK) return rc;
  DSM_REQUIRE(dsm_aligned16(x) && dsm_aligned16(s3), DSM_ERR_ALIGN);
  const long nrows = (long)B * D * H;
  const long n = nrows * W * (C / 32) * 4;
  DSM_REQUIRE(n / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(s3_from_ndhwc_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (unsigned char*)s3, nrows, W, C);
  return dsm_launch_status();
}

extern "C" int dsm_s3_to_ndhwc(const void* s3, void* x, int B, int C, int D, int H, int W,
                               dsm_stream_t stream) {
  DSM_REQUIRE(x && s3, DSM_ERR_ARG);
  int rc = check_s3_dims(B, C, D, H, W);
  if (rc != DSM_OK) return rc;
  DSM_REQUIRE(dsm_aligned16(x) && dsm_aligned16(s3), DSM_ERR_ALIGN);
  const long nrows = (long)B * D * H;
  const long n = nrows * W * (C / 32) * 4;
  DSM_REQUIRE(n / 256 < 0x7fffffffL, DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  hipLaunchKernelGGL(s3_to_ndhwc_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)s3, (float*)x, nrows, W, C);
  return dsm_launch_status();
}

extern "C" size_t dsm_concat_volume_s3_scratch_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || C % 32 || H <= 0 || W <= 0) return 0;
  return (size_t)B * H * W * 2 * C * 6;
}

extern "C" int dsm_concat_volume_s3_fwd(const void* fL, const void* fR, void* scratch, void* vol,
                                        int B, int C, int H, int W, int D, int mask_left,
                                        dsm_stream_t stream) {
  DSM_REQUIRE(fL && fR && scratch && vol, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0, DSM_ERR_ARG);
  DSM_REQUIRE(C % 32 == 0 && B <= 65535 && H <= 0x7fffffff, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(dsm_aligned16(scratch) && dsm_aligned16(vol), DSM_ERR_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  const long n = (long)B * H * 2 * (C / 32) * 4 * W;
  hipLaunchKernelGGL(feat_s3_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s, (const float*)fL,
                     (const float*)fR, (unsigned char*)scratch, B, C, H, W);
  constexpr int DG = 4;
  hipLaunchKernelGGL(volume_s3_kernel<DG>, dim3(H, dsm_cdiv(D, DG), B), dim3(256), 0, s,
                     (const unsigned char*)scratch, (unsigned char*)vol, B, C, H, W, D, mask_left);
  return dsm_launch_status();
}

extern "C" int dsm_features_s3(const void* fL, const void* fR, void* fs, int B, int C, int H, int W,
                               dsm_stream_t stream) {
  DSM_REQUIRE(fL && fR && fs, DSM_ERR_ARG);
  DSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DSM_ERR_ARG);
  DSM_REQUIRE(C % 32 == 0, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(dsm_aligned16(fs), DSM_ERR_ALIGN);
  dsm_clear_stale_error();
  const long n = (long)B * H * 2 * (C / 32) * 4 * W;
  hipLaunchKernelGGL(feat_s3_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)fL, (const float*)fR, (unsigned char*)fs, B, C, H, W);
  return dsm_launch_status();
}

extern "C" size_t dsm_conv3d_s3_packed_weight_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cin % 32 || Cout != 32) return 0;
  return (size_t)Cin * Cout * 27 * 6;
}

extern "C" int dsm_conv3d_s3_pack_weights(const void* w_torch, void* w_packed, int Cin, int Cout,
                                          dsm_stream_t stream) {
  DSM_REQUIRE(w_torch && w_packed, DSM_ERR_ARG);
  DSM_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout == 32, DSM_ERR_UNSUPPORTED);
  dsm_clear_stale_error();
  const long n = (long)Cin * 32 * 27;
  hipLaunchKernelGGL(pack_weights_s3_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)w_torch, (unsigned short*)w_packed, Cin);
  return dsm_launch_status();
}

extern "C" int dsm_conv3d_s3_fwd(const dsm_conv3d_s3_args* a, dsm_stream_t stream) {
  DSM_REQUIRE(a && a->x_s3 && a->w_packed && (a->y || a->y_s3), DSM_ERR_ARG);
  DSM_REQUIRE(a->B > 0 && a->Di > 0 && a->Hi > 0 && a->Wi > 0 && a->Do > 0 && a->Ho > 0 && a->Wo > 0,
              DSM_ERR_ARG);
  DSM_REQUIRE(a->Cin > 0 && a->Cin % 32 == 0 && a->Cout == 32, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(a->Do <= a->Di && a->Ho <= a->Hi && a->Wo <= a->Wi, DSM_ERR_ARG);
  DSM_REQUIRE(a->relu >= 0 && a->relu <= 2, DSM_ERR_ARG);
  if (a->vol_virtual) DSM_REQUIRE(a->Cin % 64 == 0, DSM_ERR_UNSUPPORTED);   // [left | right], 32-channel groups each
  if (a->residual)
    DSM_REQUIRE(a->Dr >= a->Do && a->Hr >= a->Ho && a->Wr >= a->Wo, DSM_ERR_ARG);
  DSM_REQUIRE(dsm_aligned16(a->x_s3) && dsm_aligned16(a->w_packed) && dsm_aligned16(a->y) &&
              dsm_aligned16(a->y_s3) && dsm_aligned16(a->residual), DSM_ERR_ALIGN);
  // 32-bit offsets are per input plane (the descriptor base moves with the plane), not per tensor
  const unsigned long plane_bytes = (unsigned long)a->Hi * a->Wi * a->Cin * 6ul;
  DSM_REQUIRE(plane_bytes < 0x7fffffffUL, DSM_ERR_UNSUPPORTED);
  S3ConvParams p;
  p.x = (const unsigned char*)a->x_s3; p.w = (const unsigned char*)a->w_packed;
  p.scale = a->scale; p.shift = a->shift; p.res = (const float*)a->residual;
  p.y = (float*)a->y; p.ys3 = (unsigned char*)a->y_s3;
  p.B = a->B; p.Cin = a->Cin;
  p.Di = a->Di; p.Hi = a->Hi; p.Wi = a->Wi; p.Do = a->Do; p.Ho = a->Ho; p.Wo = a->Wo;
  p.Dr = a->Dr; p.Hr = a->Hr; p.Wr = a->Wr; p.relu = a->relu;
  p.vol = a->vol_virtual ? 1 : 0; p.vol_mask_left = a->vol_mask_left ? 1 : 0;
  p.wbytes = (unsigned)dsm_conv3d_s3_packed_weight_bytes(a->Cin, 32);
  dsm_clear_stale_error();
  return launch_conv_s3<0>(p, a, (hipStream_t)stream);
}
