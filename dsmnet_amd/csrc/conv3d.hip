// 3-D convolution blocks (k = 3) of the GCNet / PSMNet regularisation trunk on
// gfx950 matrix cores -- implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// Replaces nn.Conv3d / nn.ConvTranspose3d + nn.BatchNorm3d + ReLU + the cropped
// skip additions of models/psmnet/stackhourglass.py:22-62,73-98,135-149,
// models/psmnet/submodule.py:16-19, models/gcnet.py:32-101 and
// models/util_conv.py:150-179 with one launch per layer and a fused epilogue.
//
// GEMM view: M = output voxels, N = Cout, K = 27 taps x Cin.  The f32-input MFMA is
// an exact fp32 fma chain (products and sums rounded once each, no reduced
// precision), so parity with the reference's fp32 path needs no tolerance beyond
// summation order; its rate is the fp32 vector rate (157 TF/s dense peak), 1/16 of
// bf16, which makes the kernel MFMA-issue bound by a wide margin: per MFMA (64
// cycles per SIMD) a wave needs 16 B of A per lane per 4 MFMAs from LDS and 16 B
// of B per lane per 4*TM MFMAs from L2.  The design therefore spends nothing on
// bandwidth tricks and everything on keeping the matrix pipe issuing:
//
//  * activations are NDHWC (channels_last_3d), a voxel's channels contiguous;
//  * a workgroup (4 waves) owns an output tile of 1 z x (4*TM) y x 32 x; wave w
//    owns TM rows; an M-tile is 32 consecutive x of one row (MFMA rows);
//  * K is walked in chunks of CK input channels: the chunk's halo tile
//    (3 x IY x IX voxels x CK channels) is staged global -> VGPR -> LDS while the
//    previous chunk is being multiplied (register prefetch, one LDS buffer, two
//    workgroups per CU overlap each other's barriers);
//  * LDS image is [z][y][q][x] in 16-B elements (q = 4-channel group): the 64
//    lanes of an A-fragment read touch consecutive 16-B slots -> conflict-free
//    ds_read_b128 with compile-time offsets for every tap; stride-2 convolutions
//    de-interleave even/odd x so their reads stay unit-stride;
//  * weights are pre-packed once per layer into MFMA B-fragment order
//    [cin/8][tap][cout/32][lane][4] and streamed from L2 with 16 B per lane,
//    three fragments ahead of their use;
//  * epilogue: y = acc*scale[co] + shift[co] (+ skip) (ReLU), 128 B per voxel.
//
// K order inside an 8-channel group: MFMA step j multiplies channels
// {8g + j (lanes 0-31), 8g + 4 + j (lanes 32-63)}; A and B fragments agree on it.
#include "common.hpp"
#include <type_traits>

namespace {

// Compile-time loop: f(integral_constant<int, I>) for I in [I0, N).  Used where an index must
// be a constant expression so that accumulator arrays stay in registers (a runtime-indexed
// ext-vector array goes to scratch).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int NTHREADS = 256;

// Input-tile geometry for an output tile of TY rows x 32 columns, one z.
// KZ x KXY x KXY taps (KZ = 1: a 2-D convolution on (B,1,H,W,C) volumes), dilation DIL in
// (y, x), padding = "same" ((K-1)/2 * dilation), stride S.
template <int S, int KZ = 3, int KXY = 3, int DIL = 1> struct Geo {
  static constexpr int IZ = KZ;
  static constexpr int EXT = (KXY - 1) * DIL;           // halo span in y and x
  static constexpr int PADZ = (KZ - 1) / 2, PADXY = EXT / 2;
  static constexpr int NTAP = KZ * KXY * KXY;
  static __host__ __device__ constexpr int IY(int TY) { return (TY - 1) * S + EXT + 1; }
  static constexpr int IX = 31 * S + EXT + 1;           // 34 or 65 for the 3x3x3 trunk
  static constexpr int XE = (IX + 1) / 2;               // even columns when S == 2
  static constexpr int XP = (S == 1) ? IX : 2 * XE;     // LDS row pitch in 16-B elements
  static __host__ __device__ constexpr int xmap(int x) {
    return S == 1 ? x : ((x & 1) * XE + (x >> 1));
  }
};

struct ConvParams {
  const float* x; const float* w; const float* scale; const float* shift;
  const float* res; float* y;
  int B, Cin, Cout;
  int Di, Hi, Wi, Do, Ho, Wo, Dr, Hr, Wr;
  int relu;
  int ntx, nty, ntiles;       // tile grid: x tiles, y tiles, total = B*Do*nty*ntx (x8 classes for deconv)
};

// XCD-aware persistent tile order: workgroups are dealt round-robin over the 8
// XCDs, so worker (xcd = id % 8, slot = id / 8) walks a contiguous eighth of the
// tile space -- neighbouring tiles (shared halos) meet in one XCD's L2.  Speed only.
__device__ __forceinline__ int first_tile(int ntiles, int& step, int& end) {
  const int id = blockIdx.x, G = gridDim.x;
  if ((G & 7) != 0 || ntiles < 64) { step = G; end = ntiles; return id; }
  const int xcd = id & 7, slot = id >> 3;
  const int per = (ntiles + 7) >> 3;
  const int lo = xcd * per;
  end = min(ntiles, lo + per);
  step = G >> 3;
  return lo + slot;
}

// Epilogue of one 32x32 accumulator tile: rows = 32 consecutive x of one output row (stride
// XS voxels in memory: 1 for convolutions, 2 for a transposed-conv parity class), columns =
// 32 output channels on the lanes.  y = acc*sc + sh (ReLU?) (+ skip) (ReLU?).
// Interior tiles (wave-uniform test) take a branch-free path with all 16 skip loads in flight
// and compile-time address offsets; edge tiles fall back to per-element guards.
template <int COUT, int XS>
__device__ __forceinline__ void store_tile(const f32x16& acc, float sc, float sh, int relu,
                                           float* __restrict__ yrow, const float* __restrict__ rrow,
                                           int h, int xbase, int xlimit) {
  // yrow / rrow point at voxel (row start, x = xbase-th output column), channel co of this lane
  constexpr int VS = COUT * XS;                       // floats between consecutive tile rows
  float* yp = yrow + (long)(4 * h) * VS;
  const float* rp = rrow ? rrow + (long)(4 * h) * VS : nullptr;
  if (xbase + 31 * XS < xlimit) {                     // whole tile inside the row
    float r[16];
    if (rp) {
#pragma unroll
      for (int i = 0; i < 16; ++i) r[i] = rp[((i & 3) + 8 * (i >> 2)) * VS];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = acc[i] * sc + sh;
      if (relu == 2) v = fmaxf(v, 0.f);
      if (rp) v += r[i];
      if (relu == 1) v = fmaxf(v, 0.f);
      yp[((i & 3) + 8 * (i >> 2)) * VS] = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      if (xbase + row * XS < xlimit) {
        float v = acc[i] * sc + sh;
        if (relu == 2) v = fmaxf(v, 0.f);
        if (rp) v += rp[((i & 3) + 8 * (i >> 2)) * VS];
        if (relu == 1) v = fmaxf(v, 0.f);
        yp[((i & 3) + 8 * (i >> 2)) * VS] = v;
      }
    }
  }
}

// ----------------------------------------------------------------------------
// Conv(KZ x KXY x KXY, "same" padding, dilation DIL, stride S), Cout = 32*NT, Cin = multiple
// of CK.  KZ = 3, KXY = 3: the 3-D trunk.  KZ = 1: the 2-D feature towers (3x3, 3x3
// dilated, 1x1) on (B,1,H,W,C) views of NHWC maps -- SURVEY.md section 8f-1.
// ----------------------------------------------------------------------------
template <int S, int NT, int TM, int CK, int KZ = 3, int KXY = 3, int DIL = 1>
__global__ __launch_bounds__(NTHREADS, 2) void conv3d_mfma_kernel(ConvParams p) {
  using G = Geo<S, KZ, KXY, DIL>;
  constexpr int NTAP = G::NTAP;
  constexpr int TY = 4 * TM;
  constexpr int IY = G::IY(TY), IX = G::IX, XP = G::XP, IZ = G::IZ;
  constexpr int NQ = CK / 4;                    // 16-B slots per voxel per chunk
  constexpr int NG = CK / 8;                    // 8-channel MFMA groups per chunk
  constexpr int NE = IZ * IY * IX * NQ;         // staged 16-B elements per chunk
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  constexpr int ROW = NQ * XP;                  // elements per (z, y) row
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];   // [IZ][IY][NQ][XP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nch = p.Cin / CK;
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(p.w);

  int step, end;
  int t = first_tile(p.ntiles, step, end);
  if (t >= end) return;

  // ---- staging: decode tile id, issue the global loads of one chunk ----------
  f32x4 pf[NPF];
  int tb, tz, ty0, tx0;
  auto decode = [&](int id) {
    tx0 = (id % p.ntx) * 32; id /= p.ntx;
    ty0 = (id % p.nty) * TY; id /= p.nty;
    tz = id % p.Do; tb = id / p.Do;
  };
  auto prefetch = [&](int id, int ck) {
    int xb = (id % p.ntx) * 32 * S - G::PADXY; id /= p.ntx;
    int yb = (id % p.nty) * TY * S - G::PADXY; id /= p.nty;
    int zb = (id % p.Do) * S - G::PADZ; const int b = id / p.Do;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = tid + k * NTHREADS;
      const int q = e % NQ, v = e / NQ;
      const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
      const int zi = zb + zz, yi = yb + yy, xi = xb + xx;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (e < NE && zi >= 0 && zi < p.Di && yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi)
        val = *reinterpret_cast<const f32x4*>(
            p.x + ((((long)b * p.Di + zi) * p.Hi + yi) * p.Wi + xi) * p.Cin + ck * CK + q * 4);
      pf[k] = val;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = tid + k * NTHREADS;
      const int q = e % NQ, v = e / NQ;
      const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
      if (e < NE) tile[(zz * IY + yy) * ROW + q * XP + G::xmap(xx)] = pf[k];
    }
  };

  f32x16 acc[TM][NT];
  const int lane_el = h * XP + r;               // this lane's slot within a (z,y) row
  int ck = 0;
  prefetch(t, 0);
  while (true) {
    __syncthreads();                            // every wave is done with the old chunk
    commit();
    __syncthreads();
    // next (tile, chunk) item -> registers; lands while this chunk is multiplied
    int nt_ = t, nck = ck + 1;
    if (nck == nch) { nck = 0; nt_ = t + step; }
    if (nt_ < end) prefetch(nt_, nck);

    if (ck == 0) {
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    }
    // ---- multiply: 27 taps x NG groups ------------------------------------------------
    // Software pipeline, pinned with sched_barrier(0) (left alone, hipcc sinks the loads
    // next to their first use and waits vmcnt(0) per item): B fragments are requested
    // AHEAD-1 items before use (>= 1.5k cycles of MFMA cover the L2 round trip), the next
    // item's A fragments one item before use; each MFMA cluster runs on operands that were
    // requested at least one full cluster earlier.
    constexpr int NITEM = NTAP * NG;
    constexpr int AHEAD0 = (TM * NT >= 4) ? 3 : ((TM * NT >= 2) ? 4 : 6);
    constexpr int AHEAD = AHEAD0 < NITEM ? AHEAD0 : NITEM;
    f32x4 bq[AHEAD][NT];
    f32x4 abuf[2][TM];
    const f32x4* wbase = wp + (long)ck * NG * NTAP * NT * 64 + lane;   // [g][tap][nt][lane]
    auto bload = [&](auto ic) {
      constexpr int item = decltype(ic)::value;
      constexpr int tap = item / NG, gi = item % NG;
#pragma unroll
      for (int n = 0; n < NT; ++n) bq[item % AHEAD][n] = wbase[((gi * NTAP + tap) * NT + n) * 64];
    };
    auto aload = [&](auto ic) {
      constexpr int item = decltype(ic)::value;
      constexpr int tap = item / NG, gi = item % NG;
      constexpr int dz = tap / (KXY * KXY), dy = ((tap / KXY) % KXY) * DIL, dx = (tap % KXY) * DIL;
      constexpr int xoff = (S == 1) ? dx : ((dx & 1) * G::XE + (dx >> 1));
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int yy = (wave * TM + m) * S + dy;
        abuf[item & 1][m] = tile[(dz * IY + yy) * ROW + (2 * gi) * XP + xoff + lane_el];
      }
    };
    static_for<0, AHEAD - 1>(bload);
    aload(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NITEM>([&](auto ic) {
      constexpr int item = decltype(ic)::value;
      if constexpr (item + AHEAD - 1 < NITEM) bload(std::integral_constant<int, item + AHEAD - 1>{});
      if constexpr (item + 1 < NITEM) aload(std::integral_constant<int, item + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f32x4 aa = abuf[item & 1][m];
          const f32x4 bb = bq[item % AHEAD][n];
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.x, bb.x, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.y, bb.y, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.z, bb.z, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.w, bb.w, acc[m][n], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- epilogue after the last chunk of a tile --------------------------------
    if (ck == nch - 1) {
      decode(t);
      constexpr int COUT = 32 * NT;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = n * 32 + r;
        const float sc = p.scale ? p.scale[co] : 1.f;
        const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          const int yo = ty0 + wave * TM + m;
          if (yo >= p.Ho) continue;                         // wave-uniform
          float* yrow = p.y + ((((long)tb * p.Do + tz) * p.Ho + yo) * p.Wo + tx0) * COUT + co;
          const float* rrow = p.res
              ? p.res + ((((long)tb * p.Dr + tz) * p.Hr + yo) * p.Wr + tx0) * COUT + co : nullptr;
          store_tile<COUT, 1>(acc[m][n], sc, sh, p.relu, yrow, rrow, h, tx0, p.Wo);
        }
      }
    }
    ck = nck; t = nt_;
    if (t >= end) break;
  }
}

// ----------------------------------------------------------------------------
// ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1), Cout = 32*NT.
//   out[o] += in[i] * w[k],  o = 2i - 1 + k   per dimension, so an output of parity
//   0 (o = 2m) has one tap (k=1, i=m) and of parity 1 (o = 2m+1) two taps
//   (k=2, i=m) and (k=0, i=m+1): 8 parity classes with 1..8 taps, 27 in all.
// A work item is (input-grid tile of 4 rows x 32 columns at depth m, z-parity pz).  Its four
// (py,px) classes are accumulated together: an A fragment read at input offset (iy,ix)
// feeds every class that has a tap there (4, 2, 2, 1 classes), so the tile is staged once
// per channel chunk for 9*(pz+1) tap-steps.  The step schedule is static; B fragments run
// three steps ahead; the next item's chunk is prefetched into registers while this one is
// multiplied -- the same pipeline as conv3d_mfma_kernel.
// ----------------------------------------------------------------------------
// Step schedule, ordered (input offset, channel group, class) so that one A fragment
// serves consecutive steps.  Offsets (iy,ix): (0,0) feeds classes {0,1,2,3}, (0,1) feeds
// {1,3}, (1,0) feeds {2,3}, (1,1) feeds {3}; class = py*2 + px.
struct DeconvStep { int iy, ix, gi, cls, fresh; };
__host__ __device__ constexpr DeconvStep deconv_step(int s, int NG);
// number of fresh (A-fragment-loading) steps before global step S2 (steps repeat per iz)
__host__ __device__ constexpr int deconv_fresh_before(int S2, int NG);
__host__ __device__ constexpr int deconv_aslot(int S2, int NG);
__host__ __device__ constexpr int deconv_next_fresh(int S2, int NG);
__host__ __device__ constexpr DeconvStep deconv_step(int s, int NG) {
  constexpr int n_off[4] = {4, 2, 2, 1};
  constexpr int cls_of[4][4] = {{0, 1, 2, 3}, {1, 3, 0, 0}, {2, 3, 0, 0}, {3, 0, 0, 0}};
  int o = 0, base = 0;
  while (s >= base + n_off[o] * NG) { base += n_off[o] * NG; ++o; }
  const int rel = s - base;
  return DeconvStep{o >> 1, o & 1, rel / n_off[o], cls_of[o][rel % n_off[o]],
                    (rel % n_off[o]) == 0};
}
__host__ __device__ constexpr int deconv_fresh_before(int S2, int NG) {
  int n = 0;
  for (int i = 0; i < S2; ++i) n += deconv_step(i % (9 * NG), NG).fresh;
  return n;
}
// LDS-read double buffer: the A fragment of step S2 lives in slot (index of its fresh step) & 1
__host__ __device__ constexpr int deconv_aslot(int S2, int NG) {
  return (deconv_fresh_before(S2 + 1, NG) - 1) & 1;
}
__host__ __device__ constexpr int deconv_next_fresh(int S2, int NG) {
  int i = S2 + 1;
  while (i < 2 * 9 * NG && !deconv_step(i % (9 * NG), NG).fresh) ++i;
  return i;
}

template <int NT, int CK>
__global__ __launch_bounds__(NTHREADS, 2) void deconv3d_mfma_kernel(ConvParams p) {
  constexpr int TY = 4;
  constexpr int IZ = 2, IY = TY + 1, IX = 33, XP = 34;
  constexpr int NQ = CK / 4, NG = CK / 8;
  constexpr int NE = IZ * IY * IX * NQ;
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  constexpr int ROW = NQ * XP;
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nch = p.Cin / CK;
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(p.w);
  const int lane_el = h * XP + r;

  int step, end;
  int t = first_tile(p.ntiles, step, end);
  if (t >= end) return;

  f32x4 pf[NPF];
  auto prefetch = [&](int id, int ck) {
    id >>= 1;                                            // drop the z-parity bit
    const int xb = (id % p.ntx) * 32; id /= p.ntx;
    const int yb = (id % p.nty) * TY; id /= p.nty;
    const int zb = id % p.Di; const int b = id / p.Di;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = tid + k * NTHREADS;
      const int q = e % NQ, v = e / NQ;
      const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
      const int zi = zb + zz, yi = yb + yy, xi = xb + xx;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (e < NE && zi < p.Di && yi < p.Hi && xi < p.Wi)
        val = *reinterpret_cast<const f32x4*>(
            p.x + ((((long)b * p.Di + zi) * p.Hi + yi) * p.Wi + xi) * p.Cin + ck * CK + q * 4);
      pf[k] = val;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = tid + k * NTHREADS;
      const int q = e % NQ, v = e / NQ;
      const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
      if (e < NE) tile[(zz * IY + yy) * ROW + q * XP + xx] = pf[k];
    }
  };

  f32x16 acc[4][NT];                                     // class = py*2 + px
  int ck = 0;
  prefetch(t, 0);
  while (true) {
    __syncthreads();
    commit();
    __syncthreads();
    int nt_ = t, nck = ck + 1;
    if (nck == nch) { nck = 0; nt_ = t + step; }
    if (nt_ < end) prefetch(nt_, nck);
    if (ck == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[c][n][i] = 0.f;
    }
    const int pz = t & 1;
    const f32x4* wbase = wp + (long)ck * NG * 27 * NT * 64 + lane;
    {
      // Both z taps of a z-odd item (iz = 0, 1) run as one pipelined sequence of
      // 2*NSTEP steps; a z-even item stops after NSTEP (wave-uniform branch).  Same pinned
      // software pipeline as the forward convolution.
      constexpr int NSTEP = 9 * NG;
      constexpr int AHEAD = (NT >= 2) ? 3 : 6;
      const int kz0 = pz ? 2 : 1;                        // iz = 0; iz = 1 (pz only) uses kz = 0
      f32x4 bq[AHEAD][NT];
      f32x4 abuf[2];
      auto bload = [&](auto sc) {
        constexpr int S2 = decltype(sc)::value;
        constexpr int iz = S2 / NSTEP, s = S2 % NSTEP;
        constexpr DeconvStep st = deconv_step(s, NG);
        constexpr int py = st.cls >> 1, px = st.cls & 1;
        constexpr int ky = py ? (st.iy ? 0 : 2) : 1, kx = px ? (st.ix ? 0 : 2) : 1;
        const int tap = ((iz ? 0 : kz0) * 3 + ky) * 3 + kx;
        constexpr int ring = S2 % AHEAD;
#pragma unroll
        for (int n = 0; n < NT; ++n) bq[ring][n] = wbase[((st.gi * 27 + tap) * NT + n) * 64];
      };
      auto aload = [&](auto sc) {                       // slot = parity of the fresh-read count
        constexpr int S2 = decltype(sc)::value;
        constexpr int iz = S2 / NSTEP, s = S2 % NSTEP;
        constexpr DeconvStep st = deconv_step(s, NG);
        constexpr int slot = deconv_aslot(S2, NG);
        abuf[slot] = tile[(iz * IY + wave + st.iy) * ROW + (2 * st.gi) * XP + st.ix + lane_el];
      };
      auto body = [&](auto sc) {
        constexpr int S2 = decltype(sc)::value;
        constexpr int s = S2 % NSTEP;
        constexpr DeconvStep st = deconv_step(s, NG);
        constexpr int c = st.cls;
        // prefetch: B for step S2+AHEAD-1, A for the next fresh step (one fresh step ahead)
        if constexpr (S2 + AHEAD - 1 < 2 * NSTEP) {
          if (S2 + AHEAD - 1 < NSTEP || pz) bload(std::integral_constant<int, S2 + AHEAD - 1>{});
        }
        if constexpr (st.fresh) {
          constexpr int nxt = deconv_next_fresh(S2, NG);
          if constexpr (nxt < 2 * NSTEP) {
            if (nxt < NSTEP || pz) aload(std::integral_constant<int, nxt>{});
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int slot = deconv_aslot(S2, NG);
        constexpr int ring = S2 % AHEAD;
        const f32x4 aa = abuf[slot];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f32x4 bb = bq[ring][n];
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.x, bb.x, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.y, bb.y, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.z, bb.z, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa.w, bb.w, acc[c][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      static_for<0, AHEAD - 1>(bload);                  // steps 0..AHEAD-2 are < NSTEP
      aload(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NSTEP>(body);
      if (pz) static_for<NSTEP, 2 * NSTEP>(body);
    }
    if (ck == nch - 1) {
      int id = t >> 1;
      const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
      const int ty0 = (id % p.nty) * TY; id /= p.nty;
      const int tz = id % p.Di; const int tb = id / p.Di;
      const int zo = 2 * tz + pz;
      const int ym = ty0 + wave;
      constexpr int COUT = 32 * NT;
      if (zo < p.Do && ym < p.Hi) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = n * 32 + r;
          const float sc = p.scale ? p.scale[co] : 1.f;
          const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yo = 2 * ym + (c >> 1);
            if (yo >= p.Ho) continue;                       // wave-uniform
            const int xo0 = 2 * tx0 + (c & 1);              // output x of tile row 0
            // input columns beyond Wi produce nothing: limit = min(Wo, 2*Wi)
            const int xlimit = min(p.Wo, 2 * p.Wi);
            float* yrow = p.y + ((((long)tb * p.Do + zo) * p.Ho + yo) * p.Wo + xo0) * COUT + co;
            const float* rrow = p.res
                ? p.res + ((((long)tb * p.Dr + zo) * p.Hr + yo) * p.Wr + xo0) * COUT + co : nullptr;
            store_tile<COUT, 2>(acc[c][n], sc, sh, p.relu, yrow, rrow, h, xo0, xlimit);
          }
        }
      }
    }
    ck = nck; t = nt_;
    if (t >= end) break;
  }
}

// ----------------------------------------------------------------------------
// Cout = 1 (PSMNet classif heads 32->1, GCNet l37): a 27*Cin-long dot product
// per voxel.  An MFMA tile would waste 31/32 of its columns, so this is a VALU
// kernel on the same LDS image: one thread per output voxel, weights read through
// the scalar cache (wave-uniform addresses).
// Weights packed as [tap][Cin].
// ----------------------------------------------------------------------------
// `w` is a kernel argument of its own (not a ConvParams field) and is read through the
// constant address space, so the wave-uniform weight reads compile to s_load_dwordx4.
// Same persistent pipeline as the MFMA kernels: the next (tile, chunk) is prefetched into
// registers while the current chunk is reduced; CK = 8 keeps the LDS image at 32 KB so that
// four workgroups share a CU and hide each other's barriers.
template <int CK>
__global__ __launch_bounds__(NTHREADS, 3) void conv3d_cout1_kernel(ConvParams p,
                                                                   const float* __restrict__ w) {
  constexpr int TY = 8;
  constexpr int IY = TY + 2, IX = 34, XP = 34, IZ = 3;
  constexpr int NQ = CK / 4;
  constexpr int NE = IZ * IY * IX * NQ;
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  constexpr int ROW = NQ * XP;
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];
  typedef const float __attribute__((address_space(4))) cfloat;
  const int tid = threadIdx.x;
  const int r = tid & 31, ty = tid >> 5;
  const int nch = p.Cin / CK;
  int t = blockIdx.x;
  const int step = gridDim.x, end = p.ntiles;
  if (t >= end) return;

  f32x4 pf[NPF];
  auto prefetch = [&](int id, int ck) {
    const int xb = (id % p.ntx) * 32 - 1; id /= p.ntx;
    const int yb = (id % p.nty) * TY - 1; id /= p.nty;
    const int zb = (id % p.Do) - 1; const int b = id / p.Do;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = tid + k * NTHREADS;
      const int q = e % NQ, v = e / NQ;
      const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
      const int zi = zb + zz, yi = yb + yy, xi = xb + xx;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (e < NE && zi >= 0 && zi < p.Di && yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi)
        val = *reinterpret_cast<const f32x4*>(
            p.x + ((((long)b * p.Di + zi) * p.Hi + yi) * p.Wi + xi) * p.Cin + ck * CK + q * 4);
      pf[k] = val;
    }
  };
  float acc = 0.f;
  int ck = 0;
  prefetch(t, 0);
  while (true) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = tid + k * NTHREADS;
      const int q = e % NQ, v = e / NQ;
      const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
      if (e < NE) tile[(zz * IY + yy) * ROW + q * XP + xx] = pf[k];
    }
    __syncthreads();
    int nt_ = t, nck = ck + 1;
    if (nck == nch) { nck = 0; nt_ = t + step; }
    if (nt_ < end) prefetch(nt_, nck);
    if (ck == 0) acc = 0.f;
    // Weights and activations both count on lgkmcnt, and scalar loads return out of order,
    // so mixing s_load with ds_read forces lgkmcnt(0) at every use.  Two pinned phases per
    // z-tap instead: 9 x s_load_dwordx8 (72 weights into SGPRs), then the LDS reads + FMAs.
    cfloat* wc = (cfloat*)(w) + ck * CK;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      float wv[9][CK];
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int c = 0; c < CK; ++c) wv[t9][c] = wc[(dz * 9 + t9) * p.Cin + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int dy = t9 / 3, dx = t9 % 3;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 a = tile[(dz * IY + ty + dy) * ROW + q * XP + r + dx];
          acc = fmaf(a.x, wv[t9][4 * q + 0], acc); acc = fmaf(a.y, wv[t9][4 * q + 1], acc);
          acc = fmaf(a.z, wv[t9][4 * q + 2], acc); acc = fmaf(a.w, wv[t9][4 * q + 3], acc);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ck == nch - 1) {
      int id = t;
      const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
      const int ty0 = (id % p.nty) * TY; id /= p.nty;
      const int tz = id % p.Do; const int tb = id / p.Do;
      const int yo = ty0 + ty, xo = tx0 + r;
      if (yo < p.Ho && xo < p.Wo) {
        float v = acc * (p.scale ? p.scale[0] : 1.f) + (p.shift ? p.shift[0] : 0.f);
        if (p.relu == 2) v = fmaxf(v, 0.f);
        if (p.res) v += p.res[(((long)tb * p.Dr + tz) * p.Hr + yo) * p.Wr + xo];
        if (p.relu == 1) v = fmaxf(v, 0.f);
        p.y[(((long)tb * p.Do + tz) * p.Ho + yo) * p.Wo + xo] = v;
      }
    }
    ck = nck; t = nt_;
    if (t >= end) break;
  }
}

// ConvTranspose3d(k3,s2,p1,op1) to one channel (GCNet l37, models/gcnet.py:63,100):
// thread per output voxel, gathers its 1..8 input voxels straight from L2.
__global__ __launch_bounds__(NTHREADS) void deconv3d_cout1_kernel(ConvParams p) {
  const int xo = blockIdx.x * NTHREADS + threadIdx.x;
  const int yo = blockIdx.y;
  const int zo = blockIdx.z % p.Do, b = blockIdx.z / p.Do;
  if (xo >= p.Wo) return;
  float acc = 0.f;
  const int pz = zo & 1, py = yo & 1, px = xo & 1;
  const int mz = zo >> 1, my = yo >> 1, mx = xo >> 1;
  for (int iz = 0; iz <= pz; ++iz) {
    const int zi = mz + iz, kz = pz ? (iz ? 0 : 2) : 1;
    if (zi >= p.Di) continue;
    for (int iy = 0; iy <= py; ++iy) {
      const int yi = my + iy, ky = py ? (iy ? 0 : 2) : 1;
      if (yi >= p.Hi) continue;
      for (int ix = 0; ix <= px; ++ix) {
        const int xi = mx + ix, kx = px ? (ix ? 0 : 2) : 1;
        if (xi >= p.Wi) continue;
        const int tap = (kz * 3 + ky) * 3 + kx;
        const f32x4* a = reinterpret_cast<const f32x4*>(
            p.x + ((((long)b * p.Di + zi) * p.Hi + yi) * p.Wi + xi) * p.Cin);
        const f32x4* w = reinterpret_cast<const f32x4*>(p.w + (long)tap * p.Cin);
        for (int c = 0; c < p.Cin / 4; ++c) {
          const f32x4 av = a[c], wv = w[c];
          acc = fmaf(av.x, wv.x, acc); acc = fmaf(av.y, wv.y, acc);
          acc = fmaf(av.z, wv.z, acc); acc = fmaf(av.w, wv.w, acc);
        }
      }
    }
  }
  float v = acc * (p.scale ? p.scale[0] : 1.f) + (p.shift ? p.shift[0] : 0.f);
  if (p.relu == 2) v = fmaxf(v, 0.f);
  if (p.res) v += p.res[(((long)b * p.Dr + zo) * p.Hr + yo) * p.Wr + xo];
  if (p.relu == 1) v = fmaxf(v, 0.f);
  p.y[(((long)b * p.Do + zo) * p.Ho + yo) * p.Wo + xo] = v;
}

// ----------------------------------------------------------------------------
// Weight packing (once per layer): torch layout -> MFMA B-fragment order.
// ----------------------------------------------------------------------------
// ntaps = 27 (3x3x3), 9 (3x3) or 1 (1x1).  cin_src <= Cin: source input channels; channels
// beyond it are packed as zeros (PSMNet's first convolution has 3, staged as 16).
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin,
                                    int Cout, int transposed, int ntaps, int cin_src) {
  const long n = (long)Cin * Cout * ntaps;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  int cin, cout, tap;
  if (Cout == 1) {                       // [tap][cin]
    cin = idx % Cin; tap = idx / Cin; cout = 0;
  } else {                               // [cin/8][tap][cout/32][lane][4]
    const int NT = Cout / 32;
    long i = idx;
    const int j = i & 3; i >>= 2;
    const int lane = i & 63; i >>= 6;
    const int n_ = i % NT; i /= NT;
    tap = i % ntaps; const int g = i / ntaps;
    cin = 8 * g + 4 * (lane >> 5) + j;
    cout = 32 * n_ + (lane & 31);
  }
  const long src = transposed ? (((long)cin * Cout + cout) * ntaps + tap)
                              : (((long)cout * cin_src + cin) * ntaps + tap);
  out[idx] = (cin < cin_src) ? w[src] : 0.f;
}

template <typename K>
int launch_tiles(K kernel, const ConvParams& p, size_t lds, hipStream_t s, int max_blocks) {
  if (lds > 64 * 1024) {
    static thread_local const void* configured[48];
    static thread_local int nconf = 0;
    bool seen = false;
    for (int i = 0; i < nconf; ++i) seen |= (configured[i] == (const void*)kernel);
    if (!seen) {
      if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return DSM_ERR_LAUNCH;
      if (nconf < 48) configured[nconf++] = (const void*)kernel;
    }
  }
  int blocks = p.ntiles < max_blocks ? p.ntiles : max_blocks;
  if (blocks >= 8) blocks &= ~7;                 // whole rounds over the 8 XCDs
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(NTHREADS), lds, s, p);
  return dsm_launch_status();
}

template <int S, int NT, int TM, int CK, int KZ = 3, int KXY = 3, int DIL = 1>
int run_conv(ConvParams p, hipStream_t s) {
  using G = Geo<S, KZ, KXY, DIL>;
  constexpr int TY = 4 * TM;
  p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, TY);
  const long nt = (long)p.B * p.Do * p.nty * p.ntx;
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  const size_t lds = (size_t)G::IZ * G::IY(TY) * (CK / 4) * G::XP * 16;
  return launch_tiles(conv3d_mfma_kernel<S, NT, TM, CK, KZ, KXY, DIL>, p, lds, s, 512);
}

template <int NT, int CK>
int run_deconv(ConvParams p, hipStream_t s) {
  constexpr int TY = 4;
  p.ntx = dsm_cdiv(p.Wi, 32); p.nty = dsm_cdiv(p.Hi, TY);
  const long nt = (long)p.B * p.Di * p.nty * p.ntx * 2;       // x2: z-parity
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  const size_t lds = (size_t)2 * (TY + 1) * (CK / 4) * 34 * 16;
  return launch_tiles(deconv3d_mfma_kernel<NT, CK>, p, lds, s, 512);
}

}  // namespace

extern "C" size_t dsm_conv3d_packed_weight_bytes(int Cin, int Cout, int transposed) {
  (void)transposed;
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)Cin * Cout * 27 * sizeof(float);
}

extern "C" int dsm_conv3d_pack_weights(const void* w_torch, void* w_packed, int Cin, int Cout,
                                       int transposed, dsm_stream_t stream) {
  DSM_REQUIRE(w_torch && w_packed && w_torch != w_packed, DSM_ERR_ARG);
  DSM_REQUIRE(Cin > 0 && Cout > 0, DSM_ERR_ARG);
  DSM_REQUIRE(Cin % 8 == 0 && (Cout == 1 || Cout % 32 == 0), DSM_ERR_UNSUPPORTED);
  const long n = (long)Cin * Cout * 27;
  dsm_clear_stale_error();
  hipLaunchKernelGGL(pack_weights_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)w_torch, (float*)w_packed, Cin, Cout,
                     transposed, 27, Cin);
  return dsm_launch_status();
}

extern "C" int dsm_conv_pack_weights(const void* w_torch, void* w_packed, int Cin_src, int Cin,
                                     int Cout, int kd, int k, dsm_stream_t stream) {
  DSM_REQUIRE(w_torch && w_packed && w_torch != w_packed, DSM_ERR_ARG);
  DSM_REQUIRE(Cin_src > 0 && Cin >= Cin_src && Cout > 0, DSM_ERR_ARG);
  DSM_REQUIRE((kd == 1 || kd == 3) && (k == 1 || k == 3), DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0, DSM_ERR_UNSUPPORTED);
  const int ntaps = kd * k * k;
  const long n = (long)Cin * Cout * ntaps;
  dsm_clear_stale_error();
  hipLaunchKernelGGL(pack_weights_kernel, dim3(dsm_cdiv(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)w_torch, (float*)w_packed, Cin, Cout, 0,
                     ntaps, Cin_src);
  return dsm_launch_status();
}

namespace {
// One place decides the kernel variant; dsm_conv3d_fwd launches it, dsm_conv3d_plan names it.
struct Plan { int kind; int S, NT, TM, CK; int KZ, K, DIL; };   // kind: 0 conv, 1 deconv, 2 conv cout1, 3 deconv cout1

int make_plan(const dsm_conv3d_args* a, Plan* pl) {
  DSM_REQUIRE(a && a->x && a->w_packed && a->y, DSM_ERR_ARG);
  DSM_REQUIRE(a->B > 0 && a->Cin > 0 && a->Cout > 0, DSM_ERR_ARG);
  DSM_REQUIRE(a->Di > 0 && a->Hi > 0 && a->Wi > 0 && a->Do > 0 && a->Ho > 0 && a->Wo > 0,
              DSM_ERR_ARG);
  DSM_REQUIRE(a->stride == 1 || a->stride == 2, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(!a->transposed || a->stride == 2, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(a->relu >= 0 && a->relu <= 2, DSM_ERR_ARG);
  DSM_REQUIRE(a->Cin % 16 == 0, DSM_ERR_UNSUPPORTED);   // chunk sizes 8 and 16 both divide it
  DSM_REQUIRE(dsm_aligned16(a->x) && dsm_aligned16(a->w_packed) && dsm_aligned16(a->y),
              DSM_ERR_ALIGN);
  const int kd = a->kd ? a->kd : 3, k = a->k ? a->k : 3, dil = a->dil ? a->dil : 1;
  DSM_REQUIRE((kd == 1 || kd == 3) && (k == 1 || k == 3) && (dil == 1 || dil == 2),
              DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(kd == 3 ? (k == 3 && dil == 1) : (!a->transposed && a->Di == 1 && a->Cout != 1),
              DSM_ERR_UNSUPPORTED);
  // natural output size; the caller may ask for a smaller corner (crop-add), never more
  const int nD = a->transposed ? 2 * a->Di : (a->Di - 1) / a->stride + 1;
  const int nH = a->transposed ? 2 * a->Hi : (a->Hi - 1) / a->stride + 1;
  const int nW = a->transposed ? 2 * a->Wi : (a->Wi - 1) / a->stride + 1;
  DSM_REQUIRE(a->Do <= nD && a->Ho <= nH && a->Wo <= nW, DSM_ERR_ARG);
  if (a->residual)
    DSM_REQUIRE(a->Dr >= a->Do && a->Hr >= a->Ho && a->Wr >= a->Wo, DSM_ERR_ARG);
  if (a->Cout == 1) {
    if (a->transposed) {
      DSM_REQUIRE(a->Ho <= 65535 && (long)a->B * a->Do <= 65535, DSM_ERR_UNSUPPORTED);
      *pl = Plan{3, 2, 0, 0, 0, 3, 3, 1};
    } else {
      DSM_REQUIRE(a->stride == 1, DSM_ERR_UNSUPPORTED);
      *pl = Plan{2, 1, 0, 0, 8, 3, 3, 1};
    }
    return DSM_OK;
  }
  DSM_REQUIRE(a->Cout == 32 || a->Cout == 64 || a->Cout == 128, DSM_ERR_UNSUPPORTED);
  const int NT = a->Cout / 32;
  // Tile height: 8 rows (TM = 2) when that still gives every CU two workgroups of work,
  // else 4 rows.  Stride 2 stages 8 channels per chunk so that two workgroups fit a CU.
  if (a->transposed) {
    DSM_REQUIRE(NT <= 2, DSM_ERR_UNSUPPORTED);   // 4 classes x NT accumulators must fit 256 VGPRs
    *pl = Plan{1, 2, NT, 1, 16, 3, 3, 1};
    return DSM_OK;
  }
  const bool big = (long)a->B * a->Do * dsm_cdiv(a->Ho, 8) * dsm_cdiv(a->Wo, 32) >= 1024;
  if (kd == 1) {                                   // 2-D towers: one staged slice, 16-channel chunks
    const int TM = (big && NT == 1 && a->stride == 1 && k == 3 && dil == 1) ? 2 : 1;
    *pl = Plan{0, a->stride, NT, TM, 16, 1, k, dil};
    return DSM_OK;
  }
  if (a->stride == 1) {
    const int TM = (big && NT <= 2) ? 2 : 1;
    *pl = Plan{0, 1, NT, TM, (NT == 2 && TM == 2) ? 8 : 16, 3, 3, 1};   // <1,2,2,16> would spill
  } else {
    *pl = Plan{0, 2, NT, 1, 8, 3, 3, 1};
  }
  return DSM_OK;
}
}  // namespace

extern "C" int dsm_conv3d_plan(const dsm_conv3d_args* a, char* buf, int len) {
  DSM_REQUIRE(buf && len > 0, DSM_ERR_ARG);
  Plan pl;
  int rc = make_plan(a, &pl);
  if (rc != DSM_OK) { buf[0] = 0; return rc; }
  switch (pl.kind) {
    case 0:
      if (pl.KZ == 3) snprintf(buf, len, "conv3d_mfma_kernel<S=%d,NT=%d,TM=%d,CK=%d>", pl.S, pl.NT, pl.TM, pl.CK);
      else snprintf(buf, len, "conv2d_mfma_kernel<S=%d,NT=%d,TM=%d,K=%d,DIL=%d>", pl.S, pl.NT, pl.TM, pl.K, pl.DIL);
      break;
    case 1: snprintf(buf, len, "deconv3d_mfma_kernel<NT=%d,CK=%d>", pl.NT, pl.CK); break;
    case 2: snprintf(buf, len, "conv3d_cout1_kernel<CK=%d>", pl.CK); break;
    default: snprintf(buf, len, "deconv3d_cout1_kernel"); break;
  }
  return DSM_OK;
}

extern "C" int dsm_conv3d_fwd(const dsm_conv3d_args* a, dsm_stream_t stream) {
  Plan pl;
  int rc = make_plan(a, &pl);
  if (rc != DSM_OK) return rc;
  ConvParams p;
  p.x = (const float*)a->x; p.w = (const float*)a->w_packed; p.scale = a->scale;
  p.shift = a->shift; p.res = (const float*)a->residual; p.y = (float*)a->y;
  p.B = a->B; p.Cin = a->Cin; p.Cout = a->Cout;
  p.Di = a->Di; p.Hi = a->Hi; p.Wi = a->Wi; p.Do = a->Do; p.Ho = a->Ho; p.Wo = a->Wo;
  p.Dr = a->Dr; p.Hr = a->Hr; p.Wr = a->Wr; p.relu = a->relu;
  p.ntx = p.nty = p.ntiles = 0;
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  if (pl.kind == 3) {
    dim3 grid(dsm_cdiv(a->Wo, NTHREADS), a->Ho, a->B * a->Do);
    hipLaunchKernelGGL(deconv3d_cout1_kernel, grid, dim3(NTHREADS), 0, s, p);
    return dsm_launch_status();
  }
  if (pl.kind == 2) {
    p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, 8);
    p.ntiles = p.B * p.Do * p.nty * p.ntx;
    const size_t lds = (size_t)3 * 10 * 2 * 34 * 16;              // CK = 8: 32.6 KB
    const int blocks = p.ntiles < 1024 ? p.ntiles : 1024;          // 4 workgroups per CU
    hipLaunchKernelGGL(conv3d_cout1_kernel<8>, dim3(blocks), dim3(NTHREADS), lds, s, p, p.w);
    return dsm_launch_status();
  }
#define DSM_CASE2D(S_, NT_, TM_, K_, DIL_) \
  if (pl.kind == 0 && pl.KZ == 1 && pl.S == S_ && pl.NT == NT_ && pl.TM == TM_ && pl.K == K_ && \
      pl.DIL == DIL_) return run_conv<S_, NT_, TM_, 16, 1, K_, DIL_>(p, s)
  DSM_CASE2D(1, 1, 1, 3, 1); DSM_CASE2D(1, 1, 2, 3, 1); DSM_CASE2D(1, 2, 1, 3, 1);
  DSM_CASE2D(1, 4, 1, 3, 1); DSM_CASE2D(1, 4, 1, 3, 2);
  DSM_CASE2D(2, 1, 1, 3, 1); DSM_CASE2D(2, 2, 1, 3, 1);
  DSM_CASE2D(1, 1, 1, 1, 1); DSM_CASE2D(1, 4, 1, 1, 1); DSM_CASE2D(2, 2, 1, 1, 1);
#undef DSM_CASE2D
  if (pl.KZ == 1) return DSM_ERR_UNSUPPORTED;
#define DSM_CASE(KIND, S_, NT_, TM_, CK_, CALL) \
  if (pl.kind == KIND && pl.S == S_ && pl.NT == NT_ && pl.TM == TM_ && pl.CK == CK_) return CALL
  DSM_CASE(1, 2, 1, 1, 16, (run_deconv<1, 16>(p, s)));
  DSM_CASE(1, 2, 2, 1, 16, (run_deconv<2, 16>(p, s)));
  DSM_CASE(0, 1, 1, 2, 16, (run_conv<1, 1, 2, 16>(p, s)));
  DSM_CASE(0, 1, 1, 1, 16, (run_conv<1, 1, 1, 16>(p, s)));
  DSM_CASE(0, 1, 2, 2, 8, (run_conv<1, 2, 2, 8>(p, s)));
  DSM_CASE(0, 1, 2, 1, 16, (run_conv<1, 2, 1, 16>(p, s)));
  DSM_CASE(0, 1, 4, 1, 16, (run_conv<1, 4, 1, 16>(p, s)));
  DSM_CASE(0, 2, 1, 1, 8, (run_conv<2, 1, 1, 8>(p, s)));
  DSM_CASE(0, 2, 2, 1, 8, (run_conv<2, 2, 1, 8>(p, s)));
  DSM_CASE(0, 2, 4, 1, 8, (run_conv<2, 4, 1, 8>(p, s)));
#undef DSM_CASE
  return DSM_ERR_UNSUPPORTED;
}
