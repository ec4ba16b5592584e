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
//  * MFMA operands are (weights, activations), so a lane owns one voxel and 4-channel
//    register quads: epilogue y = acc*scale + shift (+ skip) (ReLU) moves 16 B per lane.
//
// K order inside an 8-channel group: MFMA step j multiplies channels
// {8g + j (lanes 0-31), 8g + 4 + j (lanes 32-63)}; A and B fragments agree on it.
#include "conv_common.hpp"

#ifdef DSM_STAMPS
// Diagnostic build only (python dsmnet_amd/csrc/build.py --stamps): per-phase cycle totals of
// wave 0 of every workgroup, read back with dsm_debug_read_stamps.  Never in the shipped build.
__device__ unsigned long long dsm_stamp_buf[8 * 1024];
#define DSM_STAMP(slot)                                                                  \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    unsigned long long _t;                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (threadIdx.x == 0 && blockIdx.x < 1024) dsm_stamp_buf[blockIdx.x * 8 + (slot)] += _t - _tprev; \
    _tprev = _t;                                                                         \
  } while (0)
#define DSM_STAMP_INIT()                                                                 \
  unsigned long long _tprev;                                                             \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_tprev)::"memory")
extern "C" int dsm_debug_read_stamps(unsigned long long* host, int zero) {
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(dsm_stamp_buf), sizeof(unsigned long long) * 8 * 1024) != hipSuccess) return -3;
  if (zero) {
    static unsigned long long z[8 * 1024];
    if (hipMemcpyToSymbol(HIP_SYMBOL(dsm_stamp_buf), z, sizeof(z)) != hipSuccess) return -3;
  }
  return 0;
}
#else
#define DSM_STAMP(slot) do {} while (0)
#define DSM_STAMP_INIT() do {} while (0)
#endif

namespace {


template <int NPF, int NE, int NQ, int IX, int IY>
__device__ __forceinline__ void stage_offsets(unsigned (&off)[NPF], int tid, int Hi, int Wi,
                                              int Cin) {
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = min(tid + k * NTHREADS, NE - 1);
    const int q = e % NQ, v = e / NQ;
    const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
    off[k] = 4u * (unsigned)(((zz * Hi + yy) * Wi + xx) * Cin + 4 * q);      // bytes
  }
}

struct StageBox {
  const float* base;      // address of the box origin (may lie outside the volume at the edges)
  unsigned sbase;         // its byte offset from the tensor start (valid when `interior`)
  int zb, yb, xb;         // box origin in input coordinates
  int Di, Hi, Wi;
  bool interior;          // the whole box lies inside the volume (wave-uniform)
  bool active;            // there is a next item at all
};

__device__ __forceinline__ StageBox stage_box(const float* __restrict__ x, int b, int zb, int yb,
                                              int xb, int Di, int Hi, int Wi, int Cin, int c0,
                                              int IZ, int IY, int IX, bool active) {
  StageBox s;
  const long origin = ((((long)b * Di + zb) * Hi + yb) * Wi + xb) * Cin + c0;
  s.base = x + origin;
  s.sbase = (unsigned)(origin * 4);
  s.zb = zb; s.yb = yb; s.xb = xb; s.Di = Di; s.Hi = Hi; s.Wi = Wi;
  s.interior = active && zb >= 0 && zb + IZ <= Di && yb >= 0 && yb + IY <= Hi && xb >= 0 &&
               xb + IX <= Wi;
  s.active = active;
  return s;
}

// Element k of the box (k a compile-time constant).
template <int K, int NE, int NQ, int IX, int IY>
__device__ __forceinline__ f32x4 stage_load(const StageBox& s, __amdgpu_buffer_rsrc_t xrsrc,
                                            unsigned off, int tid) {
  constexpr bool tail = (K + 1) * NTHREADS > NE;       // the last slice may be partial
  if (s.interior && !tail) return buffer_load16(xrsrc, off, s.sbase);
  f32x4 val = {0.f, 0.f, 0.f, 0.f};
  const int e = tid + K * NTHREADS;
  bool ok = s.active && e < NE;
  if (!s.interior) {
    const int v = e / NQ;
    const int xx = v % IX, yy = (v / IX) % IY, zz = v / (IX * IY);
    ok = ok && (unsigned)(s.zb + zz) < (unsigned)s.Di && (unsigned)(s.yb + yy) < (unsigned)s.Hi &&
         (unsigned)(s.xb + xx) < (unsigned)s.Wi;
  }
  if (ok) val = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s.base) + off);
  return val;
}

template <int NPF, int NE, int NQ, int IX, int IY>
__device__ __forceinline__ void stage_prefetch(f32x4 (&pf)[NPF], const unsigned (&off)[NPF],
                                               const StageBox& s, __amdgpu_buffer_rsrc_t xrsrc,
                                               int tid) {
  static_for<0, NPF>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    pf[k] = stage_load<k, NE, NQ, IX, IY>(s, xrsrc, off[k], tid);
  });
}

template <int NPF, int NE>
__device__ __forceinline__ void stage_commit(f32x4* __restrict__ tile, const f32x4 (&pf)[NPF],
                                             int tid) {
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    if ((k + 1) * NTHREADS <= NE) tile[tid + k * NTHREADS] = pf[k];
    else if (tid + k * NTHREADS < NE) tile[tid + k * NTHREADS] = pf[k];
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
  constexpr int IY = G::IY(TY), IX = G::IX, IZ = G::IZ;
  constexpr int NQ = CK / 4;                    // 16-B slots per voxel per chunk
  constexpr int NG = CK / 8;                    // 8-channel MFMA groups per chunk
  constexpr int NE = IZ * IY * IX * NQ;         // staged 16-B elements per chunk
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];   // [IZ][IY][IX][NQ], element order

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nch = p.Cin / CK;

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
  unsigned goff[NPF];
  stage_offsets<NPF, NE, NQ, IX, IY>(goff, tid, p.Hi, p.Wi, p.Cin);
  const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.xbytes);
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  auto box_of = [&](int id, int ck, bool active) {
    const int xb = (id % p.ntx) * 32 * S - G::PADXY; id /= p.ntx;
    const int yb = (id % p.nty) * TY * S - G::PADXY; id /= p.nty;
    const int zb = (id % p.Do) * S - G::PADZ; const int b = id / p.Do;
    return stage_box(p.x, b, zb, yb, xb, p.Di, p.Hi, p.Wi, p.Cin, ck * CK, IZ, IY, IX, active);
  };
  auto commit = [&]() { stage_commit<NPF, NE>(tile, pf, tid); };

  f32x16 acc[TM][NT];
  float am = 0.f;                               // max |y| of this thread's outputs (y_amax)
  // folded BN of this lane's channels: kept in registers for the narrow variants (32 per
  // N-tile), re-read per tile where the accumulators need the registers
  constexpr bool KEEP_AFFINE = (NT * TM <= 2) && (NT == 1);
  Affine af[KEEP_AFFINE ? NT : 1];
  const int n0 = blockIdx.y * NT;               // first N-tile of this workgroup column (N-split)
  if constexpr (KEEP_AFFINE) {
#pragma unroll
    for (int n = 0; n < NT; ++n) af[n] = load_affine(p.scale, p.shift, (n0 + n) * 32 + 4 * h);
  }
  const int lane_el = r * S * NQ + h;           // this lane's element within an image row
  int ck = 0;
  {
    const StageBox first = box_of(t, 0, true);
    stage_prefetch<NPF, NE, NQ, IX, IY>(pf, goff, first, xrsrc, tid);
  }
  DSM_STAMP_INIT();
  while (true) {
    __syncthreads();                            // every wave is done with the old chunk
    DSM_STAMP(0);                               // [0] wait at barrier 1
    commit();
    DSM_STAMP(1);                               // [1] vmcnt wait for the prefetch + LDS commit
    __syncthreads();
    DSM_STAMP(2);                               // [2] barrier 2
    // next (tile, chunk): its NPF staged loads are issued one per item INSIDE the multiply
    // loop, in the shadow of this wave's own MFMAs -- as a separate phase they crawl whenever
    // the partner workgroup is streaming fp32 MFMAs (14k cycles instead of 3k; stamps).
    int nt_ = t, nck = ck + 1;
    if (nck == nch) { nck = 0; nt_ = t + step; }
    const StageBox nbox = box_of(nt_ < end ? nt_ : t, nck, nt_ < end);
    DSM_STAMP(3);                               // [3] next-box scalar setup
    if (ck == 0) {
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    }
    DSM_STAMP(6);                               // [6] accumulator reset

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
    // [g][tap][nt][lane]: wave-uniform base (SGPRs) + this lane's fixed 16-B slot, so that a
    // B-fragment load is `global_load_dwordx4 v, v_lane, s[base]` with no address VALU
    const unsigned wchunk = (unsigned)ck * (NG * NTAP * 64 * 16) * (unsigned)p.ntp +
                            (unsigned)n0 * (64 * 16);                    // bytes, wave-uniform
    const unsigned lane16 = lane * 16u;
    auto bload = [&](auto ic) {
      constexpr int item = decltype(ic)::value;
      constexpr int tap = item / NG, gi = item % NG;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bq[item % AHEAD][n] =
            buffer_load16(wrsrc, lane16, wchunk + (unsigned)(gi * NTAP + tap) * (64 * 16) * (unsigned)p.ntp + n * (64 * 16));
    };
    auto aload = [&](auto ic) {
      constexpr int item = decltype(ic)::value;
      constexpr int tap = item / NG, gi = item % NG;
      constexpr int dz = tap / (KXY * KXY), dy = ((tap / KXY) % KXY) * DIL, dx = (tap % KXY) * DIL;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int yy = (wave * TM + m) * S + dy;       // image element ((z*IY + y)*IX + x)*NQ + q
        abuf[item & 1][m] = tile[((dz * IY + yy) * IX + dx) * NQ + 2 * gi + lane_el];
      }
    };
    static_for<0, AHEAD - 1>(bload);
    aload(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    DSM_STAMP(7);                               // [7] B/A ring prologue issue
    static_for<0, NITEM>([&](auto ic) {
      constexpr int item = decltype(ic)::value;
      if constexpr (item + AHEAD - 1 < NITEM) bload(std::integral_constant<int, item + AHEAD - 1>{});
      if constexpr (item + 1 < NITEM) aload(std::integral_constant<int, item + 1>{});
      // staged loads of the next chunk, spread over the items (NPF <= NITEM for every variant
      // except the 1x1 kernels, whose remainder is issued with the last item)
      constexpr int PF_PER = (NPF + NITEM - 1) / NITEM;
      static_for<0, PF_PER>([&](auto jc) {
        constexpr int k = item * PF_PER + decltype(jc)::value;
        if constexpr (k < NPF) pf[k] = stage_load<k, NE, NQ, IX, IY>(nbox, xrsrc, goff[k], tid);
      });
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f32x4 aa = abuf[item & 1][m];
          const f32x4 bb = bq[item % AHEAD][n];
          // A operand = weights (rows: channels), B operand = activations (columns: voxels)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.x, aa.x, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.y, aa.y, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.z, aa.z, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.w, aa.w, acc[m][n], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    });
    DSM_STAMP(4);                               // [4] multiply (item loop)
    // ---- epilogue after the last chunk of a tile --------------------------------
    if (ck == nch - 1) {
      decode(t);
      const int COUT = 32 * p.ntp;                            // channels per voxel in memory
      const int xo = tx0 + r;                                 // this lane's output voxel
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int yo = ty0 + wave * TM + m;
        if (yo >= p.Ho || xo >= p.Wo) continue;
        const long vox = (((long)tb * p.Do + tz) * p.Ho + yo) * p.Wo + xo;
        const long rvox = (((long)tb * p.Dr + tz) * p.Hr + yo) * p.Wr + xo;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int cbase = (n0 + n) * 32 + 4 * h;
          if constexpr (KEEP_AFFINE) {
            store_tile<0>(acc[m][n], af[n], p.relu, p.y + vox * COUT + cbase,
                          p.res ? p.res + rvox * COUT + cbase : nullptr, am);
          } else {
            const Affine a1 = load_affine(p.scale, p.shift, cbase);
            store_tile<0>(acc[m][n], a1, p.relu, p.y + vox * COUT + cbase,
                          p.res ? p.res + rvox * COUT + cbase : nullptr, am);
          }
        }
      }
    }
    DSM_STAMP(5);                               // [5] epilogue
    ck = nck; t = nt_;
    if (t >= end) break;
  }
  flush_amax(p.y_amax, am, reinterpret_cast<float*>(tile));
}

#include "conv_split.hpp"   // conv_split_kernel, deconv_split_kernel, pack_weights_split_kernel
#include "conv_zs.hpp"      // conv_zs_kernel, pack_weights_zs_kernel

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
  constexpr int IZ = 2, IY = TY + 1, IX = 33;
  constexpr int NQ = CK / 4, NG = CK / 8;
  constexpr int NE = IZ * IY * IX * NQ;
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nch = p.Cin / CK;
  const int lane_el = r * NQ + h;                        // element within an image row

  int step, end;
  int t = first_tile(p.ntiles, step, end);
  if (t >= end) return;

  f32x4 pf[NPF];
  unsigned goff[NPF];
  stage_offsets<NPF, NE, NQ, IX, IY>(goff, tid, p.Hi, p.Wi, p.Cin);
  const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.xbytes);
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  auto box_of = [&](int id, int ck, bool active) {
    id >>= 1;                                            // drop the z-parity bit
    const int xb = (id % p.ntx) * 32; id /= p.ntx;
    const int yb = (id % p.nty) * TY; id /= p.nty;
    const int zb = id % p.Di; const int b = id / p.Di;
    return stage_box(p.x, b, zb, yb, xb, p.Di, p.Hi, p.Wi, p.Cin, ck * CK, IZ, IY, IX, active);
  };
  auto commit = [&]() { stage_commit<NPF, NE>(tile, pf, tid); };

  f32x16 acc[4][NT];                                     // class = py*2 + px
  float am = 0.f;
  constexpr bool KEEP_AFFINE = (NT == 1);
  Affine af[1];
  if constexpr (KEEP_AFFINE) af[0] = load_affine(p.scale, p.shift, 4 * h);
  int ck = 0;
  {
    const StageBox first = box_of(t, 0, true);
    stage_prefetch<NPF, NE, NQ, IX, IY>(pf, goff, first, xrsrc, tid);
  }
  while (true) {
    __syncthreads();
    commit();
    __syncthreads();
    int nt_ = t, nck = ck + 1;
    if (nck == nch) { nck = 0; nt_ = t + step; }
    const StageBox nbox = box_of(nt_ < end ? nt_ : t, nck, nt_ < end);   // staged inside the loop
    if (ck == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[c][n][i] = 0.f;
    }
    const int pz = t & 1;
    const unsigned wchunk = (unsigned)ck * (NG * 27 * NT * 64 * 16);      // bytes, wave-uniform
    const unsigned lane16 = lane * 16u;
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
        for (int n = 0; n < NT; ++n)
          bq[ring][n] = buffer_load16(wrsrc, lane16, wchunk + (((st.gi * 27 + tap) * NT + n) * 64) * 16);
      };
      auto aload = [&](auto sc) {                       // slot = parity of the fresh-read count
        constexpr int S2 = decltype(sc)::value;
        constexpr int iz = S2 / NSTEP, s = S2 % NSTEP;
        constexpr DeconvStep st = deconv_step(s, NG);
        constexpr int slot = deconv_aslot(S2, NG);
        abuf[slot] = tile[((iz * IY + wave + st.iy) * IX + st.ix) * NQ + 2 * st.gi + lane_el];
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
        if constexpr (S2 < NPF)                            // next chunk's staged loads, one per step
          pf[S2] = stage_load<S2, NE, NQ, IX, IY>(nbox, xrsrc, goff[S2], tid);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int slot = deconv_aslot(S2, NG);
        constexpr int ring = S2 % AHEAD;
        const f32x4 aa = abuf[slot];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f32x4 bb = bq[ring][n];
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.x, aa.x, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.y, aa.y, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.z, aa.z, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb.w, aa.w, acc[c][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      static_for<0, AHEAD - 1>(bload);                  // steps 0..AHEAD-2 are < NSTEP
      aload(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      static_assert(NPF <= NSTEP, "staged loads must fit the first z tap");
      static_for<0, NSTEP>(body);
      if (pz) static_for<NSTEP, 2 * NSTEP>(body);
    }
    if (ck == nch - 1) {
      int id = t >> 1;
      const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
      const int ty0 = (id % p.nty) * TY; id /= p.nty;
      const int tz = id % p.Di; const int tb = id / p.Di;
      const int zo = 2 * tz + pz;
      const int ym = ty0 + wave, xm = tx0 + r;               // this lane's input-grid position
      constexpr int COUT = 32 * NT;
      if (zo < p.Do && ym < p.Hi && xm < p.Wi) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int yo = 2 * ym + (c >> 1), xo = 2 * xm + (c & 1);
          if (yo >= p.Ho || xo >= p.Wo) continue;
          const long vox = (((long)tb * p.Do + zo) * p.Ho + yo) * p.Wo + xo;
          const long rvox = (((long)tb * p.Dr + zo) * p.Hr + yo) * p.Wr + xo;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int cbase = n * 32 + 4 * h;
            if constexpr (KEEP_AFFINE) {
              store_tile<COUT>(acc[c][n], af[0], p.relu, p.y + vox * COUT + cbase,
                               p.res ? p.res + rvox * COUT + cbase : nullptr, am);
            } else {
              const Affine a1 = load_affine(p.scale, p.shift, cbase);
              store_tile<COUT>(acc[c][n], a1, p.relu, p.y + vox * COUT + cbase,
                               p.res ? p.res + rvox * COUT + cbase : nullptr, am);
            }
          }
        }
      }
    }
    ck = nck; t = nt_;
    if (t >= end) break;
  }
  flush_amax(p.y_amax, am, reinterpret_cast<float*>(tile));
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
// Known limit (stamps + PMC, profiles/r01_d_pmc.md): bound by L1/TA line traffic, not by FMAs
// or LDS -- every 8-channel chunk touches 32 B of each 128-B voxel line and each input slice is
// re-read for three output planes.  Cin = 32 (every head of the reference) therefore takes
// conv3d_cout1_zslide_kernel below (83 us against 230 us at 48 x 96 x 320); this one remains
// the general-Cin form.
template <int CK>
__global__ __launch_bounds__(NTHREADS, 3) void conv3d_cout1_kernel(ConvParams p,
                                                                   const float* __restrict__ w) {
  constexpr int TY = 8;
  constexpr int IY = TY + 2, IX = 34, IZ = 3;
  constexpr int NQ = CK / 4;
  constexpr int NE = IZ * IY * IX * NQ;
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];
  typedef const float __attribute__((address_space(4))) cfloat;
  const int tid = threadIdx.x;
  const int r = tid & 31, ty = tid >> 5;
  const int nch = p.Cin / CK;
  int t = blockIdx.x;
  const int step = gridDim.x, end = p.ntiles;
  if (t >= end) return;

  f32x4 pf[NPF];
  unsigned goff[NPF];
  stage_offsets<NPF, NE, NQ, IX, IY>(goff, tid, p.Hi, p.Wi, p.Cin);
  const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.xbytes);
  auto prefetch = [&](int id, int ck) {
    const int xb = (id % p.ntx) * 32 - 1; id /= p.ntx;
    const int yb = (id % p.nty) * TY - 1; id /= p.nty;
    const int zb = (id % p.Do) - 1; const int b = id / p.Do;
    const StageBox box = stage_box(p.x, b, zb, yb, xb, p.Di, p.Hi, p.Wi, p.Cin, ck * CK, IZ, IY,
                                   IX, true);
    stage_prefetch<NPF, NE, NQ, IX, IY>(pf, goff, box, xrsrc, tid);
  };
  float acc = 0.f;
  int ck = 0;
  prefetch(t, 0);
  DSM_STAMP_INIT();
  while (true) {
    __syncthreads();
    DSM_STAMP(0);
    stage_commit<NPF, NE>(tile, pf, tid);
    DSM_STAMP(1);
    __syncthreads();
    DSM_STAMP(2);
    int nt_ = t, nck = ck + 1;
    if (nck == nch) { nck = 0; nt_ = t + step; }
    if (nt_ < end) prefetch(nt_, nck);
    DSM_STAMP(3);
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
          const f32x4 a = tile[((dz * IY + ty + dy) * IX + r + dx) * NQ + q];
          acc = fmaf(a.x, wv[t9][4 * q + 0], acc); acc = fmaf(a.y, wv[t9][4 * q + 1], acc);
          acc = fmaf(a.z, wv[t9][4 * q + 2], acc); acc = fmaf(a.w, wv[t9][4 * q + 3], acc);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    DSM_STAMP(4);
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
    DSM_STAMP(5);
    ck = nck; t = nt_;
    if (t >= end) break;
  }
}

// Cout = 1, Cin = 32, z-sliding form (PSMNet classif heads at full size).  A workgroup owns an
// 8 x 32 (y, x) column of `zseg` output planes and walks z: each input plane is staged ONCE as
// whole 128-B voxels (full-line loads) and feeds three rotating accumulators -- plane p is the
// dz = 0 / 1 / 2 tap plane of outputs p+1 / p / p-1.  Against the chunked kernel above this
// divides the L1/TA line traffic by ~12 (4 channel chunks x 3 planes) and the LDS reads by 3.
//  * LDS image [IY][IX] voxels at a pitch of 9 quads (144 B): the 16 lanes of a ds_read_b128
//    group then fall on 16 distinct slots of the 256-B bank row; element e = tid + 256k lands
//    at tid + (tid >> 3) + 288k, still an immediate offset.
//  * weights [tap][32] through the scalar cache in 24 pinned phases per plane (9 x
//    s_load_dwordx4 = 3 dz x 3 dx x 4 channels, then 3 LDS reads + 36 FMAs; 72 weights per
//    phase spilled scalars).  The pointer is
//    laundered per plane: the weight reads are loop-invariant, and hoisting 864 of them out of
//    the plane loop is what exhausted the scalar file in the first attempt at this form.
__global__ __launch_bounds__(NTHREADS, 3) void conv3d_cout1_zslide_kernel(
    ConvParams p, const float* __restrict__ w, int zseg) {
  constexpr int TY = 8, IY = TY + 2, IX = 34, NQ = 8, PITCH = 9;
  constexpr int NE = IY * IX * NQ;                       // 2720
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;    // 11
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];   // IY*IX*PITCH quads = 47.8 KB
  typedef const float __attribute__((address_space(4))) cfloat;
  typedef f32x4 __attribute__((address_space(4))) cquad;
  const int tid = threadIdx.x;
  const int r = tid & 31, ty = tid >> 5;
  int id = blockIdx.x;
  const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
  const int ty0 = (id % p.nty) * TY; id /= p.nty;
  const int nseg = (p.Do + zseg - 1) / zseg;
  const int z0 = (id % nseg) * zseg, b = id / nseg;
  const int z1 = min(z0 + zseg, p.Do);                   // output planes [z0, z1)

  f32x4 pf[NPF];
  unsigned goff[NPF];
  stage_offsets<NPF, NE, NQ, IX, IY>(goff, tid, p.Hi, p.Wi, p.Cin);
  const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.xbytes);
  auto prefetch = [&](int pz) {
    const StageBox box = stage_box(p.x, b, pz, ty0 - 1, tx0 - 1, p.Di, p.Hi, p.Wi, p.Cin, 0, 1, IY,
                                   IX, true);
    stage_prefetch<NPF, NE, NQ, IX, IY>(pf, goff, box, xrsrc, tid);
  };
  const int wbase = tid + (tid >> 3);                    // padded position of element `tid`
  const f32x4* rd = tile + (ty * IX + r) * PITCH;        // this thread's voxel, tap (0, 0)
  const int yo = ty0 + ty, xo = tx0 + r;
  const bool live = yo < p.Ho && xo < p.Wo;
  const float sc = p.scale ? p.scale[0] : 1.f, sh = p.shift ? p.shift[0] : 0.f;

  typedef float f32x2p __attribute__((ext_vector_type(2)));
  f32x2p a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f};   // outputs pz-1, pz, pz+1 (even | odd channels)
  prefetch(max(z0 - 1, 0));
  for (int pz = z0 - 1; pz <= z1; ++pz) {
    if (pz >= 0 && pz < p.Di) {                          // planes outside the volume are zero
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NPF; ++k)
        if ((k + 1) * NTHREADS <= NE || tid + k * NTHREADS < NE) tile[wbase + k * (NTHREADS / NQ) * PITCH] = pf[k];
      __syncthreads();
      if (pz + 1 <= z1 && pz + 1 < p.Di) prefetch(pz + 1);
      cfloat* wc = (cfloat*)w;
      asm volatile("" : "+s"(wc));                       // not loop-invariant as far as hipcc knows
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          f32x4 wv[3][3];                                // volatile: no merging across phases
#pragma unroll
          for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
              wv[dz][dx] = *(const volatile cquad*)(wc + ((dz * 3 + dy) * 3 + dx) * 32 + q * 4);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            // two channels per instruction (v_pk_fma_f32): even / odd channel sums, folded at the store
            const f32x4 a = rd[(dy * IX + dx) * PITCH + q];
            const f32x4 w0 = wv[0][dx], w1 = wv[1][dx], w2 = wv[2][dx];
            const f32x2p alo = {a.x, a.y}, ahi = {a.z, a.w};
            a2 = __builtin_elementwise_fma(alo, f32x2p{w0.x, w0.y}, a2); a2 = __builtin_elementwise_fma(ahi, f32x2p{w0.z, w0.w}, a2);
            a1 = __builtin_elementwise_fma(alo, f32x2p{w1.x, w1.y}, a1); a1 = __builtin_elementwise_fma(ahi, f32x2p{w1.z, w1.w}, a1);
            a0 = __builtin_elementwise_fma(alo, f32x2p{w2.x, w2.y}, a0); a0 = __builtin_elementwise_fma(ahi, f32x2p{w2.z, w2.w}, a0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int zo = pz - 1;                               // a0 has seen its three planes
    if (zo >= z0 && zo < z1 && live) {
      float v = (a0.x + a0.y) * sc + sh;
      if (p.relu == 2) v = fmaxf(v, 0.f);
      if (p.res) v += p.res[(((long)b * p.Dr + zo) * p.Hr + yo) * p.Wr + xo];
      if (p.relu == 1) v = fmaxf(v, 0.f);
      p.y[(((long)b * p.Do + zo) * p.Ho + yo) * p.Wo + xo] = v;
    }
    a0 = a1; a1 = a2; a2 = f32x2p{0.f, 0.f};
  }
}

// ConvTranspose3d(k3,s2,p1,op1) to one channel (GCNet l37, models/gcnet.py:63,100), Cin = 32.
// LDS-tiled: a workgroup stages the (2 x 5 x 33)-voxel input box of a 4 x 32 tile of input-grid
// positions (whole 128-B voxels, element order) plus the 27 x 32 weights; thread
// (position, z-parity) produces the four (y, x)-parity outputs of that z-parity: every staged
// voxel quad is read once per thread and used for all the taps it feeds; weights are LDS
// broadcasts.  (The first version -- one thread per output voxel gathering 1..8 voxels from
// L2 with 128-B-strided lanes -- ran at 0.23 TB/s: 2.2 ms for GCNet's 100 MB output.)
__global__ __launch_bounds__(NTHREADS, 3) void deconv3d_cout1_kernel(ConvParams p) {
  constexpr int TY = 4, IZ = 2, IY = TY + 1, IX = 33, NQ = 8;
  constexpr int NE = IZ * IY * IX * NQ;                // 2640
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;  // 11
  extern __shared__ __attribute__((aligned(16))) f32x4 tile[];   // the input box (weights: scalar loads)
  const int tid = threadIdx.x;
  int id = blockIdx.x;
  const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
  const int ty0 = (id % p.nty) * TY; id /= p.nty;
  const int tz = id % p.Di; const int b = id / p.Di;
  unsigned goff[NPF];
  stage_offsets<NPF, NE, NQ, IX, IY>(goff, tid, p.Hi, p.Wi, p.Cin);
  const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.xbytes);
  const StageBox box = stage_box(p.x, b, tz, ty0, tx0, p.Di, p.Hi, p.Wi, p.Cin, 0, IZ, IY, IX, true);
  f32x4 pf[NPF];
  stage_prefetch<NPF, NE, NQ, IX, IY>(pf, goff, box, xrsrc, tid);
  stage_commit<NPF, NE>(tile, pf, tid);
  __syncthreads();
  // thread -> (x position r, row ty, z-parity pz); pz is wave-uniform (waves 0-1: 0, 2-3: 1)
  const int r = tid & 31, ty = (tid >> 5) & 3, pz = tid >> 7;
  const int zo = 2 * tz + pz;
  if (zo >= p.Do) return;
  typedef float f32x2h __attribute__((ext_vector_type(2)));
  typedef const float __attribute__((address_space(4))) cfloat;
  typedef f32x4 __attribute__((address_space(4))) cquad;
  cfloat* wc = (cfloat*)p.w;
  asm volatile("" : "+s"(wc));                         // not loop-invariant as far as hipcc knows
  f32x2h acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // class = py*2 + px; (even | odd channels)
  for (int iz = 0; iz <= pz; ++iz) {
    const int kz = pz ? (iz ? 0 : 2) : 1;
#pragma unroll
    for (int o = 0; o < 4; ++o) {                      // input offset (iy, ix)
      const int iy = o >> 1, ix = o & 1;
      f32x4 v[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[q] = tile[((iz * IY + ty + iy) * IX + r + ix) * NQ + q];
#pragma unroll
      for (int c = 0; c < 4; ++c) {                    // classes fed by this offset
        const int py = c >> 1, px = c & 1;
        if (iy > py || ix > px) continue;
        const int ky = py ? (iy ? 0 : 2) : 1, kx = px ? (ix ? 0 : 2) : 1;
        // the tap's 32 weights as scalar loads (SGPR pairs), two channels per v_pk_fma_f32; even /
        // odd channel sums are folded at the store (r02: 408 -> ... us; was LDS-broadcast weights and
        // scalar FMAs)
        f32x4 w4[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) w4[q] = *(const volatile cquad*)(wc + ((kz * 3 + ky) * 3 + kx) * 32 + q * 4);
        f32x2h a = acc[c];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          a = __builtin_elementwise_fma(f32x2h{v[q].x, v[q].y}, f32x2h{w4[q].x, w4[q].y}, a);
          a = __builtin_elementwise_fma(f32x2h{v[q].z, v[q].w}, f32x2h{w4[q].z, w4[q].w}, a);
        }
        acc[c] = a;
        __builtin_amdgcn_sched_barrier(0);             // one tap's 32 SGPRs live at a time
      }
      // one offset's eight quads live at a time: left alone, hipcc hoists every LDS read of the
      // item to the top (256 VGPRs + AGPR spills, one wave per SIMD, 0.9 ms at GCNet's size)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int ym = ty0 + ty, xm = tx0 + r;
  if (ym >= p.Hi || xm >= p.Wi) return;
  const float sc = p.scale ? p.scale[0] : 1.f, sh = p.shift ? p.shift[0] : 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int yo = 2 * ym + (c >> 1), xo = 2 * xm + (c & 1);
    if (yo >= p.Ho || xo >= p.Wo) continue;
    float v = (acc[c].x + acc[c].y) * sc + sh;
    if (p.relu == 2) v = fmaxf(v, 0.f);
    if (p.res) v += p.res[(((long)b * p.Dr + zo) * p.Hr + yo) * p.Wr + xo];
    if (p.relu == 1) v = fmaxf(v, 0.f);
    p.y[(((long)b * p.Do + zo) * p.Ho + yo) * p.Wo + xo] = v;
  }
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


// NSPLIT > 1: the layer has NT * NSPLIT N-tiles and each workgroup column (blockIdx.y) computes
// NT of them -- for the tiny 128-channel layers at the bottom of GCNet's encoder (6 x 8 x 16
// voxels: 12 tiles), where one workgroup per tile left 244 CUs idle behind a 440k-cycle chain.
template <int S, int NT, int TM, int CK, int KZ = 3, int KXY = 3, int DIL = 1>
int run_conv(ConvParams p, hipStream_t s, int nsplit = 1) {
  using G = Geo<S, KZ, KXY, DIL>;
  constexpr int TY = 4 * TM;
  p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, TY);
  const long nt = (long)p.B * p.Do * p.nty * p.ntx;
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  const size_t lds = (size_t)G::IZ * G::IY(TY) * (CK / 4) * G::IX * 16;
  p.ntp = NT * nsplit;
  return launch_tiles(conv3d_mfma_kernel<S, NT, TM, CK, KZ, KXY, DIL>, p, lds, s, 512 / nsplit, nsplit);
}

// Section 2 of a packed weight buffer (the pre-split bf16 planes), present for the shapes the
// bf16x3 kernels cover: 3x3x3 with Cout 32/64, 3x3 with Cout 32/64/128; Cin % 16 == 0.
size_t bf16x3_section_bytes(int Cin, int Cout, int kd, int k) {
  if (k != 3 || Cin % 16 != 0 || Cout % 32 != 0) return 0;
  if (Cout > 128) return 0;                            // (3-D, 128 outputs: four workgroup columns of 32, small volumes only)
  return (size_t)Cin * Cout * kd * 9 * 6;
}
// Section 3 (same shapes): a 16-byte header {absolute maximum of the weights, 0, 0, 0} followed by
// the two fp16 planes of w * 2^ew (conv_split.hpp, PM = 2 | 1).
size_t f16_section_bytes(int Cin, int Cout, int kd, int k) {
  return bf16x3_section_bytes(Cin, Cout, kd, k) ? 16 + (size_t)Cin * Cout * kd * 9 * 4 : 0;
}

// Conv3d(k3, s1) to 32 channels from Cin % 32 == 0 runs on the z-sliding kernel (conv_zs.hpp):
// its split sections are packed in THAT kernel's fragment order
bool zs_layer(int Cin, int Cout, int kd, int k, int transposed) {
  return kd == 3 && k == 3 && !transposed && Cout == 32 && Cin % 32 == 0;
}

// the split sections behind the fp32 fragments of a packed buffer (`frag` = its first byte)
int pack_split_sections(const float* w, char* frag, int Cin_src, int Cin, int Cout, int kd, int k,
                        int transposed, hipStream_t s) {
  const int ntaps = kd * k * k;
  const long n = (long)Cin * Cout * ntaps;
  if (!bf16x3_section_bytes(Cin, Cout, kd, k)) return DSM_OK;
  char* sec2 = frag + n * 4;
  char* sec3 = sec2 + bf16x3_section_bytes(Cin, Cout, kd, k);
  const bool zs = zs_layer(Cin, Cout, kd, k, transposed) && Cin_src == Cin;
  if (zs)
    hipLaunchKernelGGL(pack_weights_zs_kernel<3>, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s, w,
                       (unsigned short*)sec2, (const float*)nullptr, Cin);
  else
  hipLaunchKernelGGL(pack_weights_split_kernel<3>, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s, w,
                     (unsigned short*)sec2, (const float*)nullptr, Cin, Cout, transposed, ntaps, Cin_src);
  if (hipMemsetAsync(sec3, 0, 16, s) != hipSuccess) return DSM_ERR_LAUNCH;
  const long nsrc = (long)Cin_src * Cout * ntaps;
  hipLaunchKernelGGL(absmax_kernel, dim3(dsm_cdiv(dsm_cdiv(nsrc, 4), 256) < 64 ? dsm_cdiv(dsm_cdiv(nsrc, 4), 256) : 64),
                     dim3(256), 0, s, w, nsrc, (float*)sec3);
  if (zs)
    hipLaunchKernelGGL(pack_weights_zs_kernel<2>, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s, w,
                       (unsigned short*)(sec3 + 16), (const float*)sec3, Cin);
  else
  hipLaunchKernelGGL(pack_weights_split_kernel<2>, dim3(dsm_cdiv(n, 256)), dim3(256), 0, s, w,
                     (unsigned short*)(sec3 + 16), (const float*)sec3, Cin, Cout, transposed, ntaps, Cin_src);
  return DSM_OK;
}

// dsm_conv3d_args.flags & DSM_CONV_FP32_MFMA keeps a convolution on the fp32-input MFMA (A/B and
// parity runs).  The library reads no environment variable and keeps no switch of its own: every
// tuning choice is an argument of the call (the plan is a pure function of the arguments).
inline bool bf16x3_enabled(const dsm_conv3d_args* a) { return !(a->flags & DSM_CONV_FP32_MFMA); }
inline int forced_tm(const dsm_conv3d_args* a) { return (a->flags >> DSM_CONV_TM_SHIFT) & 0xf; }

template <int NT, int CK>
int run_deconv(ConvParams p, hipStream_t s) {
  constexpr int TY = 4;
  p.ntx = dsm_cdiv(p.Wi, 32); p.nty = dsm_cdiv(p.Hi, TY);
  const long nt = (long)p.B * p.Di * p.nty * p.ntx * 2;       // x2: z-parity
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  const size_t lds = (size_t)2 * (TY + 1) * (CK / 4) * 33 * 16;
  return launch_tiles(deconv3d_mfma_kernel<NT, CK>, p, lds, s, 512);
}

}  // namespace

extern "C" size_t dsm_conv3d_packed_weight_bytes(int Cin, int Cout, int transposed) {
  (void)transposed;
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)Cin * Cout * 27 * sizeof(float) + bf16x3_section_bytes(Cin, Cout, 3, 3) +
         f16_section_bytes(Cin, Cout, 3, 3);
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
  const int rc = pack_split_sections((const float*)w_torch, (char*)w_packed, Cin, Cin, Cout, 3, 3,
                                     transposed, (hipStream_t)stream);
  return rc != DSM_OK ? rc : dsm_launch_status();
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
  const int rc = pack_split_sections((const float*)w_torch, (char*)w_packed, Cin_src, Cin, Cout, kd, k,
                                     0, (hipStream_t)stream);
  return rc != DSM_OK ? rc : dsm_launch_status();
}

extern "C" int dsm_absmax(const void* x, size_t n, float* amax, dsm_stream_t stream) {
  DSM_REQUIRE(x && amax && n > 0, DSM_ERR_ARG);
  DSM_REQUIRE(dsm_aligned16(x), DSM_ERR_ALIGN);
  dsm_clear_stale_error();
  const long blocks = dsm_cdiv(dsm_cdiv((long)n, 4), 256);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048), dim3(256), 0,
                     (hipStream_t)stream, (const float*)x, (long)n, amax);
  return dsm_launch_status();
}

extern "C" size_t dsm_conv_packed_weight_bytes(int Cin, int Cout, int kd, int k) {
  if (Cin <= 0 || Cout <= 0) return 0;
  if (!((kd == 1 || kd == 3) && (k == 1 || k == 3))) return 0;
  return (size_t)Cin * Cout * kd * k * k * sizeof(float) + bf16x3_section_bytes(Cin, Cout, kd, k) +
         f16_section_bytes(Cin, Cout, kd, k);
}

namespace {

int make_plan_f32(const dsm_conv3d_args* a, Plan* pl) {
  DSM_REQUIRE(a && a->x && a->w_packed && a->y, DSM_ERR_ARG);
  DSM_REQUIRE(a->B > 0 && a->Cin > 0 && a->Cout > 0, DSM_ERR_ARG);
  DSM_REQUIRE(a->Di > 0 && a->Hi > 0 && a->Wi > 0 && a->Do > 0 && a->Ho > 0 && a->Wo > 0,
              DSM_ERR_ARG);
  DSM_REQUIRE(a->stride == 1 || a->stride == 2, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(!a->transposed || a->stride == 2, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(a->relu >= 0 && a->relu <= 2, DSM_ERR_ARG);
  DSM_REQUIRE(a->Cin % 16 == 0, DSM_ERR_UNSUPPORTED);   // chunk sizes 8 and 16 both divide it
  DSM_REQUIRE(dsm_aligned16(a->x) && dsm_aligned16(a->w_packed) &&
              dsm_aligned16(a->y), DSM_ERR_ALIGN);
  DSM_REQUIRE(a->precision == DSM_PREC_F32 || a->precision == DSM_PREC_F16X2 || a->precision == DSM_PREC_F16, DSM_ERR_ARG);
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
      DSM_REQUIRE(a->Cin == 32, DSM_ERR_UNSUPPORTED);      // GCNet l37 (staged as whole voxels)
      *pl = Plan{3, 2, 0, 0, 0, 3, 3, 1};
    } else {
      DSM_REQUIRE(a->stride == 1, DSM_ERR_UNSUPPORTED);
      const bool zslide = !(a->flags & DSM_CONV_COUT1_CHUNKED);      // A/B runs: the chunked kernel
      *pl = Plan{(a->Cin == 32 && zslide) ? 4 : 2, 1, 0, 0, 8, 3, 3, 1};
    }
    return DSM_OK;
  }
  DSM_REQUIRE(a->Cout == 32 || a->Cout == 64 || a->Cout == 128, DSM_ERR_UNSUPPORTED);
  const int NT = a->Cout / 32;
  // Tile height: 8 rows (TM = 2) when that still gives every CU two workgroups of work,
  // else 4 rows.  Stride 2 stages 8 channels per chunk so that two workgroups fit a CU.
  if (a->transposed && NT <= 2 && a->Cin % 32 == 0 && bf16x3_enabled(a) &&
      4l * a->B * a->Di * a->Hi * a->Wi * a->Cin < 0x80000000l) {
    *pl = Plan{6, 2, NT, 1, 32, 3, 3, 1};
    return DSM_OK;
  }
  if (a->transposed) {
    DSM_REQUIRE(NT <= 2, DSM_ERR_UNSUPPORTED);   // 4 classes x NT accumulators must fit 256 VGPRs
    *pl = Plan{1, 2, NT, 1, 16, 3, 3, 1};   // (32-channel chunks measured slower: 306 vs 283 us)
    return DSM_OK;
  }
  if (a->stride == 1 && zs_layer(a->Cin, a->Cout, kd, k, 0) && bf16x3_enabled(a) &&
      (unsigned long)a->Hi * a->Wi * a->Cin * (a->vol_virtual ? 2ul : 4ul) < 0x7fffffffUL) {   // 32-bit offsets per input plane
    if (a->vol_virtual) DSM_REQUIRE(a->Cin % 64 == 0, DSM_ERR_UNSUPPORTED);   // [left | right], 32-channel groups each
    *pl = Plan{7, 1, 1, 2, 32, 3, 3, 1};
    return DSM_OK;
  }
  DSM_REQUIRE(!a->vol_virtual, DSM_ERR_UNSUPPORTED);   // only the z-sliding kernel stages a virtual volume
  const bool big = (long)a->B * a->Do * dsm_cdiv(a->Ho, 8) * dsm_cdiv(a->Wo, 32) >= 1024;
  if (a->stride == 1 && k == 3 && bf16x3_enabled(a) && bf16x3_section_bytes(a->Cin, a->Cout, kd, k) &&
      4l * a->B * a->Di * a->Hi * a->Wi * a->Cin < 0x80000000l) {       // OOBV must lie past the tensor
    // fp32 on the bf16 pipe.  16-row tiles (TM = 4) when they still give every CU a workgroup,
    // else 8-row tiles; 64 and 128 output channels always take 8-row tiles (accumulators).
    // (12-row tiles, which divide 96 rows into exactly 15 rounds instead of 11.25, measured 3 %
    // slower than 16-row tiles with their tail; 8-row tiles 12 % slower.)
    const int force_tm = forced_tm(a);             // tile-height A/B runs
    const long tiles16 = (long)a->B * a->Do * dsm_cdiv(a->Ho, 16) * dsm_cdiv(a->Wo, 32);
    int TM = (NT == 1 && dil == 1 && tiles16 >= 224) ? 4 : 2;       // 3-D, and the 32-channel 2-D maps at 1/2 resolution
    // 64 output channels on a small volume (the bottom of PSMNet's hourglass, 12 x 24 x 80:
    // 108 tiles of 8 rows): 4-row tiles when 8-row tiles leave a quarter of the CUs idle -- 83 -> 43 us
    // there.  (With 240 tiles -- the 64-channel 2-D maps -- 4-row tiles measured 10 % SLOWER: half the
    // weight reuse and more halo for nothing, every CU already had a tile.)
    const long tiles8 = (long)a->B * a->Do * dsm_cdiv(a->Ho, 8) * dsm_cdiv(a->Wo, 32);
    if (NT == 2 && dil == 1 && tiles8 < 192) TM = 1;
    if (force_tm == 2 || (force_tm == 4 && NT == 1 && dil == 1) || (force_tm == 1 && NT == 2 && dil == 1)) TM = force_tm;
    *pl = Plan{5, 1, NT, TM, 16, kd, 3, dil};
    // N-split (2-D layers, fp32 input): at most one round of 8-row tiles -- every workgroup's prologue
    // and epilogue exposed -- becomes two workgroup columns of half the output blocks, two per CU
    if (kd == 1 && TM == 2 && dil == 1 && (NT == 2 || NT == 4) && tiles8 <= 256 &&
        !(a->flags & DSM_CONV_NO_NSPLIT)) {
      pl->NT = NT / 2; pl->nsplit = 2;
      // 64 -> 64 (the towers' 31 launches at 1/4 resolution): the single-tile form, every chunk of the
      // tile requested up front (conv_once_kernel)
      if (NT == 2 && a->Cin == 64 && !(a->flags & DSM_CONV_NO_ONCE)) pl->once = 1;
    }
    // the 64-channel 3-D layers on 4-row tiles (the bottom of the hourglass: 216 tiles, one serial chain
    // of 1,296 MFMAs per wave): two columns of 32 channels halve the chain
    // (one round of tiles only: 43.0 vs 44.7 us with 216 tiles; with 432 the two columns queue: 72.5 vs 67.4)
    // 128 output channels in 3-D (GCNet's deepest encoder layers, 6 x 8 x 16 voxels): four columns of 32 on
    // 4-row tiles; larger volumes of that width stay on the fp32-input kernel (no NT = 4 3-D variant)
    if (kd == 3 && NT == 4) {
      const long t4 = (long)a->B * a->Do * dsm_cdiv(a->Ho, 4) * dsm_cdiv(a->Wo, 32);
      if (t4 <= 128 && dil == 1) { *pl = Plan{5, 1, 1, 1, 16, 3, 3, 1}; pl->nsplit = 4; return DSM_OK; }
      goto fp32_kernels;
    }
    if (kd == 3 && TM == 1 && NT == 2 && !(a->flags & DSM_CONV_NO_NSPLIT) &&
        (long)a->B * a->Do * dsm_cdiv(a->Ho, 4) * dsm_cdiv(a->Wo, 32) <= 256) {
      pl->NT = 1; pl->nsplit = 2;
    }
    return DSM_OK;
  }
  if (a->stride == 2 && kd == 3 && NT == 2 && bf16x3_enabled(a) &&
      4l * a->B * a->Di * a->Hi * a->Wi * a->Cin < 0x80000000l) {
    *pl = Plan{5, 2, 2, 1, 16, 3, 3, 1};           // stride 2 on the bf16 pipe: 4-row tiles, Cout = 64
    return DSM_OK;
  }
fp32_kernels:
  if (kd == 1) {                                   // 2-D towers: one staged slice, 16-channel chunks
    const int TM = (big && NT == 1 && a->stride == 1 && k == 3 && dil == 1) ? 2 : 1;
    *pl = Plan{0, a->stride, NT, TM, 16, 1, k, dil};
    return DSM_OK;
  }
  if (a->stride == 1) {
    const int force_tm = forced_tm(a) <= 2 ? forced_tm(a) : 0;     // tile-height A/B runs
    const int TM = force_tm ? force_tm : ((big && NT <= 2) ? 2 : 1);
    *pl = Plan{0, 1, NT, TM, (NT == 2 && TM == 2) ? 8 : 16, 3, 3, 1};   // <1,2,2,16> would spill
  } else {
    *pl = Plan{0, 2, NT, 1, 8, 3, 3, 1};
  }
  // N-split: a layer with few tiles and several N-tiles runs the NT = 1 variant with one
  // workgroup column per N-tile (4x the workgroups, each a quarter as long)
  const long tiles4 = (long)a->B * a->Do * dsm_cdiv(a->Ho, 4) * dsm_cdiv(a->Wo, 32);
  if (NT >= 2 && tiles4 <= 64 && pl->TM == 1) { pl->nsplit = NT; pl->NT = 1; }
  return DSM_OK;
}
// precision: the split kernels (kinds 5, 6) also exist on fp16 terms (conv_split.hpp); every other
// kernel computes in fp32 whatever `precision` says
int make_plan(const dsm_conv3d_args* a, Plan* pl) {
  const int rc = make_plan_f32(a, pl);
  if (rc != DSM_OK) return rc;
  pl->pm = 3;
  if ((pl->kind == 5 || pl->kind == 6 || pl->kind == 7) && a->precision != DSM_PREC_F32) {
    DSM_REQUIRE(a->x_amax != nullptr, DSM_ERR_ARG);      // the input's absolute maximum (device scalar)
    pl->pm = a->precision == DSM_PREC_F16X2 ? 2 : 1;
  }
  if (pl->pm == 3) pl->once = 0;                         // the single-tile form exists in the fp16 modes only
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
      if (pl.KZ == 3 && pl.nsplit > 1) snprintf(buf, len, "conv3d_mfma_kernel<S=%d,NT=%d,TM=%d,CK=%d>x%d", pl.S, pl.NT, pl.TM, pl.CK, pl.nsplit);
      else if (pl.KZ == 3) snprintf(buf, len, "conv3d_mfma_kernel<S=%d,NT=%d,TM=%d,CK=%d>", pl.S, pl.NT, pl.TM, pl.CK);
      else snprintf(buf, len, "conv2d_mfma_kernel<S=%d,NT=%d,TM=%d,K=%d,DIL=%d>", pl.S, pl.NT, pl.TM, pl.K, pl.DIL);
      break;
    case 1: snprintf(buf, len, "deconv3d_mfma_kernel<NT=%d,CK=%d>", pl.NT, pl.CK); break;
    case 2: snprintf(buf, len, "conv3d_cout1_kernel<CK=%d>", pl.CK); break;
    case 4: snprintf(buf, len, "conv3d_cout1_zslide_kernel"); break;
    case 5: {
      const char* pr = pl.pm == 3 ? "bf16x3" : (pl.pm == 2 ? "f16x2" : "f16");
      if (pl.KZ == 3 && pl.S == 2) snprintf(buf, len, "conv3d_%s_mfma_kernel<S=2,NT=%d,TM=%d>", pr, pl.NT, pl.TM);
      else if (pl.KZ == 3 && pl.nsplit > 1) snprintf(buf, len, "conv3d_%s_mfma_kernel<NT=%d,TM=%d>x%d", pr, pl.NT, pl.TM, pl.nsplit);
      else if (pl.KZ == 3) snprintf(buf, len, "conv3d_%s_mfma_kernel<NT=%d,TM=%d>", pr, pl.NT, pl.TM);
      else if (pl.once) snprintf(buf, len, "conv2d_%s_mfma_kernel<NT=%d,TM=%d,DIL=%d>x%d,once", pr, pl.NT, pl.TM, pl.DIL, pl.nsplit);
      else if (pl.nsplit > 1) snprintf(buf, len, "conv2d_%s_mfma_kernel<NT=%d,TM=%d,DIL=%d>x%d", pr, pl.NT, pl.TM, pl.DIL, pl.nsplit);
      else snprintf(buf, len, "conv2d_%s_mfma_kernel<NT=%d,TM=%d,DIL=%d>", pr, pl.NT, pl.TM, pl.DIL);
      break;
    }
    case 7: snprintf(buf, len, "conv3d_zs_%s_mfma_kernel%s", pl.pm == 3 ? "bf16x3" : (pl.pm == 2 ? "f16x2" : "f16"), a->vol_virtual ? "<vol>" : ""); break;
    case 6: snprintf(buf, len, "deconv3d_%s_mfma_kernel<NT=%d>", pl.pm == 3 ? "bf16x3" : (pl.pm == 2 ? "f16x2" : "f16"), pl.NT); break;
    default: snprintf(buf, len, "deconv3d_cout1_kernel"); break;
  }
  return DSM_OK;
}

extern "C" int dsm_basicblock2d_fwd(const dsm_basicblock2d_args* a, dsm_stream_t stream) {
  DSM_REQUIRE(a && a->x && a->y && a->w1_packed && a->w2_packed && a->x_amax, DSM_ERR_ARG);
  DSM_REQUIRE((a->C == 64 || a->C == 32) && a->B > 0 && a->H > 0 && a->W > 0, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(a->precision == DSM_PREC_F16X2 || a->precision == DSM_PREC_F16, DSM_ERR_UNSUPPORTED);
  DSM_REQUIRE(a->relu == 0 || a->relu == 1, DSM_ERR_ARG);
  const int C = a->C;
  const unsigned long xb = 4ul * a->B * a->H * a->W * C;
  DSM_REQUIRE(xb < 0x80000000ul, DSM_ERR_UNSUPPORTED);            // 32-bit offsets, out-of-range marker at 2^31
  const size_t sec3 = (size_t)C * C * 9 * 4 + bf16x3_section_bytes(C, C, 1, 3);
  BbParams p;
  p.x = (const float*)a->x; p.y = (float*)a->y;
  p.w1_amax = (const float*)((const char*)a->w1_packed + sec3);
  p.w2_amax = (const float*)((const char*)a->w2_packed + sec3);
  p.w1 = (const unsigned char*)a->w1_packed + sec3 + 16;
  p.w2 = (const unsigned char*)a->w2_packed + sec3 + 16;
  p.scale1 = a->scale1; p.shift1 = a->shift1; p.scale2 = a->scale2; p.shift2 = a->shift2;
  p.x_amax = a->x_amax; p.y_amax = a->y_amax;
  p.B = a->B; p.H = a->H; p.W = a->W; p.C = C; p.relu = a->relu; p.skip = a->no_skip ? 0 : 1; p.ntx = p.nty = p.ntiles = 0;
  p.xbytes = (unsigned)xb;
  p.wbytes = (unsigned)(f16_section_bytes(C, C, 1, 3) - 16);
  dsm_clear_stale_error();
  return dsmk::run_basicblock_f16(a->precision == DSM_PREC_F16X2 ? 2 : 1, p, (hipStream_t)stream);
}

extern "C" int dsm_conv3d_fwd(const dsm_conv3d_args* a, dsm_stream_t stream) {
  Plan pl;
  int rc = make_plan(a, &pl);
  if (rc != DSM_OK) return rc;
  ConvParams p;
  p.x = (const float*)a->x; p.w = (const float*)a->w_packed; p.scale = a->scale;
  p.x_amax = a->x_amax; p.w_amax = nullptr; p.y_amax = a->y_amax;
  p.shift = a->shift; p.res = (const float*)a->residual; p.y = (float*)a->y;
  p.force_blocks = (a->flags >> DSM_CONV_BLOCKS_SHIFT) & 0xffff;
  p.single_kind = (a->flags & DSM_CONV_NO_ONCE) ? 1 : 0;
  p.B = a->B; p.Cin = a->Cin; p.Cout = a->Cout;
  p.Di = a->Di; p.Hi = a->Hi; p.Wi = a->Wi; p.Do = a->Do; p.Ho = a->Ho; p.Wo = a->Wo;
  p.Dr = a->Dr; p.Hr = a->Hr; p.Wr = a->Wr; p.relu = a->relu;
  p.ntx = p.nty = p.ntiles = 0;
  if (pl.kind == 7) {
    ZsParams z;
    const size_t n = (size_t)a->Cin * 32 * 27;
    const char* sec2 = (const char*)a->w_packed + n * 4;
    const char* sec3 = sec2 + n * 6;
    z.x = (const float*)a->x; z.scale = a->scale; z.shift = a->shift; z.res = (const float*)a->residual;
    z.y = (float*)a->y; z.x_amax = a->x_amax; z.y_amax = a->y_amax;
    z.w_amax = pl.pm == 3 ? nullptr : (const float*)sec3;
    z.w = (const unsigned char*)(pl.pm == 3 ? sec2 : sec3 + 16);
    z.wbytes = (unsigned)(pl.pm == 3 ? n * 6 : n * 4);
    z.B = a->B; z.Cin = a->Cin;
    z.Di = a->Di; z.Hi = a->Hi; z.Wi = a->Wi; z.Do = a->Do; z.Ho = a->Ho; z.Wo = a->Wo;
    z.Dr = a->Dr; z.Hr = a->Hr; z.Wr = a->Wr; z.relu = a->relu;
    z.vol = a->vol_virtual ? 1 : 0; z.vol_mask_left = a->vol_mask_left ? 1 : 0;
    dsm_clear_stale_error();
    if (pl.pm == 3) return launch_conv_zs<3>(z, p.force_blocks, (hipStream_t)stream);
    return dsmk::run_zs_f16(pl.pm, z, p.force_blocks, (hipStream_t)stream);
  }
  {
    const unsigned long xb = 4ul * a->B * a->Di * a->Hi * a->Wi * a->Cin;
    const int kd_ = a->kd ? a->kd : 3, k_ = a->k ? a->k : 3;
    const unsigned long wb = 4ul * a->Cin * a->Cout * kd_ * k_ * k_;
    DSM_REQUIRE(xb < (1ul << 32) && wb < (1ul << 32), DSM_ERR_UNSUPPORTED);   // 32-bit buffer offsets
    p.xbytes = (unsigned)xb; p.wbytes = (unsigned)wb;
  }
  hipStream_t s = (hipStream_t)stream;
  dsm_clear_stale_error();
  if (pl.kind == 3) {
    p.ntx = dsm_cdiv(p.Wi, 32); p.nty = dsm_cdiv(p.Hi, 4);
    const long nt = (long)p.B * p.Di * p.nty * p.ntx;
    DSM_REQUIRE(nt < (1L << 31), DSM_ERR_UNSUPPORTED);
    const size_t lds = (size_t)(2 * 5 * 33 * 8 + 27 * 8) * 16;     // 45.7 KB
    hipLaunchKernelGGL(deconv3d_cout1_kernel, dim3((unsigned)nt), dim3(NTHREADS), lds, s, p);
    return dsm_launch_status();
  }
  if (pl.kind == 5 || pl.kind == 6) {
    const int kz = pl.kind == 6 ? 3 : pl.KZ;
    const char* sec2 = (const char*)a->w_packed + (size_t)a->Cin * a->Cout * kz * 9 * 4;
    if (pl.pm == 3) {
      p.w = (const float*)sec2;
      p.wbytes = (unsigned)bf16x3_section_bytes(a->Cin, a->Cout, kz, 3);
      return dispatch_split<3>(pl, p, s);
    }
    const char* sec3 = sec2 + bf16x3_section_bytes(a->Cin, a->Cout, kz, 3);
    p.w_amax = (const float*)sec3;
    p.w = (const float*)(sec3 + 16);
    p.wbytes = (unsigned)(f16_section_bytes(a->Cin, a->Cout, kz, 3) - 16);
    return dsmk::run_split_f16(pl, p, s);
  }
  if (pl.kind == 4) {
    p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, 8);
    // z-segment: the longest of 16/8/4/2 planes that still gives the 768 workgroup slots
    // (3 per CU) most of a round; each segment stages two halo planes on top of its own
    const long cols = (long)p.B * p.nty * p.ntx;
    int zseg = 16;
    while (zseg > 2 && cols * dsm_cdiv(p.Do, zseg) < 600) zseg >>= 1;
    const long nt = cols * dsm_cdiv(p.Do, zseg);
    DSM_REQUIRE(nt < (1L << 31), DSM_ERR_UNSUPPORTED);
    const size_t lds = (size_t)10 * 34 * 9 * 16;                   // 47.8 KB
    hipLaunchKernelGGL(conv3d_cout1_zslide_kernel, dim3((unsigned)nt), dim3(NTHREADS), lds, s, p,
                       p.w, zseg);
    return dsm_launch_status();
  }
  if (pl.kind == 2) {
    p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, 8);
    p.ntiles = p.B * p.Do * p.nty * p.ntx;
    const size_t lds = (size_t)3 * 10 * 34 * 2 * 16;              // CK = 8: 32.6 KB
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
  DSM_CASE(0, 1, 1, 1, 16, (run_conv<1, 1, 1, 16>(p, s, pl.nsplit)));
  DSM_CASE(0, 1, 2, 2, 8, (run_conv<1, 2, 2, 8>(p, s)));
  DSM_CASE(0, 1, 2, 1, 16, (run_conv<1, 2, 1, 16>(p, s)));
  DSM_CASE(0, 1, 4, 1, 16, (run_conv<1, 4, 1, 16>(p, s)));
  DSM_CASE(0, 2, 1, 1, 8, (run_conv<2, 1, 1, 8>(p, s, pl.nsplit)));
  DSM_CASE(0, 2, 2, 1, 8, (run_conv<2, 2, 1, 8>(p, s)));
  DSM_CASE(0, 2, 4, 1, 8, (run_conv<2, 4, 1, 8>(p, s)));
#undef DSM_CASE
  return DSM_ERR_UNSUPPORTED;
}
