// Two convolutions per launch: the stride-1 BasicBlock of PSMNet's towers (models/psmnet/submodule.py:24-46;
// C = 64: layer2, C = 32: layer1) and of GCNet's (models/util_conv.py:181-210; C = 32, ReLU after the add),
//   y = [ReLU](BN2(conv2(ReLU(BN1(conv1(x))))) [+ x]),   conv1, conv2 = Conv2d(C, C, 3, stride 1, pad 1)
// (without the skip: two convbn + ReLU in a row, PSMNet's firstconv[2..5], submodule.py:81-86),
// with the intermediate map kept in LDS.  Included by conv_f16.hip (fp16 modes) after conv_split.hpp.
//
// Why: the towers' 64-channel layers run ONE round of workgroups per launch, whose load, multiply and
// store phases add up instead of overlapping, and every kernel boundary sends the activations, the
// weights and the affine through the Infinity Cache again (profiles/r03_ablation.md, 4: 25.5 us per
// launch inside the forward for ~6 us of MFMAs).  Fusing a block removes one of its two boundaries, one
// write and one read of the 64-channel map, and one set of launch / set-up / tail.
//
// A workgroup (8 waves) owns an 8 x 32 output tile:
//  1. the 12 x 36 x 64 input box is requested at once (conv_once_kernel's finding: a one-tile workgroup
//     has nothing else to overlap its loads with), 16 channels at a time split into the LDS image X
//     [voxel][plane][16 ch], pitch 80 B -- conv_split_kernel's layout;
//  2. conv1 on the 10 x 34 halo of the tile: 340 intermediate voxels = 11 M-tiles of 32; wave (nw, mg)
//     takes output block nw (32 channels) of tiles mg, mg + 4, mg + 8 (measured against two other
//     mappings -- eight waves each with both blocks of tiles w, w + 8: 41.1 us; four waves with both
//     blocks of three tiles: 40.5 us; this one 38.0 us, hot caches); lane r of a tile reads ITS voxel's
//     fragment (per-lane LDS addresses: the
//     M-tiles are runs of the linearised 10 x 34 box, not image rows); 1.33 x the MFMAs of the plain
//     convolution;
//  3. BN1 + ReLU in registers; voxels outside the image are zeros (conv2's padding); the tile's own
//     maximum scales the intermediate for its fp16 split (a power of two, exact; per tile, so the
//     low bits differ from the unfused pair of launches, the error band does not); written to the
//     LDS image T [4 chunks][340 voxels][plane][16 ch];
//  4. conv2 straight from T (no staging): wave (nw, mg) owns output rows 2 mg, 2 mg + 1, block nw;
//  5. BN2 with the tile's factor, + x (re-read from L2), tensor maximum, store.
// Weights: the packed f16 sections of the two layers (dsm_conv_pack_weights), ring of three items.
#pragma once

#ifndef DSM_BB_OFF
#define DSM_BB_OFF 0           // timing-only A/B builds: 1 no conv1 MFMAs, 2 no conv2 MFMAs, 4 no activation loads, 8 no stores, 16 no intermediate split
#endif

template <int PM, int C_>
struct BbCfg {
  static constexpr int NP = Prec<PM>::NP, NPW = Prec<PM>::NPW;
  static constexpr int C = C_, NCH = C / 16, NT = C / 32;     // 64 (PSMNet layer2) or 32 (PSMNet layer1, GCNet)
  static constexpr int MG = 8 / NT;                          // waves = NT output blocks x MG tile groups
  static constexpr int TY = 8, BY = TY + 4, BX = 36;        // input box
  static constexpr int MY = TY + 2, MX = 34, NMID = MY * MX; // intermediate box: 340 voxels
  static constexpr int NMT = (NMID + 31) / 32;               // 11 M-tiles
  static constexpr int NVOX = BY * BX;                       // 432
  static constexpr int THREADS = 512;
  static constexpr int NPF = (NVOX * 4 + THREADS - 1) / THREADS;   // 4 staged quads per thread and chunk
  static constexpr int PITCH = 32 * NP + 16;
  static constexpr int XIMG = NPF * 128 * PITCH;             // 512 voxel slots
  static constexpr int TCH = NMT * 32 * PITCH;               // one 16-channel chunk of T (352 voxel slots)
  // T takes over X's bytes: X is dead when conv1 is done (the barrier of the tile-maximum reduction
  // stands between its last read and T's first write)
  static constexpr int IMGS = XIMG > NCH * TCH ? XIMG : NCH * TCH;
  static constexpr int LDS = IMGS + 4 * C * 4 + 64;        // 111.7 KB (C = 64) | 57.0 KB (C = 32: two workgroups per CU)
  static constexpr int WGS = 2 * LDS <= 160 * 1024 ? 2 : 1;
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <int PM, int C>
__global__ __launch_bounds__(512, (BbCfg<PM, C>::WGS)) void basicblock2d_kernel(BbParams p) {
  using Cf = BbCfg<PM, C>;
  using frag = typename Prec<PM>::frag;
  constexpr int NP = Cf::NP, NPW = Cf::NPW, NCH = Cf::NCH, NT = Cf::NT, TY = Cf::TY, BX = Cf::BX, MX = Cf::MX,
                NMID = Cf::NMID, NVOX = Cf::NVOX, NPF = Cf::NPF, PITCH = Cf::PITCH, TCH = Cf::TCH;
  constexpr int AHEAD = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* const ximg = lds_raw;
  unsigned char* const timg = lds_raw;          // (aliases X, see BbCfg)
  float* const aff = reinterpret_cast<float*>(lds_raw + Cf::IMGS);           // scale1, shift1, scale2, shift2
  float* const red = aff + 4 * C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  constexpr int MG = Cf::MG;
  const int nw = wave / MG, mg = wave % MG;    // this wave's 32-channel output block and M-tile group
  int id = blockIdx.x;
  if (id >= p.ntiles) return;
  const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
  const int ty0 = (id % p.nty) * TY;
  const int b = id / p.nty;

  // power-of-two factors: x and the two weight tensors (conv_split.hpp); the intermediate's comes later
  const int ex = dsm_amax_exponent(*p.x_amax), ew1 = dsm_amax_exponent(*p.w1_amax), ew2 = dsm_amax_exponent(*p.w2_amax);
  const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dsm_pow2f(ex))));
  const float so1 = dsm_pow2f(-(ex + ew1));
  if (tid < 4 * C) {
    const int c = tid % C, which = tid / C;
    const float* src = which == 0 ? p.scale1 : which == 1 ? p.shift1 : which == 2 ? p.scale2 : p.shift2;
    float v = src ? src[c] : ((which & 1) ? 0.f : 1.f);
    if (which == 0) v *= so1;
    aff[tid] = v;
  }

  // ---- 1. the whole input box, requested now (chunk-major: chunk 0 arrives first)
  constexpr unsigned OOBV = 0x80000000u;
  unsigned voff[NPF];
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + k * Cf::THREADS;
    const int v = e >> 2, q = e & 3;
    const int yy = v / BX, xx = v % BX;
    const int y = ty0 - 2 + yy, x = tx0 - 2 + xx;
    const bool ok = v < NVOX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    voff[k] = ok ? (unsigned)(4l * ((((long)b * p.H + y) * p.W + x) * C + 4 * q)) : OOBV;
  }
  f32x4 pf[NCH][NPF];
  static_for<0, NCH>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(p.x) + c * 64, p.xbytes);
    static_for<0, NPF>([&](auto kc) {
      if ((DSM_BB_OFF & 4) && p.B != 12345) pf[c][decltype(kc)::value] = f32x4{1.f, 0.5f, 0.25f, 2.f};
      else pf[c][decltype(kc)::value] = buffer_load16(rs, voff[decltype(kc)::value], 0);
    });
  });
  const unsigned lane16 = lane * 16u;
  const int wr_off = (tid >> 2) * PITCH + (tid & 3) * 8;     // voxel (tid >> 2) + 128 k of the X image

  frag wq[AHEAD][NP];
  const unsigned wlane = lane16 + (unsigned)nw * (NPW * 64 * 16);
  auto wload = [&](const __amdgpu_buffer_rsrc_t& rs, auto ic, unsigned wb) {
    constexpr int item = decltype(ic)::value;
#pragma unroll
    for (int q = 0; q < NP; ++q)
      wq[item % AHEAD][q] = __builtin_bit_cast(
          frag, buffer_load16(rs, wlane, wb + (item * NT * NPW + q) * (64 * 16)));
  };
  constexpr unsigned WCH = 9 * NT * NPW * 64 * 16;            // weight bytes per 16-channel chunk

  // ---- 2. conv1: this wave's M-tiles w, w + 8 of the linearised 10 x 34 intermediate box
  constexpr int NM1 = (Cf::NMT + MG - 1) / MG;               // 3 | 2
  int xbase[NM1];
  bool mt_on[NM1];
#pragma unroll
  for (int m = 0; m < NM1; ++m) {
    const int mt = mg + MG * m;
    mt_on[m] = mt < Cf::NMT;                                  // wave-uniform
    const int v = min(32 * mt + r, NMID - 1);
    xbase[m] = ((v / MX) * BX + v % MX) * PITCH + h * 16;     // input voxel of tap (0, 0)
  }
  f32x16 acc1[NM1];
#pragma unroll
  for (int m = 0; m < NM1; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[m][i] = 0.f;
  const __amdgpu_buffer_rsrc_t w1rs = make_rsrc(p.w1, p.wbytes);
  static_for<0, AHEAD - 1>([&](auto ic) { wload(w1rs, ic, 0u); });
  static_for<0, NCH>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (c > 0) __syncthreads();       // everyone is done reading the previous chunk's image
    static_for<0, NPF>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      unsigned lo[NP], hi[NP];
      split_pair<PM>(pf[c][k].x, pf[c][k].y, sx, lo);
      split_pair<PM>(pf[c][k].z, pf[c][k].w, sx, hi);
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        u32x2 v; v.x = lo[q]; v.y = hi[q];
        *reinterpret_cast<u32x2*>(ximg + wr_off + k * (128 * PITCH) + q * 32) = v;
      }
    });
    __syncthreads();
    static_for<0, 9>([&](auto ic) {
      constexpr int item = decltype(ic)::value;
      constexpr int tap_off = ((item / 3) * BX + item % 3) * PITCH;
      if constexpr (item + AHEAD - 1 < 9) wload(w1rs, std::integral_constant<int, item + AHEAD - 1>{}, c * WCH);
      else if constexpr (c + 1 < NCH) wload(w1rs, std::integral_constant<int, item + AHEAD - 1 - 9>{}, (c + 1) * WCH);
      frag xq[NM1][NP];
#pragma unroll
      for (int m = 0; m < NM1; ++m)
        if (mt_on[m]) {
#pragma unroll
          for (int q = 0; q < NP; ++q)
            xq[m][q] = *reinterpret_cast<const frag*>(ximg + xbase[m] + tap_off + q * 32);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NM1; ++m)
        if (mt_on[m] && (!(DSM_BB_OFF & 1) || p.B == 12345)) mma32<PM>(acc1[m], wq[item % AHEAD], xq[m]);
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  // the skip values of this wave's output rows: requested now, needed after conv2 (requested in the
  // epilogue they cost a memory latency with nothing left to hide it: 1.2 us of 39)
  constexpr int R2 = 8 / MG;                    // output rows per wave: 2 | 1
  const int xo = tx0 + r;
  Residual rr[R2];
#pragma unroll
  for (int m = 0; m < R2; ++m) {
    const int yo = ty0 + R2 * mg + m;
    if (p.skip && xo < p.W && yo < p.H) load_residual(rr[m], p.x + (((long)b * p.H + yo) * p.W + xo) * C + 32 * nw + 4 * h);
  }
  // conv2's first weights ride behind the intermediate's processing
  const __amdgpu_buffer_rsrc_t w2rs = make_rsrc(p.w2, p.wbytes);
  static_for<0, AHEAD - 1>([&](auto ic) { wload(w2rs, ic, 0u); });

  // ---- 3. BN1 + ReLU, zero outside the image, the tile's maximum, split into T
  float tmax = 0.f;
#pragma unroll
  for (int m = 0; m < NM1; ++m) {
    if (!mt_on[m]) continue;
    const int v = 32 * (mg + MG * m) + r;
    const int y = ty0 - 1 + v / MX, x = tx0 - 1 + v % MX;
    const bool inside = v < NMID && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cb = 32 * nw + 8 * g + 4 * h;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + cb), sh = *reinterpret_cast<const f32x4*>(aff + C + cb);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = fmaxf(acc1[m][4 * g + i] * sc[i] + sh[i], 0.f);
        t = inside ? t : 0.f;
        acc1[m][4 * g + i] = t;
        tmax = fmaxf(tmax, t);
      }
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  if (lane == 0) red[wave] = tmax;
  __syncthreads();                              // also: every wave is done with the X image
  tmax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
  const int et = dsm_amax_exponent(tmax);
  const float st = dsm_pow2f(et);
#pragma unroll
  for (int m = 0; m < NM1; ++m) {
    if (!mt_on[m] || ((DSM_BB_OFF & 16) && p.B != 12345)) continue;
    const int v = 32 * (mg + MG * m) + r;       // slots 340 .. 351 of the last tile: zeros, never read as real voxels
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // channels 32 nw + 8 g + 4 h + i: chunk 2 nw + (g >> 1), byte 2 (8 (g & 1) + 4 h) of the voxel's 32
      unsigned lo[NP], hi[NP];
      split_pair<PM>(acc1[m][4 * g], acc1[m][4 * g + 1], st, lo);
      split_pair<PM>(acc1[m][4 * g + 2], acc1[m][4 * g + 3], st, hi);
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        u32x2 w; w.x = lo[q]; w.y = hi[q];
        *reinterpret_cast<u32x2*>(timg + (2 * nw + (g >> 1)) * TCH + v * PITCH + q * 32 + 16 * (g & 1) + 8 * h) = w;
      }
    }
  }
  __syncthreads();

  // ---- 4. conv2 from T: wave w = output row w, both 32-channel blocks
  f32x16 acc2[R2];
#pragma unroll
  for (int m = 0; m < R2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[m][i] = 0.f;
  const int tbase = ((R2 * mg) * MX + r) * PITCH + h * 16;
  static_for<0, NCH>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    static_for<0, 9>([&](auto ic) {
      constexpr int item = decltype(ic)::value;
      constexpr int tap_off = ((item / 3) * MX + item % 3) * PITCH;
      if constexpr (item + AHEAD - 1 < 9) wload(w2rs, std::integral_constant<int, item + AHEAD - 1>{}, c * WCH);
      else if constexpr (c + 1 < NCH) wload(w2rs, std::integral_constant<int, item + AHEAD - 1 - 9>{}, (c + 1) * WCH);
      frag xq[R2][NP];
#pragma unroll
      for (int m = 0; m < R2; ++m)
#pragma unroll
        for (int q = 0; q < NP; ++q)
          xq[m][q] = *reinterpret_cast<const frag*>(timg + c * TCH + tbase + m * (MX * PITCH) + tap_off + q * 32);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < R2; ++m)
        if (!(DSM_BB_OFF & 2) || p.B == 12345) mma32<PM>(acc2[m], wq[item % AHEAD], xq[m]);
      __builtin_amdgcn_sched_barrier(0);
    });
  });

  // ---- 5. BN2 with the tile's factor, + x, store
  float am = 0.f;
  if (xo < p.W && (!(DSM_BB_OFF & 8) || p.B == 12345)) {
    const float so2 = dsm_pow2f(-(et + ew2));
    Affine af;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      af.sc[g] = *reinterpret_cast<const f32x4*>(aff + 2 * C + 32 * nw + 8 * g + 4 * h) * so2;
      af.sh[g] = *reinterpret_cast<const f32x4*>(aff + 3 * C + 32 * nw + 8 * g + 4 * h);
    }
#pragma unroll
    for (int m = 0; m < R2; ++m) {
      const int yo = ty0 + R2 * mg + m;
      if (yo < p.H)
        store_tile<C>(acc2[m], af, p.relu, p.y + (((long)b * p.H + yo) * p.W + xo) * C + 32 * nw + 4 * h, rr[m], p.skip != 0, am);
    }
  }
  flush_amax8(p.y_amax, am, red + 8);
}

template <int PM, int C>
int launch_basicblock2d(BbParams p, hipStream_t s) {
  using Cf = BbCfg<PM, C>;
  p.ntx = dsm_cdiv(p.W, 32); p.nty = dsm_cdiv(p.H, Cf::TY);
  const long nt = (long)p.B * p.nty * p.ntx;
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)basicblock2d_kernel<PM, C>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS) != hipSuccess)
      return DSM_ERR_LAUNCH;
    configured = true;
  }
  hipLaunchKernelGGL((basicblock2d_kernel<PM, C>), dim3((unsigned)nt), dim3(Cf::THREADS), Cf::LDS, s, p);
  return dsm_launch_status();
}
