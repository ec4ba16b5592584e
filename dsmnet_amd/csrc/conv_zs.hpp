// The z-sliding convolution: Conv3d(k3, s1, p1), Cin % 32 == 0 -> Cout = 32, fp32 tensors, on the
// 16-bit matrix pipe at precision mode PM (conv_split.hpp: 3 = bf16x3, 2 = f16x2, 1 = f16).
// Included by conv3d.hip (PM = 3) and conv_f16.hip (PM = 2, 1) after conv_split.hpp.
//
// Replaces convbn_3d + ReLU + myadd_3d of models/psmnet/submodule.py:16-19 and
// stackhourglass.py:10-20,73-98,135-149 for the 32-channel full-resolution layers (dres0, dres1,
// classif*.0: 652 of the PSMNet trunk's 950 GFLOP) and, with `vol` set, the volume build of
// stackhourglass.py:124-133 / gcnet.py:130-135 fused into the first of them -- the dominant kernel
// of the forward.
//
// Structure:
//  * z-sliding: a workgroup owns an (8 y x 32 x) column and walks z.  One input plane is staged
//    ONCE and contributes to three output planes (z-taps 2, 1, 0 -> accumulator sets A0, A1, A2);
//    after a plane, A0 is complete, is written out and the sets rotate.  HBM/L2 traffic for the
//    input falls from 3.6x (three z-taps x 1.2 halo) to 1.33x (the y/x halo only).
//  * the work is the linearised (column, output plane) space cut into gridDim.x EQUAL ranges
//    (one persistent workgroup per CU): no tail round; a range that crosses a column border just
//    starts a new segment.  Partial sums never leave registers: a segment's first and last
//    planes run only the z-taps whose output plane lies inside the segment.
//  * wave (ah, xh) owns output channels 16 ah .. 16 ah + 15 and the 16 columns xh of the tile, all its
//    rows, on 16x16x32 MFMAs: per (tap position, z-tap) step NP weight fragments (L2, a ring of
//    nine) feed TY product groups; the TY NP activation fragments of a tap position are read from LDS
//    once and serve its three z-taps.
//  * staging: a chunk = one input plane x 32 channels, 10 x 34 voxels x 8 fp32 quads = 11 buffer
//    loads per thread (zero address VALU: per-column offsets, per-chunk descriptor base), issued
//    one per step over the first 11 steps; over the last 11 steps each quad is split into its NP
//    16-bit planes (conv_split.hpp split_pair: the operand split of this precision mode) and written
//    to the OTHER LDS image [plane][g = quad & 3][voxel][16 B] -- the operand order of
//    v_mfma_f32_16x16x32: lane (g, x) of the B operand reads the 16-byte unit (g, voxel x), element
//    e of which is channel 16 (e >> 2) + 4 g + (e & 3) of the group.  Rows are 352 units (a multiple
//    of 256 B: conflict-free ds_read_b128 for its 16-lane groups, which span two g rows); a 16-lane
//    group of a ds_write_b64 holds the two halves of ONE unit row for 8 consecutive voxels -- 128
//    contiguous bytes, conflict-free under the stores' 32-bank rule (MI355X_MICROARCH.md, LDS).
//    Two images, one barrier per chunk.  Each voxel is split 1.33 times (once per staging), not
//    once per tap as in the r01 kernel.
//  * one workgroup per CU with one wave per SIMD (8 x 32 tiles), or two (4 x 32 tiles; the fp16 modes).
//  * `vol`: the input is a concatenation cost volume that is NEVER MATERIALISED: x is the NHWC
//    feature tensor (2B, H, W, C) [left images, then right images]; input plane d of the
//    (B, 2C, D, H, W) volume is staged as [left | right shifted by d voxels] with x < d zeroed
//    (right half always, left half iff vol_mask_left).
#pragma once

#ifndef DSM_ZS_TILING_F16
#define DSM_ZS_TILING_F16 1       // the fp16 modes' tiling (V below); 0 in A/B builds
#endif

// Two tilings: V = 0: 8 x 32 output tile, one workgroup per CU (bf16x3: its two 67.6 KB images fill the
// LDS); V = 1: 4 x 32 tile, TWO workgroups per CU (<= 256 registers, 2 x 57 KB of LDS in the fp16
// modes).  One wave per SIMD issues in order: every load, LDS access and VALU instruction of the step
// takes issue cycles the MFMAs do not get (the kernel loses the same ~90 us to them at six, three or
// one MFMA per product: profiles/r03_ablation.md); a second resident workgroup issues its MFMAs in
// those cycles.  At six MFMAs per product the chip is power-bound and the second workgroup buys
// nothing (r02); at three it does.
template <int PM, int V> struct ZsCfg {
  static constexpr int NP = Prec<PM>::NP, NPW = Prec<PM>::NPW;
  static constexpr int TY = V ? 4 : 8, IY = TY + 2, IX = 34;
  static constexpr int WGS = V ? 2 : 1;               // workgroups per CU
  static constexpr int NV = IY * IX;                  // 340 | 204 voxels of the halo box
  static constexpr int NPF = (NV * 8 + NTHREADS - 1) / NTHREADS;   // 11 | 7 staged quads per thread
  static constexpr int NVP = 32 * NPF;                // units per (plane, g) row: a multiple of 16
  static constexpr int ROW = NVP * 16;                // 5,632 | 3,584 B
  static constexpr int IMG = NP * 4 * ROW;            // V = 0: 67,584 | 45,056 | 22,528 B
  static constexpr int LDS = 2 * IMG + 256;           // two images + the folded affine
  static constexpr int NSTEP = 27;
  static constexpr int WSTEP = 2 * NPW * 1024;        // weight bytes per (tap position, z-tap) step
  static constexpr int CONV0 = NSTEP - NPF;           // first step that splits / stores a staged quad
  // weight ring: fragments are requested WAHEAD steps before their use.  A step is 8 product groups =
  // 48 (bf16x3) / 24 (f16x2) / 8 (f16) MFMAs of 16 cycles: two steps cover an L2 round trip in the
  // six-MFMA form only.  The ring has 9 slots (9 divides NSTEP: slot = step % 9 stays consistent
  // across chunks); only WAHEAD + 1 of them are live at a time.
#ifndef DSM_ZS_WAHEAD
#define DSM_ZS_WAHEAD (PM == 3 ? 2 : (V ? 3 : 4))
#endif
  static constexpr int WRING = 9, WAHEAD = DSM_ZS_WAHEAD;
  static_assert(NSTEP % WRING == 0 && WAHEAD < WRING, "weight ring");
  static_assert(32 * NPF <= NVP && NVP % 16 == 0, "image row");
  static_assert(WGS * LDS <= 160 * 1024, "LDS");
};

template <int PM>
__device__ __forceinline__ void mma16(f32x4& c, const typename Prec<PM>::frag (&w)[Prec<PM>::NP],
                                      const typename Prec<PM>::frag (&x)[Prec<PM>::NP]) {
  if constexpr (PM == 3) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], c, 0, 0, 0);
  } else if constexpr (PM == 2) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[0], c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[0], c, 0, 0, 0);
  }
}

template <int PM, int V>
__global__ __launch_bounds__(NTHREADS, (ZsCfg<PM, V>::WGS)) void conv_zs_kernel(ZsParams p) {
  using C = ZsCfg<PM, V>;
  using frag = typename Prec<PM>::frag;
  constexpr int NP = C::NP, NPW = C::NPW, TY = C::TY, IX = C::IX, NV = C::NV, NPF = C::NPF, ROW = C::ROW,
                IMG = C::IMG, NSTEP = C::NSTEP, WSTEP = C::WSTEP, CONV0 = C::CONV0, WRING = C::WRING, WAHEAD = C::WAHEAD;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int ah = wave >> 1, xh = wave & 1;    // this wave's 16 output channels (16 ah ..) and 16 columns, all TY rows
  const int ncg = p.Cin >> 5;

  // this workgroup's range of the linearised (column, output plane) space; workgroups on one XCD
  // (id % 8) take neighbouring ranges (shared halo columns meet in one L2; speed only)
  const int G = gridDim.x, id = blockIdx.x;
  const int logical = (G & 7) == 0 ? (id & 7) * (G >> 3) + (id >> 3) : id;
  const long u_begin = p.nunits * logical / G, u_end = p.nunits * (logical + 1) / G;
  if (u_begin >= u_end) return;

  // power-of-two scaling of the fp16 modes (conv_split.hpp) and the folded-BN affine of this
  // lane's 8 channels (16 a + 4 g + i), the output factor folded into the scale
  float sx = 1.f, so = 1.f;
  if constexpr (PM != 3) {
    const int ex = dsm_amax_exponent(*p.x_amax), ew = dsm_amax_exponent(*p.w_amax);
    sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dsm_pow2f(ex))));
    so = dsm_pow2f(-(ex + ew));
  }
  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG);
  if (tid < 64) aff[tid] = tid < 32 ? (p.scale ? p.scale[tid] * so : so) : (p.shift ? p.shift[tid - 32] : 0.f);
  __syncthreads();
  const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + 16 * ah + 4 * g);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + 32 + 16 * ah + 4 * g);
  float am = 0.f;

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

  // ---- staging: this thread's quads of a chunk: voxel sv + 32 k, quad sq -- a wave covers 8 voxels x 8
  // quads (1 KiB of contiguous fp32), its 16-lane groups the two halves of one unit row g
  const unsigned vstride = p.vol ? (unsigned)p.Cin * 2u : (unsigned)p.Cin * 4u;   // bytes per voxel of the staged tensor
  const unsigned plane_bytes = vstride * (unsigned)p.Hi * (unsigned)p.Wi;         // < 2 GiB: checked by the host
  constexpr unsigned OOBV = 0x80000000u;
  const int sq = (lane >> 4) + 4 * (lane & 1), sv = 8 * wave + ((lane & 15) >> 1);
  unsigned voff[NPF];
  int vx[NPF];                  // the quad's x coordinate (virtual volume: plane d masks x < d)
  auto column_offsets = [&](int col) {
    const int tx = col % p.ntx, ty = (col / p.ntx) % p.nty;
    const int y0 = ty * TY - 1, x0 = tx * 32 - 1;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int v = sv + 32 * k;
      const int yy = v / IX, xx = v % IX;
      const int y = y0 + yy, x = x0 + xx;
      const bool ok = v < NV && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
      voff[k] = ok ? ((unsigned)y * (unsigned)p.Wi + (unsigned)x) * vstride + 16u * (unsigned)sq : OOBV;
      vx[k] = x;
    }
  };
  auto chunk_rsrc = [&](const It& q) {
    const int b = q.col / (p.ntx * p.nty);
    long off; unsigned shift = 0;
    if (p.vol) {
      const int ncs = ncg >> 1;                       // channel groups per side
      const bool right = q.cg >= ncs;
      shift = right ? (unsigned)q.zi * vstride : 0u;
      off = (long)(right ? p.B + b : b) * (long)plane_bytes + (long)(right ? q.cg - ncs : q.cg) * 128 - (long)shift;
    } else {
      off = ((long)b * p.Di + q.zi) * (long)plane_bytes + (long)q.cg * 128;
    }
    return make_rsrc(reinterpret_cast<const char*>(p.x) + off, q.valid ? plane_bytes + shift : 0u);
  };
  auto chunk_xmin = [&](const It& q) {            // staged voxels with x below this are zeros
    return (p.vol && (p.vol_mask_left || 2 * q.cg >= ncg)) ? q.zi : -0x40000000;
  };
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  const unsigned lane16 = lane * 16u;
  // LDS slot of this thread's quads: unit g = sq & 3, half = sq >> 2, voxel sv + 32 k
  const int st_off = (lane >> 4) * ROW + sv * 16 + (lane & 1) * 8;                 // + 512 k, + plane * 4 ROW
  // activation fragment of this lane: voxel (r + ky, 16 xh + j + kx), unit g, plane q
  const int rd_off = g * ROW + (16 * xh + j) * 16;

  f32x4 acc[3][TY];
  frag xq[2][TY][NP];           // [tap-position parity][row][plane]
  frag wq[WRING][NP];           // [step % WRING][plane]: this wave's 16-channel block only
  f32x4 pf[NPF];

  auto zero_set = [&](auto sc_) {
    constexpr int s = decltype(sc_)::value;
#pragma unroll
    for (int r = 0; r < TY; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto wload = [&](auto sc_, unsigned wb) {                     // weights of step s (of the chunk at wb)
    constexpr int s = decltype(sc_)::value;
#if defined(DSM_ZS_ABLATE) && DSM_ZS_ABLATE == 2
    if (p.B != 12345 && s >= 0 && wb != 0xffffffffu) return;   // timing-only build: weight fragments never loaded
#endif
#pragma unroll
    for (int q = 0; q < NP; ++q)
      wq[s % WRING][q] = __builtin_bit_cast(
          frag, buffer_load16(wrsrc, lane16 + (unsigned)ah * (NPW * 1024), wb + s * WSTEP + q * 1024));
  };
  auto xload = [&](auto tpc, auto rc, const unsigned char* rd) {
    constexpr int tp = decltype(tpc)::value, r = decltype(rc)::value;
    constexpr int ky = tp / 3, kx = tp % 3;
#pragma unroll
    for (int q = 0; q < NP; ++q)
      xq[tp & 1][r][q] = *reinterpret_cast<const frag*>(rd + q * 4 * ROW + ((r + ky) * IX + kx) * 16);
  };
  // staged quad k -> its NP planes in image `wr`
  auto convert = [&](auto kc, unsigned char* wr) {
    constexpr int k = decltype(kc)::value;
    unsigned lo[NP], hi[NP];
    split_pair<PM>(pf[k].x, pf[k].y, sx, lo);
    split_pair<PM>(pf[k].z, pf[k].w, sx, hi);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      u32x2 v; v.x = lo[q]; v.y = hi[q];
      *reinterpret_cast<u32x2*>(wr + q * 4 * ROW + 512 * k) = v;
    }
  };

  // epilogue of accumulator set 0 = output plane zo of the column
  auto emit = [&](const It& q, int zo) {
    const int tx = q.col % p.ntx, ty = (q.col / p.ntx) % p.nty, b = q.col / (p.ntx * p.nty);
    const int xo = tx * 32 + 16 * xh + j;
    if (xo >= p.Wo) return;
#pragma unroll
    for (int r = 0; r < TY; ++r) {
      const int yo = ty * TY + r;
      if (yo >= p.Ho) continue;
      f32x4 v = acc[0][r] * sc + sh;
      if (p.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (p.res)
        v += *reinterpret_cast<const f32x4*>(p.res + ((((long)b * p.Dr + zo) * p.Hr + yo) * p.Wr + xo) * 32 + 16 * ah + 4 * g);
      if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
#if defined(DSM_ZS_ABLATE) && DSM_ZS_ABLATE == 7
      if (p.B == 12345)                       // timing-only build: the epilogue's arithmetic without its stores
#endif
      *reinterpret_cast<f32x4*>(p.y + ((((long)b * p.Do + zo) * p.Ho + yo) * p.Wo + xo) * 32 + 16 * ah + 4 * g) = v;
      track_amax(am, v);
    }
  };

  // ---- first chunk of the range: staged synchronously
  It cur = open_segment(u_begin);
  column_offsets(cur.col);
  {
    const __amdgpu_buffer_rsrc_t rs0 = chunk_rsrc(cur);
    const int xmin0 = chunk_xmin(cur);
#pragma unroll
    for (int k = 0; k < NPF; ++k) pf[k] = buffer_load16(rs0, vx[k] >= xmin0 ? voff[k] : OOBV, 0);
    static_for<0, NPF>([&](auto kc) { convert(kc, lds_raw + st_off); });
  }
  static_for<0, 3>([&](auto s) { zero_set(s); });
  const unsigned w0 = (unsigned)cur.cg * (NSTEP * WSTEP);
  static_for<0, WAHEAD>([&](auto sc_) { wload(sc_, w0); });
  int img = 0;

  while (true) {
    __syncthreads();            // image `img` is complete; everyone is done reading image `img ^ 1`
    const unsigned char* const rd = lds_raw + img * IMG + rd_off;
    unsigned char* const wr = lds_raw + (img ^ 1) * IMG + st_off;
    const It nxt = advance(cur);
    if (nxt.valid && nxt.col != cur.col) column_offsets(nxt.col);
    const __amdgpu_buffer_rsrc_t nrsrc = chunk_rsrc(nxt);
    const int nxmin = chunk_xmin(nxt);
    const unsigned wcur = (unsigned)cur.cg * (NSTEP * WSTEP);
    const unsigned wnext = nxt.valid ? (unsigned)nxt.cg * (NSTEP * WSTEP) : 0u;
    // z-tap kz of input plane zi feeds output plane zi - kz + 1: only inside [z0, z1)
    unsigned mask = 0;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      const int zo = cur.zi - kz + 1;
      if (zo >= cur.z0 && zo < cur.z1) mask |= 1u << kz;
    }
    static_for<0, TY>([&](auto rc) { xload(std::integral_constant<int, 0>{}, rc, rd); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NSTEP>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      constexpr int tp = s / 3, kz = s % 3;
      // unconditional part of the step: weight ring, next tap position's fragments, staging
      if constexpr (s + WAHEAD < NSTEP) wload(std::integral_constant<int, s + WAHEAD>{}, wcur);
      else wload(std::integral_constant<int, s + WAHEAD - NSTEP>{}, wnext);
      if constexpr (tp + 1 < 9) {
        if constexpr (kz < 2)             // the next tap position's TY row fragments, half per step
          static_for<0, TY / 2>([&](auto rc) {
            xload(std::integral_constant<int, tp + 1>{}, std::integral_constant<int, (TY / 2) * kz + decltype(rc)::value>{}, rd);
          });
      }
#if !(defined(DSM_ZS_ABLATE) && DSM_ZS_ABLATE == 3)
      if constexpr (s < NPF) pf[s] = buffer_load16(nrsrc, vx[s] >= nxmin ? voff[s] : OOBV, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
      constexpr int set = 2 - kz;
      if (mask & (1u << kz)) {
#pragma unroll
        for (int r = 0; r < TY; ++r) mma16<PM>(acc[set][r], wq[s % WRING], xq[tp & 1][r]);
      }
      // the operand split of one staged quad of the next chunk, in this step's issue gaps
#if !(defined(DSM_ZS_ABLATE) && DSM_ZS_ABLATE == 3)
      if constexpr (s >= CONV0) convert(std::integral_constant<int, s - CONV0>{}, wr);
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
    if (cur.cg == ncg - 1) {                                    // the plane is complete
      const int zo = cur.zi - 1;
#if defined(DSM_ZS_ABLATE) && DSM_ZS_ABLATE == 1
      if (zo >= cur.z0 && zo < cur.z1 && p.B == 12345) emit(cur, zo);    // timing-only build: no epilogue
#else
      if (zo >= cur.z0 && zo < cur.z1) emit(cur, zo);
#endif
#pragma unroll
      for (int r = 0; r < TY; ++r) { acc[0][r] = acc[1][r]; acc[1][r] = acc[2][r]; }
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
  flush_amax(p.y_amax, am, reinterpret_cast<float*>(lds_raw));
}

// weights: torch (Cout = 32, Cin, 3,3,3) -> [cg][tp = ky*3+kx][kz][a = cout/16][plane NPW][lane][8 x 16-bit]
// lane = 16 kg + m holds A[row m = cout % 16][k = 8 kg + e], k <-> channel 32 cg + 16 (e >> 2) + 4 kg + (e & 3)
template <int PM>
__global__ void pack_weights_zs_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                       const float* __restrict__ w_amax, int Cin) {
  constexpr int NPW = Prec<PM>::NPW;
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
  const float v = w[((long)cout * Cin + cin) * 27 + tap];
  unsigned short* o = out + ((((((long)cg * 9 + tp) * 3 + kz) * 2 + a) * NPW) * 64 + lane) * 8 + e;
  unsigned pl[NPW];
  split_pair<(PM == 3 ? 3 : 2)>(v, 0.f, PM == 3 ? 1.f : dsm_pow2f(dsm_amax_exponent(*w_amax)), pl);
#pragma unroll
  for (int q = 0; q < NPW; ++q) o[(long)q * 64 * 8] = (unsigned short)(pl[q] & 0xffffu);
}

template <int PM, int V = (PM == 3 ? 0 : DSM_ZS_TILING_F16)>
int launch_conv_zs(ZsParams p, int grid, hipStream_t s) {
  using C = ZsCfg<PM, V>;
  p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, C::TY);
  const long ncol = (long)p.B * p.nty * p.ntx;
  DSM_REQUIRE(ncol < (1L << 30), DSM_ERR_UNSUPPORTED);
  p.ncol = (int)ncol;
  p.nunits = ncol * p.Do;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)conv_zs_kernel<PM, V>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS) != hipSuccess)
      return DSM_ERR_LAUNCH;
    configured = true;
  }
  int blocks = grid > 0 ? grid : 256 * C::WGS;                 // persistent workgroups: WGS per CU
  if ((long)blocks > p.nunits) blocks = (int)p.nunits;
  hipLaunchKernelGGL((conv_zs_kernel<PM, V>), dim3(blocks), dim3(NTHREADS), C::LDS, s, p);
  return dsm_launch_status();
}
