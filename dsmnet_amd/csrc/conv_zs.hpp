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
//  * z-sliding: a workgroup owns a (TY y x 32 x) column and walks z.  One input plane is staged
//    ONCE and contributes to three output planes (z-taps 2, 1, 0 -> accumulator sets A0, A1, A2);
//    after a plane, A0 is complete, leaves the registers and the sets rotate.  HBM/L2 traffic for
//    the input falls from 3.6x (three z-taps x 1.2 halo) to 1.33x (the y/x halo only).
//  * the work is the linearised (column, output plane) space cut into gridDim.x EQUAL ranges
//    (one persistent workgroup per CU): no tail round; a range that crosses a column border just
//    starts a new segment.  Partial sums never leave registers: a segment's first and last
//    planes run only the z-taps whose output plane lies inside the segment.
//  * TWO KINDS OF WAVES, one of each per SIMD (512 threads).  gfx950 retires a wave's vector-memory
//    operations -- loads AND stores -- in order through one counter (vmcnt): a wave that mixes the
//    L2-latency weight fragments it needs every step with HBM-latency activation loads, residual
//    loads and output stores waits for the slowest of them at every step (the r03 single-kind kernel:
//    44 % of its wave cycles in s_waitcnt, the same ~90 us lost at six, three or one MFMA per
//    product -- profiles/r03_ablation.md).  So:
//      - MFMA waves (ah, xh): output channels 16 ah .. + 15, columns 16 xh .. + 15, all TY rows,
//        on 16x16x32 MFMAs.  Their only vector-memory traffic is the weight ring (NP fragments per
//        (tap position, z-tap) step from L2, WAHEAD steps ahead); activation fragments come from the
//        LDS image (TY NP ds_read_b128 per tap position, each row's replaced right after its last
//        use), and a finished plane goes to an LDS exchange buffer as raw fp32 accumulators.
//      - staging waves: the next chunk's activations HBM -> registers -> operand split -> LDS image
//        (loads issued a whole chunk before they are split), and the epilogue of the plane the MFMA
//        waves finished a chunk ago: folded BN, residual (prefetched a chunk ahead), ReLU, the
//        tensor maximum, the stores.
//    One barrier per chunk; images and exchange buffers are double-buffered.
//  * staging: a chunk = one input plane x 32 channels, (TY + 2) x 34 voxels x 8 fp32 quads = NPF
//    buffer loads per staging thread (zero address VALU: per-column offsets, per-chunk descriptor
//    base).  Each quad is split into its NP 16-bit planes (conv_split.hpp split_pair: the operand
//    split of this precision mode) and written to the LDS image [plane][g = quad & 3][voxel][16 B]
//    -- the operand order of v_mfma_f32_16x16x32: lane (g, x) of the B operand reads the 16-byte
//    unit (g, voxel x), element e of which is channel 16 (e >> 2) + 4 g + (e & 3) of the group.
//    Rows are 32 NPF units (a multiple of 256 B: conflict-free ds_read_b128 for its 16-lane groups,
//    which span two g rows); a 16-lane group of a ds_write_b64 holds the two halves of ONE unit row
//    for 8 consecutive voxels -- 128 contiguous bytes, conflict-free under the stores' 32-bank rule
//    (MI355X_MICROARCH.md, LDS).  Each voxel is split 1.33 times (once per staging), not once per
//    tap as in the r01 kernel.
//  * `vol`: the input is a concatenation cost volume that is NEVER MATERIALISED: x is the NHWC
//    feature tensor (2B, H, W, C) [left images, then right images]; input plane d of the
//    (B, 2C, D, H, W) volume is staged as [left | right shifted by d voxels] with x < d zeroed
//    (right half always, left half iff vol_mask_left).
#pragma once

#ifndef DSM_ZS_OFF
#define DSM_ZS_OFF 0          // timing-only A/B builds (profiles/r03_ablation.md): bits switch parts of the kernel off
#endif

template <int PM> struct ZsCfg {
  static constexpr int NP = Prec<PM>::NP, NPW = Prec<PM>::NPW;
  static constexpr int THREADS = 512, NSTAGE = 256;   // 4 MFMA waves + 4 staging waves
  static constexpr int TY = PM == 3 ? 6 : 8, IY = TY + 2, IX = 34;   // bf16x3: three planes per image
  static constexpr int NV = IY * IX;                  // 272 | 340 voxels of the halo box
  static constexpr int NPF = (NV * 8 + NSTAGE - 1) / NSTAGE;       // 9 | 11 staged quads per staging thread
  static constexpr int NVP = 32 * NPF;                // units per (plane, g) row: a multiple of 16
  static constexpr int ROW = NVP * 16;                // 4,608 | 5,632 B
  static constexpr int IMG = NP * 4 * ROW;            // 55,296 | 45,056 | 22,528 B
  static constexpr int XCH = TY * 256 * 16;           // 24,576 | 32,768 B: one output plane of the tile, fp32
  static constexpr int LDS = 2 * IMG + 2 * XCH + 256; // two images, two exchange buffers, the folded affine
  static constexpr int NSTEP = 27;
  static constexpr int WSTEP = 2 * NPW * 1024;        // weight bytes per (tap position, z-tap) step
  // weight ring: fragments are requested WAHEAD steps before their use.  A step is TY product groups =
  // 36 (bf16x3) / 24 (f16x2) / 8 (f16) MFMAs of 16 cycles.  The ring has 9 slots (9 divides NSTEP:
  // slot = step % 9 stays consistent across chunks); only WAHEAD + 1 of them are live at a time.
#ifndef DSM_ZS_WAHEAD
#define DSM_ZS_WAHEAD (PM == 3 ? 2 : 4)
#endif
  static constexpr int WRING = 9, WAHEAD = DSM_ZS_WAHEAD;
  static_assert(NSTEP % WRING == 0 && WAHEAD < WRING, "weight ring");
  static_assert(32 * NPF <= NVP && NVP % 16 == 0, "image row");
  static_assert(LDS <= 160 * 1024, "LDS");
};

// the products of one operand pair, small terms first: term t multiplies weight plane ZsTerms::w[t] by
// activation plane ZsTerms::x[t]
template <int PM> struct ZsTerms;
template <> struct ZsTerms<3> { static constexpr int N = 6; static constexpr int w[6] = {1, 2, 0, 1, 0, 0}, x[6] = {1, 0, 2, 0, 1, 0}; };
// (f16x2: the low activation plane first -- its fragments are the ones reloaded earliest, see the step loop)
template <> struct ZsTerms<2> { static constexpr int N = 3; static constexpr int w[3] = {0, 1, 0}, x[3] = {1, 0, 0}; };
template <> struct ZsTerms<1> { static constexpr int N = 1; static constexpr int w[1] = {0}, x[1] = {0}; };

template <int PM>
__device__ __forceinline__ f32x4 zs_mfma(typename Prec<PM>::frag w, typename Prec<PM>::frag x, f32x4 c) {
  if constexpr (PM == 3) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0);
}

template <int PM>
__global__ __launch_bounds__(512, 1) void conv_zs_kernel(ZsParams p) {
  using C = ZsCfg<PM>;
  using T = ZsTerms<PM>;
  using frag = typename Prec<PM>::frag;
  constexpr int NP = C::NP, NPW = C::NPW, TY = C::TY, IX = C::IX, NV = C::NV, NPF = C::NPF, ROW = C::ROW,
                IMG = C::IMG, XCH = C::XCH, NSTEP = C::NSTEP, WSTEP = C::WSTEP, WRING = C::WRING, WAHEAD = C::WAHEAD;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool staging = wave >= 4;             // wave-uniform role
  const int w4 = wave & 3;                    // staging wave w4 finishes what MFMA wave w4 computed
  const int j = lane & 15, g = lane >> 4;
  const int ah = w4 >> 1, xh = w4 & 1;        // 16 output channels (16 ah ..) and 16 columns (16 xh ..), all TY rows
  const int ncg = p.Cin >> 5;

  // this workgroup's range of the linearised (column, output plane) space; workgroups on one XCD
  // (id % 8) take neighbouring ranges (shared halo columns meet in one L2; speed only)
  const int G = gridDim.x, id = blockIdx.x;
  const int logical = (G & 7) == 0 ? (id & 7) * (G >> 3) + (id >> 3) : id;
  const long u_begin = p.nunits * logical / G, u_end = p.nunits * (logical + 1) / G;
  if (u_begin >= u_end) return;

  // power-of-two scaling of the fp16 modes (conv_split.hpp) and the folded-BN affine of this
  // lane's 4 channels (16 ah + 4 g + i), the output factor folded into the scale
  float sx = 1.f, so = 1.f;
  if constexpr (PM != 3) {
    const int ex = dsm_amax_exponent(*p.x_amax), ew = dsm_amax_exponent(*p.w_amax);
    sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dsm_pow2f(ex))));
    so = dsm_pow2f(-(ex + ew));
  }
  unsigned char* const xch = lds_raw + 2 * IMG;
  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG + 2 * XCH);
  if (tid < 64) aff[tid] = tid < 32 ? (p.scale ? p.scale[tid] * so : so) : (p.shift ? p.shift[tid - 32] : 0.f);
  __syncthreads();
  const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + 16 * ah + 4 * g);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + 32 + 16 * ah + 4 * g);
  float am = 0.f;
  const int xch_off = (w4 * 64 + lane) * 16;  // + r * 4096: this thread's (MFMA wave: own; staging wave: its twin's) row r

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
  // the output plane a chunk completes (its last channel group, z-tap 2), or -1
  auto plane_of = [&](const It& q) {
    const int zo = q.zi - 1;
    return (q.valid && q.cg == ncg - 1 && zo >= q.z0 && zo < q.z1) ? zo : -1;
  };
  // folded BN (+ ReLU / skip add) of one accumulator quad, stored; rows beyond the tensor skipped by the caller
  auto finish = [&](f32x4 v, const f32x4 res, float* dst) {
    v = v * sc + sh;
    if (p.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (p.res) v += res;
    if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(dst) = v;
    track_amax(am, v);
  };
  struct Where { int b, y0, xo; };
  auto where = [&](int col) {
    Where q;
    q.b = col / (p.ntx * p.nty);
    q.y0 = ((col / p.ntx) % p.nty) * TY;
    q.xo = (col % p.ntx) * 32 + 16 * xh + j;
    return q;
  };
  auto y_ptr = [&](const Where& q, int zo, int r) {
    return p.y + ((((long)q.b * p.Do + zo) * p.Ho + q.y0 + r) * p.Wo + q.xo) * 32 + 16 * ah + 4 * g;
  };
  auto res_ptr = [&](const Where& q, int zo, int r) {
    return p.res + ((((long)q.b * p.Dr + zo) * p.Hr + q.y0 + r) * p.Wr + q.xo) * 32 + 16 * ah + 4 * g;
  };

  if (staging) {
    // =====================================================================================
    // staging waves: chunk i + 1 -> image (i + 1) & 1 and the epilogue of plane(i - 1), while the
    // MFMA waves run chunk i
    // this thread's quads of a chunk: voxel sv + 32 k, quad sq -- a wave covers 8 voxels x 8 quads
    // (1 KiB of contiguous fp32), its 16-lane groups the two halves of one unit row g
    const unsigned vstride = p.vol ? (unsigned)p.Cin * 2u : (unsigned)p.Cin * 4u;   // bytes per voxel of the staged tensor
    const unsigned plane_bytes = vstride * (unsigned)p.Hi * (unsigned)p.Wi;         // < 2 GiB: checked by the host
    constexpr unsigned OOBV = 0x80000000u;
    const int sq = (lane >> 4) + 4 * (lane & 1), sv = 8 * w4 + ((lane & 15) >> 1);
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
    f32x4 pf[2][NPF];             // chunk k waits in set k & 1 from its request (iteration k - 2) to its split (k - 1)
    auto load_chunk = [&](auto setc, const It& q) {
      constexpr int set = decltype(setc)::value;
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
      const __amdgpu_buffer_rsrc_t rs =
          make_rsrc(reinterpret_cast<const char*>(p.x) + off, q.valid ? plane_bytes + shift : 0u);
      // staged voxels with x below xmin are zeros
      const int xmin = (p.vol && (p.vol_mask_left || 2 * q.cg >= ncg)) ? q.zi : -0x40000000;
#pragma unroll
      for (int k = 0; k < NPF; ++k) pf[set][k] = buffer_load16(rs, vx[k] >= xmin ? voff[k] : OOBV, 0);
    };
    // LDS slot of this thread's quads: unit g = sq & 3, half = sq >> 2, voxel sv + 32 k
    const int st_off = (lane >> 4) * ROW + sv * 16 + (lane & 1) * 8;                 // + 512 k, + plane * 4 ROW
    auto split_chunk = [&](auto setc, unsigned char* wr) {                           // pf[set] -> its NP planes in an image
      constexpr int set = decltype(setc)::value;
#pragma unroll
      for (int k = 0; k < NPF; ++k) {
        unsigned lo[NP], hi[NP];
        split_pair<PM>(pf[set][k].x, pf[set][k].y, sx, lo);
        split_pair<PM>(pf[set][k].z, pf[set][k].w, sx, hi);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          u32x2 v; v.x = lo[q]; v.y = hi[q];
          *reinterpret_cast<u32x2*>(wr + q * 4 * ROW + 512 * k) = v;
        }
      }
    };
    // the plane the MFMA waves left in an exchange buffer a chunk ago
    struct Pend { int col, zo; };
    f32x4 resq[TY];
#pragma unroll
    for (int r = 0; r < TY; ++r) resq[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_res = [&](const Pend& e) {
      if (e.zo < 0 || !p.res) return;
      const Where q = where(e.col);
      if (q.xo >= p.Wo) return;
#pragma unroll
      for (int r = 0; r < TY; ++r)
        if (q.y0 + r < p.Ho) resq[r] = *reinterpret_cast<const f32x4*>(res_ptr(q, e.zo, r));
    };
    auto epilogue = [&](const Pend& e, const unsigned char* src) {
      if (e.zo < 0) return;
      const Where q = where(e.col);
      if (q.xo >= p.Wo) return;
#pragma unroll
      for (int r = 0; r < TY; ++r) {
        if (q.y0 + r >= p.Ho) continue;
#if DSM_ZS_OFF & 1
        if (p.B != 12345) continue;             // timing-only build: no epilogue
#endif
        finish(*reinterpret_cast<const f32x4*>(src + xch_off + r * 4096), resq[r], y_ptr(q, e.zo, r));
      }
    };

    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    It c0 = open_segment(u_begin);
    column_offsets(c0.col);
    load_chunk(S0, c0);
    split_chunk(S0, lds_raw + st_off);
    It c1 = advance(c0);                        // requested, not yet split
    if (c1.valid && c1.col != c0.col) column_offsets(c1.col);
    load_chunk(S1, c1);
    Pend pend = {0, -1};
    int i = 0;
    __syncthreads();                            // image 0 is complete
    // iteration i (parity P = i & 1): c0 = chunk i (on the MFMA waves now), c1 = chunk i + 1 (in set P ^ 1,
    // requested an iteration ago), pend = plane of chunk i - 1 (its residual in resq).  Chunk i + 2 is
    // requested FIRST, into the set chunk i left: it has until the next iteration's split to arrive, and
    // everything this iteration consumes is older than it in the wave's in-order memory queue.
    auto iteration = [&](auto parc) {
      constexpr int P = decltype(parc)::value;
      const It c2 = advance(c1);
#if DSM_ZS_OFF & 8
      if (p.B == 12345) {                       // timing-only build: staging waves only meet the barriers
#endif
      if (c2.valid && c2.col != c1.col) column_offsets(c2.col);
#if !(DSM_ZS_OFF & 4)
      load_chunk(parc, c2);
#endif
      if (c1.valid) split_chunk(std::integral_constant<int, P ^ 1>{}, lds_raw + (P ^ 1) * IMG + st_off);
      epilogue(pend, xch + (P ^ 1) * XCH);
      pend = Pend{c0.col, plane_of(c0)};
      load_res(pend);
#if DSM_ZS_OFF & 8
      }
#endif
      __syncthreads();                          // chunk i done: image P ^ 1 complete, exchange buffer P written
      c0 = c1; c1 = c2; ++i;
      return !c0.valid;
    };
    while (true) {
      if (iteration(S0)) break;
      if (iteration(S1)) break;
    }
    epilogue(pend, xch + ((i - 1) & 1) * XCH);
  } else {
    // =====================================================================================
    // MFMA waves
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
    const unsigned wlane = lane * 16u + (unsigned)ah * (NPW * 1024);
    // activation fragment of this lane: voxel (r + ky, 16 xh + j + kx), unit g, plane q
    const int rd_off = g * ROW + (16 * xh + j) * 16;
    f32x4 acc[3][TY];
    frag xq[TY][NP];              // [row][plane] of the current tap position
    frag wq[WRING][NP];           // [step % WRING][plane]: this wave's 16-channel block only
    auto zero_set = [&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
#pragma unroll
      for (int r = 0; r < TY; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto wload = [&](auto sc_, unsigned wb) {                     // weights of step s (of the chunk at wb)
      constexpr int s = decltype(sc_)::value;
#if DSM_ZS_OFF & 2
      if (p.B != 12345 && s >= 0 && wb != 0xffffffffu) return;   // timing-only build: weight fragments never loaded
#endif
#pragma unroll
      for (int q = 0; q < NP; ++q)
        wq[s % WRING][q] = __builtin_bit_cast(frag, buffer_load16(wrsrc, wlane, wb + s * WSTEP + q * 1024));
    };
    auto xload = [&](auto tpc, auto rc, auto qc, const unsigned char* rd) {
      constexpr int tp = decltype(tpc)::value, r = decltype(rc)::value, q = decltype(qc)::value;
      constexpr int ky = tp / 3, kx = tp % 3;
      xq[r][q] = *reinterpret_cast<const frag*>(rd + q * 4 * ROW + ((r + ky) * IX + kx) * 16);
    };
    // a finished plane that cannot wait for the staging waves (the last plane of a segment that ends
    // at the tensor's last plane): finished here
    auto emit_direct = [&](const It& q, int zo, auto sc_) {
      constexpr int s = decltype(sc_)::value;
      const Where w = where(q.col);
      if (w.xo >= p.Wo) return;
#pragma unroll
      for (int r = 0; r < TY; ++r) {
        if (w.y0 + r >= p.Ho) continue;
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        if (p.res) rv = *reinterpret_cast<const f32x4*>(res_ptr(w, zo, r));
        finish(acc[s][r], rv, y_ptr(w, zo, r));
      }
    };

    // z-tap kz of input plane zi feeds output plane zi - kz + 1; bit kz: that plane lies inside the
    // segment [z0, z1)
    auto zmask_of = [&](const It& q) {
      unsigned m = 0;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const int zo = q.zi - kz + 1;
        if (zo >= q.z0 && zo < q.z1) m |= 1u << kz;
      }
      return m;
    };
    It cur = open_segment(u_begin);
    static_for<0, 3>([&](auto s) { zero_set(s); });
    static_for<0, WAHEAD>([&](auto sc_) { wload(sc_, (unsigned)cur.cg * (NSTEP * WSTEP)); });
    int i = 0;
    __syncthreads();                            // image 0 is complete
    while (true) {
      const unsigned char* const rd = lds_raw + (i & 1) * IMG + rd_off;
      const It nxt = advance(cur);
      const unsigned wcur = (unsigned)cur.cg * (NSTEP * WSTEP);
      const unsigned wnext = nxt.valid ? (unsigned)nxt.cg * (NSTEP * WSTEP) : 0u;
      const unsigned mask = zmask_of(cur);
#if DSM_ZS_OFF & 16
      if (p.B == 12345) {                       // timing-only build: MFMA waves only meet the barriers
#endif
      static_for<0, TY>([&](auto rc) {
        static_for<0, NP>([&](auto qc) { xload(std::integral_constant<int, 0>{}, rc, qc, rd); });
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NSTEP>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        constexpr int tp = s / 3, kz = s % 3;
        constexpr int set = 2 - kz;
        if constexpr (s + WAHEAD < NSTEP) wload(std::integral_constant<int, s + WAHEAD>{}, wcur);
        else wload(std::integral_constant<int, s + WAHEAD - NSTEP>{}, wnext);
        __builtin_amdgcn_sched_barrier(0);
        // term-major: neighbouring MFMAs write different accumulators; each accumulator still sums its
        // terms small-first.  z-taps 0 and 1 are skipped where their output plane is someone else's (a
        // segment's last two planes).  z-tap 2 always runs: at the last z-tap of a tap position an
        // activation plane's fragments are replaced by the next position's right after their last
        // product, one straight-line schedule; where its output plane (zi - 1) lies before the segment
        // (the segment's first two planes) the sums it feeds are rotated out without ever being emitted.
        // (Fetching zero weights for every masked z-tap instead -- no branch at all -- is 3 % slower: the
        // border planes then run all their MFMAs.)
        if (kz == 2 || ((mask >> kz) & 1u)) {
          static_for<0, T::N>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int qw = T::w[t], qx = T::x[t];
            constexpr bool last_use = [] { for (int u = t + 1; u < T::N; ++u) if (T::x[u] == qx) return false; return true; }();
            static_for<0, TY>([&](auto rc) {
              constexpr int r = decltype(rc)::value;
              acc[set][r] = zs_mfma<PM>(wq[s % WRING][qw], xq[r][qx], acc[set][r]);
              if constexpr (kz == 2 && tp + 1 < 9 && last_use && !(DSM_ZS_OFF & 32)) {
                __builtin_amdgcn_sched_barrier(0);
                xload(std::integral_constant<int, tp + 1>{}, rc, std::integral_constant<int, qx>{}, rd);
              }
            });
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
#if DSM_ZS_OFF & 16
      }
#endif
      if (cur.cg == ncg - 1) {                                    // the plane is complete
        const int zo = cur.zi - 1;
        if (zo >= cur.z0 && zo < cur.z1 && (!(DSM_ZS_OFF & 64) || p.B == 12345)) {
#pragma unroll
          for (int r = 0; r < TY; ++r)
            *reinterpret_cast<f32x4*>(xch + (i & 1) * XCH + xch_off + r * 4096) = acc[0][r];
        }
        if (cur.zi == cur.zhi) {                                  // segment ends
          if (cur.zhi >= cur.z0 && cur.zhi < cur.z1)              // only when z1 = Di: no plane Di follows
            emit_direct(cur, cur.zhi, std::integral_constant<int, 1>{});
          static_for<0, 3>([&](auto s) { zero_set(s); });
        } else {
#pragma unroll
          for (int r = 0; r < TY; ++r) { acc[0][r] = acc[1][r]; acc[1][r] = acc[2][r]; }
          zero_set(std::integral_constant<int, 2>{});
        }
      }
      __syncthreads();                          // chunk i done
      cur = nxt; ++i;
      if (!cur.valid) break;
    }
  }
  // ---- the tensor maximum: one atomic per workgroup
  if (p.y_amax) {
#pragma unroll
    for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    float* const red = reinterpret_cast<float*>(lds_raw);
    __syncthreads();
    if (lane == 0) red[wave] = am;
    __syncthreads();
    if (tid == 0) {
      am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
      if (am > __builtin_nontemporal_load(p.y_amax))
        atomicMax(reinterpret_cast<unsigned*>(p.y_amax), __builtin_bit_cast(unsigned, am));
    }
  }
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

template <int PM>
int launch_conv_zs(ZsParams p, int grid, hipStream_t s) {
  using C = ZsCfg<PM>;
  p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, C::TY);
  const long ncol = (long)p.B * p.nty * p.ntx;
  DSM_REQUIRE(ncol < (1L << 30), DSM_ERR_UNSUPPORTED);
  p.ncol = (int)ncol;
  p.nunits = ncol * p.Do;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)conv_zs_kernel<PM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS) != hipSuccess)
      return DSM_ERR_LAUNCH;
    configured = true;
  }
  int blocks = grid > 0 ? grid : 256;                          // persistent workgroups: one per CU
  if ((long)blocks > p.nunits) blocks = (int)p.nunits;
  hipLaunchKernelGGL(conv_zs_kernel<PM>, dim3(blocks), dim3(C::THREADS), C::LDS, s, p);
  return dsm_launch_status();
}
