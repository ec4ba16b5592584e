// bf16x3 kernels: fp32 convolution and transposed convolution on the bf16 matrix pipe.
// Included by conv3d.hip inside its anonymous namespace, after the shared helpers (ConvParams,
// first_tile, store_tile, buffer_load16, make_rsrc, static_for, DSM_STAMP) -- one translation
// unit, so that the diagnostic stamp buffer and the launch helpers are shared.
#pragma once

// ----------------------------------------------------------------------------
// fp32 convolution on the bf16 matrix pipe ("bf16x3"): Conv3d k3 s1 p1, Cout = 32, Cin % 16 == 0.
//
// The fp32-input MFMA runs at the vector rate (157 TF/s, 1/16 of bf16) and is the wall of this
// path.  Here every fp32 operand is split EXACTLY into three bf16 terms,
//     v = hi + mid + lo,   hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)
// (round-to-nearest; the two subtractions are exact in fp32, so the three terms carry all 24
// bits of v up to a final rounding of 2^-25 |v|), and a product is evaluated as the six largest
// of the nine cross terms
//     w*x ~= wh*xh + wh*xm + wm*xh + wh*xl + wl*xh + wm*xm
// on v_mfma_f32_32x32x16_bf16: every bf16 x bf16 product is exact in the fp32 accumulator, the
// three dropped terms are <= 3 * 2^-25 |w*x| (below the rounding of an fp32 fma chain of this
// length), and accumulation is fp32.  Six MFMAs at 16x the fp32 rate = 2.67x the throughput at
// fp32 accuracy -- parity tests run at the same tolerances as the fp32 kernel.
//
// Structure: workgroup = 4 waves = output tile 1 z x 16 y x 32 x; wave w owns rows 4w..4w+3 (four
// 32x32 accumulators).  A chunk = one z-tap plane x 16 input channels: its (18 x 34)-voxel halo is
// staged global -> VGPR (fp32, buffer loads with zero address VALU as above) -> split -> LDS as
// [voxel][plane 3][16 bf16] at a pitch of 7 x 16 B (conflict-free ds_read_b128 for the 16-lane
// read groups).  Per tap: 3 weight fragments (buffer loads, 2 items ahead, the ring running on
// across chunks) and per row 3 activation fragments (LDS) feed 6 MFMAs.
// One workgroup per CU, one wave per SIMD, TWO LDS images: the next chunk is loaded, split and
// written into the other image from inside this chunk's MFMA stream -- half an element (11 VALU
// + 3 ds_write_b64) per 6-MFMA group, in the wave's own issue gaps (a 32x32x16 MFMA holds vector
// issue for 8 of its 32 cycles).  Measured on the first version (one image, two workgroups per
// CU, split at the commit between two barriers): VALU beside a PARTNER wave's MFMA stream
// issues about once per MFMA -- the commit took 18-21 % of every wave's time -- which is the same
// effect that shaped the fp32 kernel above.  One barrier per chunk is left.
// Weights: pre-split, section 2 of the packed buffer, [Cin/16][dz][tap9][plane][lane][8 bf16].
// ----------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {        // low half = a
  const f32x2 t = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// four fp32 -> three planes of four bf16 (8 bytes each)
__device__ __forceinline__ void split3(const f32x4 v, u32x2 (&pl)[3]) {
  float r0 = v.x, r1 = v.y, r2 = v.z, r3 = v.w;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned a = pack_bf16(r0, r1), b = pack_bf16(r2, r3);
    pl[k].x = a; pl[k].y = b;
    if (k < 2) { r0 -= bf16_lo(a); r1 -= bf16_hi(a); r2 -= bf16_lo(b); r3 -= bf16_hi(b); }
  }
}

// Folded-BN scale / shift of a launch, staged once into LDS behind the two images: the epilogue
// reads them with LDS latency instead of an L2 round trip per tile (stamps: the epilogue was 15 %
// of a 64-channel 2-D layer's in-loop time, most of it waiting for these 2 x COUT floats).
__device__ __forceinline__ void stage_affine_lds(float* aff, const float* scale, const float* shift,
                                                 int cout, int tid) {
  for (int i = tid; i < 2 * cout; i += NTHREADS)
    aff[i] = i < cout ? (scale ? scale[i] : 1.f) : (shift ? shift[i - cout] : 0.f);
}
__device__ __forceinline__ Affine load_affine_lds(const float* aff, int cout, int cbase) {
  Affine a;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    a.sc[g] = *reinterpret_cast<const f32x4*>(aff + cbase + 8 * g);
    a.sh[g] = *reinterpret_cast<const f32x4*>(aff + cout + cbase + 8 * g);
  }
  return a;
}

// NT = Cout / 32; TM = 32x32 accumulator rows per wave (tile height 4 TM); KZ = 3: 3x3x3 on
// volumes, KZ = 1: 3x3 on (B,1,H,W,C) views of NHWC maps; DIL: dilation in (y, x).
// S = 2 (stride 2, DIL = 1, KZ = 3): the halo box is (2 TY + 1) x 65 voxels; its LDS image keeps
// even and odd columns of a row apart (row pitch 66 voxels: 33 even, 33 odd) so that the lanes of
// a fragment read, which step two input columns, stay 112 B apart.
// S3IN: the input arrives in the S3 format of conv_s3.hip (fp32 pre-split into three bf16 planes by
// the layer that produced it): staging is a 16-byte copy, six per voxel and chunk (3 planes x 2
// units), and the operand split -- 16 % of this kernel's time by ablation, plus the bank conflicts
// of its ds_write_b64 -- is gone from the MFMA stream.  A chunk of 16 k-slots is then units
// 2 (ck & 1), 2 (ck & 1) + 1 of channel group ck / 2: k-slot j of lane half h is channel
// 32 (ck / 2) + 16 (j / 4) + 4 (2 (ck & 1) + h) + (j & 3); the weights are packed in that order
// (dsm_conv_pack_weights_s3in).
// Variants whose registers (<= 236 at a forced occupancy of 2, no scratch) and LDS (two exact-size
// images) let TWO workgroups share a CU: the second one computes while the first one's prologue /
// epilogue runs -- what the transposed kernel gained 7-12 % from.  (The 16-row and 128-channel
// variants need one CU's LDS or registers for themselves.)
template <int NT, int TM, int S, bool S3IN, int DIL = 1>
constexpr bool conv_bf16x3_two_per_cu() { return S == 1 && !S3IN && NT <= 2 && TM <= 2 && DIL == 1; }

// NSPLIT > 1 (N-split): the layer has NT * NSPLIT 32-channel output blocks and workgroup column
// blockIdx.y computes NT of them -- a 128-channel 2-D layer with 240 tiles becomes 480 workgroups of
// the 64-channel variant, two per CU, instead of 240 that leave every CU's prologue and epilogue
// exposed (each workgroup stages the whole input tile; the weights are read once either way).
template <int NT, int TM, int KZ, int DIL, int S = 1, bool S3IN = false, int NSPLIT = 1>
__global__ __launch_bounds__(NTHREADS, (conv_bf16x3_two_per_cu<NT, TM, S, S3IN, DIL>() ? 2 : 1))
void conv_bf16x3_kernel(ConvParams p) {
  static_assert(S == 1 || (S == 2 && DIL == 1 && KZ == 3), "stride 2: 3x3x3, no dilation");
  constexpr int TY = 4 * TM, NQ = S3IN ? 6 : 4, CK = 16;
  constexpr int IY = (TY - 1) * S + 2 * DIL + 1, IX = 31 * S + 2 * DIL + 1;
  constexpr int NVOX = IY * IX;
  constexpr int NE = NVOX * NQ;                 // staged 16-B elements per chunk (fp32 quads, or S3 units)
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  constexpr int PITCH = 112;                    // bytes per voxel in LDS: 3 planes x 32 B + 16 pad
  constexpr int RP = (S == 1) ? IX : 66;        // voxels per image row
  constexpr bool TWO = conv_bf16x3_two_per_cu<NT, TM, S, S3IN, DIL>();
  // fp32 input, S = 1: the last pass's tail quads land in padding behind the image -- or, where two
  // workgroups share the CU, are not stored (exact-size image)
  constexpr int IMG = (S == 1 && !S3IN) ? (TWO ? (NVOX + 4) * PITCH : NPF * 64 * PITCH) : (IY * RP + 4) * PITCH;
  constexpr int NITEM = 9;
  constexpr int NGROUP = NITEM * TM;            // (tap, row) groups of 6 NT MFMAs per chunk
  constexpr int AHEAD = 3;                      // weight ring: two items ahead (eight measured the same, twice)
  // staging schedule.  fp32 input, S = 1: one load per group over the first NPF groups, half an
  // element split per group over the last 2 NPF.  S = 2 (ten elements for nine groups): every load
  // in group 0, four halves per group from group 4.  S3 input: LPG loads per group from group 0,
  // the same number of 16-byte stores per group over the last groups.
  constexpr bool SHORT = !S3IN && (S == 2 || NGROUP - 2 * NPF < 2);   // few groups per chunk (4-row tiles, stride 2)
  constexpr int LPG = S3IN ? (NPF + NGROUP / 2 - 1) / (NGROUP / 2) : (SHORT ? NPF : 1);   // loads per group
  constexpr int CONV0 = S3IN ? NGROUP - (NPF + LPG - 1) / LPG
                             : (SHORT ? 4 : NGROUP - 2 * NPF);   // first group that converts / stores
  constexpr int CPG = S3IN ? LPG : (2 * NPF + (NGROUP - CONV0) - 1) / (NGROUP - CONV0);   // halves (stores) per group
  static_assert(NPF <= NGROUP * LPG && CONV0 >= 2, "staging schedule");
  static_assert(!S3IN || CONV0 > (NPF + LPG - 1) / LPG, "a store must come after its load");
  constexpr int NTP = NT * NSPLIT;              // 32-channel output blocks of the layer
  constexpr int COUT = 32 * NTP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nch = p.Cin / CK;
  const int n0 = NSPLIT > 1 ? (int)blockIdx.y * NT : 0;      // first output block of this workgroup column

  int step, end;
  int t = first_tile(p.ntiles, step, end);
  if (t >= end) return;

  // Staging is branch-free: element k of this thread sits at voxel (yy, xx) of the halo box;
  // its byte offset from the box origin is fixed per launch, and a voxel outside the volume is
  // read through the buffer descriptor at an out-of-range offset (hardware returns zeros).
  f32x4 pf[NPF];
  unsigned goff[NPF], yx[NPF];
  const unsigned s3row = S3IN ? (unsigned)(p.Cin >> 5) * 12u * (unsigned)p.Wi : 0u;   // 16-B units per input row y (S3)
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + k * NTHREADS;
    const int v = e / NQ, q = e % NQ;
    const int yy = v / IX, xx = v % IX;
    if constexpr (S3IN)         // unit (plane q / 2, g = q & 1 of the chunk's unit pair) of voxel (yy, xx)
      goff[k] = 16u * ((unsigned)yy * s3row + (unsigned)(4 * (q >> 1) + (q & 1)) * (unsigned)p.Wi + (unsigned)xx);
    else
      goff[k] = 4u * (unsigned)((yy * p.Wi + xx) * p.Cin + 4 * q);
    yx[k] = e < NE ? ((unsigned)yy << 16 | (unsigned)xx) : 0x7fff0000u;   // tail: never in range
  }
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  const unsigned plane_bytes = S3IN ? 16u * s3row * (unsigned)p.Hi
                                    : 4u * (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)p.Cin;

  // (tile, dz, ck) of the chunk being multiplied and of the one being staged
  struct Pos { int t, dz, ck, yb, xb, z; unsigned base; };   // base: byte offset of the box origin in the tile's own z-plane (mod 2^32)
  auto tile_pos = [&](int id) {
    Pos q; q.t = id; q.dz = 0; q.ck = 0;
    q.xb = (id % p.ntx) * 32 * S - DIL; id /= p.ntx;
    q.yb = (id % p.nty) * TY * S - DIL; id /= p.nty;
    q.z = (id % p.Do) * S; const int b = id / p.Do;          // input plane of the centre z-tap
    if constexpr (S3IN)
      q.base = (unsigned)(16l * ((((long)b * p.Di + q.z) * p.Hi + q.yb) * (long)s3row + q.xb));
    else
      q.base = (unsigned)(4l * (((((long)b * p.Di + q.z) * p.Hi + q.yb) * p.Wi + q.xb) * p.Cin));
    return q;
  };
  auto advance = [&](Pos q) {                   // next chunk: ck fastest, then dz, then the tile
    if (++q.ck == nch) { q.ck = 0; if (++q.dz == KZ) q = tile_pos(q.t + step); }
    return q;
  };
  // Per tile: this thread's NPF byte offsets of its voxels in the tile's own z-plane, channel 0
  // (or OOBV where the voxel lies outside the volume).  Per chunk only the descriptor's base
  // moves -- by (dz - KZ/2) planes and ck channel groups, a signed 64-bit scalar add -- so a
  // staged load costs no VALU at all; a chunk whose whole plane is outside the volume gets a
  // descriptor of zero records (every load returns zeros).
  constexpr unsigned OOBV = 0x80000000u;
  unsigned voff[NPF];
  auto tile_offsets = [&](const Pos& q) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int y = q.yb + (int)(yx[k] >> 16), x = q.xb + (int)(yx[k] & 0xffffu);
      const bool ok = (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
      voff[k] = ok ? q.base + goff[k] : OOBV;
    }
  };
  auto chunk_rsrc = [&](const Pos& q, bool live) {
    const long off = (long)(q.dz - KZ / 2) * (long)plane_bytes +
                     (S3IN ? (long)((q.ck >> 1) * 12 + 2 * (q.ck & 1)) * p.Wi * 16 : (long)q.ck * (CK * 4));
    return make_rsrc(reinterpret_cast<const char*>(p.x) + off, live ? p.xbytes : 0u);
  };
  auto live_of = [&](const Pos& q) {            // wave-uniform: a tile exists and its z-tap plane is inside
    const int zin = q.z + q.dz - KZ / 2;
    return q.t < end && zin >= 0 && zin < p.Di;
  };
  // weights: [ck][dz][tap9][n][plane][lane][16 B]
  auto wbase_of = [&](const Pos& q) {
    return (unsigned)((q.ck * KZ + q.dz) * 9) * (NTP * 3 * 64 * 16) + (unsigned)n0 * (3 * 64 * 16);
  };

  // LDS write address of this thread's quad k.  S = 1: voxel (tid >> 2) + 64 k, an immediate
  // per k; S = 2: the even/odd row layout, one register per k.
  const int wr_off = (tid >> 2) * PITCH + (tid & 3) * 8;
  int wofs[(S == 1 && !S3IN) ? 1 : NPF];
  if constexpr (S3IN) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = tid + k * NTHREADS;
      const int v = min(e, NE - 1) / NQ, q = e % NQ, yy = v / IX, xx = v % IX;
      const int vox = (S == 1) ? v : yy * RP + (xx & 1) * 33 + (xx >> 1);
      wofs[k] = e < NE ? vox * PITCH + (q >> 1) * 32 + (q & 1) * 16
                       : (IY * RP) * PITCH + (tid % 28) * 16;    // the tail: spare voxels behind the image
    }
  } else if constexpr (S == 2) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = min(tid + k * NTHREADS, NE - 1);
      const int v = e / NQ, yy = v / IX, xx = v % IX;
      wofs[k] = (tid + k * NTHREADS < NE)
                    ? (yy * RP + (xx & 1) * 33 + (xx >> 1)) * PITCH + (e % NQ) * 8
                    : (IY * RP) * PITCH + (tid & 3) * 8;       // the tail: spare voxels behind the image
    }
  }
  // this lane's activation fragment: voxel (row (TM wave + m) S + dy, column r S + dx), half h
  const int rd_off = ((wave * TM * S) * RP + r) * PITCH + h * 16;

  f32x16 acc[TM][NT];
  const unsigned lane16 = lane * 16u;
  static_assert(NITEM % AHEAD == 0, "continuous weight ring");
  bf16x8 wq[AHEAD][NT][3];
  unsigned half_a[3];                           // first channel pair of the element being split

  // One element (4 channels of one voxel) -> image, split in two halves so that each rides in
  // the gaps of one MFMA group.
  auto convert = [&](auto kc, auto hc, unsigned char* img) {
    constexpr int k = decltype(kc)::value, half = decltype(hc)::value;
    if constexpr (S3IN) {                       // a plain copy: the element is already split
      *reinterpret_cast<f32x4*>(img + wofs[k]) = pf[k];
      return;
    }
    float r0 = half ? pf[k].z : pf[k].x, r1 = half ? pf[k].w : pf[k].y;
    unsigned pl[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      pl[q] = pack_bf16(r0, r1);
      if (q < 2) { r0 -= bf16_lo(pl[q]); r1 -= bf16_hi(pl[q]); }
    }
    if constexpr (half == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) half_a[q] = pl[q];
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        u32x2 v; v.x = half_a[q]; v.y = pl[q];
        if constexpr (S == 1) {
          if (!TWO || k < NPF - 1 || tid < NE - (NPF - 1) * NTHREADS)
            *reinterpret_cast<u32x2*>(img + wr_off + k * (64 * PITCH) + q * 32) = v;
        } else {
          *reinterpret_cast<u32x2*>(img + wofs[k] + q * 32) = v;
        }
      }
    }
  };
  auto wload = [&](auto ic, unsigned wb) {
    constexpr int item = decltype(ic)::value;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        wq[item % AHEAD][n][q] = __builtin_bit_cast(
            bf16x8, buffer_load16(wrsrc, lane16, wb + ((item * NTP + n) * 3 + q) * (64 * 16)));
  };

  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG);
  stage_affine_lds(aff, p.scale, p.shift, COUT, tid);        // visible after the loop's first barrier
  Pos cur_pos = tile_pos(t);
  tile_offsets(cur_pos);
  {                                             // first chunk of the launch: staged synchronously
    const __amdgpu_buffer_rsrc_t rs0 = chunk_rsrc(cur_pos, live_of(cur_pos));
    static_for<0, NPF>([&](auto kc) {
      pf[decltype(kc)::value] = buffer_load16(rs0, voff[decltype(kc)::value], 0);
    });
    static_for<0, AHEAD - 1>([&](auto ic) { wload(ic, wbase_of(cur_pos)); });
    static_for<0, NPF>([&](auto kc) {
      convert(kc, std::integral_constant<int, 0>{}, lds_raw);
      if constexpr (!S3IN) convert(kc, std::integral_constant<int, 1>{}, lds_raw);
    });
  }
  int cur = 0;                                  // image holding the current chunk
  DSM_STAMP_INIT();
  while (true) {
    __syncthreads();            // image `cur` is complete; everyone is done reading image `cur ^ 1`
    DSM_STAMP(0);
    const unsigned char* const rd = lds_raw + cur * IMG + rd_off;
    unsigned char* const nimg = lds_raw + (cur ^ 1) * IMG;
    const Pos nxt = advance(cur_pos);
    if (nxt.dz == 0 && nxt.ck == 0) tile_offsets(nxt);      // the staged chunk opens a new tile
    const __amdgpu_buffer_rsrc_t nrsrc = chunk_rsrc(nxt, live_of(nxt));
    if (cur_pos.dz == 0 && cur_pos.ck == 0) {
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    }
    const unsigned wchunk = wbase_of(cur_pos), wnext = nxt.t < end ? wbase_of(nxt) : wbase_of(cur_pos);
    bf16x8 xq[2][3];
    auto xload = [&](auto sc) {                 // s = item * TM + m
      constexpr int s = decltype(sc)::value;
      constexpr int item = s / TM, m = s % TM;
      constexpr int dy = (item / 3) * DIL, dx = (item % 3) * DIL;
      constexpr int vo = (S == 1) ? (m + dy) * RP + dx : (m * 2 + dy) * RP + (dx & 1) * 33 + (dx >> 1);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        xq[s & 1][q] = *reinterpret_cast<const bf16x8*>(rd + vo * PITCH + q * 32);
    };
    DSM_STAMP(3);
    xload(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    DSM_STAMP(7);
    static_for<0, NITEM>([&](auto ic) {
      constexpr int item = decltype(ic)::value;
      // the weight ring runs on into the next chunk (NITEM % AHEAD == 0 keeps the slots aligned)
      if constexpr (item + AHEAD - 1 < NITEM) wload(std::integral_constant<int, item + AHEAD - 1>{}, wchunk);
      else wload(std::integral_constant<int, item + AHEAD - 1 - NITEM>{}, wnext);
      static_for<0, TM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int s = item * TM + m;
#if !(defined(DSM_ABLATE) && DSM_ABLATE == 4)
        if constexpr (s + 1 < NGROUP) xload(std::integral_constant<int, s + 1>{});
#endif
        // staged loads of the next chunk
#if !(defined(DSM_ABLATE) && DSM_ABLATE == 2)
        static_for<0, LPG>([&](auto jc) {
          constexpr int k = s * LPG + decltype(jc)::value;
          if constexpr (k < NPF) pf[k] = buffer_load16(nrsrc, voff[k], 0);
        });
#endif
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 xh = xq[s & 1][0], xm = xq[s & 1][1], xl = xq[s & 1][2];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const bf16x8 wh = wq[item % AHEAD][n][0], wm = wq[item % AHEAD][n][1], wl = wq[item % AHEAD][n][2];
          // A operand = weights (rows: channels), B operand = activations (columns: voxels);
          // small terms first
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm, acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[m][n], 0, 0, 0);
        }
        // split half an element of the next chunk into the other image, in this group's gaps:
        // element j was requested in group j and is converted in groups CONV0 + 2j, + 2j + 1
#if !(defined(DSM_ABLATE) && DSM_ABLATE == 1)
        static_for<0, CPG>([&](auto jc) {
          constexpr int hidx = (s - CONV0) * CPG + decltype(jc)::value;   // half-element (S3: element) index
          if constexpr (S3IN) {
            if constexpr (s >= CONV0 && hidx < NPF)
              convert(std::integral_constant<int, hidx>{}, std::integral_constant<int, 0>{}, nimg);
          } else if constexpr (s >= CONV0 && hidx < 2 * NPF)
            convert(std::integral_constant<int, hidx / 2>{}, std::integral_constant<int, hidx % 2>{}, nimg);
        });
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    DSM_STAMP(4);
#if defined(DSM_ABLATE) && DSM_ABLATE == 3
    if (cur_pos.dz == KZ - 1 && cur_pos.ck == nch - 1 && p.B == 12345) {
#else
    if (cur_pos.dz == KZ - 1 && cur_pos.ck == nch - 1) {     // epilogue
#endif
      int id = cur_pos.t;
      const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
      const int ty0 = (id % p.nty) * TY; id /= p.nty;
      const int tz = id % p.Do, tb = id / p.Do;
      const int xo = tx0 + r;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int cbase = (n0 + n) * 32 + 4 * h;
        const Affine af = load_affine_lds(aff, COUT, cbase);
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          const int yo = ty0 + wave * TM + m;
          if (yo >= p.Ho || xo >= p.Wo) continue;
          const long vox = (((long)tb * p.Do + tz) * p.Ho + yo) * p.Wo + xo;
          const long rvox = (((long)tb * p.Dr + tz) * p.Hr + yo) * p.Wr + xo;
          if (p.ys3)
            store_tile_s3<COUT>(acc[m][n], af, p.relu, p.y ? p.y + vox * COUT + cbase : nullptr,
                                p.res ? p.res + rvox * COUT + cbase : nullptr,
                                p.ys3 + (((((long)tb * p.Do + tz) * p.Ho + yo) * NTP + n0 + n) * 12) * p.Wo * 16,
                                xo, p.Wo, h);
          else
          store_tile<COUT>(acc[m][n], af, p.relu, p.y + vox * COUT + cbase,
                           p.res ? p.res + rvox * COUT + cbase : nullptr);
        }
      }
    }
    DSM_STAMP(5);
    cur_pos = nxt; cur ^= 1;
    if (cur_pos.t >= end) break;
  }
}

// ----------------------------------------------------------------------------
// ConvTranspose3d(k3, s2, p1, op1) on the bf16 pipe (bf16x3, as conv_bf16x3_kernel above).
// Work item = (input tile 4 rows x 32 columns at depth m, z-parity pz), as in
// deconv3d_mfma_kernel: wave w owns input row w, its four (py, px) output classes are 32x32
// accumulators.  A chunk = one input z-plane (m, or m + 1 for the second z-tap of an odd output
// plane) x 32 input channels: a 5 x 33-voxel box, [voxel][k-group 2][plane 3][16 bf16] at a
// 208-B pitch, two images.  Per chunk 18 (k-group, input offset, class) steps of 6 NT MFMAs:
// offset (0,0) feeds classes {0,1,2,3}, (0,1) {1,3}, (1,0) {2,3}, (1,1) {3}; one activation
// fragment set per offset, one weight fragment set per step (ring, two steps ahead, running on
// across chunks).  Staging and the operand split ride in the MFMA stream exactly as above.
// ----------------------------------------------------------------------------
struct DcPair { int o, c; };
__host__ __device__ constexpr DcPair dc_pair(int j) {
  constexpr int o_of[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
  constexpr int c_of[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};
  return DcPair{o_of[j], c_of[j]};
}
__host__ __device__ constexpr int dc_tap9(int j) {           // ky * 3 + kx of pair j
  const DcPair q = dc_pair(j);
  const int py = q.c >> 1, px = q.c & 1, iy = q.o >> 1, ix = q.o & 1;
  const int ky = py ? (iy ? 0 : 2) : 1, kx = px ? (ix ? 0 : 2) : 1;
  return ky * 3 + kx;
}

#ifndef DSM_DECONV_WGS
#define DSM_DECONV_WGS 2
#endif
template <int NT>
__global__ __launch_bounds__(NTHREADS, NT == 1 ? DSM_DECONV_WGS : 1) void deconv_bf16x3_kernel(ConvParams p) {
  constexpr int TY = 4, IY = TY + 1, IX = 33, CK = 32;
  constexpr int NVOX = IY * IX;                 // 165
  constexpr int NE = NVOX * 8;                  // 1320 staged 16-B fp32 quads per chunk
  constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;   // 6
  constexpr int PITCH = 208;                    // 2 k-groups x 3 planes x 32 B + 16 pad
  constexpr int IMG = NPF * 32 * PITCH;         // 39,936 B
  constexpr int NSTEP = 18;
  constexpr int AHEAD = 3;
  constexpr int CONV0 = NSTEP - 2 * NPF;        // 6
  constexpr unsigned OOBV = 0x80000000u;
  constexpr int COUT = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nck = p.Cin / CK;

  int step, end;
  int t = first_tile(p.ntiles, step, end);
  if (t >= end) return;

  f32x4 pf[NPF];
  unsigned goff[NPF], yx[NPF], voff[NPF];
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + k * NTHREADS;
    const int v = e >> 3, q8 = e & 7;
    const int yy = v / IX, xx = v % IX;
    goff[k] = 4u * (unsigned)((yy * p.Wi + xx) * p.Cin + 4 * q8);
    yx[k] = e < NE ? ((unsigned)yy << 16 | (unsigned)xx) : 0x7fff0000u;
  }
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  const unsigned plane_bytes = 4u * (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)p.Cin;

  struct Pos { int t, pz, iz, ck, yb, xb, m, b; unsigned base; };
  auto item_pos = [&](int id) {
    Pos q; q.t = id; q.pz = id & 1; q.iz = 0; q.ck = 0;
    id >>= 1;
    q.xb = (id % p.ntx) * 32; id /= p.ntx;
    q.yb = (id % p.nty) * TY; id /= p.nty;
    q.m = id % p.Di; q.b = id / p.Di;
    q.base = (unsigned)(4l * (((((long)q.b * p.Di + q.m) * p.Hi + q.yb) * p.Wi + q.xb) * p.Cin));
    return q;
  };
  auto advance = [&](Pos q) {                   // ck fastest, then the z-tap plane, then the item
    if (++q.ck == nck) { q.ck = 0; if (++q.iz > q.pz) q = item_pos(q.t + step); }
    return q;
  };
  auto item_offsets = [&](const Pos& q) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int y = q.yb + (int)(yx[k] >> 16), x = q.xb + (int)(yx[k] & 0xffffu);
      voff[k] = (y < p.Hi && x < p.Wi) ? q.base + goff[k] : OOBV;
    }
  };
  auto chunk_rsrc = [&](const Pos& q) {
    const bool live = q.t < end && q.m + q.iz < p.Di;
    const long off = (long)q.iz * (long)plane_bytes + (long)q.ck * (CK * 4);
    return make_rsrc(reinterpret_cast<const char*>(p.x) + off, live ? p.xbytes : 0u);
  };
  // weights [Cin/16][tap 27][n][plane][lane][16 B]; z-tap of this chunk: pz = 0 -> kz 1; pz = 1 -> kz 2, then 0
  auto wbase_of = [&](const Pos& q) {
    const int kz = q.pz ? (q.iz ? 0 : 2) : 1;
    return (unsigned)((2 * q.ck * 27 + kz * 9) * NT) * (3 * 64 * 16);
  };
  const int wr_off = (tid >> 3) * PITCH + ((tid & 7) >> 2) * 96 + (tid & 3) * 8;
  const int rd_off = (wave * IX + r) * PITCH + h * 16;

  f32x16 acc[4][NT];
  const unsigned lane16 = lane * 16u;
  bf16x8 wq[AHEAD][NT][3];
  unsigned half_a[3];
  auto convert = [&](auto kc, auto hc, unsigned char* img) {
    constexpr int k = decltype(kc)::value, half = decltype(hc)::value;
    float r0 = half ? pf[k].z : pf[k].x, r1 = half ? pf[k].w : pf[k].y;
    unsigned pl[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      pl[q] = pack_bf16(r0, r1);
      if (q < 2) { r0 -= bf16_lo(pl[q]); r1 -= bf16_hi(pl[q]); }
    }
    if constexpr (half == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) half_a[q] = pl[q];
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        u32x2 v; v.x = half_a[q]; v.y = pl[q];
        *reinterpret_cast<u32x2*>(img + wr_off + k * (32 * PITCH) + q * 32) = v;
      }
    }
  };
  // step s = g * 9 + j: k-group g, pair j
  auto wload = [&](auto sc, unsigned wb) {
    constexpr int s = decltype(sc)::value;
    constexpr int g = s / 9, tap9 = dc_tap9(s % 9);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        wq[s % AHEAD][n][q] = __builtin_bit_cast(
            bf16x8, buffer_load16(wrsrc, lane16, wb + (((g * 27 + tap9) * NT + n) * 3 + q) * (64 * 16)));
  };

  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG);
  stage_affine_lds(aff, p.scale, p.shift, COUT, tid);
  Pos cur_pos = item_pos(t);
  item_offsets(cur_pos);
  {
    const __amdgpu_buffer_rsrc_t rs0 = chunk_rsrc(cur_pos);
    static_for<0, NPF>([&](auto kc) {
      pf[decltype(kc)::value] = buffer_load16(rs0, voff[decltype(kc)::value], 0);
    });
    const unsigned w0 = wbase_of(cur_pos);
    static_for<0, AHEAD - 1>([&](auto sc) { wload(sc, w0); });
    static_for<0, NPF>([&](auto kc) {
      convert(kc, std::integral_constant<int, 0>{}, lds_raw);
      convert(kc, std::integral_constant<int, 1>{}, lds_raw);
    });
  }
  int cur = 0;
  while (true) {
    __syncthreads();
    const unsigned char* const rd = lds_raw + cur * IMG + rd_off;
    unsigned char* const nimg = lds_raw + (cur ^ 1) * IMG;
    const Pos nxt = advance(cur_pos);
    if (nxt.iz == 0 && nxt.ck == 0) item_offsets(nxt);
    const __amdgpu_buffer_rsrc_t nrsrc = chunk_rsrc(nxt);
    if (cur_pos.iz == 0 && cur_pos.ck == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[c][n][i] = 0.f;
    }
    const unsigned wchunk = wbase_of(cur_pos), wnext = nxt.t < end ? wbase_of(nxt) : 0u;
    bf16x8 xq[2][3];
    // activation fragment set xi = g * 4 + o (k-group, input offset)
    auto xload = [&](auto xc) {
      constexpr int xi = decltype(xc)::value;
      constexpr int g = xi / 4, o = xi % 4, iy = o >> 1, ix = o & 1;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        xq[xi & 1][q] = *reinterpret_cast<const bf16x8*>(rd + (iy * IX + ix) * PITCH + (g * 3 + q) * 32);
    };
    xload(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NSTEP>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int g = s / 9, j = s % 9;
      constexpr DcPair pr = dc_pair(j);
      constexpr int xi = g * 4 + pr.o;
      constexpr bool fresh = (j == 0 || j == 4 || j == 6 || j == 8);
      if constexpr (s + AHEAD - 1 < NSTEP) wload(std::integral_constant<int, s + AHEAD - 1>{}, wchunk);
      else wload(std::integral_constant<int, s + AHEAD - 1 - NSTEP>{}, wnext);
      if constexpr (fresh && xi + 1 < 8) xload(std::integral_constant<int, xi + 1>{});
      if constexpr (s < NPF) pf[s] = buffer_load16(nrsrc, voff[s], 0);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 xh = xq[xi & 1][0], xm = xq[xi & 1][1], xl = xq[xi & 1][2];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const bf16x8 wh = wq[s % AHEAD][n][0], wm = wq[s % AHEAD][n][1], wl = wq[s % AHEAD][n][2];
        f32x16& a = acc[pr.c][n];
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, a, 0, 0, 0);
      }
      if constexpr (s >= CONV0)
        convert(std::integral_constant<int, (s - CONV0) / 2>{},
                std::integral_constant<int, (s - CONV0) % 2>{}, nimg);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (cur_pos.iz == cur_pos.pz && cur_pos.ck == nck - 1) {   // epilogue of the item
      const int zo = 2 * cur_pos.m + cur_pos.pz;
      const int ym = cur_pos.yb + wave, xm_ = cur_pos.xb + r;  // this lane's input-grid position
      if (zo < p.Do && ym < p.Hi && xm_ < p.Wi) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int cbase = n * 32 + 4 * h;
          const Affine af = load_affine_lds(aff, COUT, cbase);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yo = 2 * ym + (c >> 1), xo = 2 * xm_ + (c & 1);
            if (yo >= p.Ho || xo >= p.Wo) continue;
            const long vox = (((long)cur_pos.b * p.Do + zo) * p.Ho + yo) * p.Wo + xo;
            const long rvox = (((long)cur_pos.b * p.Dr + zo) * p.Hr + yo) * p.Wr + xo;
            if (p.ys3)
              store_tile_s3<COUT>(acc[c][n], af, p.relu, p.y ? p.y + vox * COUT + cbase : nullptr,
                                  p.res ? p.res + rvox * COUT + cbase : nullptr,
                                  p.ys3 + (((((long)cur_pos.b * p.Do + zo) * p.Ho + yo) * NT + n) * 12) * p.Wo * 16,
                                  xo, p.Wo, h);
            else
            store_tile<COUT>(acc[c][n], af, p.relu, p.y + vox * COUT + cbase,
                             p.res ? p.res + rvox * COUT + cbase : nullptr);
          }
        }
      }
    }
    cur_pos = nxt; cur ^= 1;
    if (cur_pos.t >= end) break;
  }
}

// weights -> section 2 of the packed buffer: [Cin/16][tap][Cout/32][plane][lane][8 bf16],
// tap = dz * 9 + t9 (ntaps = 27) or t9 (ntaps = 9)
// s3order: input channels in the k-slot order of an S3-input kernel (see conv_bf16x3_kernel)
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                           int Cin, int Cout, int transposed, int ntaps, int cin_src,
                                           int s3order) {
  const long n = (long)Cin * Cout * ntaps;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int NT = Cout / 32;
  long i = idx;
  const int j = i & 7; i >>= 3;
  const int lane = i & 63; i >>= 6;
  const int n_ = i % NT; i /= NT;
  const int tap = i % ntaps; const int c16 = i / ntaps;
  const int cin = s3order ? 32 * (c16 >> 1) + 16 * (j >> 2) + 4 * (2 * (c16 & 1) + (lane >> 5)) + (j & 3)
                          : 16 * c16 + 8 * (lane >> 5) + j;
  const int cout = 32 * n_ + (lane & 31);
  const long src = transposed ? (((long)cin * Cout + cout) * ntaps + tap)
                              : (((long)cout * cin_src + cin) * ntaps + tap);
  float v = cin < cin_src ? w[src] : 0.f;
  unsigned short* o = out + (((((long)c16 * ntaps + tap) * NT + n_) * 3) * 64 + lane) * 8 + j;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u = pack_bf16(v, 0.f);
    o[(long)q * 64 * 8] = (unsigned short)(u & 0xffffu);
    v -= bf16_lo(u);
  }
}

