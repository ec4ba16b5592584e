// Split-operand kernels: fp32 convolution and transposed convolution on the 16-bit matrix pipe.
// Included by conv3d.hip (PM = 3) and conv_f16.hip (PM = 2, 1) after conv_common.hpp.
//
// PM (precision mode) = number of 16-bit terms an fp32 operand is split into:
//   PM = 3  "bf16x3": v = hi + mid + lo in bf16 (exact split), six MFMAs per product -- fp32 accuracy,
//           no scaling needed (bf16 has the fp32 exponent range);
//   PM = 2  "f16x2":  v * 2^e = hi + lo in fp16 (22 significand bits), three MFMAs per product
//           (wl xh + wh xl + wh xh) -- fp32 accuracy at half the MFMAs.  fp16 has a 5-bit exponent,
//           so every tensor is scaled by a power of two chosen from its absolute maximum (activations:
//           a device scalar written by the producing kernel's epilogue, dsm_conv3d_args.x_amax /
//           y_amax; weights: at pack time) so that the maximum lands in [2^12, 2^13); the combined
//           2^-(ex + ew) is folded into the epilogue's scale.  Residuals that fall below fp16's normal
//           range become subnormals (absolute error 2^-25 against a maximum of 2^12), which the MFMA
//           keeps (f16 denormals are not flushed on gfx950: tests/test_f16_gpu.py);
//   PM = 1  "f16":    operands rounded to fp16 (hi only), one MFMA per product, fp32 accumulate --
//           the reduced-precision mode of BASELINE config #5.
#pragma once

#ifndef DSM_STAMP
#define DSM_STAMP(slot) do {} while (0)
#define DSM_STAMP_INIT() do {} while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int PM> struct Prec;
template <> struct Prec<3> { static constexpr int NP = 3, NPW = 3; typedef bf16x8 frag; };
template <> struct Prec<2> { static constexpr int NP = 2, NPW = 2; typedef f16x8 frag; };
template <> struct Prec<1> { static constexpr int NP = 1, NPW = 2; typedef f16x8 frag; };   // reads plane 0 of the f16x2 packing

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {        // low half = a
  const f32x2 t = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// two fp32 values -> their NP planes (one dword per plane, low half = a).  PM < 3: `sx` = 2^ex.
template <int PM>
__device__ __forceinline__ void split_pair(float a, float b, float sx, unsigned (&pl)[Prec<PM>::NP]) {
  if constexpr (PM == 3) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      pl[q] = pack_bf16(a, b);
      if (q < 2) { a -= bf16_lo(pl[q]); b -= bf16_hi(pl[q]); }
    }
  } else {
    // hi = f16(v * sx) straight from the fp32 value (v_fma_mix{lo,hi}_f16: an fp32 fma -- exact, sx is a
    // power of two -- rounded once to fp16), the residual with hi read as an fp16 operand
    // (v_fma_mix_f32): five plain VALU instructions per pair and no packed-fp32 ones, which issue
    // slowly beside an MFMA stream (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
    unsigned hi;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(a), "s"(sx));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(b), "s"(sx));
    pl[0] = hi;
    if constexpr (PM == 2) {
      float r0, r1;
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a), "s"(sx), "v"(hi));
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(b), "s"(sx), "v"(hi));
      const f32x2 r = {r0, r1};
      pl[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
  }
}

// one 32x32x16 product group: c += w * x on PM terms (small terms first)
template <int PM>
__device__ __forceinline__ void mma32(f32x16& c, const typename Prec<PM>::frag (&w)[Prec<PM>::NP],
                                      const typename Prec<PM>::frag (&x)[Prec<PM>::NP]) {
  if constexpr (PM == 3) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], c, 0, 0, 0);
  } else if constexpr (PM == 2) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[0], c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[0], c, 0, 0, 0);
  }
}

// scaling of a launch (PM < 3): sx = 2^ex multiplies the staged activations, `out` = 2^-(ex + ew)
// is folded into the epilogue's scale
struct SplitScale { float sx, out; };
template <int PM>
__device__ __forceinline__ SplitScale split_scale(const ConvParams& p) {
  if constexpr (PM == 3) return SplitScale{1.f, 1.f};
  else {
    const int ex = dsm_amax_exponent(*p.x_amax), ew = dsm_amax_exponent(*p.w_amax);
    return SplitScale{__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dsm_pow2f(ex)))),
                      dsm_pow2f(-(ex + ew))};
  }
}

// Folded-BN scale / shift of a launch, staged once into LDS behind the two images: the epilogue
// reads them with LDS latency instead of an L2 round trip per tile (stamps: the epilogue was 15 %
// of a 64-channel 2-D layer's in-loop time, most of it waiting for these 2 x COUT floats).
// `so`: the launch's output factor 2^-(ex + ew) (1 for PM = 3), folded into the scale here.
__device__ __forceinline__ void stage_affine_lds(float* aff, const float* scale, const float* shift,
                                                 int cout, int tid, float so) {
  for (int i = tid; i < 2 * cout; i += NTHREADS)
    aff[i] = i < cout ? (scale ? scale[i] * so : so) : (shift ? shift[i - cout] : 0.f);
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

// ----------------------------------------------------------------------------
// fp32 convolution on the 16-bit matrix pipe: Conv3d k3 p1 (KZ = 3) / Conv2d 3x3 (KZ = 1, on
// (B,1,H,W,C) views of NHWC maps), Cout = 32 NT NSPLIT, Cin % 16 == 0.
//
// PM = 3 (bf16x3): every fp32 operand is split EXACTLY into three bf16 terms,
//     v = hi + mid + lo,   hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)
// (round-to-nearest; the two subtractions are exact in fp32, so the three terms carry all 24
// bits of v up to a final rounding of 2^-25 |v|), and a product is evaluated as the six largest
// of the nine cross terms  w*x ~= wh*xh + wh*xm + wm*xh + wh*xl + wl*xh + wm*xm
// on v_mfma_f32_32x32x16_bf16: every bf16 x bf16 product is exact in the fp32 accumulator, the
// three dropped terms are <= 3 * 2^-25 |w*x| (below the rounding of an fp32 fma chain of this
// length), and accumulation is fp32.  PM = 2 / 1: the fp16 forms described at the top of the file.
//
// Structure: workgroup = 4 waves = output tile 1 z x 4 TM y x 32 x; wave w owns rows TM w .. TM w +
// TM - 1 (TM 32x32 accumulators per output block).  A chunk = one z-tap plane x 16 input channels:
// its halo is staged global -> VGPR (fp32, buffer loads with zero address VALU) -> split -> LDS as
// [voxel][plane NP][16 x 16-bit] at a pitch of (2 NP + 1) x 16 B (an odd number of 16-byte units:
// conflict-free ds_read_b128 for the 16-lane read groups).  Per tap: NP weight fragments (buffer
// loads, 2 items ahead, the ring running on across chunks) and per row NP activation fragments (LDS)
// feed one product group.
// One workgroup per CU (two where conv_split_two_per_cu says both fit), one wave per SIMD, TWO LDS
// images: the next chunk is loaded, split and written into the other image from inside this
// chunk's MFMA stream -- half an element per product group, in the wave's own issue gaps (a
// 32x32x16 MFMA holds vector issue for 8 of its 32 cycles).  Measured on the first version (one
// image, two workgroups per CU, split at the commit between two barriers): VALU beside a PARTNER
// wave's MFMA stream issues about once per MFMA -- the commit took 18-21 % of every wave's time.
// One barrier per chunk is left.
// Weights: pre-split, [Cin/16][dz][tap9][n][plane NPW][lane][8 x 16-bit].
//
// NT = output blocks per workgroup; TM = accumulator rows per wave (tile height 4 TM); DIL:
// dilation in (y, x).  S = 2 (stride 2, DIL = 1, KZ = 3): the halo box is (2 TY + 1) x 65 voxels;
// its LDS image keeps even and odd columns of a row apart (row pitch 66 voxels: 33 even, 33 odd)
// so that the lanes of a fragment read, which step two input columns, stay one pitch apart.
// NSPLIT > 1 (N-split): the layer has NT * NSPLIT 32-channel output blocks and workgroup column
// blockIdx.y computes NT of them -- a 128-channel 2-D layer with 240 tiles becomes 480 workgroups of
// the 64-channel variant, two per CU, instead of 240 that leave every CU's prologue and epilogue
// exposed (each workgroup stages the whole input tile; the weights are read once either way).
// ----------------------------------------------------------------------------
template <int NT, int TM, int S, int DIL = 1>
constexpr bool conv_split_two_per_cu() { return S == 1 && NT <= 2 && TM <= 2 && DIL == 1; }

template <int PM, int NT, int TM, int KZ, int DIL, int S = 1, int NSPLIT = 1>
struct ConvSplitCfg {
  static constexpr int NP = Prec<PM>::NP, NPW = Prec<PM>::NPW;
  static constexpr int TY = 4 * TM, CK = 16;
  static constexpr int IY = (TY - 1) * S + 2 * DIL + 1, IX = 31 * S + 2 * DIL + 1;
  static constexpr int NVOX = IY * IX;
  static constexpr int NE = NVOX * 4;                // staged 16-B fp32 quads per chunk
  static constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;
  static constexpr int PITCH = 32 * NP + 16;         // bytes per voxel in LDS: NP planes x 32 B + 16 pad
  static constexpr int RP = (S == 1) ? IX : 66;      // voxels per image row
  static constexpr bool TWO = conv_split_two_per_cu<NT, TM, S, DIL>();
  // S = 1: the last pass's tail quads land in padding behind the image -- or, where two workgroups
  // share the CU, are not stored (exact-size image)
  static constexpr int IMG = (S == 1) ? (TWO ? (NVOX + 4) * PITCH : NPF * 64 * PITCH) : (IY * RP + 4) * PITCH;
  static constexpr int NTP = NT * NSPLIT;            // 32-channel output blocks of the layer
  static constexpr int COUT = 32 * NTP;
  static constexpr int LDS = 2 * IMG + 2 * COUT * 4; // two images + scale / shift of the whole layer
  static_assert(!TWO || 2 * LDS <= 160 * 1024, "two workgroups per CU");
  static_assert(NSPLIT == 1 || TWO, "the N-split exists to put two workgroup columns on a CU");
};

template <int PM, int NT, int TM, int KZ, int DIL, int S = 1, int NSPLIT = 1>
__global__ __launch_bounds__(NTHREADS, (conv_split_two_per_cu<NT, TM, S, DIL>() ? 2 : 1))
void conv_split_kernel(ConvParams p) {
  static_assert(S == 1 || (S == 2 && DIL == 1 && KZ == 3), "stride 2: 3x3x3, no dilation");
  using C = ConvSplitCfg<PM, NT, TM, KZ, DIL, S, NSPLIT>;
  using frag = typename Prec<PM>::frag;
  constexpr int NP = C::NP, NPW = C::NPW, TY = C::TY, CK = C::CK, IX = C::IX, NE = C::NE,
                NPF = C::NPF, PITCH = C::PITCH, RP = C::RP, IMG = C::IMG, NTP = C::NTP, COUT = C::COUT;
  constexpr bool TWO = C::TWO;
  constexpr int NITEM = 9;
  constexpr int NGROUP = NITEM * TM;            // (tap, row) product groups per chunk
  constexpr int AHEAD = 3;                      // weight ring: two items ahead (eight measured the same, twice)
  // staging schedule.  S = 1: one load per group over the first NPF groups, half an element split
  // per group over the last 2 NPF.  S = 2 (ten elements for nine groups) and the 4-row tiles: every
  // load in group 0, CPG halves per group from group 4.
  constexpr bool SHORT = S == 2 || NGROUP - 2 * NPF < 2;
  constexpr int LPG = SHORT ? NPF : 1;          // loads per group
  constexpr int CONV0 = SHORT ? 4 : NGROUP - 2 * NPF;      // first group that converts / stores
  constexpr int CPG = (2 * NPF + (NGROUP - CONV0) - 1) / (NGROUP - CONV0);   // halves per group
  static_assert(NPF <= NGROUP * LPG && CONV0 >= 2, "staging schedule");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nch = p.Cin / CK;
  const int n0 = NSPLIT > 1 ? (int)blockIdx.y * NT : 0;      // first output block of this workgroup column

  int step, end;
  int t = first_tile(p.ntiles, step, end);
  if (t >= end) return;
  const SplitScale ss = split_scale<PM>(p);

  // Staging is branch-free: element k of this thread sits at voxel (yy, xx) of the halo box;
  // its byte offset from the box origin is fixed per launch, and a voxel outside the volume is
  // read through the buffer descriptor at an out-of-range offset (hardware returns zeros).
  f32x4 pf[NPF];
  unsigned goff[NPF], yx[NPF];
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + k * NTHREADS;
    const int v = e / 4, q = e % 4;
    const int yy = v / IX, xx = v % IX;
    goff[k] = 4u * (unsigned)((yy * p.Wi + xx) * p.Cin + 4 * q);
    yx[k] = e < NE ? ((unsigned)yy << 16 | (unsigned)xx) : 0x7fff0000u;   // tail: never in range
  }
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  const unsigned plane_bytes = 4u * (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)p.Cin;

  // (tile, dz, ck) of the chunk being multiplied and of the one being staged
  struct Pos { int t, dz, ck, yb, xb, z; unsigned base; };   // base: byte offset of the box origin in the tile's own z-plane (mod 2^32)
  auto tile_pos = [&](int id) {
    Pos q; q.t = id; q.dz = 0; q.ck = 0;
    q.xb = (id % p.ntx) * 32 * S - DIL; id /= p.ntx;
    q.yb = (id % p.nty) * TY * S - DIL; id /= p.nty;
    q.z = (id % p.Do) * S; const int b = id / p.Do;          // input plane of the centre z-tap
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
    const long off = (long)(q.dz - KZ / 2) * (long)plane_bytes + (long)q.ck * (CK * 4);
    return make_rsrc(reinterpret_cast<const char*>(p.x) + off, live ? p.xbytes : 0u);
  };
  auto live_of = [&](const Pos& q) {            // wave-uniform: a tile exists and its z-tap plane is inside
    const int zin = q.z + q.dz - KZ / 2;
    return q.t < end && zin >= 0 && zin < p.Di;
  };
  // weights: [ck][dz][tap9][n][plane][lane][16 B]
  auto wbase_of = [&](const Pos& q) {
    return (unsigned)((q.ck * KZ + q.dz) * 9) * (NTP * NPW * 64 * 16) + (unsigned)n0 * (NPW * 64 * 16);
  };

  // LDS write address of this thread's quad k.  S = 1: voxel (tid >> 2) + 64 k, an immediate
  // per k; S = 2: the even/odd row layout, one register per k.
  const int wr_off = (tid >> 2) * PITCH + (tid & 3) * 8;
  int wofs[S == 1 ? 1 : NPF];
  if constexpr (S == 2) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = min(tid + k * NTHREADS, NE - 1);
      const int v = e / 4, yy = v / IX, xx = v % IX;
      wofs[k] = (tid + k * NTHREADS < NE)
                    ? (yy * RP + (xx & 1) * 33 + (xx >> 1)) * PITCH + (e % 4) * 8
                    : (C::IY * RP) * PITCH + (tid & 3) * 8;    // the tail: spare voxels behind the image
    }
  }
  // this lane's activation fragment: voxel (row (TM wave + m) S + dy, column r S + dx), half h
  const int rd_off = ((wave * TM * S) * RP + r) * PITCH + h * 16;

  f32x16 acc[TM][NT];
  const unsigned lane16 = lane * 16u;
  static_assert(NITEM % AHEAD == 0, "continuous weight ring");
  frag wq[AHEAD][NT][NP];
  unsigned half_a[NP];                          // first channel pair of the element being split

  // One element (4 channels of one voxel) -> image, split in two halves so that each rides in
  // the gaps of one product group.
  auto convert = [&](auto kc, auto hc, unsigned char* img) {
    constexpr int k = decltype(kc)::value, half = decltype(hc)::value;
    unsigned pl[NP];
    split_pair<PM>(half ? pf[k].z : pf[k].x, half ? pf[k].w : pf[k].y, ss.sx, pl);
    if constexpr (half == 0) {
#pragma unroll
      for (int q = 0; q < NP; ++q) half_a[q] = pl[q];
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
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
      for (int q = 0; q < NP; ++q)
        wq[item % AHEAD][n][q] = __builtin_bit_cast(
            frag, buffer_load16(wrsrc, lane16, wb + ((item * NTP + n) * NPW + q) * (64 * 16)));
  };

  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG);
  stage_affine_lds(aff, p.scale, p.shift, COUT, tid, ss.out);   // visible after the loop's first barrier
  float am = 0.f;                               // max |y| of this thread's outputs (y_amax)
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
      convert(kc, std::integral_constant<int, 1>{}, lds_raw);
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
    frag xq[2][NP];
    auto xload = [&](auto sc) {                 // s = item * TM + m
      constexpr int s = decltype(sc)::value;
      constexpr int item = s / TM, m = s % TM;
      constexpr int dy = (item / 3) * DIL, dx = (item % 3) * DIL;
      constexpr int vo = (S == 1) ? (m + dy) * RP + dx : (m * 2 + dy) * RP + (dx & 1) * 33 + (dx >> 1);
#pragma unroll
      for (int q = 0; q < NP; ++q)
        xq[s & 1][q] = *reinterpret_cast<const frag*>(rd + vo * PITCH + q * 32);
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
        // A operand = weights (rows: channels), B operand = activations (columns: voxels)
#pragma unroll
        for (int n = 0; n < NT; ++n) mma32<PM>(acc[m][n], wq[item % AHEAD][n], xq[s & 1]);
        // split half an element of the next chunk into the other image, in this group's gaps:
        // element j was requested in group j and is converted in groups CONV0 + 2j, + 2j + 1
#if !(defined(DSM_ABLATE) && DSM_ABLATE == 1)
        static_for<0, CPG>([&](auto jc) {
          constexpr int hidx = (s - CONV0) * CPG + decltype(jc)::value;   // half-element index
          if constexpr (s >= CONV0 && hidx < 2 * NPF)
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
      // per output block: the skip values of RB rows first, then their stores (conv_common.hpp,
      // load_residual); RB = all rows where the registers allow
      constexpr int RB = (PM == 3 && NT >= 2) ? 1 : TM;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int cbase = (n0 + n) * 32 + 4 * h;
        const Affine af = load_affine_lds(aff, COUT, cbase);
#pragma unroll
        for (int m0 = 0; m0 < TM; m0 += RB) {
          Residual rr[RB];
          if (p.res) {
#pragma unroll
            for (int mi = 0; mi < RB; ++mi) {
              const int yo = ty0 + wave * TM + m0 + mi;
              if (yo >= p.Ho || xo >= p.Wo) continue;
              const long rvox = (((long)tb * p.Dr + tz) * p.Hr + yo) * p.Wr + xo;
              load_residual(rr[mi], p.res + rvox * COUT + cbase);
            }
          }
#pragma unroll
          for (int mi = 0; mi < RB; ++mi) {
            const int yo = ty0 + wave * TM + m0 + mi;
            if (yo >= p.Ho || xo >= p.Wo) continue;
            const long vox = (((long)tb * p.Do + tz) * p.Ho + yo) * p.Wo + xo;
            store_tile<COUT>(acc[m0 + mi][n], af, p.relu, p.y + vox * COUT + cbase, rr[mi], p.res != nullptr, am);
          }
        }
      }
    }
    DSM_STAMP(5);
    cur_pos = nxt; cur ^= 1;
    if (cur_pos.t >= end) break;
  }
  flush_amax(p.y_amax, am, reinterpret_cast<float*>(lds_raw));
}

// ----------------------------------------------------------------------------
// The single-tile form of the 2-D 3x3 stride-1 convolution: conv_once_kernel.
// PSMNet's 1/4-resolution towers (submodule.py:24-46,108-118: 31 Conv2d(64, 64, 3) launches at
// 2 x 96 x 320) give conv_split_kernel ONE round of 240 tiles: a workgroup lives for one tile of
// NCH = Cin / 16 = 4 chunks, each ~0.7 us of MFMAs -- shorter than the memory latency of the next
// chunk's loads, which it requests one chunk ahead.  The tile is then a chain of five exposed
// latencies (first chunk, then one per chunk): 21.8 us per launch for ~6 us of MFMAs.  Here every
// chunk of the tile is requested up front (NCH x NPF loads per thread in flight, NCH known at compile
// time, the chunk loop fully unrolled): in-order return hands chunk 0 over first, chunks 1.. arrive
// while chunk 0 is split and multiplied.  One exposed latency per tile.  Everything else -- LDS images,
// fragment order, weight ring, operand split riding in the MFMA gaps, epilogue, packed weights -- is
// conv_split_kernel's; results are the same bits.
// ----------------------------------------------------------------------------
#ifndef DSM_ONCE_OFF
#define DSM_ONCE_OFF 0          // timing-only A/B builds: 1 no MFMAs, 2 no activation loads, 4 no stores, 8 no weight loads, 16 no split
#endif
template <int PM, int NT, int TM, int NCH, int NSPLIT>
__global__ __launch_bounds__(NTHREADS, 2) void conv_once_kernel(ConvParams p) {
  using C = ConvSplitCfg<PM, NT, TM, 1, 1, 1, NSPLIT>;
  using frag = typename Prec<PM>::frag;
  static_assert(C::TWO, "two workgroups per CU");
  constexpr int NP = C::NP, NPW = C::NPW, TY = C::TY, CK = C::CK, IX = C::IX, NE = C::NE,
                NPF = C::NPF, PITCH = C::PITCH, RP = C::RP, IMG = C::IMG, NTP = C::NTP, COUT = C::COUT;
  constexpr int NITEM = 9, NGROUP = NITEM * TM, AHEAD = 3;
  constexpr int CPG = (2 * NPF + NGROUP - 1) / NGROUP;      // split halves per product group
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = NSPLIT > 1 ? (int)blockIdx.y * NT : 0;
  const int t = blockIdx.x;
  if (t >= p.ntiles) return;
  const SplitScale ss = split_scale<PM>(p);

  int id = t;
  const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
  const int ty0 = (id % p.nty) * TY; id /= p.nty;
  const int tz = id % p.Do, tb = id / p.Do;
  const int yb = ty0 - 1, xb = tx0 - 1;
  const unsigned base = (unsigned)(4l * (((((long)tb * p.Di + tz) * p.Hi + yb) * p.Wi + xb) * p.Cin));
  constexpr unsigned OOBV = 0x80000000u;
  unsigned voff[NPF];
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + k * NTHREADS;
    const int v = e / 4, q = e % 4;
    const int yy = v / IX, xx = v % IX;
    const int y = yb + yy, x = xb + xx;
    const bool ok = e < NE && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    voff[k] = ok ? base + 4u * (unsigned)((yy * p.Wi + xx) * p.Cin + 4 * q) : OOBV;
  }
  // every chunk of the tile, requested now: chunk-major, so that chunk 0 is the first to arrive
  f32x4 pf[NCH][NPF];
  static_for<0, NCH>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(p.x) + c * (CK * 4), p.xbytes);
    static_for<0, NPF>([&](auto kc) {
      if ((DSM_ONCE_OFF & 2) && p.B != 12345) pf[c][decltype(kc)::value] = f32x4{1.f, 2.f, 3.f, 4.f};
      else pf[c][decltype(kc)::value] = buffer_load16(rs, voff[decltype(kc)::value], 0);
    });
  });
  const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
  const int wr_off = (tid >> 2) * PITCH + (tid & 3) * 8;
  const int rd_off = ((wave * TM) * RP + r) * PITCH + h * 16;
  const unsigned lane16 = lane * 16u;
  f32x16 acc[TM][NT];
  frag wq[AHEAD][NT][NP];
  unsigned half_a[NP];
  auto convert = [&](auto cc, auto kc, auto hc, unsigned char* img) {
    constexpr int c = decltype(cc)::value, k = decltype(kc)::value, half = decltype(hc)::value;
    if ((DSM_ONCE_OFF & 16) && p.B != 12345 && k >= 0) return;
    unsigned pl[NP];
    split_pair<PM>(half ? pf[c][k].z : pf[c][k].x, half ? pf[c][k].w : pf[c][k].y, ss.sx, pl);
    if constexpr (half == 0) {
#pragma unroll
      for (int q = 0; q < NP; ++q) half_a[q] = pl[q];
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        u32x2 v; v.x = half_a[q]; v.y = pl[q];
        if (k < NPF - 1 || tid < NE - (NPF - 1) * NTHREADS)
          *reinterpret_cast<u32x2*>(img + wr_off + k * (64 * PITCH) + q * 32) = v;
      }
    }
  };
  auto wload = [&](auto ic, unsigned wb) {
    constexpr int item = decltype(ic)::value;
    if ((DSM_ONCE_OFF & 8) && p.B != 12345 && item >= 0) return;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < NP; ++q)
        wq[item % AHEAD][n][q] = __builtin_bit_cast(
            frag, buffer_load16(wrsrc, lane16, wb + ((item * NTP + n) * NPW + q) * (64 * 16)));
  };
  auto wbase_of = [&](int c) { return (unsigned)(c * 9) * (NTP * NPW * 64 * 16) + (unsigned)n0 * (NPW * 64 * 16); };

  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG);
  stage_affine_lds(aff, p.scale, p.shift, COUT, tid, ss.out);
  float am = 0.f;
  static_for<0, AHEAD - 1>([&](auto ic) { wload(ic, wbase_of(0)); });
  static_for<0, NPF>([&](auto kc) {
    convert(std::integral_constant<int, 0>{}, kc, std::integral_constant<int, 0>{}, lds_raw);
    convert(std::integral_constant<int, 0>{}, kc, std::integral_constant<int, 1>{}, lds_raw);
  });
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  static_for<0, NCH>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    __syncthreads();            // image c & 1 is complete; everyone is done reading the other one
    const unsigned char* const rd = lds_raw + (c & 1) * IMG + rd_off;
    unsigned char* const nimg = lds_raw + ((c + 1) & 1) * IMG;
    const unsigned wchunk = wbase_of(c), wnext = wbase_of(c + 1 < NCH ? c + 1 : c);
    frag xq[2][NP];
    auto xload = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int item = s / TM, m = s % TM;
      constexpr int vo = (m + item / 3) * RP + item % 3;
#pragma unroll
      for (int q = 0; q < NP; ++q)
        xq[s & 1][q] = *reinterpret_cast<const frag*>(rd + vo * PITCH + q * 32);
    };
    xload(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NITEM>([&](auto ic) {
      constexpr int item = decltype(ic)::value;
      if constexpr (item + AHEAD - 1 < NITEM) wload(std::integral_constant<int, item + AHEAD - 1>{}, wchunk);
      else wload(std::integral_constant<int, item + AHEAD - 1 - NITEM>{}, wnext);
      static_for<0, TM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int s = item * TM + m;
        if constexpr (s + 1 < NGROUP) xload(std::integral_constant<int, s + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (!(DSM_ONCE_OFF & 1) || p.B == 12345) mma32<PM>(acc[m][n], wq[item % AHEAD][n], xq[s & 1]);
        // the next chunk (long since requested) is split into the other image in this group's gaps
        if constexpr (c + 1 < NCH) {
          static_for<0, CPG>([&](auto jc) {
            constexpr int hidx = s * CPG + decltype(jc)::value;
            if constexpr (hidx < 2 * NPF)
              convert(std::integral_constant<int, (c + 1 < NCH ? c + 1 : c)>{}, std::integral_constant<int, hidx / 2>{},
                      std::integral_constant<int, hidx % 2>{}, nimg);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  });
  const int xo = tx0 + r;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int cbase = (n0 + n) * 32 + 4 * h;
    const Affine af = load_affine_lds(aff, COUT, cbase);
    Residual rr[TM];            // every row's skip values first, then the stores (conv_common.hpp)
    if (p.res) {
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int yo = ty0 + wave * TM + m;
        if (yo >= p.Ho || xo >= p.Wo) continue;
        const long rvox = (((long)tb * p.Dr + tz) * p.Hr + yo) * p.Wr + xo;
        load_residual(rr[m], p.res + rvox * COUT + cbase);
      }
    }
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const int yo = ty0 + wave * TM + m;
      if (yo >= p.Ho || xo >= p.Wo) continue;
      if ((DSM_ONCE_OFF & 4) && p.B != 12345) continue;
      const long vox = (((long)tb * p.Do + tz) * p.Ho + yo) * p.Wo + xo;
      store_tile<COUT>(acc[m][n], af, p.relu, p.y + vox * COUT + cbase, rr[m], p.res != nullptr, am);
    }
  }
  flush_amax(p.y_amax, am, reinterpret_cast<float*>(lds_raw));
}

// ----------------------------------------------------------------------------
// The stride-2 3x3x3 convolution with two kinds of waves: conv_s2_kernel (fp16 modes).
// conv_split_kernel<S = 2> requests a chunk (one z-tap plane x 16 channels: ~0.7 us of MFMAs) one
// chunk ahead -- less than a memory latency -- and its waves wait for vector memory in order, weights
// behind activations behind stores: MFMA pipe busy 19 %, 85-111 us per launch for 24 us of MFMAs
// (profiles/r03_a_pmc.md).  Same cure as conv_zs.hpp: 512 threads, four MFMA waves (weight ring, LDS
// fragment reads, MFMAs, the tile's epilogue) and four staging waves that keep D = 4 chunks of
// activation loads in flight in their own registers, split one chunk per iteration into the other LDS
// image and never touch a weight.  Image layout, weight layout, tile walk and results are
// conv_split_kernel<PM, 2, 1, 3, 1, 2>'s.
// ----------------------------------------------------------------------------
template <int PM>
__global__ __launch_bounds__(512, 1) void conv_s2_kernel(ConvParams p) {
  constexpr int NT = 2, TM = 1, KZ = 3, S = 2;
  using C = ConvSplitCfg<PM, NT, TM, KZ, 1, S, 1>;
  using frag = typename Prec<PM>::frag;
  constexpr int NP = C::NP, NPW = C::NPW, TY = C::TY, CK = C::CK, IX = C::IX, NE = C::NE,
                NPF = C::NPF, PITCH = C::PITCH, RP = C::RP, IMG = C::IMG, COUT = C::COUT;
  constexpr int NITEM = 9, AHEAD = 3;
  constexpr int D = 4;                          // chunks of activation loads in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool staging = wave >= 4;               // wave-uniform role
  const int stid = tid & 255;
  const int nch = p.Cin / CK;
  int step, end;
  const int t0 = first_tile(p.ntiles, step, end);
  if (t0 >= end) return;
  const int nchunks = ((end - t0 + step - 1) / step) * KZ * nch;     // this workgroup's chunks
  const SplitScale ss = split_scale<PM>(p);

  struct Pos { int t, dz, ck, yb, xb, z; unsigned base; };
  auto tile_pos = [&](int id) {
    Pos q; q.t = id; q.dz = 0; q.ck = 0;
    q.xb = (id % p.ntx) * 32 * S - 1; id /= p.ntx;
    q.yb = (id % p.nty) * TY * S - 1; id /= p.nty;
    q.z = (id % p.Do) * S; const int b = id / p.Do;
    q.base = (unsigned)(4l * (((((long)b * p.Di + q.z) * p.Hi + q.yb) * p.Wi + q.xb) * p.Cin));
    return q;
  };
  auto advance = [&](Pos q) {                   // next chunk: ck fastest, then dz, then the tile
    if (++q.ck == nch) { q.ck = 0; if (++q.dz == KZ) q = tile_pos(q.t + step); }
    return q;
  };
  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG);
  if (tid < NTHREADS) stage_affine_lds(aff, p.scale, p.shift, COUT, tid, ss.out);   // visible after the first barrier
  float am = 0.f;

  if (staging) {
    // ===================================================================== staging waves
    f32x4 pf[D][NPF];
    unsigned goff[NPF], yx[NPF], voff[NPF];
    int wofs[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int e = stid + k * NTHREADS;
      const int v = e / 4, q = e % 4;
      const int yy = v / IX, xx = v % IX;
      goff[k] = 4u * (unsigned)((yy * p.Wi + xx) * p.Cin + 4 * q);
      yx[k] = e < NE ? ((unsigned)yy << 16 | (unsigned)xx) : 0x7fff0000u;   // tail: never in range
      const int ec = min(e, NE - 1);
      const int vc = ec / 4, yc = vc / IX, xc = vc % IX;
      wofs[k] = (e < NE) ? (yc * RP + (xc & 1) * 33 + (xc >> 1)) * PITCH + (ec % 4) * 8
                         : (C::IY * RP) * PITCH + (stid & 3) * 8;           // the tail: spare voxels behind the image
    }
    const unsigned plane_bytes = 4u * (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)p.Cin;
    constexpr unsigned OOBV = 0x80000000u;
    auto tile_offsets = [&](const Pos& q) {
#pragma unroll
      for (int k = 0; k < NPF; ++k) {
        const int y = q.yb + (int)(yx[k] >> 16), x = q.xb + (int)(yx[k] & 0xffffu);
        const bool ok = (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        voff[k] = ok ? q.base + goff[k] : OOBV;
      }
    };
    auto load_chunk = [&](auto setc, const Pos& q, bool valid) {
      constexpr int set = decltype(setc)::value;
      const int zin = q.z + q.dz - 1;
      const bool live = valid && zin >= 0 && zin < p.Di;
      const long off = (long)(q.dz - 1) * (long)plane_bytes + (long)q.ck * (CK * 4);
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(p.x) + off, live ? p.xbytes : 0u);
#pragma unroll
      for (int k = 0; k < NPF; ++k) pf[set][k] = buffer_load16(rs, voff[k], 0);
    };
    auto split_chunk = [&](auto setc, unsigned char* img) {
      constexpr int set = decltype(setc)::value;
#pragma unroll
      for (int k = 0; k < NPF; ++k) {
        unsigned lo[NP], hi[NP];
        split_pair<PM>(pf[set][k].x, pf[set][k].y, ss.sx, lo);
        split_pair<PM>(pf[set][k].z, pf[set][k].w, ss.sx, hi);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          u32x2 v; v.x = lo[q]; v.y = hi[q];
          *reinterpret_cast<u32x2*>(img + wofs[k] + q * 32) = v;
        }
      }
    };
    // chunk k waits in register set k % D from its request (iteration k - D) to its split (k - 1)
    Pos lead = tile_pos(t0);                    // the chunk requested last
    int nlead = 0;                              // its index
    tile_offsets(lead);
    load_chunk(std::integral_constant<int, 0>{}, lead, true);
    split_chunk(std::integral_constant<int, 0>{}, lds_raw);
    static_for<1, D>([&](auto kc) {
      const Pos nx = advance(lead);
      ++nlead;
      if (nlead < nchunks && nx.dz == 0 && nx.ck == 0) tile_offsets(nx);
      lead = nx;
      load_chunk(kc, lead, nlead < nchunks);
    });
    int i = 0;
    __syncthreads();                            // image 0 is complete
    // iteration i (P = i % D): chunk i is on the MFMA waves; request chunk i + D into the set chunk i
    // left, split chunk i + 1 (requested D - 1 iterations ago) into the other image
    auto iteration = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
      const Pos nx = advance(lead);
      ++nlead;
      if (nlead < nchunks && nx.dz == 0 && nx.ck == 0) tile_offsets(nx);
      lead = nx;
      load_chunk(pc, lead, nlead < nchunks);
      if (i + 1 < nchunks) split_chunk(std::integral_constant<int, (P + 1) % D>{}, lds_raw + ((i + 1) & 1) * IMG);
      __syncthreads();                          // chunk i done; image (i + 1) & 1 complete
      ++i;
      return i >= nchunks;
    };
    while (true) {
      if (iteration(std::integral_constant<int, 0>{})) break;
      if (iteration(std::integral_constant<int, 1>{})) break;
      if (iteration(std::integral_constant<int, 2>{})) break;
      if (iteration(std::integral_constant<int, 3>{})) break;
    }
  } else {
    // ===================================================================== MFMA waves
    const int r = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.wbytes);
    auto wbase_of = [&](const Pos& q) { return (unsigned)((q.ck * KZ + q.dz) * 9) * (NT * NPW * 64 * 16); };
    const int rd_off = ((wave * TM * S) * RP + r) * PITCH + h * 16;
    f32x16 acc[NT];
    const unsigned lane16 = lane * 16u;
    static_assert(NITEM % AHEAD == 0, "continuous weight ring");
    frag wq[AHEAD][NT][NP];
    auto wload = [&](auto ic, unsigned wb) {
      constexpr int item = decltype(ic)::value;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < NP; ++q)
          wq[item % AHEAD][n][q] = __builtin_bit_cast(
              frag, buffer_load16(wrsrc, lane16, wb + ((item * NT + n) * NPW + q) * (64 * 16)));
    };
    Pos cur_pos = tile_pos(t0);
    static_for<0, AHEAD - 1>([&](auto ic) { wload(ic, wbase_of(cur_pos)); });
    int i = 0;
    __syncthreads();                            // image 0 is complete
    while (true) {
      const unsigned char* const rd = lds_raw + (i & 1) * IMG + rd_off;
      const Pos nxt = advance(cur_pos);
      if (cur_pos.dz == 0 && cur_pos.ck == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[n][k] = 0.f;
      }
      const unsigned wchunk = wbase_of(cur_pos), wnext = i + 1 < nchunks ? wbase_of(nxt) : wbase_of(cur_pos);
      frag xq[2][NP];
      auto xload = [&](auto sc) {
        constexpr int item = decltype(sc)::value;
        constexpr int dy = item / 3, dx = item % 3;
        constexpr int vo = dy * RP + (dx & 1) * 33 + (dx >> 1);
#pragma unroll
        for (int q = 0; q < NP; ++q)
          xq[item & 1][q] = *reinterpret_cast<const frag*>(rd + vo * PITCH + q * 32);
      };
      xload(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NITEM>([&](auto ic) {
        constexpr int item = decltype(ic)::value;
        if constexpr (item + AHEAD - 1 < NITEM) wload(std::integral_constant<int, item + AHEAD - 1>{}, wchunk);
        else wload(std::integral_constant<int, item + AHEAD - 1 - NITEM>{}, wnext);
        if constexpr (item + 1 < NITEM) xload(std::integral_constant<int, item + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n) mma32<PM>(acc[n], wq[item % AHEAD][n], xq[item & 1]);
        __builtin_amdgcn_sched_barrier(0);
      });
      if (cur_pos.dz == KZ - 1 && cur_pos.ck == nch - 1) {     // epilogue of the tile
        int id = cur_pos.t;
        const int tx0 = (id % p.ntx) * 32; id /= p.ntx;
        const int ty0 = (id % p.nty) * TY; id /= p.nty;
        const int tz = id % p.Do, tb = id / p.Do;
        const int xo = tx0 + r, yo = ty0 + wave;
        if (yo < p.Ho && xo < p.Wo) {
          const long vox = (((long)tb * p.Do + tz) * p.Ho + yo) * p.Wo + xo;
          const long rvox = (((long)tb * p.Dr + tz) * p.Hr + yo) * p.Wr + xo;
          Residual rr[NT];
          if (p.res) {
#pragma unroll
            for (int n = 0; n < NT; ++n) load_residual(rr[n], p.res + rvox * COUT + n * 32 + 4 * h);
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int cbase = n * 32 + 4 * h;
            const Affine af = load_affine_lds(aff, COUT, cbase);
            store_tile<COUT>(acc[n], af, p.relu, p.y + vox * COUT + cbase, rr[n], p.res != nullptr, am);
          }
        }
      }
      __syncthreads();                          // chunk i done
      cur_pos = nxt; ++i;
      if (i >= nchunks) break;
    }
  }
  flush_amax8(p.y_amax, am, reinterpret_cast<float*>(lds_raw));
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
template <int PM, int NT>
struct DeconvSplitCfg {
  static constexpr int NP = Prec<PM>::NP, NPW = Prec<PM>::NPW;
  static constexpr int TY = 4, IY = TY + 1, IX = 33, CK = 32;
  static constexpr int NVOX = IY * IX;                 // 165
  static constexpr int NE = NVOX * 8;                  // 1320 staged 16-B fp32 quads per chunk
  static constexpr int NPF = (NE + NTHREADS - 1) / NTHREADS;   // 6
  static constexpr int PITCH = 2 * NP * 32 + 16;       // 2 k-groups x NP planes x 32 B + 16 pad
  static constexpr int IMG = NPF * 32 * PITCH;         // 39,936 B (PM = 3)
  static constexpr int COUT = 32 * NT;
  static constexpr int LDS = 2 * IMG + 2 * COUT * 4;
  // NT = 1 fits two workgroups per CU (PM = 3: 198 registers, 2 x 80 KB of LDS): the second one
  // computes while the first one's four-class epilogue (stores, skip read) drains
  static constexpr int WGS = NT == 1 ? DSM_DECONV_WGS : 1;
};

template <int PM, int NT>
__global__ __launch_bounds__(NTHREADS, (DeconvSplitCfg<PM, NT>::WGS)) void deconv_split_kernel(ConvParams p) {
  using C = DeconvSplitCfg<PM, NT>;
  using frag = typename Prec<PM>::frag;
  constexpr int NP = C::NP, NPW = C::NPW, TY = C::TY, IX = C::IX, CK = C::CK, NE = C::NE, NPF = C::NPF,
                PITCH = C::PITCH, IMG = C::IMG, COUT = C::COUT;
  constexpr int NSTEP = 18;
  constexpr int AHEAD = 3;
  constexpr int CONV0 = NSTEP - 2 * NPF;        // 6
  constexpr unsigned OOBV = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nck = p.Cin / CK;

  int step, end;
  int t = first_tile(p.ntiles, step, end);
  if (t >= end) return;
  const SplitScale ss = split_scale<PM>(p);

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
    return (unsigned)((2 * q.ck * 27 + kz * 9) * NT) * (NPW * 64 * 16);
  };
  const int wr_off = (tid >> 3) * PITCH + ((tid & 7) >> 2) * (NP * 32) + (tid & 3) * 8;
  const int rd_off = (wave * IX + r) * PITCH + h * 16;

  f32x16 acc[4][NT];
  const unsigned lane16 = lane * 16u;
  frag wq[AHEAD][NT][NP];
  unsigned half_a[NP];
  auto convert = [&](auto kc, auto hc, unsigned char* img) {
    constexpr int k = decltype(kc)::value, half = decltype(hc)::value;
    unsigned pl[NP];
    split_pair<PM>(half ? pf[k].z : pf[k].x, half ? pf[k].w : pf[k].y, ss.sx, pl);
    if constexpr (half == 0) {
#pragma unroll
      for (int q = 0; q < NP; ++q) half_a[q] = pl[q];
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
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
      for (int q = 0; q < NP; ++q)
        wq[s % AHEAD][n][q] = __builtin_bit_cast(
            frag, buffer_load16(wrsrc, lane16, wb + (((g * 27 + tap9) * NT + n) * NPW + q) * (64 * 16)));
  };

  float* const aff = reinterpret_cast<float*>(lds_raw + 2 * IMG);
  stage_affine_lds(aff, p.scale, p.shift, COUT, tid, ss.out);
  float am = 0.f;
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
    frag xq[2][NP];
    // activation fragment set xi = g * 4 + o (k-group, input offset)
    auto xload = [&](auto xc) {
      constexpr int xi = decltype(xc)::value;
      constexpr int g = xi / 4, o = xi % 4, iy = o >> 1, ix = o & 1;
#pragma unroll
      for (int q = 0; q < NP; ++q)
        xq[xi & 1][q] = *reinterpret_cast<const frag*>(rd + (iy * IX + ix) * PITCH + (g * NP + q) * 32);
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
#pragma unroll
      for (int n = 0; n < NT; ++n) mma32<PM>(acc[pr.c][n], wq[s % AHEAD][n], xq[xi & 1]);
      if constexpr (s >= CONV0)
        convert(std::integral_constant<int, (s - CONV0) / 2>{},
                std::integral_constant<int, (s - CONV0) % 2>{}, nimg);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (cur_pos.iz == cur_pos.pz && cur_pos.ck == nck - 1) {   // epilogue of the item
      const int zo = 2 * cur_pos.m + cur_pos.pz;
      const int ym = cur_pos.yb + wave, xm_ = cur_pos.xb + r;  // this lane's input-grid position
      if (zo < p.Do && ym < p.Hi && xm_ < p.Wi) {
        // the skip values of CB classes first, then their stores (conv_common.hpp, load_residual); CB = all
        // four where the registers allow (NT = 1)
        constexpr int CB = NT == 1 ? 4 : 2;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int cbase = n * 32 + 4 * h;
          const Affine af = load_affine_lds(aff, COUT, cbase);
#pragma unroll
          for (int c0 = 0; c0 < 4; c0 += CB) {
            Residual rr[CB];
            if (p.res) {
#pragma unroll
              for (int ci = 0; ci < CB; ++ci) {
                const int c = c0 + ci;
                const int yo = 2 * ym + (c >> 1), xo = 2 * xm_ + (c & 1);
                if (yo >= p.Ho || xo >= p.Wo) continue;
                const long rvox = (((long)cur_pos.b * p.Dr + zo) * p.Hr + yo) * p.Wr + xo;
                load_residual(rr[ci], p.res + rvox * COUT + cbase);
              }
            }
#pragma unroll
            for (int ci = 0; ci < CB; ++ci) {
              const int c = c0 + ci;
              const int yo = 2 * ym + (c >> 1), xo = 2 * xm_ + (c & 1);
              if (yo >= p.Ho || xo >= p.Wo) continue;
              const long vox = (((long)cur_pos.b * p.Do + zo) * p.Ho + yo) * p.Wo + xo;
              store_tile<COUT>(acc[c][n], af, p.relu, p.y + vox * COUT + cbase, rr[ci], p.res != nullptr, am);
            }
          }
        }
      }
    }
    cur_pos = nxt; cur ^= 1;
    if (cur_pos.t >= end) break;
  }
  flush_amax(p.y_amax, am, reinterpret_cast<float*>(lds_raw));
}

// weights -> the split section of a packed buffer: [Cin/16][tap][Cout/32][plane NPW][lane][8 x 16-bit],
// tap = dz * 9 + t9 (ntaps = 27) or t9 (ntaps = 9).  PM = 3: three bf16 planes; PM = 2: two fp16
// planes of w * 2^ew, ew from the header's absolute maximum (`w_amax`, written by absmax_kernel).
template <int PM>
__global__ void pack_weights_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                          const float* __restrict__ w_amax, int Cin, int Cout,
                                          int transposed, int ntaps, int cin_src) {
  constexpr int NPW = Prec<PM>::NPW;
  const long n = (long)Cin * Cout * ntaps;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int NT = Cout / 32;
  long i = idx;
  const int j = i & 7; i >>= 3;
  const int lane = i & 63; i >>= 6;
  const int n_ = i % NT; i /= NT;
  const int tap = i % ntaps; const int c16 = i / ntaps;
  const int cin = 16 * c16 + 8 * (lane >> 5) + j;
  const int cout = 32 * n_ + (lane & 31);
  const long src = transposed ? (((long)cin * Cout + cout) * ntaps + tap)
                              : (((long)cout * cin_src + cin) * ntaps + tap);
  const float v = cin < cin_src ? w[src] : 0.f;
  unsigned short* o = out + (((((long)c16 * ntaps + tap) * NT + n_) * NPW) * 64 + lane) * 8 + j;
  unsigned pl[NPW];
  split_pair<(PM == 3 ? 3 : 2)>(v, 0.f, PM == 3 ? 1.f : dsm_pow2f(dsm_amax_exponent(*w_amax)), pl);
#pragma unroll
  for (int q = 0; q < NPW; ++q) o[(long)q * 64 * 8] = (unsigned short)(pl[q] & 0xffffu);
}

// absolute maximum of a buffer into *out (atomic max of the float bits; *out zeroed by the caller)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  float am = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    track_amax(am, reinterpret_cast<const f32x4*>(x)[i]);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) am = fmaxf(am, fabsf(x[(n4 << 2) + threadIdx.x]));
  __shared__ float red[4];
  flush_amax(out, am, red);
}

// ---------------------------------------------------------------------------- launch helpers
template <int PM, int NT, int TM, int KZ, int DIL, int S = 1, int NSPLIT = 1>
int run_conv_split(ConvParams p, hipStream_t s) {
  using C = ConvSplitCfg<PM, NT, TM, KZ, DIL, S, NSPLIT>;
  p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, C::TY);
  const long nt = (long)p.B * p.Do * p.nty * p.ntx;
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  // two images; one workgroup per CU, or two where conv_split_two_per_cu says both fit
  return launch_tiles(conv_split_kernel<PM, NT, TM, KZ, DIL, S, NSPLIT>, p, C::LDS, s,
                      (C::TWO ? 512 : 256) / NSPLIT, NSPLIT);
}

template <int PM, int NT, int TM, int NCH, int NSPLIT>
int run_conv_once(ConvParams p, hipStream_t s) {
  using C = ConvSplitCfg<PM, NT, TM, 1, 1, 1, NSPLIT>;
  p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, C::TY);
  const long nt = (long)p.B * p.Do * p.nty * p.ntx;
  if (nt >= (1L << 30) || p.Cin != 16 * NCH) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  static_assert(C::LDS <= 64 * 1024, "no dynamic-LDS attribute needed");
  hipLaunchKernelGGL((conv_once_kernel<PM, NT, TM, NCH, NSPLIT>), dim3((unsigned)nt, NSPLIT), dim3(NTHREADS), C::LDS, s, p);
  return dsm_launch_status();
}

template <int PM>
int run_conv_s2(ConvParams p, hipStream_t s) {
  using C = ConvSplitCfg<PM, 2, 1, 3, 1, 2, 1>;
  p.ntx = dsm_cdiv(p.Wo, 32); p.nty = dsm_cdiv(p.Ho, C::TY);
  const long nt = (long)p.B * p.Do * p.nty * p.ntx;
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  static thread_local bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)conv_s2_kernel<PM>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
      return DSM_ERR_LAUNCH;
    configured = true;
  }
  int blocks = p.force_blocks ? p.force_blocks : 256;          // one workgroup per CU
  if (blocks > p.ntiles) blocks = p.ntiles;
  if (blocks >= 8) blocks &= ~7;                               // whole rounds over the 8 XCDs
  hipLaunchKernelGGL(conv_s2_kernel<PM>, dim3(blocks), dim3(512), C::LDS, s, p);
  return dsm_launch_status();
}

template <int PM, int NT>
int run_deconv_split(ConvParams p, hipStream_t s) {
  using C = DeconvSplitCfg<PM, NT>;
  p.ntx = dsm_cdiv(p.Wi, 32); p.nty = dsm_cdiv(p.Hi, 4);
  const long nt = (long)p.B * p.Di * p.nty * p.ntx * 2;       // x2: z-parity
  if (nt >= (1L << 30)) return DSM_ERR_UNSUPPORTED;
  p.ntiles = (int)nt;
  return launch_tiles(deconv_split_kernel<PM, NT>, p, C::LDS, s, 256 * C::WGS);
}

// plan kinds 5 (convolution) and 6 (transposed convolution) at precision mode PM
template <int PM>
int dispatch_split(const Plan& pl, const ConvParams& p, hipStream_t s) {
  if (pl.kind == 6) return pl.NT == 1 ? run_deconv_split<PM, 1>(p, s) : run_deconv_split<PM, 2>(p, s);
  if (pl.kind != 5) return DSM_ERR_UNSUPPORTED;
  if (pl.S == 2) {
    if constexpr (PM != 3) { if (!p.single_kind) return run_conv_s2<PM>(p, s); }
    return run_conv_split<PM, 2, 1, 3, 1, 2>(p, s);
  }
  if constexpr (PM != 3) {                      // (bf16x3's three-plane images do not leave room)
    if (pl.once) {                              // one round of tiles, every chunk requested up front
      if (pl.KZ == 1 && pl.NT == 1 && pl.TM == 2 && pl.DIL == 1 && pl.nsplit == 2 && p.Cin == 64)
        return run_conv_once<PM, 1, 2, 4, 2>(p, s);
      return DSM_ERR_UNSUPPORTED;
    }
  }
  if (pl.nsplit == 2) {                         // 2-D layers of 64 / 128 channels in two workgroup columns
    if (pl.KZ == 1 && pl.NT == 1 && pl.TM == 2 && pl.DIL == 1) return run_conv_split<PM, 1, 2, 1, 1, 1, 2>(p, s);
    if (pl.KZ == 1 && pl.NT == 2 && pl.TM == 2 && pl.DIL == 1) return run_conv_split<PM, 2, 2, 1, 1, 1, 2>(p, s);
    if (pl.KZ == 3 && pl.NT == 1 && pl.TM == 1) return run_conv_split<PM, 1, 1, 3, 1, 1, 2>(p, s);
    return DSM_ERR_UNSUPPORTED;
  }
  if (pl.nsplit == 4) {                         // 3-D, 128 output channels on a small volume
    if (pl.KZ == 3 && pl.NT == 1 && pl.TM == 1) return run_conv_split<PM, 1, 1, 3, 1, 1, 4>(p, s);
    return DSM_ERR_UNSUPPORTED;
  }
#define DSM_CASE_SPLIT(NT_, TM_, KZ_, DIL_) \
  if (pl.NT == NT_ && pl.TM == TM_ && pl.KZ == KZ_ && pl.DIL == DIL_) return run_conv_split<PM, NT_, TM_, KZ_, DIL_>(p, s)
  DSM_CASE_SPLIT(1, 4, 3, 1); DSM_CASE_SPLIT(1, 2, 3, 1); DSM_CASE_SPLIT(2, 2, 3, 1); DSM_CASE_SPLIT(2, 1, 3, 1);
  DSM_CASE_SPLIT(1, 4, 1, 1); DSM_CASE_SPLIT(1, 2, 1, 1); DSM_CASE_SPLIT(2, 2, 1, 1); DSM_CASE_SPLIT(2, 1, 1, 1);
  DSM_CASE_SPLIT(4, 2, 1, 1);
  DSM_CASE_SPLIT(4, 2, 1, 2);
#undef DSM_CASE_SPLIT
  return DSM_ERR_UNSUPPORTED;
}
