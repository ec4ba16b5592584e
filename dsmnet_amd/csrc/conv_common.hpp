// Shared pieces of the convolution translation units (conv3d.hip: fp32-input MFMA + bf16x3
// kernels; conv_f16.hip: the fp16 split kernels): tile walk, epilogue, buffer loads, launch helper.
#pragma once
#include "common.hpp"
#include <type_traits>

namespace dsmk {

struct ConvParams {
  const float* x; const float* w; const float* scale; const float* shift;
  const float* res; float* y;
  int force_blocks;           // 0, or the persistent grid size asked for in dsm_conv3d_args.flags
  int single_kind;            // dsm_conv3d_args.flags & DSM_CONV_NO_ONCE: the chunk-pipelined single-kind kernels (A/B runs)
  int B, Cin, Cout;
  int Di, Hi, Wi, Do, Ho, Wo, Dr, Hr, Wr;
  int relu;
  int ntx, nty, ntiles;       // tile grid: x tiles, y tiles, total = B*Do*nty*ntx (x8 classes for deconv)
  unsigned xbytes, wbytes;    // extents of x and w for the buffer descriptors (< 4 GiB)
  int ntp;                    // Cout / 32 of the layer (packing and output stride); a launch may
                              // compute only NT of them per workgroup column (blockIdx.y: N-split)
  // fp16 split kernels (conv_split.hpp, PM < 3): absolute maxima of the input tensor (device scalar
  // written by its producer) and of the weights (header of the packed f16 section)
  const float* x_amax; const float* w_amax;
  float* y_amax;              // or null: max |y| of this launch is folded in (atomic max of the float bits)
};

// One place decides the kernel variant; dsm_conv3d_fwd launches it, dsm_conv3d_plan names it.
// kind: 0 conv, 1 deconv, 2 conv cout1, 3 deconv cout1, 4 conv cout1 z-sliding, 5 conv split (bf16x3 / f16x2 / f16), 6 deconv split,
//       7 z-sliding conv (Cout = 32, stride 1; conv_zs.hpp)
struct Plan { int kind; int S, NT, TM, CK; int KZ, K, DIL; int nsplit = 1; int pm = 3; int once = 0; };

// the z-sliding kernel (conv_zs.hpp, plan kind 7)
struct ZsParams {
  const float* x;               // fp32 NDHWC (B,Di,Hi,Wi,Cin), or (vol) the NHWC features (2B,Hi,Wi,Cin/2)
  const unsigned char* w;       // packed by pack_weights_zs_kernel<PM> (past the header)
  const float* scale; const float* shift;
  const float* res;             // fp32 NDHWC (B,Dr,Hr,Wr,32) or null
  float* y;                     // fp32 NDHWC (B,Do,Ho,Wo,32)
  const float* x_amax; const float* w_amax; float* y_amax;
  int B, Cin;
  int Di, Hi, Wi, Do, Ho, Wo, Dr, Hr, Wr;
  int relu;
  int vol, vol_mask_left;
  int ntx, nty, ncol;           // columns: (b, ty, tx), 8 x 32 outputs each
  long nunits;                  // ncol * Do
  unsigned wbytes;
};

// conv_f16.hip: the fp16 kernels (plan kinds 5 / 6 / 7 with pm = 2 | 1)
__attribute__((visibility("hidden"))) int run_split_f16(const Plan& pl, const ConvParams& p, hipStream_t s);
__attribute__((visibility("hidden"))) int run_zs_f16(int pm, const ZsParams& p, int grid, hipStream_t s);

// the fused 64-channel BasicBlock (basicblock2d.hpp)
struct BbParams {
  const float* x; float* y;
  const unsigned char* w1; const unsigned char* w2;       // fp16 planes [Cin/16][tap9][n 2][plane][lane][16 B]
  const float* w1_amax; const float* w2_amax;
  const float* scale1; const float* shift1; const float* scale2; const float* shift2;
  const float* x_amax; float* y_amax;
  int B, H, W, C, relu, skip, ntx, nty, ntiles;           // relu: 1 = ReLU at the end (GCNet's block); skip: 1 = + x
  unsigned xbytes, wbytes;
};

__attribute__((visibility("hidden"))) int run_basicblock_f16(int pm, const BbParams& p, hipStream_t s);

}  // namespace dsmk

namespace {

using dsmk::ConvParams;
using dsmk::Plan;
using dsmk::ZsParams;
using dsmk::BbParams;

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



// Power-of-two scaling of the fp16 split kernels: e such that amax * 2^e lies in [2^12, 2^13)
// (fp16 overflows at 2^16), clamped so that 2^-(ex + ew) stays a normal float.
__host__ __device__ __forceinline__ int dsm_amax_exponent(float amax) {
  const int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
  const int e = 12 - ex;
  return e < -60 ? -60 : (e > 60 ? 60 : e);
}
__host__ __device__ __forceinline__ float dsm_pow2f(int e) {
  return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}

// max |y| of a launch: a running maximum per thread in the epilogue; at the end one wave reduction,
// one LDS exchange between the workgroup's four waves and ONE atomic per workgroup (non-negative
// floats order like their bit patterns) -- and none at all when the slot already holds as much
// (hundreds of same-address atomics per launch serialise in L2).  `lds4`: four floats of LDS no
// wave still reads (the call is preceded by a barrier of its own).
__device__ __forceinline__ void flush_amax(float* slot, float am, float* lds4) {
  if (!slot) return;                                   // uniform
#pragma unroll
  for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0) {
    am = fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
    if (am > __builtin_nontemporal_load(slot))
      atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, am));
  }
}
// the same for a workgroup of eight waves (the kernels with MFMA waves and staging waves)
__device__ __forceinline__ void flush_amax8(float* slot, float am, float* lds8) {
  if (!slot) return;                                   // uniform
#pragma unroll
  for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds8[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0) {
    am = fmaxf(fmaxf(fmaxf(lds8[0], lds8[1]), fmaxf(lds8[2], lds8[3])), fmaxf(fmaxf(lds8[4], lds8[5]), fmaxf(lds8[6], lds8[7])));
    if (am > __builtin_nontemporal_load(slot))
      atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, am));
  }
}
__device__ __forceinline__ void track_amax(float& am, const f32x4 v) {
  am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

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

// Epilogue of one 32x32 accumulator tile.  The MFMAs are issued with the WEIGHT fragment as
// the A operand and the activation fragment as B, so the tile comes out transposed: lane
// (r = lane & 31, h = lane >> 5) owns output voxel r of the row and register k holds channel
// (k & 3) + 8 (k >> 2) + 4 h -- four consecutive channels per register quad, i.e. 16
// contiguous bytes of the NDHWC voxel.  Each lane therefore issues 4 dwordx4 stores (and 4
// dwordx4 skip loads) per tile instead of 16 dword ones: the store tail is issue-bound, not
// bandwidth-bound (measured: ~20k cycles per tile with dword stores).
//   y = acc*scale + shift (ReLU?) (+ skip) (ReLU?)
// `yv` / `rv` point at this lane's voxel, channel 32*n + 4*h; XS = voxel stride between
// consecutive lanes (1, or 2 for a transposed-conv parity class).
// per-lane epilogue constants: scale/shift of the 16 channels this lane owns in one N-tile
struct Affine { f32x4 sc[4], sh[4]; };
__device__ __forceinline__ Affine load_affine(const float* __restrict__ scale,
                                              const float* __restrict__ shift, int cbase) {
  Affine a;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    a.sc[g] = scale ? *reinterpret_cast<const f32x4*>(scale + cbase + 8 * g) : one;
    a.sh[g] = shift ? *reinterpret_cast<const f32x4*>(shift + cbase + 8 * g) : zero;
  }
  return a;
}

// The epilogue of one 32-voxel x 32-channel accumulator tile, in two halves: `load_residual` requests the
// skip tensor's values, `store_tile` finishes and stores.  A kernel with several tiles per epilogue
// (rows, output blocks, the four classes of the transposed convolution) calls load_residual for ALL
// of them first: left to one call per tile, the next tile's loads cannot move above the previous
// tile's stores (they may alias as far as the compiler knows), and every tile exposes a full memory
// latency -- the transposed 64 -> 32 kernel spent most of its 170 us there (profiles/r03_ablation.md, 5).
struct Residual { f32x4 v[4]; };
__device__ __forceinline__ void load_residual(Residual& r, const float* __restrict__ rv) {
#pragma unroll
  for (int g = 0; g < 4; ++g) r.v[g] = *reinterpret_cast<const f32x4*>(rv + 8 * g);
}
template <int COUT>
__device__ __forceinline__ void store_tile(const f32x16& acc, const Affine& af, int relu,
                                           float* __restrict__ yv, const Residual& res, bool has_res, float& am) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 sc = af.sc[g], sh = af.sh[g];
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    v = v * sc + sh;
    if (relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (has_res) v += res.v[g];
    if (relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(yv + 8 * g) = v;
    track_amax(am, v);
  }
}
// one tile, residual requested and consumed on the spot (the fp32-input kernels, single-tile epilogues)
template <int COUT>
__device__ __forceinline__ void store_tile(const f32x16& acc, const Affine& af, int relu,
                                           float* __restrict__ yv, const float* __restrict__ rv, float& am) {
  Residual r;
#pragma unroll
  for (int g = 0; g < 4; ++g) r.v[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (rv) load_residual(r, rv);
  store_tile<COUT>(acc, af, relu, yv, r, rv != nullptr, am);
}

// ----------------------------------------------------------------------------
// Staging of one channel chunk of a halo tile: global -> registers -> LDS.
//
// On gfx950 the fp32 MFMA shares the SIMD's vector ALU ("runs at the vector rate"): every
// VALU instruction of either resident wave is time taken from the matrix pipe (stamps: two
// workgroups per CU spend 2 x 27.6k cycles in MFMAs + 2 x 6.6k in VALU per pair of chunks,
// and staging code that runs 3k cycles alone takes 14k beside a multiplying partner).  So
// the staging path is built to issue (almost) no VALU:
//  * the LDS image is stored in element order e = ((z*IY + y)*IX + x)*NQ + q and thread
//    `tid` owns elements e = tid + 256*k: a commit is NPF ds_write_b128 with immediate
//    offsets (the A-fragment reads then have a 4-way bank conflict, irrelevant beside
//    64-cycle MFMAs);
//  * each thread's global offsets are computed once per launch; a chunk whose halo box lies
//    inside the volume (wave-uniform test) is staged by loads of the form
//    scalar base + 32-bit register offset, zero VALU; boxes that stick out of the volume
//    (edge tiles) take a guarded path that decodes coordinates on the fly;
//  * the loads are issued one per item inside the multiply loop.
// ----------------------------------------------------------------------------
// 16-byte load through a buffer descriptor: address = descriptor base + voffset (VGPR, fixed per
// lane) + soffset (SGPR).  No address VALU at all, which is the point here (T8 in the guide).
__device__ __forceinline__ f32x4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset,
                                               unsigned soffset) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voffset, (int)soffset, 0);
  static_assert(sizeof(v) == 16, "raw_buffer_load_b128 must return 16 bytes");
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}



template <typename K>
int launch_tiles(K kernel, const ConvParams& p, size_t lds, hipStream_t s, int max_blocks, int ny = 1) {
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
  if (p.force_blocks) max_blocks = p.force_blocks;     // dsm_conv3d_args.flags: persistent-grid A/B runs
  int blocks = p.ntiles < max_blocks ? p.ntiles : max_blocks;
  if (blocks >= 8) blocks &= ~7;                 // whole rounds over the 8 XCDs
  hipLaunchKernelGGL(kernel, dim3(blocks, ny), dim3(NTHREADS), lds, s, p);
  return dsm_launch_status();
}

}  // namespace
