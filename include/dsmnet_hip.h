/*
 * dsmnet_hip.h -- C ABI of the MI355X (gfx950) stereo cost-volume path.
 *
 * The reference (sunshinnnn/DSMnet) is pure Python: it defines NO plugin,
 * operator or FFI interface for this path.  Each entry
 * point below therefore cites the reference *Python code it replaces*; the
 * binding a maintainer adds on the reference side is a ctypes stub
 * (INTEGRATION.md, and dsmnet_amd/_lib.py is that stub in full).
 *
 * Conventions
 *  - plain C: device pointers + sizes; no torch types, no C++ in the signatures;
 *  - every buffer is caller-allocated device memory; the library never
 *    allocates, frees, or synchronises; all work is enqueued on `stream`
 *    (a hipStream_t passed as void*; NULL = the null stream);
 *  - return value: DSM_OK (0) or a negative DSM_ERR_* code; no exceptions
 *    cross the boundary; dsm_strerror() names a code;
 *  - dtype: tensors are fp32 (DSM_F32; the reference computes in fp32: torch.FloatTensor,
 *    models/psmnet/stackhourglass.py:124).  What the convolutions MULTIPLY in is a per-call choice,
 *    dsm_conv3d_args.precision (DSM_PREC_*): fp32-accurate (default), or operands rounded to fp16
 *    with fp32 accumulation -- the reduced-precision mode of BASELINE config #5;
 *  - 4-D feature maps are NCHW contiguous, as torch hands them over;
 *  - 5-D volumes are either DSM_NCDHW (torch contiguous) or DSM_NDHWC
 *    (torch.channels_last_3d), selected per call.
 */
#ifndef DSMNET_HIP_H
#define DSMNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSM_ABI_VERSION 7

#define DSM_OK               0
#define DSM_ERR_ARG         -1   /* null pointer, non-positive size, bad enum      */
#define DSM_ERR_UNSUPPORTED -2   /* valid request this build has no kernel for     */
#define DSM_ERR_LAUNCH      -3   /* HIP reported a launch failure                   */
#define DSM_ERR_ALIGN       -4   /* a pointer is not aligned as the kernel needs    */

enum dsm_dtype  { DSM_F32 = 0 };
enum dsm_layout { DSM_NCDHW = 0, DSM_NDHWC = 1 };

typedef void* dsm_stream_t;      /* hipStream_t */

int         dsm_abi_version(void);
const char* dsm_strerror(int code);

/* ---------------------------------------------------------------------------
 * (a1) 1-D correlation.  Replaces Corr1d.forward, models/util_conv.py:71-86
 * (python loop over D slice-multiplies) and, for the gradient, autograd through it.
 *   out[b,i,y,x] = sum_c fL[b,c,y,x] * fR[b,c,y,x - i*stride]   (x >= i*stride, i < W)
 *   ksize > 1: every plane box-filtered, zero padding counted in the divisor
 *   (nn.AvgPool2d(k, 1, k//2), util_conv.py:82-85).
 * fL, fR: (B,C,H,W); out: (B,D,H,W); tmp: (B,D,H,W) scratch, needed iff ksize > 1.
 * ------------------------------------------------------------------------- */
int dsm_corr1d_fwd(const void* fL, const void* fR, void* out, void* tmp,
                   int B, int C, int H, int W, int D, int stride, int ksize,
                   int dtype, dsm_stream_t stream);

/* grad_out: (B,D,H,W); dfL, dfR: (B,C,H,W), fully overwritten.
 * tmp: (B,D,H,W) scratch, needed iff ksize > 1. */
int dsm_corr1d_bwd(const void* grad_out, const void* fL, const void* fR,
                   void* dfL, void* dfR, void* tmp,
                   int B, int C, int H, int W, int D, int stride, int ksize,
                   int dtype, dsm_stream_t stream);

/* ---------------------------------------------------------------------------
 * (a2,a3) concatenation cost volume.  Replaces the inline loops of
 * models/gcnet.py:130-135 (mask_left = 0) and
 * models/psmnet/stackhourglass.py:124-133 (mask_left = 1):
 *   vol[b,   c, d,y,x] = fL[b,c,y,x]        (x >= d, or every x when !mask_left)
 *   vol[b, C+c, d,y,x] = fR[b,c,y,x - d]    (x >= d)          zero elsewhere
 * fL, fR: (B,C,H,W); vol: (B,2C,D,H,W) in `layout`; written in one pass, no memset.
 * mask_left bit 1 (ABI v4, NDHWC forward only): the right-referenced volume of gcnet_LR
 * (models/gcnet.py:155-164) -- pass (fR, fL) as (fL, fR): the second map is read at x + d,
 *   vol[b, C+c, d,y,x] = second[b,c,y,x + d]   (x + d < W), zero elsewhere.
 * ------------------------------------------------------------------------- */
int dsm_concat_volume_fwd(const void* fL, const void* fR, void* vol,
                          int B, int C, int H, int W, int D,
                          int mask_left, int layout, int dtype, dsm_stream_t stream);

/* gvol: (B,2C,D,H,W) in `layout`; dfL, dfR: (B,C,H,W), fully overwritten. */
int dsm_concat_volume_bwd(const void* gvol, void* dfL, void* dfR,
                          int B, int C, int H, int W, int D,
                          int mask_left, int layout, int dtype, dsm_stream_t stream);

/* ---------------------------------------------------------------------------
 * (a6,a7) soft-argmin disparity regression, fused.  Replaces
 *   PSMNet: F.upsample(trilinear) -> F.softmax -> disparityregression
 *           (stackhourglass.py:152-166, submodule.py:56-63):  negate = 0,
 *           cost (B,Dc,Hc,Wc) upsampled to (D,H,W) on the fly;
 *   GCNet:  Softmax2d(-x) -> matmul(arange) (models/gcnet.py:104-111): negate = 1,
 *           (Dc,Hc,Wc) == (D,H,W), no interpolation.
 *   disp[b,y,x] = sum_d d * softmax_d(+-cost_up[b,d,y,x])
 * cost: (B,Dc,Hc,Wc); disp: (B,H,W); stats: (B,2,H,W) or NULL -- per-pixel
 * softmax max and normaliser, saved for the backward pass.
 * align_corners follows torch.nn.functional.interpolate (0 = what F.upsample
 * resolves to on torch >= 0.4; see DESIGN.md "version drift").
 * ------------------------------------------------------------------------- */
int dsm_soft_argmin_fwd(const void* cost, void* disp, void* stats,
                        int B, int Dc, int Hc, int Wc, int D, int H, int W,
                        int negate, int align_corners, int dtype, dsm_stream_t stream);

/* gdisp: (B,H,W); dcost: (B,Dc,Hc,Wc), fully overwritten (zeroed on `stream`
 * first when the upsampling adjoint accumulates into it). */
int dsm_soft_argmin_bwd(const void* cost, const void* disp, const void* stats,
                        const void* gdisp, void* dcost,
                        int B, int Dc, int Hc, int Wc, int D, int H, int W,
                        int negate, int align_corners, int dtype, dsm_stream_t stream);

/* ---------------------------------------------------------------------------
 * (a4,a5) 3-D convolution block, k = 3, fused epilogue.  Replaces
 *   convbn_3d (+ReLU, +skip)            models/psmnet/submodule.py:16-19,
 *                                       stackhourglass.py:22-62,73-98,135-149
 *   conv3d_bn / deconv3d_bn             models/util_conv.py:150-179,
 *   myadd_3d / myAdd3d (crop + add)     stackhourglass.py:10-20, util_fun.py:41-50
 *   relu = 1:  y = relu( conv(x, w) * scale[co] + shift[co]  (+ residual, cropped) )   PSMNet order
 *   relu = 2:  y = relu( conv(x, w) * scale[co] + shift[co] )  (+ residual, cropped)   GCNet order
 *              (models/gcnet.py:78-96: the ReLU sits inside deconv3d_bn, myAdd3d follows)
 *   relu = 0:  no activation
 * `scale`/`shift` carry the folded eval-mode BatchNorm and the conv bias.
 * transposed = 0: Conv3d(k=3, padding=1, stride in {1,2})
 * transposed = 1: ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1)
 * Activations are DSM_NDHWC, fp32.  (Do,Ho,Wo) may be smaller than the natural
 * output size: only that corner is computed -- this is the crop of myadd_3d.
 * Contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32 products and sums).
 * ------------------------------------------------------------------------- */
typedef struct dsm_conv3d_args {
  const void*  x;         /* (B,Di,Hi,Wi,Cin)                                   */
  const void*  w_packed;  /* from dsm_conv3d_pack_weights                        */
  const float* scale;     /* [Cout] or NULL (= 1)                                */
  const float* shift;     /* [Cout] or NULL (= 0)                                */
  const void*  residual;  /* (B,Dr,Hr,Wr,Cout) or NULL                           */
  void*        y;         /* (B,Do,Ho,Wo,Cout)                                   */
  int B, Cin, Cout;
  int Di, Hi, Wi;
  int Do, Ho, Wo;
  int Dr, Hr, Wr;
  int stride;
  int transposed;
  int relu;               /* 0 none, 1 after the skip add, 2 before it */
  /* kernel geometry; 0 = default.  kd = 1 selects the 2-D form (Di = Do = 1, NHWC maps as
   * (B,1,H,W,C) volumes) used for the 2-D feature towers:
   * k in {1,3}, dil in {1,2}, padding = dil*(k-1)/2 ("same"), as in convbn()
   * (models/psmnet/submodule.py:10-13) for k = 3 and the towers' 1x1 convolutions. */
  int kd;                 /* depth taps: 3 (default) or 1            */
  int k;                  /* taps in y and x: 3 (default) or 1       */
  int dil;                /* dilation in y and x: 1 (default) or 2   */
  /* ABI v4: tuning / A-B switches of THIS call (0 = the plan's own choice).  The library reads no
   * environment variable and holds no global switch: a plan is a pure function of the arguments. */
  int          flags;
  /* ABI v6: what the 3x3(x3) MFMA kernels multiply in (DSM_PREC_*, below).  The fp16 modes scale
   * every tensor by a power of two taken from its absolute maximum: `x_amax` points at a device
   * float holding max |x| (or an upper bound of it) -- written by the launch that produced x
   * through ITS y_amax, or by dsm_absmax; required when precision != DSM_PREC_F32 and the layer runs
   * on a split kernel (dsm_conv3d_plan names it "..._f16x2_..." / "..._f16_..."), ignored otherwise. */
  int          precision;
  const float* x_amax;
  /* ABI v6: optional.  max |y| of this launch is folded into *y_amax with an atomic maximum on the
   * float bits: the caller zeroes it (on the stream) before the launch.  Any precision, every MFMA
   * kernel (Cout >= 32). */
  float*       y_amax;
  /* ABI v6: vol_virtual = 1 -- the input is a concatenation cost volume that is NEVER MATERIALISED
   * (models/psmnet/stackhourglass.py:124-133 with vol_mask_left = 1, models/gcnet.py:130-135 with 0,
   * fused into the first 3-D convolution).  `x` is then the NHWC feature tensor (2B, Hi, Wi, Cin/2)
   * [the B left maps, then the B right maps]; input plane d of the (B, Cin, Di, Hi, Wi) volume is
   * staged as [left | right shifted by d voxels] with x < d zeroed (right half always, left half
   * iff vol_mask_left); Di = the number of disparity planes.  Conv3d(k3, s1) to 32 channels only
   * (the z-sliding kernel, "conv3d_zs_..."): DSM_ERR_UNSUPPORTED otherwise.  x_amax: of the features. */
  int          vol_virtual;
  int          vol_mask_left;
} dsm_conv3d_args;
/* fp32 operands, fp32-accurate products and sums: three-term bf16 split (six bf16 MFMAs per
 * product) or, with DSM_CONV_FP32_MFMA, the fp32-input MFMA */
#define DSM_PREC_F32    0
/* operands rounded to fp16 (after the power-of-two scaling), one MFMA per product, fp32 accumulate:
 * relative error ~3e-4 of the result's scale.  BASELINE config #5 ("fp16 training step"). */
#define DSM_PREC_F16    1
/* fp32 accuracy on the fp16 pipe: x * 2^e = hi + lo in fp16 (22 significand bits), three MFMAs per
 * product (wl xh + wh xl + wh xh), fp32 accumulate; measured error vs float64 at or below the
 * bf16x3 and fp32-input kernels' (scripts/precision_check.py). */
#define DSM_PREC_F16X2  2
#define DSM_CONV_FP32_MFMA      0x1      /* keep the layer on the exact fp32-input MFMA (no bf16x3)   */
#define DSM_CONV_COUT1_CHUNKED  0x2      /* Cout = 1: the chunked kernel instead of the z-sliding one */
#define DSM_CONV_NO_NSPLIT      0x4      /* 2-D bf16x3 layers: one workgroup per tile (no N-split)    */
#define DSM_CONV_NO_ONCE        0x8      /* A/B: the chunk-pipelined single-kind kernels instead of the single-tile (2-D 64 -> 64) and two-kind (stride-2 3-D) forms */
#define DSM_CONV_TM_SHIFT       4        /* bits 4..7: force the tile height 4*TM rows (TM = 1, 2, 4) */
#define DSM_CONV_BLOCKS_SHIFT   16       /* bits 16..31: force the persistent grid size               */

/* bytes of the packed (MFMA-fragment-ordered) weight buffer */
size_t dsm_conv3d_packed_weight_bytes(int Cin, int Cout, int transposed);

/* w_torch: torch layout, (Cout,Cin,3,3,3) or, transposed, (Cin,Cout,3,3,3). */
int dsm_conv3d_pack_weights(const void* w_torch, void* w_packed,
                            int Cin, int Cout, int transposed, dsm_stream_t stream);

/* General packer: kd x k x k taps, torch layout (Cout, Cin_src, [kd,] k, k); input channels
 * Cin_src..Cin-1 are packed as zeros (PSMNet's first convolution: 3 staged as 16).
 * Bytes needed: dsm_conv_packed_weight_bytes (the fp32 fragments, Cin*Cout*kd*k*k*4, followed for
 * 3x3(x3) kernels by the pre-split bf16 planes the bf16x3 kernels read and by the two fp16 planes
 * of the power-of-two-scaled weights behind a 16-byte header holding their absolute maximum). */
size_t dsm_conv_packed_weight_bytes(int Cin, int Cout, int kd, int k);
int dsm_conv_pack_weights(const void* w_torch, void* w_packed, int Cin_src, int Cin, int Cout,
                          int kd, int k, dsm_stream_t stream);

/* (ABI v6) max |x| over n floats, folded into *amax (atomic maximum on the float bits; the caller
 * zeroes it first): the `x_amax` of a tensor no launch of this library produced. */
int dsm_absmax(const void* x, size_t n, float* amax, dsm_stream_t stream);

/* Two convolutions per launch: the stride-1 BasicBlock of PSMNet's towers (models/psmnet/submodule.py:24-46;
 * C = 64 or 32) and of GCNet's (models/util_conv.py:181-210; C = 32, relu = 1):
 * y = [ReLU](BN2(conv2(ReLU(BN1(conv1(x))))) + x),  both Conv2d(C, C, 3, stride 1, pad 1), folded BatchNorm as
 * scale / shift per channel (NULL: 1 / 0); relu = 1: ReLU after the skip add.  x, y: fp32 NHWC (B, H, W, C);
 * w1_packed, w2_packed: dsm_conv_pack_weights buffers of the two layers (Cin = Cout = C, kd = 1, k = 3).  precision: DSM_PREC_F16X2 or DSM_PREC_F16 (the fp16 modes only); x_amax as in
 * dsm_conv3d_args (required), y_amax optional.  The intermediate map never leaves the chip; it is scaled
 * for its fp16 split by each tile's own maximum.  Eval mode only (no backward). */
typedef struct {
  const void* x; void* y;
  const void* w1_packed; const void* w2_packed;
  const float* scale1; const float* shift1; const float* scale2; const float* shift2;
  const float* x_amax; float* y_amax;
  int B, H, W, C;
  int precision;
  int relu;
  int no_skip;       /* 1: no `+ x` (two convolution + BN + ReLU layers in a row: PSMNet's firstconv) */
} dsm_basicblock2d_args;
int dsm_basicblock2d_fwd(const dsm_basicblock2d_args* a, dsm_stream_t stream);

int dsm_conv3d_fwd(const dsm_conv3d_args* args, dsm_stream_t stream);

/* Name of the kernel variant dsm_conv3d_fwd would launch for `args` (tile shape and
 * channel chunk are chosen per layer) -- written NUL-terminated into buf[len];
 * used by bench.py to attribute per-launch timings.  No launch, no device access. */
int dsm_conv3d_plan(const dsm_conv3d_args* args, char* buf, int len);

/* ---------------------------------------------------------------------------
 * Backward of the 3-D convolution blocks (training through the trunk; in the reference this
 * is autograd through nn.Conv3d / nn.ConvTranspose3d).  bwd-data is a convolution again and
 * runs on dsm_conv3d_fwd with re-packed weights (see dsmnet_amd/costvolume.py); these are the
 * bwd-weight entry points.
 *   dW[g][c][tap] = sum_v X[v*stride + tap - 1][c] * G[v][g]
 * x: (B,Dx,Hx,Wx,Cx) NDHWC, the strided-over side; g: (B,Dg,Hg,Wg,Cg) NDHWC.
 *   Conv3d(stride s):      X = layer input, G = dY        -> dw is (Cout, Cin, 3,3,3)
 *   ConvTranspose3d(s=2):  X = dY, G = layer input, stride 2 -> dw is (Cin, Cout, 3,3,3)
 * ws: scratch of (Cx/32)*(Cg/32)*27*1024 floats; dw: Cg*Cx*27 floats, overwritten.
 * Channels must be multiples of 32.  Sums are accumulated with fp32 atomics (order varies).
 * flags (ABI v5): DSM_CONV_FP32_MFMA keeps the exact fp32-input MFMA kernel; 0 = fp32 operands on
 * the 16-bit matrix pipe, staged once as [voxel][channel] 16-bit planes and read through gfx950's
 * transposed LDS read.  precision (ABI v6, DSM_PREC_*): bf16x3 (DSM_PREC_F32), or the fp16 forms
 * f16x2 / f16, which scale X and G by powers of two taken from the device scalars x_amax / g_amax
 * (max |X|, max |G|; required then -- dsm_absmax or a producer's y_amax).
 * ------------------------------------------------------------------------- */
int dsm_conv3d_wgrad(const void* x, const void* g, void* ws, void* dw, int B, int Cx, int Cg,
                     int Dx, int Hx, int Wx, int Dg, int Hg, int Wg, int stride, int flags,
                     int precision, const float* x_amax, const float* g_amax, dsm_stream_t stream);

/* (ABI v5) The same for the 2-D towers' 3x3 layers -- autograd through convbn / BasicBlock
 * (models/psmnet/submodule.py:10-13,24-46) when the feature extraction trains.  padding = dilation;
 * (stride, dilation) in {(1,1), (1,2), (2,1)}.  x: (B,Hx,Wx,Cx) NHWC, g: (B,Hg,Wg,Cg) NHWC;
 * ws: (Cx/32)*(Cg/32)*9*32*32 floats (zeroed here); dw: (Cg, Cx, 3, 3) torch layout, overwritten. */
int dsm_conv2d_wgrad(const void* x, const void* g, void* ws, void* dw, int B, int Cx, int Cg,
                     int Hx, int Wx, int Hg, int Wg, int stride, int dilation, int flags,
                     int precision, const float* x_amax, const float* g_amax, dsm_stream_t stream);

/* Cout = 1, stride 1 (classifier heads): g (B,D,H,W); x (B,D,H,W,C); w_packed [27][C];
 * dx (B,D,H,W,C) or NULL; dw_tapmajor [27][C] or NULL (the caller transposes to (1,C,27)). */
int dsm_conv3d_cout1_bwd(const void* x, const void* g, const void* w_packed, void* dx,
                         void* dw_tapmajor, int B, int C, int D, int H, int W,
                         dsm_stream_t stream);

/* ConvTranspose3d(C -> 1, k3, s2, p1, op1) backward -- GCNet's head l37 (models/gcnet.py:63,98;
 * autograd through nn.ConvTranspose3d in the reference).  g (B,Do,Ho,Wo); x (B,Di,Hi,Wi,C)
 * NDHWC; w: torch layout (C,1,3,3,3); dx (B,Di,Hi,Wi,C) or NULL; dw (C,1,3,3,3) or NULL,
 * overwritten.  C % 4 == 0, C <= 256. */
int dsm_deconv3d_cout1_bwd(const void* x, const void* g, const void* w, void* dx, void* dw,
                           int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                           dsm_stream_t stream);

/* ---------------------------------------------------------------------------
 * Train-mode BatchNorm3d + cropped skip addition + ReLU on NDHWC fp32 volumes, forward and
 * backward (dsmnet_amd/csrc/bn3d.hip) -- what convbn_3d / conv3d_bn leave to nn.BatchNorm3d,
 * myadd_3d and F.relu in a training step of the reference (models/psmnet/submodule.py:16-19,
 * stackhourglass.py:10-20,43-62, models/util_conv.py:150-179, util_fun.py:41-50).
 *   fwd: batch statistics of y over (B,D,H,W) -> affine[4][C] = {gamma/std, beta - mean gamma/std,
 *        mean, 1/std}; running statistics updated as nn.BatchNorm3d does (momentum, unbiased var);
 *        out = relu?( y*scale + shift (+ residual) ) on the common corner of y and residual
 *        (relu: 0 none, 1 after the addition, 2 before it).
 *   bwd: dy (B,Dy,Hy,Wy,C), dresidual (B,Dr,Hr,Wr,C) or NULL, both fully written; afterwards
 *        workspace[0..C) = dbeta, workspace[C..2C) = dgamma (doubles).
 * workspace: 2*C doubles.  C % 4 == 0, C <= 256.
 * ------------------------------------------------------------------------- */
typedef struct dsm_bn3d_args {
  const void*  y;             /* conv output (B,Dy,Hy,Wy,C)                          */
  const void*  residual;      /* (B,Dr,Hr,Wr,C) or NULL                              */
  void*        out;           /* (B,Do,Ho,Wo,C), (Do,Ho,Wo) = min(y, residual)       */
  const float* gamma;         /* [C] or NULL (= 1)                                   */
  const float* beta;          /* [C] or NULL (= 0)                                   */
  float*       running_mean;  /* [C] or NULL: updated in fwd                         */
  float*       running_var;   /* [C] or NULL                                         */
  void*        affine;        /* float [4][C]: written by fwd, read by bwd           */
  void*        workspace;     /* double [2][C]                                       */
  const void*  gout;          /* bwd: d loss / d out                                 */
  void*        dy;            /* bwd                                                 */
  void*        dresidual;     /* bwd, or NULL                                        */
  int B, C;
  int Dy, Hy, Wy;
  int Dr, Hr, Wr;
  int relu;
  float momentum, eps;
  /* ABI v6, optional (the fp16 convolution modes' x_amax): max |out| (fwd), max |dy| and
   * max |dresidual| (bwd) are folded into these device floats with an atomic maximum on the float
   * bits; the caller zeroes them first. */
  float*       out_amax;
  float*       dy_amax;
  float*       dres_amax;
} dsm_bn3d_args;
int dsm_bn3d_train_fwd(const dsm_bn3d_args* args, dsm_stream_t stream);
int dsm_bn3d_train_bwd(const dsm_bn3d_args* args, dsm_stream_t stream);

/* NCDHW <-> NDHWC repack of an fp32 volume (used at the boundary with stock
 * torch modules that want contiguous NCDHW). to_ndhwc = 1: src NCDHW. */
int dsm_volume_relayout(const void* src, void* dst, int B, int C, int D, int H, int W,
                        int to_ndhwc, dsm_stream_t stream);

/* PSMNet spatial-pyramid-pooling head on NHWC fp32 maps -- replaces, in eval mode,
 * models/psmnet/submodule.py:81-99 (branch1..4: AvgPool2d(64/32/16/8) -> convbn(128,32,1,1,0,1)
 * -> ReLU; convbn pads the 1x1 convolution by 1, :10-13) and :126-137 (bilinear upsample to the
 * 1/4-resolution grid + concat [raw 64 | skip 128 | branch4 | branch3 | branch2 | branch1]).
 *   dsm_spp_pool8     skip (B,H,W,128) -> p8 (B,H/8,W/8,128), 8x8 means (floor)
 *   dsm_spp_branches  p8 -> four maps, back to back in `branches` (dsm_spp_branch_floats floats):
 *                     map i (i = 0..3 = branch4..branch1) is (B, h8/2^i + 2, w8/2^i + 2, 32);
 *                     w_t [4][128][32] (input-channel major), scale/shift [4][32] = folded BN
 *   dsm_spp_concat    raw (B,H,W,64), skip (B,H,W,128), branches -> out (B,H,W,320)
 * H, W >= 64 (the reference's 64x64 pool needs a full window). */
size_t dsm_spp_branch_floats(int B, int h8, int w8);
int dsm_spp_pool8(const void* skip, void* p8, int B, int H, int W, dsm_stream_t stream);
int dsm_spp_branches(const void* p8, const void* w_t, const void* scale, const void* shift,
                     void* branches, int B, int h8, int w8, dsm_stream_t stream);
int dsm_spp_concat(const void* raw, const void* skip, const void* branches, void* out,
                   int B, int H, int W, float* y_amax /* NULL, or the device scalar raised to max |out| (ABI v7) */,
                   dsm_stream_t stream);

/* Disparity warp, optionally fused with the reconstruction error -- utils/imwrap.py:37-72
 * (imwrap_BCHW with its default arguments) and models/iresnet.py:169-170.
 *   R (B,C,H0,W0), disp (B,1,H,W), out (B,C,H,W), all NCHW fp32;
 *   out = grid_sample(R + delt, grid(disp), bilinear, zeros, align_corners=False)
 *   L (B,C,H,W) non-NULL: out = |L - that|.  delt: the reference's random epsilon, drawn by the
 *   caller (1e-4 * (U[0,1) + 0.1)). */
int dsm_warp_abs_error(const void* L, const void* R, const void* disp, void* out, int B, int C,
                       int H, int W, int H0, int W0, float delt, dsm_stream_t stream);

/* One decoder level of DispNetC / iResNet (models/dispnetcorr.py:89-132, iresnet.py:119-161,186-193):
 *   out = myCat2d( relu?(up + bias), upsample_x2_bilinear(pr), skip )      (util_fun.py:7-15)
 * up (B,Cu,Hu,Wu): the transposed convolution's output WITHOUT its bias; bias [Cu] or NULL;
 * pr (B,Cp,Hp,Wp) or NULL (Cp = 0); skip (B,Cs,Hs,Ws) or NULL (Cs = 0); all NCHW fp32.
 * out (B, Cu+Cp+Cs, h, w) with h = min(Hu, 2 Hp, Hs), w = min(Wu, 2 Wp, Ws) -- the caller
 * allocates it from those rules.  align_corners = False (what nn.Upsample resolves to today). */
int dsm_decoder_cat(const void* up, const void* bias, const void* pr, const void* skip, void* out,
                    int B, int Cu, int Cp, int Cs, int Hu, int Wu, int Hp, int Wp, int Hs, int Ws,
                    int relu, dsm_stream_t stream);

/* (ABI v5) Image staging of the 2-D towers (models/psmnet/stackhourglass.py:118-121 feeds the two
 * views through feature_extraction one after the other; in eval mode they share one batch):
 * left, right (B,C,H,W) NCHW fp32, C <= 16 -> out (2B,16,H,W) in NHWC memory, channels C..15 zero.
 * right = NULL stages one view (B images). */
int dsm_stage_images_nhwc16(const void* left, const void* right, void* out, int B, int C, int H, int W,
                            dsm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DSMNET_HIP_H */
