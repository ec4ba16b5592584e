// The fp16 forms of the split-operand convolution kernels (conv_split.hpp, PM = 2 "f16x2" and
// PM = 1 "f16") -- a translation unit of their own so that they compile beside conv3d.hip.
//   f16x2: fp32 accuracy at three MFMAs per product instead of bf16x3's six (the default of the
//          PSMNet / GCNet eval forwards when the precision gate of scripts/precision_check.py holds);
//   f16:   operands rounded to fp16, fp32 accumulate -- the reduced-precision mode of BASELINE
//          config #5 ("PSMNet fp16 training step"); the reference itself computes in fp32
//          (models/psmnet/stackhourglass.py:124).
// Replaces the same reference code as conv3d.hip: convbn_3d / conv3d_bn / deconv3d_bn and the
// towers' convbn (models/psmnet/submodule.py:10-19, stackhourglass.py:22-62,73-98,135-149,
// models/util_conv.py:150-179).
#include "conv_common.hpp"

namespace {
#include "conv_split.hpp"
#include "conv_zs.hpp"
#include "basicblock2d.hpp"
}  // namespace

int dsmk::run_zs_f16(int pm, const ZsParams& p, int grid, hipStream_t s) {
  if (pm == 2) return launch_conv_zs<2>(p, grid, s);
  if (pm == 1) return launch_conv_zs<1>(p, grid, s);
  return DSM_ERR_UNSUPPORTED;
}

int dsmk::run_split_f16(const Plan& pl, const ConvParams& p, hipStream_t s) {
  if (pl.pm == 2) return dispatch_split<2>(pl, p, s);
  if (pl.pm == 1) return dispatch_split<1>(pl, p, s);
  return DSM_ERR_UNSUPPORTED;
}

int dsmk::run_basicblock_f16(int pm, const BbParams& p, hipStream_t s) {
  if (pm == 2) return p.C == 64 ? launch_basicblock2d<2, 64>(p, s) : launch_basicblock2d<2, 32>(p, s);
  if (pm == 1) return p.C == 64 ? launch_basicblock2d<1, 64>(p, s) : launch_basicblock2d<1, 32>(p, s);
  return DSM_ERR_UNSUPPORTED;
}
