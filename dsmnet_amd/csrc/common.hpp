// Shared helpers for the gfx950 kernels of the stereo cost-volume path.
// Written for CDNA4 only: wave = 64 lanes, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/dsmnet_hip.h"

#define DSM_WAVE 64

// ext-vector float4 with 4-byte alignment: gfx950 global/LDS dwordx4 accesses
// only need dword alignment, so shifted (x - d) rows can still move 16 B per lane.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipGetLastError() is sticky per thread and is shared with everything else in the
// process (PyTorch, MIOpen ...): drop whatever an earlier, unrelated call left behind
// before launching, so that dsm_launch_status() reports only this library's launch.
static inline void dsm_clear_stale_error() { (void)hipGetLastError(); }

static inline int dsm_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DSM_OK : DSM_ERR_LAUNCH;
}

static inline bool dsm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

static inline int dsm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#define DSM_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
