// One (variant, EXTRA, guided) instantiation of the bf16 persistent integrator per translation unit,
// so the ~1 minute device compilations run in parallel.
// Built with -DB16_VARIANT=0|1 -DB16_EXTRA=0|1|2 (kernel MODE) -DB16_GUIDE=0|1.
#include <hip/hip_runtime.h>

#if B16_GUIDE == 0
#define B16_ASM_BODY 1   // unguided kernels run the generated assembly body
#endif
#include "plan.hpp"
#include "engine_bf16.hip.inc"
#include "engine_bf16_variants.hpp"

namespace grlhip {
#if B16_VARIANT == 0
using CfgSel = CfgQgpo;
#else
using CfgSel = CfgSwiss;
#endif
#if B16_GUIDE == 0
using GuideSel = b16::NoGuide;
using BodySel = AsmBodyFor<CfgSel>::type;   // unguided kernels run the hand-scheduled assembly body
#else
using GuideSel = GuideQgpo;
using BodySel = b16::NoAsmBody;             // guided kernels: C++ body (net + energy-net gradient)
#endif
#define B16_CAT2(a, b, c, d, e, f) a##b##c##d##e##f
#define B16_CAT(a, b, c, d, e, f) B16_CAT2(a, b, c, d, e, f)
void B16_CAT(b16_launch_v, B16_VARIANT, _e, B16_EXTRA, _g, B16_GUIDE)(b16::SampleArgs16& a, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((b16::k_sample_bf16<CfgSel, GuideSel, (B16_GUIDE == 0 ? B16_NSLOT_ASM : B16_NSLOT), (B16_GUIDE == 0 ? B16_CF_ASM : B16_CF), B16_P, B16_EXTRA, BodySel>), dim3(blocks), dim3(256), 0, st, a);
}
}  // namespace grlhip
