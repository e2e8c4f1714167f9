// One (variant, EXTRA, guided) instantiation of the bf16 persistent integrator per translation unit,
// so the ~1 minute device compilations run in parallel.
// Built with -DB16_VARIANT=0|1 -DB16_EXTRA=0|1|2 (kernel MODE) -DB16_GUIDE=0|1.
#include <hip/hip_runtime.h>

#define B16_ASM_BODY 1   // every kernel runs a generated assembly body (gen_bf16_body.py)
#include "plan.hpp"
#include "engine_bf16.hip.inc"
#include "engine_bf16_variants.hpp"

// the product library carries the shipped schedule only: a body generated under GEN_* experiment knobs (some of which
// give wrong results by design) compiles only in an experiment build (csrc/xbuild.sh passes -DB16_EXPERIMENT)
#if !defined(B16_EXPERIMENT)
static_assert(B16_GEN_DEFAULT_KNOBS == 1, "engine_bf16_body_gen.inc was generated with non-default GEN_* knobs");
#endif

namespace grlhip {
#if B16_VARIANT == 0
using CfgSel = CfgQgpo;
#else
using CfgSel = CfgSwiss;
#endif
#if B16_GUIDE == 0
using GuideSel = b16::NoGuide;
#else
using GuideSel = GuideQgpo;
#endif
using BodySel = AsmBodyFor<CfgSel, GuideSel>::type;
#define B16_CAT2(a, b, c, d, e, f) a##b##c##d##e##f
#define B16_CAT(a, b, c, d, e, f) B16_CAT2(a, b, c, d, e, f)
#if B16_VARIANT == 0 && B16_EXTRA == 0 && B16_GUIDE == 0
// what the generated bodies of this library expect from the packer (engine_bf16_host.hip.inc, bf16_pack): 1 = the
// time-MLP bias of every residual block folded into its dense2 bias
int b16_body_folds_time_bias() { return B16_GEN_FOLD_TIME_BIAS; }
#endif
void B16_CAT(b16_launch_v, B16_VARIANT, _e, B16_EXTRA, _g, B16_GUIDE)(b16::SampleArgs16& a, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((b16::k_sample_bf16<CfgSel, GuideSel, B16_NSLOT, B16_CF, B16_EXTRA, BodySel>), dim3(blocks), dim3(256), 0, st, a);
}
}  // namespace grlhip
