// The network shapes the bf16 engine is instantiated for, and its ring geometry.
#pragma once
namespace grlhip {
// Shape classes (padded I/O widths; actual widths are run-time arguments of the kernel):
//   CfgQgpo   conditional [512,256,128] nets of the QGPO / GMPO benchmark configs: action <= 16, observation <= 32
//             when energy-guided, any width unguided (HalfCheetah / Walker2d 6/17, Hopper 3/11, Ant 8/27, ...)
//   CfgSwiss  unconditional [512,256,128] nets, D <= 16 (swiss-roll tutorial: D = 2)
using CfgQgpo = b16::Cfg<16, 32, 32, 32, 128, 512, 256, 128>;
using CfgSwiss = b16::Cfg<16, 0, 0, 32, 32, 512, 256, 128>;
using GuideQgpo = b16::GuideCfg<32, 256, 256>;                // QGPO energy net: ConcatenateMLP [55, 256, 256] -> 1
#if defined(B16_ASM_BODY)
// the generated assembly body of each sampler shape (engine_bf16_body_gen.inc)
template <class C, class G> struct AsmBodyFor;
template <> struct AsmBodyFor<CfgQgpo, b16::NoGuide> { using type = b16::AsmBody_cond; };
template <> struct AsmBodyFor<CfgSwiss, b16::NoGuide> { using type = b16::AsmBody_uncond; };
template <> struct AsmBodyFor<CfgQgpo, GuideQgpo> { using type = b16::AsmBody_cond_guided; };
#endif
// Ring geometry (must match gen_bf16_body.py's GEN_NSLOT / GEN_CF).  Measured (profiles/r02_asm_body_experiments.md):
// 4, 6 and 8 slots run at the same speed, so the smallest is kept and the LDS goes to the input park slots.
constexpr int B16_NSLOT = 4, B16_CF = 16;   // ring slots, fragments per chunk
}  // namespace grlhip
