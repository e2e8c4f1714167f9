// The network shapes the bf16 engine is instantiated for, and its ring geometry.
#pragma once
namespace grlhip {
using CfgQgpo = b16::Cfg<6, 17, 32, 32, 128, 512, 256, 128>;  // QGPO / GMPO HalfCheetah: act 6, obs 17
using CfgSwiss = b16::Cfg<2, 0, 0, 32, 32, 512, 256, 128>;    // swiss-roll tutorial: D = 2, unconditional
constexpr int B16_NSLOT = 4, B16_CF = 16, B16_P = 8;  // ring slots, fragments per chunk, register prefetch depth
}  // namespace grlhip
