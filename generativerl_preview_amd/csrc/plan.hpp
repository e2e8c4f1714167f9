// Host-side plan: parameter list, op program, activation-slab layout and packed-weight layout
// for one network (TemporalSpatialResidualNet or ConcatenateMLP).  Pure C++ (no HIP types) so the
// layout logic can be unit-tested on the CPU.
//
// Activation slab (fp32 engine): every wave owns 32 batch rows; a buffer of F features is stored
// feature-major as slab[(off + f) * 32 + row].  Offsets below are in FEATURES.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/grlhip.h"

namespace grlhip {

constexpr int GRLHIP_BF16_MAX_STEPS = 4096;  // rows of the per-step Fourier table in the bf16 workspace
int device_cus();                            // CU count of the current device (256 when none is visible)

enum OpType : int32_t {
  OP_LINEAR = 0,         // out[n] = act(sum_k W[n][k] in[k] + b[n]) (+ slab[add + n])
  OP_SILU = 1,           // out[n] = silu(in[n])
  OP_COPY = 2,           // out[n] = in[n]
  OP_SILU_BWD = 3,       // out[n] = slab[add + n] * silu'(in[n])
  OP_SILU_BWD_CONST = 4, // out[n] = wts[w + n] * silu'(in[n])
  OP_DW = 5,             // grad[w + n*pre + k] += sum_rows slab[add + n] * slab[in + k]; grad[b + n] += sum_rows slab[add + n]
  OP_ADD = 6,            // out[n] += in[n]
  OP_RELU_BWD = 7,       // out[n] = slab[add + n] * (in[n] > 0)
  // the 1-wide output layer of an energy / Q net as matrix-vector work (K features):
  OP_HEAD_FWD = 8,       // slab[out] = wts[b] + sum_k wts[w + k] * slab[in + k]
  // its backward from g = slab[add] (one feature): slab[out + k] = g * wts[w + k] * act'(slab[pre + k]) (the dZ of the
  // layer below; act = ReLU with F_RELU_OUT, else SiLU); grad[lin's weight + k] += sum_rows g * slab[in + k];
  // grad[bias] += sum_rows g.  Op.lin = the output linear (its parameter indices give the gradient offsets)
  OP_HEAD_BWD = 9
};
enum OpFlags : int32_t { F_SILU_OUT = 1, F_SAVE_PRE = 2, F_ACCUM = 4, F_RELU_OUT = 8,
                         // backward: besides out = A in (+bias) (+ slab[add]) (accumulated with F_ACCUM), features
                         // n < zN of it also give pre[n] = out[n] * SiLU'(slab[z + n]) -- the dZ of the layer below,
                         // whose pre-activation sits at `z`
                         // (with F_RELU_OUT: ReLU' instead of SiLU'; with F_NO_OUT: out itself is not stored)
                         F_DSILU = 16, F_NO_OUT = 32,
                         // the op only feeds the gradient of the Fourier frequencies: skipped unless Plan::fourier_grad
                         F_FOURIER_ONLY = 64,
                         F_PAIR_NEXT = 128 };   // independent of the next OP_LINEAR of the program: the engine may launch the two together

struct Op {
  int32_t type, K, N, w, b, in, out, add, flags, pre;  // pre: where z (pre-activation) is kept (F_SAVE_PRE)
  int32_t lin;                                         // index into linears (OP_LINEAR), else -1
  int32_t z, zN;                                       // F_DSILU: pre-activation offset, number of leading features
};

struct ParamInfo {
  std::string name;
  int32_t rows, cols;  // cols == 0: vector
  int64_t grad_off;    // offset in the flat gradient buffer
};

// One packed matrix: M[n][k] (n < N, k < K) gathered from parameter p_w as
//   transposed ? P[k * src_cols + col0 + n] : P[n * src_cols + col0 + k]
struct LinearDesc {
  int32_t N, K, Npad, Kpad;
  int32_t p_w, p_b;  // parameter indices (p_b = -1: zero bias)
  int32_t col0, src_cols, transposed;
  int64_t w_off, b_off;    // fp32 image, in floats
  int64_t w_off16, b_off16;  // bf16 image: weights offset in bf16 elements, bias offset in floats of the f32 tail
};

struct Plan {
  grlhip_net_desc desc;
  std::vector<ParamInfo> params;
  std::vector<LinearDesc> linears;
  std::vector<Op> fwd;       // forward program
  std::vector<Op> grad;      // CONCAT_MLP only: forward (pre-activations) + backward to x
  // RESNET training step: forward keeping every pre-activation, then the backward program
  std::vector<Op> train_fwd, train_bwd;
  int32_t train_slab_features = 0;
  int32_t tr_temb = 0, tr_x = 0, tr_c = -1, tr_out = 0, tr_gout = 0, tr_gtc = 0;
  int32_t tr_gx = -1;            // d loss / d x (RESNET; written by the last train_bwd_dx_ops ops of train_bwd)
  int32_t train_bwd_dx_ops = 0;  // trailing ops of train_bwd that only the differentiable integrator needs
  bool fourier_grad = true;      // compute d loss / d (Fourier frequencies) in the training entry points
  int32_t bwd_pieces = 3;        // bf16 pieces per fp32 value in the GEMMs of the training-backward program (3 or 2)
  void* d_train_fwd = nullptr;
  void* d_train_bwd = nullptr;
  int32_t slab_features = 0;
  int32_t temb_off = 0, x_off = 0, c_off = 0, out_off = 0, gradx_off = 0;
  int64_t fourier_off = 0;   // offset of W (t_dim/2 floats) in the fp32 image
  int64_t packed_floats = 0; // size of the fp32 image
  // GRLHIP_F32 packs carry, behind the fp32 image, every matrix pre-split into its three bf16 pieces in the order the
  // fp32 GEMM stages them: [16-k tile][piece][row, padded to 128][16 k] (engine_batch: k_bt_gemm_x6).  Offsets in bf16
  // elements from the start of that region, per linear
  std::vector<int64_t> split_off;
  int64_t split_elems = 0;
  int64_t grad_numel = 0;
  // device copies of the programs (uploaded at creation)
  void* d_fwd = nullptr;
  void* d_grad = nullptr;
  void* bf16 = nullptr;      // Bf16Layout* (engine_bf16.hip.inc), null when the shape has no bf16 kernel
  std::string error;
};

// Returns nullptr and fills err on invalid descriptors.
Plan* build_plan(const grlhip_net_desc& d, std::string& err);

}  // namespace grlhip
