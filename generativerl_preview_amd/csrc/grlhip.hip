// C ABI of the MI355X generative-policy hot path (declarations + contract: include/grlhip.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>

#include "plan.hpp"
using namespace grlhip;

namespace grlhip {
int device_cus() {  // CU count of the CURRENT device (cached per device ordinal)
  static int cus[64] = {0};
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] <= 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus[dev] = v;
    else cus[dev] = 256;
  }
  return cus[dev];
}
}  // namespace grlhip

#include "pack_f32.hip.inc"
#include "engine_batch.hip.inc"
#include "engine_wide.hip.inc"
#include "engine_bf16_host.hip.inc"

struct grlhip_plan {
  Plan* p;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(GRLHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

int upload(const std::vector<Op>& ops, void** d) {
  *d = nullptr;
  if (ops.empty()) return 0;
  HIP_TRY(hipMalloc(d, ops.size() * sizeof(Op)));
  HIP_TRY(hipMemcpy(*d, ops.data(), ops.size() * sizeof(Op), hipMemcpyHostToDevice));
  return 0;
}

int check_ready(const grlhip_plan* plan, const void* packed) {
  if (!plan || !plan->p) return fail(GRLHIP_ERR_INVALID, "null plan");
  if (!packed) return fail(GRLHIP_ERR_INVALID, "null packed weight image");
  if (!plan->p->d_fwd) return fail(GRLHIP_ERR_HIP, "plan was created without a usable HIP device");
  return 0;
}

}  // namespace

// torch.optim.Adam's update (amsgrad / maximize off) over flat buffers, one thread per 4 elements
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct AdamArgs {
  float* p; const float* g; float* m; float* v; int64_t n;
  float b1, b2, eps, wd, step_size, bc2_sqrt;
  const float* coef;   // null, or {step_size, bc2_sqrt} on the device (grlhip_adam_step_counted: the step count lives there)
};
// the device-resident step record of grlhip_adam_step_counted: one thread advances it and derives the bias corrections
// (in double, as the host does for grlhip_adam_step)
struct AdamRecord { long long step; float step_size, bc2_sqrt; };
__global__ void k_adam_tick(AdamRecord* r, float lr, float b1, float b2) {
  const long long t = r->step + 1;
  r->step = t;
  r->step_size = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
  r->bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
__global__ __launch_bounds__(256) void k_adam_step(AdamArgs a) {
  if (a.coef) { a.step_size = a.coef[0]; a.bc2_sqrt = a.coef[1]; }
  auto one = [&](int64_t i) {
    float g = a.g[i];
    const float p = a.p[i];
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);
    const float m = a.m[i] + (g - a.m[i]) * (1.f - a.b1);                 // exp_avg.lerp_(grad, 1 - beta1)
    const float v = fmaf(a.b2, a.v[i], (1.f - a.b2) * g * g);              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    a.m[i] = m; a.v[i] = v;
    a.p[i] = p - a.step_size * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
  };
  const int64_t n4 = a.n / 4;
  const bool aligned = (((size_t)a.p | (size_t)a.g | (size_t)a.m | (size_t)a.v) & 15) == 0;
  if (aligned) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
      const f32x4 g4 = *(const f32x4*)(a.g + 4 * q), p4 = *(const f32x4*)(a.p + 4 * q);
      const f32x4 m4 = *(const f32x4*)(a.m + 4 * q), v4 = *(const f32x4*)(a.v + 4 * q);
      f32x4 po, mo, vo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g = g4[e];
        if (a.wd != 0.f) g = fmaf(a.wd, p4[e], g);
        mo[e] = m4[e] + (g - m4[e]) * (1.f - a.b1);
        vo[e] = fmaf(a.b2, v4[e], (1.f - a.b2) * g * g);
        po[e] = p4[e] - a.step_size * (mo[e] / (sqrtf(vo[e]) / a.bc2_sqrt + a.eps));
      }
      *(f32x4*)(a.m + 4 * q) = mo; *(f32x4*)(a.v + 4 * q) = vo; *(f32x4*)(a.p + 4 * q) = po;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) one(i);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) one(i);
  }
}

extern "C" {

const char* grlhip_last_error(void) { return g_err.c_str(); }

int grlhip_device_available(void) {
  int n = 0;
  return (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 1 : 0;
}

int grlhip_plan_create(const grlhip_net_desc* net, grlhip_plan** out_plan) {
  if (!net || !out_plan) return fail(GRLHIP_ERR_INVALID, "null argument");
  std::string err;
  Plan* p = build_plan(*net, err);
  if (!p) return fail(GRLHIP_ERR_INVALID, err);
  if (grlhip_device_available()) {
    int rc = upload(p->fwd, &p->d_fwd);
    if (rc == 0) rc = upload(p->grad, &p->d_grad);
    if (rc == 0) rc = upload(p->train_fwd, &p->d_train_fwd);
    if (rc == 0) rc = upload(p->train_bwd, &p->d_train_bwd);
    if (rc != 0) { delete p; return rc; }
  }
  bf16_plan_init(p, p->d_fwd != nullptr);
  *out_plan = new grlhip_plan{p};
  return 0;
}

void grlhip_plan_destroy(grlhip_plan* plan) {
  if (!plan) return;
  if (plan->p) {
    if (plan->p->d_fwd) (void)hipFree(plan->p->d_fwd);
    if (plan->p->d_grad) (void)hipFree(plan->p->d_grad);
    if (plan->p->d_train_fwd) (void)hipFree(plan->p->d_train_fwd);
    if (plan->p->d_train_bwd) (void)hipFree(plan->p->d_train_bwd);
    bf16_plan_free(plan->p);
    delete plan->p;
  }
  delete plan;
}

int grlhip_plan_num_params(const grlhip_plan* plan) { return plan && plan->p ? (int)plan->p->params.size() : 0; }

int grlhip_plan_param_info(const grlhip_plan* plan, int index, char* name_buf, size_t name_buf_len,
                           int32_t* rows, int32_t* cols) {
  if (!plan || !plan->p || index < 0 || index >= (int)plan->p->params.size())
    return fail(GRLHIP_ERR_INVALID, "parameter index out of range");
  const ParamInfo& pi = plan->p->params[index];
  if (name_buf && name_buf_len) {
    std::strncpy(name_buf, pi.name.c_str(), name_buf_len - 1);
    name_buf[name_buf_len - 1] = 0;
  }
  if (rows) *rows = pi.rows;
  if (cols) *cols = pi.cols;
  return 0;
}

int grlhip_plan_program_info(const grlhip_plan* plan, int program, int32_t* n_ops, int32_t* n_launches) {
  if (!plan || !plan->p || program < 0 || program > 3) return fail(GRLHIP_ERR_INVALID, "program must be 0..3");
  const Plan* p = plan->p;
  const std::vector<Op>& prog = program == 0 ? p->fwd : program == 1 ? p->grad : program == 2 ? p->train_fwd : p->train_bwd;
  // the engine's own schedule in count-only mode (nothing is dereferenced or launched)
  static float dummy[4];
  int launches = 0;
  if (!prog.empty()) batch_run_program(p, prog, dummy, dummy, 128, dummy, nullptr, false, 0, 3, &launches);
  if (n_ops) *n_ops = (int32_t)prog.size();
  if (n_launches) *n_launches = launches;
  return 0;
}

int grlhip_plan_schedule_check(const grlhip_plan* plan, int program) {
  if (!plan || !plan->p || program < 0 || program > 3) return fail(GRLHIP_ERR_INVALID, "program must be 0..3");
  const Plan* p = plan->p;
  const std::vector<Op>& prog = program == 0 ? p->fwd : program == 1 ? p->grad : program == 2 ? p->train_fwd : p->train_bwd;
  return program_schedule_violation(p, prog);
}

int grlhip_plan_set_fourier_grad(grlhip_plan* plan, int enable) {
  if (!plan || !plan->p) return fail(GRLHIP_ERR_INVALID, "null plan");
  plan->p->fourier_grad = enable != 0;
  return 0;
}

int grlhip_plan_set_backward_split(grlhip_plan* plan, int pieces) {
  if (!plan || !plan->p) return fail(GRLHIP_ERR_INVALID, "null plan");
  if (pieces != 2 && pieces != 3) return fail(GRLHIP_ERR_INVALID, "backward split: 2 or 3 bf16 pieces per fp32 value");
  plan->p->bwd_pieces = pieces;
  return 0;
}

int64_t grlhip_grad_numel(const grlhip_plan* plan) { return plan && plan->p ? plan->p->grad_numel : 0; }
int64_t grlhip_grad_offset(const grlhip_plan* plan, int index) {
  if (!plan || !plan->p || index < 0 || index >= (int)plan->p->params.size()) return -1;
  return plan->p->params[index].grad_off;
}

size_t grlhip_packed_bytes(const grlhip_plan* plan, int dtype) {
  if (!plan || !plan->p) return 0;
  if (dtype == GRLHIP_F32) return f32_packed_bytes(plan->p);
  // bf16 on a shape without a fused kernel runs the batch engine with bf16 MFMAs on the fp32 image
  if (dtype == GRLHIP_BF16)
    return bf16_fused_ok(plan->p) ? bf16_packed_bytes(plan->p) : wide_ok(plan->p) ? wide_packed_bytes(plan->p) : batch_bf16_packed_bytes(plan->p);
  return 0;
}

int grlhip_pack_weights(const grlhip_plan* plan, const void* const* params, int n, int dtype, void* packed,
                        void* stream) {
  if (!plan || !plan->p || !params || !packed) return fail(GRLHIP_ERR_INVALID, "null argument");
  const Plan* p = plan->p;
  if (n != (int)p->params.size()) return fail(GRLHIP_ERR_INVALID, "expected " + std::to_string(p->params.size()) + " parameters");
  for (int i = 0; i < n; ++i)
    if (!params[i]) return fail(GRLHIP_ERR_INVALID, "null parameter pointer: " + p->params[i].name);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GRLHIP_BF16 && bf16_fused_ok(p)) return bf16_pack(p, params, packed, st, g_err);
  if (dtype != GRLHIP_F32 && dtype != GRLHIP_BF16) return fail(GRLHIP_ERR_UNSUPPORTED, "unknown dtype");
  float* dst = (float*)packed;
  PackBatch pb{};
  auto flush = [&]() {
    if (pb.n) hipLaunchKernelGGL(k_pack_f32_many, dim3((unsigned)pb.first_block[pb.n]), dim3(256), 0, st, pb);
    pb.n = 0;
  };
  for (const LinearDesc& l : p->linears) {
    PackArgs& a = pb.a[pb.n];
    a = PackArgs{};
    a.src_w = (const float*)params[l.p_w];
    a.src_b = l.p_b >= 0 ? (const float*)params[l.p_b] : nullptr;
    a.dst = dst; a.N = l.N; a.K = l.K; a.Npad = l.Npad; a.Kpad = l.Kpad;
    a.col0 = l.col0; a.src_cols = l.src_cols; a.transposed = l.transposed;
    a.w_off = l.w_off; a.b_off = l.b_off;
    const int64_t total = (int64_t)l.Npad * l.Kpad + l.Npad;
    const int blocks = (int)std::min<int64_t>((total + 1023) / 1024, 256);
    if (pb.n == 0) pb.first_block[0] = 0;
    pb.first_block[pb.n + 1] = pb.first_block[pb.n] + blocks;
    if (++pb.n == PACK_BATCH) flush();
  }
  flush();
  const int half = p->desc.t_dim / 2;
  if (half)
    hipLaunchKernelGGL(k_copy_pad_f32, dim3(1), dim3(64), 0, st, (const float*)params[0], dst + p->fourier_off, half,
                       (int)(p->packed_floats - p->fourier_off));
  if (dtype == GRLHIP_F32) {   // the split image of the fp32 GEMMs (after the fp32 image on the same stream)
    SplitBatch sb{};
    auto sflush = [&]() {
      if (sb.n) hipLaunchKernelGGL(k_split_many, dim3((unsigned)sb.first_block[sb.n]), dim3(256), 0, st, sb);
      sb.n = 0;
    };
    __bf16* base = (__bf16*)((char*)packed + split_base_bytes(p));
    for (size_t i = 0; i < p->linears.size(); ++i) {
      const LinearDesc& l = p->linears[i];
      SplitArgs& a = sb.a[sb.n];
      a.src = dst + l.w_off; a.dst = base + p->split_off[i]; a.Npad = l.Npad; a.Kpad = l.Kpad;
      a.np3 = (l.Npad + 127) / 128 * 128; a.tiles = (l.Kpad + 15) / 16;
      const int64_t total = (int64_t)a.tiles * a.np3;
      const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256);
      if (sb.n == 0) sb.first_block[0] = 0;
      sb.first_block[sb.n + 1] = sb.first_block[sb.n] + blocks;
      if (++sb.n == PACK_BATCH) sflush();
    }
    sflush();
  }
  if (dtype == GRLHIP_BF16) batch_mirror_weights(p, dst, st);   // (after the fp32 image on the same stream)
  if (dtype == GRLHIP_BF16 && wide_ok(p)) wide_pack(p, dst, st);   // row-major bf16 engine of the wide nets (engine_wide.hip.inc)
  HIP_TRY(hipGetLastError());
  return 0;
}

size_t grlhip_workspace_bytes(const grlhip_plan* plan, const grlhip_plan* guidance, int64_t rows, int dtype) {
  if (!plan || !plan->p || rows <= 0) return 0;
  if (dtype == GRLHIP_BF16) {
    if (bf16_fused_ok(plan->p)) return bf16_workspace_bytes(plan->p, guidance ? guidance->p : nullptr, rows);
    const size_t need = batch_workspace_bytes(plan->p, guidance ? guidance->p : nullptr, rows);
    return wide_ok(plan->p) ? std::max(need, wide_workspace_bytes(plan->p, rows)) : need;
  }
  return batch_workspace_bytes(plan->p, guidance ? guidance->p : nullptr, rows);
}

int grlhip_net_forward(const grlhip_plan* plan, const void* packed, int dtype, const float* t, const float* x,
                       const float* cond, float* out, int64_t rows, void* workspace, size_t workspace_bytes,
                       void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (rows <= 0) return 0;
  const bool qnet = p->desc.kind == GRLHIP_NET_QNET;
  if ((!t && !qnet) || !x || !out) return fail(GRLHIP_ERR_INVALID, "null t/x/out");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "net_forward: only GRLHIP_F32 is implemented");
  const size_t need = grlhip_workspace_bytes(plan, nullptr, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  hipStream_t st = (hipStream_t)stream;
  batch_forward(p, (const float*)packed, t, x, cond, out, rows, 0, 1.0f, (float*)workspace, st);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_energy_grad(const grlhip_plan* plan, const void* packed, int dtype, const float* t, const float* x,
                       const float* cond, float scale, float* grad, int64_t rows, void* workspace,
                       size_t workspace_bytes, void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (p->desc.kind != GRLHIP_NET_CONCAT_MLP) return fail(GRLHIP_ERR_INVALID, "energy_grad needs a CONCAT_MLP plan");
  if (rows <= 0) return 0;
  if (!t || !x || !grad) return fail(GRLHIP_ERR_INVALID, "null t/x/grad");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "energy_grad: only GRLHIP_F32 is implemented");
  const size_t need = grlhip_workspace_bytes(plan, nullptr, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  hipStream_t st = (hipStream_t)stream;
  batch_forward(p, (const float*)packed, t, x, cond, grad, rows, 1, scale, (float*)workspace, st);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int sample_common(const grlhip_plan* plan, const void* packed, int dtype, const grlhip_plan* guidance,
                         const void* guidance_packed, const float* x0, const float* cond, const float* coef,
                         const float* dW, int32_t n_steps, int64_t rows, float* out_final, float* out_traj,
                         void* workspace, size_t workspace_bytes, void* stream, bool stages = false) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (p->desc.kind != GRLHIP_NET_RESNET) return fail(GRLHIP_ERR_INVALID, "sampling needs a RESNET plan");
  if (p->desc.out_dim != p->desc.x_dim) return fail(GRLHIP_ERR_INVALID, "sampling needs out_dim == x_dim");
  if (rows <= 0) return 0;
  if (n_steps < 0) return fail(GRLHIP_ERR_INVALID, "n_steps < 0");
  if (!x0 || !out_final || (n_steps > 0 && !coef)) return fail(GRLHIP_ERR_INVALID, "null x0/out_final/coef");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  const Plan* g = nullptr;
  if (guidance) {
    if (int rc = check_ready(guidance, guidance_packed)) return rc;
    g = guidance->p;
    if (g->desc.kind != GRLHIP_NET_CONCAT_MLP || g->desc.x_dim != p->desc.x_dim || g->desc.cond_dim != p->desc.cond_dim)
      return fail(GRLHIP_ERR_INVALID, "guidance plan must be a CONCAT_MLP over the same x/cond dims");
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GRLHIP_BF16) {
    const bool nf = bf16_fused_ok(p), gf = g && bf16_fused_ok(g);
    if (nf)  // the persistent fused integrator (rejects pairs it has no kernel for)
      return bf16_sample(p, packed, g, guidance_packed, x0, cond, coef, dW, n_steps, rows, out_final, out_traj,
                         workspace, workspace_bytes, st, g_err, stages);
    if (gf) return fail(GRLHIP_ERR_UNSUPPORTED, "bf16: this energy net only has a fused kernel but the sampler net has none; use GRLHIP_F32");
    // any other shape: the batch engine with bf16 MFMAs (both images are in the fp32 layout)
    const size_t need = grlhip_workspace_bytes(plan, guidance, rows, dtype);
    if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
    SampleCall c{x0, cond, coef, dW, out_final, out_traj, rows, n_steps};
    if (!g && !stages && wide_ok(p)) wide_sample(p, (const float*)packed, c, workspace, st);   // wide nets: row-major bf16 slab
    else batch_sample(p, (const float*)packed, g, (const float*)guidance_packed, c, (float*)workspace, st, true, stages);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "unknown dtype");
  const size_t need = grlhip_workspace_bytes(plan, guidance, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  SampleCall c{x0, cond, coef, dW, out_final, out_traj, rows, n_steps};
  batch_sample(p, (const float*)packed, g, (const float*)guidance_packed, c, (float*)workspace, st, false, stages);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_sample_ode(const grlhip_plan* plan, const void* packed, int dtype, const grlhip_plan* guidance,
                      const void* guidance_packed, const float* x0, const float* cond, const float* coef,
                      int32_t n_steps, int64_t rows, float* out_final, float* out_traj, void* workspace,
                      size_t workspace_bytes, void* stream) {
  return sample_common(plan, packed, dtype, guidance, guidance_packed, x0, cond, coef, nullptr, n_steps, rows,
                       out_final, out_traj, workspace, workspace_bytes, stream);
}

int grlhip_sample_fixed(const grlhip_plan* plan, const void* packed, int dtype, const grlhip_plan* guidance,
                        const void* guidance_packed, const float* x0, const float* cond, const float* coef,
                        const float* dW, int32_t n_steps, int64_t rows, const float* fixed_x, const float* fixed_mask,
                        const float* fix_sc, const float* fix_noise, float* out_final, float* out_traj,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (p->desc.kind != GRLHIP_NET_RESNET) return fail(GRLHIP_ERR_INVALID, "sampling needs a RESNET plan");
  if (p->desc.out_dim != p->desc.x_dim) return fail(GRLHIP_ERR_INVALID, "sampling needs out_dim == x_dim");
  if (rows <= 0) return 0;
  if (n_steps < 0) return fail(GRLHIP_ERR_INVALID, "n_steps < 0");
  if (!x0 || !out_final || (n_steps > 0 && (!coef || !fix_sc))) return fail(GRLHIP_ERR_INVALID, "null x0/out_final/coef/fix_sc");
  if (!fixed_x || !fixed_mask) return fail(GRLHIP_ERR_INVALID, "null fixed_x / fixed_mask");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  const Plan* g = nullptr;
  if (guidance) {
    if (int rc = check_ready(guidance, guidance_packed)) return rc;
    g = guidance->p;
    if (g->desc.kind != GRLHIP_NET_CONCAT_MLP || g->desc.x_dim != p->desc.x_dim || g->desc.cond_dim != p->desc.cond_dim)
      return fail(GRLHIP_ERR_INVALID, "guidance plan must be a CONCAT_MLP over the same x/cond dims");
  }
  if (dtype != GRLHIP_F32 && dtype != GRLHIP_BF16) return fail(GRLHIP_ERR_UNSUPPORTED, "unknown dtype");
  if (dtype == GRLHIP_BF16 && (bf16_fused_ok(p) || (g && bf16_fused_ok(g))))
    return fail(GRLHIP_ERR_UNSUPPORTED, "sample_fixed runs on the batch engine: the fused bf16 integrator has no fixed-component form; use GRLHIP_F32 images for this network");
  const size_t need = grlhip_workspace_bytes(plan, guidance, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  SampleCall c{x0, cond, coef, dW, out_final, out_traj, rows, n_steps};
  c.fixed = fixed_x; c.mask = fixed_mask; c.fix_sc = fix_sc; c.fix_noise = fix_noise;
  batch_sample(p, (const float*)packed, g, (const float*)guidance_packed, c, (float*)workspace, (hipStream_t)stream, dtype == GRLHIP_BF16, false);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_bf16_fused_kernel(const grlhip_plan* plan, const grlhip_plan* guidance, char* name_buf, size_t name_buf_len) {
  if (!plan || !plan->p) return fail(GRLHIP_ERR_INVALID, "null plan");
  const Plan* g = guidance ? guidance->p : nullptr;
  const bool fused = bf16_fused_ok(plan->p);
  const char* nm = fused ? bf16_kernel_name(plan->p, g)
                   : (!g && wide_ok(plan->p)) ? "layer engine: k_w_gemm (row-major bf16 activations; Euler / Euler-Maruyama calls)"
                                              : "layer engine: k_bt_gemm_bf16 (batch engine, fp32 activations)";
  if (name_buf && name_buf_len) snprintf(name_buf, name_buf_len, "%s", nm);
  return fused && nm[0] ? 1 : 0;
}

// ---- allocation-free variant of grlhip_sample_ode (include/grlhip.h): the fused bf16 integrator with its
// workspace (per-step Fourier tables, spliced guided weight stream) in a file-scope device array instead of a
// caller workspace.  In host code `g_static_scratch` is only a stand-in for the symbol: the device address is
// looked up once per device with hipGetSymbolAddress and only that pointer is ever passed to launches / copies.
static __device__ __align__(256) unsigned char g_static_scratch[GRLHIP_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {nullptr};

int static_scratch(void** out) {
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(GRLHIP_ERR_UNSUPPORTED, "device ordinal out of range for the static scratch table");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {  // first call on this device, which is the current one
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_static_scratch)));
    if (!ptr) return fail(GRLHIP_ERR_HIP, "hipGetSymbolAddress returned a null address for the static scratch");
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace

size_t grlhip_static_scratch_bytes(void) { return (size_t)GRLHIP_STATIC_SCRATCH_BYTES; }

int grlhip_sample_ode_static(const grlhip_plan* plan, const void* packed, const grlhip_plan* guidance,
                             const void* guidance_packed, const float* x0, const float* cond, const float* coef,
                             int32_t n_steps, int64_t rows, float* out_final, float* out_traj, void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  if (!bf16_fused_ok(plan->p))
    return fail(GRLHIP_ERR_UNSUPPORTED, "sample_ode_static: only networks with a fused bf16 integrator (GRLHIP_BF16 image) are supported");
  if (rows > 0) {
    const size_t need = grlhip_workspace_bytes(plan, guidance, rows, GRLHIP_BF16);
    if (need > (size_t)GRLHIP_STATIC_SCRATCH_BYTES)
      return fail(GRLHIP_ERR_WORKSPACE, "sample_ode_static: this call needs " + std::to_string(need) + " bytes of scratch, the static scratch holds " +
                                            std::to_string((size_t)GRLHIP_STATIC_SCRATCH_BYTES));
  }
  void* scratch = nullptr;
  if (int rc = static_scratch(&scratch)) return rc;
  // every byte of the scratch the kernels read is rewritten by this call first (Fourier tables by k_temb_table, the
  // spliced stream by the copies + memset in bf16_sample): nothing has to start at zero
  return sample_common(plan, packed, GRLHIP_BF16, guidance, guidance_packed, x0, cond, coef, nullptr, n_steps, rows,
                       out_final, out_traj, scratch, (size_t)GRLHIP_STATIC_SCRATCH_BYTES, stream);
}

int grlhip_sample_sde(const grlhip_plan* plan, const void* packed, int dtype, const grlhip_plan* guidance,
                      const void* guidance_packed, const float* x0, const float* cond, const float* coef,
                      const float* dW, int32_t n_steps, int64_t rows, float* out_final, float* out_traj,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!dW && n_steps > 0) return fail(GRLHIP_ERR_INVALID, "sample_sde needs the Brownian increment buffer dW");
  return sample_common(plan, packed, dtype, guidance, guidance_packed, x0, cond, coef, dW, n_steps, rows,
                       out_final, out_traj, workspace, workspace_bytes, stream);
}

int grlhip_sample_stages(const grlhip_plan* plan, const void* packed, int dtype, const grlhip_plan* guidance,
                         const void* guidance_packed, const float* x0, const float* cond, const float* stage_table,
                         int32_t n_stages, int64_t rows, float* out_final, float* out_traj, void* workspace,
                         size_t workspace_bytes, void* stream) {
  return sample_common(plan, packed, dtype, guidance, guidance_packed, x0, cond, stage_table, nullptr, n_stages, rows,
                       out_final, out_traj, workspace, workspace_bytes, stream, true);
}

size_t grlhip_loss_workspace_bytes(const grlhip_plan* plan, int64_t rows, int dtype) {
  if (!plan || !plan->p || rows <= 0 || dtype != GRLHIP_F32 || plan->p->train_fwd.empty()) return 0;
  return train_workspace_bytes(plan->p, rows);
}

int grlhip_loss_fwd_bwd(const grlhip_plan* plan, const void* packed, int dtype, const float* t, const float* x_t,
                        const float* cond, const float* alpha, const float* beta, const float* target,
                        const float* row_weight, float* loss_rows, float* grad_flat, int64_t rows, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  const bool qnet = p->desc.kind == GRLHIP_NET_QNET;
  if ((p->desc.kind != GRLHIP_NET_RESNET && !qnet) || p->train_fwd.empty())
    return fail(GRLHIP_ERR_UNSUPPORTED, "loss_fwd_bwd needs a RESNET or QNET plan");
  if (!qnet && p->desc.out_dim != p->desc.x_dim) return fail(GRLHIP_ERR_INVALID, "loss_fwd_bwd needs out_dim == x_dim");
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "loss_fwd_bwd: only GRLHIP_F32 is implemented");
  if (rows <= 0) return 0;
  if (qnet) { t = nullptr; beta = nullptr; }
  if ((!t && !qnet) || !x_t || !alpha || (!beta && !qnet) || !target || !loss_rows) return fail(GRLHIP_ERR_INVALID, "null t/x_t/alpha/beta/target/loss_rows");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  const size_t need = grlhip_loss_workspace_bytes(plan, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  hipStream_t st = (hipStream_t)stream;
  TrainCall c{t, x_t, cond, alpha, beta, target, row_weight, loss_rows, grad_flat, rows};
  train_step(p, (const float*)packed, c, (float*)workspace, st);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_cep_loss_fwd_bwd(const grlhip_plan* plan, const void* packed, int dtype, const float* t, const float* x_t,
                            const float* cond, const float* logits0, int32_t group_size, const float* group_weight,
                            float* loss_groups, float* grad_flat, int64_t rows, void* workspace, size_t workspace_bytes,
                            void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (p->desc.kind != GRLHIP_NET_CONCAT_MLP || p->train_fwd.empty())
    return fail(GRLHIP_ERR_UNSUPPORTED, "cep_loss_fwd_bwd needs a CONCAT_MLP plan");
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "cep_loss_fwd_bwd: only GRLHIP_F32 is implemented");
  if (group_size < 1 || group_size > TRAIN_CHUNK_ROWS) return fail(GRLHIP_ERR_INVALID, "group_size must be in [1, 65536]");
  if (rows < 0 || rows % group_size) return fail(GRLHIP_ERR_INVALID, "rows must be a multiple of group_size");
  if (rows == 0) return 0;
  if (!t || !x_t || !logits0 || !loss_groups) return fail(GRLHIP_ERR_INVALID, "null t/x_t/logits0/loss_groups");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  const size_t need = grlhip_loss_workspace_bytes(plan, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  CepCall c{t, x_t, cond, logits0, group_weight, loss_groups, grad_flat, rows, group_size};
  cep_step(p, (const float*)packed, c, (float*)workspace, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_net_backward(const grlhip_plan* plan, const void* packed, int dtype, const float* t, const float* x,
                        const float* cond, const float* grad_out, float* grad_flat, float* grad_x, int64_t rows,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (p->train_fwd.empty()) return fail(GRLHIP_ERR_UNSUPPORTED, "net_backward: the plan has no training programs");
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "net_backward: only GRLHIP_F32 is implemented");
  if (rows <= 0) return 0;
  const bool qnet = p->desc.kind == GRLHIP_NET_QNET;
  if (qnet) t = nullptr;
  if ((!t && !qnet) || !x || !grad_out || (!grad_flat && !grad_x)) return fail(GRLHIP_ERR_INVALID, "null t/x/grad_out, or neither grad_flat nor grad_x");
  if (grad_x && p->tr_gx < 0) return fail(GRLHIP_ERR_UNSUPPORTED, "net_backward: this plan has no input-gradient program");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  const size_t need = grlhip_loss_workspace_bytes(plan, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  backward_step(p, (const float*)packed, t, x, cond, grad_out, grad_flat, grad_x, rows, (float*)workspace, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_sample_ode_backward(const grlhip_plan* plan, const void* packed, int dtype, const float* traj,
                               const float* cond, const float* coef, int32_t n_steps, int64_t rows,
                               const float* grad_traj, float* grad_x0, float* grad_flat, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (p->desc.kind != GRLHIP_NET_RESNET || p->tr_gx < 0) return fail(GRLHIP_ERR_INVALID, "sample_ode_backward needs a RESNET plan");
  if (p->desc.out_dim != p->desc.x_dim) return fail(GRLHIP_ERR_INVALID, "sample_ode_backward needs out_dim == x_dim");
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "sample_ode_backward: only GRLHIP_F32 is implemented");
  if (rows <= 0 || n_steps <= 0) return 0;
  if (!traj || !coef || !grad_traj || !grad_flat) return fail(GRLHIP_ERR_INVALID, "null traj/coef/grad_traj/grad_flat");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  const size_t need = grlhip_loss_workspace_bytes(plan, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  OdeBwdCall c{traj, cond, coef, grad_traj, grad_x0, grad_flat, rows, n_steps};
  ode_backward(p, (const float*)packed, c, (float*)workspace, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_sample_stages_backward(const grlhip_plan* plan, const void* packed, int dtype, const float* states,
                                  const float* cond, const float* stage_table, int32_t n_stages, int64_t rows,
                                  const float* grad_states, float* grad_x0, float* grad_flat, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (int rc = check_ready(plan, packed)) return rc;
  const Plan* p = plan->p;
  if (p->desc.kind != GRLHIP_NET_RESNET || p->tr_gx < 0) return fail(GRLHIP_ERR_INVALID, "sample_stages_backward needs a RESNET plan");
  if (p->desc.out_dim != p->desc.x_dim) return fail(GRLHIP_ERR_INVALID, "sample_stages_backward needs out_dim == x_dim");
  if (dtype != GRLHIP_F32) return fail(GRLHIP_ERR_UNSUPPORTED, "sample_stages_backward: only GRLHIP_F32 is implemented");
  if (rows <= 0 || n_stages <= 0) return 0;
  if (!states || !stage_table || !grad_states || !grad_flat) return fail(GRLHIP_ERR_INVALID, "null states/stage_table/grad_states/grad_flat");
  if ((p->desc.cond_dim > 0) != (cond != nullptr))
    return fail(GRLHIP_ERR_INVALID, "condition pointer does not match the plan's cond_dim");
  const size_t need = grlhip_loss_workspace_bytes(plan, rows, dtype);
  if (!workspace || workspace_bytes < need) return fail(GRLHIP_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  StagesBwdCall c{states, cond, stage_table, grad_states, grad_x0, grad_flat, rows, n_stages};
  stages_backward(p, (const float*)packed, c, (float*)workspace, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
  if (n <= 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq) return fail(GRLHIP_ERR_INVALID, "null params/grads/exp_avg/exp_avg_sq");
  if (step < 1) return fail(GRLHIP_ERR_INVALID, "adam_step: step counts from 1");
  // bias corrections in double on the host, as torch.optim.Adam computes them for a Python-float step
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
  AdamArgs a{params, grads, exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay, (float)(lr / bc1), (float)std::sqrt(bc2), nullptr};
  const int64_t blocks = std::min<int64_t>((n + 1023) / 1024, 4096);
  hipLaunchKernelGGL(k_adam_step, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

int grlhip_adam_step_counted(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, void* step_record, int advance,
                             void* stream) {
  if (!step_record || ((size_t)step_record & 7)) return fail(GRLHIP_ERR_INVALID, "adam_step_counted: step_record must be an 8-byte aligned device pointer to 16 bytes");
  if (advance) hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, (AdamRecord*)step_record, lr, beta1, beta2);
  if (n > 0) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail(GRLHIP_ERR_INVALID, "null params/grads/exp_avg/exp_avg_sq");
    AdamArgs a{params, grads, exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay, 0.f, 1.f, &((const AdamRecord*)step_record)->step_size};
    const int64_t blocks = std::min<int64_t>((n + 1023) / 1024, 4096);
    hipLaunchKernelGGL(k_adam_step, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"

#ifdef GEMM_STAMPS
extern "C" int grlhip_dbg_gemm(unsigned long long* dst, int reset) {   // experiment builds only (scripts/gemm_clock.py)
  if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(grlhip::bt::gemm_dbg), z, sizeof z); }
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(grlhip::bt::gemm_dbg), sizeof(unsigned long long) * 8);
}
#endif
#ifdef WIDE_STAMPS
extern "C" int grlhip_dbg_wide(unsigned long long* dst) {   // experiment builds only
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(grlhip::wd::wide_dbg), sizeof(unsigned long long) * 1024 * 8);
}
#endif
