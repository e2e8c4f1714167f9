/* grlhip -- C ABI of the MI355X-native generative-policy hot path.
 *
 * The reference (zjowowen/GenerativeRL_Preview) is pure Python and has no FFI; this ABI is what
 * the Python classes that mirror its API (generativerl_preview_amd/) bind through ctypes, one
 * entry point per reference code path (cited per function, paths relative to the reference root).
 *
 * Conventions: every function returns 0 on success or a negative grlhip_status; the text of the
 * last failure on the calling thread is available from grlhip_last_error().  Nothing throws.
 * All data pointers are DEVICE pointers (fp32, row-major, contiguous) unless a comment says
 * "host"; the library never allocates or frees caller memory: scratch space is a caller-owned
 * workspace whose size comes from grlhip_workspace_bytes().  Every launch goes to the explicit
 * `stream` argument (a hipStream_t passed as void*; NULL = the null stream).  Calls on distinct
 * plans/workspaces may run concurrently; a workspace must not be shared by overlapping calls.
 */
#ifndef GRLHIP_H
#define GRLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRLHIP_MAX_HIDDEN 8

typedef enum {
  GRLHIP_OK = 0,
  GRLHIP_ERR_INVALID = -1,     /* bad argument / inconsistent sizes */
  GRLHIP_ERR_UNSUPPORTED = -2, /* shape or mode this build has no kernel for */
  GRLHIP_ERR_HIP = -3,         /* a HIP runtime call failed */
  GRLHIP_ERR_WORKSPACE = -4    /* workspace missing or too small */
} grlhip_status;

typedef enum {
  GRLHIP_NET_RESNET = 0,    /* TemporalSpatialResidualNet, grl/neural_network/__init__.py:237-342 */
  GRLHIP_NET_CONCAT_MLP = 1, /* ConcatenateMLP(SiLU) -> 1 output, grl/neural_network/__init__.py:558-583 */
  GRLHIP_NET_QNET = 2        /* QNetwork, grl/rl_modules/value_network/q_network.py:12-64: ConcatenateMLP(ReLU) over
                                cat[action, state] -> 1, no time input (fp32 batch engine only) */
} grlhip_net_kind;

typedef enum {
  GRLHIP_F32 = 0, /* fp32 storage, fp32-input MFMA (exact fp32 fma chains) */
  GRLHIP_BF16 = 1 /* sampling only: bf16 MFMA operands, fp32 accumulate, fp32 state and schedule math.
                     [512,256,128] nets (+ the [256,256] energy net) run the persistent fused integrator,
                     every other shape the layer-by-layer batch engine with bf16 MFMAs. */
} grlhip_dtype;

/* Network description = the constructor arguments of the reference module.
 * RESNET:     x_dim -> hidden[0..n_hidden) U-Net of TemporalSpatialResBlocks -> out_dim; t_dim is the
 *             width of the GaussianFourierProjectionTimeEncoder output (encoders.py:36-86);
 *             cond_dim == 0 means unconditional (then cond_hidden is ignored, t_cond_hidden = t_dim).
 * CONCAT_MLP: input cat[t_emb(t_dim), x(x_dim), c(cond_dim)] -> hidden[0..n_hidden) with SiLU -> out_dim (1).
 * QNET:       input cat[action(x_dim), state(cond_dim)] -> hidden[0..n_hidden) with ReLU -> out_dim (1); t_dim = 0,
 *             the `t` argument of the entry points is ignored (may be NULL). */
typedef struct {
  int32_t kind;
  int32_t x_dim;
  int32_t out_dim;
  int32_t cond_dim;
  int32_t t_dim;
  int32_t cond_hidden;
  int32_t t_cond_hidden;
  int32_t n_hidden;
  int32_t hidden[GRLHIP_MAX_HIDDEN];
} grlhip_net_desc;

typedef struct grlhip_plan grlhip_plan; /* opaque; immutable after creation */

const char* grlhip_last_error(void);
/* 1 when a HIP device is usable by this process, else 0 (never fails). */
int grlhip_device_available(void);

/* Replaces: module construction through get_module()/get_encoder(),
 * grl/generative_models/intrinsic_model.py:19-52.  Builds the op program, the activation-slab
 * layout and the packed-weight layout for `net`; uploads the (tiny) program to the current device. */
int grlhip_plan_create(const grlhip_net_desc* net, grlhip_plan** out_plan);
void grlhip_plan_destroy(grlhip_plan* plan);

/* The parameters the plan consumes, in packing order, named as in the reference state_dict
 * relative to the IntrinsicModel's ModuleDict ("t_encoder.W", "backbone.sort_t.0.weight", ...;
 * SURVEY.md section 5 "Checkpoint").  rows/cols: cols == 0 for vectors. */
int grlhip_plan_num_params(const grlhip_plan* plan);
int grlhip_plan_param_info(const grlhip_plan* plan, int index, char* name_buf, size_t name_buf_len,
                           int32_t* rows, int32_t* cols);

/* Replaces: the `requires_grad` argument of GaussianFourierProjectionTimeEncoder (grl/neural_network/encoders.py:53-66;
 * False by default and in every shipped configuration): with enable == 0 the training entry points
 * (grlhip_loss_fwd_bwd, grlhip_cep_loss_fwd_bwd, grlhip_net_backward, grlhip_sample_*_backward) leave the gradient slot
 * of "t_encoder.W" untouched and skip the work that only feeds it.  Default: enabled.  The one mutable property of a
 * plan: do not change it while a call on the same plan is in progress on another thread. */
int grlhip_plan_set_fourier_grad(grlhip_plan* plan, int enable);

/* Introspection for the tests (no counterpart in the reference): the engine launches some ops of a program out of program
 * order -- weight gradients are held back and launched together, two independent GEMMs go as one launch.  This replays
 * the launch schedule of a program (0..3 as in grlhip_plan_program_info) without launching anything and checks it against
 * the program's sequential meaning (every read-after-write, write-after-write and write-after-read pair of ops keeps
 * its order).  Returns 0, or 1 + the index of the first op that would run too early (negative: invalid arguments). */
int grlhip_plan_schedule_check(const grlhip_plan* plan, int program);

/* No counterpart in the reference (its GPU matmuls are whatever precision torch is configured for).  The fp32 engine
 * forms every fp32 product from bf16 pieces on the matrix cores: three pieces per value, six MFMAs per product = fp32
 * rounding level.  pieces == 2 (three MFMAs per product, |error| <= ~2^-16 of each product,
 * random sign) applies to the GEMMs of the BACKWARD programs only -- weight gradients and input gradients of
 * grlhip_loss_fwd_bwd, grlhip_cep_loss_fwd_bwd, grlhip_net_backward, grlhip_sample_*_backward; forwards, losses and
 * sampling are not affected.  Gradients stay bit-reproducible.  Default 3.  Change it only while no call on the plan
 * is in progress. */
int grlhip_plan_set_backward_split(grlhip_plan* plan, int pieces);

/* Introspection (no counterpart in the reference; used by the tests and to size stream captures): the number of
 * ops of one of the plan's programs (0 forward, 1 energy forward + input gradient, 2 training forward,
 * 3 training backward) and the number of kernel launches the batch engine turns it into per pass (a plain
 * linear whose only consumer is the residual input of the next linear is fused into that GEMM). */
int grlhip_plan_program_info(const grlhip_plan* plan, int program, int32_t* n_ops, int32_t* n_launches);

/* Introspection (no counterpart in the reference): 1 when GRLHIP_BF16 sampling of `plan` (energy-guided by
 * `guidance`, may be NULL) runs the persistent fused integrator k_sample_bf16 -- any conditional or unconditional
 * [512,256,128] TemporalSpatialResidualNet with at most 16 state features (guided: at most 32 condition features and
 * a [256,256] energy net) -- and 0 when it runs layer by layer.  name_buf (optional) receives the kernel class; for
 * layer-by-layer plans the engine: the row-major bf16 engine of wide nets (k_w_gemm: a hidden width >= 256, all widths
 * multiples of 8, unguided Euler / Euler-Maruyama calls) or the batch engine (k_bt_gemm_bf16). */
int grlhip_bf16_fused_kernel(const grlhip_plan* plan, const grlhip_plan* guidance, char* name_buf, size_t name_buf_len);

/* Bytes of the packed weight image for `dtype` (the image layout is private to the library and differs
 * per dtype and shape: always produce it with grlhip_pack_weights for the same plan and dtype; GRLHIP_BF16 of a
 * network without a fused integrator is the fp32 image followed by a bf16 mirror of its matrices). */
size_t grlhip_packed_bytes(const grlhip_plan* plan, int dtype);
/* Derive the packed (padded / transposed / fragment-ordered) weight image from the reference-layout
 * parameters.  `params` is a HOST array of n DEVICE pointers (fp32), in grlhip_plan_param_info order. */
int grlhip_pack_weights(const grlhip_plan* plan, const void* const* params, int n, int dtype,
                        void* packed, void* stream);

/* Workspace bytes for a call over `rows` rows (sampling / forward); `guidance` may be NULL.  fp32 calls
 * stream activations through the workspace (a feature-major slab of up to 262 144 rows: gigabytes for the
 * QGPO net); the fused bf16 integrator needs < 5 MB. */
size_t grlhip_workspace_bytes(const grlhip_plan* plan, const grlhip_plan* guidance, int64_t rows, int dtype);

/* Replaces: IntrinsicModel.forward, grl/generative_models/intrinsic_model.py:54-75 --
 * out[R,out_dim] = backbone(fourier(t[R]), x[R,x_dim], cond[R,cond_dim] | NULL). */
int grlhip_net_forward(const grlhip_plan* plan, const void* packed, int dtype, const float* t,
                       const float* x, const float* cond, float* out, int64_t rows,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: EnergyGuidance.calculate_energy_guidance,
 * grl/generative_models/diffusion_model/energy_conditional_diffusion_model.py:67-92 --
 * grad[R,x_dim] = scale * d/dx sum_rows E(t, x, cond) for a CONCAT_MLP plan. */
int grlhip_energy_grad(const grlhip_plan* plan, const void* packed, int dtype, const float* t,
                       const float* x, const float* cond, float scale, float* grad, int64_t rows,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Per-step coefficients of the fused integrator, a HOST-computed DEVICE table coef[n_steps][GRLHIP_COEF_STRIDE]:
 *   [0] tau    time fed to the network at this step
 *   [1] dt     step length
 *   [2] a      x_{k+1} = x_k + dt * (a*x_k + b*net(tau,x_k,c) + g*dE/dx(tau,x_k,c)) + s*dW_k
 *   [3] b
 *   [4] g      guidance coefficient (0.5*g^2(tau)*guidance_scale for the reverse ODE)
 *   [5] s      diffusion coefficient multiplying the caller's Brownian increment (0 for ODEs)
 *   [6] emit   >= 0: after this step copy the state to out_traj[emit] (only if out_traj != NULL);
 *              stored as float
 *   [7] lerp   weight w for the emitted state: out = x_prev + w*(x_new - x_prev) (1 = x_new),
 *              reproducing torchsde's interpolation to requested output times.
 *   [8] emit2, [9] lerp2   a second emission of the same step (emit2 < 0: none); when a step spans
 *              more than two requested output times the caller interpolates the rest from these two.
 *   [10], [11] reserved (0).
 * This covers every (model head x path x solver) closure of DiffusionProcess.reverse_ode / reverse_sde
 * (grl/generative_models/diffusion_process.py:742-919) and the ICFM drift
 * (conditional_flow_model/independent_conditional_flow_model.py:285-288). */
#define GRLHIP_COEF_STRIDE 12

/* Replaces: ODESolver.integrate with Euler on the caller's grid
 * (grl/numerical_methods/numerical_solvers/ode_solver.py:51-79,168-185) driven from
 * DiffusionModel.sample_forward_process (diffusion_model/diffusion_model.py:282-304) /
 * EnergyConditionalDiffusionModel (:406-435) / IndependentConditionalFlowModel (:285-319):
 * all n_steps Euler steps of all rows enqueued by ONE call (one launch on the fused bf16 path, one launch
 * per layer and step on the batch engine; never a host synchronisation).  guidance/guidance_packed may be NULL.
 * out_final[R,D] always written; out_traj[n_emit+1, R, D] optional (slot 0 = x0). */
int grlhip_sample_ode(const grlhip_plan* plan, const void* packed, int dtype,
                      const grlhip_plan* guidance, const void* guidance_packed,
                      const float* x0, const float* cond, const float* coef, int32_t n_steps,
                      int64_t rows, float* out_final, float* out_traj,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: the Euler / Euler-Maruyama loop of DiffusionModel.sample_forward_process_with_fixed_x
 * (grl/generative_models/diffusion_model/diffusion_model.py:521-772; energy-guided:
 * diffusion_model/energy_conditional_diffusion_model.py:669-960) -- sampling with some components pinned
 * ("inpainting"): fixed_mask[R,D] is 1 where a component is generated and 0 where it is pinned to fixed_x[R,D]
 * (any value in between blends, as the reference's arithmetic does).
 *   state_0 = fixed_x*(1 - mask) + x0*mask
 *   xin_k   = (s_k*fixed_x + sigma_k*noise_k)*(1 - mask) + state_k*mask      (diffusion_process.direct_sample, :939-956)
 *   state_{k+1} = state_k + mask * (dt*(a*xin_k + b*net(tau, xin_k, c) + g*dE/dx(tau, xin_k, c)) + s*dW_k)
 * with the coef table of grlhip_sample_ode / _sde, fix_sc[n_steps][2] = {s_k, sigma_k} (scale and std of the forward
 * path at the step's network time) and fix_noise[n_steps][R][D] the Gaussian draws of direct_sample (NULL = zeros; the
 * reference draws them inside the drift, so passing them is what makes a run reproducible).  dW may be NULL (ODE).
 * Runs on the batch engine (GRLHIP_F32, or GRLHIP_BF16 for networks without a fused integrator; a network whose
 * GRLHIP_BF16 image is the fused integrator's: GRLHIP_ERR_UNSUPPORTED).  Workspace: grlhip_workspace_bytes.
 * out_final[R,D] = state after the last step; out_traj as in grlhip_sample_ode (slot 0 = state_0). */
int grlhip_sample_fixed(const grlhip_plan* plan, const void* packed, int dtype,
                        const grlhip_plan* guidance, const void* guidance_packed,
                        const float* x0, const float* cond, const float* coef, const float* dW, int32_t n_steps,
                        int64_t rows, const float* fixed_x, const float* fixed_mask, const float* fix_sc,
                        const float* fix_noise, float* out_final, float* out_traj,
                        void* workspace, size_t workspace_bytes, void* stream);

/* No counterpart in the reference.  For callers that cannot allocate device memory or hand over a workspace --
 * tight serving loops, harnesses that forbid hipMalloc, code that runs under HIP graph capture (make the first call on
 * each device OUTSIDE a capture: it looks up the array's device address with hipGetSymbolAddress, which may load the
 * code object; after that the call captures and replays, guided or not: tests/test_zz_no_alloc.py): grlhip_sample_ode on the fused
 * bf16 integrator (GRLHIP_BF16 images of networks that have one; anything else: GRLHIP_ERR_UNSUPPORTED) with the
 * workspace (per-step Fourier tables, spliced guided weight stream) in a device array of GRLHIP_STATIC_SCRATCH_BYTES
 * that is part of the library's code object.  Each GPU the library runs on holds its own copy until the process ends.
 * The call performs no hipMalloc / hipFree; a call whose scratch need (grlhip_workspace_bytes for GRLHIP_BF16)
 * exceeds the array returns GRLHIP_ERR_WORKSPACE.  Every call on a GPU shares that GPU's array: calls that overlap
 * on one GPU (different streams or threads) race -- serialise them, or use grlhip_sample_ode with workspaces. */
#define GRLHIP_STATIC_SCRATCH_BYTES (6u << 20)
size_t grlhip_static_scratch_bytes(void);
int grlhip_sample_ode_static(const grlhip_plan* plan, const void* packed,
                             const grlhip_plan* guidance, const void* guidance_packed,
                             const float* x0, const float* cond, const float* coef, int32_t n_steps,
                             int64_t rows, float* out_final, float* out_traj, void* stream);

/* Replaces: SDESolver.integrate -> torchsde.sdeint(method="euler")
 * (grl/numerical_methods/numerical_solvers/sde_solver.py:92-144) for the reverse SDE closures
 * (diffusion_process.py:742-851): Euler-Maruyama y += f*dt + g*dW with the Brownian increments read
 * from the caller's buffer dW[n_steps, R, D] (already scaled by sqrt(dt)), so runs are reproducible. */
int grlhip_sample_sde(const grlhip_plan* plan, const void* packed, int dtype,
                      const grlhip_plan* guidance, const void* guidance_packed,
                      const float* x0, const float* cond, const float* coef, const float* dW,
                      int32_t n_steps, int64_t rows, float* out_final, float* out_traj,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: DPMSolver.integrate, method "singlestep", orders 1-3, both solver types
 * (grl/numerical_methods/numerical_solvers/dpm_solver.py:250-523,854-904) and any other solver whose step is a
 * sequence of STAGES, each one evaluation of the network (and energy gradient) followed by an affine update with
 * one saved state and two saved model outputs.  stage_table[n_stages][GRLHIP_COEF_STRIDE], a HOST-computed DEVICE
 * table with the layout
 *   [0] tau   time fed to the network(s) at this stage
 *   [5] cn, [6] cx, [7] cg     m = cn*net(tau,x,c) + cx*x + cg*dE/dx(tau,x,c)   (the solver's "model" output: the
 *                              head conversion and the energy guidance folded into three scalars)
 *   [1] P, [2] Q, [3] R, [4] W, [10] W2    x_new = P*x + Q*xs + R*m + W*ms + W2*ms2
 *   [8] save  1: afterwards xs = x (the state BEFORE the update), ms = m;  2: afterwards ms2 = m;  0: nothing
 *             (xs = ms = ms2 = 0 initially)
 *   [9] emit  >= 0: out_traj[emit] = x_new (stored as float; only if out_traj != NULL; slot 0 = x0)
 *   [11] reserved (0).
 * One call runs every stage of every row (one launch on the fused bf16 path). */
int grlhip_sample_stages(const grlhip_plan* plan, const void* packed, int dtype,
                         const grlhip_plan* guidance, const void* guidance_packed,
                         const float* x0, const float* cond, const float* stage_table, int32_t n_stages,
                         int64_t rows, float* out_final, float* out_traj,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Workspace bytes for grlhip_loss_fwd_bwd over `rows` rows. */
size_t grlhip_loss_workspace_bytes(const grlhip_plan* plan, int64_t rows, int dtype);
/* Number of fp32 elements of the flat gradient buffer (all parameters in param_info order). */
int64_t grlhip_grad_numel(const grlhip_plan* plan);
/* Offset (in elements) of parameter `index` inside the flat gradient buffer. */
int64_t grlhip_grad_offset(const grlhip_plan* plan, int index);

/* Replaces: the forward + autograd backward of ScoreFunction.score_matching_loss
 * (grl/generative_models/model_functions/score_function.py:83-251), VelocityFunction.flow_matching_loss
 * (velocity_function.py:81-218) and flow_matching_loss_icfm (:220-302).  Every variant is
 *   pred = alpha[r]*net(t[r], x_t[r], c[r]) + beta[r]*x_t[r];  loss_r = sum_d 0.5*(pred - target)^2
 * loss_rows[R] receives loss_r; grad_flat (+=) receives sum_r row_weight[r] * d loss_r / d theta
 * (row_weight NULL = 1).  grad_flat is NOT zeroed by the call.
 * Gradients are bit-reproducible from call to call: the contraction over the rows is split into a fixed number of
 * slices (independent of the device), whose partial sums are added in order (no float atomics).
 * Arithmetic of every GRLHIP_F32 contraction: fp32 operands split exactly into three bf16 pieces, six bf16 MFMAs with
 * fp32 accumulation per product -- fp32 rounding level, on the bf16 matrix cores.
 * QNET plans (the two regressions of QGPOCritic.q_loss, grl/algorithms/qgpo.py:132-139): x_t = action, cond = state,
 * t and beta ignored (may be NULL), pred = alpha[r]*q(action, state), target[R,1]. */
int grlhip_loss_fwd_bwd(const grlhip_plan* plan, const void* packed, int dtype, const float* t,
                        const float* x_t, const float* cond, const float* alpha, const float* beta,
                        const float* target, const float* row_weight, float* loss_rows,
                        float* grad_flat, int64_t rows, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Replaces: torch autograd's backward through IntrinsicModel.forward / QNetwork.forward
 * (grl/generative_models/intrinsic_model.py:54-75, grl/rl_modules/value_network/q_network.py:48-64) for losses this
 * library has no fused kernel for (expectile / weighted regressions of the IQL critics, grl/algorithms/gmpo.py): the
 * vector-Jacobian products grad_flat (+=) sum_r grad_out[r] . d out[r] / d theta (NULL: skipped) and
 * grad_x[R,x_dim] = grad_out[r] . d out[r] / d x[r] (NULL: skipped; for a QNET plan x is the action -- the path of the
 * GMPG policy gradient, grl/algorithms/gmpg.py:517-534), with the forward recomputed inside the call (nothing is kept
 * between grlhip_net_forward and this).  No gradient with respect to cond / t.
 * Workspace: grlhip_loss_workspace_bytes.  t is ignored for QNET plans. */
int grlhip_net_backward(const grlhip_plan* plan, const void* packed, int dtype, const float* t, const float* x,
                        const float* cond, const float* grad_out, float* grad_flat, float* grad_x, int64_t rows,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: torch autograd's backward through the Euler chain of a differentiable sample
 * (`sample(..., with_grad=True)`: diffusion_model/diffusion_model.py:282-304,
 * conditional_flow_model/independent_conditional_flow_model.py:285-319 -> ODESolver.integrate, used by GMPG,
 * grl/algorithms/gmpg.py:333,517).  traj[n_steps+1][R][D] are the states a grlhip_sample_ode / _sde call with the
 * same coef table stored (every step emitting slot k+1 with lerp 1, slot 0 = x0); grad_traj has the same shape
 * (d loss / d traj, zeros where unused).  Walks the chain backwards, recomputing each step's forward:
 * grad_flat (+=) d loss / d theta, grad_x0[R][D] (may be NULL) = d loss / d x0.  Unguided RESNET plans, GRLHIP_F32.
 * Workspace: grlhip_loss_workspace_bytes. */
int grlhip_sample_ode_backward(const grlhip_plan* plan, const void* packed, int dtype, const float* traj,
                               const float* cond, const float* coef, int32_t n_steps, int64_t rows,
                               const float* grad_traj, float* grad_x0, float* grad_flat,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: torch autograd's backward through DPMSolver.integrate's stages when the caller differentiates a
 * DPM-Solver sample (`sample(..., with_grad=True)` with a DPMSolver: diffusion_model/diffusion_model.py:261-279 ->
 * grl/numerical_methods/numerical_solvers/dpm_solver.py:83-905, singlestep orders 1-3).  stage_table is the table a
 * grlhip_sample_stages call ran (same row layout); states[n_stages+1][R][D] holds every stage's INPUT state
 * (states[i] = the state stage i started from, states[n_stages] = the final one: the trajectory of a forward call
 * whose table emits slot i+1 at every stage i); grad_states has the same shape (d loss / d states, zeros where
 * unused).  Walks the table backwards, recomputing each stage's model evaluation and carrying the adjoints of the
 * saved state / model values (xs, ms, ms2): grad_flat (+=) d loss / d theta, grad_x0[R][D] (may be NULL) =
 * d loss / d x0.  The guidance column cg is not differentiated (the reference detaches the guidance term,
 * energy_conditional_diffusion_model.py:87-92).  RESNET plans, GRLHIP_F32.  Workspace: grlhip_loss_workspace_bytes. */
int grlhip_sample_stages_backward(const grlhip_plan* plan, const void* packed, int dtype, const float* states,
                                  const float* cond, const float* stage_table, int32_t n_stages, int64_t rows,
                                  const float* grad_states, float* grad_x0, float* grad_flat,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: the energy-guidance network's forward + autograd backward inside
 * EnergyConditionalDiffusionModel.energy_guidance_loss (contrastive energy prediction,
 * grl/generative_models/diffusion_model/energy_conditional_diffusion_model.py:1354-1394) for a CONCAT_MLP plan.
 * Rows come in groups of `group_size` consecutive support actions of one state (x.reshape(B*K, D) order);
 *   f = net(t, x_t, c);  loss_g = -sum_k softmax(logits0)_gk * log_softmax(f)_gk     (both over the group)
 * with logits0[R] = alpha * energy_model(x_0, c) supplied by the caller (detached in the reference).
 * loss_groups[R/group_size] receives loss_g; grad_flat (+=, may be NULL) receives
 * sum_g group_weight[g] * d loss_g / d theta (group_weight NULL = 1).  Workspace: grlhip_loss_workspace_bytes. */
int grlhip_cep_loss_fwd_bwd(const grlhip_plan* plan, const void* packed, int dtype, const float* t,
                            const float* x_t, const float* cond, const float* logits0, int32_t group_size,
                            const float* group_weight, float* loss_groups, float* grad_flat, int64_t rows,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: torch.optim.Adam.step (amsgrad and maximize off) for the optimizers the reference's training loops build
 * over one network's parameters (grl/algorithms/qgpo.py:554,637,642, grl/algorithms/gmpo.py:788,942,946,1061), when the
 * parameters and their gradients live in flat fp32 buffers (generativerl_preview_amd/optim.py: the gradient layout of
 * grlhip_grad_offset): ONE launch over n elements instead of the optimizer's per-tensor kernels,
 *   g += weight_decay p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g g;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps).
 * `step` counts from 1.  All four buffers are device pointers of n floats; exp_avg / exp_avg_sq start at zero. */
int grlhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);

/* grlhip_adam_step with the step count on the device, for training steps replayed from a captured HIP graph (a by-value
 * `step` would be frozen into the graph): `step_record` points to 16 device bytes {int64 step; float; float}, zeroed by
 * the caller before the first step.  advance != 0: a one-thread kernel first does ++step and derives the two bias
 * corrections from it (in double, as grlhip_adam_step does on the host); advance == 0 reuses them (further parameter
 * ranges of the same optimizer step).  Same update rule; the reference's counterpart is torch.optim.Adam(capturable=True). */
int grlhip_adam_step_counted(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, void* step_record, int advance,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRLHIP_H */
