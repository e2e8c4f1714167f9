// Plan builder: turns a grlhip_net_desc into the parameter list, the op program and the layouts.
// Reference structure being described: TemporalSpatialResidualNet
// (grl/neural_network/__init__.py:245-342) and ConcatenateMLP (:558-583).
#include "plan.hpp"

#include <algorithm>

namespace grlhip {
namespace {

inline int32_t up(int32_t v, int32_t m) { return (v + m - 1) / m * m; }

struct Builder {
  Plan* p;
  int32_t bump = 0;
  int64_t woff = 0;

  int32_t alloc(int32_t features) {  // even-sized so a K-pad feature stays inside the buffer (and zero)
    int32_t o = bump;
    bump += up(features, 2);
    return o;
  }
  int add_param(const std::string& name, int rows, int cols) {
    ParamInfo pi{name, rows, cols, p->grad_numel};
    p->grad_numel += (int64_t)rows * (cols ? cols : 1);
    p->params.push_back(pi);
    return (int)p->params.size() - 1;
  }
  // registers weight+bias parameters "<name>.weight"/"<name>.bias" and the packed matrix
  int add_linear_params(const std::string& name, int N, int K, bool bias = true) {
    int pw = add_param(name + ".weight", N, K);
    int pb = bias ? add_param(name + ".bias", N, 0) : -1;
    return add_matrix(N, K, pw, pb, 0, K, 0);
  }
  int add_matrix(int N, int K, int pw, int pb, int col0, int src_cols, int transposed) {
    LinearDesc l{};
    l.N = N; l.K = K; l.Npad = up(N, 32); l.Kpad = up(K, 2);
    l.p_w = pw; l.p_b = pb; l.col0 = col0; l.src_cols = src_cols; l.transposed = transposed;
    l.w_off = woff; woff += (int64_t)l.Npad * l.Kpad;
    l.b_off = woff; woff += l.Npad;
    p->linears.push_back(l);
    return (int)p->linears.size() - 1;
  }
  void linear(std::vector<Op>& prog, int lin, int in, int out, int add, int flags, int pre = -1) {
    const LinearDesc& l = p->linears[lin];
    Op o{};
    o.type = OP_LINEAR; o.K = l.Kpad; o.N = l.N; o.w = (int32_t)l.w_off; o.b = (int32_t)l.b_off;
    o.in = in; o.out = out; o.add = add; o.flags = flags; o.pre = pre; o.lin = lin;
    prog.push_back(o);
  }
  void eltwise(std::vector<Op>& prog, int type, int n, int in, int out, int add = -1, int w = 0) {
    Op o{};
    o.type = type; o.K = 0; o.N = n; o.w = w; o.b = 0; o.in = in; o.out = out; o.add = add; o.lin = -1; o.pre = -1;
    prog.push_back(o);
  }
};

}  // namespace

Plan* build_plan(const grlhip_net_desc& d, std::string& err) {
  if (d.n_hidden < 1 || d.n_hidden > GRLHIP_MAX_HIDDEN) { err = "n_hidden must be in [1, 8]"; return nullptr; }
  if (d.x_dim < 1 || d.out_dim < 1 || d.t_dim < 2 || (d.t_dim & 1)) { err = "x_dim/out_dim >= 1 and even t_dim >= 2 required"; return nullptr; }
  if (d.cond_dim < 0) { err = "cond_dim < 0"; return nullptr; }
  for (int i = 0; i < d.n_hidden; ++i)
    if (d.hidden[i] < 1) { err = "hidden sizes must be positive"; return nullptr; }

  Plan* p = new Plan();
  p->desc = d;
  Builder b{p};
  b.add_param("t_encoder.W", d.t_dim / 2, 0);

  if (d.kind == GRLHIP_NET_RESNET) {
    const int L = d.n_hidden;
    const int* h = d.hidden;
    for (int i = 0; i + 1 < L; ++i)
      if (h[i] != 2 * h[i + 1]) {  // up_block[i] is fed cat(u, d): needs hidden[i] == 2*hidden[i+1]
        err = "TemporalSpatialResidualNet wiring needs hidden[i] == 2*hidden[i+1] (grl/neural_network/__init__.py:304-311,340-341)";
        delete p; return nullptr;
      }
    const bool cond = d.cond_dim > 0;
    const int ch = cond ? d.cond_hidden : 0;
    const int tcd = d.t_dim + ch;
    const int tch = cond ? (d.t_cond_hidden > 0 ? d.t_cond_hidden : tcd) : d.t_dim;
    if (cond && ch < 1) { err = "condition_hidden_dim must be >= 1 for a conditional net"; delete p; return nullptr; }

    // ---- slab: [t_emb | c_emb] contiguous so sort_t.0 reads the concatenation in place
    p->temb_off = b.alloc(d.t_dim + ch);
    const int cemb = p->temb_off + d.t_dim;
    p->x_off = b.alloc(d.x_dim);
    p->c_off = cond ? b.alloc(d.cond_dim) : -1;
    const int z = b.alloc(tch), te = b.alloc(tch), ste = b.alloc(tch);
    std::vector<int> cat(L);
    for (int i = 0; i < L; ++i) cat[i] = b.alloc(2 * h[i]);  // [u_i | d_i]
    const int hmax = *std::max_element(h, h + L);
    const int sA = b.alloc(hmax), sB = b.alloc(hmax);
    p->out_off = b.alloc(d.out_dim);

    auto& F = p->fwd;
    if (cond) {
      int l = b.add_linear_params("backbone.pre_sort_condition.0", ch, d.cond_dim);
      b.linear(F, l, p->c_off, cemb, -1, F_SILU_OUT);
    }
    int l0 = b.add_linear_params("backbone.sort_t.0", tch, tcd);
    int l2 = b.add_linear_params("backbone.sort_t.2", tch, tch);
    b.linear(F, l0, p->temb_off, z, -1, F_SILU_OUT);
    b.linear(F, l2, z, te, -1, 0);
    b.eltwise(F, OP_SILU, tch, te, ste);  // every time_mlp starts with SiLU(te)

    auto block = [&](const std::string& name, int in, int K, int out, int N) {
      int lt = b.add_linear_params(name + ".time_mlp.1", N, tch);
      int l1 = b.add_linear_params(name + ".dense1.0", N, K);
      int ld = b.add_linear_params(name + ".dense2.0", N, N);
      int lm = (K != N) ? b.add_linear_params(name + ".modify_x", N, K) : -1;
      b.linear(F, lt, ste, sA, -1, 0);
      b.linear(F, l1, in, sB, sA, F_SILU_OUT);           // h1 = silu(dense1 x) + time_mlp(te)
      int res = in;
      if (lm >= 0) { b.linear(F, lm, in, sA, -1, 0); res = sA; }
      b.linear(F, ld, sB, out, res, F_SILU_OUT);         // out = silu(dense2 h1) + modify_x(x)
    };
    block("backbone.first_block", p->x_off, d.x_dim, cat[0] + h[0], h[0]);
    for (int i = 0; i + 1 < L; ++i)
      block("backbone.down_block." + std::to_string(i), cat[i] + h[i], h[i], cat[i + 1] + h[i + 1], h[i + 1]);
    block("backbone.middle_block", cat[L - 1] + h[L - 1], h[L - 1], cat[L - 1], h[L - 1]);
    for (int j = 0; j + 1 < L; ++j) {
      int i = L - 2 - j;
      block("backbone.up_block." + std::to_string(j), cat[i + 1], h[i], cat[i], h[i]);
    }
    int ll = b.add_linear_params("backbone.last_block", d.out_dim, 2 * h[0]);
    b.linear(F, ll, cat[0], p->out_off, -1, 0);
  } else if (d.kind == GRLHIP_NET_CONCAT_MLP) {
    const int L = d.n_hidden;
    const int K0 = d.t_dim + d.x_dim + d.cond_dim;
    if (d.out_dim != 1) { err = "CONCAT_MLP energy nets must have out_dim == 1"; delete p; return nullptr; }
    p->temb_off = b.alloc(K0);  // [t_emb | x | c] contiguous (concat order intrinsic_model.py:69)
    p->x_off = p->temb_off + d.t_dim;
    p->c_off = d.cond_dim > 0 ? p->x_off + d.x_dim : -1;
    std::vector<int> zb(L), ab(L), dz(L);
    for (int i = 0; i < L; ++i) { zb[i] = b.alloc(d.hidden[i]); ab[i] = b.alloc(d.hidden[i]); dz[i] = b.alloc(d.hidden[i]); }
    p->out_off = b.alloc(1);
    p->gradx_off = b.alloc(d.x_dim);
    std::vector<int> lin(L + 1);
    int kin = K0;
    for (int i = 0; i < L; ++i) {
      lin[i] = b.add_linear_params("backbone.model.model.linear" + std::to_string(i), d.hidden[i], kin);
      kin = d.hidden[i];
    }
    lin[L] = b.add_linear_params("backbone.model.model.linear" + std::to_string(L), 1, kin);
    // forward (energy value)
    auto& F = p->fwd;
    int in = p->temb_off;
    for (int i = 0; i < L; ++i) { b.linear(F, lin[i], in, ab[i], -1, F_SILU_OUT); in = ab[i]; }
    b.linear(F, lin[L], in, p->out_off, -1, 0);
    // gradient program: pre-activations, then dE/dx
    auto& G = p->grad;
    in = p->temb_off;
    for (int i = 0; i < L; ++i) {
      b.linear(G, lin[i], in, zb[i], -1, 0);
      if (i + 1 < L) { b.eltwise(G, OP_SILU, d.hidden[i], zb[i], ab[i]); in = ab[i]; }
    }
    // dz_L = w_last * silu'(z_L): w_last is row 0 of the packed last linear, i.e. its column n=0
    // of W^T -- stored strided, so pack a contiguous copy as an extra "matrix" [h_L x 1] transposed
    const LinearDesc last = p->linears[lin[L]];
    int lw = b.add_matrix(d.hidden[L - 1], 1, last.p_w, -1, 0, d.hidden[L - 1], 1);  // M[n][0] = W[0][n]
    // that matrix is stored as W^T[k=0][n] = M[n][0] -> contiguous over n at w_off
    b.eltwise(G, OP_SILU_BWD_CONST, d.hidden[L - 1], zb[L - 1], dz[L - 1], -1, (int32_t)p->linears[lw].w_off);
    for (int i = L - 1; i >= 1; --i) {
      const LinearDesc li = p->linears[lin[i]];
      int lt = b.add_matrix(li.K, li.N, li.p_w, -1, 0, li.K, 1);  // W_i^T : [K_i x N_i]
      b.linear(G, lt, dz[i], ab[i - 1], -1, 0);                    // da_{i-1} (reuses the a buffer)
      b.eltwise(G, OP_SILU_BWD, d.hidden[i - 1], zb[i - 1], dz[i - 1], ab[i - 1]);
    }
    const LinearDesc l1 = p->linears[lin[0]];
    int lx = b.add_matrix(d.x_dim, l1.N, l1.p_w, -1, d.t_dim, l1.K, 1);  // (W_1[:, x cols])^T
    b.linear(G, lx, dz[0], p->gradx_off, -1, 0);
  } else {
    err = "unknown net kind";
    delete p; return nullptr;
  }
  p->fourier_off = b.woff;
  b.woff += up(d.t_dim / 2, 2);
  p->packed_floats = b.woff;
  p->slab_features = up(b.bump, 2);
  if (p->packed_floats >= (int64_t)1 << 31) { err = "network too large"; delete p; return nullptr; }
  return p;
}

}  // namespace grlhip
