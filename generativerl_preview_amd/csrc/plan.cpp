// Plan builder: turns a grlhip_net_desc into the parameter list, the op program and the layouts.
// Reference structure being described: TemporalSpatialResidualNet
// (grl/neural_network/__init__.py:245-342) and ConcatenateMLP (:558-583).
#include "plan.hpp"

#include <algorithm>
#include <map>

namespace grlhip {
namespace {

inline int32_t up(int32_t v, int32_t m) { return (v + m - 1) / m * m; }

struct Builder {
  Plan* p;
  int32_t bump = 0;
  int64_t woff = 0;
  std::map<std::string, int> by_name;  // linear layer name -> index into p->linears

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
    int l = add_matrix(N, K, pw, pb, 0, K, 0);
    by_name[name] = l;
    return l;
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
  void linear(std::vector<Op>& prog, int lin, int in, int out, int add, int flags, int pre = -1, int z = -1, int zN = 0) {
    const LinearDesc& l = p->linears[lin];
    Op o{};
    o.type = OP_LINEAR; o.K = l.Kpad; o.N = l.N; o.w = (int32_t)l.w_off; o.b = (int32_t)l.b_off;
    o.in = in; o.out = out; o.add = add; o.flags = flags; o.pre = pre; o.lin = lin;
    o.z = z; o.zN = (flags & F_DSILU) ? (zN > 0 ? zN : l.N) : 0;
    prog.push_back(o);
  }
  void eltwise(std::vector<Op>& prog, int type, int n, int in, int out, int add = -1, int w = 0) {
    Op o{};
    o.type = type; o.K = 0; o.N = n; o.w = w; o.b = 0; o.in = in; o.out = out; o.add = add; o.lin = -1; o.pre = -1; o.z = -1;
    prog.push_back(o);
  }
};

}  // namespace

Plan* build_plan(const grlhip_net_desc& d, std::string& err) {
  if (d.n_hidden < 1 || d.n_hidden > GRLHIP_MAX_HIDDEN) { err = "n_hidden must be in [1, 8]"; return nullptr; }
  const bool qnet = d.kind == GRLHIP_NET_QNET;
  if (qnet && d.t_dim != 0) { err = "a QNET plan has no time input: t_dim must be 0"; return nullptr; }
  if (d.x_dim < 1 || d.out_dim < 1 || (!qnet && (d.t_dim < 2 || (d.t_dim & 1)))) { err = "x_dim/out_dim >= 1 and even t_dim >= 2 required"; return nullptr; }
  if (d.cond_dim < 0) { err = "cond_dim < 0"; return nullptr; }
  for (int i = 0; i < d.n_hidden; ++i)
    if (d.hidden[i] < 1) { err = "hidden sizes must be positive"; return nullptr; }

  Plan* p = new Plan();
  p->desc = d;
  Builder b{p};
  if (!qnet) b.add_param("t_encoder.W", d.t_dim / 2, 0);

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

    // ------------------------------------------------------------------ training step programs
    // Own slab layout (every pre-activation kept, gradient buffers appended); weights W^T are packed
    // as extra matrices so that dX = W^T dZ is the same LINEAR op.
    {
      Builder t{p};
      t.woff = b.woff;
      auto& TF = p->train_fwd;
      auto& TB = p->train_bwd;
      auto T_of = [&](int lin) {  // transposed copy of a packed linear
        const LinearDesc l = p->linears[lin];
        return t.add_matrix(l.K, l.N, l.p_w, -1, 0, l.K, 1);
      };
      auto dw = [&](int lin, int dz, int in) {  // weight + bias gradient of a linear
        const LinearDesc l = p->linears[lin];
        Op o{};
        o.type = OP_DW; o.K = l.K; o.N = l.N; o.w = (int32_t)p->params[l.p_w].grad_off;
        o.b = l.p_b >= 0 ? (int32_t)p->params[l.p_b].grad_off : -1;
        o.in = in; o.add = dz; o.out = -1; o.pre = l.K; o.lin = lin;
        TB.push_back(o);
      };
      auto lin_t = [&](int lin, int in, int out, bool accum, int more = 0) { t.linear(TB, T_of(lin), in, out, -1, (accum ? F_ACCUM : 0) | more); };
      const int L_ = L;
      p->tr_temb = t.alloc(d.t_dim + ch);
      const int tc = p->tr_temb, cemb_t = tc + d.t_dim;
      p->tr_x = t.alloc(d.x_dim);
      p->tr_c = cond ? t.alloc(d.cond_dim) : -1;
      const int zc = cond ? t.alloc(ch) : -1;
      const int z0 = t.alloc(tch), a0 = t.alloc(tch), te_t = t.alloc(tch), ste_t = t.alloc(tch);
      std::vector<int> catf(L_), gcat(L_);
      for (int i = 0; i < L_; ++i) catf[i] = t.alloc(2 * h[i]);
      p->tr_out = t.alloc(d.out_dim);
      // gradient buffers
      p->tr_gout = t.alloc(d.out_dim);
      for (int i = 0; i < L_; ++i) gcat[i] = t.alloc(2 * h[i]);
      const int g_ste = t.alloc(tch), g_te = t.alloc(tch), g_a0 = t.alloc(tch), g_z0 = t.alloc(tch);
      p->tr_gtc = t.alloc(d.t_dim + ch);
      const int g_zc = cond ? t.alloc(ch) : -1;
      const int hmax_t = *std::max_element(h, h + L_);
      const int g_h1 = t.alloc(hmax_t), g_z1 = t.alloc(hmax_t), sM = t.alloc(hmax_t);
      // dz2 of consecutive blocks alternates between two buffers: a block's last GEMM writes the NEXT block's dz2 while this
      // block's weight gradients (queued by the engine until an operand is about to be overwritten, then ONE launch) still
      // read its own
      const int g_zs[2] = {t.alloc(hmax_t), t.alloc(hmax_t)};
      int zsel = 0;

      struct Blk { int lt, l1, ld, lm, in, K, out, N, T, z1, h1, z2; };
      std::vector<Blk> blks;
      auto fblock = [&](const std::string& name, int in, int K, int out, int N) {
        Blk k{};
        k.lt = b.by_name[name + ".time_mlp.1"]; k.l1 = b.by_name[name + ".dense1.0"]; k.ld = b.by_name[name + ".dense2.0"];
        k.lm = (K != N) ? b.by_name[name + ".modify_x"] : -1;
        k.in = in; k.K = K; k.out = out; k.N = N;
        k.T = t.alloc(N); k.z1 = t.alloc(N); k.h1 = t.alloc(N); k.z2 = t.alloc(N);
        t.linear(TF, k.lt, ste_t, k.T, -1, 0);
        t.linear(TF, k.l1, in, k.h1, k.T, F_SILU_OUT | F_SAVE_PRE, k.z1);
        int res = in;
        if (k.lm >= 0) { t.linear(TF, k.lm, in, sM, -1, 0); res = sM; }
        t.linear(TF, k.ld, k.h1, out, res, F_SILU_OUT | F_SAVE_PRE, k.z2);
        blks.push_back(k);
      };
      if (cond) t.linear(TF, b.by_name["backbone.pre_sort_condition.0"], p->tr_c, cemb_t, -1, F_SILU_OUT | F_SAVE_PRE, zc);
      t.linear(TF, b.by_name["backbone.sort_t.0"], tc, a0, -1, F_SILU_OUT | F_SAVE_PRE, z0);
      t.linear(TF, b.by_name["backbone.sort_t.2"], a0, te_t, -1, 0);
      t.eltwise(TF, OP_SILU, tch, te_t, ste_t);
      fblock("backbone.first_block", p->tr_x, d.x_dim, catf[0] + h[0], h[0]);
      for (int i = 0; i + 1 < L_; ++i)
        fblock("backbone.down_block." + std::to_string(i), catf[i] + h[i], h[i], catf[i + 1] + h[i + 1], h[i + 1]);
      fblock("backbone.middle_block", catf[L_ - 1] + h[L_ - 1], h[L_ - 1], catf[L_ - 1], h[L_ - 1]);
      for (int j = 0; j + 1 < L_; ++j) {
        int i = L_ - 2 - j;
        fblock("backbone.up_block." + std::to_string(j), catf[i + 1], h[i], catf[i], h[i]);
      }
      t.linear(TF, b.by_name["backbone.last_block"], catf[0], p->tr_out, -1, 0);

      // ---- backward.  gin_mode: 0 = input gradient not needed, 1 = write, 2 = accumulate
      bool ste_written = false;
      // next: the block whose backward follows (its output gradient IS this block's gin): its dz2 = dout SiLU'(z2) comes
      // out of the epilogue of this block's last GEMM (F_DSILU over the leading next->N features), so it starts with
      // dz2 already in g_z (dz2_ready) instead of an elementwise pass over dout
      auto bblock = [&](const Blk& k, int gout, int gin, int gin_mode, const Blk* next, bool dz2_ready) {
        const int g_z = g_zs[zsel], g_z_next = g_zs[zsel ^ 1];
        zsel ^= 1;
        if (k.lm >= 0) {
          dw(k.lm, gout, k.in);
          // (reads dout, writes d in: independent of the dense2 input gradient that follows)
          if (gin_mode) lin_t(k.lm, gout, gin, gin_mode == 2, F_PAIR_NEXT);
        }
        // (identity residual, k.lm < 0: d in += d out rides as the residual operand of the dense1 GEMM below)
        if (!dz2_ready) t.eltwise(TB, OP_SILU_BWD, k.N, k.z2, g_z, gout);       // dz2 = dout * silu'(z2)
        dw(k.ld, g_z, k.h1);
        // dh1 (= dT) and, from the same epilogue, dz1 = dh1 * silu'(z1) (g_z is this GEMM's input: dz1 -> g_z1)
        t.linear(TB, T_of(k.ld), g_z, g_h1, -1, F_DSILU, g_z1, k.z1);
        dw(k.lt, g_h1, ste_t);
        // (reads dh1, adds into d ste: independent of the dense1 input gradient that follows)
        lin_t(k.lt, g_h1, g_ste, ste_written, gin_mode ? F_PAIR_NEXT : 0);
        ste_written = true;
        dw(k.l1, g_z1, k.in);
        if (gin_mode) {
          // gin (+)= W1^T dz1 (+ dout for an identity residual); g_z (this block's dz2) is dead by now
          const int flags = ((k.lm >= 0 || gin_mode == 2) ? F_ACCUM : 0) | (next ? F_DSILU : 0);
          t.linear(TB, T_of(k.l1), g_z1, gin, k.lm >= 0 ? -1 : gout, flags, next ? g_z_next : -1, next ? next->z2 : -1, next ? next->N : 0);
        }
      };
      const int llast = b.by_name["backbone.last_block"];
      dw(llast, p->tr_gout, catf[0]);
      // d cat[0] = W_last^T dout; the leading h[0] features are the output gradient of the block processed first, whose
      // dz2 = dout SiLU'(z2) comes out of the same epilogue (as between blocks, see bblock)
      t.linear(TB, T_of(llast), p->tr_gout, gcat[0], -1, F_DSILU, g_zs[0], blks.back().z2, blks.back().N);
      int bi = (int)blks.size() - 1;                              // blocks were pushed in forward order
      bool ready = true;
      for (int j = L_ - 2; j >= 0; --j, --bi) {                    // up blocks, reversed
        int i = L_ - 2 - j;
        bblock(blks[bi], gcat[i], gcat[i + 1], 1, &blks[bi - 1], ready);
        ready = true;
      }
      bblock(blks[bi], gcat[L_ - 1], gcat[L_ - 1] + h[L_ - 1], 2, &blks[bi - 1], ready); --bi;   // middle
      ready = true;
      for (int i = L_ - 2; i >= 0; --i, --bi)                      // down blocks, reversed
        bblock(blks[bi], gcat[i + 1] + h[i + 1], gcat[i] + h[i], 2, &blks[bi - 1], true);
      bblock(blks[bi], gcat[0] + h[0], -1, 0, nullptr, true);      // first block (no dx)
      t.eltwise(TB, OP_SILU_BWD, tch, te_t, g_te, g_ste);          // ste = silu(te)
      dw(b.by_name["backbone.sort_t.2"], g_te, a0);
      lin_t(b.by_name["backbone.sort_t.2"], g_te, g_a0, false);
      t.eltwise(TB, OP_SILU_BWD, tch, z0, g_z0, g_a0);
      dw(b.by_name["backbone.sort_t.0"], g_z0, tc);
      lin_t(b.by_name["backbone.sort_t.0"], g_z0, p->tr_gtc, false);   // d[t_emb | c_emb]
      if (cond) {
        t.eltwise(TB, OP_SILU_BWD, ch, zc, g_zc, p->tr_gtc + d.t_dim);
        dw(b.by_name["backbone.pre_sort_condition.0"], g_zc, p->tr_c);
      }
      // ---- input gradient (differentiable sampling only; skipped by the loss entry points): the first block's
      // output gradient is still in gcat[0] + h[0] and its dz1 in g_z1 (nothing after that block writes them)
      {
        const Blk& k = blks[0];
        p->tr_gx = t.alloc(d.x_dim);
        const size_t before = TB.size();
        if (k.lm >= 0) lin_t(k.lm, gcat[0] + h[0], p->tr_gx, false);
        else t.eltwise(TB, OP_COPY, k.N, gcat[0] + h[0], p->tr_gx);
        lin_t(k.l1, g_z1, p->tr_gx, true);
        p->train_bwd_dx_ops = (int32_t)(TB.size() - before);
      }
      p->train_slab_features = up(t.bump, 2);
      b.woff = t.woff;
    }
  } else if (d.kind == GRLHIP_NET_CONCAT_MLP || qnet) {
    const int L = d.n_hidden;
    const int ACT = qnet ? F_RELU_OUT : F_SILU_OUT;
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
    for (int i = 0; i < L; ++i) { b.linear(F, lin[i], in, ab[i], -1, ACT); in = ab[i]; }
    b.linear(F, lin[L], in, p->out_off, -1, 0);
    // gradient program (energy nets only): pre-activations, then dE/dx
    auto& G = p->grad;
    in = p->temb_off;
    for (int i = 0; i < L && !qnet; ++i) {
      b.linear(G, lin[i], in, zb[i], -1, 0);
      if (i + 1 < L) { b.eltwise(G, OP_SILU, d.hidden[i], zb[i], ab[i]); in = ab[i]; }
    }
    // dz_L = w_last * silu'(z_L): w_last is row 0 of the packed last linear, i.e. its column n=0
    // of W^T -- stored strided, so pack a contiguous copy as an extra "matrix" [h_L x 1] transposed
    const LinearDesc last = p->linears[lin[L]];
    int lw = b.add_matrix(d.hidden[L - 1], 1, last.p_w, -1, 0, d.hidden[L - 1], 1);  // M[n][0] = W[0][n]
    // that matrix is stored as W^T[k=0][n] = M[n][0] -> contiguous over n at w_off
    if (!qnet) b.eltwise(G, OP_SILU_BWD_CONST, d.hidden[L - 1], zb[L - 1], dz[L - 1], -1, (int32_t)p->linears[lw].w_off);
    for (int i = L - 1; i >= 1 && !qnet; --i) {
      const LinearDesc li = p->linears[lin[i]];
      int lt = b.add_matrix(li.K, li.N, li.p_w, -1, 0, li.K, 1);  // W_i^T : [K_i x N_i]
      b.linear(G, lt, dz[i], ab[i - 1], -1, 0);                    // da_{i-1} (reuses the a buffer)
      b.eltwise(G, OP_SILU_BWD, d.hidden[i - 1], zb[i - 1], dz[i - 1], ab[i - 1]);
    }
    const LinearDesc l1 = p->linears[lin[0]];
    if (!qnet) {
      int lx = b.add_matrix(d.x_dim, l1.N, l1.p_w, -1, d.t_dim, l1.K, 1);  // (W_1[:, x cols])^T
      b.linear(G, lx, dz[0], p->gradx_off, -1, 0);
    }

    // ---- training programs (CEP energy_guidance_loss, energy_conditional_diffusion_model.py:1308-1394):
    // forward keeping every pre-activation; backward from d loss / d f (tr_gout, one feature) to every
    // weight / bias gradient and to the Fourier features (tr_gtc)
    {
      Builder t{p};
      t.woff = b.woff;
      auto& TF = p->train_fwd;
      auto& TB = p->train_bwd;
      p->tr_temb = t.alloc(K0);
      p->tr_x = p->tr_temb + d.t_dim;
      p->tr_c = d.cond_dim > 0 ? p->tr_x + d.x_dim : -1;
      std::vector<int> tz(L), ta(L);
      for (int i = 0; i < L; ++i) { tz[i] = t.alloc(d.hidden[i]); ta[i] = t.alloc(d.hidden[i]); }
      p->tr_out = t.alloc(1);
      p->tr_gout = t.alloc(1);
      const int hmax = *std::max_element(d.hidden, d.hidden + L);
      const int gz[2] = {t.alloc(hmax), t.alloc(hmax)};               // dZ of consecutive layers alternate between the two
      p->tr_gtc = qnet ? -1 : t.alloc(d.t_dim);
      int tin = p->tr_temb;
      for (int i = 0; i < L; ++i) { t.linear(TF, lin[i], tin, ta[i], -1, ACT | F_SAVE_PRE, tz[i]); tin = ta[i]; }
      {                                                                // f = w_last . a_L + b: matrix-vector
        Op o{};
        o.type = OP_HEAD_FWD; o.K = d.hidden[L - 1]; o.N = 1; o.w = (int32_t)p->linears[lw].w_off;
        o.b = (int32_t)p->linears[lin[L]].b_off; o.in = tin; o.out = p->tr_out; o.add = -1; o.pre = -1; o.lin = lin[L];
        TF.push_back(o);
      }
      auto dw = [&](int li, int dzo, int ino) {
        const LinearDesc l = p->linears[li];
        Op o{};
        o.type = OP_DW; o.K = l.K; o.N = l.N; o.w = (int32_t)p->params[l.p_w].grad_off;
        o.b = l.p_b >= 0 ? (int32_t)p->params[l.p_b].grad_off : -1;
        o.in = ino; o.add = dzo; o.out = -1; o.pre = l.K; o.lin = li;
        TB.push_back(o);
      };
      int cur = 0;
      {                                                                // dz_L, d w_last, d b_last in one pass
        Op o{};
        o.type = OP_HEAD_BWD; o.K = d.hidden[L - 1]; o.N = 1; o.w = (int32_t)p->linears[lw].w_off; o.b = -1;
        o.in = ta[L - 1]; o.out = gz[cur]; o.add = p->tr_gout; o.pre = tz[L - 1]; o.flags = qnet ? F_RELU_OUT : 0;
        o.lin = lin[L];
        TB.push_back(o);
      }
      for (int i = L - 1; i >= 0; --i) {
        dw(lin[i], gz[cur], i ? ta[i - 1] : p->tr_temb);
        const LinearDesc li = p->linears[lin[i]];
        if (i) {
          // dz_{i-1} = (W_i^T dz_i) * act'(z_{i-1}) from the GEMM's epilogue; da_{i-1} itself is not stored
          int lt = t.add_matrix(li.K, li.N, li.p_w, -1, 0, li.K, 1);  // W_i^T
          t.linear(TB, lt, gz[cur], -1, -1, F_DSILU | F_NO_OUT | (qnet ? F_RELU_OUT : 0), gz[cur ^ 1], tz[i - 1]);
          cur ^= 1;
        } else if (!qnet) {
          int lt = t.add_matrix(d.t_dim, li.N, li.p_w, -1, 0, li.K, 1);  // (W_1[:, t_emb cols])^T
          t.linear(TB, lt, gz[cur], p->tr_gtc, -1, F_FOURIER_ONLY);
        }
      }
      // trailing op, run only when the caller asks for d/dx (grlhip_net_backward's grad_x): gz[cur] still holds dz_0
      {
        const LinearDesc l0 = p->linears[lin[0]];
        p->tr_gx = t.alloc(d.x_dim);
        int lx0 = t.add_matrix(d.x_dim, l0.N, l0.p_w, -1, d.t_dim, l0.K, 1);  // (W_1[:, x cols])^T
        t.linear(TB, lx0, gz[cur], p->tr_gx, -1, 0);
        p->train_bwd_dx_ops = 1;
      }
      p->train_slab_features = up(t.bump, 2);
      b.woff = t.woff;
    }
  } else {
    err = "unknown net kind";
    delete p; return nullptr;
  }
  p->fourier_off = b.woff;
  b.woff += up(d.t_dim / 2, 2);
  p->packed_floats = b.woff;
  p->slab_features = up(b.bump, 2);
  if (p->packed_floats >= (int64_t)1 << 31) { err = "network too large"; delete p; return nullptr; }
  for (const LinearDesc& l : p->linears) {
    p->split_off.push_back(p->split_elems);
    p->split_elems += (int64_t)up(l.Kpad, 16) / 16 * 3 * up(l.Npad, 128) * 16;
  }
  return p;
}

}  // namespace grlhip
