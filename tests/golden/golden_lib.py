"""Deterministic inputs shared by ``make_golden.py`` (which runs the reference) and the
parity tests (which must not see the reference).

Weights are NOT stored in the fixtures (the canonical nets hold 1.6-3.5 M parameters):
they are regenerated from ``numpy.random.RandomState`` -- a frozen legacy generator whose
stream is stable across numpy versions -- from the (key, shape) lists recorded in
``manifest.json``, which themselves come from the reference's ``state_dict()``.
"""
import zlib

import numpy as np

T_DIM = 32


def t_encoder_cfg(embed_dim=T_DIM, scale=30.0):
    return dict(type="GaussianFourierProjectionTimeEncoder",
                args=dict(embed_dim=embed_dim, scale=scale))


def resnet_cfg(hidden, x_dim, cond_dim=None, cond_hidden=32, t_cond_hidden=128, t_dim=T_DIM):
    args = dict(hidden_sizes=list(hidden), output_dim=x_dim, t_dim=t_dim)
    if cond_dim:
        args.update(condition_dim=cond_dim, condition_hidden_dim=cond_hidden,
                    t_condition_hidden_dim=t_cond_hidden)
    return dict(t_encoder=t_encoder_cfg(t_dim),
                backbone=dict(type="TemporalSpatialResidualNet", args=args))


def energy_cfg(x_dim, cond_dim, hidden=256, t_dim=T_DIM):
    return dict(t_encoder=t_encoder_cfg(t_dim),
                backbone=dict(type="ConcatenateMLP",
                              args=dict(hidden_sizes=[x_dim + cond_dim + t_dim, hidden, hidden],
                                        output_size=1, activation="silu")))


VP = dict(type="linear_vp_sde", beta_0=0.1, beta_1=20.0)
GVP = dict(type="gvp")


def diffusion_cfg(x_dim, path, head, hidden, cond_dim=None, solver=None, energy=False, **net_kw):
    cfg = dict(device="cpu", x_size=x_dim, alpha=1.0, path=dict(path), reverse_path=dict(path),
               model=dict(type=head, args=resnet_cfg(hidden, x_dim, cond_dim, **net_kw)),
               solver=solver or dict(type="ODESolver", args=dict(library="torchdyn")))
    if energy:
        cfg["energy_guidance"] = energy_cfg(x_dim, cond_dim)
    return cfg


def icfm_cfg(x_dim, hidden, cond_dim=None, sigma=0.1, **net_kw):
    return dict(device="cpu", x_size=x_dim, alpha=1.0, path=dict(sigma=sigma),
                model=dict(type="velocity_function", args=resnet_cfg(hidden, x_dim, cond_dim, **net_kw)),
                solver=dict(type="ODESolver", args=dict(library="torchdyn")))


# name -> (kind, config).  kinds: "diffusion", "energy", "icfm".
def model_zoo():
    zoo = {}
    # C2/C4: QGPO HalfCheetah (benchmark/qgpo/vpsde/halfcheetah_medium_expert.py:99-130)
    zoo["qgpo_noise"] = ("energy", diffusion_cfg(6, VP, "noise_function", [512, 256, 128], 17, energy=True))
    # other environments of the same benchmark family hit the same fused kernel class (padded I/O widths):
    # Hopper (benchmark/qgpo/vpsde/hopper_medium.py:4-5,99-130: action 3, observation 11), energy-guided, and a wide
    # observation (antmaze-like: action 8, observation 44 = three condition k-steps), unguided
    zoo["hopper_noise"] = ("energy", diffusion_cfg(3, VP, "noise_function", [512, 256, 128], 11, energy=True))
    zoo["wideobs_noise"] = ("diffusion", diffusion_cfg(8, VP, "noise_function", [512, 256, 128], 44))
    # C1: swiss roll tutorial (unconditional, D=2)
    zoo["swiss_noise"] = ("diffusion", diffusion_cfg(2, VP, "noise_function", [512, 256, 128]))
    # C3: GMPO ICFM
    zoo["icfm_cond"] = ("icfm", icfm_cfg(6, [512, 256, 128], 17))
    # C5: wide single-level GVP velocity net
    zoo["gvp_wide"] = ("diffusion", diffusion_cfg(64, GVP, "velocity_function", [1024]))
    # small nets: every head x path, full-gradient goldens
    for head in ("noise_function", "score_function", "velocity_function", "data_prediction_function"):
        zoo["small_vp_" + head.split("_")[0]] = (
            "diffusion", diffusion_cfg(6, VP, head, [64, 32, 16], 17, cond_hidden=8, t_cond_hidden=24))
    zoo["small_gvp_velocity"] = ("diffusion", diffusion_cfg(6, GVP, "velocity_function", [64, 32, 16], 17,
                                                          cond_hidden=8, t_cond_hidden=24))
    zoo["small_gvp_score"] = ("diffusion", diffusion_cfg(6, GVP, "score_function", [64, 32, 16], 17,
                                                       cond_hidden=8, t_cond_hidden=24))
    zoo["small_uncond_noise"] = ("diffusion", diffusion_cfg(3, VP, "noise_function", [48, 24]))
    zoo["small_icfm"] = ("icfm", icfm_cfg(6, [64, 32, 16], 17, cond_hidden=8, t_cond_hidden=24))
    zoo["small_energy"] = ("energy", diffusion_cfg(6, VP, "noise_function", [64, 32, 16], 17, energy=True,
                                                   cond_hidden=8, t_cond_hidden=24))
    return zoo


def _seed_for(name, salt):
    return (zlib.crc32(name.encode()) + 7919 * salt) % (2 ** 31 - 1)


def make_weights(model_name, key_shapes):
    """key_shapes: list of (state_dict key, shape).  torch-default-like init
    (uniform +-1/sqrt(fan_in)); the output Linear is scaled by 0.1 so that sampled
    trajectories stay O(1); Fourier W = randn * 30 * 2 pi."""
    out = {}
    for i, (key, shape) in enumerate(key_shapes):
        rs = np.random.RandomState(_seed_for(model_name + "/" + key, i))
        shape = tuple(shape)
        if key.endswith("t_encoder.W"):
            w = rs.randn(*shape) * 30.0 * 2.0 * np.pi
        elif len(shape) == 0:
            w = np.asarray(1.0)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:  # bias: fan_in of its weight is not known here; use its own length
                bound = 1.0 / np.sqrt(max(shape[0], 1))
            else:
                bound = 1.0 / np.sqrt(fan_in)
            w = rs.uniform(-bound, bound, size=shape)
            if "last_block" in key:
                w = w * 0.1
        out[key] = np.ascontiguousarray(w, dtype=np.float32)
    return out


# VP-path models whose weights are TRAINED (a few hundred Adam steps of the reference's own loss on a bounded toy set,
# make_trained_weights.py) and stored in weights_<name>.npz: with the seeded init above the VP reverse ODE amplifies
# the state by exp(0.5 * int beta) ~ 150 over [0, 1], so sampled "actions" end at |x| ~ 300 and a max-abs criterion
# on them says nothing.  Trained nets keep |x| = O(1), as a real policy does.
# Not trained: small_vp_score (the reference's score-head loss regresses score * sigma onto +eps, score_function.py
# :206-217 as written, so training drives the reverse ODE to diverge: 2e12 after 100 steps), small_vp_data (no loss
# of the reference accepts a data-prediction head), small_gvp_score (tan(pi/2) at the first step).  Those keep the
# seeded init and their trajectory checks stay RELATIVE to max|ref|.
TRAINED = ("qgpo_noise", "swiss_noise", "small_vp_noise", "small_vp_velocity", "small_energy", "small_uncond_noise",
           "hopper_noise", "wideobs_noise")
# trained weights rounded to fp16 before they are stored and used (halves the fixture; exact in fp32 on both sides)
STORED_F16 = ("hopper_noise", "wideobs_noise")


def toy_data(name, rows, x_dim, cond_dim, seed):
    """Bounded toy training set: x in [-1, 1]^D, a smooth function of the condition (or one bent blob)."""
    rs = np.random.RandomState(_seed_for(name + "/toy", seed))
    if cond_dim:
        A = np.random.RandomState(_seed_for(name + "/toyA", 0)).randn(cond_dim, x_dim) / np.sqrt(cond_dim)
        cond = rs.randn(rows, cond_dim)
        x = 0.9 * np.tanh(1.5 * cond @ A) + 0.05 * rs.randn(rows, x_dim)
        return x.astype(np.float32), cond.astype(np.float32)
    # unconditional: ONE smooth, bent ("banana") blob.  A multi-modal target makes the sampling ODE bifurcate between
    # modes, where rounding-level differences grow without bound -- that tests chaos, not kernels.
    z = rs.randn(rows, x_dim)
    x = 0.3 * z
    x[:, 0] = 0.5 * z[:, 0]
    if x_dim >= 2:
        x[:, 1] = 0.25 * z[:, 1] + 0.2 * (z[:, 0] ** 2 - 1.0)
    return np.clip(x, -1.0, 1.0).astype(np.float32), None


def load_weights(model_name, key_shapes, golden_dir=None):
    """The weights the goldens were produced with: the stored trained ones for TRAINED models, else the seeded init."""
    import os
    golden_dir = golden_dir or os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(golden_dir, "weights_%s.npz" % model_name)
    if model_name in TRAINED and os.path.exists(path):
        z = np.load(path)
        out = {}
        for key, shape in key_shapes:
            w = z[key]
            assert tuple(w.shape) == tuple(shape), (key, w.shape, shape)
            out[key] = np.ascontiguousarray(w, dtype=np.float32)
        return out
    return make_weights(model_name, key_shapes)


def make_inputs(tag, rows, x_dim, cond_dim=None):
    rs = np.random.RandomState(_seed_for(tag, 1))
    d = dict(x=rs.randn(rows, x_dim).astype(np.float32),
             t=rs.uniform(0.02, 0.98, size=rows).astype(np.float32),
             eps=rs.randn(rows, x_dim).astype(np.float32),
             eps2=rs.randn(rows, x_dim).astype(np.float32))
    if cond_dim:
        d["cond"] = rs.randn(rows, cond_dim).astype(np.float32)
    return d
