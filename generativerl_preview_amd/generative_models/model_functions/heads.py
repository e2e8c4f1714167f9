"""Head conversions between the four parameterisations (score / noise / velocity / data prediction) of a
Gaussian-path model: ``ScoreFunction.forward`` (grl/generative_models/model_functions/score_function.py:44-81),
``NoiseFunction.forward`` (noise_function.py:62-78), ``VelocityFunction.forward`` (velocity_function.py:54-79)
and ``DataPredictionFunction.forward`` (data_prediction_function.py:62-84).

Each is ``alpha(t) * net(t, x, c) + beta(t) * x`` with scalar schedule coefficients; the network evaluation is
the HIP forward (``IntrinsicModel.__call__`` -> ``grlhip_net_forward``), the affine part stays in torch.
Formulas are kept AS WRITTEN in the reference (e.g. the score -> data-prediction branch uses ``-std * x``)."""
import torch


def _b(v, x):
    """DiffusionProcess broadcast (diffusion_process.py:55-57): (B,) -> x.shape."""
    v = v.to(x.device)
    return v[(...,) + (None,) * (x.dim() - v.dim())].expand(x.shape) if x.dim() > v.dim() else v


def score(model_type, path, net, t, x):
    if model_type == "noise_function":
        return -net / _b(path.std(t), x)
    if model_type == "score_function":
        return net
    if model_type == "velocity_function":
        return -(net - _b(path.drift_coefficient(t), x) * x) * 2.0 / _b(path.diffusion_squared(t), x)
    if model_type == "data_prediction_function":
        return -(x - _b(path.scale(t), x) * net) / _b(path.covariance(t), x)
    raise NotImplementedError("Unknown type of function {}".format(model_type))


def noise(model_type, path, net, t, x):
    if model_type == "noise_function":
        return net
    if model_type == "score_function":
        return -net * _b(path.std(t), x)
    if model_type == "velocity_function":
        return (net - _b(path.drift_coefficient(t), x) * x) * 2.0 * _b(path.std(t), x) / _b(path.diffusion_squared(t), x)
    if model_type == "data_prediction_function":
        return (x - _b(path.scale(t), x) * net) / _b(path.std(t), x)
    raise NotImplementedError("Unknown type of function {}".format(model_type))


def velocity(model_type, path, net, t, x):
    f, g2 = _b(path.drift_coefficient(t), x), _b(path.diffusion_squared(t), x)
    if model_type == "noise_function":
        return f * x + 0.5 * g2 / _b(path.std(t), x) * net
    if model_type == "score_function":
        return f * x - 0.5 * g2 * net
    if model_type == "velocity_function":
        return net
    # the reference's data_prediction branch fails on its own (shape bug, velocity_function.py:70-77)
    raise NotImplementedError("velocity from {} is not usable in the reference either".format(model_type))


def data_prediction(model_type, path, net, t, x):
    s, sig = _b(path.scale(t), x), _b(path.std(t), x)
    if model_type == "noise_function":
        return (x - sig * net) / s
    if model_type == "score_function":  # as written: -std * x
        return (-sig * x + _b(path.covariance(t), x) * net) / s
    if model_type == "velocity_function":
        return ((_b(path.drift_coefficient(t), x) * x - net) * 2.0 * _b(path.covariance(t), x)
                / _b(path.diffusion_squared(t), x) + x) / s
    if model_type == "data_prediction_function":
        return net
    raise NotImplementedError("Unknown type of function {}".format(model_type))
