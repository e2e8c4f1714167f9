"""FlatAdam (grlhip_adam_step: Adam over one flat parameter buffer, one launch per step) against torch.optim.Adam on the
same model, losses and draws."""
import copy

import pytest
import torch

import util_models as um

DEV = "cuda:0"


@pytest.mark.gpu
@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_gpu_flat_adam_matches_torch_adam(weight_decay):
    from generativerl_preview_amd.optim import FlatAdam
    name = "small_icfm"
    kind, cfg, a = um.build(name, DEV)
    kind, cfg, b = um.build(name, DEV)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    ref = torch.optim.Adam([p for p in a.parameters() if p.requires_grad], lr=3e-3, betas=(0.8, 0.95), eps=1e-7,
                           weight_decay=weight_decay)
    opt = FlatAdam(b.model, lr=3e-3, betas=(0.8, 0.95), eps=1e-7, weight_decay=weight_decay)
    frozen = b.model.model["t_encoder"].W.detach().clone()
    gen = torch.Generator().manual_seed(4)
    B = 300
    for step in range(4):
        x0, x1 = torch.randn(B, 6, generator=gen).to(DEV), torch.rand(B, 6, generator=gen).to(DEV)
        cond, t, eps = torch.randn(B, 17, generator=gen).to(DEV), torch.rand(B, generator=gen).to(DEV), torch.randn(B, 6, generator=gen).to(DEV)
        ref.zero_grad()
        la = a.flow_matching_loss(x0, x1, cond, t=t, noise=eps)
        la.backward()
        ref.step()
        opt.zero_grad()
        lb = b.flow_matching_loss(x0, x1, cond, t=t, noise=eps)
        lb.backward()
        opt.step()
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la)), (step, float(la), float(lb))
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    assert len(pa) > 10
    for k in pa:
        scale = float(pa[k].abs().max()) + 1e-12
        assert float((pa[k] - pb[k]).abs().max()) <= 2e-5 * scale, k
    assert torch.equal(b.model.model["t_encoder"].W.detach(), frozen)      # frozen parameters are not touched
    # the parameters are views of one buffer, the state dict still has the reference's keys and shapes
    assert all(p.data_ptr() >= opt.flat_params.data_ptr() for p in b.model.parameters())
    assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {k: tuple(v.shape) for k, v in b.state_dict().items()}
    sd = opt.state_dict()
    assert sd["step"] == 4 and float(sd["exp_avg"].abs().max()) > 0


@pytest.mark.gpu
def test_gpu_flat_adam_on_a_q_network():
    """FlatAdam over a QNetwork (ReLU ConcatenateMLP, no frozen parameters): regression of Q(a, s) onto a target through
    the differentiable HIP forward, against torch.optim.Adam on a twin; the packed-weight cache of the owner is dropped
    after every step (the parameters changed behind torch's version counters)."""
    from generativerl_preview_amd.optim import FlatAdam
    from generativerl_preview_amd.rl_modules.value_network.q_network import QNetwork
    cfg = dict(backbone=dict(type="ConcatenateMLP", args=dict(hidden_sizes=[3 + 5, 64, 32], output_size=1, activation="relu")))
    torch.manual_seed(11)
    a = QNetwork(cfg).to(DEV)
    b = QNetwork(cfg).to(DEV)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    gen = torch.Generator().manual_seed(2)
    act, st, tgt = torch.randn(500, 3, generator=gen).to(DEV), torch.randn(500, 5, generator=gen).to(DEV), torch.randn(500, 1, generator=gen).to(DEV)
    b(act, st)                                     # plans the twin (FlatAdam needs the plan's parameter order)
    ref = torch.optim.Adam(a.parameters(), lr=1e-2)
    opt = FlatAdam(b, lr=1e-2)
    for _ in range(5):
        ref.zero_grad()
        la = (a(act, st) - tgt).pow(2).mean()
        la.backward()
        ref.step()
        opt.zero_grad()
        lb = (b(act, st) - tgt).pow(2).mean()
        lb.backward()
        opt.step()
        assert abs(float(la) - float(lb)) <= 2e-5 * abs(float(la)), (float(la), float(lb))
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p - q).abs().max()) <= 5e-5 * (float(p.abs().max()) + 1e-12), k


@pytest.mark.gpu
def test_gpu_training_step_replays_from_a_captured_hip_graph():
    """A whole training step -- gradient zero, flow_matching_loss forward + backward (every launch of the library goes to
    the current stream, no host synchronisation, no allocation by the library), FlatAdam(capturable=True) with its step
    count on the device (grlhip_adam_step_counted), the repack of the changed weights -- captured ONCE with
    torch.cuda.graph and replayed: the parameters after 3 eager + 5 replayed steps equal those of a twin after 8 eager
    steps with the host-counted optimizer, bit for bit (Adam's coefficients are derived in double either way)."""
    from generativerl_preview_amd.optim import FlatAdam
    name = "small_icfm"
    kind, cfg, a = um.build(name, DEV)
    kind, cfg, b = um.build(name, DEV)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    ref = FlatAdam(a.model, lr=2e-3)
    opt = FlatAdam(b.model, lr=2e-3, capturable=True)
    gen = torch.Generator().manual_seed(9)
    B = 1000
    x0, x1 = torch.randn(B, 6, generator=gen).to(DEV), torch.rand(B, 6, generator=gen).to(DEV)
    cond, t, eps = torch.randn(B, 17, generator=gen).to(DEV), torch.rand(B, generator=gen).to(DEV), torch.randn(B, 6, generator=gen).to(DEV)

    def step(m, o):
        o.zero_grad()
        loss = m.flow_matching_loss(x0, x1, cond, t=t, noise=eps)
        loss.backward()
        o.step()
        return loss.detach()

    for _ in range(8):
        la = step(a, ref)
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):        # warm-up on the capture stream's side (workspaces, autograd, LDS opt-ins)
        for _ in range(3):
            step(b, opt)
    torch.cuda.current_stream(DEV).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step(b, opt)
    # (capture runs nothing: the step count has not moved)
    assert opt.step_count == 3
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    assert opt.step_count == 8
    assert torch.equal(static_loss, la), (float(static_loss), float(la))
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k


@pytest.mark.gpu
def test_gpu_capturable_flat_adam_matches_the_host_counted_one_and_restores_its_state():
    """FlatAdam(capturable=True) -- step count and bias corrections on the device -- takes the same steps as the
    host-counted optimizer (both derive the corrections in double), and a state_dict round trip restores the count."""
    from generativerl_preview_amd.optim import FlatAdam
    name = "small_icfm"
    kind, cfg, a = um.build(name, DEV)
    kind, cfg, b = um.build(name, DEV)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    oa, ob = FlatAdam(a.model, lr=1e-3, weight_decay=0.01), FlatAdam(b.model, lr=1e-3, weight_decay=0.01, capturable=True)
    gen = torch.Generator().manual_seed(1)
    B = 257
    for _ in range(5):
        x0, x1 = torch.randn(B, 6, generator=gen).to(DEV), torch.rand(B, 6, generator=gen).to(DEV)
        cond, t, eps = torch.randn(B, 17, generator=gen).to(DEV), torch.rand(B, generator=gen).to(DEV), torch.randn(B, 6, generator=gen).to(DEV)
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m.flow_matching_loss(x0, x1, cond, t=t, noise=eps).backward()
            o.step()
    assert oa.step_count == ob.step_count == 5
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), k
    sd = ob.state_dict()
    ob.step_count = 0
    assert ob.step_count == 0
    ob.load_state_dict(sd)
    assert ob.step_count == 5
