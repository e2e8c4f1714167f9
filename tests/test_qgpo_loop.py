"""The three QGPO training phases and the guided sampler composed as the reference's training loop composes them
(grl/algorithms/qgpo.py:393-756, shapes from benchmark/qgpo/vpsde/halfcheetah_medium_expert.py), entirely on the
package's classes: behaviour-policy score matching, fake-action generation with the DPM-Solver at guidance 0, the
in-support softmax Q-learning step with its soft target update, the CEP energy-guidance step, guided sampling."""
import pytest
import torch

import golden_lib as gl

DEV = "cuda:0"


@pytest.mark.gpu
def test_gpu_qgpo_phases_compose_and_learn():
    import generativerl_preview_amd as G
    from generativerl_preview_amd.algorithms import QGPOCritic
    torch.manual_seed(0)
    A, S, B, K = 6, 17, 512, 16
    kind, cfg = gl.model_zoo()["qgpo_noise"]
    critic = QGPOCritic(dict(q_alpha=1.0, DoubleQNetwork=dict(backbone=dict(type="ConcatenateMLP", args=dict(
        hidden_sizes=[A + S, 256, 256], output_size=1, activation="relu"))))).to(DEV)
    model = G.EnergyConditionalDiffusionModel(dict(cfg, device=DEV), energy_model=critic).to(DEV)
    model.alpha = 3.0
    dpm = dict(type="DPMSolver", args=dict(order=2, device=DEV, steps=17))
    state, next_state = torch.randn(B, S, device=DEV), torch.randn(B, S, device=DEV)
    action = torch.tanh(state[:, :A] * 0.5 + 0.1 * torch.randn(B, A, device=DEV))
    reward, done = -(action ** 2).sum(-1, keepdim=True), torch.zeros(B, 1, device=DEV)

    # phase 1: behaviour policy (score matching on the noise head)
    opt = torch.optim.Adam(model.model.parameters(), lr=1e-3)
    first = last = None
    for it in range(30):
        opt.zero_grad()
        loss = model.score_matching_loss(action, state, weighting_scheme="vanilla")
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
        last = float(loss.detach())
    assert last < first, (first, last)

    # fake actions for every state and next state: (K, B, A) -> (B, K, A)   (qgpo.py:469-486)
    fake = model.sample(t_span=None, condition=state, batch_size=K, guidance_scale=0.0, solver_config=dpm)
    fake_next = model.sample(t_span=None, condition=next_state, batch_size=K, guidance_scale=0.0, solver_config=dpm)
    assert fake.shape == (K, B, A) and torch.isfinite(fake).all()
    fake, fake_next = fake.transpose(0, 1).contiguous(), fake_next.transpose(0, 1).contiguous()

    # phase 2: critic, with the reference's soft target update through .data (qgpo.py:640-655)
    qopt = torch.optim.Adam(critic.q.parameters(), lr=3e-4)
    q_first = q_last = None
    for it in range(30):
        qopt.zero_grad()
        ql = critic.q_loss(action, state, reward, next_state, done, fake_next, discount_factor=0.99)
        ql.backward()
        qopt.step()
        for p, pt in zip(critic.q.parameters(), critic.q_target.parameters()):
            pt.data.mul_(0.995)
            pt.data.add_(0.005 * p.data)
        q_first = float(ql.detach()) if q_first is None else q_first
        q_last = float(ql.detach())
    assert q_last < q_first, (q_first, q_last)

    # phase 3: energy guidance by contrastive energy prediction (qgpo.py:690-705)
    eopt = torch.optim.Adam(model.energy_guidance.parameters(), lr=1e-3)
    e_first = e_last = None
    torch.manual_seed(1)
    for it in range(40):
        eopt.zero_grad()
        el = model.energy_guidance_loss(fake, state)
        el.backward()
        eopt.step()
        e_first = float(el.detach()) if e_first is None else e_first
        e_last = float(el.detach())
    assert torch.isfinite(torch.tensor(e_last)) and e_last < e_first + 0.05, (e_first, e_last)

    # deployment: guided sampling with the shipped solver (agents/qgpo.py:63-79)
    act = model.sample(t_span=None, condition=state[:32], guidance_scale=1.0, solver_config=dpm)
    assert act.shape == (32, A) and torch.isfinite(act).all()
    q_guided = critic(act, state[:32]).mean()
    assert torch.isfinite(q_guided)
