"""Policy inference next to the env (ppg_policy_*).

The MFMA kernels compute in bf16 with fp32 accumulation and round activations to bf16 between the six layers; the reference is
`PolicyNet` -- the same architecture in float32 PyTorch -- on the same weights and the same observation rows.
TOLERANCE (stated here, checked below): |logit_hip - logit_fp32| <= 0.02 * max(1, max|logit_fp32|) for every logit, and the
greedy action agrees wherever the fp32 margin between the best and the second-best logit exceeds twice that bound.  The
reference's own RLlib module cannot be imported in this container: parity vs RLlib itself is unpinned."""
import numpy as np
import pytest
import torch

from predpreygrass_amd import _abi
from predpreygrass_amd.config import config_env
from predpreygrass_amd.policy import PolicyNet

REL_TOL = 0.02


def make_nets(Rp=7, Rq=9, scale=3.0, seed=0):
    """Default-initialised networks with the weights scaled up so that the logits are O(1) and distinct."""
    torch.manual_seed(seed)
    nets = [PolicyNet(Rp), PolicyNet(Rq)]
    with torch.no_grad():
        for net in nets:
            for m in list(net.conv) + list(net.fc):
                m.weight.mul_(scale)
                m.bias.uniform_(-0.2, 0.2)
    return nets


def test_policy_net_is_the_reference_architecture():
    """tune_ppo_base_environment.py:106-141: conv 3x3 [16, 32, 64] stride 1, fcnet_hiddens [256, 256], ReLU."""
    net = PolicyNet(9)
    assert [tuple(c.weight.shape) for c in net.conv] == [(16, 4, 3, 3), (32, 16, 3, 3), (64, 32, 3, 3)]
    assert [tuple(f.weight.shape) for f in net.fc] == [(256, 64 * 81), (256, 256), (9, 256)]
    out = net(torch.zeros(5, 4, 9, 9, dtype=torch.float64))
    assert out.shape == (5, 9) and out.dtype == torch.float32


def test_fused_policy_fails_loudly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from predpreygrass_amd.policy import FusedPolicy
    with pytest.raises(RuntimeError):
        FusedPolicy(PolicyNet(7), PolicyNet(9))


def rows_in_use(env):
    es = env.env_state
    mp = torch.arange(env.pred_capacity, device=env.device)[None, :] < es[:, _abi.ENV_N_PRED_ROWS:_abi.ENV_N_PRED_ROWS + 1]
    mq = torch.arange(env.prey_capacity, device=env.device)[None, :] < es[:, _abi.ENV_N_PREY_ROWS:_abi.ENV_N_PREY_ROWS + 1]
    return mp, mq


def check_against_fp32(envs, nets, fused, dense_obs=False, seed=0):
    """One ppg_policy_act over `envs`; returns (max relative logit error, greedy agreement rate) after asserting the tolerance."""
    if dense_obs:   # every window cell and every channel non-zero: all nine taps and the halo handling matter
        g = torch.Generator(device="cuda:0").manual_seed(seed)
        for e in envs:
            e.obs_pred.copy_(torch.rand(e.obs_pred.shape, generator=g, device="cuda:0", dtype=torch.float64).to(e.obs_pred.dtype) * 4 - 1)
            e.obs_prey.copy_(torch.rand(e.obs_prey.shape, generator=g, device="cuda:0", dtype=torch.float64).to(e.obs_prey.dtype) * 4 - 1)
    for e in envs:
        e.actions.fill_(-7)
    lg = fused.act(envs, want_logits=True)
    torch.cuda.synchronize()
    worst, agree, total = 0.0, 0, 0
    masks = [rows_in_use(e) for e in envs]
    for t, net in enumerate(nets):
        obs = torch.cat([(e.obs_prey if t else e.obs_pred)[m[t]] for e, m in zip(envs, masks)])   # env-major, row order
        n = obs.shape[0]
        assert n > 0
        with torch.no_grad():
            ref = net.to("cuda:0")(obs)
        got = lg[t][:n]
        bound = REL_TOL * max(1.0, float(ref.abs().max()))
        err = float((got - ref).abs().max())
        assert err <= bound, (t, err, bound)
        assert not bool(lg[t][n:].any())          # nothing written behind the rows in use
        worst = max(worst, err / max(1.0, float(ref.abs().max())))
        # greedy actions: stored where the env expects them (row -> slot), equal to the argmax of the kernel's own logits, and equal
        # to the fp32 argmax wherever the fp32 decision is not within the error bound
        acts = torch.cat([(e.actions[:, e.pred_capacity:] if t else e.actions[:, :e.pred_capacity])[m[t]] for e, m in zip(envs, masks)])
        assert torch.equal(acts.long(), got.argmax(1))
        top2 = ref.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2 * bound
        assert torch.equal(acts.long()[clear], ref.argmax(1)[clear])
        agree += int((acts.long() == ref.argmax(1)).sum())
        total += n
    for e, m in zip(envs, masks):   # slots not in use keep their old content
        unused = torch.cat([~m[0], ~m[1]], dim=1)
        assert bool((e.actions[unused] == -7).all())
    return worst, agree / total


@pytest.mark.gpu
@pytest.mark.parametrize("obs_dtype", [torch.float64, torch.float32])
def test_policy_logits_and_actions_match_fp32_reference(obs_dtype):
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets()
    fused = FusedPolicy(nets[0], nets[1])
    env = BatchedPredPreyGrass(dict(config_env), batch_size=37, device="cuda:0", obs_dtype=obs_dtype, seed=5)
    env.reset()
    for _ in range(60):
        env.step(random_actions=True, auto_reset=True)
    w1, a1 = check_against_fp32([env], nets, fused)                    # the env's own (sparse) observations
    w2, a2 = check_against_fp32([env], nets, fused, dense_obs=True)   # dense random windows
    print(f"max relative logit error {max(w1, w2):.2e}; greedy agreement {a1:.4f} (env obs) {a2:.4f} (dense obs)")
    assert a1 > 0.97 and a2 > 0.97


@pytest.mark.gpu
def test_policy_over_sub_batches_partial_tiles_and_other_window_sizes():
    """Three handles (sub-batches) in one call; row totals that are not multiples of the 128-sample tile; 5x5 / 11x11 windows."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
    nets = make_nets(seed=1)
    fused = FusedPolicy(nets[0], nets[1])
    grp = SubBatchedPredPreyGrass(dict(config_env), batch_size=1000, n_sub=3, device="cuda:0", seed=9)
    grp.reset()
    for _ in range(40):
        grp.step(random_actions=True, auto_reset=True)
    grp.synchronize()
    w, a = check_against_fp32(grp.subs, nets, fused, dense_obs=True, seed=3)
    assert a > 0.97
    cfg = {**config_env, "predator_obs_range": 5, "prey_obs_range": 11}
    nets2 = make_nets(5, 11, seed=2)
    fused2 = FusedPolicy(nets2[0], nets2[1])
    env = BatchedPredPreyGrass(cfg, batch_size=9, device="cuda:0", seed=1)
    env.reset()
    for _ in range(30):
        env.step(random_actions=True, auto_reset=True)
    check_against_fp32([env], nets2, fused2, dense_obs=True, seed=4)
    with pytest.raises(ValueError):
        fused.act(env)            # 7x7 / 9x9 networks on 5x5 / 11x11 observations


@pytest.mark.gpu
def test_policy_sampling_follows_the_softmax():
    """PPG_POLICY_SAMPLE: Gumbel-max with Philox.  One env, 4000 seeds: the empirical action frequencies of a row match
    softmax(logits) of that row (chi-square well below the 0.001 quantile for 8 degrees of freedom: 26.1)."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=4)
    fused = FusedPolicy(nets[0], nets[1])
    env = BatchedPredPreyGrass(dict(config_env), batch_size=2, device="cuda:0", seed=3)
    env.reset()
    lg = fused.act(env, want_logits=True)
    probs = torch.softmax(lg[1][0], 0).cpu().numpy()        # prey row 0 of env 0
    counts = np.zeros(9)
    n = 4000
    acts = torch.empty((n,), dtype=torch.int8, device="cuda:0")
    for s in range(n):
        fused.act(env, sample=True, seed=1000 + s)
        acts[s] = env.actions[0, env.pred_capacity]
    a = acts.cpu().numpy()
    assert a.min() >= 0 and a.max() <= 8
    counts = np.bincount(a, minlength=9)
    chi2 = float((((counts - n * probs) ** 2) / (n * probs + 1e-9)).sum())
    assert chi2 < 26.1, (chi2, counts, probs)
    fused.act(env, sample=True, seed=5)
    first = env.actions.clone()
    fused.act(env, sample=True, seed=5)
    assert torch.equal(first, env.actions)                   # same seed, same draw


@pytest.mark.gpu
def test_policy_closes_the_loop_for_a_whole_rollout():
    """env.step(actions from the policy) for 200 steps with auto-reset, 512 envs: every action the env consumed was legal
    (status word clean) and the episode statistics move (the policy is not the uniform random one)."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=7)
    fused = FusedPolicy(nets[0], nets[1])
    env = BatchedPredPreyGrass(dict(config_env), batch_size=512, device="cuda:0", seed=2)
    env.reset()
    for t in range(200):
        fused.act(env, sample=True, seed=t)
        env.step(env.actions, auto_reset=True)
    torch.cuda.synchronize()
    es = env.env_state.cpu().numpy()
    assert (es[:, _abi.ENV_STATUS] & _abi.STATUS_BAD_ACTION == 0).all()
    assert (es[:, _abi.ENV_CALLS] == 200).all()
