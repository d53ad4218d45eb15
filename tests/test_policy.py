"""Policy inference next to the env (ppg_policy_*).

WHAT IS PINNED: the architecture.  The reference tree holds one real RLlib checkpoint of its PPO setup (ray 2.52.1); its key names,
shapes and float32 actor weights are committed under tests/golden/rllib_checkpoint/ (make_fixture.py) and `PolicyNet` -- the float32
PyTorch restatement the kernels are compared with -- must load them with `strict=True`.  RLlib itself cannot be imported here
(SURVEY.md 8(c)), so the FORWARD semantics (zero-pad + conv + ReLU per layer, channels-last flatten, one Linear head) are restated
from ray/rllib/core/models/torch/{primitives,encoder}.py, not executed.
TOLERANCE of the MFMA kernels (bf16 operands, fp32 accumulation, activations rounded to bf16 between layers), stated here and checked
below: |logit_hip - logit_fp32| <= 0.02 * max(1, max|logit_fp32|) for every logit, and the greedy action agrees wherever the fp32
margin between the best and the second-best logit exceeds twice that bound."""
import json
import os

import ctypes

import numpy as np
import pytest
import torch

from predpreygrass_amd import _abi
from predpreygrass_amd.config import config_env
from predpreygrass_amd.policy import PolicyNet

REL_TOL = 0.02
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rllib_checkpoint")


def make_nets(Rp=7, Rq=9, scale=3.0, seed=0, **kw):
    """Default-initialised networks with the weights scaled up so that the logits are O(1) and distinct."""
    torch.manual_seed(seed)
    nets = [PolicyNet(Rp, **kw), PolicyNet(Rq, **kw)]
    with torch.no_grad():
        for net in nets:
            for m in net.conv + net.fc:
                m.weight.mul_(scale)
                m.bias.uniform_(-0.2, 0.2)
    return nets


def test_policy_net_is_what_rllib_builds():
    """tune_ppo_base_environment.py:106-141 -> RLlib: the (4,9,9) Box read as a 4 x 9 image with 9 channels, conv 3x3 [16, 32, 64]
    with explicit zero padding, channels-last flatten, ONE Linear head (fcnet_hiddens is ignored for image observations)."""
    net = PolicyNet(9)
    assert list(net.state_dict()) == [f"encoder.actor_encoder.net.0.cnn.{i}.{w}" for i in (1, 4, 7) for w in ("weight", "bias")] + \
        ["pi.net.mlp.0.weight", "pi.net.mlp.0.bias"]
    assert [tuple(c.weight.shape) for c in net.conv] == [(16, 9, 3, 3), (32, 16, 3, 3), (64, 32, 3, 3)]
    assert [tuple(f.weight.shape) for f in net.fc] == [(9, 64 * 4 * 9)]
    x = torch.rand(5, 4, 9, 9, dtype=torch.float64)
    out = net(x)
    assert out.shape == (5, 9) and out.dtype == torch.float32
    # the forward pass, written out: [N, H=4, W=9, C=9] -> NCHW -> (pad, conv, relu) x 3 -> NHWC -> flatten -> Linear
    y = x.float().permute(0, 3, 1, 2)
    for c in net.conv:
        y = torch.relu(torch.nn.functional.conv2d(torch.nn.functional.pad(y, (1, 1, 1, 1)), c.weight, c.bias))
    assert y.shape == (5, 64, 4, 9)
    want = torch.nn.functional.linear(y.permute(0, 2, 3, 1).reshape(5, -1), net.fc[0].weight, net.fc[0].bias)
    assert torch.equal(out, want)


def test_real_rllib_checkpoint_pins_the_architecture():
    """The checkpoint in the reference tree (shared_prey PPO run, Box(5,9,9), conv_filters 16/32/64/64, fcnet_hiddens [256,256]):
    its state holds four convolutions with conv1 taking NINE input channels (channels-last reading) and a single-Linear head
    [9, 2880] -- no 256-wide layer anywhere.  PolicyNet takes the actor entries with strict=True, and load_rllib_state_dict
    recovers the same network from the full state (critic and value head ignored)."""
    from predpreygrass_amd.policy import load_rllib_state_dict
    listing = json.load(open(os.path.join(FIXTURE, "state_listing.json")))
    assert listing["metadata"]["ray_version"] == "2.52.1"
    for sp in ("type_1_predator", "type_1_prey"):
        ctor, state = listing[sp]["ctor"], listing[sp]["state"]
        assert ctor["module_class"] == "DefaultPPOTorchRLModule" and ctor["observation_box_shape"] == [5, 9, 9]
        assert ctor["conv_filters"] == [[16, [3, 3], 1], [32, [3, 3], 1], [64, [3, 3], 1], [64, [3, 3], 1]]
        assert ctor["fcnet_hiddens"] == [256, 256] and "head_fcnet_hiddens" not in ctor["model_config_keys"]
        assert state["encoder.actor_encoder.net.0.cnn.1.weight"]["shape"] == [16, 9, 3, 3]
        assert state["pi.net.mlp.0.weight"]["shape"] == [9, 2880] and "pi.net.mlp.2.weight" not in state
        assert not any(256 in v["shape"] for v in state.values())          # fcnet_hiddens built nothing
    actor = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(FIXTURE, "type_1_predator_actor.npz")).items()}
    net = PolicyNet(9, 9, "hwc", obs_channels=5, conv_channels=(16, 32, 64, 64))
    net.load_state_dict(actor, strict=True)                                # names AND shapes
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == \
        {k: v["shape"] for k, v in listing["type_1_predator"]["state"].items() if k in actor}
    full = dict(actor)
    for k, v in listing["type_1_predator"]["state"].items():
        if k not in full:
            full[k] = torch.randn(v["shape"])                              # critic encoder, value head, log_std constants
    got = load_rllib_state_dict(full)
    assert (got.layout, got.flatten, got.obs_range, got.obs_channels, got.n_actions) == ("hwc", "nhwc", 9, 5, 9)
    assert got.conv_channels == (16, 32, 64, 64) and got.head_hiddens == ()
    x = torch.rand(7, 5, 9, 9) * 3
    assert torch.equal(got(x), net(x))
    assert float(net(x).abs().max()) > 1e-3
    # a state dict in SORTED key order (cnn.1, cnn.10, cnn.4, cnn.7 -- what a re-serialised checkpoint or the listing itself holds):
    # the layers are taken in the numeric order of their indices
    resorted = {k: full[k] for k in sorted(full)}
    assert [k for k in resorted if "actor_encoder" in k and k.endswith("weight")][1].endswith("cnn.10.weight")
    assert torch.equal(load_rllib_state_dict(resorted)(x), net(x))


def rllib_style_state_dict(net, shared_encoder=False, numpy_values=False):
    """The state of an RLlib PPO RLModule around `net`'s actor network: the critic's encoder and the value head next to it (or one
    shared encoder, vf_share_layers), key names as in the real checkpoint."""
    sd = {}
    for k, v in net.state_dict().items():
        if shared_encoder:
            k = k.replace("encoder.actor_encoder.", "encoder.encoder.")
        sd[k] = v.detach().clone()
        if not shared_encoder and k.startswith("encoder.actor_encoder."):
            sd[k.replace("actor_encoder", "critic_encoder")] = torch.randn_like(v)
    sd["pi.log_std_clip_param_const"] = torch.zeros(1)
    dims = [net.fc[0].in_features, *net.head_hiddens, 1]
    for l in range(len(dims) - 1):
        sd[f"vf.net.mlp.{2 * l}.weight"] = torch.randn(dims[l + 1], dims[l])
        sd[f"vf.net.mlp.{2 * l}.bias"] = torch.randn(dims[l + 1])
    if numpy_values:
        sd = {k: v.numpy() for k, v in sd.items()}
    return sd


@pytest.mark.parametrize("layout,R", [("chw", 7), ("chw", 9), ("hwc", 7), ("hwc", 9), ("hwc", 5)])
@pytest.mark.parametrize("hiddens", [(), (256,), (256, 256), (128, 64)])
@pytest.mark.parametrize("shared,as_numpy", [(False, False), (True, True)])
def test_load_rllib_state_dict_takes_the_architecture_from_the_shapes(layout, R, hiddens, shared, as_numpy):
    """conv1 [16,4,3,3] = channel-first, [16,R,3,3] = RLlib's channels-last reading of the (4,R,R) Box; 0, 1 or 2 hidden head layers
    (head_fcnet_hiddens); the loaded network gives the same logits as the one the state dict was taken from."""
    from predpreygrass_amd.policy import load_rllib_state_dict
    torch.manual_seed(R)
    src = PolicyNet(R, 9, layout, head_hiddens=hiddens)
    net = load_rllib_state_dict(rllib_style_state_dict(src, shared, as_numpy))
    assert (net.layout, net.obs_range, net.n_actions, net.obs_channels, net.head_hiddens) == (layout, R, 9, 4, hiddens)
    assert tuple(net.conv[0].weight.shape) == ((16, 4, 3, 3) if layout == "chw" else (16, R, 3, 3))
    assert net.fc[0].weight.shape[1] == (64 * R * R if layout == "chw" else 64 * 4 * R)
    x = torch.rand(6, 4, R, R, dtype=torch.float64)
    assert torch.equal(net(x), src(x))


def test_load_rllib_state_dict_other_depths_and_channels():
    """The second-generation tune scripts size the encoder by the window ((R - 1) // 2 layers of 16, 32, 64, 64 ..,
    red_queen/utils/networks.py:18-25) and the walls variant can add a fifth observation channel."""
    from predpreygrass_amd.policy import load_rllib_state_dict
    for R, chans, C in [(5, (16, 32), 4), (9, (16, 32, 64, 64), 4), (11, (16, 32, 64, 64, 64), 5), (3, (16,), 4)]:
        src = PolicyNet(R, 25, "hwc", obs_channels=C, conv_channels=chans)
        net = load_rllib_state_dict(rllib_style_state_dict(src))
        assert (net.conv_channels, net.obs_channels, net.obs_range, net.n_actions) == (chans, C, R, 25)
        x = torch.rand(3, C, R, R)
        assert torch.equal(net(x), src(x))


def test_load_rllib_state_dict_fails_loudly():
    from predpreygrass_amd.policy import load_rllib_state_dict
    good = rllib_style_state_dict(PolicyNet(7, 9, "hwc", head_hiddens=(256, 256)))
    for mutate, what in [
        (lambda d: [d.pop(k) for k in list(d) if k.startswith("pi.net.mlp")], "linear weights"),          # no head at all
        (lambda d: d.pop("encoder.actor_encoder.net.0.cnn.4.bias"), "no bias"),
        (lambda d: d.update({"encoder.actor_encoder.net.0.cnn.4.weight": torch.zeros(32, 16, 5, 5)}), "not 3x3"),   # another filter size
        (lambda d: d.update({"pi.net.mlp.0.weight": torch.zeros(256, 64 * 4 * 7 + 64), "pi.net.mlp.0.bias": torch.zeros(256)}), "neither"),
        (lambda d: d.update({"pi.net.mlp.2.weight": torch.zeros(128, 255), "pi.net.mlp.2.bias": torch.zeros(128)}), "chain of Linear"),
        (lambda d: d.update({"pi.net.mlp.0.weight": torch.zeros(512, 64 * 4 * 7), "pi.net.mlp.0.bias": torch.zeros(512),
                             "pi.net.mlp.2.weight": torch.zeros(256, 512)}), "up to 256"),
        (lambda d: d.update({"encoder.actor_encoder.net.0.cnn.7.weight": torch.zeros(128, 32, 3, 3),
                             "encoder.actor_encoder.net.0.cnn.7.bias": torch.zeros(128)}), "output channels"),
    ]:
        d = dict(good)
        mutate(d)
        with pytest.raises(ValueError, match=what):
            load_rllib_state_dict(d)
    with pytest.raises(ValueError, match="obs_range 9"):
        load_rllib_state_dict(good, obs_range=9)     # a 7-window network for 9-window observations
    deep = rllib_style_state_dict(PolicyNet(9, 9, "hwc", conv_channels=(16, 32, 64, 64)))
    deep.update({"pi.net.mlp.0.weight": torch.zeros(256, 64 * 36), "pi.net.mlp.0.bias": torch.zeros(256),
                 "pi.net.mlp.2.weight": torch.zeros(9, 256), "pi.net.mlp.2.bias": torch.zeros(9)})
    with pytest.raises(ValueError, match="exactly three convolutions"):
        load_rllib_state_dict(deep)


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
@pytest.mark.parametrize("Rp,Rq", [(7, 9), (5, 11), (9, 15)])
@pytest.mark.parametrize("hiddens", [(), (256,), (256, 256), (96, 160)])
def test_rllib_networks_with_any_head_depth_match_fp32_reference(Rp, Rq, hiddens):
    """RLlib's reading of the (4,R,R) Box -- a 4 x R image with R channels, channels-last flatten -- with the single-Linear head RLlib
    builds by default (the all-in-LDS kernels) and with 1 or 2 hidden head layers (head_fcnet_hiddens; widths below 256 are
    zero-padded): one and two channel blocks into conv1 (R <= 8 / R >= 9), loaded from an RLlib-style state dict, float64 and
    float32 observations, sparse and dense windows."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy, load_rllib_state_dict
    src = make_nets(Rp, Rq, seed=11, head_hiddens=hiddens)
    nets = [load_rllib_state_dict(rllib_style_state_dict(n)) for n in src]
    assert [(n.layout, n.flatten, n.head_hiddens) for n in nets] == [("hwc", "nhwc", hiddens)] * 2
    fused = FusedPolicy(nets[0], nets[1])
    cfg = {**config_env, "predator_obs_range": Rp, "prey_obs_range": Rq}
    for dt in (torch.float64, torch.float32):
        env = BatchedPredPreyGrass(cfg, batch_size=41, device="cuda:0", obs_dtype=dt, seed=6)
        env.reset()
        for _ in range(50):
            env.step(random_actions=True, auto_reset=True)
        w1, a1 = check_against_fp32([env], nets, fused)
        w2, a2 = check_against_fp32([env], nets, fused, dense_obs=True, seed=2)
        print(f"hwc {Rp}/{Rq} {hiddens} {dt}: max relative logit error {max(w1, w2):.2e}; greedy agreement {a1:.4f} / {a2:.4f}")
        assert a1 > 0.97 and a2 > 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(layout="chw", head_hiddens=(256, 256)), dict(layout="chw"), dict(layout="chw", flatten="nhwc"),
                                dict(layout="hwc", flatten="nchw"), dict(conv_channels=(16, 32)), dict(conv_channels=(16,)),
                                dict(conv_channels=(16, 32, 64, 64)), dict(conv_channels=(16, 32, 64, 64, 64)),
                                dict(conv_channels=(12, 24, 40)), dict(conv_channels=(16, 32, 64), head_hiddens=(256, 256), flatten="nchw")])
def test_other_network_shapes_match_fp32_reference(kw):
    """Everything ppg_policy_create_spec takes: channel-first images (rounds 2-3's network: R x R image, 256/256 head, channel-major
    flatten), either flatten order with either reading, 1 to 5 convolutions (the second-generation tune scripts use (R - 1) // 2
    layers), channel counts that are not multiples of 8."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=13, **kw)
    fused = FusedPolicy(nets[0], nets[1])
    env = BatchedPredPreyGrass(dict(config_env), batch_size=23, device="cuda:0", seed=7)
    env.reset()
    for _ in range(40):
        env.step(random_actions=True, auto_reset=True)
    w1, a1 = check_against_fp32([env], nets, fused)
    w2, a2 = check_against_fp32([env], nets, fused, dense_obs=True, seed=5)
    print(f"{kw}: max relative logit error {max(w1, w2):.2e}; greedy agreement {a1:.4f} / {a2:.4f}")
    assert a1 > 0.97 and a2 > 0.97


@pytest.mark.gpu
def test_real_rllib_checkpoint_weights_on_the_matrix_cores():
    """The actor of the checkpoint the reference tree holds (Box(5,9,9), four convolutions, Linear(2880, 9)) evaluated on the
    5-channel 9x9 observations of the walls variant with its visibility channel: the kernels' logits against the float32 module
    holding the same real weights, on the env's own observations and on dense ones."""
    from predpreygrass_amd.policy import FusedPolicy, load_rllib_state_dict
    from predpreygrass_amd.red_queen import BatchedRedQueen
    from predpreygrass_amd.walls_occlusion import config_env_zigzag_walls
    actor = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(FIXTURE, "type_1_predator_actor.npz")).items()}
    net = load_rllib_state_dict(actor)
    assert (net.obs_channels, net.obs_range, net.conv_channels, net.head_hiddens) == (5, 9, (16, 32, 64, 64), ())
    fused = FusedPolicy(net, net)
    cfg = {**config_env_zigzag_walls, "predator_obs_range": 9, "prey_obs_range": 9}
    env = BatchedRedQueen(cfg, batch_size=29, device="cuda:0", walls=True, obs_dtype=torch.float32, seed=3)   # (the reference's dtype)
    assert env.obs_channels == 5
    env.set_walls(cfg["manual_wall_positions"])
    env.reset()
    for _ in range(40):
        env.step(random_actions=True, auto_reset=True)
    w1, a1 = check_against_fp32([env], [net, net], fused)
    print(f"real checkpoint, env observations: max relative logit error {w1:.2e}; greedy agreement {a1:.4f}")
    assert a1 > 0.95
    # Dense random windows in [-1, 3) are far from anything this network was trained on, and its four trained layers amplify the
    # rounding of bf16 operands beyond the 2 % bound (a float32 PyTorch forward pass with weights and activations rounded to
    # bf16 the way the kernels round them is itself 3-5 % away from the float32 one).  So on these inputs the kernels are compared
    # with THAT emulation -- same roundings, only the summation order differs -- within 0.5 % of the largest logit.
    g = torch.Generator(device="cuda:0").manual_seed(8)
    for t in (env.obs_pred, env.obs_prey):
        t.copy_((torch.rand(t.shape, generator=g, device="cuda:0", dtype=torch.float32) * 4 - 1).to(t.dtype))
    lg = fused.act(env, want_logits=True)
    torch.cuda.synchronize()
    mp, mq = rows_in_use(env)
    net = net.to("cuda:0")

    def bf(t):
        return t.bfloat16().float()
    for got, obs in ((lg[0], env.obs_pred[mp]), (lg[1], env.obs_prey[mq])):
        with torch.no_grad():
            y = bf(obs.float()).permute(0, 3, 1, 2)
            for c in net.conv:
                y = bf(torch.relu(torch.nn.functional.conv2d(torch.nn.functional.pad(y, (1, 1, 1, 1)), bf(c.weight), c.bias)))
            emu = torch.nn.functional.linear(y.permute(0, 2, 3, 1).flatten(1), bf(net.fc[0].weight), net.fc[0].bias)
            ref = net(obs)
        n = obs.shape[0]
        err_emu, err_fp32 = float((got[:n] - emu).abs().max()), float((got[:n] - ref).abs().max())
        print(f"real checkpoint, dense windows: |hip - bf16 emulation| {err_emu:.3e}, |hip - fp32| {err_fp32:.3e}, max |logit| {float(ref.abs().max()):.2f}")
        assert err_emu <= 0.005 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError, match="channel"):
        from predpreygrass_amd.batched import BatchedPredPreyGrass
        fused.act(BatchedPredPreyGrass({**config_env, "predator_obs_range": 9}, batch_size=2, device="cuda:0").reset())


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(head_hiddens=(256, 256)), dict(layout="chw", head_hiddens=(256, 256))])
def test_bfloat16_observation_rows_give_bit_identical_logits(kw):
    """obs_dtype bfloat16: ppg_step writes the rows the policy stages -- the float64 value rounded exactly as the policy kernels
    round it -- a quarter of the bytes, the SAME logits and actions bit for bit (no float64 row is written or read
    in a rollout whose policy runs next to the env)."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=21, **kw)
    fused = FusedPolicy(nets[0], nets[1])
    out = []
    for dt in (torch.float64, torch.bfloat16):
        env = BatchedPredPreyGrass(dict(config_env), batch_size=300, device="cuda:0", obs_dtype=dt, seed=4)
        env.reset()
        for _ in range(50):
            env.step(random_actions=True, auto_reset=True)
        lg = fused.act(env, want_logits=True)
        torch.cuda.synchronize()
        out.append((env, lg[0].clone(), lg[1].clone(), env.actions.clone()))
    (e64, p64, q64, a64), (e16, p16, q16, a16) = out
    assert torch.equal(e64.obs_prey.float().bfloat16().view(torch.int16), e16.obs_prey.view(torch.int16))
    assert torch.equal(p64, p16) and torch.equal(q64, q16) and torch.equal(a64, a16)
    assert bool(q16.abs().sum() > 0)
    # and a closed loop on the bfloat16 rows stays legal
    for t in range(30):
        fused.act(e16, sample=True, seed=t)
        e16.step(e16.actions, auto_reset=True)
    torch.cuda.synchronize()
    assert (e16.env_state[:, _abi.ENV_STATUS] & _abi.STATUS_BAD_ACTION == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("Rp,Rq", [(5, 11), (9, 15), (3, 13)])
def test_bfloat16_rows_of_other_windows_through_the_pipeline_kernels(Rp, Rq):
    """The two-role pipeline kernels fetch bfloat16 rows as aligned 8-byte chunks through LDS when a sub-group's rows fit three chunks
    per thread (5x5: ten samples per load; 11x11 / 13x13: two / one), and with one load per channel otherwise (15x15; 3x3: too many
    samples per load for the table's padding): both against the fp32 module on the same rows, sparse and dense windows -- and the
    float64 rows of the same state give the same logits bit for bit."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(Rp, Rq, seed=31)
    fused = FusedPolicy(nets[0], nets[1])
    cfg = {**config_env, "predator_obs_range": Rp, "prey_obs_range": Rq}
    logits = []
    for dt in (torch.bfloat16, torch.float64):
        env = BatchedPredPreyGrass(cfg, batch_size=150, device="cuda:0", obs_dtype=dt, seed=8)
        env.reset()
        for _ in range(40):
            env.step(random_actions=True, auto_reset=True)
        lg = fused.act(env, want_logits=True)
        torch.cuda.synchronize()
        logits.append((lg[0].clone(), lg[1].clone()))
        if dt == torch.bfloat16:
            w1, a1 = check_against_fp32([env], nets, fused)
            w2, a2 = check_against_fp32([env], nets, fused, dense_obs=True, seed=5)
            assert a1 > 0.97 and a2 > 0.97
    assert torch.equal(logits[0][0], logits[1][0]) and torch.equal(logits[0][1], logits[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float64])
def test_pipeline_and_one_role_kernels_agree_and_each_is_bit_stable(dt, monkeypatch):
    """The two-role pipeline kernels (ppg_policy_pipe.h) against the one-role direct-head kernels (PPG_POLICY_PIPE=0 at creation time
    selects the latter for the same network).  conv2, conv3 and the head do the same arithmetic in the same order in both; since round 5
    the pipeline's FIRST convolution runs on v_mfma_f32_16x16x32_bf16 with one k-step per kernel row and the bias in the accumulators
    (Conv1X), where the one-role kernels keep the 32x32x16 form with the bias as a bf16 pair in the GEMM: the same products summed in
    another order.  Bit for bit: both launches of either family (greedy twice, sampled twice); across the families: logits within 2e-3
    of the largest logit, and the same action wherever the one-role kernels' margin exceeds twice that."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=51)
    env = BatchedPredPreyGrass(dict(config_env), batch_size=700, device="cuda:0", obs_dtype=dt, seed=14)
    env.reset()
    for _ in range(60):
        env.step(random_actions=True, auto_reset=True)
    out = []
    for pipe in ("1", "0"):
        monkeypatch.setenv("PPG_POLICY_PIPE", pipe)
        fused = FusedPolicy(nets[0], nets[1])
        runs = []
        for _ in range(2):
            lg = fused.act(env, want_logits=True)
            greedy = env.actions.clone()
            fused.act(env, sample=True, seed=77)
            torch.cuda.synchronize()
            runs.append((lg[0].clone(), lg[1].clone(), greedy, env.actions.clone()))
        for a, b in zip(*runs):
            assert torch.equal(a, b)            # run-to-run: bit for bit
        out.append(runs[0])
    n_rows = [int(env.env_state[:, w].sum()) for w in (_abi.ENV_N_PRED_ROWS, _abi.ENV_N_PREY_ROWS)]
    for t in range(2):
        a, b = out[0][t][:n_rows[t]], out[1][t][:n_rows[t]]
        bound = 2e-3 * max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= bound, (t, float((a - b).abs().max()), bound)
        top = b.topk(2, dim=1).values
        sure = (top[:, 0] - top[:, 1]) > 2 * bound
        assert bool(sure.float().mean() > 0.9) and torch.equal(a.argmax(1)[sure], b.argmax(1)[sure])
    assert bool(out[0][1].abs().sum() > 0) and not torch.equal(out[0][2], out[0][3])


@pytest.mark.gpu
@pytest.mark.parametrize("batch,steps,dt", [(4096, 150, torch.bfloat16), (700, 60, torch.float64), (700, 60, torch.float32),
                                            (6500, 500, torch.bfloat16), (3, 30, torch.bfloat16), (1, 5, torch.bfloat16)])
def test_one_fused_launch_gives_what_the_separate_launches_give(batch, steps, dt, monkeypatch):
    """Round 5: ppg_policy_act with both species = ONE launch (ppg_policy_pipe2_*: the plan computed by every workgroup for itself, the
    launch's workgroups divided between the species).  PPG_POLICY_FUSED=0 sends the same call through the plan launch + two forward
    launches of rounds 2-4: identical logits, greedy AND sampled actions -- at the benchmark's size, with float64 / float32 rows, with
    shares longer than a workgroup's table (6500 envs: several tiles per workgroup), with fewer rows than workgroups, with one env."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=52)
    env = BatchedPredPreyGrass(dict(config_env), batch_size=batch, device="cuda:0", obs_dtype=dt, seed=15)
    env.reset()
    for _ in range(steps):
        env.step(random_actions=True, auto_reset=True)
    fused = FusedPolicy(nets[0], nets[1])
    out = []
    for mode in ("1", "0"):
        monkeypatch.setenv("PPG_POLICY_FUSED", mode)
        env.actions.fill_(_abi.ACTION_NONE)
        lg = fused.act(env, want_logits=True)
        greedy = env.actions.clone()
        fused.act(env, sample=True, seed=78)
        torch.cuda.synchronize()
        out.append((lg[0].clone(), lg[1].clone(), greedy, env.actions.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert bool(out[0][1].abs().sum() > 0) and (batch < 10 or not torch.equal(out[0][2], out[0][3]))
    env.step(env.actions, auto_reset=True)
    torch.cuda.synchronize()
    assert (env.env_state[:, _abi.ENV_STATUS] & _abi.STATUS_BAD_ACTION == 0).all()


@pytest.mark.gpu
def test_fused_launch_over_several_sub_batches_and_an_extinct_species(monkeypatch):
    """The fused launch over three handles (the envs of a GPU as sub-batches), and with one species extinct everywhere (every
    workgroup then serves the other one)."""
    from predpreygrass_amd.policy import FusedPolicy
    from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
    nets = make_nets(seed=53)
    group = SubBatchedPredPreyGrass(dict(config_env), batch_size=900, n_sub=3, device="cuda:0", obs_dtype=torch.bfloat16, seed=3)
    group.reset()
    for _ in range(80):
        group.step(random_actions=True, auto_reset=True)
    group.synchronize()
    fused = FusedPolicy(nets[0], nets[1])
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("PPG_POLICY_FUSED", mode)
        lg = fused.act(group.subs, want_logits=True)
        torch.cuda.synchronize()
        res.append((lg[0].clone(), lg[1].clone(), torch.cat([e.actions for e in group.subs]).clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    cfg = dict(config_env, n_initial_active_predator=0)
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    env = BatchedPredPreyGrass(cfg, batch_size=300, device="cuda:0", obs_dtype=torch.bfloat16, seed=4)
    env.reset()
    env.step(random_actions=True)
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("PPG_POLICY_FUSED", mode)
        lg = fused.act(env, want_logits=True)
        torch.cuda.synchronize()
        res.append((lg[0].clone(), lg[1].clone(), env.actions.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][0].abs().sum()) == 0.0 and float(res[0][1].abs().sum()) > 0.0   # (no predator row anywhere)


@pytest.mark.gpu
def test_a_workgroups_share_larger_than_its_sample_table():
    """6500 envs: ~200 k prey rows on 256 workgroups = shares of ~800 samples, more than the pipeline kernels' LDS table holds (693 at
    9x9) -- every workgroup walks two tiles (the pipeline drains and refills); and 3 envs: most workgroups get nothing."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=41)
    fused = FusedPolicy(nets[0], nets[1])
    for batch, steps in ((6500, 500), (3, 30)):   # (500 steps: the prey have grown from 10 to ~32 per env)
        env = BatchedPredPreyGrass(dict(config_env), batch_size=batch, device="cuda:0", obs_dtype=torch.bfloat16, seed=12)
        env.reset()
        for _ in range(steps):
            env.step(random_actions=True, auto_reset=True)
        n_prey = int(env.env_state[:, _abi.ENV_N_PREY_ROWS].sum())
        assert (n_prey > 256 * 700) == (batch == 6500)
        w, a = check_against_fp32([env], nets, fused)
        assert a > 0.97
        fused.act(env, sample=True, seed=3)
        env.step(env.actions, auto_reset=True)
        torch.cuda.synchronize()
        assert (env.env_state[:, _abi.ENV_STATUS] & _abi.STATUS_BAD_ACTION == 0).all()
        del env


@pytest.mark.gpu
def test_sampled_actions_do_not_depend_on_where_they_are_written():
    """PPG_POLICY_SAMPLE is keyed by (seed, env, row): the same seed gives the same actions in another tensor, in another
    process, on another rank; split into sub-batches the envs keep their draws."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy
    nets = make_nets(seed=5)
    fused = FusedPolicy(nets[0], nets[1])
    env = BatchedPredPreyGrass(dict(config_env), batch_size=64, device="cuda:0", seed=8)
    env.reset()
    for _ in range(20):
        env.step(random_actions=True, auto_reset=True)
    fused.act(env, sample=True, seed=77)
    own = env.actions.clone()
    pad = torch.empty((1 << 20,), dtype=torch.int8, device="cuda:0")     # (moves the next allocation somewhere else)
    other = torch.full_like(env.actions, -1)
    fused.act(env, actions=[other], sample=True, seed=77)
    assert other.data_ptr() != env.actions.data_ptr()
    assert torch.equal(own, other)
    fused.act(env, actions=[other], sample=True, seed=78)
    assert not torch.equal(own, other)
    del pad


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


@pytest.mark.gpu
def test_pipeline_barrier_status_stays_clean_over_a_rollout():
    """The pipeline kernels' private barrier is a spin on an LDS counter (ppg_policy_pipe.h: pipe_wait) -- correct while all eight
    wavefronts are resident, a silent hang otherwise.  The -DPPG_PIPE_DEBUG build (tools/build_pipe_debug.sh) bounds the spin and reports
    through status words that ppg_policy_act checks after every launch: over a closed-loop rollout (fused launch and separate launches,
    shares of one and of several tiles) no wait ever gives up; and a report, when there is one, surfaces as an error."""
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_build", "libppg_hip_pipedbg.so")
    if not os.path.exists(lib):
        pytest.skip("tools/_build/libppg_hip_pipedbg.so has not been built (tools/build_pipe_debug.sh)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from predpreygrass_amd import _abi\n"
        "from predpreygrass_amd.batched import BatchedPredPreyGrass\n"
        "from predpreygrass_amd.config import config_env\n"
        "from predpreygrass_amd.policy import FusedPolicy, PolicyNet\n"
        "torch.manual_seed(7)\n"
        "nets = [PolicyNet(7), PolicyNet(9)]\n"
        "fused = FusedPolicy(nets[0], nets[1])\n"
        "for batch, steps, mode in ((512, 120, '1'), (512, 40, '0'), (6500, 12, '1')):\n"
        "    os.environ['PPG_POLICY_FUSED'] = mode\n"
        "    env = BatchedPredPreyGrass(dict(config_env), batch_size=batch, device='cuda:0', obs_dtype=torch.bfloat16, seed=2)\n"
        "    env.reset()\n"
        "    if batch > 1000:\n"
        "        for _ in range(400): env.step(random_actions=True, auto_reset=True)\n"
        "    for t in range(steps):\n"
        "        fused.act(env, sample=True, seed=t)\n"
        "        env.step(env.actions, auto_reset=True)\n"
        "    torch.cuda.synchronize()\n"
        "    assert (env.env_state[:, _abi.ENV_STATUS] & _abi.STATUS_BAD_ACTION == 0).all()\n"
        "os.environ['PPG_PIPE_DEBUG_INJECT'] = '1'\n"
        "try:\n"
        "    fused.act(env, sample=True, seed=1)\n"
        "    print('NO ERROR')\n"
        "except RuntimeError as exc:\n"
        "    assert 'gave up waiting' in str(exc), exc\n"
        "    print('PIPE-STATUS-CLEAN')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, PPG_HIP_LIB=lib))
    assert out.returncode == 0 and "PIPE-STATUS-CLEAN" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


# ---- the weight repack (MFMA operand order), restated in NumPy and checked against a plain convolution ON THE CPU --------------------
def _bf16_round(a):
    """float32 -> the nearest bfloat16 (ties to even), as float64."""
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def _bf16_bits_to_f64(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def _pack(lib, spec, what):
    n = ctypes.c_uint64(0)
    assert lib.ppg_policy_pack(ctypes.byref(spec), what, None, 0, ctypes.byref(n)) == 0
    out = np.zeros(n.value, dtype=np.uint16)
    assert lib.ppg_policy_pack(ctypes.byref(spec), what, out.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n)) == 0
    return _bf16_bits_to_f64(out)


def _row_feature(r):   # MFMA row r of a 32-row tile computes this channel of the tile (csrc/ppg_policy.h: ORIENTATION)
    return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3)


@pytest.mark.parametrize("layout,C,R,chans", [("hwc", 4, 9, (16, 32, 64)), ("hwc", 4, 7, (16, 32, 64)), ("chw", 4, 9, (16, 32, 64)),
                                             ("hwc", 5, 9, (16, 32, 64, 64)), ("hwc", 4, 5, (12, 20, 40)), ("chw", 6, 5, (16, 24))])
def test_weight_fragments_restated_in_numpy_equal_a_plain_convolution(layout, C, R, chans):
    """what ppg_policy_create_spec uploads (ppg_policy_pack: device-free) read back through a NumPy restatement of
    the MFMA operand order -- 32x32x16: lane (row = lane & 31, half = lane >> 5) holds A[row][8 half + j], K block q = 2 ks + half =
    tap * CBIN + channel block, block 9 CBIN = the bias as bf16 head + remainder against B = {1, 1, 0..}, MFMA row -> channel by
    _row_feature; the pipeline's first convolution as 16x16x32: one k-step per kernel row, lane quarters 0-2 = the three taps of
    channels 0-7, quarter 3 = the ninth channel's three taps; the head as 16x16x32 over area F = [position][channels] -- against
    torch's conv2d / linear on the bf16-rounded weights in float64.  A wrong lane, tap or channel block dies here, on the CPU."""
    import __graft_entry__ as g
    from predpreygrass_amd.policy import PolicyNet
    lib = _abi.bind(ctypes.CDLL(g.build_hip()))
    torch.manual_seed(C * 100 + R)
    net = PolicyNet(R, 9, layout, obs_channels=C, conv_channels=chans)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    convs = sorted((k for k in sd if k.endswith("weight") and sd[k].dim() == 4), key=lambda k: int(k.split("cnn.")[1].split(".")[0]))
    head = [k for k in sd if k.startswith("pi.") and k.endswith("weight")][0]
    spec = _abi.PpgPolicySpec()
    spec.obs_channels, spec.obs_range, spec.n_actions = C, R, 9
    spec.layout = _abi.POLICY_LAYOUT_HWC if layout == "hwc" else _abi.POLICY_LAYOUT_CHW
    spec.flatten = _abi.POLICY_FLATTEN_NHWC
    spec.n_conv, spec.n_fc = len(chans), 1
    keep = []
    for l, k in enumerate(convs):
        w, b = sd[k].numpy().astype(np.float32).copy(), sd[k[:-6] + "bias"].numpy().astype(np.float32).copy()
        keep += [w, b]
        spec.conv_out[l], spec.conv_w[l], spec.conv_b[l] = w.shape[0], w.ctypes.data, b.ctypes.data
    hw, hb = sd[head].numpy().astype(np.float32).copy(), sd[head[:-6] + "bias"].numpy().astype(np.float32).copy()
    keep += [hw, hb]
    spec.fc_out[0], spec.fc_w[0], spec.fc_b[0] = 9, hw.ctypes.data, hb.ctypes.data
    IH, IW, CIN = (C, R, R) if layout == "hwc" else (R, R, C)
    rng = np.random.default_rng(7)
    cin = CIN
    for l, k in enumerate(convs):
        w, b = keep[2 * l].astype(np.float64), keep[2 * l + 1].astype(np.float64)
        cout = w.shape[0]
        x = _bf16_round(rng.uniform(-2, 2, (cin, IH, IW)))
        xp = np.zeros((cin + 16, IH + 2, IW + 2))
        xp[:cin, 1:-1, 1:-1] = x
        want = torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(_bf16_round(w)), torch.from_numpy(b), padding=1)[0].numpy()
        # ---- 32x32x16 fragments [row tile][k-step][lane][8]
        CBIN, MT = (2 if CIN > 8 else 1, 1) if l == 0 else (2, 1) if l == 1 else (4, 2) if l == 2 else (8, 2)
        Q = 9 * CBIN
        KS = (Q + 2) // 2
        frag = _pack(lib, spec, l).reshape(MT, KS, 64, 8)
        got = np.zeros((cout, IH, IW))
        for mt in range(MT):
            for r in range(32):
                co = 32 * mt + _row_feature(r)
                if co >= cout:
                    assert not frag[mt, :, [r, r + 32], :].any()     # padding rows hold zeros
                    continue
                for ks in range(KS):
                    for h in range(2):
                        q, a = 2 * ks + h, frag[mt, ks, r + 32 * h]
                        if q > Q:
                            assert not a.any()
                        elif q == Q:                                  # the bias block: B = {1, 1, 0, ...}
                            got[co] += a[0] + a[1]
                        else:
                            tap, cb = divmod(q, CBIN)
                            ky, kx = divmod(tap, 3)
                            got[co] += np.einsum("j,jyx->yx", a, xp[8 * cb:8 * cb + 8, ky:ky + IH, kx:kx + IW])
        assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max()), (l, np.abs(got - want).max())
        # ---- the pipeline's first convolution: 16x16x32 fragments [kernel row][lane][8], bias in the accumulators
        if l == 0 and CIN <= 9:
            fx = _pack(lib, spec, _abi.POLICY_PACK_CONV1X).reshape(3, 64, 8)
            gx = np.tile(b[:, None, None], (1, IH, IW)) if cout <= 16 else None
            for ky in range(3):
                for lane in range(64):
                    co, kq, a = lane & 15, lane >> 4, fx[ky, lane]
                    if co >= cout:
                        assert not a.any()
                        continue
                    if kq < 3:      # the tap kx = kq of channels 0-7
                        gx[co] += np.einsum("j,jyx->yx", a, xp[0:8, ky:ky + IH, kq:kq + IW])
                    else:           # the ninth channel at the taps kx = 0, 1, 2; elements 3-7 unused
                        assert not a[3:].any()
                        for kx in range(3):
                            gx[co] += a[kx] * xp[8, ky:ky + IH, kx:kx + IW]
            assert np.abs(gx - want).max() < 1e-9 * max(1.0, np.abs(want).max())
        cin = cout
    # ---- the head: 16x16x32 fragments [action tile][k-step][lane][8] over area F = [position][channels padded to blocks of 8]
    P, cl = IH * IW, chans[-1]
    flat_c = (cl + 7) // 8 * 8
    ksteps = (P * flat_c + 31) // 32
    fh = _pack(lib, spec, _abi.POLICY_PACK_HEAD).reshape(1, ksteps, 64, 8)
    feat = _bf16_round(rng.uniform(0, 2, (cl, IH, IW)))           # conv output [channel][row][column]
    F = np.zeros(ksteps * 32)
    for q in range(P):
        F[q * flat_c:q * flat_c + cl] = feat[:, q // IW, q % IW]
    got = np.zeros(9)
    for ks in range(ksteps):
        for lane in range(64):
            a, kq = lane & 15, lane >> 4
            if a < 9:
                got[a] += fh[0, ks, lane] @ F[32 * ks + 8 * kq:32 * ks + 8 * kq + 8]
            else:
                assert not fh[0, ks, lane].any()
    flat = torch.from_numpy(feat).permute(1, 2, 0).reshape(-1)      # RLlib: channels-last flatten
    want = (torch.from_numpy(_bf16_round(hw)) @ flat).numpy()
    assert np.abs(got - want).max() < 1e-9 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("layout,C,R", [("hwc", 4, 9), ("hwc", 4, 7), ("hwc", 5, 9), ("hwc", 4, 5), ("hwc", 6, 3), ("chw", 4, 9), ("chw", 4, 7), ("chw", 8, 5)])
def test_slot_table_spreads_every_tile_over_all_bank_groups(layout, C, R):
    """The two-role pipeline's slot table (ppg_policy_pack: PPG_POLICY_PACK_SLOTS, device-free): every position of a sub-group exactly
    once; and where the table exists (the reference's 7x7 and 9x9 windows among them) slot n's cell of the padded image -- 16-byte cell
    sample * stride + (y + 1) * (W + 1) + x + 1 -- is congruent to n modulo 16, i.e. the sixteen lanes of a ds_read_b128 group (lanes
    0-3, 12-15, 20-27 / 4-11, 16-19, 28-31 of a half) read sixteen different bank groups and the eight lanes of a store group eight."""
    import __graft_entry__ as g
    lib = _abi.bind(ctypes.CDLL(g.build_hip()))
    spec = _abi.PpgPolicySpec()
    spec.obs_channels, spec.obs_range, spec.n_actions = C, R, 9
    spec.layout = _abi.POLICY_LAYOUT_HWC if layout == "hwc" else _abi.POLICY_LAYOUT_CHW
    spec.flatten, spec.n_conv, spec.n_fc = _abi.POLICY_FLATTEN_NHWC, 3, 1
    IH, IW, CIN = (C, R, R) if layout == "hwc" else (R, R, C)
    keep = []
    cin = CIN
    for l, co in enumerate((16, 32, 64)):
        w, b = np.zeros((co, cin, 3, 3), np.float32), np.zeros(co, np.float32)
        keep += [w, b]
        spec.conv_out[l], spec.conv_w[l], spec.conv_b[l] = co, w.ctypes.data, b.ctypes.data
        cin = co
    hw, hb = np.zeros((9, IH * IW * 64), np.float32), np.zeros(9, np.float32)
    spec.fc_out[0], spec.fc_w[0], spec.fc_b[0] = 9, hw.ctypes.data, hb.ctypes.data
    out = (ctypes.c_int32 * 12)()
    assert lib.ppg_policy_describe(ctypes.byref(spec), out, 12) == 0
    if out[0] != 3:
        pytest.skip("not a network of the two-role pipeline")
    st, stride_cells = out[1], out[5] // 16
    n = ctypes.c_uint64(0)
    assert lib.ppg_policy_pack(ctypes.byref(spec), _abi.POLICY_PACK_SLOTS, None, 0, ctypes.byref(n)) == 0
    tab = np.zeros(n.value, dtype=np.uint16)
    assert lib.ppg_policy_pack(ctypes.byref(spec), _abi.POLICY_PACK_SLOTS, tab.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n)) == 0
    P = IH * IW
    assert len(tab) % 32 == 0 and len(tab) >= st * P and len(tab) < st * P + 32
    used = tab[tab != 0xFFFF]
    assert sorted(used.tolist()) == sorted((s << 8 | p) for s in range(st) for p in range(P))     # every position exactly once
    s_, p_ = used >> 8, used & 255
    cell = s_.astype(np.int64) * stride_cells + (p_ // IW + 1) * (IW + 1) + p_ % IW + 1
    spread = np.array_equal(cell % 16, np.nonzero(tab != 0xFFFF)[0] % 16)
    if (layout, R) in (("hwc", 9), ("hwc", 7)):
        assert spread                                   # the reference's networks get conflict-free tiles
    if spread:
        for t in range(len(tab) // 32):                 # restated per lane group of one 32-lane half
            for group in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
                c = [int(x) for k in group for x in [tab[32 * t + k]] if x != 0xFFFF]
                cells = [(v >> 8) * stride_cells + ((v & 255) // IW + 1) * (IW + 1) + (v & 255) % IW + 1 for v in c]
                assert len({v % 16 for v in cells}) == len(cells)
            for w in range(4):
                c = [int(x) for x in tab[32 * t + 8 * w:32 * t + 8 * w + 8] if x != 0xFFFF]
                cells = [(v >> 8) * stride_cells + ((v & 255) // IW + 1) * (IW + 1) + (v & 255) % IW + 1 for v in c]
                assert len({v % 8 for v in cells}) == len(cells)
    else:                                               # the plain order
        assert np.array_equal(used, np.array([(s << 8 | p) for s in range(st) for p in range(P)], dtype=np.uint16))


@pytest.mark.gpu
def test_a_step_replayed_from_a_hip_graph_equals_the_launches_it_was_captured_from():
    """GraphedPolicyStep: policy launch + ppg_step + the Philox key's increment captured once, replayed 40 times -- the envs end up in
    exactly the state 40 explicit (act, step) pairs with keys seed, seed + 1, ... leave them in (row tables and observations bit for
    bit); the key word has advanced by one per replay."""
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.policy import FusedPolicy, GraphedPolicyStep
    nets = make_nets(seed=61)
    fused = FusedPolicy(nets[0], nets[1])
    envs = [BatchedPredPreyGrass(dict(config_env), batch_size=600, device="cuda:0", obs_dtype=torch.bfloat16, seed=21) for _ in range(2)]
    for e in envs:
        e.reset()
        for _ in range(30):
            e.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    a, b = envs
    for name in ("row_xy", "row_energy", "env_state"):
        assert torch.equal(getattr(a, name), getattr(b, name))
    n = 40
    loop = GraphedPolicyStep(fused, a, seed=1000)        # (its warm-up pass is step 0 with key 1000, the capture pass is not executed)
    loop.replay(n - 1)
    for t in range(n):
        fused.act(b, sample=True, seed=1000 + t)
        b.step(b.actions, auto_reset=True)
    torch.cuda.synchronize()
    assert int(loop.seed.item()) == 1000 + n
    for name in ("row_xy", "row_energy", "row_id", "row_flags", "row_reward", "row_cumrew", "env_state", "grass_energy", "obs_pred", "obs_prey", "actions"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert (a.env_state[:, _abi.ENV_STATUS] & _abi.STATUS_BAD_ACTION == 0).all() and (a.env_state[:, _abi.ENV_CALLS] == 30 + n).all()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["random_init_base_env", "rllib_checkpoint_walls_env"])
def test_report_greedy_agreement_with_fp32_inference_over_a_closed_loop_rollout(which):
    """A REPORT, not a tolerance (the figures are quoted in INTEGRATION.md): a population of envs is rolled out
    for 1000 steps with the kernels' own sampled actions closing the loop (bfloat16 MFMA operands, `tune_ppo_*.py`'s rollout
    setting), and on every 20th step the greedy action of the kernels is compared with the argmax of the float32 PyTorch module
    (RLlib's inference arithmetic) on the same observation rows -- the observation distribution the policy itself produces,
    not i.i.d. windows.  Asserted: the only sanity bounds the other tests already state (> 0.95)."""
    from predpreygrass_amd.policy import FusedPolicy
    if which == "random_init_base_env":
        from predpreygrass_amd.batched import BatchedPredPreyGrass
        nets = make_nets(seed=11)
        env = BatchedPredPreyGrass(dict(config_env), batch_size=256, device="cuda:0", obs_dtype=torch.float64, seed=4)
    else:
        from predpreygrass_amd.policy import load_rllib_state_dict
        from predpreygrass_amd.red_queen import BatchedRedQueen
        from predpreygrass_amd.walls_occlusion import config_env_zigzag_walls
        actor = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(FIXTURE, "type_1_predator_actor.npz")).items()}
        net = load_rllib_state_dict(actor)
        nets = (net, net)
        cfg = {**config_env_zigzag_walls, "predator_obs_range": 9, "prey_obs_range": 9}
        env = BatchedRedQueen(cfg, batch_size=256, device="cuda:0", walls=True, obs_dtype=torch.float32, seed=4)
        env.set_walls(cfg["manual_wall_positions"])
    fused = FusedPolicy(nets[0], nets[1])
    refs = [n.to("cuda:0") for n in nets]
    env.reset()
    agree = total = 0
    margins = []     # float32 top-1 minus top-2 logit where the two greedy actions differ
    scale = 0.0
    for t in range(1000):
        if t % 20 == 0:
            lg = fused.act(env, want_logits=True)
            torch.cuda.synchronize()
            mp, mq = rows_in_use(env)
            for sp, m in enumerate((mp, mq)):
                obs = (env.obs_prey if sp else env.obs_pred)[m]
                if obs.shape[0] == 0:
                    continue
                with torch.no_grad():
                    ref = refs[sp](obs)
                got = lg[sp][: obs.shape[0]]
                same = got.argmax(1) == ref.argmax(1)
                agree += int(same.sum()); total += int(same.numel())
                top2 = ref.topk(2, dim=1).values
                margins.append((top2[:, 0] - top2[:, 1])[~same].float().cpu())
                scale = max(scale, float(ref.abs().max()))
        fused.act(env, sample=True, seed=t)
        env.step(env.actions, auto_reset=True)
    torch.cuda.synchronize()
    mg = torch.cat(margins) if margins else torch.zeros(0)
    rate = agree / max(total, 1)
    print(f"\n[policy precision] {which}: greedy agreement bf16-MFMA vs float32 module {rate:.5f} over {total} observations of a 1000-step "
          f"closed-loop rollout (256 envs, every 20th step); {total - agree} differ, float32 margin there: median "
          f"{float(mg.median()) if len(mg) else 0.0:.2e}, max {float(mg.max()) if len(mg) else 0.0:.2e} (largest |logit| {scale:.2f})")
    assert total > 20000 and rate > 0.95
