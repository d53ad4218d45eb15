"""Policy inference next to the env: the observations never leave the GPU.

The reference trains one PPO policy per species with RLlib's `DefaultPPOTorchRLModule`
(base_environment/tune_ppo_base_environment.py:106-141: conv_filters [[16,[3,3],1],[32,[3,3],1],[64,[3,3],1]],
fcnet_hiddens [256,256], ReLU).  WHAT RLLIB BUILDS FROM THAT is pinned by the checkpoint the reference tree itself holds
(.../shared_prey/experiments/PPO_v_APPO/.../checkpoint_000099/learner_group/learner/rl_module/type_1_predator/module_state.pkl,
ray 2.52.1; tests/golden/rllib_checkpoint/): the (C,R,R) Box is read channels-last -- a C x R image with R channels --, the CNN
encoder is one [ZeroPad2d, Conv2d 3x3, ReLU] per conv_filters entry (`encoder.actor_encoder.net.0.cnn.{1,4,7,..}`), its output is
permuted back to channels-last and flattened, `fcnet_hiddens` is IGNORED for image observations, and the policy head is a single
`Linear(flat, n_actions)` (`pi.net.mlp.0`).  `PolicyNet` is that network as a plain float32 `torch.nn.Module` WITH RLLIB'S
PARAMETER NAMES (a checkpoint's actor entries load with `load_state_dict(strict=True)`) -- the container a user loads weights
into and the reference the parity tests compare against.  `FusedPolicy` hands the weights of two such modules to libppg_hip.so
(`ppg_policy_create_spec`: repacked into MFMA fragment order, bf16) and `act()` launches `ppg_policy_act`: hand-written
matrix-core kernels that read `obs_pred` / `obs_prey` where `ppg_step` wrote them and write one int8 action per agent row into
the env's action tensor.  bf16 operands, fp32 accumulation; tolerance in tests/test_policy.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi


class _Named(torch.nn.Module):
    """An empty module: only there to give the parameters RLlib's key names."""


class _TorchCNN(torch.nn.Module):
    """ray.rllib.core.models.torch.primitives.TorchCNN as the checkpoint's keys describe it: `cnn` = Sequential of
    (ZeroPad2d, Conv2d, ReLU) per layer -- the convolutions sit at cnn.1, cnn.4, cnn.7, ..."""

    def __init__(self, cin, channels):
        super().__init__()
        layers = []
        for c in channels:
            layers += [torch.nn.ZeroPad2d(1), torch.nn.Conv2d(cin, c, 3, 1), torch.nn.ReLU()]
            cin = c
        self.cnn = torch.nn.Sequential(*layers)


class PolicyNet(torch.nn.Module):
    """L x [zero-pad 1, conv3x3 stride 1, ReLU] -> flatten -> [Linear, ReLU] per hidden head layer -> Linear -> n_actions logits,
    on a (C,R,R) observation read as an image in one of two ways:

    layout "hwc"  channels-last, RLlib's reading of a 3-D Box: a C x R image with R channels (conv1 weight [16,R,3,3]); the
                  encoder output is flattened [row][column][channel] (flatten "nhwc").  THE DEFAULT: with conv_channels
                  (16,32,64) and no hidden head layer this is the module tune_ppo_base_environment.py:106-141 trains.
    layout "chw"  channel-first: an R x R image with C channels (conv1 weight [16,C,3,3]), flattened channel-major ("nchw") -- a
                  plain PyTorch network of one's own, not what RLlib builds.
    Parameter names are RLlib's: encoder.actor_encoder.net.0.cnn.{1+3l}.{weight,bias}, pi.net.mlp.{2l}.{weight,bias}.
    `forward` takes the observation rows as the env writes them, [N,C,R,R], in either layout; `.conv` / `.fc` list the layers."""

    def __init__(self, obs_range: int, n_actions: int = 9, layout: str = "hwc", obs_channels: int = 4,
                 conv_channels=(16, 32, 64), head_hiddens=(), flatten: str = None):
        super().__init__()
        if layout not in ("chw", "hwc"):
            raise ValueError("layout must be 'chw' or 'hwc'")
        flatten = flatten or ("nhwc" if layout == "hwc" else "nchw")
        if flatten not in ("nhwc", "nchw"):
            raise ValueError("flatten must be 'nhwc' or 'nchw'")
        self.obs_range, self.n_actions, self.layout, self.flatten = int(obs_range), int(n_actions), layout, flatten
        self.obs_channels = int(obs_channels)
        self.conv_channels, self.head_hiddens = tuple(int(c) for c in conv_channels), tuple(int(h) for h in head_hiddens)
        R, Cn = self.obs_range, self.obs_channels
        cin, positions = (Cn, R * R) if layout == "chw" else (R, Cn * R)
        self.encoder = _Named()
        self.encoder.actor_encoder = _Named()
        self.encoder.actor_encoder.net = torch.nn.Sequential(_TorchCNN(cin, self.conv_channels))
        dims = [self.conv_channels[-1] * positions, *self.head_hiddens, self.n_actions]
        mlp = []
        for l in range(len(dims) - 1):
            mlp.append(torch.nn.Linear(dims[l], dims[l + 1]))
            if l + 2 < len(dims):
                mlp.append(torch.nn.ReLU())
        self.pi = _Named()
        self.pi.net = _Named()
        self.pi.net.mlp = torch.nn.Sequential(*mlp)

    @property
    def conv(self):
        return [m for m in self.encoder.actor_encoder.net[0].cnn if isinstance(m, torch.nn.Conv2d)]

    @property
    def fc(self):
        return [m for m in self.pi.net.mlp if isinstance(m, torch.nn.Linear)]

    def forward(self, obs):
        x = obs.to(torch.float32)
        if self.layout == "hwc":
            x = x.permute(0, 3, 1, 2)   # [N, H=C, W=R, channels=R] -> NCHW, as TorchCNN.forward does with its channels-last input
        x = self.encoder.actor_encoder.net[0].cnn(x)
        if self.flatten == "nhwc":
            x = x.permute(0, 2, 3, 1)   # "permute back to channels_last", then nn.Flatten
        return self.pi.net.mlp(x.flatten(1))


def load_rllib_state_dict(state_dict, obs_range: int = None, obs_channels: int = None) -> PolicyNet:
    """A `PolicyNet` holding the policy (actor) network of an RLlib PPO RLModule state dict -- what
    `RLModule.from_checkpoint(...)` holds in evaluate_ppo_from_checkpoint_debug.py:129 and greedy actions are taken from at its
    lines 69-96 (`module.get_state()` / the unpickled `module_state.pkl` of a checkpoint; tensors or numpy arrays).

    Parameter discovery is by ROLE, because the key names depend on the RLlib version and on `vf_share_layers`:
      conv layers  = the 4-D weights (with their biases) whose key contains "encoder" and not "critic" / "vf", in the natural
                     (numeric) order of their keys
                     (encoder.actor_encoder.net.0.cnn.{1,4,7,...}.weight, or encoder.encoder... with a shared encoder);
      head layers  = the 2-D weights whose key starts with "pi." (pi.net.mlp.0.weight alone in what RLlib builds by default;
                     pi.net.mlp.{0,2,..} with `head_fcnet_hiddens`).
    The ARCHITECTURE IS TAKEN FROM THE SHAPES: 1-6 convolutions 3x3 (up to 16 / 32 / 64 / 64 .. channels), 0-2 hidden head layers
    (up to 256 features).  conv1's input channels and the head's input size decide the reading of the (C,R,R) Box: channels-last
    ("hwc", RLlib: conv1 takes R channels, flat = C * R * cout) or channel-first ("chw": conv1 takes C channels, flat = R * R *
    cout); where both fit, `obs_range` / `obs_channels` decide and channels-last wins otherwise.  Anything else raises ValueError
    naming the keys and shapes found."""
    def arr(v):
        return v.detach().cpu().to(torch.float32) if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v), dtype=torch.float32)
    sd = {k: arr(v) for k, v in state_dict.items() if hasattr(v, "shape")}
    listing = ", ".join(f"{k}{tuple(v.shape)}" for k, v in sd.items())

    def is_actor_encoder(k):
        kl = k.lower()
        return "encoder" in kl and "critic" not in kl and ".vf" not in kl and not kl.startswith("vf")
    conv_w = [k for k, v in sd.items() if v.dim() == 4 and is_actor_encoder(k)]
    if not conv_w:   # a bare state dict of the network itself
        conv_w = [k for k, v in sd.items() if v.dim() == 4]
    head_w = [k for k, v in sd.items() if v.dim() == 2 and k.lower().startswith("pi.")]
    if not head_w:
        head_w = [k for k, v in sd.items() if v.dim() == 2 and "critic" not in k.lower() and not k.lower().startswith("vf")]
    # layer order = the NUMERIC layer indices in the keys (cnn.1, cnn.4, cnn.7, cnn.10), not the dict's order: a state dict that has
    # been sorted or re-serialised lists cnn.1, cnn.10, cnn.4, cnn.7 (tests/golden/rllib_checkpoint/state_listing.json)
    def natural(k):
        import re
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", k)]
    conv_w.sort(key=natural)
    head_w.sort(key=natural)
    if not 1 <= len(conv_w) <= _abi.POLICY_MAX_CONV or not 1 <= len(head_w) <= _abi.POLICY_MAX_FC:
        raise ValueError(f"expected 1-{_abi.POLICY_MAX_CONV} convolution and 1-{_abi.POLICY_MAX_FC} linear weights of the policy "
                         f"network, found {len(conv_w)} / {len(head_w)}: {listing}")

    def bias_of(k):
        kb = k[: -len("weight")] + "bias" if k.endswith("weight") else None
        if kb is None or kb not in sd:
            raise ValueError(f"no bias next to {k}: {listing}")
        return sd[kb]
    cw, hw = [sd[k] for k in conv_w], [sd[k] for k in head_w]
    cin = int(cw[0].shape[1])
    chans, prev = [], cin
    for l, w in enumerate(cw):
        if tuple(w.shape[2:]) != (3, 3) or int(w.shape[1]) != prev:
            raise ValueError(f"convolutions {[tuple(w.shape) for w in cw]} are not 3x3 layers feeding each other "
                             "(tune_ppo_base_environment.py:112-116)")
        prev = int(w.shape[0])
        if prev > (16, 32, 64)[min(l, 2)]:
            raise ValueError(f"convolution {l + 1} has {prev} output channels: the kernels take up to 16 / 32 / 64 / 64 ...")
        chans.append(prev)
    flat, n_actions = int(hw[0].shape[1]), int(hw[-1].shape[0])
    hidden, prev = [], flat
    for l, w in enumerate(hw):
        if int(w.shape[1]) != prev:
            raise ValueError(f"policy head {[tuple(w.shape) for w in hw]} is not a chain of Linear layers")
        prev = int(w.shape[0])
        if l + 1 < len(hw):
            if prev > 256:
                raise ValueError(f"policy head {[tuple(w.shape) for w in hw]}: hidden layers of up to 256 features")
            hidden.append(prev)
    if hidden and len(cw) != 3:
        raise ValueError(f"a policy head with hidden layers needs exactly three convolutions, found {len(cw)}")
    if flat % chans[-1]:
        raise ValueError(f"the head takes {flat} features, not a multiple of the last convolution's {chans[-1]} channels")
    positions = flat // chans[-1]
    layouts = []   # (layout, R, C)
    if positions % cin == 0 and 1 <= positions // cin <= 8:
        layouts.append(("hwc", cin, positions // cin))
    r = int(round(positions ** 0.5))
    if r * r == positions and 1 <= cin <= 8:
        layouts.append(("chw", r, cin))
    if obs_range is not None:
        layouts = [t for t in layouts if t[1] == int(obs_range)]
    if obs_channels is not None:
        layouts = [t for t in layouts if t[2] == int(obs_channels)]
    if not layouts:
        raise ValueError(f"conv1 has {cin} input channels and the head takes {flat} = {positions} x {chans[-1]} features: neither "
                         f"C x R positions with R channels (channels-last) nor R x R positions with C channels (channel-first)"
                         + (f" for obs_range {obs_range}" if obs_range is not None else "")
                         + (f" for obs_channels {obs_channels}" if obs_channels is not None else ""))
    layout, R, Cn = layouts[0]
    net = PolicyNet(R, n_actions, layout, obs_channels=Cn, conv_channels=chans, head_hiddens=hidden)
    with torch.no_grad():
        for l, m in enumerate(net.conv):
            m.weight.copy_(cw[l]); m.bias.copy_(bias_of(conv_w[l]))
        for l, m in enumerate(net.fc):
            m.weight.copy_(hw[l]); m.bias.copy_(bias_of(head_w[l]))
    return net


class FusedPolicy:
    """The two species' networks on the matrix cores, next to the envs.

    pred_net / prey_net: `PolicyNet`s (any device; the float32 weights are copied once).  Either may be None: that species
    keeps whatever the action tensor holds."""

    def __init__(self, pred_net: PolicyNet = None, prey_net: PolicyNet = None, device="cuda:0"):
        self._lib = _abi.load_hip_library()
        if not hasattr(self._lib, "ppg_policy_create"):
            raise RuntimeError("libppg_hip.so has no ppg_policy_* entry points: rebuild with __graft_entry__.build()")
        if not torch.cuda.is_available():
            raise RuntimeError("predpreygrass_amd.policy needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._handles = [self._create(pred_net), self._create(prey_net)]
        self.nets = (pred_net, prey_net)

    def _create(self, net):
        if net is None:
            return None
        keep = []   # the host arrays must live until ppg_policy_create_spec returns

        def host(t):
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            keep.append(a)
            return a.ctypes.data
        convs, fcs = net.conv, net.fc
        if len(convs) > _abi.POLICY_MAX_CONV or len(fcs) > _abi.POLICY_MAX_FC:
            raise ValueError(f"{len(convs)} convolutions / {len(fcs)} linear layers: the kernels take up to "
                             f"{_abi.POLICY_MAX_CONV} / {_abi.POLICY_MAX_FC}")
        sp = _abi.PpgPolicySpec()
        sp.obs_channels, sp.obs_range, sp.n_actions = net.obs_channels, net.obs_range, net.n_actions
        sp.layout = _abi.POLICY_LAYOUT_HWC if net.layout == "hwc" else _abi.POLICY_LAYOUT_CHW
        sp.flatten = _abi.POLICY_FLATTEN_NHWC if net.flatten == "nhwc" else _abi.POLICY_FLATTEN_NCHW
        sp.n_conv, sp.n_fc = len(convs), len(fcs)
        for l, m in enumerate(convs):
            sp.conv_out[l], sp.conv_w[l], sp.conv_b[l] = m.out_channels, host(m.weight), host(m.bias)
        for l, m in enumerate(fcs):
            sp.fc_out[l], sp.fc_w[l], sp.fc_b[l] = m.out_features, host(m.weight), host(m.bias)
        h = C.c_void_p()
        rc = self._lib.ppg_policy_create_spec(self.device.index, C.byref(sp), C.byref(h))
        if rc != 0:
            msg = self._lib.ppg_policy_last_error(None).decode()
            if rc == -1:
                raise ValueError(msg)
            raise RuntimeError(f"ppg_policy_create_spec failed ({rc}): {msg}")
        return h

    def macs_per_observation(self, species: int) -> int:
        h = self._handles[species]
        return int(self._lib.ppg_policy_macs_per_observation(h)) if h else 0

    def act(self, envs, actions=None, sample=False, seed=0, want_logits=False, stream=None):
        """Actions for every row in use of `envs` (a BatchedPredPreyGrass or a list of sub-batches of one GPU), written into
        each env's `.actions` tensor (or the given list of int8 [B_k, S] tensors).  Stream-ordered like a step: launch it on
        a stream that is ordered behind the envs' last step.  want_logits: also return (logits_pred, logits_prey) float32
        [rows in use, n_actions] in env-major row order -- sized for the row capacity, the caller slices by the counts.
        seed: an int, or a one-element int64 tensor on the device -- the kernel then reads the Philox key from it when it RUNS
        (`PPG_POLICY_SEED_ON_DEVICE`): what a step captured into a HIP graph needs (`GraphedPolicyStep`)."""
        envs = list(envs) if isinstance(envs, (list, tuple)) else [envs]
        acts = [e.actions for e in envs] if actions is None else list(actions)
        for e, a in zip(envs, acts):
            if a.dtype != torch.int8 or tuple(a.shape) != (e.batch_size, e.S) or not a.is_contiguous() or a.device != e.device:
                raise ValueError("actions must be contiguous int8 tensors [B_k, S] on the envs' device")
        n = len(envs)
        handles = (C.c_void_p * n)(*[e._handle for e in envs])
        aptr = (C.c_void_p * n)(*[a.data_ptr() for a in acts])
        lg = [None, None]
        if want_logits:
            # the buffers are zero-filled on the stream the kernels run on (so the fill cannot race with their writes and the
            # caching allocator knows which stream uses the blocks); a raw hipStream_t handle: fill on the current stream and
            # let the host wait for it
            on = stream if isinstance(stream, torch.cuda.Stream) else torch.cuda.current_stream(self.device)
            with torch.cuda.stream(on):
                for t, net in enumerate(self.nets):
                    if net is not None:
                        cap = sum(e.batch_size * (e.prey_capacity if t else e.pred_capacity) for e in envs)
                        lg[t] = torch.zeros((cap, net.n_actions), dtype=torch.float32, device=self.device)
            if stream is not None and not isinstance(stream, torch.cuda.Stream):
                on.synchronize()
        flags = _abi.POLICY_SAMPLE if sample else _abi.POLICY_ARGMAX
        if isinstance(seed, torch.Tensor):
            if seed.dtype != torch.int64 or seed.numel() != 1 or seed.device != self.device:
                raise ValueError("a device-resident seed is a one-element int64 tensor on the policy's device")
            flags |= _abi.POLICY_SEED_ON_DEVICE
            seed = seed.data_ptr()
        rc = self._lib.ppg_policy_act(self._handles[0], self._handles[1], handles, n, aptr,
                                      flags, int(seed) & (2 ** 64 - 1),
                                      C.c_void_p(lg[0].data_ptr()) if lg[0] is not None else None,
                                      C.c_void_p(lg[1].data_ptr()) if lg[1] is not None else None, envs[0]._stream(stream))
        if rc != 0:
            msg = self._lib.ppg_policy_last_error(None).decode()
            if rc == -1:
                raise ValueError(msg)
            raise RuntimeError(f"ppg_policy_act failed ({rc}): {msg}")
        return (lg[0], lg[1]) if want_logits else None

    def close(self):
        for h in self._handles:
            if h:
                self._lib.ppg_policy_destroy(h)
        self._handles = [None, None]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GraphedPolicyStep:
    """One closed-loop step -- the policy on every observation row, `ppg_step` with its actions, auto-reset -- captured ONCE into a HIP
    graph and replayed: two kernel launches and an increment of the Philox key per replay, no launch arguments built on the host, no
    gaps between the kernels.  (`tune_ppo_base_environment.py:106-141` + `BASE:219-473` as one device-side loop body.)

        loop = GraphedPolicyStep(fused, env, seed=0); loop.replay(1000)

    The key of step t is seed + t, kept in a device word the captured increment advances; results equal `fused.act(env, sample=True,
    seed=seed + t); env.step(env.actions, auto_reset=True)` step by step (tests/test_policy.py)."""

    def __init__(self, fused: "FusedPolicy", env, seed: int = 0, sample: bool = True, auto_reset: bool = True):
        self.fused, self.env = fused, env
        dev = env.device
        self.seed = torch.full((1,), int(seed), dtype=torch.int64, device=dev)
        self._one = torch.ones((1,), dtype=torch.int64, device=dev)

        def body():
            fused.act(env, sample=sample, seed=self.seed)
            env.step(env.actions, auto_reset=auto_reset)
            self.seed.add_(self._one)
        # (one uncaptured pass on a side stream first: lazy allocations and attribute calls of the library must not fall into the capture)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()

    def replay(self, n: int = 1):
        for _ in range(n):
            self.graph.replay()
        return self
