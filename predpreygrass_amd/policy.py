"""Policy inference next to the env: the observations never leave the GPU.

The reference trains one PPO policy per species with RLlib
(base_environment/tune_ppo_base_environment.py:106-141: conv_filters [[16,[3,3],1],[32,[3,3],1],[64,[3,3],1]],
fcnet_hiddens [256,256], ReLU).  `PolicyNet` is that architecture as a plain float32 `torch.nn.Module` -- the container a
user trains / loads weights into and the reference the parity tests compare against.  `FusedPolicy` hands the weights of two
such modules to libppg_hip.so (`ppg_policy_create`: repacked into MFMA fragment order, bf16) and `act()` launches
`ppg_policy_act`: hand-written matrix-core kernels that read `obs_pred` / `obs_prey` where `ppg_step` wrote them and write
one int8 action per agent row into the env's action tensor.  bf16 operands, fp32 accumulation; tolerance in
tests/test_policy.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi


class PolicyNet(torch.nn.Module):
    """conv3x3 16/32/64 "same" + ReLU -> flatten (channel-major) -> 256 -> 256 -> n_actions logits, on a (4,R,R) observation read
    as an image in one of two ways:

    layout "chw"  channel-first: an R x R image with 4 channels (conv1 weight [16,4,3,3], 64 R^2 flat features);
    layout "hwc"  channels-last: a 4 x R image with R channels (conv1 weight [16,R,3,3], 64*4*R flat features).  RLlib's CNN
                  encoder takes 3-D Box spaces as [H, W, C], so this is the reading a module trained by
                  tune_ppo_base_environment.py:106-141 on Box(0, 100, (4,R,R)) has; its [256, 256] MLP is the policy head.
    `forward` takes the observation rows as the env writes them, [N,4,R,R], in either layout."""

    def __init__(self, obs_range: int, n_actions: int = 9, layout: str = "chw"):
        super().__init__()
        if layout not in ("chw", "hwc"):
            raise ValueError("layout must be 'chw' or 'hwc'")
        self.obs_range, self.n_actions, self.layout = int(obs_range), int(n_actions), layout
        R = self.obs_range
        cin, positions = (4, R * R) if layout == "chw" else (R, 4 * R)
        self.conv = torch.nn.ModuleList([torch.nn.Conv2d(cin, 16, 3, padding=1), torch.nn.Conv2d(16, 32, 3, padding=1),
                                         torch.nn.Conv2d(32, 64, 3, padding=1)])
        self.fc = torch.nn.ModuleList([torch.nn.Linear(64 * positions, 256), torch.nn.Linear(256, 256),
                                       torch.nn.Linear(256, self.n_actions)])

    def forward(self, obs):
        x = obs.to(torch.float32)
        if self.layout == "hwc":
            x = x.permute(0, 3, 1, 2)   # [N, H=4, W=R, C=R] -> NCHW, as RLlib's TorchCNN does with its channels-last input
        for c in self.conv:
            x = torch.relu(c(x))
        x = x.flatten(1)
        x = torch.relu(self.fc[0](x))
        x = torch.relu(self.fc[1](x))
        return self.fc[2](x)


def load_rllib_state_dict(state_dict, obs_range: int = None) -> PolicyNet:
    """A `PolicyNet` holding the policy (actor) network of an RLlib PPO RLModule state dict -- what
    `RLModule.from_checkpoint(...)` holds in evaluate_ppo_from_checkpoint_debug.py:129 and greedy actions are taken from at its
    lines 69-96 (`module.get_state()` / `module.state_dict()`, tensors or numpy arrays).

    Parameter discovery is by ROLE, because the key names depend on the RLlib version and on `vf_share_layers`:
      conv layers  = the 4-D weights (with their biases) whose key contains "encoder" and not "critic" / "vf", in key order
                     (e.g. encoder.actor_encoder.net.0.cnn.{1,4,7}.weight, or encoder.encoder... with a shared encoder);
      head layers  = the 2-D weights whose key starts with "pi." (e.g. pi.net.mlp.{0,2,4}.weight).
    Shapes are checked strictly: three 3x3 convolutions C -> 16 -> 32 -> 64, then Linear(flat -> 256), (256 -> 256),
    (256 -> n_actions).  conv1's input channels decide the layout: 4 = channel-first ("chw"), R = channels-last ("hwc", RLlib's
    own reading of a (4,R,R) Box; R is then also read off the weight).  With R == 4 the two are indistinguishable by shape and
    `obs_range` plus the flat size decide.  Anything else raises ValueError naming the keys and shapes found."""
    def arr(v):
        return v.detach().cpu().to(torch.float32) if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v), dtype=torch.float32)
    sd = {k: arr(v) for k, v in state_dict.items() if hasattr(v, "shape")}
    listing = ", ".join(f"{k}{tuple(v.shape)}" for k, v in sd.items())

    def is_actor_encoder(k):
        kl = k.lower()
        return "encoder" in kl and "critic" not in kl and ".vf" not in kl and not kl.startswith("vf")
    conv_w = [k for k, v in sd.items() if v.dim() == 4 and is_actor_encoder(k)]
    if not conv_w:   # a bare state dict of the network itself (e.g. a PolicyNet's own)
        conv_w = [k for k, v in sd.items() if v.dim() == 4]
    head_w = [k for k, v in sd.items() if v.dim() == 2 and k.lower().startswith("pi.")]
    if not head_w:
        head_w = [k for k, v in sd.items() if v.dim() == 2 and "critic" not in k.lower() and not k.lower().startswith("vf")]
    if len(conv_w) != 3 or len(head_w) != 3:
        raise ValueError(f"expected 3 convolution and 3 linear weights of the policy network, found {len(conv_w)} / {len(head_w)}: {listing}")

    def bias_of(k):
        kb = k[: -len("weight")] + "bias" if k.endswith("weight") else None
        if kb is None or kb not in sd:
            raise ValueError(f"no bias next to {k}: {listing}")
        return sd[kb]
    cw, hw = [sd[k] for k in conv_w], [sd[k] for k in head_w]
    cin = int(cw[0].shape[1])
    want_conv = [(16, cin, 3, 3), (32, 16, 3, 3), (64, 32, 3, 3)]
    if [tuple(w.shape) for w in cw] != want_conv:
        raise ValueError(f"convolutions {[tuple(w.shape) for w in cw]} are not 3x3 {cin} -> 16 -> 32 -> 64 "
                         "(tune_ppo_base_environment.py:112-116)")
    flat, n_actions = int(hw[0].shape[1]), int(hw[2].shape[0])
    if tuple(hw[0].shape) != (256, flat) or tuple(hw[1].shape) != (256, 256) or tuple(hw[2].shape) != (n_actions, 256):
        raise ValueError(f"policy head {[tuple(w.shape) for w in hw]} is not Linear(flat, 256), (256, 256), (256, n_actions)")
    layouts = []
    if cin == 4 and flat % 64 == 0 and int(round((flat // 64) ** 0.5)) ** 2 == flat // 64:
        layouts.append(("chw", int(round((flat // 64) ** 0.5))))
    if flat == 64 * 4 * cin:
        layouts.append(("hwc", cin))
    if obs_range is not None:
        layouts = [(l, r) for l, r in layouts if r == int(obs_range)]
    if not layouts:
        raise ValueError(f"conv1 has {cin} input channels and the head takes {flat} features: neither 64 R^2 with 4 channels "
                         f"(channel-first) nor 64*4*R with R channels (channels-last)"
                         + (f" for obs_range {obs_range}" if obs_range is not None else ""))
    layout, R = layouts[-1] if cin != 4 else layouts[0]
    net = PolicyNet(R, n_actions, layout)
    with torch.no_grad():
        for l in range(3):
            net.conv[l].weight.copy_(cw[l]); net.conv[l].bias.copy_(bias_of(conv_w[l]))
            net.fc[l].weight.copy_(hw[l]); net.fc[l].bias.copy_(bias_of(head_w[l]))
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
        keep = []   # the host arrays must live until ppg_policy_create returns

        def host(t):
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            keep.append(a)
            return a.ctypes.data
        w = _abi.PpgPolicyWeights()
        for l in range(3):
            w.conv_w[l], w.conv_b[l] = host(net.conv[l].weight), host(net.conv[l].bias)
            w.fc_w[l], w.fc_b[l] = host(net.fc[l].weight), host(net.fc[l].bias)
        h = C.c_void_p()
        layout = _abi.POLICY_LAYOUT_HWC if getattr(net, "layout", "chw") == "hwc" else _abi.POLICY_LAYOUT_CHW
        rc = self._lib.ppg_policy_create_layout(self.device.index, net.obs_range, net.n_actions, layout, C.byref(w), C.byref(h))
        if rc != 0:
            msg = self._lib.ppg_policy_last_error(None).decode()
            if rc == -1:
                raise ValueError(msg)
            raise RuntimeError(f"ppg_policy_create failed ({rc}): {msg}")
        return h

    def macs_per_observation(self, species: int) -> int:
        h = self._handles[species]
        return int(self._lib.ppg_policy_macs_per_observation(h)) if h else 0

    def act(self, envs, actions=None, sample=False, seed=0, want_logits=False, stream=None):
        """Actions for every row in use of `envs` (a BatchedPredPreyGrass or a list of sub-batches of one GPU), written into
        each env's `.actions` tensor (or the given list of int8 [B_k, S] tensors).  Stream-ordered like a step: launch it on
        a stream that is ordered behind the envs' last step.  want_logits: also return (logits_pred, logits_prey) float32
        [rows in use, n_actions] in env-major row order -- sized for the row capacity, the caller slices by the counts."""
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
        rc = self._lib.ppg_policy_act(self._handles[0], self._handles[1], handles, n, aptr,
                                      _abi.POLICY_SAMPLE if sample else _abi.POLICY_ARGMAX, int(seed) & (2 ** 64 - 1),
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
