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
    """(4,R,R) -> conv3x3 16/32/64 "same" + ReLU -> flatten (channel-major) -> 256 -> 256 -> n_actions logits."""

    def __init__(self, obs_range: int, n_actions: int = 9):
        super().__init__()
        self.obs_range, self.n_actions = int(obs_range), int(n_actions)
        self.conv = torch.nn.ModuleList([torch.nn.Conv2d(4, 16, 3, padding=1), torch.nn.Conv2d(16, 32, 3, padding=1),
                                         torch.nn.Conv2d(32, 64, 3, padding=1)])
        self.fc = torch.nn.ModuleList([torch.nn.Linear(64 * self.obs_range ** 2, 256), torch.nn.Linear(256, 256),
                                       torch.nn.Linear(256, self.n_actions)])

    def forward(self, obs):
        x = obs.to(torch.float32)
        for c in self.conv:
            x = torch.relu(c(x))
        x = x.flatten(1)
        x = torch.relu(self.fc[0](x))
        x = torch.relu(self.fc[1](x))
        return self.fc[2](x)


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
        rc = self._lib.ppg_policy_create(self.device.index, net.obs_range, net.n_actions, C.byref(w), C.byref(h))
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
            for t, net in enumerate(self.nets):
                if net is not None:
                    cap = sum(e.batch_size * (e.prey_capacity if t else e.pred_capacity) for e in envs)
                    lg[t] = torch.zeros((cap, net.n_actions), dtype=torch.float32, device=self.device)
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
