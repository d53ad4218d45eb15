"""Fixture generator (run in the build container only; reads /root/reference, which does not exist on the GPU box).

The reference tree holds ONE real RLlib checkpoint of its PPO setup (ray 2.52.1, DefaultPPOTorchRLModule, PPOCatalog):
  predpreygrass/non_evolutionary/project_cooperation/shared_prey/experiments/PPO_v_APPO/
    GRID_30_PRED_OBS_RANGE_9_INITS_15_INIT_PREY_ENERGY_2_5/PPO/PPO_PredPreyGrass_5c0be_00000_0_2025-12-19_23-59-23/
    checkpoint_000099/learner_group/learner/rl_module/{type_1_predator,type_1_prey}/{module_state.pkl,class_and_ctor_args.pkl,metadata.json}
It pins what RLlib builds from the model_config the reference's tune scripts pass (conv_filters + fcnet_hiddens [256, 256]) for a
3-D Box observation: which parameters exist, their names and shapes.  This script copies DATA only:
  state_listing.json             every key -> shape / dtype of both modules' state, the checkpoint metadata, and what the pickled
                                 constructor arguments say (observation Box shape, conv_filters, fcnet_hiddens)
  type_1_predator_actor.npz      the float32 values of the predator module's actor entries (encoder.actor_encoder.*, pi.*)
Usage: python tests/golden/rllib_checkpoint/make_fixture.py
"""
import json
import os
import pickle
import pickletools
import warnings

import numpy as np

REF = ("/root/reference/predpreygrass/non_evolutionary/project_cooperation/shared_prey/experiments/PPO_v_APPO/"
       "GRID_30_PRED_OBS_RANGE_9_INITS_15_INIT_PREY_ENERGY_2_5/PPO/PPO_PredPreyGrass_5c0be_00000_0_2025-12-19_23-59-23/"
       "checkpoint_000099/learner_group/learner/rl_module")
HERE = os.path.dirname(os.path.abspath(__file__))


def ctor_facts(path):
    """class_and_ctor_args.pkl cannot be unpickled without ray / gymnasium: read the opcode stream instead."""
    ops = [(op.name, arg) for op, arg, _ in pickletools.genops(open(path, "rb"))]
    strings = [a for n, a in ops if isinstance(a, str)]
    facts = {"module_class": next(s for s in strings if s.endswith("RLModule")),
             "catalog_class": next((s for s in strings if s.endswith("Catalog")), None)}
    i = next(k for k, (n, a) in enumerate(ops) if a == "_shape")
    facts["observation_box_shape"] = [a for n, a in ops[i + 1:i + 8] if n.startswith("BININT")][:3]
    i = next(k for k, (n, a) in enumerate(ops) if a == "conv_filters")
    j = next(k for k, (n, a) in enumerate(ops) if a == "fcnet_hiddens")
    ints = [a for n, a in ops[i + 1:j] if n.startswith("BININT")]
    facts["conv_filters"] = [[ints[k], [ints[k + 1], ints[k + 2]], ints[k + 3]] for k in range(0, len(ints), 4)]
    k = next(k for k, (n, a) in enumerate(ops) if a == "fcnet_activation")
    facts["fcnet_hiddens"] = [a for n, a in ops[j + 1:k] if n.startswith("BININT")]
    facts["fcnet_activation"] = next(a for n, a in ops[k + 1:] if isinstance(a, str))
    facts["model_config_keys"] = [s for s in strings if s in ("vf_share_layers", "conv_filters", "fcnet_hiddens", "fcnet_activation",
                                                              "head_fcnet_hiddens", "conv_activation")]
    return facts


def main():
    listing = {"source": REF.replace("/root/reference/", ""), "metadata": json.load(open(os.path.join(REF, "metadata.json")))}
    for sp in ("type_1_predator", "type_1_prey"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            state = pickle.load(open(os.path.join(REF, sp, "module_state.pkl"), "rb"))
        listing[sp] = {"state": {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in state.items()},
                       "ctor": ctor_facts(os.path.join(REF, sp, "class_and_ctor_args.pkl"))}
        if sp == "type_1_predator":
            actor = {k: np.asarray(v, dtype=np.float32) for k, v in state.items()
                     if (k.startswith("encoder.actor_encoder.") or k.startswith("pi.")) and v.ndim >= 1 and "log_std" not in k}
            np.savez_compressed(os.path.join(HERE, sp + "_actor.npz"), **actor)
    json.dump(listing, open(os.path.join(HERE, "state_listing.json"), "w"), indent=1, sort_keys=True)
    print("written:", os.listdir(HERE))


if __name__ == "__main__":
    main()
