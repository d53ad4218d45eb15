"""The environment configuration dict: same keys and defaults as the reference's
predpreygrass/non_evolutionary/base_environment/config_env.py:1-38, and the in-code
fallbacks the reference's __init__ applies to a partial dict
(predpreygrass_rllib_env.py:20-61)."""
from __future__ import annotations

# config_env.py:1-38
config_env = {
    "max_steps": 1000,
    "grid_size": 25,
    "num_obs_channels": 4,
    "predator_obs_range": 7,
    "prey_obs_range": 9,
    "reward_predator_catch_prey": 0.0,
    "reward_prey_eat_grass": 0.0,
    "reward_predator_step": 0.0,
    "reward_prey_step": 0.0,
    "penalty_prey_caught": 0.0,
    "reproduction_reward_predator": 10.0,
    "reproduction_reward_prey": 10.0,
    "energy_loss_per_step_predator": 0.15,
    "energy_loss_per_step_prey": 0.05,
    "predator_creation_energy_threshold": 12.0,
    "prey_creation_energy_threshold": 8.0,
    "n_possible_predators": 2000,
    "n_possible_prey": 2000,
    "n_initial_active_predator": 6,
    "n_initial_active_prey": 8,
    "initial_energy_predator": 5.0,
    "initial_energy_prey": 3.0,
    "initial_num_grass": 100,
    "initial_energy_grass": 2.0,
    "energy_gain_per_step_grass": 0.04,
    "verbose_engagement": False,
    "verbose_movement": False,
    "verbose_spawning": False,
}

# `config.get(key, default)` fallbacks in predpreygrass_rllib_env.py:22-61 (they differ from config_env)
_IN_CODE_DEFAULTS = {
    "max_steps": 10000,
    "grid_size": 10,
    "num_obs_channels": 4,
    "predator_obs_range": 7,
    "prey_obs_range": 5,
    "reward_predator_catch_prey": 0.0,
    "reward_prey_eat_grass": 0.0,
    "reward_predator_step": 0.0,
    "reward_prey_step": 0.0,
    "penalty_prey_caught": 0.0,
    "reproduction_reward_predator": 10.0,
    "reproduction_reward_prey": 10.0,
    "energy_loss_per_step_predator": 0.15,
    "energy_loss_per_step_prey": 0.05,
    "predator_creation_energy_threshold": 12.0,
    "prey_creation_energy_threshold": 8.0,
    "n_possible_predators": 50,
    "n_possible_prey": 50,
    "n_initial_active_predator": 6,
    "n_initial_active_prey": 8,
    "initial_energy_predator": 5.0,
    "initial_energy_prey": 3.0,
    "initial_num_grass": 25,
    "initial_energy_grass": 2.0,
    "energy_gain_per_step_grass": 0.2,
    "verbose_engagement": False,
    "verbose_movement": False,
    "verbose_spawning": False,
}


def resolve_config(config: dict | None) -> dict:
    """`config = config or config_env` (predpreygrass_rllib_env.py:20), then per-key
    `config.get(key, in_code_default)` (predpreygrass_rllib_env.py:22-61)."""
    base = config or config_env  # a falsy dict ({}) selects the defaults, as in the reference
    out = dict(_IN_CODE_DEFAULTS)
    out.update({k: v for k, v in base.items()})
    if int(out["num_obs_channels"]) != 4:
        raise ValueError("num_obs_channels must be 4 (border, predator, prey, grass)")
    return out
