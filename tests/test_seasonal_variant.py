"""Known answers for the seasonal grass-regrowth variant (a square wave on energy_gain_per_step_grass keyed on current_step:
base_environment_seasonal/predpreygrass_rllib_env.py:224-234,268-271 of the reference), run on the emulated kernel.

Table-driven: every row states a season configuration and what the square wave must do.  Bit-exact parity with that variant's
step() is the job of the seasonal_* golden cases (tests/test_oracle_golden.py, tests/test_emulated_kernel.py); these rows pin the
arithmetic of the wave itself and of the regrowth it scales, with numbers derived here from the definition -- gain(step) =
base_gain * (high if (step // length) is even else low), capped at initial_energy_grass."""
import pytest

from predpreygrass_amd.config import config_env as base_config
from predpreygrass_amd.env import PredPreyGrass
from tests.emu_backend import library

GAIN = base_config["energy_gain_per_step_grass"]
CAP = base_config["initial_energy_grass"]

# (season_length_steps, high, low, {step: expected multiplier})
WAVE = [
    (5, 1.5, 0.5, {0: 1.5, 4: 1.5, 5: 0.5, 9: 0.5, 10: 1.5, 14: 1.5, 15: 0.5, 999: 0.5}),
    (1, 2.0, 0.25, {0: 2.0, 1: 0.25, 2: 2.0, 7: 0.25}),
    (40, 1.5, 0.5, {39: 1.5, 40: 0.5, 79: 0.5, 80: 1.5}),        # the variant's own config (its config_env.py:35-40)
    (3, 1.0, 1.0, {s: 1.0 for s in range(12)}),                    # both multipliers 1: the base environment
    (0, 7.0, 9.0, {s: 1.0 for s in (0, 1, 45, 1000)}),             # no cycle configured: the base environment
]


def make(**season):
    return PredPreyGrass({**base_config, **season}, _library=library())


@pytest.mark.parametrize("length,high,low,expect", WAVE)
def test_square_wave(length, high, low, expect):
    env = make(season_length_steps=length, season_high_multiplier=high, season_low_multiplier=low)
    for step, want in expect.items():
        env.current_step = step
        assert env._current_season_multiplier() == want, (length, step)


def test_base_config_is_flat():
    env = PredPreyGrass(base_config, _library=library())
    for step in (0, 3, 45, 400):
        env.current_step = step
        assert env._current_season_multiplier() == 1.0


# (season length, high, low, calls, start energy of the tracked patch)
REGROWTH = [
    (3, 1.5, 0.5, 9, 0.0),       # three phases: high, low, high
    (2, 2.0, 0.0, 6, 0.3),       # a low phase in which nothing grows
    (4, 10.0, 10.0, 8, 1.0),     # runs into the cap (initial_energy_grass) during the first phase and stays there
]


@pytest.mark.parametrize("length,high,low,calls,start", REGROWTH)
def test_regrowth_of_an_untouched_patch_follows_the_wave(length, high, low, calls, start):
    """A patch nobody stands on, everybody told to stay put: after every call its energy is min(cap, previous + gain(step))."""
    env = make(season_length_steps=length, season_high_multiplier=high, season_low_multiplier=low)
    obs, _ = env.reset(seed=3)
    taken = set(env.agent_positions.values())
    patch = next(g for g, pos in env.grass_positions.items() if pos not in taken)
    env.set_grass_energy(patch, start)
    live, want = list(obs), start
    for step in range(calls):
        o, _, term, _, _ = env.step({a: 4 for a in live})     # action 4 = (0, 0)
        live = [a for a in o if not term[a]]
        mult = high if (step // length) % 2 == 0 else low
        want = min(CAP, want + GAIN * mult)                    # the reference's float64 arithmetic, in its order
        assert env.grass_energies[patch] == want, (step, env.grass_energies[patch], want)
