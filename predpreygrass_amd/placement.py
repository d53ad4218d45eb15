"""Initial placement exactly as the reference draws it.

reset(seed) of the reference (base_environment/predpreygrass_rllib_env.py:157-178, red_queen/...:868-891) draws
(x, y) pairs from ``np.random.default_rng(seed)`` into a Python ``set`` until enough distinct cells exist and then uses
``list(that_set)``: predators take the first cells of the list, then prey, then grass.  The order of that list is the
set's iteration order, i.e. a property of CPython's set (hash of the tuple, table size, insertion history).  The only
way to reproduce it is to do the same on the same interpreter, which is what this host-side helper does (a hundred
cells per episode: nothing to accelerate).  tests/test_placement.py pins it to the placements captured from the
reference in tests/golden/.
"""
from __future__ import annotations

import numpy as np


def reference_placement(grid_size: int, num_positions: int, seed) -> list[tuple[int, int]]:
    if num_positions > grid_size * grid_size:
        raise ValueError("Cannot place more unique positions than grid cells.")  # :167-168 / RQ:881-882
    rng = np.random.default_rng(seed)
    cells = set()
    while len(cells) < num_positions:
        cells.add(tuple(rng.integers(0, grid_size, size=2)))
    return [(int(x), int(y)) for x, y in cells]
