"""`reference_placement` must give the cells the reference's reset(seed) picked (captured in the golden vectors)."""
import json

import numpy as np
import pytest

from predpreygrass_amd.placement import reference_placement
from tests import golden_io, golden_io_rq


@pytest.mark.parametrize("name", golden_io.case_names())
def test_base_family_placement(name):
    z = golden_io.GoldenCase(name).z
    cfg = json.loads(str(z["config_json"]))
    want = np.concatenate([z["pred_xy"].reshape(-1, 2), z["prey_xy"].reshape(-1, 2), z["grass_xy"].reshape(-1, 2)])
    G = cfg.get("grid_size", 25)
    got = np.array(reference_placement(G, len(want), int(z["seed"])))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", golden_io_rq.case_names())
def test_second_generation_placement(name):
    c = golden_io_rq.RQGoldenCase(name)
    want = np.concatenate([a.reshape(-1, 2) for a in c.placement])
    got = np.array(reference_placement(c.config["grid_size"], len(want), int(c.z["seed"])))
    assert np.array_equal(got, want)


def test_too_many_positions():
    with pytest.raises(ValueError, match="Cannot place more unique positions"):
        reference_placement(3, 10, 0)
