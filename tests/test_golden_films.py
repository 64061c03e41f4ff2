"""Golden films (tests/golden/render_fixtures.npz, written by tests/golden/make_render_fixtures.py from the deterministic-math
oracle): the oracle must still produce them (CPU), and the HIP path must reproduce them bit for bit (-m gpu)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_render_fixtures import CASES, render_case  # noqa: E402

from fountain_amd import _abi as A  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_fixtures.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden_film(orc_det, name):
    px, st = render_case(orc_det, name)
    assert np.array_equal(px.view(np.uint32), GOLD[name].view(np.uint32))
    assert [st["rays_closest"], st["rays_any"], st["camera_samples"]] == GOLD[name + "__rays"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_reproduces_golden_film(gpu, name):
    pipelines = [A.FTN_PIPELINE_MEGAKERNEL] if name in ("furnace_tile_serial", "cornell_direct") else [A.FTN_PIPELINE_WAVEFRONT, A.FTN_PIPELINE_MEGAKERNEL]
    for pl in pipelines:
        px, st = render_case(gpu, name, pipeline=pl)
        diff = (px.view(np.uint32) != GOLD[name].view(np.uint32)).any(axis=-1)
        assert int(diff.sum()) <= 4 * st["spill_samples"], (name, pl, int(diff.sum()))
        assert np.allclose(px, GOLD[name], rtol=2e-6, atol=1e-7)
        assert [st["rays_closest"], st["rays_any"], st["camera_samples"]] == GOLD[name + "__rays"].tolist()
