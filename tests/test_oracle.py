"""CPU suite: the oracle (our restatement of render0) against the reference's golden frames.

The golden frames were produced by the UNMODIFIED reference engine (tests/golden/make_golden.py);
the bar is bit-exact 0x00RRGGBB equality.
"""
import numpy as np
import pytest

from conftest import MANIFEST, SMALL_CASES, load_blob, load_frame


@pytest.mark.parametrize("name", SMALL_CASES)
def test_oracle_matches_reference_frame(oracle, name):
    blob = load_blob(name)
    ref = load_frame(name)
    ours, _, counts = oracle.render(blob, threads=4)
    assert ours.shape == ref.shape
    diff = int((ours != (ref & 0xFFFFFF)).sum())
    assert diff == 0, f"{name}: {diff} pixels differ from the reference frame"
    assert int(MANIFEST[name]["hash"], 16) == oracle.frame_hash(ours)
    assert counts["primary"] >= ref.size


@pytest.mark.parametrize("name", ["c2_demo01_1080p_d0", "c3_demo02_1080p_gf_d3"])
def test_oracle_matches_reference_hash_full_size(oracle, name):
    """BASELINE.json configs 2 and 3 at full 1920x1080: frame hash equals the reference's."""
    blob = load_blob(name)
    ours, _, _ = oracle.render(blob, threads=8)
    assert oracle.frame_hash(ours) == int(MANIFEST[name]["hash"], 16)


def test_oracle_row_interleave_matches_whole_frame(oracle):
    """index/thnum slicing (tracer.cpp:1144-1145, 5385-5386) composes to the whole frame."""
    blob = load_blob("demo01_160")
    whole, _, _ = oracle.render(blob, threads=2)
    acc = np.zeros_like(whole)
    for idx in range(3):
        part, _, _ = oracle.render(blob, threads=2, index=idx, thnum=3)
        acc |= part
    assert (acc == whole).all()


def test_oracle_depth_zero_has_no_secondary_rays(oracle):
    blob = load_blob("demo02_160_gf_d3")
    _, _, c = oracle.render(blob, depth=0, threads=2)
    assert c["reflect"] == 0 and c["refract"] == 0 and c["shadow"] > 0


@pytest.mark.parametrize("name", SMALL_CASES)
def test_deferred_shading_is_pixel_identical(oracle, name):
    """The HIP backend shades only the final hit of a list walk; the reference shades every
    depth-test winner.  Check on the CPU that both give the same pixels and hit ids, and that
    deferred shading never traces more rays."""
    blob = load_blob(name)
    eager, ids_e, c_e = oracle.render(blob, threads=4, want_ids=True)
    lazy, ids_l, c_l = oracle.render(blob, threads=4, want_ids=True, deferred=True)
    assert (eager == lazy).all() and (ids_e == ids_l).all()
    assert c_l["primary"] == c_e["primary"]
    assert all(c_l[k] <= c_e[k] for k in c_e)


def test_work_counter_is_deterministic_and_matches_committed_table(oracle):
    """The fp32-operation counter behind bench.py's VALU roofline (SURVEY.md 8(d) weights): independent
    of the thread count, smaller in deferred mode, and equal to tests/golden/work.json at full size."""
    import json, os
    blob = load_blob("demo02_160_gf_aa4")
    _, _, a = oracle.render(blob, threads=1)
    _, _, b = oracle.render(blob, threads=4)
    _, _, d = oracle.render(blob, threads=4, deferred=True)
    assert a == b and a["flops"] > 0
    assert 0 < d["flops"] <= a["flops"]
    work = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "work.json")))
    blob = load_blob("c2_demo01_1080p_d0")
    _, _, c = oracle.render(blob, threads=8, deferred=True)
    w = work["c2_demo01_1080p_d0"]["deferred"]
    assert c["flops"] == w["flops"]
    assert c["primary"] + c["shadow"] + c["reflect"] + c["refract"] == w["rays"]
