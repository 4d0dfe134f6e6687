"""CPU tests of the build planning bench.py relies on (sh-assembly_amd/shk/plan.py): the occupancy model against
the oracle on a scaled-down run of the bench workload, and the plans of the driver's command lines."""
import importlib.util
import os

import pytest

from shk import plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, L, ERR, G, R, NCH = 47, 150, 0.00234, 119_157_843, 8_000_000, 302   # bench.py defaults (302 chunks of 8 MiB per step)


def _plan_check():
    spec = importlib.util.spec_from_file_location("plan_check", os.path.join(ROOT, "tools", "plan_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_model_tracks_oracle_schedule():
    """the bench workload at 1/2048 scale through the t = 1 schedule on the oracle: occupied slots, distinct count
    and rounds fired after every step agree with the prediction (slots within 4 %)"""
    r = _plan_check().run(scale=2048, steps=10, verbose=False)
    assert r["full_at_step"] is None
    assert len(r["per_step"]) == len(r["pred_per_step"]) == 10
    same = 0
    for (u, d, f), (pu, pd, pf) in zip(r["per_step"], r["pred_per_step"]):
        assert abs(f - pf) <= 1
        if f != pf:      # a round that fires within a few chunks of a step boundary may land on either side of it
            continue
        same += 1
        assert abs(u - pu) <= 0.04 * u, (r["per_step"], r["pred_per_step"])
        assert abs(d - pd) <= 0.06 * d
    assert same >= 6
    assert abs(r["peak_load"] - r["pred_peak_load"]) <= 0.04
    assert r["fired"] >= 2   # the comparison covers deNoise rounds


def test_model_flags_the_round1_overflow():
    """BENCH_r01: 25 batches into one filter sized by the formula for exactly those k-mers (qb 29, 20 rounds) overflowed
    in step 24 on the GPU and in the oracle at 1/256 scale; the model must say so too"""
    S = R * (L - K + 1)
    qb, nd, trig = plan.sizing(K, G, 25 * S, ERR)
    assert (qb, nd) == (29, 20)
    p = plan.predict_build(K, G, L, ERR, S / NCH, NCH * 25, trig, nd)
    assert p["peak_slots"] > plan.xnslots(qb)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("steps", [1, 6, 20, 25])
def test_planned_builds_fit(steps, world):
    """every build bench.py plans (the driver runs --steps 20 --warmup 5 at 1, 2, 4 and 8 GPUs; the default is 6 steps)
    stays at or below 95 % of the slots at its predicted peak -- well clear of xnslots"""
    S = R * (L - K + 1)
    pl = plan.plan_build(K, G * world, L, ERR, S * world / NCH, NCH * steps, world=world)
    assert pl["predicted_peak_load"] <= 0.95
    assert pl["predicted_peak_slots"] < 0.96 * (1 << pl["qb"]) < plan.xnslots(pl["qb"])
    assert pl["rounds"] >= pl["formula_rounds"]
    if steps == 20 and world == 1:
        # BASELINE config 1: the README's table size and (nearly) its schedule
        assert pl["qb"] == 29 and pl["formula_rounds"] == 8 and pl["rounds"] <= 14


@pytest.mark.parametrize("profile", [(0.0005, 0.00418)])     # (None = the uniform rate: tools/plan_check.py, same outcome)
def test_readme_recipe_overflows_whatever_the_error_spectrum(profile):
    """is the 12-round plan an artefact of the uniform-substitution read model? The README recipe as
    written (rounds and trigger straight from src/CQF-deNoise.cpp:96-161: 8 rounds), the whole 20-step build at 1/2048
    scale on the oracle, with the bench's uniform error rate and with rates rising linearly along the read (mean unchanged,
    what an --errorProfile file describes): the table fills up in the last step either way. What overflows it are error
    k-mers seen TWICE between two rounds (they survive every later round; the formula budgets every false k-mer as
    removable), and how often an error recurs depends on coverage x rate per genome position, which the shape of the
    profile along the read does not change -- the README's own ntCard figures show the same population in the real data
    (f2 = 26,122,317 doubletons next to n = 119,157,843 true k-mers, README.md:81-92)."""
    r = _plan_check().run(scale=2048, steps=20, verbose=False, profile=profile)
    assert (r["rounds"], r["fired"]) == (8, 8)
    assert r["full_at_step"] is not None and r["full_at_step"] >= 17
    assert r["pred_peak_load"] > 1.0
