"""Training quality with everything stochastic ON (north star: "AUC within +-0.002 of reference"; VERDICT r2 missing #2).

`tests/golden/train_band_{naml,nrms}.json` hold what the REAL reference reaches on `synthetic.make_learnable_world` --
16 seeds (24 for NRMS at D = 64; 16 for the headline-width bands `*_d256`: D = 256, B = 64; 8 for the MIND-shaped world `naml_mind`:
titles of up to 30 tokens, histories of up to 50 clicks) of its own training loop (torch dropout at its three sites,
python-random negatives, DataLoader(shuffle=True), Adam +
linear schedule), dev rows scored by its own forward + MetricPool (generator: tests/golden/make_train_band.py).  Here the
MI355X trainer path (`TrainStep`: device sampler, Philox dropout, per-epoch reshuffle, fused Adam; `Evaluator` + the metrics
kernel) runs the same world / hyper-parameters from the SAME initial parameters, one run per seed.  The streams differ, so
the comparison is between means:

    |mean_hip - mean_ref| <= 0.002 + 2 * sqrt(se_ref^2 + se_hip^2)        (se = std / sqrt(n_seeds); never above 0.002 + the
                                                                           reference's own max-min seed spread)

and training must have moved the metric the way it moved the reference's (well above both 0.5 and the untrained model)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(kind, band, seed, dev, world, glove):
    from legommenders_amd.evaluate import Evaluator
    from legommenders_amd.synthetic import init_naml_params, init_nrms_params
    from legommenders_amd.train_step import DeviceData, TrainStep
    h = band["hyper"]
    if kind == "naml":
        P = init_naml_params(D=h["D"], A=h["A"], V=world["V"], n_cat=world["n_cat"], seed=seed, glove=glove)
    else:
        P = init_nrms_params(D=h["D"], A=h["A"], V=world["V"], n_cat=world["n_cat"], heads=h["heads"], seed=seed, glove=glove)
    data = DeviceData(world, dev, seed=seed + 1000)
    steps = world["n_rows"] // h["B"] * h["epochs"]
    ts = TrainStep(kind, P, data, h["B"], K=h["K"], lr=h["lr"], total_steps=steps, warmup=0, seed=seed, heads=h["heads"],
                   glove=True, dropout=True, tail="keep")
    assert ts.steps_per_epoch * h["epochs"] == steps == band["runs"][0]["steps"]
    ev = Evaluator(kind, ts.fp.P, data, item_page=256, user_page=128, heads=h["heads"], glove=True)
    v = world["valid"]
    before, _ = ev.evaluate(v["user"], v["item"], v["label"], metrics=tuple(band["metrics"]))
    losses = torch.stack([ts.step().clone() for _ in range(steps)]).flatten().cpu().numpy()
    after, _ = ev.evaluate(v["user"], v["item"], v["label"], metrics=tuple(band["metrics"]))
    return before, after, float(losses[-50:].mean())


BANDS = sorted(os.path.basename(f)[len("train_band_"):-len(".json")] for f in glob.glob(os.path.join(HERE, "golden", "train_band_*.json")))


@pytest.mark.parametrize("name", BANDS)
def test_trained_gauc_matches_the_reference_band(name):
    _check_band(name)


@pytest.mark.parametrize("name", ["naml", "nrms_d256"])
def test_trained_gauc_in_split_bf16_mode(name):
    """the opt-in split-bf16 product mode (tests/test_split_bf16.py) trains to the same bands, same tolerances: NAML (D = 64, 16 seeds)
    and NRMS at the headline width (D = 256, B = 64, 16 seeds)"""
    from legommenders_amd import _lib
    _lib.set_product_mode(_lib.SPLIT_BF16)
    try:
        _check_band(name)
    finally:
        _lib.set_product_mode(_lib.EXACT_F32)


def _keep_report(name, band, seeds, report, runs):
    """the measured means beside the reference's, kept as a file (they lived only in prose): merged into
    $LEGO_BAND_REPORT, or gpurun_out/train_band_report.json when that directory exists (copied to profiles/r05_train_band.json)"""
    from legommenders_amd import _lib
    path = os.environ.get("LEGO_BAND_REPORT")
    if path is None:
        d = os.path.join(os.path.dirname(HERE), "gpurun_out")
        if not os.path.isdir(d):
            return
        path = os.path.join(d, "train_band_report.json")
    try:
        out = json.load(open(path))
    except (OSError, ValueError):
        out = {}
    key = name + ("" if _lib.product_mode() == _lib.EXACT_F32 else "@split_bf16")
    out[key] = {"kind": band["kind"], "seeds": len(seeds), "hyper": {k: band["hyper"][k] for k in ("D", "B", "epochs")},
                "world": {k: band["world"][k] for k in ("T", "S", "n_items", "n_users", "n_rows", "V", "n_dev_users")},
                "metrics": {m: {"mi355x_mean": r[0], "reference_mean": r[1], "tolerance": r[2], "abs_diff": round(abs(r[0] - r[1]), 4),
                                "reference_seed_std": round(float(band["std"][m]), 4)} for m, r in report.items()},
                "mi355x_per_seed_GAUC": [round(float(a["GAUC"]), 4) for _, a, _ in runs],
                "reference_per_seed_GAUC": [round(float(r["after"]["GAUC"]), 4) for r in band["runs"]]}
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)


def _check_band(name):
    from legommenders_amd.synthetic import glove_table_np, make_learnable_world
    dev = torch.device("cuda:0")
    band = json.load(open(os.path.join(HERE, "golden", f"train_band_{name}.json")))
    kind = band["kind"]
    world = make_learnable_world(**band["world"])
    glove = torch.from_numpy(glove_table_np(band["hyper"]["glove_seed"], world["V"]))
    seeds = [r["seed"] for r in band["runs"]]
    runs = [_run(kind, band, s, dev, world, glove) for s in seeds]
    # (1) same starting point: the untrained model scores the dev rows as the reference's untrained model does (eval is exact)
    for (before, _, _), r in zip(runs, band["runs"]):
        for m in band["metrics"]:
            assert abs(before[m] - r["before"][m]) < 1e-3, (kind, r["seed"], m, before[m], r["before"][m])
    n = len(seeds)
    report = {}
    for m in band["metrics"]:
        got = np.array([a[m] for _, a, _ in runs])
        ref = np.array([r["after"][m] for r in band["runs"]])
        se = float(np.sqrt(got.std(ddof=1) ** 2 / n + ref.std(ddof=1) ** 2 / n))
        tol = 0.002 + min(2.0 * se, band["spread"][m])
        report[m] = (round(float(got.mean()), 4), round(float(ref.mean()), 4), round(tol, 4))
        assert abs(got.mean() - ref.mean()) <= tol, (name, m, got.tolist(), ref.tolist(), tol)
        if m == "GAUC":
            # the pin itself has to be tight: the north-star bar is +-0.002; with >= 16 seeds, a 4 000-user dev split and NRMS at its
            # plateau the statistical slack on top of it stays below 0.002 (tol <= 0.004)
            assert tol <= 0.004 or len(seeds) < 16, (name, tol)
    print(name, "mean (hip, reference, tolerance):", report)
    _keep_report(name, band, seeds, report, runs)
    # (2) training did what it did for the reference: clearly above chance and above the untrained model
    g_after = np.mean([a["GAUC"] for _, a, _ in runs])
    g_before = np.mean([b["GAUC"] for b, _, _ in runs])
    assert g_after > 0.6 and g_after - g_before > 0.5 * (band["mean"]["GAUC"] - band["mean_before"]["GAUC"]) > 0.01
    # (3) and the training loss ends where the reference's ends
    ref_loss = np.array([r["last_loss"] for r in band["runs"]])
    got_loss = np.array([l for _, _, l in runs])
    assert abs(got_loss.mean() - ref_loss.mean()) <= 0.02 + 2.0 * float(np.sqrt(got_loss.var(ddof=1) / n + ref_loss.var(ddof=1) / n))
