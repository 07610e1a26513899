"""The parts of the reference that ARE importable in the build container (pure pandas / numpy: ``src/batt_models/battgp.py``,
``fault_probabilities.py``, ``ref_strategy.py``, ``cellnr.py``, ``src/operating_point.py``) sit on both sides of the hot path:
they choose the operating point a system is evaluated at and they consume its result frame.  Two kinds of test:

* differential (only where ``/root/reference`` exists, i.e. in the build container): the package's ``BattGPResult`` / ``Op`` /
  ``RefStrategy`` / tags against the reference's own objects on the same inputs, the reference's ``RefStrategy`` / ``Op`` passed
  INTO ``BattGP_Full``, and the reference's ``calc_fault_probabilities`` (``gp_runner.py:111-124``) run unchanged on a result the
  engine produced (CPU build of the kernel sources) next to the same call on an oracle-made frame;
* golden (anywhere): ``tests/golden/system_contract.json``, written by ``make_golden.py::make_system_contract`` from those same
  reference objects - column names, display strings, the operating point each strategy selects.

Test infrastructure only; nothing of the reference is copied - it is imported where it lies, or its OUTPUTS are stored as data.
"""

import json
import os
import sys

import numpy as np
import pandas as pd
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
REF_ROOT = "/root/reference"

from battgp_amd import synthetic  # noqa: E402
from battgp_amd.battgp_full import BattGPResult, RefStrategy  # noqa: E402
from battgp_amd.operating_point import Op, get_causal_tag, get_cell_tag  # noqa: E402


def contract_frame(n_rows: int = 7, cells=(1, 2, 3, 4), seed: int = 3) -> pd.DataFrame:
    """A result frame with every column family the reference's two modes write (acausal for full_gp; causal and the
    ``dr0`` pair only exist for the spatio-temporal mode) - and one hole (no ``dr0var`` of cell 3)."""
    rng = np.random.default_rng(seed)
    cols = {"t": np.linspace(0.0, 100.0, n_rows), "ds_count": np.arange(n_rows)}
    for c in (-1, *cells):
        for causal in (False, True):
            for sig in ("r0", "r0var", "dr0", "dr0var"):
                if sig == "dr0var" and c == 3:
                    continue
                cols[f"{sig}_{get_causal_tag(causal)}_{get_cell_tag(c)}"] = rng.random(n_rows)
    return pd.DataFrame(cols)


CELL_DATA_CALLS = [  # (cellnrs, signals, causal, missing_behaviour)
    (2, ["t", "r0", "r0var"], False, "error"),
    (2, None, False, "error"),
    (-1, ["r0", "t"], True, "error"),
    ([1, 2, 3, 4], ["t", "r0", "r0var"], False, "error"),
    ([4, 1], ["r0var", "t", "ds_count", "r0"], True, "error"),
    ([1, 2, 3], ["t", "dr0var"], False, "ignore"),
    ([1, 2, 3], ["t", "dr0var"], False, "error"),  # cell 3 has none -> ValueError
    (3, None, True, "error"),  # dito
    (3, None, True, "ignore"),
    ([2], ["r0"], False, "error"),  # a list of one keeps the cell tag
    ([-1, 1], ["t", "r0"], False, "error"),
    (1, ["t", "nonsense"], False, "error"),
    (1, ["t", "nonsense"], False, "ignore"),
    ((1, 2), ["t", "t", "r0"], False, "error"),
]


def _outcome(result_obj, call):
    cellnrs, signals, causal, missing = call
    try:
        out = result_obj.get_cell_data(cellnrs, signals, causal, missing)
    except ValueError:
        return "ValueError"
    return out


@pytest.fixture(scope="module")
def ref():
    if not os.path.isdir(os.path.join(REF_ROOT, "src")):
        pytest.skip("the reference tree is only present in the build container")
    sys.path.insert(0, REF_ROOT)
    try:
        from types import SimpleNamespace

        from src.batt_models import battgp, cellnr, fault_probabilities, ref_strategy
        from src import operating_point

        yield SimpleNamespace(battgp=battgp, cellnr=cellnr, faults=fault_probabilities, ref_strategy=ref_strategy, op=operating_point)
    finally:
        sys.path.remove(REF_ROOT)


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("needs the ROCm host clang to build the CPU stand-in")
    from inject import installed

    with installed() as lib:
        yield lib


# ---- differential: against the reference's own objects ---------------------------------------------------------------
def test_get_cell_data_equals_the_reference(ref):
    df = contract_frame()
    ours = BattGPResult(None, [], Op(-15.0, 90.0, 25.0), df)
    theirs = ref.battgp.BattGPResult(None, [], ref.op.Op(-15.0, 90.0, 25.0), df)
    for call in CELL_DATA_CALLS:
        a, b = _outcome(ours, call), _outcome(theirs, call)
        if isinstance(b, str):
            assert a == b, call
        else:
            assert not isinstance(a, str), call
            pd.testing.assert_frame_equal(a, b)


def test_op_and_tags_equal_the_reference(ref):
    for vals in [(-15.0, 90.0, 25.0), (-27.123456, 73.5, 18.004), (0, 100, -5), (np.float64(-80.0), np.float32(40.5), 45)]:
        a, b = Op(*vals), ref.op.Op(*vals)
        assert a.disp_str() == b.disp_str()
        assert np.array_equal(a.into_array(), b.into_array()) and a.into_array().dtype == b.into_array().dtype
        assert np.array_equal(a.into_row_vector(), b.into_row_vector()) and a.into_row_vector().shape == (1, 3)
        assert repr(a) == repr(b) and str(a) == str(b)
        assert a == b and b == a and not (a != b)
        assert a != Op(vals[0], vals[1], vals[2] + 1) and b != Op(vals[0], vals[1], vals[2] + 1)
    kw = Op(T=1.0, I=2.0, SOC=3.0)
    assert kw == ref.op.Op(T=1.0, I=2.0, SOC=3.0)
    for c in (-1, 0, 1, 7, 12, 108):
        assert get_cell_tag(c) == ref.cellnr.get_cell_tag(c)
    for causal in (True, False, 1, 0):
        assert get_causal_tag(causal) == ref.cellnr.get_causal_tag(causal)


class _Data(synthetic.SyntheticBattData):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.mean_op = Op(-21.5, 71.25, 24.75)
        self.median_op = Op(-18.0, 74.0, 26.5)


def test_ref_strategy_selects_the_same_operating_point(ref, emu, tmp_path, capsys):
    """``BattGP.__init__`` (battgp.py:96-128; the base class builds no model, so it runs here) against ``BattGP_Full`` -
    with this package's ``RefStrategy``, with the REFERENCE's ``RefStrategy`` / ``Op`` objects as ``gp_runner.py:53-76``
    passes them, and with the bare strings."""
    from battgp_amd.battgp_full import BattGP_Full

    bd = _Data("sysA", n_cells=2, seed=5)
    manual_ref, manual_own = ref.op.Op(-15.0, 90.0, 25.0), Op(-15.0, 90.0, 25.0)
    for ref_arg, own_args in [
        ("mean", ["mean", RefStrategy("mean"), ref.ref_strategy.RefStrategy("mean")]),
        ("median", ["median", RefStrategy("median"), ref.ref_strategy.RefStrategy("median")]),
        (manual_ref, [manual_own, manual_ref, RefStrategy(manual_own), RefStrategy(manual_ref), ref.ref_strategy.RefStrategy(manual_ref)]),
    ]:
        base = ref.battgp.BattGP(bd, max_training_data=64, save_path=str(tmp_path / "ref"), ref_strategy=ref.ref_strategy.RefStrategy(ref_arg))
        printed_ref = capsys.readouterr().out
        for own in own_args:
            sysm = BattGP_Full(bd, max_training_data=64, device=0, save_path=str(tmp_path / "own"), ref_strategy=own)
            printed_own = capsys.readouterr().out
            assert sysm.ref_op == base.ref_op and sysm.get_operating_point().disp_str() == base.get_operating_point().disp_str()
            assert printed_own == printed_ref  # "Reference operating point: Op(I=..., SOC=..., T=...)"
            assert sysm.max_age == base.max_age and sysm.max_training_data == base.max_training_data
            assert os.path.relpath(sysm.save_path, tmp_path / "own") == os.path.relpath(base.save_path, tmp_path / "ref")
    # the default is the mean (battgp_full.py:22); the mean / median setters and the explicit one (battgp.py:131-170)
    sysm = BattGP_Full(bd, max_training_data=64, device=0)
    base = ref.battgp.BattGP(bd, max_training_data=64)
    capsys.readouterr()
    for name in ("set_operating_point_to_median", "set_operating_point_to_mean"):
        getattr(sysm, name)()
        own_out = capsys.readouterr().out
        getattr(base, name)()
        assert own_out == capsys.readouterr().out and sysm.get_operating_point() == base.get_operating_point()
    with pytest.raises(ValueError):
        RefStrategy("mode")
    with pytest.raises(ValueError):
        ref.ref_strategy.RefStrategy("mode")
    with pytest.raises(ValueError):
        RefStrategy(3.0)
    with pytest.raises(ValueError):
        RefStrategy("mean").get_manual_value()
    assert sysm.get_cell_model(-1) is sysm.packmodel
    with pytest.raises(ValueError, match="does not exist"):
        sysm.get_cell_model(99)


def test_reference_fault_probabilities_consume_the_engine_result(ref, emu, tmp_path):
    """``gp_runner.py``'s full_gp branch after the hot path (``:96-124``): ``calc_fault_probabilities(gp_res, causal=False,
    r0_band=, r0_upper_threshold=)`` of the reference, run UNCHANGED on what ``BattGP_Full.predict_cell_r0_op`` returns
    (kernels: the CPU build of csrc/*.hip), next to the same call on the reference's own ``BattGPResult`` around a frame
    whose columns the oracle computed; and the saved feather read back into the reference's result type."""
    from battgp_amd.battgp_full import BattGP_Full
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP

    bd = synthetic.SyntheticBattData("sysF", n_cells=3, seed=11)
    n_train = 150
    sysm = BattGP_Full(bd, max_training_data=n_train, device=0, save_path=str(tmp_path), ref_strategy=ref.ref_strategy.RefStrategy(ref.op.Op(-15.0, 90.0, 25.0)))
    res = sysm.predict_cell_r0_op()
    t = res.df["t"].to_numpy()
    xq = np.column_stack((t, np.full(t.size, -15.0), np.full(t.size, 90.0), np.full(t.size, 25.0)))
    cols = {"t": t}
    for c in (-1, *bd.cell_nrs):
        x, y = bd.generateTrainingData(c, n_train)
        mean, var = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit().predict(xq)
        cols[f"r0_acausal_{get_cell_tag(c)}"] = mean
        cols[f"r0var_acausal_{get_cell_tag(c)}"] = np.maximum(var, 1e-10)
    oracle_res = ref.battgp.BattGPResult(bd, [], ref.op.Op(-15.0, 90.0, 25.0), pd.DataFrame(cols))
    assert list(res.df.columns) == list(oracle_res.df.columns)
    for band, thr in [(0.5e-3, 2.0e-3), (1.0e-3, 13.0e-3)]:
        got = ref.faults.calc_fault_probabilities(res, causal=False, r0_band=band, r0_upper_threshold=thr)
        want = ref.faults.calc_fault_probabilities(oracle_res, causal=False, r0_band=band, r0_upper_threshold=thr)
        assert list(got.columns) == list(want.columns) and got.shape == (300, want.shape[1])
        g, w = got.to_numpy(dtype=float), want.to_numpy(dtype=float)
        assert np.isfinite(g).all()
        assert np.abs(g - w).max() <= 1e-6 * max(1.0, np.abs(w).max())  # probabilities in [0, 1], resistances ~1e-2
    # the artefact on disk (battgp.py:233-262) loads into the reference's result type and answers the same queries
    saved = pd.read_feather(tmp_path / "sysF" / "battgpf_df.feather")
    reloaded = ref.battgp.BattGPResult(bd, [], res.ref_op, saved)
    pd.testing.assert_frame_equal(reloaded.get_cell_data(bd.cell_nrs, ["t", "r0", "r0var"]), res.get_cell_data(bd.cell_nrs, ["t", "r0", "r0var"]))
    info = json.load(open(tmp_path / "sysF" / "battgpf_info.json"))
    assert info["ref_point"] == ref.op.Op(-15.0, 90.0, 25.0).disp_str()


# ---- golden: the same facts as data, wherever the suite runs ---------------------------------------------------------
def test_system_contract_golden(golden_dir):
    with open(os.path.join(golden_dir, "system_contract.json")) as fil:
        gold = json.load(fil)
    df = contract_frame(**gold["frame"])
    assert list(df.columns) == gold["frame_columns"]
    ours = BattGPResult(None, [], Op(-15.0, 90.0, 25.0), df)
    assert len(gold["cell_data"]) == len(CELL_DATA_CALLS)
    for call, want in zip(CELL_DATA_CALLS, gold["cell_data"]):
        got = _outcome(ours, call)
        if want == "ValueError":
            assert isinstance(got, str), call
            continue
        assert list(got.columns) == want["columns"], call
        # the values under each output column are those of the recorded source column
        for pos, src in enumerate(want["sources"]):
            assert np.array_equal(got.iloc[:, pos].to_numpy(), df[src].to_numpy()), (call, pos)
    for vals, text, rep in gold["op"]:
        assert Op(*vals).disp_str() == text and repr(Op(*vals)) == rep
    for c, tag in gold["cell_tags"]:
        assert get_cell_tag(c) == tag
    assert [get_causal_tag(True), get_causal_tag(False)] == gold["causal_tags"]
    bd = _Data("g", n_cells=1)
    from battgp_amd.battgp_full import _resolve_ref_op

    for strategy, want in gold["strategy_picks"].items():
        assert _resolve_ref_op(bd, strategy, None).disp_str() == want
    assert _resolve_ref_op(bd, RefStrategy(Op(1.0, 2.0, 3.0)), None).disp_str() == gold["manual_pick"]


# ---- the reference's own ETL in front of the engine: gp_runner.py's full_gp flow, reference code on both sides -----------
def _write_raw_field_data(folder, batt_id="syn", rows=2600, seed=21):
    """A raw field-data file in the layout ``src/batt_data/data_utils.py`` reads (``data_sys_<id>.csv``, the column names of
    ``data_columns.py``): an 8s1p pack over ~430 days whose cell voltages follow ``U = OCV(SOC) + I R0`` with a known
    ``R0(t, T)`` per cell, a share of rows outside the segment limits of ``config.py:99-111`` and a few NaNs for the
    clean-up to remove.  Returns the true resistance function."""
    rng = np.random.default_rng(seed)
    ocv = pd.read_csv(os.path.join(REF_ROOT, "data", "ocv_linear_approx.csv"))
    t_days = np.sort(rng.uniform(0.0, 430.0, rows))
    index = pd.Timestamp("2021-03-01") + pd.to_timedelta(np.round(t_days * 86400.0), unit="s")
    soc = rng.uniform(35.0, 99.0, rows)
    i_batt = rng.uniform(-90.0, 0.0, rows)
    temps = 25.0 + 12.0 * np.sin(2 * np.pi * t_days / 365.0)[:, None] + rng.normal(0.0, 1.0, (rows, 4))
    ocv_cell = np.interp(soc, ocv["SOC"], ocv["OCV"])

    def r0_true(cell, t, temp):
        return 0.012 + 0.002 * np.exp(-(temp - 10.0) / 20.0) + 1.5e-6 * t + 1.5e-4 * ((cell * 7) % 5 - 2)

    cols = {"SOC_Battery": soc, "I_Battery": i_batt}
    for k in range(4):
        cols[f"Temperature_{k + 1}"] = temps[:, k]
    u_sum = np.zeros(rows)
    for c in range(1, 9):
        i_cnv = rng.normal(0.0, 1.0, rows)
        i_cnv[rng.random(rows) < 0.02] = 25.0  # balancing current outside CNV_LIMIT: the row is not a valid segment
        u = ocv_cell + (i_batt + i_cnv) * r0_true(c, t_days, temps[:, (c - 1) // 2]) + rng.normal(0.0, 2e-4, rows)
        cols[f"U_Cell_{c}"] = u
        cols[f"I_CNV_Cell_{c}"] = i_cnv
        cols[f"T_CNV_Cell_{c}"] = temps[:, (c - 1) // 2] + 5.0  # read, not imported (data_columns.py: None)
        cols[f"SOC_Cell_{c}"] = soc
        u_sum += u
    cols["U_Battery"] = u_sum
    cols["U_CR"] = np.zeros(rows)
    cols["I_CR"] = np.zeros(rows)
    df = pd.DataFrame(cols, index=index)
    df.index.name = "time_stamp"
    df.iloc[5, 3] = np.nan
    df.iloc[77, 0] = np.inf
    df = df[~df.index.duplicated(keep="first")]
    df.to_csv(os.path.join(folder, f"data_sys_{batt_id}.csv"))
    return r0_true


def test_gp_runner_full_gp_flow_with_the_reference_etl_in_front(ref, emu, tmp_path, monkeypatch):
    """``gp_runner.py:44-124`` (``MODE = "full_gp"``, plots aside): raw field data -> the REFERENCE's ``BattData`` (clean-up,
    float32 cache, segment selection, gap removal, ``R = (U - OCV) / I``, even-index sub-sampling) -> ``BattGP_Full`` of this
    package with the reference's ``RefStrategy(Op)`` -> ``train`` off (``HYPER_OPT_PARAMS["optimize"]`` is False in
    ``config.py``) -> ``predict_cell_r0_op`` -> the REFERENCE's ``calc_fault_probabilities``.  The engine (CPU build of the
    kernel sources here) sits between reference code on both sides, unchanged; every result column equals an oracle GP on
    what the reference's ``generateTrainingData`` hands over, and the resistance it reports is the one the file was made
    with."""
    import src.config as cfg
    from src.batt_data.batt_data import BattData
    from src.batt_data.data_utils import read_cell_characteristics

    from battgp_amd.battgp_full import BattGP_Full
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP

    raw_dir, cache_dir = tmp_path / "raw", tmp_path / "cache"
    raw_dir.mkdir()
    cache_dir.mkdir()
    r0_true = _write_raw_field_data(str(raw_dir))
    monkeypatch.setattr(cfg, "PATH_FIELDDATA_DATA", str(raw_dir))
    monkeypatch.setattr(cfg, "PATH_DATA_CACHE", str(cache_dir))
    cell_ocv = read_cell_characteristics(os.path.join(REF_ROOT, "data", "ocv_linear_approx.csv"))
    battdata = BattData("syn", cell_ocv, segment_selection=True, gap_removal=cfg.GAP_REMOVAL, min_data_threshold=1000)
    assert battdata.cell_nrs == list(range(1, 9)) and 1000 <= len(battdata.df) < 2600  # the segment filter removed rows
    # the stand-in of this package offers the same contract as the real class
    stand_in = synthetic.SyntheticBattData("x", n_cells=8)
    for attr in ("id", "age", "cell_nrs", "mean_op", "median_op", "generateTrainingData"):
        assert hasattr(stand_in, attr) and hasattr(battdata, attr)

    n_train = 160
    op = ref.op.Op(-15, 90, 25)  # gp_runner.py:32
    model = BattGP_Full(battdata, ref_strategy=ref.ref_strategy.RefStrategy(op), max_training_data=n_train, max_age=None,
                        device=0, save_path=str(tmp_path / "results"))
    gp_res = model.predict_cell_r0_op()
    df = gp_res.df
    assert df.shape == (300, 1 + 2 * 9) and np.isfinite(df.to_numpy()).all()
    t = df["t"].to_numpy()
    xq = np.column_stack((t, np.full(300, float(op.I)), np.full(300, float(op.SOC)), np.full(300, float(op.T))))
    hyp = np.array([cfg.NOISE_VARIANCE[0] if isinstance(cfg.NOISE_VARIANCE, tuple) else cfg.NOISE_VARIANCE, cfg.OUTPUTSCALE_WIENER,
                    cfg.OUTPUTSCALE_RBF, *cfg.LENGTHSCALE_RBF], dtype=np.float64)
    for c in (-1, *battdata.cell_nrs):
        x, y = battdata.generateTrainingData(c, n_train, None)
        assert x.shape == (n_train, 4) and x[0, 0] == 0.0  # time restarts at the first kept sample (batt_data.py:90-94)
        mean, var = OracleGP(K.KERNEL_BATTGP, hyp, x, y).fit().predict(xq)
        tag = get_cell_tag(c)
        assert np.linalg.norm(df[f"r0_acausal_{tag}"] - mean) <= 1e-6 * np.linalg.norm(mean), tag
        assert np.max(np.abs(df[f"r0var_acausal_{tag}"] - var)) <= 1e-6 * np.max(var) + 1e-12, tag
        if c > 0:  # the engine reports the resistance the file was generated with (25 degC, inside the data's time span)
            inner = (t > 30.0) & (t < t[-1] - 30.0)
            assert np.max(np.abs(df[f"r0_acausal_{tag}"].to_numpy()[inner] - r0_true(c, t[inner], 25.0))) < 6e-4, tag
    for band in 10 ** (-3) * np.array([0.55]):  # gp_runner.py:31
        faults = ref.faults.calc_fault_probabilities(gp_res, causal=False, r0_band=band, r0_upper_threshold=2.0e-3)
        vals = faults.filter(like="fault prob").to_numpy(dtype=float)
        assert faults.shape[0] == 300 and np.isfinite(vals).all() and vals.min() >= 0.0 and vals.max() <= 1.0
    assert os.path.exists(tmp_path / "results" / "syn" / "battgpf_df.feather")
