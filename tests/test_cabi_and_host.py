"""CPU tests: the C-ABI library loads and exports everything include/battgp.h declares; the product
fails loudly without a GPU; host-side logic of the reference-shaped adaptor."""

import math
import os
import re
import sys

import numpy as np
import pytest

from battgp_amd import _lib, synthetic
from battgp_amd import config as cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "battgp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bgp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libbattgp.so does not export {name}"
    assert sorted(_lib.SIGNATURES) == declared, "battgp_amd/_lib.py and include/battgp.h disagree"
    assert lib.bgp_version() >= 100


def test_header_constants_match_python_binding():
    text = open(os.path.join(ROOT, "include", "battgp.h")).read()
    for name, val in (("BGP_T_FILL", _lib.T_FILL), ("BGP_T_POTRF", _lib.T_POTRF), ("BGP_T_TRAIL", _lib.T_TRAIL),
                      ("BGP_T_TRAIL_FLOP", _lib.T_TRAIL_FLOP), ("BGP_T_FILL_BYTES", _lib.T_FILL_BYTES),
                      ("BGP_T_COUNT", _lib.T_COUNT)):
        m = re.search(rf"{name}\s*=\s*(\d+)", text)
        assert m and int(m.group(1)) == val, name
    import battgp_amd

    for name, val in (("BGP_KERNEL_BATTGP", battgp_amd.KERNEL_BATTGP), ("BGP_KERNEL_SCALED_RBF", battgp_amd.KERNEL_SCALED_RBF),
                      ("BGP_KERNEL_MATERN32", battgp_amd.KERNEL_MATERN32), ("BGP_KERNEL_ARD_RBF", battgp_amd.KERNEL_ARD_RBF)):
        m = re.search(rf"{name}\s*=\s*(\d+)", text)
        assert m and int(m.group(1)) == val, name


def _has_gpu():
    import torch

    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    from battgp_amd.engine import EngineError, ExactGPEngine

    with pytest.raises(EngineError, match="no HIP device|bgp_create"):
        ExactGPEngine(0, synthetic.HYP_BATTGP)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "battgp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_device_argument_forms():
    import torch

    from battgp_amd.engine import EngineError, as_device_index

    assert as_device_index(None) == 0
    assert as_device_index(3) == 3
    assert as_device_index(torch.device("cuda", 2)) == 2
    assert as_device_index("cuda:5") == 5
    with pytest.raises(EngineError):
        as_device_index(torch.device("cpu"))


def test_adaptor_construction_is_lazy_and_validates_kwargs():
    from battgp_amd.battcellgp_full import BatteryCellGP_Full

    x, y = synthetic.make_cell_data(50)
    cell = BatteryCellGP_Full(x, y, cellnr=3)  # no GPU needed: the engine is created on first use
    assert cell.cellnr == 3
    p = cell.get_parameters()
    assert set(p) == set(BatteryCellGP_Full.get_default_parameters())
    assert p["noise_variance"] == cfg.NOISE_VARIANCE and p["lengthscale_rbf"] == cfg.LENGTHSCALE_RBF
    p["max_iter"] = -1  # deep copy: does not leak back
    assert cell.get_parameters()["max_iter"] == cfg.OPTIM_MAX_ITER
    xt, yt = cell.get_training_data()
    assert np.array_equal(xt, x) and np.array_equal(yt, y)
    assert tuple(cell.model.train_inputs[0].shape) == (50, 4)
    assert float(cell.model.train_inputs[0][:, 0][0].detach().cpu()) == 0.0  # battgp_full.py:84,98
    assert np.allclose(cell.model.hyp_vector(), synthetic.HYP_BATTGP, rtol=0, atol=0)
    with pytest.raises(ValueError, match="unknown keyword parameter 'nois'"):
        BatteryCellGP_Full(x, y, 1, nois=1.0)
    cell2 = BatteryCellGP_Full(x, y, 1, noise_variance=1e-3, lengthscale_rbf=(1.0, 2.0, 3.0))
    assert np.allclose(cell2.model.hyp_vector(), [1e-3, cfg.OUTPUTSCALE_WIENER, cfg.OUTPUTSCALE_RBF, 1, 2, 3])
    del cell.model  # what BattGP_Full does between cells (battgp_full.py:103,118)
    assert not hasattr(cell, "model")


def test_hyperparameter_csv_layout(tmp_path):
    import pandas as pd

    from battgp_amd.battcellgp_full import BatteryCellGP_Full

    x, y = synthetic.make_cell_data(20)
    cell = BatteryCellGP_Full(x, y, cellnr=-1)
    with pytest.raises(AttributeError):  # like the reference: only set by train_hyperparameters
        cell.save_hyperparameters(str(tmp_path))
    cell.marginallikelihood = 12.5
    cell.save_hyperparameters(str(tmp_path))
    df = pd.read_csv(tmp_path / "-1hyperparams.csv", index_col=0)
    assert list(df.index) == ["Noise Variance", "Wiener Outputscale", "RBF Outputscale", "RBF Lengthscale 1",
                              "RBF Lengthscale 2", "RBF Lengthscale 3", "Marginal Likelihood"]
    assert float(df.loc["Marginal Likelihood", "params"]) == 12.5


def test_constraints_match_gpytorch_transforms():
    from battgp_amd.cell_gp import Constraint, constraint_from_range

    iv = constraint_from_range((1e-15, 1e4))  # Interval -> sigmoid
    assert iv.kind == "interval"
    raw = np.array([-3.0, 0.0, 2.5])
    val = iv.transform(raw)
    assert np.allclose(val, 1e-15 + (1e4 - 1e-15) / (1 + np.exp(-raw)))
    assert np.allclose(iv.inverse_transform(val), raw)
    pos = constraint_from_range((0.0, math.inf))  # Positive -> softplus
    assert pos.kind == "greater"
    assert np.allclose(pos.transform(raw), np.log1p(np.exp(raw)))
    assert np.allclose(pos.inverse_transform(pos.transform(raw)), raw)
    vec = constraint_from_range(((1e-5, 1e4), (1e-5, 1e4), (1e-5, 1e4)))
    assert vec.kind == "interval" and vec.lower_bound.shape == (3,)
    for c in (iv, pos, Constraint(-math.inf, 5.0)):
        h = 1e-6
        fd = (c.transform(raw + h) - c.transform(raw - h)) / (2 * h)
        assert np.allclose(c.dvalue_draw(raw), fd, rtol=1e-6)


def test_raw_vector_round_trip_and_lengthscale_softplus_quirk():
    from battgp_amd.battcellgp_full import BatteryCellGP_Full

    x, y = synthetic.make_cell_data(10)
    m = BatteryCellGP_Full(x, y, 1).model
    raw = m.raw_vector()
    m.set_raw_vector(raw)
    assert np.allclose(m.hyp_vector(), synthetic.HYP_BATTGP, rtol=1e-12)
    # lengthscale keeps GPyTorch's default Positive() transform (reference bug cell_gp.py:194 vs :182)
    assert np.allclose(np.log1p(np.exp(raw[3:])), cfg.LENGTHSCALE_RBF)


def test_training_loops_follow_reference_stop_rule(monkeypatch):
    """training.train_exact_gp_adam with a stub loss: history length, NaN padding, rel_ftol stop and the
    final entry (src/gp/training.py:35-65) - no GPU involved."""
    from battgp_amd import training
    from battgp_amd.battcellgp_full import BatteryCellGP_Full

    x, y = synthetic.make_cell_data(10)
    model = BatteryCellGP_Full(x, y, 1).model
    target = model.raw_vector() + 0.5

    def fake_neg_mll():
        return 1.0 + float(np.sum((model.raw_vector() - target) ** 2))

    def fake_loss_and_grad():
        return fake_neg_mll(), 2.0 * (model.raw_vector() - target)

    monkeypatch.setattr(model, "neg_mll", fake_neg_mll)
    monkeypatch.setattr(model, "neg_mll_and_raw_grad", fake_loss_and_grad)
    losses = training.train_exact_gp_adam(model, max_iter=200, rel_ftol=0.0, loss_scale=10, lr=0.05, messages=False)
    assert losses.shape == (201,) and not np.isnan(losses).any()
    assert losses[0] == pytest.approx(10 * (1 + 6 * 0.25))
    assert losses[-1] < 10.2  # converged towards the minimum value 1.0 * loss_scale
    model.set_raw_vector(target - 0.5)
    losses = training.train_exact_gp_adam(model, max_iter=500, rel_ftol=1e-3, loss_scale=1, lr=0.05, messages=False)
    assert 2 < len(losses) < 500  # stopped by the relative-change rule; array truncated like the reference
    assert model.training is False and model.likelihood.training is False


def test_synthetic_inputs_are_deterministic_and_in_range():
    x1, y1 = synthetic.make_cell_data(300)
    x2, y2 = synthetic.make_cell_data(300)
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2)
    assert x1[0, 0] == 0.0 and np.all(np.diff(x1[:, 0]) >= 0)
    assert x1[:, 1].min() >= -80 and x1[:, 1].max() <= -5
    assert x1[:, 2].min() >= 40 and x1[:, 2].max() <= 95
    assert x1[:, 3].min() >= 10 and x1[:, 3].max() <= 45
    q = synthetic.make_query(x1)
    assert q.shape == (300, 4) and np.all(q[:, 1:] == np.array(synthetic.REF_OP))


def test_cells_for_rank_deals_nine_gps_over_eight_gpus():
    from battgp_amd.parallel import cells_for_rank

    cells = [-1, 1, 2, 3, 4, 5, 6, 7, 8]
    assert cells_for_rank(cells, 0, 8) == [-1, 8]
    assert cells_for_rank(cells, 7, 8) == [7]
    assert sorted(sum((cells_for_rank(cells, r, 8) for r in range(8)), [])) == sorted(cells)


def test_system_driver_concurrency_is_a_scheduling_choice_only(monkeypatch):
    """BattGP_Full.predict_cell_r0_op: sequential, automatic and explicit ``in_flight`` / ``devices=[0]*k`` must hand
    back the same frame (each GP is computed by its own model whatever runs next to it), run at most the requested
    number of models at a time, free every model but the last, and re-raise a worker's exception in the caller."""
    import threading
    import time

    import pandas as pd

    from battgp_amd import battgp_full
    from battgp_amd.synthetic import SyntheticBattData

    state = {"now": 0, "peak": 0, "freed": [], "lock": threading.Lock(), "fail": None}

    class FakeModel:
        def __init__(self, cellnr, n):
            import torch

            self.cellnr = cellnr
            self.device_ = "cuda:0"
            self.model = type("M", (), {})()
            self.model.train_inputs = (torch.arange(4.0 * n, dtype=torch.float64).reshape(n, 4),)
            self.model.train_targets = torch.zeros(n, dtype=torch.float64)

        def predict_r0_op(self, op, t):
            with state["lock"]:
                state["now"] += 1
                state["peak"] = max(state["peak"], state["now"])
            time.sleep(0.02)
            with state["lock"]:
                state["now"] -= 1
            if state["fail"] == self.cellnr:
                raise RuntimeError(f"cell {self.cellnr} failed")
            tag = "pack" if self.cellnr == -1 else f"c{self.cellnr}"
            return pd.DataFrame({"t": t, f"r0_acausal_{tag}": t * (self.cellnr + 2), f"r0var_acausal_{tag}": t * 0 + self.cellnr})

        def __delattr__(self, name):
            if name == "model":
                state["freed"].append(self.cellnr)
            super().__delattr__(name)

    monkeypatch.setattr(battgp_full, "build_cellmodel_full", lambda c, bd, max_training_data, max_age=None, device=None, **kw: FakeModel(c, 50))
    bd = SyntheticBattData("conc", n_cells=8, seed=1)

    def run(**kw):
        state.update(now=0, peak=0, freed=[])
        sysm = battgp_full.BattGP_Full(bd, max_training_data=50, **kw)
        return sysm.predict_cell_r0_op(save=False).df

    ref = run(device=0, in_flight=1)
    assert state["peak"] == 1 and sorted(state["freed"]) == [-1, 1, 2, 3, 4, 5, 6, 7]  # the last cell model stays alive
    assert ref.shape == (300, 1 + 2 * 9)
    for kw, peak in (({"device": 0, "in_flight": 4}, 4), ({"devices": [0, 0, 0]}, 3), ({"devices": [0, 1], "in_flight": 2}, 4)):
        df = run(**kw)
        assert df.equals(ref), kw
        assert 1 < state["peak"] <= peak, (kw, state["peak"])
        assert sorted(state["freed"]) == [-1, 1, 2, 3, 4, 5, 6, 7]
    # automatic: no GPU in this process -> mem_get_info fails -> sequential, never a crash
    df = run(device=0)
    assert df.equals(ref)
    state["fail"] = 3
    with pytest.raises(RuntimeError, match="cell 3 failed"):
        run(device=0, in_flight=9)


def test_operating_point_record_and_tags():
    """Op(I, SOC, T): the members the callers use (src/operating_point.py, src/batt_models/cellnr.py) - vector views,
    the display string that is written to battgpf_info.json, dataclass-like equality / repr, mutability; any object
    with .I / .SOC / .T compares equal (the reference's own Op can be handed to the plugin)."""
    from types import SimpleNamespace

    from battgp_amd.operating_point import Op, get_causal_tag, get_cell_tag

    op = Op(-15.0, SOC=90.0, T=25.0)
    assert op.disp_str() == "I = -15.00 A, SOC = 90.00 %, T = 25.00 °C"
    assert op.into_array().tolist() == [-15.0, 90.0, 25.0] and op.into_array().dtype == np.float64
    assert op.into_row_vector().shape == (1, 3)
    assert repr(op) == "Op(I=-15.0, SOC=90.0, T=25.0)"
    assert op == Op(-15.0, 90.0, 25.0) and op != Op(-15.0, 90.0, 26.0)
    assert op == SimpleNamespace(I=-15.0, SOC=90.0, T=25.0) and op != "x"
    op.T = 30.0
    assert op.into_array()[2] == 30.0
    assert [get_cell_tag(c) for c in (-1, 0, 12)] == ["pack", "c0", "c12"]
    assert (get_causal_tag(True), get_causal_tag(False)) == ("causal", "acausal")


def test_header_is_plain_c_and_a_c_program_drives_the_abi(tmp_path):
    """include/battgp.h compiles as pedantic C99 (it is the boundary an FFI binds, not a C++ header), and a plain-C
    caller (tests/cabi/c_caller.c) drives create / set_kernel / fit / predict / refit / lml_grad / destroy through it:
    linked against the CPU build of the kernel sources it reproduces the known answers of the reference's own unit
    test and the oracle's LML and gradient of a production-kernel problem; linked against the product library on this
    GPU-less box it must stop at bgp_create with an error code and a message - never compute anything."""
    import shutil
    import subprocess

    from battgp_amd import _lib, synthetic
    from oracle import kernels as K
    from oracle.exact_gp import lml_and_grad

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "tests", "cabi", "c_caller.c")
    flags = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), src]
    # (a) the product library: links (every symbol the program uses is exported), and without a GPU fails loudly
    if os.path.exists(_lib.LIB_PATH):
        exe = str(tmp_path / "c_caller_product")
        subprocess.run(flags + ["-L" + os.path.dirname(_lib.LIB_PATH), "-lbattgp", "-lm", "-o", exe], check=True, capture_output=True)
        try:
            import torch

            have_gpu = torch.cuda.is_available()
        except Exception:
            have_gpu = False
        if not have_gpu:
            env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(_lib.LIB_PATH) + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
            r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
            assert r.returncode == 77 and "bgp_create failed" in r.stderr and "c_caller ok" not in r.stdout, (r.returncode, r.stderr)
    # (b) the CPU build of the same sources: the program's checks, plus LML and gradient against the oracle
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("needs the ROCm host clang to build the CPU stand-in")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu

    emu_so = build_emu.build()
    exe = str(tmp_path / "c_caller_emu")
    name = os.path.splitext(os.path.basename(emu_so))[0][3:]
    subprocess.run(flags + ["-L" + os.path.dirname(emu_so), "-l" + name, "-lm", "-o", exe], check=True, capture_output=True)
    n, d = 90, 4
    x, y = synthetic.make_cell_data(n, seed=31)
    lml, grad = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
    args = [str(n), str(d), repr(float(lml))] + [repr(float(v)) for v in x.ravel()] + [repr(float(v)) for v in y] + [repr(float(v)) for v in synthetic.HYP_BATTGP]
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(emu_so) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c_caller ok" in r.stdout, r.stdout + r.stderr
    got = np.array([float(v) for v in r.stdout.splitlines()[0].split("grad")[1].split()])
    assert np.all(np.abs(got - grad) <= 1e-5 * np.abs(grad) + 1e-7 * np.abs(grad).max()), (got, grad)


def test_gradient_tests_sit_behind_the_core_gpu_modules():
    """The driver runs `pytest -m gpu -x`: nothing in the modules ordered BEFORE tests/test_gpu_grad.py may reach the
    LML-gradient kernels (the youngest device code), or a fault there leaves BASELINE configs 2 and 3 unreached.
    (The dynamic rehearsal of the same thing: `pytest -m gpu -x --emu --emu-fault bgp_lml_grad`)"""
    import ast

    import conftest

    order = conftest.GPU_ORDER
    assert "test_gpu_grad" in order
    core = order[: order.index("test_gpu_grad")]
    assert core[:3] == ["test_gpu_parity", "test_gpu_pins_and_sizes", "test_gpu_slab_layout"]
    reaches_gradient = {"lml_grad", "bgp_lml_grad", "train_hyperparameters", "optimize", "train_exact_gp_adam", "train_exact_gp_lbfgs",
                        "train_exact_gp_botorch", "bgp_grad_reduce_block_dev", "set_keep_factor"}
    for mod in core:
        tree = ast.parse(open(os.path.join(ROOT, "tests", mod + ".py")).read())
        used = {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute)} | {n.id for n in ast.walk(tree) if isinstance(n, ast.Name)}
        assert not (used & reaches_gradient), (mod, used & reaches_gradient)
        # (tests/problem_gen.py::check_problem evaluates the gradient only on request)
        assert not any(isinstance(n, ast.keyword) and n.arg == "grad" for n in ast.walk(tree)), mod
    # and the gradient module does hold them
    src = open(os.path.join(ROOT, "tests", "test_gpu_grad.py")).read()
    assert src.count("lml_grad()") >= 8
