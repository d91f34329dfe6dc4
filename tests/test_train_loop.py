"""learning-rate schedule and checkpoint bookkeeping of the training driver (host logic; the GPU run is in
tests/test_gpu_train_loop.py)"""
import importlib
import types

import pytest
import torch

solver = importlib.import_module("3dod_amd.cubercnn.solver")
syn = importlib.import_module("3dod_amd.synthetic")


def _opt():
    return types.SimpleNamespace(lr_scale=1.0, param_groups=[{"lr": 0.02}, {"lr": 0.04}])


def test_warmup_multistep_values():
    cfg = syn.make_cfg()                      # Base.yaml: steps (19200, 25600), max_iter 32000, warmup 1000? (see below)
    o = _opt()
    s = solver.WarmupMultiStepLR(o, [19200, 25600], 0.1, 0.001, 1000, "linear", 32000)
    assert s.factor(0) == pytest.approx(0.001) and s.factor(500) == pytest.approx(0.5005) and s.factor(1000) == 1.0
    assert s.factor(19199) == 1.0 and s.factor(19200) == pytest.approx(0.1) and s.factor(25600) == pytest.approx(0.01)
    assert o.lr_scale == pytest.approx(0.001)                 # constructed at iteration 0
    for _ in range(500):
        s.step()
    assert s.last_iter == 500 and o.lr_scale == pytest.approx(0.5005) and s.get_last_lr() == pytest.approx([0.02 * 0.5005, 0.04 * 0.5005])
    # resume
    o2 = _opt()
    s2 = solver.WarmupMultiStepLR(o2, [19200, 25600], 0.1, 0.001, 1000, "linear", 32000)
    s2.load_state_dict(s.state_dict())
    assert s2.last_iter == 500 and o2.lr_scale == pytest.approx(0.5005)
    # milestones past MAX_ITER are dropped; a milestone inside the warm-up is interpolated towards
    assert solver.WarmupMultiStepLR(_opt(), [10, 50], 0.1, 0.001, 0, "linear", 20).factor(60) == pytest.approx(0.1)
    w = solver.WarmupMultiStepLR(_opt(), [5], 0.1, 0.5, 10, "linear", 100)
    assert w.factor(0) == pytest.approx(0.5) and w.factor(10) == pytest.approx(0.1) and w.factor(5) == pytest.approx(0.5 * 0.5 + 0.1 * 0.5)
    assert solver.WarmupMultiStepLR(_opt(), [5], 0.1, 0.3, 10, "constant", 100).factor(7) == pytest.approx(0.3)
    with pytest.raises(ValueError):
        solver.WarmupMultiStepLR(_opt(), [5, 3])
    # from the config, like tools/train_net.py:135
    sch = solver.build_lr_scheduler(cfg, _opt())
    assert isinstance(sch, solver.WarmupMultiStepLR) and sch.milestones == [s for s in cfg.SOLVER.STEPS if s <= cfg.SOLVER.MAX_ITER]
    cfg.SOLVER.LR_SCHEDULER_NAME = "WarmupCosineLR"
    c = solver.build_lr_scheduler(cfg, _opt())
    assert c.factor(cfg.SOLVER.MAX_ITER) == pytest.approx(0.0, abs=1e-9) and 0.49 < c.factor(cfg.SOLVER.MAX_ITER // 2) < 0.51
    cfg.SOLVER.LR_SCHEDULER_NAME = "nope"
    with pytest.raises(ValueError):
        solver.build_lr_scheduler(cfg, _opt())


def test_checkpointer_files(tmp_path):
    model = torch.nn.Linear(3, 2)
    ck = solver.Checkpointer(model, str(tmp_path))
    assert not ck.has_checkpoint()
    per = solver.PeriodicCheckpointerOnlyOne(ck, 3, max_iter=7)
    for it in range(7):
        per.step(it)
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["last_checkpoint", "model_final.pth", "model_recent.pth"]
    assert ck.get_checkpoint_file().endswith("model_final.pth")
    w = model.weight.detach().clone()
    with torch.no_grad():
        model.weight.zero_()
    data = ck.resume_or_load("", resume=True)
    assert data["iteration"] == 6 and torch.equal(model.weight, w)
    # rank != 0 writes nothing
    other = solver.Checkpointer(model, str(tmp_path / "r1"), save_to_disk=False)
    assert other.save("x") is None and not (tmp_path / "r1").exists()


def test_checkpointer_reports_incompatible_keys(tmp_path, caplog):
    """a pretrained checkpoint whose keys do not match must not load silently as a random initialisation: missing /
    unexpected / wrongly shaped keys are logged and recorded, a checkpoint that matches nothing is an error"""
    import logging
    model = torch.nn.Sequential(torch.nn.Linear(3, 2), torch.nn.Linear(2, 2))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    part = {"0.weight": sd["0.weight"] + 1.0, "0.bias": torch.zeros(5), "renamed.weight": sd["1.weight"]}
    torch.save({"model": part}, str(tmp_path / "part.pth"))
    ck = solver.Checkpointer(model, str(tmp_path))
    with caplog.at_level(logging.WARNING):
        ck.load(str(tmp_path / "part.pth"), checkpointables=[])
    assert torch.equal(model.state_dict()["0.weight"], sd["0.weight"] + 1.0)
    assert set(ck.incompatible["missing"]) == {"0.bias", "1.weight", "1.bias"} and ck.incompatible["unexpected"] == ["renamed.weight"]
    text = caplog.text
    assert "not in the checkpoint" in text and "1.weight" in text and "renamed.weight" in text and "shape of '0.bias'" in text
    torch.save({"model": {"module." + k: v for k, v in sd.items()}}, str(tmp_path / "none.pth"))
    with pytest.raises(RuntimeError, match="matches none"):
        ck.load(str(tmp_path / "none.pth"), checkpointables=[])
