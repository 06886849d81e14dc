"""Host-side logic of temporalstereo_amd.train that needs no GPU."""
import os

import pytest
import torch


def test_package_import_leaves_the_environment_alone(monkeypatch):
    import importlib
    import temporalstereo_amd
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    importlib.reload(temporalstereo_amd)
    # ROCm 7.2 replays pre-built graph packets incorrectly for graphs of the training step's size (train.py); switching that
    # off is an explicit opt-in (train.enable_graph_replay), not a side effect of importing the library
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is None


def test_enable_graph_replay_is_explicit_and_checked(monkeypatch):
    from temporalstereo_amd import train
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    monkeypatch.setattr(train, "_set_before_runtime", False)
    monkeypatch.setattr(train, "_initial_environment", lambda: {})
    assert not train.graph_replay_safe()
    # the variable alone is not enough: it must have been there before the runtime started
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    assert not train.graph_replay_safe()
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        train.TrainStep(torch.nn.Linear(2, 2), graph=True)
    monkeypatch.setattr(train, "_initial_environment", lambda: {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"})
    assert train.graph_replay_safe()
    monkeypatch.setattr(train, "_initial_environment", lambda: {})
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
    assert not torch.cuda.is_initialized()
    train.enable_graph_replay()                      # no GPU call yet in this process: effective
    assert os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0" and train.graph_replay_safe()
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    monkeypatch.setattr(train, "_set_before_runtime", False)
    with pytest.raises(RuntimeError, match="before the first GPU call"):
        train.enable_graph_replay()


def test_train_step_defaults_are_the_reference_configuration():
    # sceneflow.yaml:21-24, :61-67 (the objective must be the reference's)
    from temporalstereo_amd.train import TrainStep
    step = TrainStep(torch.nn.Linear(2, 2))
    assert step.l1.weights == [2.0, 1.0, 0.7, 0.5] and step.l1.global_weight == 1.0
    assert step.wars.weights == [1.0, 0.7, 0.5] and step.wars.global_weight == 2.0
    assert step.opt.param_groups[0]["lr"] == 1e-3
    custom = TrainStep(torch.nn.Linear(2, 2), lr=3e-4, l1_weights=None, wars_weights=(1.0,), wars_global_weight=1.0)
    assert custom.l1.weights is None and custom.wars.weights == [1.0] and custom.opt.param_groups[0]["lr"] == 3e-4


def test_graph_mode_refuses_an_unsafe_runtime(monkeypatch):
    from temporalstereo_amd.train import TrainStep
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        TrainStep(torch.nn.Linear(2, 2), graph=True)


def test_weight_layouts_requires_the_library_path_only_on_use():
    # constructing the per-step layout cache touches neither the GPU nor the library
    from temporalstereo_amd import functional as TF
    w = TF.WeightLayouts()
    assert w.entries == {} and w.lazy is False
    w.refresh()                      # nothing registered: no launch, no error
    assert w.epoch == 1
