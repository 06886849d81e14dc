"""Host-side logic of temporalstereo_amd.train that needs no GPU."""
import os

import pytest
import torch


def test_package_import_switches_graph_packet_capture_off():
    import temporalstereo_amd  # noqa: F401
    # ROCm 7.2 replays pre-built graph packets incorrectly for graphs of the training step's size (train.py); the package
    # sets the runtime switch when nothing else has
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is not None


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
