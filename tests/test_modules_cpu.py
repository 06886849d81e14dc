"""CPU: the host-side mirror of the reference interface -- registry/config seam, state-dict
compatibility (names and shapes recorded from the real reference modules by tools/gen_golden.py)."""
import pytest
import torch

import temporalstereo_amd as ts
from helpers import state_shapes


def _sceneflow_cfg():
    return ts.CfgView({"MODEL": {"AGGREGATION": {
        "NAME": "TEMPORALSTEREO",
        "COARSE": {"IN_PLANES": 256, "C": 32, "NUM_SAMPLE": 12, "DELTA": 1.0, "BLOCK_COST_SCALE": 3, "TOPK": 2,
                   "SPATIAL_FUSION": True},
        "FINE": {"IN_PLANES": 128, "C": 16, "NUM_SAMPLE": 5, "DELTA": 1.0, "BLOCK_COST_SCALE": 3, "TOPK": 2,
                 "SPATIAL_FUSION": True},
        "PRECISE": {"IN_PLANES": 64, "C": 8, "NUM_SAMPLE": 5, "DELTA": 1.0, "BLOCK_COST_SCALE": 3, "TOPK": 2}}}})


def test_build_aggregation_from_config_matches_reference_state_dict():
    net = ts.build_aggregation(_sceneflow_cfg())
    assert isinstance(net, ts.TEMPORALSTEREO)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == state_shapes("32x16x8")          # sceneflow.yaml dims, recorded from the reference
    assert sum(p.numel() for p in net.parameters()) == 1041986


def test_tiny_dims_state_dict_and_strict_load():
    net = ts.TEMPORALSTEREO(coarse=ts.CoarseAggregation(32, 8, 4), fine=ts.FineAggregation(16, 8, 5),
                            precise=ts.PreciseAggregation(8, 8, 5))
    ref = state_shapes("8x8x8")
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == ref
    net.load_state_dict({k: torch.zeros(s) if "num_batches" not in k else torch.zeros(s, dtype=torch.long)
                         for k, s in ref.items()}, strict=True)


def test_registry_and_prediction_modules():
    assert "TEMPORALSTEREO" in ts.AGGREGATION_REGISTRY
    with pytest.raises(KeyError):
        ts.AGGREGATION_REGISTRY.get("NOPE")
    cfg = ts.CfgView({"MODEL": {"PREDICTION": {"NAME": "SOFTARGMIN", "TEMPERATURE": 2.0}}})
    p = ts.build_prediction(cfg)
    assert isinstance(p, ts.SOFTARGMIN) and p.temperature == 2.0 and p.normalize is True
    assert ts.SOFTARGMIN(temperature=0.5).temperature == 0.5

    class FakeRefRegistry:
        def __init__(self): self.items = {}
        def register(self, cls): self.items[cls.__name__] = cls; return cls
    r = ts.register_into(FakeRefRegistry())
    assert "TEMPORALSTEREO_HIP" in r.items and issubclass(r.items["TEMPORALSTEREO_HIP"], ts.TEMPORALSTEREO)


def test_reference_initialiser_statistics():
    torch.manual_seed(0)
    lvl = ts.CoarseAggregation(64, 16, 4)
    w = lvl.init3d[0].conv[0].weight            # (1,3,3) kernel, Cout=16 -> std = sqrt(2/(9*16))
    assert abs(float(w.std()) - (2.0 / (9 * 16)) ** 0.5) < 0.01
    bn = lvl.init3d[0].conv[0].norm
    assert float(bn.weight.min()) == 1.0 and float(bn.bias.abs().max()) == 0.0


def test_layers_constructor_contract():
    from temporalstereo_amd.layers import Conv3d, get_activation, get_norm
    c = Conv3d(4, 8, 3, 1, 1, bias=False, norm=('BN3d', 8), activation=('LeakyReLU', 0.2))
    assert isinstance(c.norm, torch.nn.BatchNorm3d) and c.activation.negative_slope == 0.2
    assert get_norm(None, 4) is None and get_activation("") is None
    with pytest.raises(KeyError):
        get_norm("nope", 4)
    y = c(torch.randn(2, 4, 3, 5, 5))
    assert y.shape == (2, 8, 3, 5, 5)
