"""Import the read-only reference (/root/reference) in THIS container only.

Used by tools/gen_golden.py to produce the fixtures under tests/golden/.  Nothing here (and
nothing it imports from /root/reference) ever travels to the GPU box or is used at test /
bench / product run time: the committed artefacts are the .npz vectors only.

The reference's python dependencies detectron2 / cupy / timm are absent from this image, so
the *import-time* names it needs from them are bound to minimal behavioural equivalents:
  detectron2.config.configurable          cfg-or-kwargs ctor decorator (calls cls.from_config)
  detectron2.utils.registry.Registry      name -> class table
  detectron2.utils.env.TORCH_VERSION      tuple
  detectron2.layers.{NaiveSyncBatchNorm,FrozenBatchNorm2d}   never instantiated on this path
  cupy.memoize / cupy.cuda / cupy.int32   decorators only; the cupy splat itself is CUDA-only
                                          and is NOT runnable here (parity for it: see DESIGN.md)
  timm (+ timm.models.efficientnet_blocks) import-time names only; backbone is out of scope
None of these touch the arithmetic of the hot path, which is pure torch ops.
"""
import functools
import inspect
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TS_REFERENCE_ROOT", "/root/reference")


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _configurable(init_func=None, *, from_config=None):
    # same calling convention as detectron2's decorator: Class(cfg) -> Class(**from_config(cfg))
    def _looks_like_cfg(args, kwargs):
        a = args[0] if args else kwargs.get("cfg", None)
        return a is not None and hasattr(a, "MODEL") and (len(args) + len(kwargs)) == 1

    assert init_func is not None and inspect.isfunction(init_func)

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        if _looks_like_cfg(args, kwargs):
            cfg = args[0] if args else kwargs["cfg"]
            init_func(self, **type(self).from_config(cfg))
        else:
            init_func(self, *args, **kwargs)

    return wrapped


class _Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(cls):
                self[cls.__name__] = cls
                return cls
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        if name not in self:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return self[name]


def install_stubs():
    import torch

    if "detectron2" in sys.modules:
        return
    _module("detectron2")
    _module("detectron2.config", configurable=_configurable)
    _module("detectron2.utils")
    _module("detectron2.utils.registry", Registry=_Registry)
    tv = tuple(int(x) for x in torch.__version__.split("+")[0].split(".")[:2])
    _module("detectron2.utils.env", TORCH_VERSION=tv)
    sys.modules["detectron2.utils"].env = sys.modules["detectron2.utils.env"]
    _module("detectron2.layers", NaiveSyncBatchNorm=torch.nn.BatchNorm2d,
            FrozenBatchNorm2d=torch.nn.BatchNorm2d)

    def _memoize(**_kw):
        return lambda f: f
    cuda = types.SimpleNamespace(compile_with_cache=None)
    _module("cupy", memoize=_memoize, cuda=cuda, int32=int)

    class _Missing:  # placeholder names for the (out-of-scope) backbone import
        def __init__(self, *a, **k):
            raise RuntimeError("timm is not available; the backbone is out of scope")
    _module("timm", create_model=_Missing)
    _module("timm.models")
    _module("timm.models.efficientnet_blocks", InvertedResidual=_Missing, drop_path=None)


def import_reference():
    """Returns the imported `architecture.modeling` package of the reference."""
    sys.dont_write_bytecode = True
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import architecture.modeling as M  # noqa
    return M
