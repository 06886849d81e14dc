"""Plug-in seam of the hot path.

The reference builds its aggregator with detectron2's Registry + @configurable
(architecture/modeling/aggregation/builder.py:3-20,
 architecture/modeling/aggregation/TemporalStereo/TemporalStereo.py:14,23,38-78).  This module gives
the same two mechanisms without detectron2, and `register_into(reference_registry)` adds our classes
to the reference's own registry object when both live in one process.
"""
import functools


class Registry:
    """name -> class table with the register()/get() calls the reference uses."""

    def __init__(self, name):
        self._name = name
        self._table = {}

    def register(self, obj=None, name=None):
        def _add(cls):
            key = name or cls.__name__
            if key in self._table and self._table[key] is not cls:
                raise KeyError("'%s' is already registered in %s" % (key, self._name))
            self._table[key] = cls
            return cls
        return _add if obj is None else _add(obj)

    def get(self, name):
        try:
            return self._table[name]
        except KeyError:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name)) from None

    def __contains__(self, name):
        return name in self._table

    def names(self):
        return sorted(self._table)


def configurable(init):
    """`Class(cfg)` -> `Class(**Class.from_config(cfg))`; explicit kwargs pass straight through."""
    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        first = args[0] if args else kwargs.get("cfg")
        if first is not None and (len(args) + len(kwargs)) == 1 and hasattr(first, "MODEL"):
            init(self, **type(self).from_config(first))
        else:
            init(self, *args, **kwargs)
    return wrapper


AGGREGATION_REGISTRY = Registry("AGGREGATION")
PREDICTION_REGISTRY = Registry("PREDICTION")


def build_aggregation(cfg):
    """cfg.MODEL.AGGREGATION.NAME selects the class (aggregation/builder.py:12-20)."""
    return AGGREGATION_REGISTRY.get(cfg.MODEL.AGGREGATION.NAME)(cfg)


def build_prediction(cfg):
    return PREDICTION_REGISTRY.get(cfg.MODEL.PREDICTION.NAME)(cfg)


def register_into(reference_registry, registry=AGGREGATION_REGISTRY, suffix="_HIP"):
    """Add our classes to a detectron2-style registry of the reference under NAME+suffix."""
    for key in registry.names():
        cls = registry.get(key)
        alias = type(key + suffix, (cls,), {})
        reference_registry.register(alias)
    return reference_registry


class CfgView(dict):
    """Tiny attribute/dict config node (`cfg.MODEL.AGGREGATION.COARSE.get('C', 32)`), enough to drive
    from_config() without fvcore/yacs.  Keys follow projects/TemporalStereo/config.py:120-160."""

    def __init__(self, mapping=None):
        super().__init__()
        for k, v in (mapping or {}).items():
            self[k] = CfgView(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None
