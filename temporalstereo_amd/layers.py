"""conv -> norm -> activation wrappers with the reference's constructor contract and state-dict
names (architecture/modeling/layers/basic_layers.py:10-103,151-235,289-388): `Conv3d(*conv_args,
norm=(name, channels) | module | None, activation=str | (str, coeff) | module | None)`; parameters
live at `<name>.weight/.bias` and `<name>.norm.*`.

On the GPU every wrapper of the model runs on the HIP kernels, forward and backward: the two separable 3-D families ((1,3,3) and
(k,1,1), their stride-2 and transposed forms), 3x3 Conv2d (the (1,3,3) family on one plane) and ConvTranspose2d(4, stride 2, padding 1)
(forward kernel of its own, backward as a 3x3 convolution of the space-to-depth gradient).  conv -> BatchNorm -> activation is ONE
autograd node (functional.conv_bn_act): train-mode batch statistics from the BatchNorm kernels (exchangeable between launches:
dist.SyncBatchNorm), running statistics in eval mode, and inside `functional.BNFolds` (the previous frames of a training step) the
BatchNorm folded into the convolution's epilogue.  Whole-model eval execution with folded weights, fused concatenations and a recorded
launch plan happens one level up, in aggregation.native.  CPU tensors (host-logic tests) take torch's own operators, as does anything
outside the kernels' limits (groups, other kernel sizes, GroupNorm / InstanceNorm, momentum=None).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_CONV_BACKEND = {"name": "hip"}


def set_conv_backend(name):
    """'hip' (default): GPU tensors use the HIP convolution Functions; 'torch': F.conv3d (MIOpen) everywhere --
    kept so that benchmarks can time the framework's own kernels next to ours."""
    if name not in ("hip", "torch"):
        raise ValueError("conv backend must be 'hip' or 'torch'")
    _CONV_BACKEND["name"] = name


def _hip_conv(x):
    return _CONV_BACKEND["name"] == "hip" and x.is_cuda and x.dtype == torch.float32

_NORMS = {
    "BN1d": nn.BatchNorm1d, "BN": nn.BatchNorm2d, "BN3d": nn.BatchNorm3d,
    "IN1d": nn.InstanceNorm1d, "IN": nn.InstanceNorm2d, "IN3d": nn.InstanceNorm3d,
    "SyncBN": nn.SyncBatchNorm, "nnSyncBN": nn.SyncBatchNorm,
    "GN": lambda c: nn.GroupNorm(32, c),
}
_ACTS = {
    "ReLU": lambda a: nn.ReLU(inplace=True),
    "LeakyReLU": lambda a: nn.LeakyReLU(negative_slope=0.1 if a is None else a, inplace=True),
    "ELU": lambda a: nn.ELU(alpha=1.0 if a is None else a, inplace=True),
    "SELU": lambda a: nn.SELU(inplace=True),
    "SiLU": lambda a: nn.SiLU(inplace=True),
    "Hardswish": lambda a: nn.Hardswish(inplace=True),
    "Mish": lambda a: nn.Mish(inplace=True),
}


def get_norm(norm, out_channels):
    """basic_layers.py:10-39."""
    if norm is None or (isinstance(norm, str) and not norm):
        return None
    if isinstance(norm, str):
        if norm not in _NORMS:
            raise KeyError("unknown norm '%s'" % norm)
        return _NORMS[norm](out_channels)
    return norm(out_channels)


def get_activation(activation, coeff=None):
    """basic_layers.py:42-73."""
    if activation is None or (isinstance(activation, str) and not activation):
        return None
    if isinstance(activation, str):
        if activation not in _ACTS:
            raise KeyError("unknown activation '%s'" % activation)
        return _ACTS[activation](coeff)
    return activation


def _split_kwargs(kwargs):
    """basic_layers.py:76-103: pops `norm` / `activation` and builds the modules."""
    norm = kwargs.pop("norm", None)
    if isinstance(norm, (tuple, list)):
        if len(norm) != 2:
            raise ValueError("norm must be (name, channels)")
        norm = get_norm(*norm)
    act = kwargs.pop("activation", None)
    if isinstance(act, (tuple, list)):
        if len(act) not in (1, 2):
            raise ValueError("activation must be name | (name,) | (name, coeff)")
        act = get_activation(act[0], act[1] if len(act) == 2 else None)
    elif isinstance(act, str):
        act = get_activation(act)
    return norm, act, kwargs


class _NormAct:
    def _finish(self, x):
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x

    def _fusable(self):
        """Name of the activation when conv -> BatchNorm -> activation can run as the one fused HIP node
        (functional.conv_bn_act: BatchNorm with affine parameters and running statistics; no activation, SiLU or ReLU)."""
        bn = self.norm
        if not isinstance(bn, nn.modules.batchnorm._BatchNorm) or not bn.affine or not bn.track_running_stats:
            return False
        if bn.momentum is None:           # cumulative moving average: 1 / num_batches_tracked per step -- the framework's module does that
            return False
        if self.activation is None:
            return None
        if isinstance(self.activation, nn.SiLU):
            return "SiLU"
        if isinstance(self.activation, nn.ReLU):
            return "ReLU"
        return False


def _static(m, make):
    """Per-module memo of what depends only on the constructor arguments and on WHICH norm / activation modules are attached (not
    on their state): the kernel family, its geometry tuple and the fused activation's name.  A training step walks ~180 wrappers per
    frame; recomputing these (tuple building, isinstance chains, nn.Module.__getattr__ for every parameter and submodule) was ~1 ms
    of its host time."""
    d = m.__dict__
    mods = d["_modules"]
    key = (id(mods.get("norm")), id(mods.get("activation")))
    memo = d.get("_ts_static")
    if memo is None or memo[0] != key:
        memo = d["_ts_static"] = (key, make())
    return memo[1]


class Conv2d(nn.Conv2d, _NormAct):
    def __init__(self, *args, **kwargs):
        norm, act, kwargs = _split_kwargs(kwargs)
        super().__init__(*args, **kwargs)
        self.norm, self.activation = norm, act

    def forward(self, x):
        if _hip_conv(x):
            from . import functional as TF

            def make():
                st, pd, dl = (1,) + tuple(self.stride), (0,) + tuple(self.padding), (1,) + tuple(self.dilation)
                w5 = (self.weight.shape[0], self.weight.shape[1], 1) + tuple(self.weight.shape[2:])
                return TF.conv3d_supported(w5, st, pd, dl, self.groups), self._fusable(), st, pd, dl
            kind, act, st, pd, dl = _static(self, make)
            if kind == "hw":      # a (1,3,3) convolution on one plane
                w5 = self._parameters["weight"].unsqueeze(2)
                if act is not False:
                    return TF.conv_bn_act(x.unsqueeze(2), w5, self._parameters["bias"], self._modules["norm"], act, "hw",
                                          (st[1], dl[1], False)).squeeze(2)
                return self._finish(TF.conv3d(x.unsqueeze(2), w5, self.bias, st, pd, dl).squeeze(2))
        return self._finish(F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups))


class Conv3d(nn.Conv3d, _NormAct):
    def __init__(self, *args, **kwargs):
        norm, act, kwargs = _split_kwargs(kwargs)
        super().__init__(*args, **kwargs)
        self.norm, self.activation = norm, act

    def forward_then(self, x, activation):
        """forward(x) followed by `activation` ('SiLU' | 'ReLU') for a wrapper that has none of its own -- on the HIP path inside the
        same conv -> BatchNorm -> activation node (ResidualBlock3D applies F.silu to conv4's output, module.py:276)."""
        if _hip_conv(x) and self.activation is None:
            from . import functional as TF
            kind = TF.conv3d_supported(tuple(self.weight.shape), self.stride, self.padding, self.dilation, self.groups)
            if kind and self._fusable() is None:
                geom = (self.stride[1], self.dilation[1], False) if kind == "hw" else (self.stride[0], self.dilation[0], self.padding[0], False)
                return TF.conv_bn_act(x, self._parameters["weight"], self._parameters["bias"], self._modules["norm"], activation, kind, geom)
        y = self.forward(x)
        return F.silu(y) if activation == "SiLU" else F.relu(y)

    def forward(self, x):
        if _hip_conv(x):
            from . import functional as TF

            def make():
                kind = TF.conv3d_supported(tuple(self.weight.shape), self.stride, self.padding, self.dilation, self.groups)
                geom = (self.stride[1], self.dilation[1], False) if kind == "hw" else \
                    (self.stride[0], self.dilation[0], self.padding[0], False)
                return kind, self._fusable(), geom
            kind, act, geom = _static(self, make)
            if kind:
                if act is not False:      # conv -> BatchNorm (train or eval statistics) -> activation as one autograd node
                    return TF.conv_bn_act(x, self._parameters["weight"], self._parameters["bias"], self._modules["norm"], act, kind, geom)
                return self._finish(TF.conv3d(x, self.weight, self.bias, self.stride, self.padding, self.dilation))
        return self._finish(F.conv3d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups))


class ConvTranspose2d(nn.ConvTranspose2d, _NormAct):
    def __init__(self, *args, **kwargs):
        norm, act, kwargs = _split_kwargs(kwargs)
        super().__init__(*args, **kwargs)
        self.norm, self.activation = norm, act

    def forward(self, x, output_size=None):
        if self.padding_mode != 'zeros':
            raise ValueError('Only `zeros` padding mode is supported for ConvTranspose2d')
        op = self._output_padding(x, output_size, self.stride, self.padding, self.kernel_size, 2, self.dilation)
        if _hip_conv(x):
            from . import functional as TF
            if TF.deconv2d_k4s2_supported(tuple(self.weight.shape), self.stride, self.padding, tuple(op), self.dilation, self.groups):
                act = self._fusable()
                if act is not False:      # UNet.deconv4 (module.py:453-456): transposed conv -> BatchNorm -> ReLU as one autograd node
                    return TF.conv_bn_act(x, self.weight, self.bias, self.norm, act, "dc", None)
                return self._finish(TF.conv_transpose2d_k4s2(x, self.weight, self.bias))
        return self._finish(F.conv_transpose2d(x, self.weight, self.bias, self.stride, self.padding, op,
                                               self.groups, self.dilation))


class ConvTranspose3d(nn.ConvTranspose3d, _NormAct):
    def __init__(self, *args, **kwargs):
        norm, act, kwargs = _split_kwargs(kwargs)
        super().__init__(*args, **kwargs)
        self.norm, self.activation = norm, act

    def forward(self, x, output_size=None):
        if self.padding_mode != 'zeros':
            raise ValueError('Only `zeros` padding mode is supported for ConvTranspose3d')
        op = self._output_padding(x, output_size, self.stride, self.padding, self.kernel_size, 3, self.dilation)
        if _hip_conv(x):
            from . import functional as TF
            kind = TF.conv3d_supported(tuple(self.weight.shape), self.stride, self.padding, self.dilation, self.groups, True, tuple(op))
            if kind:
                act = self._fusable()
                if act is not False:
                    geom = (2, 1, True) if kind == "hw" else (2, 1, 1, True)
                    return TF.conv_bn_act(x, self.weight, self.bias, self.norm, act, kind, geom)
                return self._finish(TF.conv_transpose3d(x, self.weight, self.bias, self.stride, self.padding, tuple(op)))
        return self._finish(F.conv_transpose3d(x, self.weight, self.bias, self.stride, self.padding, op,
                                               self.groups, self.dilation))
