"""Mirror of the reference's building blocks (models/modules.py) -- same class names, constructor arguments, parameter
names and shapes (the state_dict contract of SURVEY.md 8b).  On the fused path the backbones read these modules'
parameters and run libflid_tg kernels; the modules' own forward() methods are HIP-backed too (ops.py), for callers that
use a block stand-alone."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMM (tg_gemm_f32), with its two transposed products in backward."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty((x2.shape[0], w.shape[0]), device=x.device)
        ops.gemm(x2, w, y, tb=True, bias=b, relu=relu)
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias, ctx.shape = b is not None, x.shape
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if y is not None:
            dy2 = ops.relu_bwd_(dy2.clone(), y)
        dx = torch.empty_like(x2)
        ops.gemm(dy2, w, dx)
        dw = torch.empty_like(w)
        ops.gemm(dy2, x2, dw, ta=True)
        db = ops.colsum(dy2) if ctx.has_bias else None
        return dx.reshape(ctx.shape), dw, db, None


def linear(x, w, b=None, relu=False):
    return _LinearFn.apply(x, w, b, relu)


class _TimeEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, w, b, fused):
        out = ops.time_encode(t, w.reshape(-1), b, fused_fma=fused)
        ctx.save_for_backward(t, w, b)
        return out

    @staticmethod
    def backward(ctx, g):
        t, w, b = ctx.saved_tensors
        dw, db = ops.time_encode_bwd(t, None, w.reshape(-1), b, g)      # timestamps are data: no gradient w.r.t. t
        return None, dw.reshape(w.shape), db, None


class TimeEncoder(nn.Module):
    """cos(w t + b), w_j = 10^(-9 j / (T-1))   (reference models/modules.py:7-40)"""

    def __init__(self, time_dim: int, parameter_requires_grad: bool = True):
        super().__init__()
        self.time_dim = time_dim
        self.w = nn.Linear(1, time_dim)
        self.w.weight = nn.Parameter((torch.from_numpy(1 / 10 ** np.linspace(0, 9, time_dim, dtype=np.float32))).reshape(time_dim, -1))
        self.w.bias = nn.Parameter(torch.zeros(time_dim))
        if not parameter_requires_grad:
            self.w.weight.requires_grad = False
            self.w.bias.requires_grad = False

    def forward(self, timestamps: torch.Tensor):
        # (batch, seq) -> (batch, seq, time_dim); t*w+b is rounded once (fma), as the reference's CPU Linear(1,T) does
        return _TimeEncodeFn.apply(timestamps.contiguous().float(), self.w.weight, self.w.bias, True)


class MergeLayer(nn.Module):
    """fc2(relu(fc1([a | b])))   (reference models/modules.py:43-69)"""

    def __init__(self, input_dim1: int, input_dim2: int, hidden_dim: int, output_dim: int):
        super().__init__()
        self.fc1 = nn.Linear(input_dim1 + input_dim2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self.act = nn.ReLU()

    def forward(self, input_1: torch.Tensor, input_2: torch.Tensor):
        x = torch.cat([input_1, input_2], dim=1)
        return linear(linear(x, self.fc1.weight, self.fc1.bias, relu=True), self.fc2.weight, self.fc2.bias)


class MLPClassifier(nn.Module):
    """decoder head (reference models/modules.py:72-97) -- caller-side, HIP GEMMs + torch dropout"""

    def __init__(self, input_dim: int, dropout: float = 0.1, num_classes: int = 2):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, 80)
        self.fc2 = nn.Linear(80, 10)
        self.fc3 = nn.Linear(10, num_classes)
        self.act = nn.ReLU()
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor):
        x = self.dropout(linear(x, self.fc1.weight, self.fc1.bias, relu=True))
        x = self.dropout(linear(x, self.fc2.weight, self.fc2.bias, relu=True))
        return linear(x, self.fc3.weight, self.fc3.bias)


class MultiHeadAttention(nn.Module):
    """Parameter holder with the reference's names/shapes (models/modules.py:126-165).  The backbones run it through the
    gather-fused kernels (flid_amd/engine.py); forward() on materialised inputs uses the same kernels with identity
    indices."""

    def __init__(self, node_feat_dim: int, edge_feat_dim: int, time_feat_dim: int, num_heads: int = 2, dropout: float = 0.1):
        super().__init__()
        self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim, self.num_heads = node_feat_dim, edge_feat_dim, time_feat_dim, num_heads
        self.query_dim = node_feat_dim + time_feat_dim
        self.key_dim = node_feat_dim + edge_feat_dim + time_feat_dim
        assert self.query_dim % num_heads == 0, "The sum of node_feat_dim and time_feat_dim should be divided by num_heads!"
        self.head_dim = self.query_dim // num_heads
        self.query_projection = nn.Linear(self.query_dim, num_heads * self.head_dim, bias=False)
        self.key_projection = nn.Linear(self.key_dim, num_heads * self.head_dim, bias=False)
        self.value_projection = nn.Linear(self.key_dim, num_heads * self.head_dim, bias=False)
        self.scaling_factor = self.head_dim ** -0.5
        self.layer_norm = nn.LayerNorm(self.query_dim)
        self.residual_fc = nn.Linear(num_heads * self.head_dim, self.query_dim)
        self.dropout = nn.Dropout(dropout)

    def fused_params(self):
        return [self.query_projection.weight, self.key_projection.weight, self.value_projection.weight,
                self.layer_norm.weight, self.layer_norm.bias, self.residual_fc.weight, self.residual_fc.bias]
