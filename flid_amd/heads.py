"""Caller-side heads + losses of the trainers in fused-step form: `loss_fn(emb) -> (loss, d loss / d emb)` objects for
TGAT.train_step / MemoryModel.train_step, so that a whole trainer step -- backbone forward, head, loss, head backward, backbone
backward -- runs without an autograd graph.

replaces: PTCL/EM_warmup.py:212-231 (MergeLayer link predictor -> sigmoid -> nn.BCELoss -> backward: LinkPredictionLoss),
PTCL/M_step.py:285-318 (MLPClassifier -> masked / weighted cross entropy: ClassifierLoss) and, through `AutogradHeadLoss`, any other
head written with torch ops (MLPClassifier_BN, NPL/NPL.py:280-307)."""
import torch

from . import ops
from ._lib import check, lib


def _acc_grad(p, g):
    p.grad = g if p.grad is None else p.grad + g


class LinkPredictionLoss:
    """The link-prediction warm-up's head and loss on an embedding block [src | dst | negative dst] (3 B rows):
        p+ = sigmoid(head(src, dst)), p- = sigmoid(head(src, neg)),  loss = BCELoss(cat[p+, p-], cat[1, 0])      (EM_warmup.py:212-222)
    with head = MergeLayer(D, D, H, 1) = fc2(relu(fc1(cat[a, b]))) (models/modules.py:58-69).  Forward and backward are explicit
    products on the library's kernels (no autograd); the head's parameter gradients are ADDED to their .grad, the gradient w.r.t. the
    embedding block is returned -- the source rows collect both uses, as in the reference, which embeds them twice.  The reference
    computes the positive and the negative pair in two backbone calls (4 roots per edge); here the sources are embedded once."""

    def __init__(self, head):
        self.head = head
        self.loss_out = None

    def __call__(self, emb: torch.Tensor):
        fc1, fc2 = self.head.fc1, self.head.fc2
        n3, D = emb.shape
        B = n3 // 3
        assert n3 == 3 * B and fc1.weight.shape[1] == 2 * D and fc2.weight.shape[0] == 1
        H = fc1.weight.shape[0]
        dev = emb.device
        src, dst, neg = emb[:B], emb[B:2 * B], emb[2 * B:]
        X = torch.cat([torch.cat([src, dst], 1), torch.cat([src, neg], 1)], 0)                 # (2 B, 2 D): positive pairs, negative pairs
        h = torch.empty((2 * B, H), device=dev)
        ops.gemm(X, fc1.weight.detach(), h, tb=True, bias=fc1.bias.detach(), relu=True)
        z = torch.empty((2 * B, 1), device=dev)
        ops.gemm(h, fc2.weight.detach(), z, tb=True, bias=fc2.bias.detach())
        if self.loss_out is None or self.loss_out.device != dev:
            self.loss_out = torch.zeros(1, device=dev)
        dz = torch.empty((2 * B, 1), device=dev)
        check(lib().tg_bce_logits(ops._p(z), B, 2 * B, ops._p(self.loss_out), ops._p(dz), ops._stream()), "tg_bce_logits")
        # backward
        dW2 = torch.empty_like(fc2.weight)
        ops.gemm(dz, h, dW2, ta=True)                                                            # (1, H)
        dh = torch.empty((2 * B, H), device=dev)
        ops.gemm(dz, fc2.weight.detach(), dh)                                                    # (2 B, 1) x (1, H)
        ops.relu_bwd_(dh, h)
        dW1 = torch.empty_like(fc1.weight)
        ops.gemm(dh, X, dW1, ta=True)
        dX = torch.empty((2 * B, 2 * D), device=dev)
        ops.gemm(dh, fc1.weight.detach(), dX)
        _acc_grad(fc2.weight, dW2)
        _acc_grad(fc2.bias, ops.colsum(dz))
        _acc_grad(fc1.weight, dW1)
        _acc_grad(fc1.bias, ops.colsum(dh))
        d_emb = torch.cat([dX[:B, :D] + dX[B:, :D], dX[:B, D:], dX[B:, D:]], 0)
        return self.loss_out, d_emb


class PairLinkLoss:
    """One half of the link-prediction warm-up's loss on an embedding block [src | dst] (2 B rows) of ONE backbone call -- the form a
    memory model needs, whose negative and positive pairs come from two calls (PTCL/EM_warmup.py:159-175):
        p = sigmoid(head(src, dst)),  this call's share of BCELoss over the 2 B samples of the step = sum_i BCE(p_i, label) / (2 B)
    (`share` = B / (2 B) = 0.5 rescales the mean over this call's B samples).  Explicit forward / backward on the library's kernels;
    the head's parameter gradients are ADDED to their .grad; returns (loss share, d loss / d emb)."""

    def __init__(self, head, positive: bool, share: float = 0.5):
        self.head, self.positive, self.share = head, bool(positive), float(share)
        self.loss_out = None

    def __call__(self, emb: torch.Tensor):
        fc1, fc2 = self.head.fc1, self.head.fc2
        n2, D = emb.shape
        B = n2 // 2
        assert n2 == 2 * B and fc1.weight.shape[1] == 2 * D and fc2.weight.shape[0] == 1
        H = fc1.weight.shape[0]
        dev = emb.device
        X = torch.cat([emb[:B], emb[B:]], 1)                                                     # (B, 2 D)
        h = torch.empty((B, H), device=dev)
        ops.gemm(X, fc1.weight.detach(), h, tb=True, bias=fc1.bias.detach(), relu=True)
        z = torch.empty((B, 1), device=dev)
        ops.gemm(h, fc2.weight.detach(), z, tb=True, bias=fc2.bias.detach())
        if self.loss_out is None or self.loss_out.device != dev:
            self.loss_out = torch.zeros(1, device=dev)
        dz = torch.empty((B, 1), device=dev)
        check(lib().tg_bce_logits(ops._p(z), B if self.positive else 0, B, ops._p(self.loss_out), ops._p(dz), ops._stream()), "tg_bce_logits")
        if self.share != 1.0:
            dz.mul_(self.share)
        dW2 = torch.empty_like(fc2.weight)
        ops.gemm(dz, h, dW2, ta=True)
        dh = torch.empty((B, H), device=dev)
        ops.gemm(dz, fc2.weight.detach(), dh)
        ops.relu_bwd_(dh, h)
        dW1 = torch.empty_like(fc1.weight)
        ops.gemm(dh, X, dW1, ta=True)
        dX = torch.empty((B, 2 * D), device=dev)
        ops.gemm(dh, fc1.weight.detach(), dX)
        _acc_grad(fc2.weight, dW2)
        _acc_grad(fc2.bias, ops.colsum(dz))
        _acc_grad(fc1.weight, dW1)
        _acc_grad(fc1.bias, ops.colsum(dh))
        return self.loss_out * self.share, torch.cat([dX[:, :D], dX[:, D:]], 0).contiguous()


class ClassifierLoss:
    """The M-step's head and loss on the rows `rows` (a slice; default: the first len(labels) rows = the source embeddings,
    PTCL/M_step.py:285) of an embedding block:
        z = MLPClassifier(x) = fc3(drop(relu(fc2(drop(relu(fc1(x)))))))                                  (models/modules.py:72-97)
        loss = sum_i w_i CE(z_i, y_i)  over rows with y_i >= 0
    where the caller folds the trainer's masks and weights into w (PTCL/M_step.py:296-312: ground-truth rows 1 / n_gt, pseudo-labelled
    rows (1 - gt_weight) * exp(-alpha (patience_i - iter)) / n_ps, filtered rows 0 or y_i = -1).  Explicit forward / backward on
    the library's kernels; dropout masks are a hash of (seed, element) recomputed in the backward.  The head's parameter gradients
    are ADDED to their .grad; returns (loss, d loss / d emb) with zero rows outside `rows`.  `logits` holds z of the last call
    (the trainers collect predictions for their metrics)."""

    def __init__(self, head, labels: torch.Tensor, weights: torch.Tensor, rows=None, dropout=None):
        self.head, self.labels, self.weights, self.rows = head, labels.to(torch.int32).contiguous(), weights.to(torch.float32).contiguous(), rows
        self.p = float(head.dropout.p if dropout is None else dropout)
        self.loss_out, self.logits = None, None

    def __call__(self, emb: torch.Tensor):
        from . import engine
        fc1, fc2, fc3 = self.head.fc1, self.head.fc2, self.head.fc3
        n = self.labels.numel()
        rows = self.rows if self.rows is not None else slice(0, n)
        X = emb[rows]
        assert X.shape[0] == n and X.is_contiguous() and fc1.weight.shape[1] == X.shape[1]
        dev, C_ = emb.device, fc3.weight.shape[0]
        p = self.p if self.head.training else 0.0
        s1, s2 = engine._next_seeds(2) if p > 0 else (0, 0)
        h1 = torch.empty((n, fc1.weight.shape[0]), device=dev)
        ops.gemm(X, fc1.weight.detach(), h1, tb=True, bias=fc1.bias.detach(), relu=True)
        h1d = ops.dropout(h1, p, s1) if p > 0 else h1
        h2 = torch.empty((n, fc2.weight.shape[0]), device=dev)
        ops.gemm(h1d, fc2.weight.detach(), h2, tb=True, bias=fc2.bias.detach(), relu=True)
        h2d = ops.dropout(h2, p, s2) if p > 0 else h2
        z = torch.empty((n, C_), device=dev)
        ops.gemm(h2d, fc3.weight.detach(), z, tb=True, bias=fc3.bias.detach())
        if self.loss_out is None or self.loss_out.device != dev:
            self.loss_out = torch.zeros(1, device=dev)
        dz = torch.empty((n, C_), device=dev)
        check(lib().tg_weighted_ce(ops._p(z), C_, ops._p(self.labels), ops._p(self.weights), n, C_, ops._p(self.loss_out), ops._p(dz), C_,
                                   ops._stream()), "tg_weighted_ce")
        self.logits = z
        # backward
        dW3 = torch.empty_like(fc3.weight)
        ops.gemm(dz, h2d, dW3, ta=True)
        dh2 = torch.empty_like(h2)
        ops.gemm(dz, fc3.weight.detach(), dh2)
        if p > 0:
            dh2 = ops.dropout(dh2, p, s2)
        ops.relu_bwd_(dh2, h2)
        dW2 = torch.empty_like(fc2.weight)
        ops.gemm(dh2, h1d, dW2, ta=True)
        dh1 = torch.empty_like(h1)
        ops.gemm(dh2, fc2.weight.detach(), dh1)
        if p > 0:
            dh1 = ops.dropout(dh1, p, s1)
        ops.relu_bwd_(dh1, h1)
        dW1 = torch.empty_like(fc1.weight)
        ops.gemm(dh1, X, dW1, ta=True)
        d_emb = torch.zeros_like(emb)
        ops.gemm(dh1, fc1.weight.detach(), d_emb[rows])
        for lin, dW, dy in ((fc3, dW3, dz), (fc2, dW2, dh2), (fc1, dW1, dh1)):
            _acc_grad(lin.weight, dW)
            _acc_grad(lin.bias, ops.colsum(dy))
        return self.loss_out, d_emb


class AutogradHeadLoss:
    """Any head + loss written with torch ops (the M-step's MLPClassifier with its ground-truth / pseudo-label masks and per-sample
    weights, PTCL/M_step.py:285-318): the head runs as a small autograd island on the detached embedding block; its parameters get
    their .grad as usual, the backbone receives d loss / d emb and runs its backward without a graph.
    `fn(emb) -> scalar loss` (emb requires grad)."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, emb: torch.Tensor):
        x = emb.detach().requires_grad_(True)
        with torch.enable_grad():
            loss = self.fn(x)
            loss.backward()
        return loss.detach(), x.grad
