"""PyTorch-ROCm counterpart of /root/reference model/loss.py (PPLoss).

Same values and gradients as model/loss.py:24-63 (pinned by
tests/golden/model_golden.npz); the only structural change is that the tanh on
regression channel 6 is not written back in place into the model's output
(model/loss.py:50) -- the forward value and gradient are identical.  Note that
the reference's tanh touches channel 6 of the whole A_c*8 channel axis (the
first anchor's dt only); that behaviour is kept.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PPLoss(nn.Module):
    """focal-weighted BCE (alpha 25 on positives, gamma 2, weights detached) +
    smooth-L1 over the positives' 7 regression dims + orientation BCE;
    total = b_cls*cls + b_reg*reg + b_ort*ort (config.py:144-146: 250, 1, 0)."""

    def __init__(self, b_ort=0.0, b_reg=1.0, b_cls=250.0, gamma=2.0, reg_dims=8):
        super().__init__()
        self.b_ort, self.b_reg, self.b_cls, self.gamma = b_ort, b_reg, b_cls, gamma
        self.reg_dims = reg_dims

    def forward(self, cls_tensor, reg_tensor, cls_targets, reg_targets):
        # cls_tensor [B, A_c*9, H, W] -> [B, H*W*A_c*9]   (model/loss.py:31-36)
        cls_tensor = cls_tensor.permute(0, 2, 3, 1)
        bsz = cls_tensor.size(0)
        cls_tensor = cls_tensor.reshape(bsz, -1)
        cls_targets = cls_targets.reshape(bsz, -1)
        p = torch.sigmoid(cls_tensor)
        is_pos = cls_targets == 1
        pt = torch.where(is_pos, p, 1 - p)
        at = torch.where(is_pos, torch.full_like(p, 25.0), torch.ones_like(p))   # :40
        w = (at * (1 - pt) ** self.gamma).detach()                               # :43
        cls_loss = F.binary_cross_entropy_with_logits(cls_tensor, cls_targets, weight=w)

        reg_tensor = reg_tensor.permute(0, 2, 3, 1)
        # model/loss.py:50 applies tanh to index 6 of the PERMUTED channel axis,
        # i.e. to channel 6 of all A_c*8 channels -- the first anchor's dt only.
        # Reproduced as is (bug-compatible), before the reshape to (..., 8).
        reg_tensor = torch.cat((reg_tensor[..., :6], torch.tanh(reg_tensor[..., 6:7]),
                                reg_tensor[..., 7:]), dim=-1)
        reg_tensor = reg_tensor.reshape(bsz, -1, self.reg_dims)
        pos = reg_targets[..., 0] == 1                                           # :53
        reg_scores = reg_tensor[pos][..., :7]
        loss_targs = reg_targets[pos][..., 1:8]
        # mean over an empty selection is NaN, exactly like the reference
        reg_loss = F.smooth_l1_loss(reg_scores, loss_targs, reduction="mean")
        ort_scores = reg_tensor[pos][..., 7]
        ort_targets = reg_targets[pos][..., 8]
        ort_loss = F.binary_cross_entropy_with_logits(ort_scores, ort_targets)
        total = self.b_cls * cls_loss + self.b_reg * reg_loss + self.b_ort * ort_loss
        return p, cls_loss, reg_loss, ort_loss, total
