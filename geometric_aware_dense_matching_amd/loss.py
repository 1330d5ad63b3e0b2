"""Training losses of the geoMatch path, same names / semantics as /root/reference/models/loss.py:
CircleLoss :433-494, FocalLoss :15-46, AutomaticWeightedLoss :496-516.
Plain torch on the device the inputs live on (the reference hard-codes `.cuda()`, :509)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class FocalLoss(nn.Module):
    def __init__(self, gamma=0, alpha=None, size_average=True):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha
        if isinstance(alpha, (float, int)):
            self.alpha = torch.tensor([alpha, 1 - alpha])
        if isinstance(alpha, list):
            self.alpha = torch.tensor(alpha)
        self.size_average = size_average

    def forward(self, input, target):
        input = input.transpose(1, 2)                       # [B,C,N] -> [B,N,C]
        input = input.contiguous().view(-1, input.size(2))
        target = target.view(-1, 1)
        logpt = F.log_softmax(input, dim=-1).gather(1, target).view(-1)
        pt = logpt.detach().exp()
        if self.alpha is not None:
            at = self.alpha.to(input).gather(0, target.view(-1))
            logpt = logpt * at
        loss = -1 * (1 - pt) ** self.gamma * logpt
        return loss.mean() if self.size_average else loss.sum()


class CircleLoss(nn.Module):
    def __init__(self, gamma):
        super().__init__()
        self.gamma = gamma
        self.soft_plus = nn.Softplus()

    @staticmethod
    def log_sum_exp(inputs, mask):
        """loss.py:441-459: masked LSE; `mask` is 1.0 where the entry takes part."""
        inv = 1.0 - mask
        s, _ = torch.max(inputs + (-1e7 * inv), dim=-1, keepdim=True)
        off = (inputs - s).masked_fill(inv.to(torch.bool), -float("inf"))
        return (s + off.exp().sum(dim=-1, keepdim=True).log()).squeeze(-1)

    def rows(self, sim, mask, m):
        """Per-row loss f32[rows] (forward() is its mean)."""
        ap = torch.clamp_min(-sim.detach() + 1 + m, min=0.0).masked_fill(~mask, 0)
        an = torch.clamp_min(sim.detach() + m, min=0.0).masked_fill(mask, 0)
        delta_p, delta_n = 1 - m, m
        logit_p = -ap * (sim - delta_p) * self.gamma
        logit_n = an * (sim - delta_n) * self.gamma
        lse_p = self.log_sum_exp(logit_p, mask.to(torch.float))
        lse_n = self.log_sum_exp(logit_n, (~mask).to(torch.float))
        return self.soft_plus(lse_p + lse_n)

    def forward(self, sim, mask, m):
        return self.rows(sim, mask, m).mean()


class AutomaticWeightedLoss(nn.Module):
    def __init__(self, num=2):
        super().__init__()
        self.params = nn.Parameter(torch.ones(num))

    def forward(self, *x):
        loss_sum = 0
        for i, loss in enumerate(x):
            loss_sum = loss_sum + 0.5 / (self.params[i] ** 2) * loss + torch.log(1 + self.params[i] ** 2)
        return loss_sum
