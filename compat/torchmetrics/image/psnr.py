import torch


class PeakSignalNoiseRatio(torch.nn.Module):
    """10 log10(range^2 / MSE) over everything passed in one call (torchmetrics' default reduction over the whole batch)."""

    def __init__(self, data_range=None, **_unused):
        super().__init__()
        if isinstance(data_range, (tuple, list)):
            self.lo, self.hi = float(data_range[0]), float(data_range[1])
        else:
            self.lo, self.hi = None, (float(data_range) if data_range is not None else None)

    def forward(self, preds, target):
        if self.lo is not None:
            preds, target = preds.clamp(self.lo, self.hi), target.clamp(self.lo, self.hi)
            rng = self.hi - self.lo
        else:
            rng = self.hi if self.hi is not None else float(target.max() - target.min())
        mse = (preds - target).square().mean()
        return 10.0 * torch.log10(rng * rng / mse)
