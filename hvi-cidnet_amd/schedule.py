"""Epoch learning-rate schedule of the reference's training script ("next" row f2 of SURVEY section 8).

Reference: `make_scheduler`, train.py:165-181, with the default options (`--cos_restart True`):
    start_warmup=True : GradualWarmupScheduler(multiplier=1, total_epoch=warmup_epochs,
                            after_scheduler=CosineAnnealingRestartLR(periods=[nEpochs - warmup_epochs - start_epoch],
                                                                     restart_weights=[1], eta_min=1e-7))
    start_warmup=False: CosineAnnealingRestartLR(periods=[nEpochs - start_epoch], restart_weights=[1], eta_min=1e-7)
(data/scheduler.py:5-63, 133-172), stepped once per epoch (train.py:220).  `lr_after(n)` is the learning rate the
optimizer holds after n calls of `scheduler.step()` -- including the reference's quirks: the warm-up starts from lr = 0
at construction, and the first epoch after the warm-up runs at the full base lr a second time because the wrapped cosine
scheduler only starts counting one step later.  Pinned against the reference classes in tests/golden/lr_schedule.npz."""
import math


class WarmupCosineLR:
    def __init__(self, base_lr, nEpochs, warmup_epochs=3, start_epoch=0, start_warmup=True, eta_min=1e-7):
        self.base_lr = float(base_lr)
        self.warmup = int(warmup_epochs) if start_warmup else 0
        self.start_warmup = bool(start_warmup)
        self.period = int(nEpochs) - self.warmup - int(start_epoch)
        self.eta_min = float(eta_min)
        if self.period <= 0:
            raise ValueError("nEpochs must exceed warmup_epochs + start_epoch")

    def _cosine(self, e):
        # CosineAnnealingRestartLR.get_lr with one period and restart weight 1 (data/scheduler.py:160-172);
        # past the period get_position_from_periods returns None in the reference and indexing fails: not reproduced
        if e > self.period:
            raise ValueError("the reference's scheduler is undefined past its single period")
        return self.eta_min + 0.5 * (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * (e / self.period)))

    def lr_after(self, n):
        """learning rate after n scheduler.step() calls (n = 0: right after construction)"""
        n = int(n)
        if not self.start_warmup:
            return self._cosine(n)
        if n <= self.warmup:
            return self.base_lr * (float(n) / self.warmup)          # multiplier == 1: 0 -> base_lr (scheduler.py:37)
        return self._cosine(n - self.warmup - 1)                     # the wrapped scheduler starts one step late

    def apply(self, trainer, n):
        """set the fused Adam's lr for the epoch that follows n scheduler steps"""
        lr = self.lr_after(n)
        trainer.set_lr(lr)
        return lr
