"""Learning-rate schedule of the training loop.

The reference chains a linear warm-up (schedulers/warmup.py:7-47, used with multiplier 1:
lr = base * it / nr_warmup_iters) to MultiStepLR(gamma 0.3) (base_method.py:71-76,
volsurfs.py:774-783) and steps it once per training iteration (trainer.py:306-308).
Here the schedule is one closed form, `lr_at`, plus a small stepper with the scheduler
call shape the trainer uses (`step()`, `get_last_lr()`); `GradualWarmupScheduler` is kept
as a constructor-compatible name.  Pinned by tests/golden/lr_schedule.npz — the lr the
reference's own scheduler classes produce, stepped in the build container."""
from bisect import bisect_right

from torch.optim.lr_scheduler import MultiStepLR  # noqa: F401  (the reference vendors torch 1.10's)


def lr_at(it, base_lr, nr_warmup_iters, milestones, gamma=0.3):
    """lr used BY iteration `it` (0-based; the scheduler has been stepped `it` times).

    Warm-up covers steps 0..nr_warmup_iters inclusive (the reference switches over when
    last_epoch > total_epoch); the decay scheduler only starts counting its own epochs one
    step after that, so milestone m takes effect at it = nr_warmup_iters + 1 + m."""
    if nr_warmup_iters > 0:
        if it <= nr_warmup_iters:
            return base_lr * (float(it) / nr_warmup_iters)
        decay_epoch = max(it - nr_warmup_iters - 1, 0)
    else:
        decay_epoch = it
    return base_lr * gamma ** bisect_right(sorted(milestones), decay_epoch)


class WarmupMultiStep:
    """Stepper over `lr_at` for every param group of an optimizer."""

    def __init__(self, optimizer, nr_warmup_iters, milestones, gamma=0.3):
        self.optimizer = optimizer
        self.nr_warmup_iters, self.milestones, self.gamma = nr_warmup_iters, list(milestones), gamma
        self.base_lrs = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.it = 0
        self._apply()

    def _apply(self):
        self._last = [lr_at(self.it, b, self.nr_warmup_iters, self.milestones, self.gamma)
                      for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, self._last):
            g["lr"] = lr

    def step(self):
        self.it += 1
        self._apply()

    def get_last_lr(self):
        return list(self._last)

    def state_dict(self):
        return {"it": self.it}

    def load_state_dict(self, sd):
        self.it = int(sd["it"])
        self._apply()


def GradualWarmupScheduler(optimizer, multiplier, total_epoch, after_scheduler=None):
    """Constructor-compatible with schedulers/warmup.py for the way VolSurfs uses it
    (multiplier 1, MultiStepLR afterwards)."""
    if multiplier != 1:
        raise NotImplementedError("the K-shell path only uses multiplier=1 (volsurfs.py:776-781)")
    ms = sorted(after_scheduler.milestones.elements()) if after_scheduler is not None else []
    gamma = after_scheduler.gamma if after_scheduler is not None else 1.0
    return WarmupMultiStep(optimizer, total_epoch, ms, gamma)
