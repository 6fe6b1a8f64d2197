"""AverageMeter as the agents use it (reference: metrics.py:27-49).  Values may be 0-dim device
tensors: the running sum then stays on the device and nothing synchronises until ``val`` is
formatted at the end of the epoch."""


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.value = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        if hasattr(val, "detach"):
            val = val.detach()
        self.value = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count

    @property
    def val(self):
        return self.avg
