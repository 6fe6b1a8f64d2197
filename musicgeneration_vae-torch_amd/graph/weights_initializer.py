"""weights_init with the reference's exact (and odd) semantics, SURVEY defect D4
(reference: graph/weights_initializer.py:5-23): class names containing 'Conv2',
'BatchNorm' or 'Linear' get weight ~ N(mean=-1, std=1); the "bias" branch re-draws the
WEIGHT, so biases keep their defaults; ConvTranspose2d / InstanceNorm2d / Embedding are
not matched at all."""


def weights_init(m):
    name = type(m).__name__
    for key in ("Conv2", "BatchNorm", "Linear"):
        if key in name:
            w = getattr(m, "weight", None)
            if w is None:
                return
            w.data.normal_(-1.0, 1.0)
            if getattr(m, "bias", None) is not None:
                w.data.normal_(-1.0, 1.0)
            return
