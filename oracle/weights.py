"""TEST INFRASTRUCTURE ONLY -- parameter manifests and the deterministic weight function.

``manifest_*`` list (name, shape, kind) in the reference's ``state_dict()`` order; the
names/shapes are checked against the reference import by ``oracle/make_golden.py``
and frozen into ``tests/golden/manifest.json``.

``make_state_dict(manifest, seed, mode)`` regenerates identical weights anywhere
(NumPy ``default_rng`` keyed by crc32(name) ^ seed), so neither the reference nor
382 MB of weights ever has to travel to the GPU box.

mode "d4": the statistics the reference's ``weights_init`` really produces (SURVEY
defect D4): Conv2d / Linear weights ~ N(-1, 1); ConvTranspose2d weights, all biases,
Embedding: framework defaults (U(+-1/sqrt(fan_in)), U(-1,1) for the embedding);
InstanceNorm affine = (1, 0).
mode "wc": well-conditioned: weights ~ N(0, 1/fan_in), affine (1 + 0.1 N, 0.1 N),
biases 0.1 N -- exercises every parameter with O(1) activations.
"""
import zlib
import numpy as np
import torch

ENC_LAYERS = (64, 128, 256, 512, 1024)
DEC_LAYERS = (1024, 512, 256, 128, 64)


def _cbam(p, c):
    return [(p + "channel_attention.conv1.weight", (c // 16, c, 1, 1), "conv_w"),
            (p + "channel_attention.conv2.weight", (c, c // 16, 1, 1), "conv_w"),
            (p + "spatial_attention.conv.weight", (1, 2, 3, 3), "conv_w")]


def _in(p, c):
    return [(p + "weight", (c,), "in_w"), (p + "bias", (c,), "in_b")]


def _trunk(p):
    m = []
    m += [(p + "time_pitch.time.weight", (32, 1, 4, 1), "conv_w"), (p + "time_pitch.pitch.weight", (32, 32, 1, 4), "conv_w")]
    m += _in(p + "time_pitch.bn.", 32) + _cbam(p + "time_pitch.cbam.", 32)
    m += [(p + "pitch_time.pitch.weight", (32, 1, 1, 4), "conv_w"), (p + "pitch_time.time.weight", (32, 32, 4, 1), "conv_w")]
    m += _in(p + "pitch_time.bn.", 32) + _cbam(p + "pitch_time.cbam.", 32)
    for i in range(len(ENC_LAYERS) - 1):
        ci, co = ENC_LAYERS[i], ENC_LAYERS[i + 1]
        q = p + "layers.%d." % (2 * i)
        m += [(q + "conv1.weight", (ci, ci, 3, 3), "conv_w"), (q + "conv2.weight", (ci, ci, 3, 3), "conv_w")]
        m += _in(q + "bn.", ci) + _cbam(q + "cbam.", ci)
        q = p + "layers.%d." % (2 * i + 1)
        m += [(q + "conv.weight", (co, ci, 3, 3), "conv_w")]
        m += _in(q + "bn.", co) + _cbam(q + "cbam.", co)
    return m


def manifest_encoder(p=""):
    return _trunk(p) + [(p + "linear.weight", (1152, 1024), "lin_w"), (p + "linear.bias", (1152,), "lin_b")]


def manifest_phrase_model(p=""):
    q = p + "phrase_encoder."
    return _trunk(q) + [(q + "linear.weight", (1152, 1024), "lin_w")]


def manifest_decoder(p=""):
    m = [(p + "bar_linear.weight", (1152, 2304), "lin_w"), (p + "bar_linear.bias", (1152,), "lin_b"),
         (p + "phrase_linear.weight", (1152, 2304), "lin_w"), (p + "phrase_linear.bias", (1152,), "lin_b")]
    m += [(p + "time.time.weight", (2304, 1024, 6, 1), "convT_w"), (p + "time.pitch.weight", (1024, 1024, 1, 3), "convT_w")]
    m += _in(p + "time.bn.", 1024) + _cbam(p + "time.cbam.", 1024)
    m += [(p + "pitch.pitch.weight", (2304, 1024, 1, 3), "convT_w"), (p + "pitch.time.weight", (1024, 1024, 6, 1), "convT_w")]
    m += _in(p + "pitch.bn.", 1024) + _cbam(p + "pitch.cbam.", 1024)
    m += [(p + "fit1.weight", (1024, 2048, 1, 1), "conv_w")] + _in(p + "bn.", 1024)
    m += [(p + "fit2.weight", (1, 64, 1, 1), "conv_w")]
    for i in range(1, len(DEC_LAYERS)):
        ci, co = DEC_LAYERS[i - 1], DEC_LAYERS[i]
        q = p + "layers.%d." % (i - 1)
        if i < 3:
            m += [(q + "deConv1.weight", (ci, co, 4, 4), "convT_w"), (q + "deConv1.bias", (co,), "convT_b"),
                  (q + "deConv2.weight", (ci, co, 4, 4), "convT_w"), (q + "deConv2.bias", (co,), "convT_b"),
                  (q + "conv.weight", (co, ci, 1, 1), "conv_w")]
            m += _in(q + "bn1.", co) + _in(q + "bn2.", co) + _in(q + "bn3.", co)
            m += _cbam(q + "cbam1.", co) + _cbam(q + "cbam2.", co)
        else:
            m += [(q + "deConv1.weight", (ci, co, 4, 4), "convT_w"),
                  (q + "deConv2.weight", (ci, co, 3, 3), "convT_w"), (q + "deConv2.bias", (co,), "convT_b"),
                  (q + "conv.weight", (co, ci, 1, 1), "conv_w")]
            m += _in(q + "bn1.", co) + _in(q + "bn2.", co) + _in(q + "bn3.", co)
            m += _cbam(q + "cbam.", co)
    m += _cbam(p + "cbam.", 1024)
    m += [(p + "position_embedding.weight", (332, 1152), "emb")]
    return m


def manifest_generator(p=""):
    """graph/model_with_gan.py Model (== graph/model.py Model without the Refiner)."""
    return manifest_encoder(p + "encoder.") + manifest_decoder(p + "decoder.") + manifest_phrase_model(p + "phrase_encoder.")


def manifest_z_discriminator(p=""):
    m, dims = [], (1152, 512, 512, 512, 512, 1)
    for j, i in enumerate((0, 2, 4, 6, 8)):
        m += [(p + "net.%d.weight" % i, (dims[j + 1], dims[j]), "lin_w"), (p + "net.%d.bias" % i, (dims[j + 1],), "lin_b")]
    return m


def manifest_bar_feature_discriminator(p=""):
    return [(p + "linear1.weight", (512, 1152), "lin_w"), (p + "linear2.weight", (1, 512), "lin_w")]


def _bnm(p, c):
    return [(p + "weight", (c,), "bn_w"), (p + "bias", (c,), "bn_b"), (p + "running_mean", (c,), "bn_rm"),
            (p + "running_var", (c,), "bn_rv"), (p + "num_batches_tracked", (), "bn_nbt")]


def manifest_bar_discriminator(p=""):
    q = p + "chord."
    m = [(q + "chord_conv1.weight", (8, 1, 3, 1), "conv_w"), (q + "chord_conv2.weight", (16, 8, 3, 1), "conv_w"),
         (q + "chord_fit.weight", (16, 16, 1, 1), "conv_w"), (q + "chord_conv3.weight", (32, 16, 3, 3), "conv_w"),
         (q + "chord_conv4.weight", (64, 32, 3, 3), "conv_w")]
    for i, c in enumerate((8, 16, 16, 32, 64)):
        m += _bnm(q + "batch_norm%d." % (i + 1), c)
    q = p + "onoff."
    m += [(q + "onoff_conv1.weight", (8, 1, 3, 3), "conv_w"), (q + "onoff_conv2.weight", (8, 8, 3, 3), "conv_w")]
    m += _bnm(q + "batch_norm2.", 8)
    m += [(q + "onoff_conv3.weight", (16, 8, 3, 3), "conv_w"), (q + "onoff_conv4.weight", (32, 16, 3, 3), "conv_w"),
          (q + "onoff_fit.weight", (32, 32, 1, 1), "conv_w"), (q + "onoff_conv5.weight", (64, 32, 3, 3), "conv_w")]
    q = p + "basic."
    m += [(q + "pitch1.weight", (8, 1, 1, 4), "conv_w"), (q + "pitch2.weight", (8, 8, 4, 1), "conv_w"),
          (q + "time1.weight", (8, 1, 4, 1), "conv_w"), (q + "time2.weight", (8, 8, 1, 4), "conv_w"),
          (q + "fit.weight", (8, 16, 1, 1), "conv_w")]
    m += _bnm(q + "bn.", 8)
    chans = (8, 16, 32, 64)
    for i in range(3):
        r = q + "layers.%d." % i
        if i < 2:
            m += [(r + "conv1.weight", (chans[i], chans[i], 3, 3), "conv_w")]
        m += [(r + "conv2.weight", (chans[i + 1], chans[i], 3, 3), "conv_w")]
        m += _bnm(r + "bn1.", chans[i]) + _bnm(r + "bn2.", chans[i + 1])
    m += [(p + "linear.weight", (1, 192), "lin_w")]
    return m


def manifest_refiner(p=""):
    """graph/refiner.py state_dict with the D2 fix (layer2.0.weight is [8,2,4,4], not [8,1,4,4])"""
    m = [(p + "layer1.0.weight", (2, 1, 4, 4), "conv_w"), (p + "layer1.0.bias", (2,), "lin_b")] + _bnm(p + "layer1.1.", 2)
    m += [(p + "layer2.0.weight", (8, 2, 4, 4), "conv_w"), (p + "layer2.0.bias", (8,), "lin_b")] + _bnm(p + "layer2.1.", 8)
    m += [(p + "layer3.0.weight", (1024, 2880), "lin_w"), (p + "layer3.0.bias", (1024,), "lin_b")]
    m += [(p + "layer4.0.weight", (2880, 1024), "lin_w"), (p + "layer4.0.bias", (2880,), "lin_b")]
    m += [(p + "layer5.0.weight", (8, 2, 4, 4), "convT_w")] + _bnm(p + "layer5.1.", 2)
    m += [(p + "layer6.0.weight", (2, 1, 4, 4), "convT_w")] + _bnm(p + "layer6.1.", 1)
    return m


def _fan_in(shape, kind):
    if kind in ("conv_w",):
        return shape[1] * shape[2] * shape[3]
    if kind == "convT_w":
        # torch computes fan_in of a ConvTranspose2d weight [Cin, Cout, kh, kw] from dim 1
        return shape[1] * shape[2] * shape[3]
    if kind == "lin_w":
        return shape[1]
    return 1


def make_tensor(name, shape, kind, seed=0, mode="d4", owner_fan_in=None):
    rng = np.random.default_rng((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)
    n = int(np.prod(shape)) if len(shape) else 1
    fi = owner_fan_in if owner_fan_in is not None else _fan_in(shape, kind)
    if kind == "bn_nbt":
        return torch.zeros((), dtype=torch.int64)
    if kind == "bn_rm":
        a = np.zeros(n, np.float32)
    elif kind == "bn_rv":
        a = np.ones(n, np.float32)
    elif mode == "d4":
        if kind in ("conv_w", "lin_w", "bn_w"):
            a = rng.normal(-1.0, 1.0, n)
        elif kind == "convT_w":
            b = 1.0 / np.sqrt(fi)
            a = rng.uniform(-b, b, n)
        elif kind in ("lin_b", "convT_b"):
            b = 1.0 / np.sqrt(fi)
            a = rng.uniform(-b, b, n)
        elif kind == "in_w":
            a = np.ones(n)
        elif kind in ("in_b", "bn_b"):
            a = np.zeros(n)
        elif kind == "emb":
            a = rng.uniform(-1.0, 1.0, n)
        else:
            raise ValueError(kind)
    elif mode == "wc":
        if kind in ("conv_w", "lin_w", "convT_w"):
            a = rng.normal(0.0, 1.0 / np.sqrt(fi), n)
        elif kind in ("in_w", "bn_w"):
            a = 1.0 + 0.1 * rng.normal(size=n)
        elif kind in ("in_b", "bn_b", "lin_b", "convT_b"):
            a = 0.1 * rng.normal(size=n)
        elif kind == "emb":
            a = rng.uniform(-1.0, 1.0, n)
        else:
            raise ValueError(kind)
    else:
        raise ValueError(mode)
    return torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))


def make_state_dict(manifest, seed=0, mode="d4", dtype=torch.float32):
    """Deterministic weights for a manifest.  Bias fan-in follows its weight (the
    entry just before it in the manifest, as in torch's reset_parameters)."""
    sd, last_fi = {}, 1
    for name, shape, kind in manifest:
        if kind in ("conv_w", "convT_w", "lin_w"):
            last_fi = _fan_in(shape, kind)
        t = make_tensor(name, shape, kind, seed, mode, owner_fan_in=last_fi if kind in ("lin_b", "convT_b") else None)
        sd[name] = t.to(dtype) if t.is_floating_point() else t
    return sd


def split_generator(gsd):
    """generator state_dict -> (encoder, decoder, phrase_model) sub-dicts with the prefix stripped"""
    out = []
    for pref in ("encoder.", "decoder.", "phrase_encoder."):
        out.append({k[len(pref):]: v for k, v in gsd.items() if k.startswith(pref)})
    return tuple(out)


def make_inputs(batch, seed=1234, p_on=0.05):
    """Synthetic piano-roll batch: Bernoulli(p_on) rolls, uniform positions."""
    rng = np.random.default_rng(seed)
    note = (rng.random((batch, 1, 96, 60)) < p_on).astype(np.float32)
    pre_note = (rng.random((batch, 1, 96, 60)) < p_on).astype(np.float32)
    phrase = (rng.random((batch, 1, 384, 60)) < p_on).astype(np.float32)
    position = rng.integers(0, 332, size=(batch,), dtype=np.int64)
    return (torch.from_numpy(note), torch.from_numpy(pre_note), torch.from_numpy(phrase), torch.from_numpy(position))
