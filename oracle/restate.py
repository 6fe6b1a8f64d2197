"""TEST INFRASTRUCTURE ONLY -- functional torch-CPU restatement of the reference hot path.

Every function takes a flat ``state_dict``-style mapping ``sd`` (same key names and
tensor layouts as the reference's ``nn.Module.state_dict()``) plus a key prefix, and
restates one reference ``forward``.  Device-agnostic (fixes SURVEY defects D1/D3),
literal about D5 (``bn2`` used on both branches).  Works under autograd when the
tensors in ``sd`` require grad, which is how ``bench.py``'s ``cpu_baseline`` leg and
the gradient fixtures use it.

Pinned bit-for-bit against the reference import by ``oracle/make_golden.py``.
"""
import numpy as np
import torch
import torch.nn.functional as F

# Test hook for the bf16-STORAGE mode (BASELINE.json configs 3-4; tests/test_hip_parity.py's bf16 step): a pair of callables
# (round_activation, round_weight).  When set, every convolution of the encoder trunks and of the decoder's blocks / fit1
# -- the channels-last island, where the HIP path keeps activations and their gradients as bf16 tensors and multiplies
# bf16 copies of the fp32 master weights -- sees rounded operands and stores a rounded result:
#     y = round_activation(conv(round_activation(x), round_weight(w)))
# None (the default, and the only state the golden vectors / CPU baseline ever see): plain arithmetic.
ISLAND_ROUNDING = None


def _island(fn, x, w, *a, **k):
    if ISLAND_ROUNDING is None:
        return fn(x, w, *a, **k)
    qa, qw = ISLAND_ROUNDING
    return qa(fn(qa(x), qw(w), *a, **k))


ENC_LAYERS = (64, 128, 256, 512, 1024)
DEC_LAYERS = (1024, 512, 256, 128, 64)
Z_DIM = 1152


# --------------------------------------------------------------------------- blocks
def instance_norm(sd, p, x):
    # nn.InstanceNorm2d(C, eps=1e-5, affine=True), no running stats
    return F.instance_norm(x, None, None, sd[p + "weight"], sd[p + "bias"], True, 0.01, 1e-5)


def channel_attention(sd, p, x):
    """graph/cbam.py:22-29"""
    w1, w2 = sd[p + "conv1.weight"], sd[p + "conv2.weight"]
    a = F.conv2d(F.relu(F.conv2d(F.adaptive_avg_pool2d(x, 1), w1)), w2)
    m = F.conv2d(F.relu(F.conv2d(F.adaptive_max_pool2d(x, 1), w1)), w2)
    return x * torch.sigmoid(a + m)


def spatial_attention(sd, p, x):
    """graph/cbam.py:43-52"""
    avg = torch.mean(x, dim=1, keepdim=True)
    mx, _ = torch.max(x, dim=1, keepdim=True)
    g = F.conv2d(torch.cat([avg, mx], dim=1), sd[p + "conv.weight"], padding=1)
    return x * torch.sigmoid(g)


def cbam(sd, p, x):
    """graph/cbam.py:63-67"""
    return spatial_attention(sd, p + "spatial_attention.", channel_attention(sd, p + "channel_attention.", x))


def enc_time_pitch(sd, p, x):
    """graph/encodingBlock.py:25-36"""
    o = F.conv2d(x, sd[p + "time.weight"], stride=(2, 1), padding=(1, 0))
    o = F.leaky_relu(o, 0.01)
    o = F.conv2d(o, sd[p + "pitch.weight"], stride=(1, 2), padding=(0, 1))
    o = instance_norm(sd, p + "bn.", o)
    o = o + cbam(sd, p + "cbam.", o)
    return F.leaky_relu(o, 0.01)


def enc_pitch_time(sd, p, x):
    """graph/encodingBlock.py:56-67"""
    o = F.conv2d(x, sd[p + "pitch.weight"], stride=(1, 2), padding=(0, 1))
    o = F.leaky_relu(o, 0.01)
    o = F.conv2d(o, sd[p + "time.weight"], stride=(2, 1), padding=(1, 0))
    o = instance_norm(sd, p + "bn.", o)
    o = o + cbam(sd, p + "cbam.", o)
    return F.leaky_relu(o, 0.01)


def residual_module(sd, p, x):
    """graph/encodingBlock.py:87-100"""
    o = F.relu(_island(F.conv2d, x, sd[p + "conv1.weight"], padding=1))
    o = _island(F.conv2d, o, sd[p + "conv2.weight"], padding=1)
    o = instance_norm(sd, p + "bn.", o)
    o = cbam(sd, p + "cbam.", o)
    return F.relu(x + o)


def pooling_module(sd, p, x):
    """graph/encodingBlock.py:118-126"""
    o = _island(F.conv2d, x, sd[p + "conv.weight"], stride=2, padding=1)
    o = instance_norm(sd, p + "bn.", o)
    o = o + cbam(sd, p + "cbam.", o)
    return F.relu(o)


def _trunk(sd, p, x, taps=None):
    pitch = enc_pitch_time(sd, p + "pitch_time.", x)
    time = enc_time_pitch(sd, p + "time_pitch.", x)
    if taps is not None:
        taps[p + "pitch_time"] = pitch
        taps[p + "time_pitch"] = time
    o = torch.cat((pitch, time), dim=1)
    for i in range(len(ENC_LAYERS) - 1):
        o = residual_module(sd, p + "layers.%d." % (2 * i), o)
        if taps is not None:
            taps[p + "layers.%d" % (2 * i)] = o
        o = pooling_module(sd, p + "layers.%d." % (2 * i + 1), o)
        if taps is not None:
            taps[p + "layers.%d" % (2 * i + 1)] = o
    return o


def encoder(sd, p, x, taps=None):
    """graph/encoder.py:26-40 -- AvgPool(3,2) then Linear(1024,1152)+bias"""
    o = _trunk(sd, p, x, taps)
    o = F.avg_pool2d(o, (3, 2)).view(-1, 1024)
    return F.linear(o, sd[p + "linear.weight"], sd[p + "linear.bias"])


def phrase_encoder(sd, p, x, taps=None):
    """graph/phrase_encoder.py:27-41 -- AvgPool(12,2) then Linear without bias"""
    o = _trunk(sd, p, x, taps)
    o = F.avg_pool2d(o, (12, 2)).view(-1, 1024)
    return F.linear(o, sd[p + "linear.weight"])


def phrase_model(sd, p, x, taps=None):
    """graph/phrase_encoder.py:52-55"""
    return phrase_encoder(sd, p + "phrase_encoder.", x, taps)


def dec_pitch_time(sd, p, x):
    """graph/decoder.py:55-66"""
    o = F.relu(F.conv_transpose2d(x, sd[p + "pitch.weight"], stride=(1, 3)))
    o = F.conv_transpose2d(o, sd[p + "time.weight"], stride=(6, 1))
    o = instance_norm(sd, p + "bn.", o)
    o = o + cbam(sd, p + "cbam.", o)
    return F.relu(o)


def dec_time_pitch(sd, p, x):
    """graph/decoder.py:25-36"""
    o = F.relu(F.conv_transpose2d(x, sd[p + "time.weight"], stride=(6, 1)))
    o = F.conv_transpose2d(o, sd[p + "pitch.weight"], stride=(1, 3))
    o = instance_norm(sd, p + "bn.", o)
    o = o + cbam(sd, p + "cbam.", o)
    return F.relu(o)


def deconv_pitch_padding(sd, p, x):
    """graph/decoder.py:135-154 (bn2 on both branches, bn1 unused: defect D5)"""
    o1 = _island(F.conv_transpose2d, x, sd[p + "deConv1.weight"], sd[p + "deConv1.bias"], stride=2, padding=1,
                            output_padding=(0, 1))
    o1 = instance_norm(sd, p + "bn2.", o1)
    o1 = F.relu(o1 + cbam(sd, p + "cbam1.", o1))
    o2 = _island(F.conv_transpose2d, x, sd[p + "deConv2.weight"], sd[p + "deConv2.bias"], stride=2, padding=1,
                            output_padding=(0, 1))
    o2 = F.relu(instance_norm(sd, p + "bn2.", o2))
    o = _island(F.conv2d, torch.cat((o1, o2), dim=1), sd[p + "conv.weight"])
    o = instance_norm(sd, p + "bn3.", o)
    return F.relu(o + cbam(sd, p + "cbam2.", o))


def deconv_module(sd, p, x):
    """graph/decoder.py:91-109"""
    o1 = _island(F.conv_transpose2d, x, sd[p + "deConv1.weight"], stride=2, padding=1)
    o1 = F.relu(instance_norm(sd, p + "bn1.", o1))
    o2 = _island(F.conv_transpose2d, x, sd[p + "deConv2.weight"], sd[p + "deConv2.bias"], stride=2, padding=1,
                            output_padding=1)
    o2 = F.relu(instance_norm(sd, p + "bn2.", o2))
    o = _island(F.conv2d, torch.cat((o1, o2), dim=1), sd[p + "conv.weight"])
    o = instance_norm(sd, p + "bn3.", o)
    return F.relu(o + cbam(sd, p + "cbam.", o))


def decoder_head(sd, p, z, pre_z, phrase_feature, position, train=False, drop_masks=None):
    """graph/decoder.py:192-205: embedding gather, the two Linear + ReLU + Dropout(0.3) branches, concat -> [B, 2304, 1, 1].
    ``drop_masks`` (two pre-scaled {0, 1/0.7} tensors [B,1152]) replaces torch's RNG so the HIP path can be compared on the
    same mask."""
    def drop(t, i):
        if drop_masks is not None:
            return t * drop_masks[i]
        return F.dropout(t, 0.3, train)

    pf = torch.cat((phrase_feature, F.embedding(position, sd[p + "position_embedding.weight"])), dim=1)
    pf = drop(F.relu(F.linear(pf, sd[p + "phrase_linear.weight"], sd[p + "phrase_linear.bias"])), 0)
    bf = torch.cat((z, pre_z), dim=1)
    bf = drop(F.relu(F.linear(bf, sd[p + "bar_linear.weight"], sd[p + "bar_linear.bias"])), 1)
    return torch.cat((bf, pf), dim=1).view(-1, 2304, 1, 1)


def decoder_fit1(sd, p, o):
    """graph/decoder.py:213-215: fit1 -> InstanceNorm -> +CBAM -> ReLU"""
    o = _island(F.conv2d, o, sd[p + "fit1.weight"])
    o = instance_norm(sd, p + "bn.", o)
    return F.relu(o + cbam(sd, p + "cbam.", o))


def decoder(sd, p, z, pre_z, phrase_feature, position, train=False, drop_masks=None, taps=None,
            return_logits=False):
    """graph/decoder.py:192-222.  ``train`` enables Dropout(0.3); ``drop_masks``: see decoder_head."""
    x = decoder_head(sd, p, z, pre_z, phrase_feature, position, train, drop_masks)
    if taps is not None:
        taps[p + "head"] = x
    pitch = dec_pitch_time(sd, p + "pitch.", x)
    time = dec_time_pitch(sd, p + "time.", x)
    if taps is not None:
        taps[p + "pitch"] = pitch
        taps[p + "time"] = time
    o = decoder_fit1(sd, p, torch.cat((pitch, time), dim=1))
    if taps is not None:
        taps[p + "fit1"] = o
    for i in range(1, len(DEC_LAYERS)):
        q = p + "layers.%d." % (i - 1)
        o = deconv_pitch_padding(sd, q, o) if i < 3 else deconv_module(sd, q, o)
        if taps is not None:
            taps[p + "layers.%d" % (i - 1)] = o
    pre = F.conv2d(o, sd[p + "fit2.weight"])
    gen = torch.sigmoid(pre)
    return (gen, pre) if return_logits else gen


def refiner(sd, p, x, train=True):
    """graph/refiner.py:49-58 with layer2 taking 2 input channels (the evident intent; the reference's
    Conv2d(1, 8) raises: defect D2).  UNPINNED: the reference cannot produce a value."""
    def bn(q, t):
        return _bn(sd, q, t, train, 0.1)
    x_2 = F.max_pool2d(F.leaky_relu(bn(p + "layer1.1.", F.conv2d(x, sd[p + "layer1.0.weight"], sd[p + "layer1.0.bias"], padding=2)), 0.2), 2)
    x_8 = F.max_pool2d(F.leaky_relu(bn(p + "layer2.1.", F.conv2d(x_2, sd[p + "layer2.0.weight"], sd[p + "layer2.0.bias"], padding=2)), 0.2), 2)
    f = F.relu(F.linear(x_8.reshape(-1, 2880), sd[p + "layer3.0.weight"], sd[p + "layer3.0.bias"]))
    f = F.relu(F.linear(f, sd[p + "layer4.0.weight"], sd[p + "layer4.0.bias"]))
    x_8_t = x_8 + f.view(-1, 8, 24, 15)
    x_2_t = x_2 + F.relu(bn(p + "layer5.1.", F.conv_transpose2d(x_8_t, sd[p + "layer5.0.weight"], stride=2, padding=1)))
    y = torch.sigmoid(bn(p + "layer6.1.", F.conv_transpose2d(x_2_t, sd[p + "layer6.0.weight"], stride=2, padding=1)))
    return (x + y) * 0.5


# ---------------------------------------------------------------- generator wrappers
def generator_train(sd, note, pre_note, phrase, position, train=False, drop_masks=None, p=""):
    """graph/model.py:22-33 minus the Refiner (defect D2: the reference's Refiner raises;
    the timed configs exclude it).  Returns (gen, z, pre_z, phrase_feature)."""
    phrase_feature = phrase_model(sd, p + "phrase_encoder.", phrase)
    z = encoder(sd, p + "encoder.", note)
    pre_z = encoder(sd, p + "encoder.", pre_note)
    gen = decoder(sd, p + "decoder.", z, pre_z, phrase_feature, position, train, drop_masks)
    return gen, z, pre_z, phrase_feature


def generator_sample(sd, latent, pre_note, phrase, position, p=""):
    """graph/model.py:34-41 minus the Refiner."""
    phrase_feature = phrase_model(sd, p + "phrase_encoder.", phrase)
    pre_z = encoder(sd, p + "encoder.", pre_note)
    return decoder(sd, p + "decoder.", latent, pre_z, phrase_feature, position)


def generator_gan(sd, note, pre_note, phrase, position, is_note=True, train=False, drop_masks=None, p=""):
    """graph/model_with_gan.py:20-38, device-agnostic (D3).  UNPINNED as a whole (the
    reference hard-codes torch.cuda.FloatTensor); pinned by composition."""
    phrase_feature = phrase_model(sd, p + "phrase_encoder.", phrase)
    pre_z = encoder(sd, p + "encoder.", pre_note)
    if is_note:
        z = encoder(sd, p + "encoder.", note)
        gen = decoder(sd, p + "decoder.", z, pre_z, phrase_feature, position, train, drop_masks)
        fake = torch.gt(gen, 0.3).to(gen.dtype)
        return gen, z, pre_z, phrase_feature, encoder(sd, p + "encoder.", fake)
    gen = decoder(sd, p + "decoder.", note, pre_z, phrase_feature, position, train, drop_masks)
    fake = torch.gt(gen, 0.3).to(gen.dtype)
    return gen, encoder(sd, p + "encoder.", fake)


# ------------------------------------------------------------------- discriminators
def z_discriminator(sd, p, x):
    """graph/z_discriminator.py:28-29,53-54 -- net.{0,2,4,6,8} Linear, ReLU between, sigmoid"""
    for i in (0, 2, 4, 6):
        x = F.relu(F.linear(x, sd[p + "net.%d.weight" % i], sd[p + "net.%d.bias" % i]))
    return torch.sigmoid(F.linear(x, sd[p + "net.8.weight"], sd[p + "net.8.bias"]))


def bar_feature_discriminator(sd, p, x):
    """graph/bar_discriminator_with_feature.py:17-25 (no activation between the Linears)"""
    x = F.linear(x.view(-1, Z_DIM), sd[p + "linear1.weight"])
    return torch.sigmoid(F.linear(x, sd[p + "linear2.weight"]))


def _bn(sd, p, x, train, momentum, eps=1e-5):
    if train and (p + "num_batches_tracked") in sd:
        sd[p + "num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        train, momentum, eps)


def bar_discriminator(sd, p, x, train=True):
    """graph/bar_discriminator.py:200-217 and the three feature towers (:31-58, :85-100,
    :122-183).  Restated literally, including the channel-dim slice in OnOffFeature
    (``x[:, :-1]`` is empty so onoff_x == sum over pitch of x).  With
    ``train=True`` the BatchNorm running statistics in ``sd`` are updated in place like
    the reference's modules."""
    x = x.view(-1, 1, 192, 60)
    # chord tower
    q = p + "chord."
    c = torch.sum(x.view(-1, 1, 192, 12, 5), 4, keepdim=True).view(-1, 1, 192, 12)
    c = F.relu(_bn(sd, q + "batch_norm1.", F.conv2d(c, sd[q + "chord_conv1.weight"], stride=(2, 1), padding=(1, 0)), train, 0.01))
    c = F.relu(_bn(sd, q + "batch_norm2.", F.conv2d(c, sd[q + "chord_conv2.weight"], stride=(2, 1), padding=(1, 0)), train, 0.01))
    c = F.relu(_bn(sd, q + "batch_norm3.", F.conv2d(c, sd[q + "chord_fit.weight"]), train, 0.01))
    c = F.relu(_bn(sd, q + "batch_norm4.", F.conv2d(c, sd[q + "chord_conv3.weight"], stride=2, padding=1), train, 0.01))
    c = F.relu(_bn(sd, q + "batch_norm5.", F.conv2d(c, sd[q + "chord_conv4.weight"], stride=2, padding=1), train, 0.01))
    c = F.avg_pool2d(c, (12, 3))
    # on/off tower
    q = p + "onoff."
    o = F.pad(x[:, :-1], (0, 0, 0, 0, 1, 0))
    o = torch.sum(x - o, 3, keepdim=True)
    o = F.relu(F.conv2d(o, sd[q + "onoff_conv1.weight"], stride=(2, 1), padding=1))
    o = F.relu(F.conv2d(o, sd[q + "onoff_conv2.weight"], stride=(2, 1), padding=1))
    o = _bn(sd, q + "batch_norm2.", o, train, 0.1)
    o = F.relu(F.conv2d(o, sd[q + "onoff_conv3.weight"], stride=(2, 1), padding=1))
    o = F.relu(F.conv2d(o, sd[q + "onoff_conv4.weight"], stride=(2, 1), padding=1))
    o = F.relu(F.conv2d(o, sd[q + "onoff_fit.weight"]))
    o = F.relu(F.conv2d(o, sd[q + "onoff_conv5.weight"], stride=(2, 1), padding=1))
    o = F.avg_pool2d(o, (6, 1))
    # basic tower
    q = p + "basic."
    pt = F.relu(F.conv2d(x, sd[q + "pitch1.weight"], stride=(1, 2), padding=(0, 1)))
    pt = F.relu(F.conv2d(pt, sd[q + "pitch2.weight"], stride=(2, 1), padding=(1, 0)))
    tm = F.relu(F.conv2d(x, sd[q + "time1.weight"], stride=(2, 1), padding=(1, 0)))
    tm = F.relu(F.conv2d(tm, sd[q + "time2.weight"], stride=(1, 2), padding=(0, 1)))
    b = F.conv2d(torch.cat((pt, tm), dim=1), sd[q + "fit.weight"])
    b = F.relu(_bn(sd, q + "bn.", b, train, 0.01))
    for i, basic in enumerate((False, False, True)):
        r = q + "layers.%d." % i
        if not basic:
            b = F.relu(_bn(sd, r + "bn1.", F.conv2d(b, sd[r + "conv1.weight"], padding=1), train, 0.01))
        b = F.relu(_bn(sd, r + "bn2.", F.conv2d(b, sd[r + "conv2.weight"], stride=2, padding=1), train, 0.01))
    b = F.avg_pool2d(b, (12, 4))
    f = torch.cat((c, o, b), dim=1).view(-1, 192)
    return torch.sigmoid(F.linear(f, sd[p + "linear.weight"]))


# --------------------------------------------------------------------------- losses
PITCH_PRIOR = np.array(
    [0.0079033, 0.00712255, 0.01189558, 0.00953322, 0.01102056, 0.01156428, 0.01136433, 0.01637716, 0.01211462,
     0.01776168, 0.01644157, 0.0171948, 0.01922302, 0.01582762, 0.02385192, 0.02001634, 0.02312213, 0.02348127,
     0.02263083, 0.0268141, 0.02373071, 0.02942328, 0.0272045, 0.0304963, 0.03032582, 0.02782333, 0.03458292,
     0.03230801, 0.03388906, 0.03283811, 0.03093611, 0.03616363, 0.03006419, 0.03296618, 0.02867032, 0.02654072,
     0.02609579, 0.01954488, 0.02251165, 0.01813882, 0.01599178, 0.01313839, 0.01104167, 0.01169814, 0.00756204,
     0.00793332, 0.00601032, 0.00540243, 0.00512497, 0.00286655, 0.00308927, 0.00260029, 0.00184589, 0.00166959,
     0.00103728, 0.00112497, 0.00071164, 0.00052543, 0.00072274, 0.00038808], dtype=np.float32)
"""The 60-bin pitch prior table of graph/loss/bar_loss.py:10-17 (data, not code)."""


def bar_loss(gen, labels, is_pretraining=False):
    """graph/loss/bar_loss.py:23-33, device-agnostic (D3).  UNPINNED directly (``Loss()``
    needs a GPU in the reference); composed of nn.BCELoss semantics which are pinned by
    ``dloss``.  BCE clamps log at -100 (torch semantics)."""
    if is_pretraining:
        recon = F.binary_cross_entropy(gen, labels)
    else:
        dist = torch.from_numpy(PITCH_PRIOR * np.float32(0.08)).to(gen.device)
        default = torch.tensor(np.array([0.1 / 60], dtype=np.float32)).to(gen.device)
        recon = F.binary_cross_entropy(gen, (labels * 0.82) + default + dist)
    out = torch.gt(gen, 0.3).to(gen.dtype)
    extra = torch.gt(labels - out, 0.0001).to(gen.dtype).sum() * 0.005
    return recon + extra


def dloss(outputs, targets):
    """graph/loss/bar_loss.py:41-42"""
    return F.binary_cross_entropy(outputs, targets)


def reparameterize(mean, logvar, eps):
    """old/graphs/models/bar_v1/encoder.py:60-63 with the noise passed in."""
    return mean + eps * torch.exp(0.5 * logvar)


def kl_term(mean, logvar):
    """old/graphs/losses/loss.py:14-17 -- -0.5 * sum(1 + logvar - mean^2 - exp(logvar))"""
    return -0.5 * torch.sum(1 + logvar - mean.pow(2) - logvar.exp())


# ------------------------------------------------------------ one training step (CPU)
def pretrain_step_loss(gsd, zb_sd, zp_sd, note, pre_note, phrase, position, is_pretraining=True,
                       train=False, drop_masks=None):
    """agent/barGen2.py:267-288: generator forward, three frozen z-discriminator
    forwards with 'valid' targets, Loss.  Returns the scalar the agent back-propagates."""
    gen, z, pre_z, pf = generator_train(gsd, note, pre_note, phrase, position, train, drop_masks)
    valid = torch.ones(note.size(0), dtype=gen.dtype, device=gen.device)
    loss = dloss(z_discriminator(zp_sd, "", pf).view(-1), valid)
    loss = loss + dloss(z_discriminator(zb_sd, "", z).view(-1), valid) + dloss(z_discriminator(zb_sd, "", pre_z).view(-1), valid)
    loss = loss + bar_loss(gen, note, is_pretraining)
    return loss, gen


def adam_step(params, grads, m, v, step, lr=0.002, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (agent/barGen2.py:60) written out; in place."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    with torch.no_grad():
        for p, g, mi, vi in zip(params, grads, m, v):
            mi.mul_(b1).add_(g, alpha=1 - b1)
            vi.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (vi.sqrt() / (bc2 ** 0.5)).add_(eps)
            p.addcdiv_(mi, denom, value=-lr / bc1)
