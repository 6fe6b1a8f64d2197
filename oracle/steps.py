"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's agent iterations (agent/barGen.py adversarial
branch, agent/barGen_with_gan.py train_wae / train_gan, agent/barGen_with_gan2.py train_discriminator + train_add_gan)
and of the sampling loop.

The reference's agents cannot be imported on Python >= 3.7 (``async=`` keyword, SURVEY defect D1) and hard-code
``.cuda()`` (D3), so these functions restate their per-iteration arithmetic from the source text on top of the
forwards in oracle/restate.py (which ARE pinned bit-for-bit to the reference import).  PARITY UNPINNED as whole
steps: no reference test or runnable reference produces a value for them; pinned by composition.

Everything that the reference draws from an RNG is an argument here (prior noise, dropout masks), so the HIP path can
be run on the same draws.  Optimizers are torch.optim.Adam defaults with lr = config.learning_rate
(agent/barGen_with_gan.py:63-77), written out in restate.adam_step.
"""
import torch

from . import restate as R


def _params(sd):
    """names of the trainable entries of a state-dict style mapping (floating point, not BatchNorm statistics)"""
    return [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]


def leaf_copy(sd, dtype=torch.float64):
    """independent copy of ``sd``: parameters become autograd leaves of ``dtype``, BatchNorm statistics plain tensors"""
    out = {}
    for k, v in sd.items():
        if not v.is_floating_point():
            out[k] = v.clone()
        elif "running_" in k:
            out[k] = v.clone().to(dtype)
        else:
            out[k] = v.clone().to(dtype).requires_grad_(True)
    return out


class AdamState:
    """torch.optim.Adam(lr) over the parameters of one state dict (restate.adam_step), stepping in place"""

    def __init__(self, sd, lr):
        self.sd, self.lr, self.names = sd, lr, _params(sd)
        self.m = {n: torch.zeros_like(sd[n]) for n in self.names}
        self.v = {n: torch.zeros_like(sd[n]) for n in self.names}
        self.t = 0

    def step(self, grads):
        """``grads``: name -> tensor or None (torch skips parameters whose grad is None)"""
        self.t += 1
        names = [n for n in self.names if grads.get(n) is not None]
        R.adam_step([self.sd[n] for n in names], [grads[n] for n in names], [self.m[n] for n in names],
                    [self.v[n] for n in names], self.t, lr=self.lr)


def _grads(loss, sd):
    names = _params(sd)
    g = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True, retain_graph=True)
    return dict(zip(names, g))


def _add(a, b):
    """sum of two gradient dicts (loss_a.backward(); loss_b.backward() accumulate into .grad)"""
    out = dict(a)
    for k, v in b.items():
        if v is not None:
            out[k] = v if out.get(k) is None else out[k] + v
    return out


def _ones(x):
    return torch.ones(x.shape[0], dtype=x.dtype)


def _zeros(x):
    return torch.zeros(x.shape[0], dtype=x.dtype)


def wae_iteration(gsd, zb_sd, zp_sd, opts, batch, noise, drop_masks, run_disc):
    """agent/barGen_with_gan.py:381-460 (train_wae), one iteration.

    gsd / zb_sd / zp_sd: leaf_copy'd generator and latent-discriminator state dicts (updated IN PLACE like the
    reference's modules); opts: {"generator", "z_bar", "z_phrase"} -> AdamState; noise = (phrase_fake, bar_fake), the
    two N(0, sigma^2) draws of :399,:406 in that order; drop_masks: the decoder's two dropout masks (the generator is
    in train() mode in both of its forwards); run_disc: (epoch + curr_it) % 2 of :394.
    Returns the quantities the HIP agent is compared on."""
    note, pre_note, pre_phrase, position = batch
    out = {}
    if run_disc:
        # generator frozen: its outputs carry no graph (:396-402)
        with torch.no_grad():
            _, z, pre_z, pf, _ = R.generator_gan(gsd, note, pre_note, pre_phrase, position, True, True, drop_masks)
        d_phrase_fake = R.z_discriminator(zp_sd, "", noise[0]).view(-1)
        d_phrase_real = R.z_discriminator(zp_sd, "", pf).view(-1)
        phrase_loss = R.dloss(d_phrase_real, _zeros(pf)) + R.dloss(d_phrase_fake, _ones(pf))          # :404-405
        d_bar_fake = R.z_discriminator(zb_sd, "", noise[1]).view(-1)
        d_bar_real = R.z_discriminator(zb_sd, "", z).view(-1)
        bar_loss = R.dloss(d_bar_real, _zeros(z)) + R.dloss(d_bar_fake, _ones(z))                      # :411
        out["phrase_loss"], out["bar_loss"] = phrase_loss.detach(), bar_loss.detach()
        out["grad_z_phrase"] = _grads(phrase_loss, zp_sd)
        out["grad_z_bar"] = _grads(bar_loss, zb_sd)
        opts["z_bar"].step(out["grad_z_bar"])                                                          # :422-423
        opts["z_phrase"].step(out["grad_z_phrase"])
    # generator step against the (just updated) frozen latent discriminators (:426-452)
    gen, z, pre_z, pf, _ = R.generator_gan(gsd, note, pre_note, pre_phrase, position, True, True, drop_masks)
    zp = {k: v.detach() for k, v in zp_sd.items()}
    zb = {k: v.detach() for k, v in zb_sd.items()}
    loss = R.dloss(R.z_discriminator(zp, "", pf).view(-1), _ones(pf))
    loss = loss + R.dloss(R.z_discriminator(zb, "", z).view(-1), _ones(z)) + R.dloss(R.z_discriminator(zb, "", pre_z).view(-1), _ones(z))
    loss = loss + R.bar_loss(gen, note, False)
    out["generator_loss"], out["gen"] = loss.detach(), gen.detach()
    out["grad_generator"] = _grads(loss, gsd)
    opts["generator"].step(out["grad_generator"])
    return out


def gan_iteration(gsd, d_sd, f_sd, opts, batch, noise, drop_masks, run_disc):
    """agent/barGen_with_gan.py:462-537 (train_gan), one iteration.

    d_sd: BarDiscriminator state dict (BatchNorm running statistics are updated in place: the module is in train()
    mode in the discriminator step AND in the generator step, :466-467); f_sd: BarFeatureDiscriminator; opts:
    {"generator", "discriminator", "discriminator_feature"}; noise: the N(0, 1.5^2) latent of :517."""
    note, pre_note, pre_phrase, position = batch
    out = {}
    if run_disc:
        with torch.no_grad():           # generator frozen (:479-483)
            gen, z, pre_z, pf, gen_z = R.generator_gan(gsd, note, pre_note, pre_phrase, position, True, True, drop_masks)
        fake_pair = torch.cat((pre_note, gen), dim=2)                                                   # :486-487
        real_pair = torch.cat((pre_note, note), dim=2)
        d_note_fake = R.bar_discriminator(d_sd, "", fake_pair, train=True).view(-1)                     # fake first
        d_note_real = R.bar_discriminator(d_sd, "", real_pair, train=True).view(-1)
        note_loss = R.dloss(d_note_real, _zeros(z)) + R.dloss(d_note_fake, _ones(z))                    # :490
        d_feat_fake = R.bar_feature_discriminator(f_sd, "", gen_z).view(-1)
        d_feat_real = R.bar_feature_discriminator(f_sd, "", z).view(-1)
        feat_loss = R.dloss(d_feat_real, _zeros(z)) + R.dloss(d_feat_fake, _ones(z))                    # :495-496
        out["note_loss"], out["feature_loss"] = note_loss.detach(), feat_loss.detach()
        out["grad_discriminator"] = _grads(note_loss, d_sd)
        out["grad_discriminator_feature"] = _grads(feat_loss, f_sd)
        opts["discriminator"].step(out["grad_discriminator"])                                           # :503-504
        opts["discriminator_feature"].step(out["grad_discriminator_feature"])
    # generator step from prior noise, judged by both (frozen, train-mode) discriminators (:507-529)
    gen, gen_z = R.generator_gan(gsd, noise, pre_note, pre_phrase, position, False, True, drop_masks)
    dd = {k: (v.detach() if v.is_floating_point() and "running_" not in k else v) for k, v in d_sd.items()}
    ff = {k: v.detach() for k, v in f_sd.items()}
    loss = R.dloss(R.bar_discriminator(dd, "", torch.cat((pre_note, gen), dim=2), train=True).view(-1), _ones(gen_z))
    loss = loss + R.dloss(R.bar_feature_discriminator(ff, "", gen_z).view(-1), _ones(gen_z))
    out["generator_loss"], out["gen"], out["gen_z"] = loss.detach(), gen.detach(), gen_z.detach()
    out["grad_generator"] = _grads(loss, gsd)
    opts["generator"].step(out["grad_generator"])
    return out


def bargen_iteration(gsd, d_sd, zb_sd, zp_sd, opts, batch, noise, drop_masks, run_disc):
    """agent/barGen.py:254-327, one ADVERSARIAL iteration (epoch > pretraining_step_size) of the first agent.

    The generator is graph.model.Model (4-tuple; Refiner left out, defect D2) in train() mode in both of its forwards
    (:218); BarDiscriminator is in train() mode throughout (:219): its BatchNorm statistics move in the discriminator
    block (fake pair first, then real: :281-284) AND in the generator block (:320-322), where its parameters are frozen.
    This agent labels real -> valid and prior -> fake (:268,:275-276), the opposite of barGen2, and draws its priors
    with sigma = 1 (:265,:271).  The bar discriminator sees the fake bar BINARISED at 0.3 (:279,:320): no gradient
    reaches the generator through it.  Defect D7 restated: the smoothed reconstruction loss of :314 is overwritten at
    :316, so the generator's adversarial loss is the three latent terms + the (gradient-free) bar term, and only the
    encoders receive a gradient; ``opt_gen2`` steps (:327).
    opts: {"gen2", "discriminator", "z_bar", "z_phrase"}; noise = (phrase_fake [B, 1152], bar_fake [2B, 1152])."""
    note, pre_note, pre_phrase, position = batch
    out = {}
    if run_disc:                                                                                         # :254
        with torch.no_grad():                                                                            # generator frozen (:260-262)
            gen, z, pre_z, pf = R.generator_train(gsd, note, pre_note, pre_phrase, position, True, drop_masks)
        d_phrase_fake = R.z_discriminator(zp_sd, "", noise[0]).view(-1)                                  # :265-267
        d_phrase_real = R.z_discriminator(zp_sd, "", pf).view(-1)
        phrase_loss = R.dloss(d_phrase_real, _ones(pf)) + R.dloss(d_phrase_fake, _zeros(pf))             # :268
        d_bar_fake = R.z_discriminator(zb_sd, "", noise[1]).view(-1)                                     # :271-274
        d_bar_real1 = R.z_discriminator(zb_sd, "", z).view(-1)
        d_bar_real2 = R.z_discriminator(zb_sd, "", pre_z).view(-1)
        bar_loss = R.dloss(d_bar_real1, _ones(z)) + R.dloss(d_bar_real2, _ones(z)) + R.dloss(d_bar_fake, _zeros(noise[1]))   # :275-276
        fake = torch.cat((pre_note, torch.gt(gen, 0.3).to(gen.dtype)), dim=2)                            # :279-280
        d_fake = R.bar_discriminator(d_sd, "", fake, train=True).view(-1)                                # :281
        d_real = R.bar_discriminator(d_sd, "", torch.cat((pre_note, note), dim=2), train=True).view(-1)  # :283-284
        disc_loss = R.dloss(d_fake, _zeros(z)) + R.dloss(d_real, _ones(z))                               # :286
        out["disc_loss"], out["phrase_loss"], out["bar_loss"] = disc_loss.detach(), phrase_loss.detach(), bar_loss.detach()
        out["grad_discriminator"] = _grads(disc_loss, d_sd)                                              # :289-291
        out["grad_z_phrase"] = _grads(phrase_loss, zp_sd)
        out["grad_z_bar"] = _grads(bar_loss, zb_sd)
        opts["discriminator"].step(out["grad_discriminator"])                                            # :293-295
        opts["z_bar"].step(out["grad_z_bar"])
        opts["z_phrase"].step(out["grad_z_phrase"])
    # generator block (:299-327) against the just-updated, frozen discriminators
    gen, z, pre_z, pf = R.generator_train(gsd, note, pre_note, pre_phrase, position, True, drop_masks)
    zp = {k: v.detach() for k, v in zp_sd.items()}
    zb = {k: v.detach() for k, v in zb_sd.items()}
    dd = {k: (v.detach() if v.is_floating_point() and "running_" not in k else v) for k, v in d_sd.items()}
    loss = R.dloss(R.z_discriminator(zp, "", pf).view(-1), _ones(pf))                                    # :316 (overwrites :314, D7)
    loss = loss + R.dloss(R.z_discriminator(zb, "", z).view(-1), _ones(z)) + R.dloss(R.z_discriminator(zb, "", pre_z).view(-1), _ones(z))
    fake = torch.cat((pre_note, torch.gt(gen, 0.3).to(gen.dtype)), dim=2)                                # :320-321
    loss = loss + R.dloss(R.bar_discriminator(dd, "", fake, train=True).view(-1), _ones(z))              # :322-324
    out["generator_loss"], out["gen"] = loss.detach(), gen.detach()
    out["grad_generator"] = _grads(loss, gsd)
    opts["gen2"].step(out["grad_generator"])                                                             # :327
    return out


def gan2_iteration(gsd, d_sd, f_sd, zb_sd, zp_sd, opts, batch, noise, drop_masks, sigma=1.0):
    """agent/barGen_with_gan2.py:345-404 (train_discriminator) followed by :468-519 (train_add_gan): what every iteration
    after pre-training runs (:287-293).

    Modes: train_add_gan puts the generator and the two latent discriminators in train() mode and BarDiscriminator /
    BarFeatureDiscriminator in eval() mode (:469-474), and nothing ever switches the latter two back: BarDiscriminator's
    BatchNorm therefore normalises with its RUNNING statistics in every pass of this agent, and they never move.
    train_discriminator itself sets no mode: it runs with what the previous iteration's train_add_gan left (the generator
    in train() mode: dropout on) -- restated here; the first iteration of an epoch, which follows the all-eval sampling
    block of :308-313, differs only in the generator's dropout being off.
    This agent labels real -> fake_target (0) and fake / prior -> valid_target (1) in all four discriminators.
    noise = (phrase_fake [B,1152], bar_fake [B,1152], both already scaled by config.sigma (:358,:365); gan_noise [B,1152]
    ~ N(0, 1.5^2) of :503).  opts: {"generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase"}."""
    note, pre_note, pre_phrase, position = batch
    out = {}
    # ---- train_discriminator (:345-404)
    with torch.no_grad():                                                                                # generator frozen (:353-355)
        gen, z, pre_z, pf, gen_z = R.generator_gan(gsd, note, pre_note, pre_phrase, position, True, True, drop_masks)
    d_phrase_fake = R.z_discriminator(zp_sd, "", noise[0]).view(-1)                                      # :358-360
    d_phrase_real = R.z_discriminator(zp_sd, "", pf).view(-1)
    phrase_loss = R.dloss(d_phrase_real, _zeros(pf)) + R.dloss(d_phrase_fake, _ones(pf))                 # :361-362
    d_bar_fake = R.z_discriminator(zb_sd, "", noise[1]).view(-1)                                         # :365-367
    d_bar_real = R.z_discriminator(zb_sd, "", z).view(-1)
    bar_loss = R.dloss(d_bar_real, _zeros(z)) + R.dloss(d_bar_fake, _ones(z))                            # :368
    binar = torch.gt(gen, 0.3).to(gen.dtype)                                                             # :371
    d_note_fake = R.bar_discriminator(d_sd, "", torch.cat((pre_note, binar), dim=2), train=False).view(-1)     # :372-374
    d_note_real = R.bar_discriminator(d_sd, "", torch.cat((pre_note, note), dim=2), train=False).view(-1)      # :376
    note_loss = R.dloss(d_note_real, _zeros(z)) + R.dloss(d_note_fake, _ones(z))                         # :376
    d_feat_fake = R.bar_feature_discriminator(f_sd, "", gen_z).view(-1)                                  # :379-380
    d_feat_real = R.bar_feature_discriminator(f_sd, "", z).view(-1)
    feat_loss = R.dloss(d_feat_real, _zeros(z)) + R.dloss(d_feat_fake, _ones(z))                         # :381-382
    out["phrase_loss"], out["bar_loss"] = phrase_loss.detach(), bar_loss.detach()
    out["note_loss"], out["feature_loss"] = note_loss.detach(), feat_loss.detach()
    out["grad_z_phrase"] = _grads(phrase_loss, zp_sd)                                                    # :385-388
    out["grad_z_bar"] = _grads(bar_loss, zb_sd)
    out["grad_discriminator"] = _grads(note_loss, d_sd)
    out["grad_discriminator_feature"] = _grads(feat_loss, f_sd)
    opts["z_bar"].step(out["grad_z_bar"])                                                                # :390-393
    opts["z_phrase"].step(out["grad_z_phrase"])
    opts["discriminator"].step(out["grad_discriminator"])
    opts["discriminator_feature"].step(out["grad_discriminator_feature"])
    # ---- train_add_gan (:468-519): every discriminator frozen, at its just-updated weights
    zp = {k: v.detach() for k, v in zp_sd.items()}
    zb = {k: v.detach() for k, v in zb_sd.items()}
    dd = {k: (v.detach() if v.is_floating_point() else v) for k, v in d_sd.items()}
    ff = {k: v.detach() for k, v in f_sd.items()}
    gen, z, pre_z, pf, _ = R.generator_gan(gsd, note, pre_note, pre_phrase, position, True, True, drop_masks)   # :488
    loss = R.dloss(R.z_discriminator(zp, "", pf).view(-1), _ones(pf))                                    # :491-492
    loss = loss + R.dloss(R.z_discriminator(zb, "", z).view(-1), _ones(z)) + R.dloss(R.z_discriminator(zb, "", pre_z).view(-1), _ones(z))   # :495-497
    loss = loss + R.bar_loss(gen, note, False)                                                           # :500
    gen2, gen_z = R.generator_gan(gsd, noise[2], pre_note, pre_phrase, position, False, True, drop_masks)        # :503-504
    binar = torch.gt(gen2, 0.3).to(gen2.dtype)                                                           # :506
    d_note_fake = R.bar_discriminator(dd, "", torch.cat((pre_note, binar), dim=2), train=False).view(-1)         # :507
    loss = loss + R.dloss(d_note_fake, _ones(z)) * 0.05                                                  # :508
    loss = loss + R.dloss(R.bar_feature_discriminator(ff, "", gen_z).view(-1), _ones(z)) * 0.05          # :510-511
    out["generator_loss"], out["gen"], out["gen_z"] = loss.detach(), gen2.detach(), gen_z.detach()      # returns gen_note[:3] of :504
    out["grad_generator"] = _grads(loss, gsd)
    opts["generator"].step(out["grad_generator"])                                                        # :513-515
    return out


def sample_phrases(gsd, latents, music_length, songs=1, refiner=False):
    """maker_bar.py:31-44 (== agent/barGen2.py:317-336) with the prior draws passed in: ``latents[idx][bar]`` is the
    [songs, 1152] latent of bar ``bar`` of phrase ``idx``.  Songs are independent batch entries.  ``refiner``: apply the
    D2-fixed Refiner (graph/refiner.py:49-58 through graph/model.py:39-40; ``gsd`` then carries its ``refiner.*`` entries) in
    eval mode like the rest of the generator (agent/barGen2.py:318) -- BASELINE.json configs[4] as worded; without it the
    loop is the refiner-less generator of the timed training configs (defect D2).
    Returns ([songs, music_length * 384, 60] binary roll, list of every bar's output in (0, 1))."""
    dt = latents[0][0].dtype
    pre_phrase = torch.zeros(songs, 1, 384, 60, dtype=dt)
    pre_bar = torch.zeros(songs, 1, 96, 60, dtype=dt)
    phrase_idx = [330] + [i for i in range(music_length - 2, -1, -1)]
    outputs, raw = [], []
    with torch.no_grad():
        for idx in range(music_length):
            pos = torch.full((songs,), phrase_idx[idx], dtype=torch.long)
            bars = []
            for b in range(4):
                gen = R.generator_sample(gsd, latents[idx][b], pre_bar, pre_phrase, pos)
                if refiner:
                    gen = R.refiner(gsd, "refiner.", gen, train=False)
                raw.append(gen)
                pre_bar = torch.gt(gen, 0.3).to(dt)
                bars.append(pre_bar.reshape(songs, 96, 60))
            phrase = torch.cat(bars, dim=1)
            outputs.append(phrase)
            pre_phrase = phrase.reshape(songs, 1, 384, 60)
    return torch.cat(outputs, dim=1), raw
