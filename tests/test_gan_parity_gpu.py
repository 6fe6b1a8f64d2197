"""GPU suite (-m gpu): the parts of BASELINE.json configs 4-5 that round 1 only smoke-ran, against the oracle.

  * graph.model_with_gan.Model.forward, both modes (reference graph/model_with_gan.py:20-38) vs
    oracle.restate.generator_gan on both weight sets -- gen, z, pre_z, phrase_feature, gen_z, the 0.3 threshold;
  * graph.model.Model.forward(is_train=False) and maker_bar.sample (reference graph/model.py:34-41,
    maker_bar.py:31-44) at 32 songs vs an oracle loop on oracle.restate.generator_sample with injected prior noise;
  * one train_wae and one train_gan iteration through the AGENT'S OWN methods (agent/barGen_with_gan.py ==
    reference agent/barGen_with_gan.py:381-537) vs oracle.steps: every loss value, every parameter gradient of every
    network that steps, BatchNorm running statistics, and the Adam update.

The whole-step oracles are restated from the reference's source text (its agents cannot be imported: SURVEY D1/D3) on
top of forwards that are pinned bit-for-bit to the reference import (oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import restate as R
from oracle import steps as S
from oracle import weights as W
from parity_util import REPORT, TOL, check, check_grad, flush_report, pop_margins, rel

pytestmark = pytest.mark.gpu
dev = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _env():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available(), "GPU suite needs a ROCm device"
    yield
    flush_report()


def _gan_generator(mode):
    from graph.model_with_gan import Model
    gsd = W.make_state_dict(W.manifest_generator(), 0, mode)
    m = Model()
    m.load_state_dict(gsd)
    return m.to(dev).eval(), gsd


@pytest.mark.parametrize("mode", ["wc", "d4"])
def test_model_with_gan_forward_both_modes(mode):
    m, gsd = _gan_generator(mode)
    g64 = {k: v.double() for k, v in gsd.items()}
    note, pre, phrase, pos = W.make_inputs(4, seed=77)
    tol = TOL if mode == "wc" else 7e-3       # d4 decoder: SURVEY section 7 (torch fp32 itself is 2.6e-3 from fp64 there)
    # ---- is_note=True: the 5-tuple
    with torch.no_grad():
        gen, z, pz, pf, gz = m(note.to(dev), pre.to(dev), phrase.to(dev), pos.to(dev))
        og, oz, opz, opf, ogz = R.generator_gan(g64, note.double(), pre.double(), phrase.double(), pos, True)
        o32 = R.generator_gan(gsd, note, pre, phrase, pos, True)
    check("gan-model[%s] z" % mode, z, oz); check("gan-model[%s] pre_z" % mode, pz, opz)
    check("gan-model[%s] phrase_feature" % mode, pf, opf)
    check("gan-model[%s] gen (torch fp32 itself: %.2e)" % (mode, rel(o32[0], og)), gen, og, max(tol, 2 * rel(o32[0], og)))
    # the threshold: bit-exact wherever |gen - 0.3| is outside fp32 noise of the decoder output
    fake_h = (gen > 0.3).cpu()
    fake_o = og > 0.3
    far = (og - 0.3).abs() > max(tol, 2 * rel(o32[0], og)) * og.abs().max()
    assert torch.equal(fake_h[far], fake_o[far]), "threshold differs away from 0.3"
    flips = int((fake_h != fake_o).sum())
    REPORT.append("gan-model[%s] is_note=True: %d of %d cells binarise differently from fp64 (all within noise of 0.3)" % (mode, flips, gen.numel()))
    # gen_z: the third encoder pass over the binarised bar -- compared on the HIP path's OWN binary bar (exact input)
    with torch.no_grad():
        want_gz = R.encoder(g64, "encoder.", fake_h.double())
    check("gan-model[%s] gen_z = encoder(gen > 0.3)" % mode, gz, want_gz)
    if flips == 0:
        check("gan-model[%s] gen_z vs free-running oracle" % mode, gz, ogz)
    # ---- is_note=False: latent in, 2-tuple out
    lat = torch.randn(4, 1152, generator=torch.Generator().manual_seed(5)) * 1.5
    with torch.no_grad():
        gen2, gz2 = m(lat.to(dev), pre.to(dev), phrase.to(dev), pos.to(dev), False)
        og2, ogz2 = R.generator_gan(g64, lat.double(), pre.double(), phrase.double(), pos, False)
        o32b = R.generator_gan(gsd, lat, pre, phrase, pos, False)
    t2 = max(tol, 2 * rel(o32b[0], og2))
    check("gan-model[%s] is_note=False gen (torch fp32 itself: %.2e)" % (mode, rel(o32b[0], og2)), gen2, og2, t2)
    fh2 = (gen2 > 0.3).cpu()
    far2 = (og2 - 0.3).abs() > t2 * og2.abs().max()
    assert torch.equal(fh2[far2], (og2 > 0.3)[far2])
    with torch.no_grad():
        check("gan-model[%s] is_note=False gen_z" % mode, gz2, R.encoder(g64, "encoder.", fh2.double()))


def test_sampling_forward_and_loop_at_32_songs(monkeypatch):
    """config 5's arithmetic: Model.forward(is_train=False) teacher-forced on the oracle's states, then the whole
    free-running maker_bar.sample loop (2 phrases x 4 bars, 32 independent songs) on the same prior draws."""
    import maker_bar
    from graph.model import Model
    songs, length = 32, 2
    gsd = W.make_state_dict(W.manifest_generator(), 0, "wc")
    gen = Model()
    gen.load_state_dict(gsd)
    gen = gen.to(dev).eval()
    g = torch.Generator().manual_seed(2024)
    lat = [[torch.randn(songs, 1152, generator=g) for _ in range(4)] for _ in range(length)]
    want_roll, raw = S.sample_phrases(gsd, lat, length, songs)          # fp32 oracle (the reference's arithmetic)
    # teacher-forced single calls: the oracle's (binary, hence exact) states in, sigmoid output compared
    pre_phrase = torch.zeros(songs, 1, 384, 60); pre_bar = torch.zeros(songs, 1, 96, 60)
    phrase_idx = [330] + list(range(length - 2, -1, -1))
    n = 0
    for idx in range(length):
        pos = torch.full((songs,), phrase_idx[idx], dtype=torch.long)
        for b in range(4):
            with torch.no_grad():
                out = gen(lat[idx][b].to(dev), pre_bar.to(dev), pre_phrase.to(dev), pos.to(dev), False)
            check("sampling forward phrase %d bar %d (32 songs, teacher-forced)" % (idx, b), out, raw[n])
            pre_bar = (raw[n] > 0.3).float()
            n += 1
        pre_phrase = want_roll[:, idx * 384:(idx + 1) * 384].reshape(songs, 1, 384, 60)
    # free-running loop on the same draws
    draws = [t for ph in lat for t in ph]
    it = iter(draws)
    monkeypatch.setattr(maker_bar.HF, "randn", lambda shape, sigma=1.0, device="cuda", out=None: next(it).to(device) * sigma)
    roll = maker_bar.sample(gen, music_length=length, songs=songs, device=dev).cpu()
    assert tuple(roll.shape) == (songs, length * 384, 60)
    diff = int((roll != want_roll).sum())
    REPORT.append("sampling loop 32 songs x %d phrases: %d of %d cells differ from the oracle loop" % (length, diff, roll.numel()))
    # An autoregressive loop amplifies a single threshold flip: a sigmoid output within fp32 noise of 0.3 binarises the
    # other way, the next bar is conditioned on a different previous bar, and that song's rolls part ways for good.
    # Songs are independent batch entries, so every song that differs must be EXPLAINED by such a flip: in the first bar
    # where it differs, every differing cell's oracle output lies within 1e-4 of the threshold (the single calls above
    # pin the forward at 1e-3 of the largest output; measured 1e-6 .. 1e-5); at most 3 of the 32 songs may do so.
    bars_h = roll.reshape(songs, length * 4, 96, 60)
    bars_o = want_roll.reshape(songs, length * 4, 96, 60)
    parted = 0
    for sng in range(songs):
        bad = [b for b in range(length * 4) if not torch.equal(bars_h[sng, b], bars_o[sng, b])]
        if not bad:
            continue
        parted += 1
        b0 = bad[0]
        cells = bars_h[sng, b0] != bars_o[sng, b0]
        margin = (raw[b0][sng, 0][cells] - 0.3).abs().max().item()
        REPORT.append("    song %d parts ways in bar %d: %d cells, farthest oracle output %.2e from the 0.3 threshold" % (sng, b0, int(cells.sum()), margin))
        assert margin < 1e-4, (sng, b0, margin)
    assert parted <= 3, parted


def test_sampling_loop_with_refiner_at_32_songs(monkeypatch):
    """BASELINE.json configs[4] as it is worded -- "phrase_encoder + REFINER long-sequence sampling, batch=32": the
    generator with the D2-fixed Refiner (graph/refiner.py:49-58 with layer2 taking 2 channels; parity unpinned, the
    reference raises) in the maker_bar.py:31-44 loop at 32 songs, against the oracle loop that applies restate.refiner
    after every decoder call.  Teacher-forced single calls first (the refined output is what gets thresholded), then the
    free-running loop with the same explained-flip rule as the refiner-less test above."""
    import maker_bar
    from graph.model import Model
    songs, length = 32, 2
    gsd = W.make_state_dict(W.manifest_generator(), 0, "wc")
    gsd.update(W.make_state_dict(W.manifest_refiner("refiner."), 3, "wc"))
    # informative BatchNorm statistics for the eval-mode refiner (the manifest's defaults are mean 0 / var 1)
    gq = torch.Generator().manual_seed(77)
    for k in list(gsd):
        if k.startswith("refiner.") and k.endswith("running_mean"):
            gsd[k] = 0.1 * torch.randn(gsd[k].shape, generator=gq)
        if k.startswith("refiner.") and k.endswith("running_var"):
            gsd[k] = 0.5 + torch.rand(gsd[k].shape, generator=gq)
    gen = Model(use_refiner=True)
    gen.load_state_dict(gsd)
    gen = gen.to(dev).eval()
    g = torch.Generator().manual_seed(2025)
    lat = [[torch.randn(songs, 1152, generator=g) for _ in range(4)] for _ in range(length)]
    want_roll, raw = S.sample_phrases(gsd, lat, length, songs, refiner=True)
    with torch.no_grad():           # the refiner must matter for this check to mean anything
        first_plain = R.generator_sample(gsd, lat[0][0], torch.zeros(songs, 1, 96, 60), torch.zeros(songs, 1, 384, 60),
                                         torch.full((songs,), 330, dtype=torch.long))
    assert float((first_plain - raw[0]).abs().max()) > 1e-2
    pre_phrase = torch.zeros(songs, 1, 384, 60); pre_bar = torch.zeros(songs, 1, 96, 60)
    phrase_idx = [330] + list(range(length - 2, -1, -1))
    n = 0
    for idx in range(length):
        pos = torch.full((songs,), phrase_idx[idx], dtype=torch.long)
        for b in range(4):
            with torch.no_grad():
                out = gen(lat[idx][b].to(dev), pre_bar.to(dev), pre_phrase.to(dev), pos.to(dev), False)
            check("sampling+refiner forward phrase %d bar %d (32 songs, teacher-forced)" % (idx, b), out, raw[n])
            pre_bar = (raw[n] > 0.3).float()
            n += 1
        pre_phrase = want_roll[:, idx * 384:(idx + 1) * 384].reshape(songs, 1, 384, 60)
    draws = [t for ph in lat for t in ph]
    it = iter(draws)
    monkeypatch.setattr(maker_bar.HF, "randn", lambda shape, sigma=1.0, device="cuda", out=None: next(it).to(device) * sigma)
    roll = maker_bar.sample(gen, music_length=length, songs=songs, device=dev).cpu()
    assert tuple(roll.shape) == (songs, length * 384, 60)
    diff = int((roll != want_roll).sum())
    REPORT.append("sampling loop WITH refiner, 32 songs x %d phrases: %d of %d cells differ from the oracle loop" % (length, diff, roll.numel()))
    bars_h = roll.reshape(songs, length * 4, 96, 60)
    bars_o = want_roll.reshape(songs, length * 4, 96, 60)
    parted = 0
    for sng in range(songs):
        bad = [b for b in range(length * 4) if not torch.equal(bars_h[sng, b], bars_o[sng, b])]
        if not bad:
            continue
        parted += 1
        b0 = bad[0]
        cells = bars_h[sng, b0] != bars_o[sng, b0]
        margin = (raw[b0][sng, 0][cells] - 0.3).abs().max().item()
        REPORT.append("    song %d parts ways in bar %d: %d cells, farthest oracle output %.2e from the 0.3 threshold" % (sng, b0, int(cells.sum()), margin))
        assert margin < 1e-4, (sng, b0, margin)
    assert parted <= 3, parted
    # the HIP-graph sampler runs the same two programs
    it2 = iter(draws)
    monkeypatch.setattr(maker_bar.HF, "randn",
                        lambda shape, sigma=1.0, device="cuda", out=None: (out.copy_(next(it2).to(device) * sigma) if out is not None else next(it2).to(device) * sigma))
    gs = maker_bar.GraphSampler(gen, songs=songs, device=dev)
    it2 = iter(draws)
    roll_g = gs.sample(length).cpu()
    same = float((roll_g == roll).float().mean())
    REPORT.append("GraphSampler with refiner vs eager loop: %.6f of the cells equal" % same)
    assert same > 0.999


# ------------------------------------------------------------------------------------------ agent iterations
def _agent(tmp_path, monkeypatch, compute_dtype="f32", which="barGen_with_gan"):
    import importlib
    from test_agent_gpu import _make_dataset
    from config import Config
    BarGen = importlib.import_module("agent." + which).BarGen
    root = str(tmp_path)
    _make_dataset(root, n_files=2, per_file=2)

    class Cfg(Config):
        root_path = root
        batch_size = 2
        epoch = 1
        pretraining_step_size = 0
        seed = 5
        log_file = os.path.join(root, "train_epoch.log")

    Cfg.compute_dtype = compute_dtype
    agent = BarGen(Cfg())
    sds = {"generator": W.make_state_dict(W.manifest_generator(), 0, "wc"),
           "discriminator": W.make_state_dict(W.manifest_bar_discriminator(), 0, "wc"),
           "discriminator_feature": W.make_state_dict(W.manifest_bar_feature_discriminator(), 0, "wc"),
           "z_discriminator_bar": W.make_state_dict(W.manifest_z_discriminator(), 1, "wc"),
           "z_discriminator_phrase": W.make_state_dict(W.manifest_z_discriminator(), 2, "wc")}
    nets = getattr(agent, "nets", None)
    if nets is None:        # agent/barGen.py: no feature discriminator, two Adam states over the generator
        nets = {"discriminator": agent.net_disc, "z_discriminator_bar": agent.net_zbar, "z_discriminator_phrase": agent.net_zphrase}
        sds.pop("discriminator_feature")
    for n, sd in sds.items():
        getattr(agent, n).load_state_dict(sd)
    # record the flat gradient of every network at the moment it steps
    grads = {}
    if which == "barGen":
        def gen_step(opt, orig=agent._gen_step):
            torch.cuda.synchronize()
            grads["generator"] = {k: p.grad.detach().clone() for k, p in agent.generator.named_parameters()}
            grads["generator_opt"] = "gen2" if opt is agent.opt_gen2 else "gen1"
            orig(opt)
        agent._gen_step = gen_step
    for n, net in nets.items():
        def wrapped(net=net, n=n, orig=net.step):
            torch.cuda.synchronize()
            grads[n] = {k: p.grad.detach().clone() for k, p in net.module.named_parameters()}
            orig()
        net.step = wrapped
    return agent, sds, grads


def _masks(b, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(b, 1152, generator=g) >= 0.3).float() / 0.7 for _ in range(2)]


def _compare_net(tag, module, hip_grads, oracle_grads, ref32_grads, osd, sd0, lr, segment=None):
    """gradients of one network at its step (fp64 oracle is the judge, its fp32 run the reference's own arithmetic),
    then the Adam update the network received"""
    named = dict(module.named_parameters())
    gs = max(g.abs().max().item() for g in oracle_grads.values() if g is not None)
    names = [n for n, g in oracle_grads.items() if g is not None]
    if len(names) > 20:
        # the whole gradient as one vector: direction and size against fp64, next to what the reference's own fp32
        # arithmetic achieves (plain run and perturbed runs); the HIP step may not be further off than 3 x the worst of them
        flat = lambda d: torch.cat([d[n].detach().double().cpu().reshape(-1) for n in names])
        fo, fh = flat(oracle_grads), flat(hip_grads)
        cos_h = float((fh * fo).sum() / (fh.norm() * fo.norm())); l2_h = float((fh - fo).norm() / fo.norm())
        runs = len(next(v for v in ref32_grads.values() if v is not None)) if isinstance(next(v for v in ref32_grads.values() if v is not None), list) else 1
        worst_l2, worst_cos = 0.0, 1.0
        for r in range(runs):
            f32 = torch.cat([(ref32_grads[n][r] if runs > 1 or isinstance(ref32_grads[n], list) else ref32_grads[n]).detach().double().reshape(-1) for n in names])
            worst_l2 = max(worst_l2, float((f32 - fo).norm() / fo.norm())); worst_cos = min(worst_cos, float((f32 * fo).sum() / (f32.norm() * fo.norm())))
        REPORT.append("%s whole flat gradient vs fp64: hip cos %.8f l2-rel %.3e | torch fp32 (worst of %d runs) cos %.8f l2-rel %.3e" % (
            tag, cos_h, l2_h, runs, worst_cos, worst_l2))
        assert l2_h <= max(2e-3, 3 * worst_l2), (tag, l2_h, worst_l2)
        assert 1 - cos_h <= max(2e-6, 9 * (1 - worst_cos)), (tag, cos_h, worst_cos)
    for n, g in oracle_grads.items():
        if g is None:
            assert hip_grads[n].abs().max().item() == 0, (tag, n)
            continue
        if g.abs().max().item() <= 1e-6 * gs:
            continue
        check_grad("%s d%s" % (tag, n), hip_grads[n], g, 2 * TOL, atol=1e-6 * gs, ref32=ref32_grads[n],
                   segment=(segment or {}).get(n))
        # first Adam step: dw = -lr * g / (|g| + eps) -> compare where the gradient is clear of the noise floor: 1e-3 of the
        # tensor's largest entry, or ten times the gradient error just measured when the row passed through a relaxed rule
        big = g.abs() > max(1e-3, 10 * check_grad.last_rel) * g.abs().max()
        if not big.any():
            continue
        dw_h = (named[n].detach().cpu().double() - sd0[n].double())[big]
        dw_o = (osd[n].detach() - sd0[n].double())[big]
        bad = ((dw_h - dw_o).abs() > 0.02 * lr).double().mean().item()
        assert bad <= 2e-2, (tag, n, bad)


def _with_perturbed(o32, run, seeds=(101, 102, 103)):
    """the gradient dictionaries of the plain fp32 oracle run and of a few perturbed ones (weights moved by <= 2e-6 relative), row by row as lists
    (parity_util.check_grad takes the worst of them as the row's fp32 noise floor)"""
    extra = [run(s) for s in seeds]
    out = dict(o32)
    for k, v in o32.items():
        if k.startswith("grad_") and isinstance(v, dict):
            out[k] = {n: ([t] + [e[k][n] for e in extra] if t is not None else None) for n, t in v.items()}
    return out


def _oracle(kind, sds, lr, batch, noise, masks, dtype, perturb=None):
    osd = {n: S.leaf_copy(sd, dtype) for n, sd in sds.items()}
    if perturb is not None:
        # every weight moved by <= 2e-6 relative (2^-19): the size of an fp32 dot product's own accumulation error over the
        # model's K = 576 .. 4608 -- a stand-in for "the same arithmetic, summed in another order / on another pipe"
        g = torch.Generator().manual_seed(perturb)
        with torch.no_grad():
            for sd in osd.values():
                for t in sd.values():
                    if t.is_floating_point():
                        t.mul_(1.0 + 2.0 ** -19 * (2.0 * torch.rand(t.shape, generator=g, dtype=torch.float32).to(t.dtype) - 1.0))
    b = tuple(t.to(dtype) if t.is_floating_point() else t for t in batch)
    mk = [m.to(dtype) for m in masks]
    if kind == "bargen":
        opts = {"gen2": S.AdamState(osd["generator"], lr), "discriminator": S.AdamState(osd["discriminator"], lr),
                "z_bar": S.AdamState(osd["z_discriminator_bar"], lr), "z_phrase": S.AdamState(osd["z_discriminator_phrase"], lr)}
        o = S.bargen_iteration(osd["generator"], osd["discriminator"], osd["z_discriminator_bar"], osd["z_discriminator_phrase"],
                               opts, b, [t.to(dtype) for t in noise], mk, True)
    elif kind == "gan2":
        opts = {"generator": S.AdamState(osd["generator"], lr), "discriminator": S.AdamState(osd["discriminator"], lr),
                "discriminator_feature": S.AdamState(osd["discriminator_feature"], lr),
                "z_bar": S.AdamState(osd["z_discriminator_bar"], lr), "z_phrase": S.AdamState(osd["z_discriminator_phrase"], lr)}
        o = S.gan2_iteration(osd["generator"], osd["discriminator"], osd["discriminator_feature"], osd["z_discriminator_bar"],
                             osd["z_discriminator_phrase"], opts, b, [t.to(dtype) for t in noise], mk)
    elif kind == "wae":
        opts = {"generator": S.AdamState(osd["generator"], lr), "z_bar": S.AdamState(osd["z_discriminator_bar"], lr),
                "z_phrase": S.AdamState(osd["z_discriminator_phrase"], lr)}
        o = S.wae_iteration(osd["generator"], osd["z_discriminator_bar"], osd["z_discriminator_phrase"], opts, b,
                            [t.to(dtype) for t in noise], mk, True)
    else:
        opts = {"generator": S.AdamState(osd["generator"], lr), "discriminator": S.AdamState(osd["discriminator"], lr),
                "discriminator_feature": S.AdamState(osd["discriminator_feature"], lr)}
        o = S.gan_iteration(osd["generator"], osd["discriminator"], osd["discriminator_feature"], opts, b, noise.to(dtype), mk, True)
    return o, osd


def test_bf16_generator_step_segmented_against_rounding_oracle():
    """BASELINE.json configs 2-3 (bf16 storage inside the island): the pre-training generator step cut at the block
    boundaries like the fp32 one (tests/segmented.py), every segment in bf16 STORAGE against the fp64 oracle under the
    island's rounding model (oracle.restate.ISLAND_ROUNDING), teacher-forced on that oracle's activations and activation
    gradients.  Through ~40 layers the whole bf16 step is chaotic at bf16 resolution (the round-2 whole-step test can only
    ask for "as close to the rounding oracle as that is to exact arithmetic", i.e. ~0.2); one segment is not: forward within
    1e-2 of the largest activation, input and parameter gradients within 2e-2 of the tensor's largest entry (the per-block
    tests measure 2e-3 / 3-7e-3), the fp32 segments outside the island (stems, heads, Linears, fit2) at the fp32 bounds."""
    import segmented as SG
    from graph.model import Model
    from hipops import FlatParams
    from hipops import functional as HF
    from parity_util import RoundBf16, RoundBf16Forward
    B = 4
    gsd = W.make_state_dict(W.manifest_generator(), 0, "wc")
    zb = {k: v.double() for k, v in W.make_state_dict(W.manifest_z_discriminator(), 1, "wc").items()}
    zp = {k: v.double() for k, v in W.make_state_dict(W.manifest_z_discriminator(), 2, "wc").items()}
    note, pre_note, pre_phrase, position = W.make_inputs(B, seed=61)
    masks = [m.double() for m in _masks(B, 9)]
    ones = torch.ones(B, dtype=torch.float64)
    HF.set_compute_dtype("bf16")
    R.ISLAND_ROUNDING = (RoundBf16.apply, RoundBf16Forward.apply)
    try:
        gen = Model()
        gen.load_state_dict(gsd)
        gen = gen.to(dev).train()
        opt = FlatParams(list(gen.parameters()))
        g64 = {k: v.double().requires_grad_(True) for k, v in gsd.items()}

        def decode(zz, pf, taps):
            return R.decoder(g64, "decoder.", zz[:B], zz[B:], pf, position, True, masks, taps)

        def loss_of(g, zz, pf):
            loss = R.dloss(R.z_discriminator(zp, "", pf).view(-1), ones)
            loss = loss + R.dloss(R.z_discriminator(zb, "", zz[:B]).view(-1), ones) + R.dloss(R.z_discriminator(zb, "", zz[B:]).view(-1), ones)
            return loss + R.bar_loss(g, note.double(), True)

        enc_in = torch.cat((note, pre_note), 0).double()
        bound, gr, _ = SG.oracle_step(g64, enc_in, pre_phrase.double(), decode, loss_of)
        bound["__enc_in"], bound["__phrase_in"] = enc_in, pre_phrase.double()
        rep = SG.segmented_generator_check(gen, opt, gsd, bound, gr, position, masks, B=B, tol=1e-2)
    finally:
        R.ISLAND_ROUNDING = None
        HF.set_compute_dtype("f32")
    rules = {k: v for k, v in rep.items() if ":dx" not in k}
    bad = sorted((k, v) for k, v in rules.items() if v != "strict")
    REPORT.append("bf16 pre-training generator step, segmented vs the rounding oracle: %d parameter rows + %d boundary gradients; "
                  "not within 2e-2 in max norm (admitted on relative L2 <= 5e-2 or as <= 2 %% outliers): %s" % (len(rules), len(rep) - len(rules), bad))
    # every row passed one of: 2e-2 max-norm, <= 2 % outliers with L2 <= 5e-2, L2 <= 5e-2 (island rows only): i.e. every row of the
    # bf16 step is within 5e-2 of the rounding oracle once its segment boundaries are pinned -- the whole step's bound was ~0.3
    assert len(rules) >= 180 and len(bad) <= 0.25 * len(rules), bad


def _segmented_wae_generator_step(agent, sds, osd, batch, masks):
    """the generator step of the WAE iteration (agent/barGen_with_gan.py:426-452), cut at the block boundaries: the HIP
    generator holds the INITIAL weights (this runs before the agent's own iteration), the latent discriminators are the
    fp64 oracle's already-updated ones -- exactly what the whole-step generator gradient is taken against.  Every segment
    strict (tests/segmented.py); returns {parameter name: rule} for the whole-step comparison."""
    import segmented as SG
    note, pre_note, pre_phrase, position = batch
    B = note.shape[0]
    g64 = {k: v.double() for k, v in sds["generator"].items()}
    zb = {k: v.detach() for k, v in osd["z_discriminator_bar"].items()}
    zp = {k: v.detach() for k, v in osd["z_discriminator_phrase"].items()}
    m64 = [m.double() for m in masks]
    ones = torch.ones(B, dtype=torch.float64)

    def decode(zz, pf, taps):
        return R.decoder(g64, "decoder.", zz[:B], zz[B:], pf, position, True, m64, taps)

    def loss_of(gen, zz, pf):
        loss = R.dloss(R.z_discriminator(zp, "", pf).view(-1), ones)
        loss = loss + R.dloss(R.z_discriminator(zb, "", zz[:B]).view(-1), ones) + R.dloss(R.z_discriminator(zb, "", zz[B:]).view(-1), ones)
        return loss + R.bar_loss(gen, note.double(), False)

    enc_in = torch.cat((note, pre_note), 0).double()
    g64 = {k: v.requires_grad_(True) for k, v in g64.items()}
    bound, gr, loss = SG.oracle_step(g64, enc_in, pre_phrase.double(), decode, loss_of)
    bound["__enc_in"], bound["__phrase_in"] = enc_in, pre_phrase.double()
    rep = SG.segmented_generator_check(agent.generator, agent.opt_generator, sds["generator"], bound, gr, position, m64, B=B)
    rules = {k: v for k, v in rep.items() if ":dx" not in k}
    bad = sorted(k for k, v in rules.items() if v != "strict")
    REPORT.append("WAE generator step, segmented: %d parameter rows + %d boundary gradients; not strict: %s" % (
        len(rules), len(rep) - len(rules), bad))
    assert len(rules) >= 180, len(rules)
    assert len(bad) <= 0.05 * len(rules), bad           # (an in-segment arg-max tie may still fall the other way)
    return rules


def _segmented_gan_generator_step(agent, sds, osd, batch, noise, masks):
    """the generator step of the GAN iteration (agent/barGen_with_gan.py:507-529), cut at the block boundaries like the WAE
    one: the bar is decoded from the N(0, 1.5^2) latent, judged by the (already updated, frozen, train-mode) bar
    discriminator on the pair (pre_note, gen) and by the feature discriminator on encoder(gen > 0.3).  The bar encoder runs
    twice -- on pre_note and on the binarised fake bar, which depends on the decoder's output -- so the oracle step is
    assembled here and its two encoder passes are stacked along the batch for the segments (the encoder has no
    cross-sample op)."""
    import segmented as SG
    note, pre_note, pre_phrase, position = batch
    B = note.shape[0]
    g64 = {k: v.double().requires_grad_(True) for k, v in sds["generator"].items()}
    dd = {k: (v.detach().clone() if v.is_floating_point() else v.clone()) for k, v in osd["discriminator"].items()}
    ff = {k: v.detach() for k, v in osd["discriminator_feature"].items()}
    m64 = [m.double() for m in masks]
    ones = torch.ones(B, dtype=torch.float64)
    tp, ta, tb, td = {}, {}, {}, {}
    pf = R.phrase_model(g64, "phrase_encoder.", pre_phrase.double(), tp)
    pre_z = R.encoder(g64, "encoder.", pre_note.double(), ta)
    gen = R.decoder(g64, "decoder.", noise.double(), pre_z, pf, position, True, m64, td)
    fake = torch.gt(gen, 0.3).to(gen.dtype)
    gen_z = R.encoder(g64, "encoder.", fake, tb)
    loss = R.dloss(R.bar_discriminator(dd, "", torch.cat((pre_note.double(), gen), dim=2), train=True).view(-1), ones)
    loss = loss + R.dloss(R.bar_feature_discriminator(ff, "", gen_z).view(-1), ones)
    bound = {"pf": pf, "gen": gen, "pre_z": pre_z, "gen_z": gen_z}
    bound.update(tp); bound.update(td)
    for k in ta:
        bound["A:" + k] = ta[k]; bound["B:" + k] = tb[k]
    names = list(bound)
    gr = dict(zip(names, torch.autograd.grad(loss, [bound[n] for n in names], allow_unused=True)))
    zero = lambda t, g: torch.zeros_like(t) if g is None else g
    b2, g2 = {n: bound[n].detach() for n in names if n[:2] not in ("A:", "B:")}, {n: gr[n] for n in names if n[:2] not in ("A:", "B:")}
    for k in ta:
        b2[k] = torch.cat((bound["A:" + k].detach(), bound["B:" + k].detach()), 0)
        g2[k] = torch.cat((zero(bound["A:" + k], gr["A:" + k]), zero(bound["B:" + k], gr["B:" + k])), 0)
    b2["zz"] = torch.cat((pre_z.detach(), gen_z.detach()), 0)
    g2["zz"] = torch.cat((zero(pre_z, gr["pre_z"]), zero(gen_z, gr["gen_z"])), 0)
    b2["__enc_in"], b2["__phrase_in"] = torch.cat((pre_note.double(), fake.detach()), 0), pre_phrase.double()
    rep = SG.segmented_generator_check(agent.generator, agent.opt_generator, sds["generator"], b2, g2, position, m64, B=B,
                                       decoder_latent=noise.double())
    rules = {k: v for k, v in rep.items() if ":dx" not in k}
    bad = sorted(k for k, v in rules.items() if v != "strict")
    REPORT.append("GAN generator step, segmented: %d parameter rows + %d boundary gradients; not strict: %s" % (
        len(rules), len(rep) - len(rules), bad))
    assert len(rules) >= 180 and len(bad) <= 0.05 * len(rules), bad
    return rules


def test_train_wae_iteration_against_oracle(tmp_path, monkeypatch):
    agent, sds, grads = _agent(tmp_path, monkeypatch)
    lr = agent.config.learning_rate
    batch = W.make_inputs(4, seed=31)
    g = torch.Generator().manual_seed(8)
    noise = [torch.randn(4, 1152, generator=g) * agent.config.sigma for _ in range(2)]
    masks = _masks(4, 3)
    it = iter(noise)
    monkeypatch.setattr(agent, "prior", lambda rows, sigma: next(it).to(dev))
    agent.generator.decoder._drop_masks = [m.to(dev) for m in masks]
    agent.epoch = 1
    from metrics import AverageMeter
    meters = {k: AverageMeter() for k in ("generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase")}
    o, osd = _oracle("wae", sds, lr, batch, noise, masks, torch.float64)
    seg = _segmented_wae_generator_step(agent, sds, osd, batch, masks)
    out = agent.train_wae(*(t.to(dev) for t in batch), meters, 0)          # (epoch + curr_it) % 2 == 1: both halves run
    torch.cuda.synchronize()
    o32, _ = _oracle("wae", sds, lr, batch, noise, masks, torch.float32)
    o32 = _with_perturbed(o32, lambda seed: _oracle("wae", sds, lr, batch, noise, masks, torch.float32, perturb=seed)[0])
    check("train_wae phraseZ discriminator loss", meters["z_phrase"].val, o["phrase_loss"])
    check("train_wae barZ discriminator loss", meters["z_bar"].val, o["bar_loss"])
    check("train_wae generator loss", meters["generator"].val, o["generator_loss"])
    check("train_wae returned sample", out, o["gen"][:3], max(TOL, 2e-3))
    _compare_net("train_wae z_phrase", agent.z_discriminator_phrase, grads["z_discriminator_phrase"], o["grad_z_phrase"],
                 o32["grad_z_phrase"], osd["z_discriminator_phrase"], sds["z_discriminator_phrase"], lr)
    _compare_net("train_wae z_bar", agent.z_discriminator_bar, grads["z_discriminator_bar"], o["grad_z_bar"],
                 o32["grad_z_bar"], osd["z_discriminator_bar"], sds["z_discriminator_bar"], lr)
    _compare_net("train_wae generator", agent.generator, grads["generator"], o["grad_generator"], o32["grad_generator"], osd["generator"],
                 sds["generator"], lr, segment=seg)
    from parity_util import TALLY
    REPORT.append("train_wae tally after the generator rows: %s" % dict(TALLY))
    pop_margins("train_wae iteration vs fp64 oracle", 10)
    assert set(grads) == {"generator", "z_discriminator_bar", "z_discriminator_phrase"}      # nothing else stepped
    assert agent.opt_discriminator.step_count == 0 and agent.opt_generator.step_count == 1


def test_train_gan_iteration_against_oracle(tmp_path, monkeypatch):
    agent, sds, grads = _agent(tmp_path, monkeypatch)
    lr = agent.config.learning_rate
    batch = W.make_inputs(4, seed=32)
    noise = torch.randn(4, 1152, generator=torch.Generator().manual_seed(9)) * 1.5
    masks = _masks(4, 4)
    monkeypatch.setattr(agent, "prior", lambda rows, sigma: noise.to(dev))
    agent.generator.decoder._drop_masks = [m.to(dev) for m in masks]
    agent.epoch = 1
    from metrics import AverageMeter
    meters = {k: AverageMeter() for k in ("generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase")}
    o, osd = _oracle("gan", sds, lr, batch, noise, masks, torch.float64)
    seg = _segmented_gan_generator_step(agent, sds, osd, batch, noise, masks)
    out = agent.train_gan(*(t.to(dev) for t in batch), meters, 0)
    torch.cuda.synchronize()
    o32, _ = _oracle("gan", sds, lr, batch, noise, masks, torch.float32)
    o32 = _with_perturbed(o32, lambda seed: _oracle("gan", sds, lr, batch, noise, masks, torch.float32, perturb=seed)[0])
    check("train_gan bar discriminator loss", meters["discriminator"].val, o["note_loss"])
    check("train_gan feature discriminator loss", meters["discriminator_feature"].val, o["feature_loss"])
    check("train_gan generator loss", meters["generator"].val, o["generator_loss"])
    check("train_gan returned sample", out, o["gen"][:3], max(TOL, 2e-3))
    _compare_net("train_gan discriminator", agent.discriminator, grads["discriminator"], o["grad_discriminator"],
                 o32["grad_discriminator"], osd["discriminator"], sds["discriminator"], lr)
    _compare_net("train_gan discriminator_feature", agent.discriminator_feature, grads["discriminator_feature"],
                 o["grad_discriminator_feature"], o32["grad_discriminator_feature"], osd["discriminator_feature"], sds["discriminator_feature"], lr)
    _compare_net("train_gan generator", agent.generator, grads["generator"], o["grad_generator"], o32["grad_generator"], osd["generator"],
                 sds["generator"], lr, segment=seg)
    pop_margins("train_gan iteration vs fp64 oracle", 10)
    # BatchNorm running statistics after THREE train-mode passes (fake, real, generator step) and their counters
    hsd = agent.discriminator.state_dict()
    for k, v in osd["discriminator"].items():
        if "running_" in k:
            check("train_gan BatchNorm " + k, hsd[k], v)
        elif k.endswith("num_batches_tracked") and int(v) > 0:
            assert int(hsd[k]) == int(v) == 3, (k, int(hsd[k]), int(v))
    assert set(grads) == {"generator", "discriminator", "discriminator_feature"}


def _bn_stats(tag, module, osd_d, passes):
    """BatchNorm running statistics and counters of the bar discriminator after ``passes`` train-mode forwards (0: the
    module must still hold its initial statistics)"""
    hsd = module.state_dict()
    for k, v in osd_d.items():
        if "running_" in k:
            check("%s BatchNorm %s" % (tag, k), hsd[k], v)
        elif k.endswith("num_batches_tracked"):
            assert int(hsd[k]) == int(v), (k, int(hsd[k]), int(v))
            assert int(v) in (0, passes), (k, int(v))


def test_bargen_adversarial_iteration_against_oracle(tmp_path, monkeypatch):
    """row a17, agent/barGen.py:254-327 through the agent's own ``train_iteration``: the discriminator block (three
    networks step: BarDiscriminator on the BINARISED fake pair, both latent discriminators with real -> valid labels and
    sigma-1 priors) and the generator block with the D7 loss overwrite on ``opt_gen2`` -- every loss, every gradient, the
    Adam updates, and BatchNorm statistics after three train-mode passes."""
    agent, sds, grads = _agent(tmp_path, monkeypatch, which="barGen")
    lr = agent.config.learning_rate
    batch = W.make_inputs(4, seed=41)
    g = torch.Generator().manual_seed(18)
    noise = [torch.randn(4, 1152, generator=g), torch.randn(8, 1152, generator=g)]
    masks = _masks(4, 6)
    it = iter(noise)

    def prior(rows, sigma):
        t = next(it)
        assert sigma == 1.0 and t.shape[0] == rows       # agent/barGen.py:265,271: unscaled randn, B then 2B rows
        return t.to(dev)
    monkeypatch.setattr(agent, "prior", prior)
    agent.generator.decoder._drop_masks = [m.to(dev) for m in masks]
    agent.epoch = 1                                      # > pretraining_step_size (0): adversarial
    for m in (agent.generator, agent.discriminator, agent.z_discriminator_bar, agent.z_discriminator_phrase):
        m.train()                                        # agent/barGen.py:218-221
    from metrics import AverageMeter
    meters = tuple(AverageMeter() for _ in range(4))
    out = agent.train_iteration(*(t.to(dev) for t in batch), 0, 2, meters)        # (0 + 1) % 2 == 1: both blocks run
    torch.cuda.synchronize()
    o, osd = _oracle("bargen", sds, lr, batch, noise, masks, torch.float64)
    o32, _ = _oracle("bargen", sds, lr, batch, noise, masks, torch.float32)
    o32 = _with_perturbed(o32, lambda seed: _oracle("bargen", sds, lr, batch, noise, masks, torch.float32, perturb=seed)[0])
    avg_gen, avg_disc, avg_zbar, avg_zphrase = meters
    check("barGen bar discriminator loss", avg_disc.val, o["disc_loss"])
    check("barGen barZ discriminator loss", avg_zbar.val, o["bar_loss"])
    check("barGen phraseZ discriminator loss", avg_zphrase.val, o["phrase_loss"])
    check("barGen generator loss (D7: latent + bar terms only)", avg_gen.val, o["generator_loss"])
    check("barGen returned sample", out, o["gen"], max(TOL, 2e-3))
    _compare_net("barGen discriminator", agent.discriminator, grads["discriminator"], o["grad_discriminator"],
                 o32["grad_discriminator"], osd["discriminator"], sds["discriminator"], lr)
    _compare_net("barGen z_bar", agent.z_discriminator_bar, grads["z_discriminator_bar"], o["grad_z_bar"], o32["grad_z_bar"],
                 osd["z_discriminator_bar"], sds["z_discriminator_bar"], lr)
    _compare_net("barGen z_phrase", agent.z_discriminator_phrase, grads["z_discriminator_phrase"], o["grad_z_phrase"],
                 o32["grad_z_phrase"], osd["z_discriminator_phrase"], sds["z_discriminator_phrase"], lr)
    _compare_net("barGen generator", agent.generator, grads["generator"], o["grad_generator"], o32["grad_generator"], osd["generator"],
                 sds["generator"], lr)
    pop_margins("barGen adversarial iteration vs fp64 oracle", 10)
    # D7 + the threshold: nothing of the decoder receives a gradient in the adversarial generator block
    dec = [n for n, gr in o["grad_generator"].items() if n.startswith("decoder.")]
    assert dec and all(o["grad_generator"][n] is None for n in dec)
    assert all(float(grads["generator"][n].abs().max()) == 0.0 for n in dec)
    assert grads["generator_opt"] == "gen2" and agent.opt_gen2.step_count == 1 and agent.opt_gen1.step_count == 0
    _bn_stats("barGen", agent.discriminator, osd["discriminator"], 3)
    assert agent.opt_discriminator.step_count == agent.opt_Zdiscriminator_bar.step_count == agent.opt_Zdiscriminator_phrase.step_count == 1


def test_bargen_with_gan2_iteration_against_oracle(tmp_path, monkeypatch):
    """row a17, agent/barGen_with_gan2.py:345-404 + :468-519 through the agent's own ``train_discriminator`` and
    ``train_add_gan``: all four discriminators step every iteration (BarDiscriminator in eval() mode on the binarised fake
    pair), then the generator on reconstruction + latent terms + 0.05 x (bar + feature) terms of a bar decoded from
    N(0, 1.5^2) noise."""
    agent, sds, grads = _agent(tmp_path, monkeypatch, which="barGen_with_gan2")
    lr = agent.config.learning_rate
    batch = W.make_inputs(4, seed=43)
    g = torch.Generator().manual_seed(28)
    sigma = agent.config.sigma
    noise = [torch.randn(4, 1152, generator=g) * sigma, torch.randn(4, 1152, generator=g) * sigma, torch.randn(4, 1152, generator=g) * 1.5]
    want_sigma = [sigma, sigma, 1.5]
    masks = _masks(4, 7)
    k = [0]

    def prior(rows, sg):
        assert rows == 4 and sg == want_sigma[k[0]], (rows, sg, k[0])
        k[0] += 1
        return noise[k[0] - 1].to(dev)
    monkeypatch.setattr(agent, "prior", prior)
    agent.generator.decoder._drop_masks = [m.to(dev) for m in masks]
    agent.epoch = 1
    # the modes the previous iteration's train_add_gan left (agent/barGen_with_gan2.py:469-474)
    agent.modes(train=("generator", "z_discriminator_bar", "z_discriminator_phrase"), evaluate=("discriminator", "discriminator_feature"))
    from metrics import AverageMeter
    meters = {k_: AverageMeter() for k_ in ("generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase")}
    dbatch = tuple(t.to(dev) for t in batch)
    agent.train_discriminator(*dbatch, meters)
    out = agent.train_add_gan(*dbatch, meters)
    torch.cuda.synchronize()
    assert k[0] == 3
    o, osd = _oracle("gan2", sds, lr, batch, noise, masks, torch.float64)
    o32, _ = _oracle("gan2", sds, lr, batch, noise, masks, torch.float32)
    o32 = _with_perturbed(o32, lambda seed: _oracle("gan2", sds, lr, batch, noise, masks, torch.float32, perturb=seed)[0])
    check("gan2 phraseZ discriminator loss", meters["z_phrase"].val, o["phrase_loss"])
    check("gan2 barZ discriminator loss", meters["z_bar"].val, o["bar_loss"])
    check("gan2 bar discriminator loss", meters["discriminator"].val, o["note_loss"])
    check("gan2 feature discriminator loss", meters["discriminator_feature"].val, o["feature_loss"])
    check("gan2 generator loss", meters["generator"].val, o["generator_loss"])
    check("gan2 returned sample", out, o["gen"][:3], max(TOL, 2e-3))
    for net, key in (("z_discriminator_phrase", "grad_z_phrase"), ("z_discriminator_bar", "grad_z_bar"),
                     ("discriminator", "grad_discriminator"), ("discriminator_feature", "grad_discriminator_feature"),
                     ("generator", "grad_generator")):
        _compare_net("gan2 " + net, getattr(agent, net), grads[net], o[key], o32[key], osd[net], sds[net], lr)
    pop_margins("barGen_with_gan2 iteration vs fp64 oracle", 10)
    assert set(grads) == {"generator", "discriminator", "discriminator_feature", "z_discriminator_bar", "z_discriminator_phrase"}
    assert all(n.opt.step_count == 1 for n in agent.nets.values())
    _bn_stats("gan2", agent.discriminator, osd["discriminator"], 0)      # eval mode: the statistics never move


def test_train_gan_iteration_bf16_storage(tmp_path, monkeypatch):
    """BASELINE.json configs[3] (barGen_with_gan, bf16): the same GAN iteration with ``config.compute_dtype = 'bf16'`` --
    the generator's island in bf16 storage, the discriminators (outside the island) fp32 -- against the fp64 oracle WITH
    the island's rounding model.  The discriminator steps see only the generator's forward: their losses must match to
    1 % and their gradients in direction; the generator's own gradient is judged like the 32-bar step
    (test_bf16_storage_step_against_bf16_rounding_oracle): as close to the rounding oracle as that is to exact arithmetic."""
    from hipops import functional as HF
    from parity_util import RoundBf16, RoundBf16Forward
    agent, sds, grads = _agent(tmp_path, monkeypatch, "bf16")
    try:
        assert HF.get_compute_dtype() == "bf16"
        lr = agent.config.learning_rate
        batch = W.make_inputs(4, seed=32)
        noise = torch.randn(4, 1152, generator=torch.Generator().manual_seed(9)) * 1.5
        masks = _masks(4, 4)
        monkeypatch.setattr(agent, "prior", lambda rows, sigma: noise.to(dev))
        agent.generator.decoder._drop_masks = [m.to(dev) for m in masks]
        agent.epoch = 1
        from metrics import AverageMeter
        meters = {k: AverageMeter() for k in ("generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase")}
        agent.train_gan(*(t.to(dev) for t in batch), meters, 0)
        torch.cuda.synchronize()
    finally:
        HF.set_compute_dtype("f32")
    o_x, _ = _oracle("gan", sds, lr, batch, noise, masks, torch.float64)
    R.ISLAND_ROUNDING = (RoundBf16.apply, RoundBf16Forward.apply)
    try:
        o_r, _ = _oracle("gan", sds, lr, batch, noise, masks, torch.float64)
    finally:
        R.ISLAND_ROUNDING = None
    for key, meter in (("note_loss", "discriminator"), ("feature_loss", "discriminator_feature"), ("generator_loss", "generator")):
        got, want, exact = float(meters[meter].val), float(o_r[key]), float(o_x[key])
        REPORT.append("bf16 train_gan %-16s hip %.6f  rounding oracle %.6f  exact %.6f" % (key, got, want, exact))
        # the feature discriminator reads the BINARISED fake pair: one bf16 rounding that falls the other way flips notes, and the
        # same build on the same box gives 1.378 - 1.405 from run to run (tools/r3_run25.sh and the full-suite runs: -1.9 %, -1.6 %,
        # -0.4 %, -0.2 %, 0.0 % of the oracle's 1.4048); the bound is twice the worst of those.  The two other losses sit within
        # 1e-3 of the oracle in every run
        rel = 4e-2 if key == "feature_loss" else 1e-2
        assert abs(got - want) <= max(2 * abs(want - exact), rel * abs(exact)), (key, got, want, exact)

    def flat(d, names):
        return torch.cat([d[n].detach().double().cpu().reshape(-1) for n in names])
    for net, okey in (("discriminator", "grad_discriminator"), ("discriminator_feature", "grad_discriminator_feature"),
                      ("generator", "grad_generator")):
        names = [n for n, g in o_x[okey].items() if g is not None and n in grads[net]]
        fh, fr, fx = flat(grads[net], names), flat(o_r[okey], names), flat(o_x[okey], names)
        e_hip = float((fh - fr).norm() / fr.norm()); e_model = float((fr - fx).norm() / fx.norm())
        cos = float((fh * fr).sum() / (fh.norm() * fr.norm()))
        REPORT.append("bf16 train_gan d%-22s |hip - rounding oracle| %.3e   |rounding oracle - exact| %.3e   cos %.4f" % (net, e_hip, e_model, cos))
        # (the fake bar is binarised at 0.3 before the discriminators see it: bf16 rounding flips cells, so even the
        # discriminators' gradients move by 20 - 40 % between the rounding oracle and exact arithmetic)
        assert e_hip <= 1.5 * max(e_model, 2e-2), (net, e_hip, e_model)
        assert cos >= 1.0 - 1.5 * max(e_model, 2e-2) ** 2, (net, cos, e_model)
    assert agent.opt_discriminator.step_count == 1 and agent.opt_generator.step_count == 1

