"""TEST INFRASTRUCTURE ONLY -- pins the oracle to the reference and writes tests/golden/.

Run in the build container (the only place /root/reference exists):

    python oracle/make_golden.py

What it does
  1. imports the reference's importable modules (graph.encoder.Encoder,
     graph.phrase_encoder.PhraseModel, graph.decoder.Decoder, the discriminators,
     graph.loss.bar_loss.DLoss) from /root/reference,
  2. checks oracle/weights.py manifests against their state_dict() names + shapes,
  3. loads make_state_dict(...) weights into them, runs seeded inputs, and asserts the
     functional restatement in oracle/restate.py reproduces every output BIT FOR BIT,
  4. writes small fixtures (inputs are regenerated from seeds, never stored).

Nothing from the reference is copied: fixtures hold numbers only.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import restate as R          # noqa: E402
from oracle import weights as W          # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
B = 4


def summary(t):
    """(sum, abs-sum, 64 strided samples) of a tensor, float64 accumulators."""
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, 64).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def check_manifest(mod, manifest, what):
    ref = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
    mine = [(n, tuple(s)) for n, s, _ in manifest]
    assert sorted(ref) == sorted(mine), (what, set(ref) ^ set(mine))
    if ref != mine:
        print("  note: %s manifest order differs from state_dict order (names/shapes equal)" % what)
    return ref


def beq(a, b, what):
    assert a.shape == b.shape and torch.equal(a, b), "oracle != reference for %s (max abs %g)" % (
        what, (a - b).abs().max().item())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    from graph.encoder import Encoder
    from graph.phrase_encoder import PhraseModel
    from graph.decoder import Decoder
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.bar_discriminator_with_feature import BarFeatureDiscriminator
    from graph.bar_discriminator import BarDiscriminator
    from graph.loss.bar_loss import DLoss
    import graph.cbam as ref_cbam
    import graph.encodingBlock as ref_eb
    import graph.decoder as ref_dec

    enc, phr, dec = Encoder(list(R.ENC_LAYERS)), PhraseModel(list(R.ENC_LAYERS)), Decoder(list(R.DEC_LAYERS))
    zb, zp, fd, bd = BarZDiscriminator(), PhraseZDiscriminator(), BarFeatureDiscriminator(), BarDiscriminator()
    manifests = {
        "encoder": (enc, W.manifest_encoder()), "phrase_encoder": (phr, W.manifest_phrase_model()),
        "decoder": (dec, W.manifest_decoder()), "z_discriminator_bar": (zb, W.manifest_z_discriminator()),
        "z_discriminator_phrase": (zp, W.manifest_z_discriminator()),
        "discriminator_feature": (fd, W.manifest_bar_feature_discriminator()),
        "discriminator": (bd, W.manifest_bar_discriminator()),
    }
    mjson = {}
    for k, (mod, man) in manifests.items():
        mjson[k] = [[n, list(s)] for n, s in check_manifest(mod, man, k)]
    # D4 check: which classes weights_init touches, by statistics of a fresh module
    fresh = Encoder(list(R.ENC_LAYERS))
    assert abs(fresh.layers[0].conv1.weight.mean().item() + 1.0) < 0.05      # Conv2d ~ N(-1,1)
    assert abs(fresh.linear.weight.mean().item() + 1.0) < 0.05              # Linear ~ N(-1,1)
    assert torch.equal(fresh.layers[0].bn.weight, torch.ones(64))            # InstanceNorm untouched
    assert fresh.linear.bias.abs().max().item() <= 1.0 / 32 + 1e-6           # bias keeps default U(+-1/sqrt(1024))
    fd_ = Decoder(list(R.DEC_LAYERS))
    assert fd_.layers[3].deConv1.weight.abs().max().item() < 0.1             # ConvTranspose2d untouched
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(mjson, f)

    note, pre_note, phrase, position = W.make_inputs(B, seed=1234)
    for mode in ("d4", "wc"):
        print("mode", mode)
        fx = {}
        sds = {k: W.make_state_dict(man, seed=0, mode=mode) for k, (_, man) in manifests.items()}
        # canonical generator weights are keyed by their full generator names
        sds["encoder"], sds["decoder"], sds["phrase_encoder"] = W.split_generator(
            W.make_state_dict(W.manifest_generator(), seed=0, mode=mode))
        for k, (mod, _) in manifests.items():
            mod.load_state_dict(sds[k])
            mod.eval()
        bd.train()   # BatchNorm batch statistics, like the agents
        with torch.no_grad():
            # ---- reference forward (eval: dropout off)
            r_pf = phr(phrase)
            r_z, r_pz = enc(note), enc(pre_note)
            r_gen = dec(r_z, r_pz, r_pf, position)
            # ---- oracle forward
            taps, taps_p, taps_d = {}, {}, {}
            o_pf = R.phrase_model(sds["phrase_encoder"], "", phrase, taps_p)
            o_z = R.encoder(sds["encoder"], "", note, taps)
            o_pz = R.encoder(sds["encoder"], "", pre_note)
            o_gen, o_logit = R.decoder(sds["decoder"], "", o_z, o_pz, o_pf, position, taps=taps_d, return_logits=True)
            beq(o_pf, r_pf, "phrase_feature"); beq(o_z, r_z, "z"); beq(o_pz, r_pz, "pre_z"); beq(o_gen, r_gen, "gen")
            # ---- block-level pins (reference leaf blocks vs restatement on the oracle's own taps)
            blk = ref_eb.ResidualModule(64); blk.load_state_dict({k[len("layers.0."):]: v for k, v in sds["encoder"].items() if k.startswith("layers.0.")})
            x0 = torch.cat((taps["pitch_time"], taps["time_pitch"]), 1)
            beq(R.residual_module(sds["encoder"], "layers.0.", x0), blk(x0.clone()), "ResidualModule(64)")
            blk = ref_eb.PoolingModule(64, 128); blk.load_state_dict({k[len("layers.1."):]: v for k, v in sds["encoder"].items() if k.startswith("layers.1.")})
            beq(R.pooling_module(sds["encoder"], "layers.1.", taps["layers.0"]), blk(taps["layers.0"].clone()), "PoolingModule")
            cb = ref_cbam.CBAM(64); cb.load_state_dict({k[len("layers.0.cbam."):]: v for k, v in sds["encoder"].items() if k.startswith("layers.0.cbam.")})
            beq(R.cbam(sds["encoder"], "layers.0.cbam.", x0), cb(x0), "CBAM(64)")
            blk = ref_dec.DeConvPitchPadding(1024, 512); blk.load_state_dict({k[len("layers.0."):]: v for k, v in sds["decoder"].items() if k.startswith("layers.0.")})
            beq(R.deconv_pitch_padding(sds["decoder"], "layers.0.", taps_d["fit1"]), blk(taps_d["fit1"].clone()), "DeConvPitchPadding")
            blk = ref_dec.DeConvModule(256, 128); blk.load_state_dict({k[len("layers.2."):]: v for k, v in sds["decoder"].items() if k.startswith("layers.2.")})
            beq(R.deconv_module(sds["decoder"], "layers.2.", taps_d["layers.1"]), blk(taps_d["layers.1"].clone()), "DeConvModule")
            # ---- discriminators
            r_dzb, r_dzp, r_dfd = zb(r_z), zp(r_pf), fd(r_z)
            pair = torch.cat((pre_note, note), dim=2)
            r_dbd = bd(pair)
            bsd = {k: v.clone() for k, v in sds["discriminator"].items()}
            beq(R.z_discriminator(sds["z_discriminator_bar"], "", o_z), r_dzb, "BarZDiscriminator")
            beq(R.z_discriminator(sds["z_discriminator_phrase"], "", o_pf), r_dzp, "PhraseZDiscriminator")
            beq(R.bar_feature_discriminator(sds["discriminator_feature"], "", o_z), r_dfd, "BarFeatureDiscriminator")
            beq(R.bar_discriminator(bsd, "", pair, train=True), r_dbd, "BarDiscriminator")
            for k, v in bd.state_dict().items():
                assert torch.equal(bsd[k], v), ("BarDiscriminator running stats", k)
            tgt = torch.ones(B)
            r_dl = DLoss()(r_dzb.view(-1), tgt)
            beq(R.dloss(r_dzb.view(-1), tgt), r_dl, "DLoss")
        fx.update(z=o_z.numpy(), pre_z=o_pz.numpy(), phrase_feature=o_pf.numpy(), gen=o_gen.numpy(),
                  logits=o_logit.numpy(), d_zbar=r_dzb.numpy(), d_zphrase=r_dzp.numpy(), d_feature=r_dfd.numpy(),
                  d_bar=r_dbd.numpy(), dloss=np.array(r_dl.item()),
                  bd_running=np.concatenate([v.numpy().ravel() for k, v in bd.state_dict().items() if "running" in k]))
        for pref, tp in (("encoder.", taps), ("phrase_encoder.", taps_p), ("decoder.", taps_d)):
            for k, t in tp.items():
                fx["tap/" + pref + k] = summary(t)

        # ---- gradients: reference modules' autograd vs the oracle's, same composed loss
        for m in (enc, phr, dec):
            m.zero_grad()
        rp = phr(phrase); rz = enc(note); rpz = enc(pre_note); rg = dec(rz, rpz, rp, position)
        ones = torch.ones(B)
        loss_r = DLoss()(zp(rp).view(-1), ones) + DLoss()(zb(rz).view(-1), ones) + DLoss()(zb(rpz).view(-1), ones) \
            + R.bar_loss(rg, note, True)
        gparams = [p for m in (enc, dec, phr) for p in m.parameters()]
        gr = torch.autograd.grad(loss_r, gparams, allow_unused=True)
        gsd = {}
        for pref, k in (("encoder.", "encoder"), ("decoder.", "decoder"), ("phrase_encoder.", "phrase_encoder")):
            for n, v in sds[k].items():
                gsd[pref + n] = v.clone().requires_grad_(True)
        loss_o, _ = R.pretrain_step_loss(gsd, sds["z_discriminator_bar"], sds["z_discriminator_phrase"],
                                         note, pre_note, phrase, position, is_pretraining=True)
        names = [pref + n for pref, m in (("encoder.", enc), ("decoder.", dec), ("phrase_encoder.", phr))
                 for n, _ in m.named_parameters()]
        go = torch.autograd.grad(loss_o, [gsd[n] for n in names], allow_unused=True)
        beq(loss_o.detach(), loss_r.detach(), "pretrain loss")
        gnorm, unused = {}, []
        for n, a, b in zip(names, gr, go):
            assert (a is None) == (b is None), n
            if a is None:
                unused.append(n)
                continue
            beq(b, a, "grad " + n)
            gnorm[n] = [a.double().norm().item(), a.double().sum().item()]
        print("  loss %.6f, %d grads bit-equal, grad-less params: %s" % (loss_r.item(), len(gnorm), unused))
        fx["loss_pretrain"] = np.array(loss_r.item())
        fx["loss_smoothed"] = np.array(R.bar_loss(rg.detach(), note, False).item())
        with open(os.path.join(OUT, "gradnorm_%s.json" % mode), "w") as f:
            json.dump({"grad": gnorm, "unused": unused}, f)

        # ---- fp64 run of the same functions (tolerance rule of SURVEY section 7)
        with torch.no_grad():
            d = lambda s: {k: (v.double() if v.is_floating_point() else v) for k, v in s.items()}
            e64, p64, c64 = d(sds["encoder"]), d(sds["phrase_encoder"]), d(sds["decoder"])
            pf64 = R.phrase_model(p64, "", phrase.double())
            z64, pz64 = R.encoder(e64, "", note.double()), R.encoder(e64, "", pre_note.double())
            g64, l64 = R.decoder(c64, "", z64, pz64, pf64, position, return_logits=True)
            # decoder alone from the fp32 latents: isolates the decoder's own rounding
            g64d, l64d = R.decoder(c64, "", o_z.double(), o_pz.double(), o_pf.double(), position, return_logits=True)
        fx.update(z64=z64.numpy(), pre_z64=pz64.numpy(), phrase_feature64=pf64.numpy(), gen64=g64.numpy(),
                  logits64=l64.numpy(), gen64_dec=g64d.numpy(), logits64_dec=l64d.numpy())
        rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
        print("  torch fp32 vs fp64: z %.2e  phrase %.2e  gen(e2e) %.2e  gen(dec only) %.2e  logits(dec only) %.2e; "
              "binarised mismatches %d" % (rel(o_z, z64), rel(o_pf, pf64), rel(o_gen, g64), rel(o_gen, g64d),
                                           rel(o_logit, l64d), int(((o_gen > 0.3) != (g64d > 0.3)).sum())))
        np.savez_compressed(os.path.join(OUT, "generator_%s.npz" % mode), **fx)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
