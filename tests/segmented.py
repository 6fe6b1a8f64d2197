"""Segmented backward of one generator step: every trunk segment of the HIP generator against the fp64 oracle, TEACHER-FORCED
at the segment boundaries.

Why: in a whole 4-bar step roughly half of the generator's gradient rows are chaotic at fp32 resolution -- a single ReLU /
arg-max decision near the top of a trunk moves every row below it by a common 0.3 - 1.2 %, and the reference's own fp32
arithmetic does the same (parity_util.check_grad: "uninformative").  Cutting the backward pass at the block boundaries
removes the propagation: each segment gets the ORACLE's fp64 input activation and the ORACLE's fp64 gradient of its output
(both cast to fp32), runs forward + backward on the HIP path alone, and is held to the strict bound -- output, input
gradient and every parameter gradient of the segment (the oracle's parameter gradients of the isolated segment ARE its
whole-step ones, because the boundary values are the whole step's).  The segments tile the generator: both encoder trunks
(stems, 8 blocks, pool + Linear), the decoder (head, stems, fit1 stage, 4 blocks, fit2), i.e. all 217 trained tensors.
"""
import torch

from oracle import restate as R
from parity_util import REPORT, TOL, check, check_grad


def _leaf(sd, prefix):
    return {k[len(prefix):]: v.detach().clone().double().requires_grad_(True) for k, v in sd.items() if k.startswith(prefix)}


def oracle_step(gsd64, enc_inputs, phrase, decode, loss_of):
    """fp64 generator step with taps.  ``enc_inputs``: the bars the (shared) bar encoder sees, stacked [k*B,1,96,60];
    ``decode(zz, pf, taps)`` -> gen (calls R.decoder with the taps dict); ``loss_of(gen, zz, pf)`` -> scalar.
    Returns the boundary tensors and their gradients."""
    tp, te, td = {}, {}, {}
    pf = R.phrase_model(gsd64, "phrase_encoder.", phrase, tp)
    zz = R.encoder(gsd64, "encoder.", enc_inputs, te)
    gen = decode(zz, pf, td)
    loss = loss_of(gen, zz, pf)
    bound = {"pf": pf, "zz": zz, "gen": gen}
    bound.update(tp); bound.update(te); bound.update(td)
    names = list(bound)
    grads = torch.autograd.grad(loss, [bound[n] for n in names], allow_unused=True, retain_graph=False)
    return {n: bound[n].detach() for n in names}, dict(zip(names, grads)), float(loss)


def _run_segment(tag, hip_fn, oracle_fn, osd, hip_params, opt, inputs, in_grads, dy, out_ref, tol, l2_ok=None):
    """one segment.  inputs: list of fp64 tensors (int tensors pass through); in_grads: the oracle's gradients of those
    inputs (None: not checked); dy: the oracle's gradient of the segment output; osd: fresh fp64 leaves of its parameters.
    Returns ({parameter name within the segment: rule}, {input index: rule})"""
    dev = next(iter(hip_params.values())).device
    # oracle, isolated on the same boundary values
    xin = [t.detach().clone().requires_grad_(True) if t.is_floating_point() else t for t in inputs]
    yo = oracle_fn(osd, *xin)
    yo.backward(dy)
    # HIP
    opt.zero_grad()
    xd = [t.float().to(dev).requires_grad_(g is not None) if t.is_floating_point() else t.to(dev) for t, g in zip(inputs, in_grads)]
    y = hip_fn(*xd)
    check("%s fwd (teacher-forced input)" % tag, y, out_ref, tol)
    y.backward(dy.float().to(dev))
    torch.cuda.synchronize()
    dx_rules, p_rules = {}, {}
    for i, (t, g) in enumerate(zip(xd, in_grads)):
        if g is not None:
            dx_rules[i] = check_grad("%s dx%d" % (tag, i), t.grad, g, 2 * tol, l2_ok=l2_ok)
    gs = max([v.grad.abs().max().item() for v in osd.values() if v.grad is not None] + [1e-300])
    # analytically-zero gradients (the bias of a conv in front of an InstanceNorm) are pure rounding noise: 1e-6 of the
    # segment's largest gradient entry in fp32, 5e-3 of it where the segment's tensors are STORED in bf16 (l2_ok is only
    # given for those segments) -- such rows carry no information and are skipped, smaller entries of other rows get the
    # same absolute allowance
    floor = (5e-3 if l2_ok is not None else 1e-6) * gs
    for n, p in hip_params.items():
        og = osd[n].grad
        if og is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, (tag, n)      # D5: never used
            continue
        if og.abs().max().item() <= floor:
            continue
        p_rules[n] = check_grad("%s d%s" % (tag, n), p.grad, og, 2 * tol, atol=floor, l2_ok=l2_ok)
    return p_rules, dx_rules


def segmented_generator_check(gen, opt, gsd, bound, grads, position, masks, enc_pool=(3, 2), B=None, decoder_latent=None, tol=TOL):
    """``gen``: the HIP generator (graph.model.Model / graph.model_with_gan.Model) whose parameters hold ``gsd`` and live in
    the flat optimizer ``opt``; ``bound`` / ``grads``: from oracle_step.  ``decoder_latent``: None -> the decoder's z / pre_z
    are the two halves of zz (training forward); a tensor -> z is that latent and pre_z = zz[:B] (the GAN generator step).
    Returns {generator parameter name: rule} -- "strict" wherever the segment passed the strict bound."""
    import torch.nn.functional as F
    from hipops import functional as HF
    report = {}
    cl = HF.to_channels_last

    def params_of(module):
        return dict(module.named_parameters())

    def seg(name, module, hip_fn, oracle_fn, prefix, inputs, in_grads, dy, out_ref, island=False):
        # ``tol`` is the bound of the segments INSIDE the channels-last island (it differs from TOL only in bf16 storage);
        # the stems, heads, Linears and fit2 compute in fp32 in every mode
        osd = _leaf(gsd, prefix)
        # bf16 storage (tol > TOL): a row of an island segment that misses the max-norm bound may still pass on its relative L2
        # error (<= 5e-2): a small aggregate such as a CBAM gate-MLP weight sums bf16-rounded terms over whole maps
        p_rules, dx_rules = _run_segment(name, hip_fn, oracle_fn, osd, params_of(module), opt, inputs, in_grads, dy, out_ref,
                                         tol if island else TOL, l2_ok=5e-2 if (island and tol > TOL) else None)
        for n, v in p_rules.items():
            report[prefix + n] = v
        for i, v in dx_rules.items():
            report["%s:dx%d" % (name, i)] = v

    for trunk, p, x_in, pool in ((gen.encoder, "encoder.", bound["__enc_in"], enc_pool),
                                 (gen.phrase_encoder.phrase_encoder, "phrase_encoder.phrase_encoder.", bound["__phrase_in"], (12, 2))):
        class Stems(torch.nn.Module):
            def __init__(self, t):
                super().__init__()
                self.pitch_time, self.time_pitch = t.pitch_time, t.time_pitch
        cat_ref = torch.cat((bound[p + "pitch_time"], bound[p + "time_pitch"]), dim=1)
        cat_grad = torch.cat((grads[p + "pitch_time"], grads[p + "time_pitch"]), dim=1)
        seg(p + "stems", Stems(trunk), lambda x, trunk=trunk: trunk.stem_cat(x, cast=False),
            lambda sd, x: torch.cat((R.enc_pitch_time(sd, "pitch_time.", x), R.enc_time_pitch(sd, "time_pitch.", x)), dim=1),
            p, [x_in], [None], cat_grad, cat_ref)
        prev, prev_g = cat_ref, cat_grad
        for i, blk in enumerate(trunk.layers):
            ofn = R.residual_module if i % 2 == 0 else R.pooling_module
            out_ref, out_g = bound[p + "layers.%d" % i], grads[p + "layers.%d" % i]
            seg(p + "layers.%d" % i, blk, (lambda x, blk=blk: HF.to_nchw(blk(cl(x)))) if trunk.channels_last else blk,
                lambda sd, x, ofn=ofn: ofn(sd, "", x), p + "layers.%d." % i, [prev], [prev_g], out_g, out_ref, island=True)
            prev, prev_g = out_ref, out_g
        key = "zz" if p == "encoder." else "pf"
        seg(p + "linear", trunk.linear, lambda x, t=trunk: t.linear(HF.global_avg_pool(x)),
            lambda sd, x, pool=pool: F.linear(F.avg_pool2d(x, pool).view(-1, 1024), sd["weight"], sd.get("bias")),
            p + "linear.", [prev], [prev_g], grads[key], bound[key])
    # ---- decoder
    dec, p = gen.decoder, "decoder."
    zz, pf = bound["zz"], bound["pf"]
    if decoder_latent is None:
        z, pre_z = zz[:B], zz[B:2 * B]
    else:
        z, pre_z = decoder_latent, zz[:B]

    class Head(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.position_embedding, self.phrase_linear, self.bar_linear = d.position_embedding, d.phrase_linear, d.bar_linear
    dec._drop_masks = [m.float().to(next(dec.parameters()).device) for m in masks]
    seg(p + "head", Head(dec), lambda a, b, c, pos: dec.head(a, b, c, pos),
        lambda sd, a, b, c, pos: R.decoder_head(sd, "", a, b, c, pos, True, masks), p,
        [z, pre_z, pf, position], [None, None, None, None], grads[p + "head"], bound[p + "head"])
    # (the latents' whole-step gradients also collect the latent discriminators' loss terms: not this segment's dx)

    class DStems(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.pitch, self.time = d.pitch, d.time
    cat_ref = torch.cat((bound[p + "pitch"], bound[p + "time"]), dim=1)
    cat_grad = torch.cat((grads[p + "pitch"], grads[p + "time"]), dim=1)
    seg(p + "stems", DStems(dec), dec.stems,
        lambda sd, x: torch.cat((R.dec_pitch_time(sd, "pitch.", x), R.dec_time_pitch(sd, "time.", x)), dim=1),
        p, [bound[p + "head"]], [grads[p + "head"]], cat_grad, cat_ref)

    class Fit(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.fit1, self.bn, self.cbam = d.fit1, d.bn, d.cbam
    wrap = (lambda f: (lambda x: HF.to_nchw(f(cl(x))))) if dec.channels_last else (lambda f: f)
    seg(p + "fit1", Fit(dec), wrap(dec.fit_stage), lambda sd, x: R.decoder_fit1(sd, "", x), p, [cat_ref], [cat_grad],
        grads[p + "fit1"], bound[p + "fit1"], island=True)
    prev, prev_g = bound[p + "fit1"], grads[p + "fit1"]
    for i, blk in enumerate(dec.layers):
        ofn = R.deconv_pitch_padding if i < 2 else R.deconv_module
        out_ref, out_g = bound[p + "layers.%d" % i], grads[p + "layers.%d" % i]
        seg(p + "layers.%d" % i, blk, wrap(blk), lambda sd, x, ofn=ofn: ofn(sd, "", x), p + "layers.%d." % i, [prev], [prev_g], out_g, out_ref,
            island=True)
        prev, prev_g = out_ref, out_g
    seg(p + "fit2", dec.fit2, (lambda x: HF.conv2d_to1_cl(cl(x), dec.fit2.weight, HF.ACT_SIGMOID)) if dec.channels_last else (lambda x: dec.fit2(x, act=HF.ACT_SIGMOID)),
        lambda sd, x: torch.sigmoid(F.conv2d(x, sd["weight"])),
        p + "fit2.", [prev], [prev_g], grads["gen"], bound["gen"], island=dec.channels_last)      # reads (bf16 mode: bf16) island storage
    opt.zero_grad()
    rules = {k: v for k, v in report.items() if ":dx" not in k}
    n_strict = sum(1 for v in rules.values() if v == "strict")
    REPORT.append("segmented backward: %d parameter rows, %d strict, others: %s" % (
        len(rules), n_strict, sorted((k, v) for k, v in rules.items() if v != "strict")))
    return report
