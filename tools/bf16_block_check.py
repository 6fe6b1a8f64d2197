"""diagnosis: each channels-last block in bf16 STORAGE against the fp64 oracle with the island rounding model
(oracle.restate.ISLAND_ROUNDING) and against exact fp64 -- forward, dx and the largest parameter-gradient error."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "musicgeneration_vae-torch_amd")); sys.path.insert(0, os.path.join(R_, "tests"))
import torch
import __graft_entry__ as g; g.build()
import graph.encodingBlock as EB
import graph.decoder as DD
from hipops import FlatParams
from hipops import functional as HF
from oracle import restate as R, weights as W
from parity_util import RoundBf16, RoundBf16Forward
dev = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "wc"
gsd = W.make_state_dict(W.manifest_generator(), 0, mode)
torch.manual_seed(3)
cases = [("residual64", lambda: EB.ResidualModule(64, True), "encoder.layers.0.", R.residual_module, (3, 64, 48, 30)),
         ("pooling64", lambda: EB.PoolingModule(64, 128, True), "encoder.layers.1.", R.pooling_module, (3, 64, 48, 30)),
         ("residual128", lambda: EB.ResidualModule(128, True), "encoder.layers.2.", R.residual_module, (2, 128, 24, 15)),
         ("residual512", lambda: EB.ResidualModule(512, True), "encoder.layers.6.", R.residual_module, (3, 512, 6, 4)),
         ("pooling512", lambda: EB.PoolingModule(512, 1024, True), "encoder.layers.7.", R.pooling_module, (3, 512, 6, 4)),
         ("deconv_pp1024", lambda: DD.DeConvPitchPadding(1024, 512, True), "decoder.layers.0.", R.deconv_pitch_padding, (3, 1024, 6, 3)),
         ("deconv_pp512", lambda: DD.DeConvPitchPadding(512, 256, True), "decoder.layers.1.", R.deconv_pitch_padding, (3, 512, 12, 7)),
         ("deconv256", lambda: DD.DeConvModule(256, 128, True), "decoder.layers.2.", R.deconv_module, (2, 256, 24, 15)),
         ("deconv128", lambda: DD.DeConvModule(128, 64, True), "decoder.layers.3.", R.deconv_module, (2, 128, 48, 30))]
rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-300))
HF.set_compute_dtype("bf16")
print("%-16s %-34s %-34s %s" % ("block", "fwd: hip-round / round-exact / hip-exact", "dx: same three", "worst dparam: hip-round / round-exact"))
for tag, mk, prefix, ofn, shape in cases:
    mod = mk()
    sub = {k[len(prefix):]: v for k, v in gsd.items() if k.startswith(prefix)}
    mod.load_state_dict(sub); mod = mod.to(dev)
    opt = FlatParams(list(mod.parameters())); opt.zero_grad()
    x = torch.randn(shape).relu_().bfloat16().float()
    res = {}
    for name, rounding in (("round", (RoundBf16.apply, RoundBf16Forward.apply)), ("exact", None)):
        osd = {k: v.clone().double().requires_grad_(True) for k, v in sub.items()}
        xr = x.double().requires_grad_(True)
        R.ISLAND_ROUNDING = rounding
        try:
            yr = ofn(osd, "", xr)
            if name == "round":
                dy = torch.randn_like(yr).bfloat16().double()
            yr.backward(dy)
        finally:
            R.ISLAND_ROUNDING = None
        res[name] = (yr.detach(), xr.grad, {k: v.grad for k, v in osd.items()})
    xd = x.to(dev).requires_grad_(True)
    y = mod(HF.to_channels_last(xd))
    y.backward(dy.float().to(dev).to(y.dtype))
    torch.cuda.synchronize()
    f = (rel(y.float(), res["round"][0]), rel(res["round"][0], res["exact"][0]), rel(y.float(), res["exact"][0]))
    d = (rel(xd.grad, res["round"][1]), rel(res["round"][1], res["exact"][1]), rel(xd.grad, res["exact"][1]))
    worst = (0, "", 0)
    for n, p in mod.named_parameters():
        gr, gx = res["round"][2][n], res["exact"][2][n]
        if gr is None or gr.numel() < 256: continue
        e = rel(p.grad, gr)
        if e > worst[0]: worst = (e, n, rel(gr, gx))
    print("%-16s %.2e / %.2e / %.2e       %.2e / %.2e / %.2e       %s %.2e / %.2e" % ((tag,) + f + d + (worst[1], worst[0], worst[2])), flush=True)
HF.set_compute_dtype("f32")
