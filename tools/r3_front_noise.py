"""Run-to-run spread of the Decoder (per-op and chained) in bf16 storage: max and L2 measures (diagnostic for tests/test_nhwc_gpu.py)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "musicgeneration_vae-torch_amd"))
import hipops.functional as HF, hipops.blocks as HB
from hipops.flat import FlatParams
from graph.decoder import Decoder
dev = torch.device("cuda:0")
for storage in ("f32", "bf16"):
    HF.set_compute_dtype(storage)
    for how in ("rng", "masks", "eval"):
        res = []
        for run in range(6):
            HB.ENABLED = run >= 3
            torch.manual_seed(8)
            dec = Decoder([1024, 512, 256, 128, 64]).to(dev)
            with torch.no_grad():
                for prm in dec.parameters():
                    if prm.dim() > 1:
                        prm.copy_(torch.randn_like(prm) * (1.2 / (prm[0].numel() ** 0.5)))
            dec.train(how != "eval")
            gm = torch.Generator().manual_seed(4)
            dec._drop_masks = [((torch.rand(3, 1152, generator=gm) >= 0.3).float() / 0.7).to(dev) for _ in range(2)] if how == "masks" else None
            HF.manual_seed(77)
            opt = FlatParams(list(dec.parameters()))
            opt.zero_grad()
            gi = torch.Generator().manual_seed(5)
            zz = torch.randn(6, 1152, generator=gi).to(dev).requires_grad_(True)
            pf = torch.randn(3, 1152, generator=gi).to(dev).requires_grad_(True)
            pos = torch.tensor([3, 330, 17], device=dev)
            y = dec(zz[:3], zz[3:], pf, pos)
            y.backward(torch.linspace(-1, 1, y.numel(), device=dev).view_as(y))
            torch.cuda.synchronize()
            res.append((y.detach().clone(), zz.grad.clone(), pf.grad.clone(), opt.grad.clone()))
        for i, name in enumerate(("y", "d(z,pre_z)", "d(phrase)", "param grads")):
            a = res[0][i]
            s, l = float(a.abs().max()), float(a.norm())
            out = []
            for r in res[1:]:
                d = r[i] - a
                out.append("%.1e/%.1e/%d" % (float(d.abs().max()) / s, float(d.norm()) / l, int((d.abs() > 1e-2 * s).sum())))
            print(storage, how, "%-12s" % name, "per-op x2:", out[0], out[1], " chained x3:", out[2], out[3], out[4], flush=True)
