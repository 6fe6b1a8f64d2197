"""GPU suite: checkpoint compatibility with the reference's files.  The fixture is a checkpoint in the
reference's exact layout built WITHOUT this build's code (tests/ref_checkpoint.py: plain torch modules named per
tests/golden/manifest.json + the 30 Refiner entries, nn.DataParallel ``module.`` prefixes, torch.optim.Adam state
after two steps).  The agent must load it (agent/barGen2.py:145-166), continue Adam exactly where torch would, train,
save in the same layout, and the saved file must load back into torch modules + torch.optim.Adam."""
import os

import numpy as np
import pytest
import torch

import ref_checkpoint as RC

pytestmark = pytest.mark.gpu


def test_reference_layout_checkpoint_round_trip(tmp_path):
    import __graft_entry__ as g
    g.build()
    from test_agent_gpu import _make_dataset
    from config import Config
    from agent.barGen2 import BarGen
    import maker_bar
    root = str(tmp_path)
    _make_dataset(root, n_files=2, per_file=2)
    ck, nets = RC.build(seed=3, lr=0.0016)
    assert len(ck["generator_state_dict"]) == 251 and all(k.startswith("module.") for k in ck["generator_state_dict"])
    assert sum(k.startswith("module.refiner.") for k in ck["generator_state_dict"]) == 30
    assert len(ck["generator_optimizer"]["state"]) == 239 and ck["generator_optimizer"]["param_groups"][0]["lr"] == 0.0016
    os.makedirs(os.path.join(root, "model"))
    torch.save(ck, os.path.join(root, "model", "checkpoint.pth.tar"))

    class Cfg(Config):
        root_path = root
        batch_size = 2
        epoch = 1
        pretraining_step_size = 5
        seed = 2
        log_file = os.path.join(root, "train_epoch.log")

    agent = BarGen(Cfg())                    # loads model/checkpoint.pth.tar in its constructor
    dp, topt = nets["generator"]
    ref = {k[len("module."):]: v for k, v in dp.state_dict().items()}
    # (1) weights, both Adam moments, step counter and learning rate arrived
    for k, v in agent.generator.state_dict().items():
        assert torch.equal(v.cpu(), ref[k]), k
    opt = agent.opt_generator
    tparams = [p for p in dp.parameters()]
    for i, (p, o) in enumerate(zip(opt.params, opt.offsets)):
        st = topt.state[tparams[i]]
        assert torch.equal(opt._view(opt.exp_avg, p, o).cpu(), st["exp_avg"]), i          # (logical comparison: a
        assert torch.equal(opt._view(opt.exp_avg_sq, p, o).cpu(), st["exp_avg_sq"]), i    # channels-last weight is permuted)
    assert opt.step_count == 2 and abs(opt.param_groups[0]["lr"] - 0.0016) < 1e-12
    assert agent.opt_Zdiscriminator_bar.step_count == 2
    # (2) the next Adam step continues torch's: same gradient into both, third step's bias corrections
    rng = np.random.default_rng(0)
    opt.zero_grad()
    for i, p in enumerate(opt.params):
        gnp = (rng.standard_normal(tuple(p.shape)) * 0.01).astype(np.float32)
        p.grad.copy_(torch.from_numpy(gnp))
        tparams[i].grad = torch.from_numpy(gnp.copy())
    for p in tparams[len(opt.params):]:
        p.grad = None                        # refiner entries: no gradient, torch skips them
    opt.step()
    topt.step()
    torch.cuda.synchronize()
    worst = 0.0
    for i, p in enumerate(opt.params):
        d = (p.detach().cpu() - tparams[i].detach()).abs().max().item()
        worst = max(worst, d / 0.0016)
    assert worst < 1e-3, worst               # a step moves an entry by ~lr; the two agree to 1e-3 of that
    # (3) train one epoch, save: the reference's keys and layouts again
    agent.run()
    torch.cuda.synchronize()
    agent.save_checkpoint(Cfg.checkpoint_file, agent.epoch)
    ck2 = torch.load(os.path.join(root, "model", "checkpoint.pth.tar"), weights_only=False)
    assert set(ck) <= set(ck2)
    steps = agent.opt_generator.step_count
    assert steps > 3
    # (4) the saved file loads into plain torch modules + torch.optim.Adam (refiner-less generator: 221 entries)
    man = RC.manifest()
    dp2 = RC.DataParallelKeys(RC.bag(RC.generator_entries(man), np.random.default_rng(9)))
    dp2.load_state_dict(ck2["generator_state_dict"])            # strict
    topt2 = torch.optim.Adam(dp2.parameters(), lr=1.0)
    topt2.load_state_dict(ck2["generator_optimizer"])
    p0 = next(iter(dp2.parameters()))
    assert int(topt2.state[p0]["step"]) == steps and abs(topt2.param_groups[0]["lr"] - agent.get_lr(agent.opt_generator)) < 1e-12
    g0 = agent.opt_generator
    assert torch.equal(topt2.state[p0]["exp_avg"], g0._view(g0.exp_avg, g0.params[0], g0.offsets[0]).cpu())
    for p in dp2.parameters():
        p.grad = torch.zeros_like(p)
    topt2.step()                                                 # usable, not only loadable
    # (5) the sampling script's loader takes the reference file (refiner.* entries ignored) and refuses another layout
    torch.save(ck, os.path.join(root, "model", "checkpoint.pth.tar"))
    gen = maker_bar.load_generator(Cfg(), torch.device("cuda", 0))
    assert torch.equal(gen.encoder.linear.weight.cpu(), ref["encoder.linear.weight"])
    bad = dict(ck, generator_state_dict={k.replace("encoder.linear", "encoder.fc"): v for k, v in ck["generator_state_dict"].items()})
    torch.save(bad, os.path.join(root, "model", "checkpoint.pth.tar"))
    with pytest.raises(RuntimeError, match="does not match the generator"):
        maker_bar.load_generator(Cfg(), torch.device("cuda", 0))
