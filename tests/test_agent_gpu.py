"""GPU suite: the BarGen agent main.py runs (agent/barGen2.py restated) on a tiny synthetic
dataset -- pre-training and adversarial epochs, checkpoint round trip in the reference's key
layout, and the sampling loop."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_dataset(root, n_files=6, per_file=2, seed=0):
    d = os.path.join(root, "data", "dataset")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(seed)
    for i in range(n_files):
        np.savez(os.path.join(d, "bar_%03d.npz" % i),
                 note=(rng.random((per_file, 1, 96, 60)) < 0.05).astype(np.float32),
                 pre_note=(rng.random((per_file, 1, 96, 60)) < 0.05).astype(np.float32),
                 pre_phrase=(rng.random((per_file, 1, 384, 60)) < 0.05).astype(np.float32),
                 position=rng.integers(0, 332, size=(per_file,)))


def test_bargen2_trains_checkpoints_and_samples(tmp_path):
    import __graft_entry__ as g
    g.build()
    from config import Config
    from agent.barGen2 import BarGen
    root = str(tmp_path)
    _make_dataset(root)

    class Cfg(Config):
        root_path = root
        batch_size = 2          # 2 files x 2 bars = 4 bars per step
        epoch = 2
        pretraining_step_size = 1   # epoch 1 pre-trains, epoch 2 runs the adversarial schedule
        seed = 7
        log_file = os.path.join(root, "train_epoch.log")

    agent = BarGen(Cfg())
    before = agent.opt_generator.flat.clone()
    zb_before = agent.opt_Zdiscriminator_bar.flat.clone()
    agent.run()
    torch.cuda.synchronize()
    assert agent.epoch == 2 and agent.iteration == 6
    assert not torch.equal(before, agent.opt_generator.flat), "generator did not move"
    assert not torch.equal(zb_before, agent.opt_Zdiscriminator_bar.flat), "z-discriminator step never ran in epoch 2"
    assert torch.isfinite(agent.opt_generator.flat).all()
    assert agent.opt_generator.step_count == 6
    log = open(Cfg.log_file).read()
    assert "loss info - generator" in log and "lr info" in log
    # checkpoint: the reference's keys, DataParallel "module." prefixes, torch-Adam state layout
    agent.save_checkpoint(Cfg.checkpoint_file, agent.epoch)
    ck = torch.load(os.path.join(root, "model", "checkpoint.pth.tar"), weights_only=False)
    assert set(ck) >= {"generator_state_dict", "generator_optimizer", "z_discriminator_bar_state_dict",
                       "opt_Zdiscriminator_bar_optimizer", "z_discriminator_phrase_state_dict",
                       "opt_Zdiscriminator_phrase_optimizer"}
    assert all(k.startswith("module.") for k in ck["generator_state_dict"])
    assert set(ck["generator_optimizer"]) == {"state", "param_groups"}
    agent2 = BarGen(Cfg())       # loads checkpoint.pth.tar in its constructor
    assert torch.equal(agent2.opt_generator.flat, agent.opt_generator.flat)
    assert torch.equal(agent2.opt_generator.exp_avg, agent.opt_generator.exp_avg)
    assert agent2.epoch == 2 and agent2.opt_generator.step_count == 6
    # sampling loop (reference agent/barGen2.py:317-336): 3 phrases x 4 bars, binary rolls
    agent2.generator.eval()
    out = agent2.sample_phrases(agent2.generator, 3)
    assert len(out) == 3 and out[0].shape == (384, 60) and set(np.unique(out[0])) <= {0.0, 1.0}


def test_schedule_conditions():
    from agent.barGen2 import BarGen
    assert not BarGen.runs_discriminator_step(220, 0, 220)
    assert BarGen.runs_discriminator_step(221, 1, 220) and not BarGen.runs_discriminator_step(221, 0, 220)
    assert BarGen.is_pretraining(220, 220) and not BarGen.is_pretraining(221, 220)


@pytest.mark.parametrize("which", ["barGen_with_gan", "barGen_with_gan2", "barGen"])
def test_gan_agents_run_every_phase(tmp_path, which):
    """three tiny epochs: pre-training, then the adversarial phases of each agent variant"""
    import importlib
    import __graft_entry__ as g
    g.build()
    from config import Config
    root = str(tmp_path)
    _make_dataset(root, n_files=4, per_file=2)

    class Cfg(Config):
        root_path = root
        batch_size = 2
        epoch = 3
        pretraining_step_size = 1
        seed = 11
        log_file = os.path.join(root, "train_epoch.log")

    mod = importlib.import_module("agent." + which)
    agent = mod.BarGen(Cfg())
    if which == "barGen_with_gan":
        agent.flag_gan = False
    nets0 = {n: getattr(agent, n).state_dict() for n in ("generator", "discriminator", "z_discriminator_bar")}
    nets0 = {n: {k: v.clone() for k, v in sd.items()} for n, sd in nets0.items()}
    agent.run()
    if which == "barGen_with_gan":        # force one epoch of the GAN phase too
        agent.flag_gan = True
        agent.epoch += 1
        agent.train_epoch()
    torch.cuda.synchronize()
    for n, sd0 in nets0.items():
        sd1 = getattr(agent, n).state_dict()
        moved = any(not torch.equal(sd0[k], sd1[k]) for k in sd0 if sd0[k].is_floating_point())
        # with the reference's N(-1,1) init a discriminator's sigmoid can saturate to exactly 0/1 on
        # such a tiny batch, and BCE then gives an exactly-zero gradient (torch semantics): for the
        # discriminators require that their optimizer stepped, for the generator that it moved
        if n == "generator":
            assert moved, "generator never trained in %s" % which
        assert all(torch.isfinite(v).all() for v in sd1.values() if v.is_floating_point()), n
    assert agent.opt_discriminator.step_count > 0 and agent.opt_Zdiscriminator_bar.step_count > 0
    agent.save_checkpoint(Cfg.checkpoint_file, agent.epoch)
    agent2 = mod.BarGen(Cfg())
    for k, v in agent.generator.state_dict().items():
        assert torch.equal(v, agent2.generator.state_dict()[k]), k
    bn = [k for k in agent.discriminator.state_dict() if k.endswith("num_batches_tracked")]
    assert bn
    if which != "barGen_with_gan2":   # that variant keeps the bar discriminator in eval() throughout, like the reference
        assert int(agent2.discriminator.state_dict()[bn[0]]) > 0


def test_maker_bar_sampling_batch_of_songs():
    """config 5 shape: the sampling loop with several independent songs in one batch"""
    import __graft_entry__ as g
    g.build()
    import maker_bar
    from graph.model import Model
    gen = Model().to("cuda").eval()
    roll = maker_bar.sample(gen, music_length=2, songs=3, device="cuda")
    assert tuple(roll.shape) == (3, 768, 60) and set(roll.unique().tolist()) <= {0.0, 1.0}


def test_graph_sampler_replays_the_eager_loop():
    """the HIP-graph sampler (two captured device programs over static buffers) reproduces the eager sampling loop
    bit for bit when the prior noise stream is the same, with and without the D2-fixed refiner"""
    import __graft_entry__ as g
    g.build()
    import maker_bar
    from graph.model import Model
    from hipops import functional as HF
    for refine in (False, True):
        torch.manual_seed(3)
        gen = Model(use_refiner=refine).to("cuda").eval()
        with torch.no_grad():                       # tame the D4 N(-1,1) init so the bars are not all-on / all-off
            for p in gen.parameters():
                p.mul_(0.05)
        HF.manual_seed(11)
        want = maker_bar.sample(gen, music_length=3, songs=2, device="cuda")
        gs = maker_bar.GraphSampler(gen, songs=2, device="cuda")
        HF.manual_seed(11)
        got = gs.sample(3)
        assert tuple(got.shape) == (2, 3 * 384, 60)
        assert float((got != want).float().mean()) < 2e-3, float((got != want).float().mean())
        HF.manual_seed(11)
        again = gs.sample(3)                         # replays are repeatable
        assert float((again != got).float().mean()) < 2e-3


def test_packed_dataset_feeds_the_same_batches(tmp_path):
    """bit-packed input pipeline: the agent with ``config.packed_data_file`` sees exactly the batches of the
    reference-format dataset (bits shipped, expanded on the GPU by mgvae_unpack_bits) and trains on them"""
    import __graft_entry__ as g
    g.build()
    from config import Config
    from data.bar_dataset import pack_dataset
    from agent.barGen2 import BarGen
    from hipops import functional as HF
    root = str(tmp_path)
    _make_dataset(root, n_files=8, per_file=1)      # one sample per file: batch_size then means the same in both datasets
    pack_dataset(os.path.join(root, "data", "dataset"), os.path.join(root, "data", "packed.npz"))
    x = (np.random.default_rng(0).random((3, 5760)) < 0.1)
    bits = torch.from_numpy(np.packbits(x.astype(np.uint8), axis=1, bitorder="little")).cuda()
    assert torch.equal(HF.unpack_bits(bits, (3, 1, 96, 60)).cpu().view(3, -1), torch.from_numpy(x.astype(np.float32)))

    def cfg(packed):
        class Cfg(Config):
            root_path = root
            batch_size = 4
            epoch = 1
            pretraining_step_size = 5
            seed = 11
            packed_data_file = "data/packed.npz" if packed else None
            log_file = os.path.join(root, "train_epoch.log")
        return Cfg()

    a, b = BarGen(cfg(False)), BarGen(cfg(True))
    for ba, bb in zip(a.dataloader, b.dataloader):
        ta, tb = a.to_device(*ba), b.to_device(*bb)
        for u, v in zip(ta, tb):
            assert u.dtype == v.dtype and torch.equal(u, v)
    b.run()
    assert b.epoch == 1
