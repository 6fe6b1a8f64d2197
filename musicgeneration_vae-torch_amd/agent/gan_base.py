"""Shared set-up of the agents that train the generator against four discriminators
(reference: agent/barGen_with_gan.py:31-217, agent/barGen_with_gan2.py:30-240): the generator
of graph/model_with_gan.py, BarDiscriminator, BarFeatureDiscriminator and the two latent
discriminators, each with its flat HIP Adam, RCCL reducer and ReduceLROnPlateau."""
import os

import torch
from tqdm import tqdm

from agent.base import AgentBase, Net, make_summary_writer
from data.bar_dataset import NoteDataset
from graph.bar_discriminator import BarDiscriminator
from graph.bar_discriminator_with_feature import BarFeatureDiscriminator
from graph.loss.bar_loss import DLoss, Loss
from graph.model_with_gan import Model
from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
from hipops import dist as hdist
from hipops import functional as HF
from metrics import AverageMeter

NETS = (("generator", "generator_state_dict", "generator_optimizer"),
        ("discriminator", "discriminator_state_dict", "discriminator_optimizer"),
        ("discriminator_feature", "discriminator_feature_state_dict", "discriminator_feature_optimizer"),
        ("z_discriminator_bar", "z_discriminator_bar_state_dict", "opt_Zdiscriminator_bar_optimizer"),
        ("z_discriminator_phrase", "z_discriminator_phrase_state_dict", "opt_Zdiscriminator_phrase_optimizer"))


class GanAgentBase(AgentBase):
    batch_bonus = 0           # barGen_with_gan2 pre-trains with batch_size + 2
    save_after = 20           # checkpoints once epoch > pretraining_step_size + save_after

    def __init__(self, config):
        super().__init__(config)
        self.flag_gan = False
        self.train_count = 0
        self.batch_size = self.config.batch_size + self.batch_bonus
        self.dataset = self.make_dataset()
        self.dataloader = self.make_loader(self.dataset)
        self.generator = Model().to(self.device)
        self.discriminator = BarDiscriminator().to(self.device)
        self.discriminator_feature = BarFeatureDiscriminator().to(self.device)
        self.z_discriminator_phrase = PhraseZDiscriminator().to(self.device)
        self.z_discriminator_bar = BarZDiscriminator().to(self.device)
        self.loss_generator = Loss().to(self.device)
        self.loss_disc = self.loss_feature_disc = self.loss_bar = self.loss_phrase = DLoss()
        lr, mb = self.config.learning_rate, getattr(self.config, "grad_bucket_mb", 64)
        self.nets = {name: Net(getattr(self, name), lr, mb) for name, _, _ in NETS}
        self.opt_generator = self.nets["generator"].opt
        self.opt_discriminator = self.nets["discriminator"].opt
        self.opt_discriminator_feature = self.nets["discriminator_feature"].opt
        self.opt_Zdiscriminator_bar = self.nets["z_discriminator_bar"].opt
        self.opt_Zdiscriminator_phrase = self.nets["z_discriminator_phrase"].opt
        self.scheduler_generator = self.nets["generator"].scheduler
        self.scheduler_discriminator = self.nets["discriminator"].scheduler
        self.scheduler_discriminator_feature = self.nets["discriminator_feature"].scheduler
        self.scheduler_Zdiscriminator_bar = self.nets["z_discriminator_bar"].scheduler
        self.scheduler_Zdiscriminator_phrase = self.nets["z_discriminator_phrase"].scheduler
        self.seed_everything()
        self.load_checkpoint(self.config.checkpoint_file)
        self.summary_writer = make_summary_writer(os.path.join(self.config.root_path, self.config.summary_dir), "BarGen") \
            if self.rank == 0 else make_summary_writer(None)

    # -------------------------------------------------------------- helpers
    def modes(self, train=(), evaluate=()):
        for n in train:
            getattr(self, n).train()
        for n in evaluate:
            getattr(self, n).eval()

    def zero(self, *names):
        for n in names:
            self.nets[n].zero_grad()

    def only_trainable(self, *names):
        for n, _, _ in NETS:
            (self.free if n in names else self.frozen)(getattr(self, n))

    def prior(self, rows, sigma):
        return HF.randn((rows, 1152), sigma, self.device)

    def pair(self, pre_note, bar):
        return HF.cat_time(pre_note, bar)

    # -------------------------------------------------------------- checkpoints (agent/barGen_with_gan.py:169-217)
    def load_checkpoint(self, file_name):
        ck = self._load(file_name)
        if ck is None:
            return
        for name, sk, ok in NETS:
            self.nets[name].load_state_dict(ck[sk])
            self.nets[name].opt.load_state_dict(ck[ok])
        extra = ck.get("mgvae_extra")
        if extra:
            self.epoch, self.iteration = extra["epoch"], extra["iteration"]
            self.flag_gan, self.train_count = extra["flag_gan"], extra["train_count"]
            for name, _, _ in NETS:
                self.nets[name].scheduler.load_state_dict(extra["sched"][name])

    def save_checkpoint(self, file_name, epoch):
        state = {}
        for name, sk, ok in NETS:
            state[sk] = self.nets[name].state_dict()
            state[ok] = self.nets[name].opt.state_dict()
        state["mgvae_extra"] = {"epoch": self.epoch, "iteration": self.iteration, "flag_gan": self.flag_gan,
                                "train_count": self.train_count,
                                "sched": {n: self.nets[n].scheduler.state_dict() for n, _, _ in NETS}}
        self._save(state, epoch)

    # -------------------------------------------------------------- epoch skeleton
    def train(self):
        for _ in range(self.config.epoch):
            self.epoch += 1
            self.train_epoch()
            if self.epoch > self.pretraining_step_size + self.save_after:
                self.save_checkpoint(self.config.checkpoint_file, self.epoch)
            self.after_epoch()

    def after_epoch(self):
        pass

    def step_batch(self, batch, curr_it, meters):
        raise NotImplementedError

    def train_epoch(self):
        if self.epoch > self.pretraining_step_size:
            self.train_count += 1
        it_total = (len(self.dataloader.sampler) + self.batch_size - 1) // self.batch_size if self.world > 1 \
            else (len(self.dataset) + self.batch_size - 1) // self.batch_size
        batches = tqdm(self.dataloader, total=it_total, desc="epoch-{}".format(self.epoch), disable=self.rank != 0)
        meters = {k: AverageMeter() for k in ("generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase")}
        image_sample = origin_image = None
        for curr_it, batch in enumerate(batches):
            self.iteration += 1
            batch = self.to_device(*batch)
            origin_image = batch[0]
            image_sample = self.step_batch(batch, curr_it, meters)
        batches.close()
        if image_sample is None:
            return
        self.toggle_phase()
        vals = {k: hdist.all_reduce_mean_scalar(float(m.val)) for k, m in meters.items()}
        w = self.summary_writer
        for k, v in vals.items():
            w.add_scalar("train/%s_loss" % k, v, self.epoch)
        if self.rank == 0:
            for n, _, _ in NETS:
                getattr(self, n).eval()
            outputs = self.sample_phrases(self.generator, 10)
            self.record_image(image_sample[:3], origin_image[:3], outputs)
        self.scheduler_generator.step(vals["generator"])
        if self.epoch > self.pretraining_step_size:
            self.scheduler_discriminator.step(vals["discriminator"])
            self.scheduler_discriminator_feature.step(vals["discriminator_feature"])
            self.scheduler_Zdiscriminator_bar.step(vals["z_bar"])
            self.scheduler_Zdiscriminator_phrase.step(vals["z_phrase"])
        self.logger.warning("loss info - gen: {}, barZ disc: {},  phraseZ disc: {}, bar disc: {}, bar_seq disc: {}".format(
            vals["generator"], vals["z_bar"], vals["z_phrase"], vals["discriminator_feature"], vals["discriminator"]))
        self.logger.warning("lr info - gen: {}, barZ disc: {},  phraseZ disc: {}, bar disc: {}, bar_seq disc: {}".format(
            self.get_lr(self.opt_generator), self.get_lr(self.opt_Zdiscriminator_bar), self.get_lr(self.opt_Zdiscriminator_phrase),
            self.get_lr(self.opt_discriminator_feature), self.get_lr(self.opt_discriminator)))

    def toggle_phase(self):
        pass

    def record_image(self, samples, origins, outputs):
        w = self.summary_writer
        binar = torch.gt(samples, 0.3).float()
        for i in range(samples.size(0)):
            w.add_image("train/sample %d" % (i + 1), samples[i].detach().reshape(1, 96, 60).cpu(), self.epoch)
            w.add_image("train/sample_binarization %d" % (i + 1), binar[i].reshape(1, 96, 60).cpu(), self.epoch)
            w.add_image("train/origin %d" % (i + 1), origins[i].reshape(1, 96, 60).cpu(), self.epoch)
        w.add_image("eval/generated 1", outputs[0].reshape(1, 96 * 4, 60), self.epoch)
        w.add_image("eval/generated 2", outputs[1].reshape(1, 96 * 4, 60), self.epoch)
