"""BarGen agent that main.py runs (reference: agent/barGen2.py).

Per-step schedule restated from agent/barGen2.py:198-299:
  * every iteration: a generator step -- Model forward, three frozen z-discriminator forwards
    judged "valid", Loss(gen, note, is_pretraining = epoch <= pretraining_step_size), backward,
    Adam;
  * once epoch > pretraining_step_size and (epoch + it) is even: first a z-discriminator step
    on the frozen generator's latents vs N(0, sigma^2) priors (real -> fake_target,
    prior -> valid_target, exactly as the reference labels them).
Every tensor op runs on the HIP library; data-parallel ranks exchange flat gradient buckets
over RCCL."""
import os

import torch
from tqdm import tqdm

from agent.base import AgentBase, Net, make_summary_writer
from data.bar_dataset import NoteDataset
from graph.loss.bar_loss import DLoss, Loss
from graph.model import Model
from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
from hipops import dist as hdist
from hipops import functional as HF
from metrics import AverageMeter


class BarGen(AgentBase):
    def __init__(self, config):
        super().__init__(config)
        self.dataset = self.make_dataset()
        self.dataloader = self.make_loader(self.dataset)

        self.generator = Model().to(self.device)
        self.z_discriminator_phrase = PhraseZDiscriminator().to(self.device)
        self.z_discriminator_bar = BarZDiscriminator().to(self.device)

        self.loss_generator = Loss().to(self.device)
        self.loss_bar = DLoss()
        self.loss_phrase = DLoss()

        lr, mb = self.config.learning_rate, getattr(self.config, "grad_bucket_mb", 64)
        self.lr_generator = self.lr_Zdiscriminator_bar = self.lr_Zdiscriminator_phrase = lr
        self.net_generator = Net(self.generator, lr, mb)
        self.net_zbar = Net(self.z_discriminator_bar, lr, mb)
        self.net_zphrase = Net(self.z_discriminator_phrase, lr, mb)
        self.opt_generator = self.net_generator.opt
        self.opt_Zdiscriminator_bar = self.net_zbar.opt
        self.opt_Zdiscriminator_phrase = self.net_zphrase.opt
        self.scheduler_generator = self.net_generator.scheduler
        self.scheduler_Zdiscriminator_bar = self.net_zbar.scheduler
        self.scheduler_Zdiscriminator_phrase = self.net_zphrase.scheduler

        self.seed_everything()
        self.load_checkpoint(self.config.checkpoint_file)
        self.summary_writer = make_summary_writer(os.path.join(self.config.root_path, self.config.summary_dir), "BarGen") \
            if self.rank == 0 else make_summary_writer(None)
        if self.rank == 0:
            for name, m in (("generator", self.generator), ("barZ discriminator", self.z_discriminator_bar),
                            ("phraseZ discriminator", self.z_discriminator_phrase)):
                print("Number of {} parameters: {}".format(name, sum(p.numel() for p in m.parameters())))

    # ------------------------------------------------------------------ checkpoints
    def load_checkpoint(self, file_name):
        ck = self._load(file_name)
        if ck is None:
            return
        self.net_generator.load_state_dict(ck["generator_state_dict"])
        self.opt_generator.load_state_dict(ck["generator_optimizer"])
        self.net_zbar.load_state_dict(ck["z_discriminator_bar_state_dict"])
        self.opt_Zdiscriminator_bar.load_state_dict(ck["opt_Zdiscriminator_bar_optimizer"])
        self.net_zphrase.load_state_dict(ck["z_discriminator_phrase_state_dict"])
        self.opt_Zdiscriminator_phrase.load_state_dict(ck["opt_Zdiscriminator_phrase_optimizer"])
        extra = ck.get("mgvae_extra")      # what the reference forgets to save
        if extra:
            self.epoch, self.iteration = extra["epoch"], extra["iteration"]
            for s, k in ((self.scheduler_generator, "sched_g"), (self.scheduler_Zdiscriminator_bar, "sched_zb"),
                         (self.scheduler_Zdiscriminator_phrase, "sched_zp")):
                s.load_state_dict(extra[k])

    def save_checkpoint(self, file_name, epoch):
        self._save({
            "generator_state_dict": self.net_generator.state_dict(),
            "generator_optimizer": self.opt_generator.state_dict(),
            "z_discriminator_bar_state_dict": self.net_zbar.state_dict(),
            "opt_Zdiscriminator_bar_optimizer": self.opt_Zdiscriminator_bar.state_dict(),
            "z_discriminator_phrase_state_dict": self.net_zphrase.state_dict(),
            "opt_Zdiscriminator_phrase_optimizer": self.opt_Zdiscriminator_phrase.state_dict(),
            "mgvae_extra": {"epoch": self.epoch, "iteration": self.iteration,
                            "sched_g": self.scheduler_generator.state_dict(),
                            "sched_zb": self.scheduler_Zdiscriminator_bar.state_dict(),
                            "sched_zp": self.scheduler_Zdiscriminator_phrase.state_dict()},
        }, epoch)

    # ------------------------------------------------------------------ schedule (agent/barGen2.py:233,288)
    @staticmethod
    def runs_discriminator_step(epoch, curr_it, pretraining_step_size):
        return epoch > pretraining_step_size and (epoch + curr_it) % 2 == 0

    @staticmethod
    def is_pretraining(epoch, pretraining_step_size):
        return epoch <= pretraining_step_size

    # ------------------------------------------------------------------ training
    def train(self):
        for _ in range(self.config.epoch):
            self.epoch += 1
            self.train_epoch()
            if self.epoch > self.pretraining_step_size + 50:
                self.save_checkpoint(self.config.checkpoint_file, self.epoch)

    def discriminator_step(self, note, pre_note, pre_phrase, position):
        """agent/barGen2.py:233-265"""
        self.free(self.z_discriminator_bar)
        self.free(self.z_discriminator_phrase)
        self.frozen(self.generator)
        _, z, pre_z, phrase_feature = self.generator(note, pre_note, pre_phrase, position)
        b, sigma = z.size(0), self.config.sigma
        # phrase latent: real -> fake_target (0), prior sample -> valid_target (1)
        phrase_fake = HF.randn((b, phrase_feature.size(1)), sigma, self.device)
        d_fake = self.z_discriminator_phrase(phrase_fake).view(-1)
        d_real = self.z_discriminator_phrase(phrase_feature).view(-1)
        phrase_loss = DLoss.constant(d_real, 0.0) + DLoss.constant(d_fake, 1.0)
        bar_fake = HF.randn((b * 2, z.size(1)), sigma, self.device)
        d_bar_fake = self.z_discriminator_bar(bar_fake).view(-1)
        d_bar_real1 = self.z_discriminator_bar(z).view(-1)
        d_bar_real2 = self.z_discriminator_bar(pre_z).view(-1)
        bar_loss = DLoss.constant(d_bar_real1, 0.0) + DLoss.constant(d_bar_real2, 0.0) + DLoss.constant(d_bar_fake, 1.0)
        phrase_loss.backward()
        bar_loss.backward()
        self.net_zbar.step()
        self.net_zphrase.step()
        return bar_loss, phrase_loss

    def generator_step(self, note, pre_note, pre_phrase, position):
        """agent/barGen2.py:267-292"""
        self.free(self.generator)
        self.frozen(self.z_discriminator_bar)
        self.frozen(self.z_discriminator_phrase)
        gen_note, z, pre_z, phrase_feature = self.generator(note, pre_note, pre_phrase, position)
        if hdist.is_dist():          # decoder gradients are complete once its three inputs have theirs
            left = [3]

            def fire(_g, left=left):
                left[0] -= 1
                if left[0] == 0:
                    self.net_generator.reducer.reduce_range(*self.decoder_range())
            for t in (z, pre_z, phrase_feature):
                t.register_hook(fire)
        loss = DLoss.constant(self.z_discriminator_phrase(phrase_feature).view(-1), 1.0)
        loss = loss + DLoss.constant(self.z_discriminator_bar(z).view(-1), 1.0) \
            + DLoss.constant(self.z_discriminator_bar(pre_z).view(-1), 1.0)
        loss = loss + self.loss_generator(gen_note, note, self.is_pretraining(self.epoch, self.pretraining_step_size))
        loss.backward()
        self.net_generator.step()
        return loss, gen_note

    def decoder_range(self):
        if not hasattr(self, "_dec_range"):
            opt = self.opt_generator
            ids = {id(p): i for i, p in enumerate(opt.params)}
            idx = [ids[id(p)] for p in self.generator.decoder.parameters()]
            lo, hi = min(idx), max(idx)
            self._dec_range = (opt.offsets[lo], (opt.offsets[hi] + opt.params[hi].numel() + 63) // 64 * 64)
        return self._dec_range

    def train_epoch(self):
        it_total = (len(self.dataloader.sampler) + self.batch_size - 1) // self.batch_size if self.world > 1 \
            else self.dataset.num_iterations
        batches = tqdm(self.dataloader, total=it_total, desc="epoch-{}".format(self.epoch), disable=self.rank != 0)
        image_sample = origin_image = None
        avg_generator_loss, avg_barZ_disc_loss, avg_phraseZ_disc_loss = AverageMeter(), AverageMeter(), AverageMeter()
        curr_it = 0
        for curr_it, batch in enumerate(batches):
            note, pre_note, pre_phrase, position = self.to_device(*batch)
            self.iteration += 1
            self.generator.train(); self.z_discriminator_bar.train(); self.z_discriminator_phrase.train()
            self.net_generator.zero_grad(); self.net_zbar.zero_grad(); self.net_zphrase.zero_grad()
            if self.runs_discriminator_step(self.epoch, curr_it, self.pretraining_step_size):
                bar_loss, phrase_loss = self.discriminator_step(note, pre_note, pre_phrase, position)
                avg_barZ_disc_loss.update(bar_loss)
                avg_phraseZ_disc_loss.update(phrase_loss)
            loss, gen_note = self.generator_step(note, pre_note, pre_phrase, position)
            image_sample, origin_image = gen_note, note
            avg_generator_loss.update(loss)
        batches.close()
        if image_sample is None:
            return
        g_loss = hdist.all_reduce_mean_scalar(float(avg_generator_loss.val))
        zb_loss = hdist.all_reduce_mean_scalar(float(avg_barZ_disc_loss.val))
        zp_loss = hdist.all_reduce_mean_scalar(float(avg_phraseZ_disc_loss.val))
        w = self.summary_writer
        w.add_scalar("train/Generator_loss", g_loss, self.epoch)
        if self.epoch > self.pretraining_step_size:
            w.add_scalar("train/Bar_Z_Discriminator_loss", zb_loss, self.epoch)
            w.add_scalar("train/Phrase_Z_discriminator_loss", zp_loss, self.epoch)
        if self.rank == 0:
            k = min(3, image_sample.size(0))
            binar = torch.gt(image_sample, 0.3).float()
            for i in range(k):
                w.add_image("train/sample %d" % (i + 1), image_sample[i].detach().reshape(1, 96, 60).cpu(), self.epoch)
                w.add_image("train/sample_binarization %d" % (i + 1), binar[i].reshape(1, 96, 60).cpu(), self.epoch)
                w.add_image("train/origin %d" % (i + 1), origin_image[i].reshape(1, 96, 60).cpu(), self.epoch)
            self.generator.eval(); self.z_discriminator_bar.eval(); self.z_discriminator_phrase.eval()
            outputs = self.sample_phrases(self.generator, 10)
            w.add_image("eval/generated 1", outputs[0].reshape(1, 96 * 4, 60), self.epoch)
            w.add_image("eval/generated 2", outputs[1].reshape(1, 96 * 4, 60), self.epoch)
        self.scheduler_generator.step(g_loss)
        if self.epoch > self.pretraining_step_size and (self.epoch + curr_it) % 2 == 0:
            self.scheduler_Zdiscriminator_bar.step(zb_loss)
            self.scheduler_Zdiscriminator_phrase.step(zp_loss)
        self.logger.warning("loss info - generator: {}, barZ disc: {},  phraseZ disc: {}".format(g_loss, zb_loss, zp_loss))
        self.logger.warning("lr info - generator: {}, barZ disc: {},  phraseZ disc: {}".format(
            self.get_lr(self.opt_generator), self.get_lr(self.opt_Zdiscriminator_bar), self.get_lr(self.opt_Zdiscriminator_phrase)))
