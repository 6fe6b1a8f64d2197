"""First BarGen agent (reference: agent/barGen.py): generator + BarDiscriminator + the two latent
discriminators, two Adam optimizers over the generator (opt_gen1 while pre-training, opt_gen2
afterwards).  Restated literally, including the reference's own defects:
the reconstruction loss computed after pre-training is OVERWRITTEN by the adversarial terms
(agent/barGen.py:314-316) and the two generator schedulers are stepped swapped (:361-364)."""
import os
import random

import torch
from tqdm import tqdm

from agent.base import AgentBase, Net, ReduceLROnPlateau, make_summary_writer
from data.bar_dataset import NoteDataset
from graph.bar_discriminator import BarDiscriminator
from graph.loss.bar_loss import DLoss, Loss
from graph.model import Model
from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
from hipops import dist as hdist
from hipops import functional as HF
from metrics import AverageMeter

C = DLoss.constant


class BarGen(AgentBase):
    def __init__(self, config):
        super().__init__(config)
        self.dataset = self.make_dataset()
        self.dataloader = self.make_loader(self.dataset)
        self.generator = Model().to(self.device)
        self.discriminator = BarDiscriminator().to(self.device)
        self.z_discriminator_phrase = PhraseZDiscriminator().to(self.device)
        self.z_discriminator_bar = BarZDiscriminator().to(self.device)
        self.loss_gen = Loss().to(self.device)
        self.loss_disc = DLoss()
        lr, mb = self.config.learning_rate, getattr(self.config, "grad_bucket_mb", 64)
        self.lr_gen1 = self.lr_gen2 = self.lr_discriminator = self.lr_Zdiscriminator_bar = self.lr_Zdiscriminator_phrase = lr
        self.net_gen = Net(self.generator, lr, mb)
        self.opt_gen1 = self.net_gen.opt
        self.opt_gen2 = self.opt_gen1.second_state(lr)
        self.scheduler_gen1 = self.net_gen.scheduler
        self.scheduler_gen2 = ReduceLROnPlateau(self.opt_gen2, mode="min", factor=0.8, cooldown=6)
        self.net_disc = Net(self.discriminator, lr, mb)
        self.net_zbar = Net(self.z_discriminator_bar, lr, mb)
        self.net_zphrase = Net(self.z_discriminator_phrase, lr, mb)
        self.opt_discriminator, self.scheduler_discriminator = self.net_disc.opt, self.net_disc.scheduler
        self.opt_Zdiscriminator_bar, self.scheduler_Zdiscriminator_bar = self.net_zbar.opt, self.net_zbar.scheduler
        self.opt_Zdiscriminator_phrase, self.scheduler_Zdiscriminator_phrase = self.net_zphrase.opt, self.net_zphrase.scheduler
        self.seed_everything()
        self.load_checkpoint(self.config.checkpoint_file)
        self.summary_writer = make_summary_writer(os.path.join(self.config.root_path, self.config.summary_dir), "BarGen") \
            if self.rank == 0 else make_summary_writer(None)

    # reference key set: agent/barGen.py:178-193
    def save_checkpoint(self, file_name, epoch):
        self._save({
            "epoch": epoch,
            "generator_state_dict": self.net_gen.state_dict(),
            "gen_optimizer1": self.opt_gen1.state_dict(), "gen_optimizer2": self.opt_gen2.state_dict(),
            "discriminator_state_dict": self.net_disc.state_dict(), "disc_optimizer": self.opt_discriminator.state_dict(),
            "z_discriminator_bar_state_dict": self.net_zbar.state_dict(),
            "opt_Zdiscriminator_bar_optimizer": self.opt_Zdiscriminator_bar.state_dict(),
            "z_discriminator_phrase_state_dict": self.net_zphrase.state_dict(),
            "opt_Zdiscriminator_phrase_optimizer": self.opt_Zdiscriminator_phrase.state_dict(),
            "lr_gen1": self.get_lr(self.opt_gen1), "lr_gen2": self.get_lr(self.opt_gen2),
            "lr_discriminator": self.get_lr(self.opt_discriminator),
            "lr_Zdiscriminator_bar": self.get_lr(self.opt_Zdiscriminator_bar),
            "lr_Zdiscriminator_phrase": self.get_lr(self.opt_Zdiscriminator_phrase),
        }, epoch)

    def load_checkpoint(self, file_name):
        ck = self._load(file_name)
        if ck is None:
            return
        self.net_gen.load_state_dict(ck["generator_state_dict"])
        self.opt_gen1.load_state_dict(ck["gen_optimizer1"]); self.opt_gen2.load_state_dict(ck["gen_optimizer2"])
        self.net_disc.load_state_dict(ck["discriminator_state_dict"]); self.opt_discriminator.load_state_dict(ck["disc_optimizer"])
        self.net_zbar.load_state_dict(ck["z_discriminator_bar_state_dict"])
        self.opt_Zdiscriminator_bar.load_state_dict(ck["opt_Zdiscriminator_bar_optimizer"])
        self.net_zphrase.load_state_dict(ck["z_discriminator_phrase_state_dict"])
        self.opt_Zdiscriminator_phrase.load_state_dict(ck["opt_Zdiscriminator_phrase_optimizer"])
        self.epoch = int(ck.get("epoch", 0))

    def train(self):
        for _ in range(self.config.epoch):
            self.epoch += 1
            self.train_epoch()
            if self.epoch > self.pretraining_step_size + 50:
                self.save_checkpoint(self.config.checkpoint_file, self.epoch)

    def _gen_step(self, opt):
        self.net_gen.reducer.reduce_rest()
        self.net_gen.reducer.wait()
        opt.step(grad_scale=1.0 / hdist.world_size())

    def train_iteration(self, note, pre_note, pre_phrase, position, curr_it, div_flag, meters):
        """one pass of the loop body of agent/barGen.py:231-335; ``meters`` = (gen, disc, z_bar, z_phrase) AverageMeters"""
        avg_gen, avg_disc, avg_zbar, avg_zphrase = meters
        adversarial = self.epoch > self.pretraining_step_size
        for n in (self.net_gen, self.net_disc, self.net_zbar, self.net_zphrase):
            n.zero_grad()
        if (curr_it + self.epoch) % div_flag == 1 and adversarial:
            for m in (self.discriminator, self.z_discriminator_bar, self.z_discriminator_phrase):
                self.free(m)
            self.frozen(self.generator)
            gen_note, z, pre_z, phrase_feature = self.generator(note, pre_note, pre_phrase, position)
            b = z.size(0)
            # this agent labels real -> valid and prior -> fake (opposite of barGen2), sigma = 1 (:265,:271)
            phrase_fake = self.prior(b, 1.0)
            phrase_loss = C(self.z_discriminator_phrase(phrase_feature).view(-1), 1.0) + \
                C(self.z_discriminator_phrase(phrase_fake).view(-1), 0.0)
            bar_fake = self.prior(2 * b, 1.0)
            bar_loss = C(self.z_discriminator_bar(z).view(-1), 1.0) + C(self.z_discriminator_bar(pre_z).view(-1), 1.0) + \
                C(self.z_discriminator_bar(bar_fake).view(-1), 0.0)
            fake = HF.cat_time(pre_note, torch.gt(gen_note, 0.3).float())
            d_fake = self.discriminator(fake).view(-1)              # fake pair first (BatchNorm statistics: :281,:284)
            disc_loss = C(d_fake, 0.0) + C(self.discriminator(HF.cat_time(pre_note, note)).view(-1), 1.0)
            disc_loss.backward(); phrase_loss.backward(); bar_loss.backward()
            self.net_disc.step(); self.net_zbar.step(); self.net_zphrase.step()
            avg_disc.update(disc_loss); avg_zbar.update(bar_loss); avg_zphrase.update(phrase_loss)
        self.free(self.generator)
        for m in (self.discriminator, self.z_discriminator_bar, self.z_discriminator_phrase):
            self.frozen(m)
        gen_note, z, pre_z, phrase_feature = self.generator(note, pre_note, pre_phrase, position)
        if adversarial:
            # D7: the smoothed reconstruction loss is computed by the reference and then overwritten
            gen_loss = C(self.z_discriminator_phrase(phrase_feature).view(-1), 1.0)
            gen_loss = gen_loss + C(self.z_discriminator_bar(z).view(-1), 1.0) + C(self.z_discriminator_bar(pre_z).view(-1), 1.0)
            fake = HF.cat_time(pre_note, torch.gt(gen_note, 0.3).float())
            gen_loss = gen_loss + C(self.discriminator(fake).view(-1), 1.0)
            gen_loss.backward()
            self._gen_step(self.opt_gen2)
        else:
            gen_loss = self.loss_gen(gen_note, note, True)
            gen_loss.backward()
            self._gen_step(self.opt_gen1)
        avg_gen.update(gen_loss)
        return gen_note

    def prior(self, rows, sigma):
        return HF.randn((rows, 1152), sigma, self.device)

    def train_epoch(self):
        it_total = (len(self.dataset) + self.batch_size * self.world - 1) // (self.batch_size * self.world)
        batches = tqdm(self.dataloader, total=it_total, desc="epoch-{}".format(self.epoch), disable=self.rank != 0)
        for m in (self.generator, self.discriminator, self.z_discriminator_bar, self.z_discriminator_phrase):
            m.train()
        avg_gen, avg_disc, avg_zbar, avg_zphrase = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
        div_flag = random.randrange(2, 5)
        image_sample = origin_image = None
        adversarial = self.epoch > self.pretraining_step_size
        for curr_it, batch in enumerate(batches):
            note, pre_note, pre_phrase, position = self.to_device(*batch)
            self.iteration += 1
            image_sample = self.train_iteration(note, pre_note, pre_phrase, position, curr_it, div_flag,
                                                (avg_gen, avg_disc, avg_zbar, avg_zphrase))
            origin_image = note
        batches.close()
        if image_sample is None:
            return
        g, d = hdist.all_reduce_mean_scalar(float(avg_gen.val)), hdist.all_reduce_mean_scalar(float(avg_disc.val))
        zb, zp = hdist.all_reduce_mean_scalar(float(avg_zbar.val)), hdist.all_reduce_mean_scalar(float(avg_zphrase.val))
        w = self.summary_writer
        w.add_scalar("train/Generator_loss" if adversarial else "pre_train/Generator_loss", g, self.iteration)
        # D7: schedulers of the two generator optimizers are stepped swapped in the reference
        (self.scheduler_gen1 if adversarial else self.scheduler_gen2).step(g)
        self.scheduler_discriminator.step(d)
        self.scheduler_Zdiscriminator_bar.step(zb)
        self.scheduler_Zdiscriminator_phrase.step(zp)
        self.logger.warning("loss info - gen: {}, disc: {}, barZ disc: {}, phraseZ disc: {}".format(g, d, zb, zp))
