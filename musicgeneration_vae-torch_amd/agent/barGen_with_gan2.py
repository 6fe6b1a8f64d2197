"""Variant that trains all four discriminators every step once pre-training is over
(reference: agent/barGen_with_gan2.py): pre-training uses batch_size + 2 and also trains the
latent discriminators (:407-468); afterwards every iteration runs train_discriminator
(:345-405) and then one generator step whose loss is reconstruction + latent adversarial +
0.05 x (bar + feature) GAN terms on a sample decoded from N(0, 1.5^2) noise (:470-519)."""
import torch

from agent.gan_base import GanAgentBase
from graph.loss.bar_loss import DLoss

C = DLoss.constant


class BarGen(GanAgentBase):
    batch_bonus = 2
    save_after = 20

    def after_epoch(self):
        """agent/barGen_with_gan2.py:251-254: drop the +2 once pre-training ends"""
        if self.epoch == self.pretraining_step_size:
            self.batch_size -= 2
            self.dataloader = self.make_loader(self.dataset)

    def step_batch(self, batch, curr_it, meters):
        note, pre_note, pre_phrase, position = batch
        if self.epoch <= self.pretraining_step_size:
            return self.train_pretrain(note, pre_note, pre_phrase, position, meters)
        self.train_discriminator(note, pre_note, pre_phrase, position, meters)
        return self.train_add_gan(note, pre_note, pre_phrase, position, meters)

    def _latent_disc_losses(self, z, phrase_feature):
        b = z.size(0)
        d_phrase_fake = self.z_discriminator_phrase(self.prior(b, self.config.sigma)).view(-1)
        d_phrase_real = self.z_discriminator_phrase(phrase_feature).view(-1)
        phrase_loss = C(d_phrase_real, 0.0) + C(d_phrase_fake, 1.0)
        d_bar_fake = self.z_discriminator_bar(self.prior(b, self.config.sigma)).view(-1)
        d_bar_real = self.z_discriminator_bar(z).view(-1)
        bar_loss = C(d_bar_real, 0.0) + C(d_bar_fake, 1.0)
        return phrase_loss, bar_loss

    def train_discriminator(self, note, pre_note, pre_phrase, position, meters):
        self.zero("discriminator", "discriminator_feature", "z_discriminator_bar", "z_discriminator_phrase")
        self.only_trainable("discriminator", "discriminator_feature", "z_discriminator_bar", "z_discriminator_phrase")
        gen_note, z, pre_z, phrase_feature, gen_z = self.generator(note, pre_note, pre_phrase, position)
        phrase_loss, bar_loss = self._latent_disc_losses(z, phrase_feature)
        out = torch.gt(gen_note, 0.3).float()
        d_note_fake = self.discriminator(self.pair(pre_note, out)).view(-1)
        d_note_real = self.discriminator(self.pair(pre_note, note)).view(-1)
        note_loss = C(d_note_real, 0.0) + C(d_note_fake, 1.0)
        feature_loss = C(self.discriminator_feature(z).view(-1), 0.0) + C(self.discriminator_feature(gen_z).view(-1), 1.0)
        for l in (phrase_loss, bar_loss, note_loss, feature_loss):
            l.backward()
        for n in ("z_discriminator_bar", "z_discriminator_phrase", "discriminator", "discriminator_feature"):
            self.nets[n].step()
        meters["z_bar"].update(bar_loss); meters["z_phrase"].update(phrase_loss)
        meters["discriminator"].update(note_loss); meters["discriminator_feature"].update(feature_loss)

    def train_pretrain(self, note, pre_note, pre_phrase, position, meters):
        self.modes(train=("generator", "z_discriminator_bar", "z_discriminator_phrase"), evaluate=("discriminator", "discriminator_feature"))
        self.zero("generator", "z_discriminator_bar", "z_discriminator_phrase")
        # the reference runs this forward with whatever requires_grad flags the previous
        # generator step left (generator trainable); gradients into the generator are zeroed below
        gen_note, z, pre_z, phrase_feature, gen_z = self.generator(note, pre_note, pre_phrase, position)
        phrase_loss, bar_loss = self._latent_disc_losses(z.detach(), phrase_feature.detach())
        phrase_loss.backward()
        bar_loss.backward()
        self.nets["z_discriminator_bar"].step()
        self.nets["z_discriminator_phrase"].step()
        meters["z_bar"].update(bar_loss); meters["z_phrase"].update(phrase_loss)
        self.only_trainable("generator")
        self.zero("generator")
        gen_note, z, pre_z, phrase_feature, _ = self.generator(note, pre_note, pre_phrase, position)
        loss = self.loss_generator(gen_note, note, True)
        loss.backward()
        self.nets["generator"].step()
        meters["generator"].update(loss)
        return gen_note[:3]

    def train_add_gan(self, note, pre_note, pre_phrase, position, meters):
        self.modes(train=("generator", "z_discriminator_bar", "z_discriminator_phrase"), evaluate=("discriminator", "discriminator_feature"))
        self.zero("generator", "z_discriminator_bar", "z_discriminator_phrase")
        self.only_trainable("generator")
        gen_note, z, pre_z, phrase_feature, _ = self.generator(note, pre_note, pre_phrase, position)
        loss = C(self.z_discriminator_phrase(phrase_feature).view(-1), 1.0)
        loss = loss + C(self.z_discriminator_bar(z).view(-1), 1.0) + C(self.z_discriminator_bar(pre_z).view(-1), 1.0)
        loss = loss + self.loss_generator(gen_note, note, False)
        noise = self.prior(note.size(0), 1.5)
        gen_note, gen_z = self.generator(noise, pre_note, pre_phrase, position, False)
        out = torch.gt(gen_note, 0.3).float()          # binarised: no gradient reaches the generator here
        loss = loss + C(self.discriminator(self.pair(pre_note, out)).view(-1), 1.0) * 0.05
        loss = loss + C(self.discriminator_feature(gen_z).view(-1), 1.0) * 0.05
        loss.backward()
        self.nets["generator"].step()
        meters["generator"].update(loss)
        return gen_note[:3]
