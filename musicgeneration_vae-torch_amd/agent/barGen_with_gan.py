"""BarGen with bar / feature GAN phases (reference: agent/barGen_with_gan.py).

Schedule restated from agent/barGen_with_gan.py:252-305,351-537:
  epoch <= pretraining_step_size          -> train_pretrain (reconstruction only)
  afterwards, blocks of 50 epochs of train_wae (latent discriminators on odd (epoch+it), then a
  generator step with the smoothed reconstruction loss) alternate with blocks of 100 epochs of
  train_gan (bar + feature discriminators on odd (epoch+it), then a generator step from
  N(0, 1.5^2) prior noise judged by both discriminators).  Label conventions are the
  reference's (real -> fake_target, generated -> valid_target in the discriminator steps)."""
from agent.gan_base import GanAgentBase
from graph.loss.bar_loss import DLoss

C = DLoss.constant


class BarGen(GanAgentBase):
    save_after = 20

    def toggle_phase(self):
        """agent/barGen_with_gan.py:300-305"""
        if self.flag_gan and self.train_count >= 100:
            self.flag_gan, self.train_count = False, 0
        elif not self.flag_gan and self.train_count >= 50:
            self.flag_gan, self.train_count = True, 0

    def step_batch(self, batch, curr_it, meters):
        note, pre_note, pre_phrase, position = batch
        if self.epoch <= self.pretraining_step_size:
            return self.train_pretrain(note, pre_note, pre_phrase, position, meters)
        if self.flag_gan:
            return self.train_gan(note, pre_note, pre_phrase, position, meters, curr_it)
        return self.train_wae(note, pre_note, pre_phrase, position, meters, curr_it)

    def train_pretrain(self, note, pre_note, pre_phrase, position, meters):
        self.modes(train=("generator",), evaluate=("discriminator", "discriminator_feature", "z_discriminator_bar", "z_discriminator_phrase"))
        self.zero("generator")
        self.only_trainable("generator")
        gen_note, z, pre_z, phrase_feature, _ = self.generator(note, pre_note, pre_phrase, position)
        loss = self.loss_generator(gen_note, note, True)
        loss.backward()
        self.nets["generator"].step()
        meters["generator"].update(loss)
        return gen_note[:3]

    def train_wae(self, note, pre_note, pre_phrase, position, meters, curr_it):
        self.modes(train=("generator", "z_discriminator_bar", "z_discriminator_phrase"), evaluate=("discriminator", "discriminator_feature"))
        self.zero("generator", "z_discriminator_bar", "z_discriminator_phrase")
        if (self.epoch + curr_it) % 2:
            self.only_trainable("z_discriminator_bar", "z_discriminator_phrase")
            _, z, pre_z, phrase_feature, _ = self.generator(note, pre_note, pre_phrase, position)
            b = z.size(0)
            d_phrase_fake = self.z_discriminator_phrase(self.prior(b, self.config.sigma)).view(-1)
            d_phrase_real = self.z_discriminator_phrase(phrase_feature).view(-1)
            phrase_loss = C(d_phrase_real, 0.0) + C(d_phrase_fake, 1.0)
            d_bar_fake = self.z_discriminator_bar(self.prior(b, self.config.sigma)).view(-1)
            d_bar_real = self.z_discriminator_bar(z).view(-1)
            bar_loss = C(d_bar_real, 0.0) + C(d_bar_fake, 1.0)
            phrase_loss.backward()
            bar_loss.backward()
            self.nets["z_discriminator_bar"].step()
            self.nets["z_discriminator_phrase"].step()
            meters["z_bar"].update(bar_loss)
            meters["z_phrase"].update(phrase_loss)
        self.only_trainable("generator")
        gen_note, z, pre_z, phrase_feature, _ = self.generator(note, pre_note, pre_phrase, position)
        loss = C(self.z_discriminator_phrase(phrase_feature).view(-1), 1.0)
        loss = loss + C(self.z_discriminator_bar(z).view(-1), 1.0) + C(self.z_discriminator_bar(pre_z).view(-1), 1.0)
        loss = loss + self.loss_generator(gen_note, note, False)
        loss.backward()
        self.nets["generator"].step()
        meters["generator"].update(loss)
        return gen_note[:3]

    def train_gan(self, note, pre_note, pre_phrase, position, meters, curr_it):
        self.modes(train=("generator", "discriminator", "discriminator_feature"), evaluate=("z_discriminator_bar", "z_discriminator_phrase"))
        self.zero("generator", "discriminator", "discriminator_feature")
        if (self.epoch + curr_it) % 2:
            self.only_trainable("discriminator", "discriminator_feature")
            gen_note, z, pre_z, phrase_feature, gen_z = self.generator(note, pre_note, pre_phrase, position)
            d_note_fake = self.discriminator(self.pair(pre_note, gen_note)).view(-1)
            d_note_real = self.discriminator(self.pair(pre_note, note)).view(-1)
            note_loss = C(d_note_real, 0.0) + C(d_note_fake, 1.0)
            d_feature_fake = self.discriminator_feature(gen_z).view(-1)
            d_feature_real = self.discriminator_feature(z).view(-1)
            feature_loss = C(d_feature_real, 0.0) + C(d_feature_fake, 1.0)
            note_loss.backward()
            feature_loss.backward()
            self.nets["discriminator"].step()
            self.nets["discriminator_feature"].step()
            meters["discriminator"].update(note_loss)
            meters["discriminator_feature"].update(feature_loss)
        self.only_trainable("generator")
        noise = self.prior(note.size(0), 1.5)
        gen_note, gen_z = self.generator(noise, pre_note, pre_phrase, position, False)
        loss = C(self.discriminator(self.pair(pre_note, gen_note)).view(-1), 1.0)
        loss = loss + C(self.discriminator_feature(gen_z).view(-1), 1.0)
        loss.backward()
        self.nets["generator"].step()
        meters["generator"].update(loss)
        return gen_note[:3]
