"""Machinery shared by the BarGen agents of the MI355X build.

What the reference's agents do per process with nn.DataParallel / Horovod
(agent/barGen2.py:31-107,128-196; agent/barGen_horovod.py:35-36,49-50,91-134) is done here
with one process per GPU: the device is cuda:LOCAL_RANK, the dataset is sharded by rank
(DistributedSampler semantics), every network's parameters live in one flat buffer with a
fused HIP Adam (hipops.FlatParams), gradients are all-reduced over RCCL in large buckets
(hipops.dist.GradReducer) and only rank 0 writes logs / summaries / checkpoints."""
import logging
import os
import random
import shutil

import numpy as np
import torch
from torch.utils.data import DataLoader

from hipops import FlatParams
from hipops import dist as hdist
from hipops import functional as HF
from hipops.dist import GradReducer


class _NullWriter:
    """tensorboard shim: tensorboardX is not a dependency of this build"""

    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass

    def close(self):
        pass


def make_summary_writer(log_dir, comment=""):
    for mod in ("tensorboardX", "torch.utils.tensorboard"):
        try:
            m = __import__(mod, fromlist=["SummaryWriter"])
            return m.SummaryWriter(log_dir=log_dir, comment=comment)
        except Exception:
            continue
    return _NullWriter()


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, cooldown) restated for
    FlatParams (reference: agent/barGen2.py:67-76; torch defaults patience=10, threshold=1e-4
    relative, min_lr=0, eps=1e-8)."""

    def __init__(self, optimizer, mode="min", factor=0.8, patience=10, cooldown=6, threshold=1e-4, min_lr=0.0, eps=1e-8):
        assert mode == "min"
        self.opt, self.factor, self.patience, self.cooldown = optimizer, factor, patience, cooldown
        self.threshold, self.min_lr, self.eps = threshold, min_lr, eps
        self.best, self.num_bad, self.cooldown_counter = float("inf"), 0, 0

    def step(self, metric):
        cur = float(metric)
        if cur < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = cur, 0
        else:
            self.num_bad += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad = 0
        if self.num_bad > self.patience:
            for g in self.opt.param_groups:
                new = max(g["lr"] * self.factor, self.min_lr)
                if g["lr"] - new > self.eps:
                    g["lr"] = new
            self.cooldown_counter, self.num_bad = self.cooldown, 0

    def state_dict(self):
        return {"best": self.best, "num_bad": self.num_bad, "cooldown_counter": self.cooldown_counter}

    def load_state_dict(self, sd):
        self.best, self.num_bad, self.cooldown_counter = sd["best"], sd["num_bad"], sd["cooldown_counter"]


def collate_batch(samples):
    """the reference's ``make_batch`` (agent/barGen2.py:128-135): a batch is the concatenation of its items along axis 0.
    Module-level so that spawned loader workers can unpickle it (AgentBase.make_loader)."""
    cat = lambda k: np.concatenate([s[k] for s in samples], axis=0)
    if "note_bits" in samples[0]:           # packed: ship the bits, expand them in to_device
        return (torch.from_numpy(cat("note_bits")), torch.from_numpy(cat("pre_note_bits")),
                torch.from_numpy(cat("pre_phrase_bits")), torch.from_numpy(cat("position").astype(np.int64)))
    return (torch.tensor(cat("note"), dtype=torch.float), torch.tensor(cat("pre_note"), dtype=torch.float),
            torch.tensor(cat("pre_phrase"), dtype=torch.float), torch.tensor(cat("position"), dtype=torch.long))


class Net:
    """a network + its flat Adam + its gradient reducer + its LR scheduler"""

    def __init__(self, module, lr, bucket_mb, plateau=True):
        self.module = module
        self.opt = FlatParams(list(module.parameters()), lr=lr)
        hdist.broadcast_flat(self.opt.flat)
        self.reducer = GradReducer(self.opt.grad, int(bucket_mb) * 1024 * 1024 // 4)
        self.scheduler = ReduceLROnPlateau(self.opt, mode="min", factor=0.8, cooldown=6) if plateau else None

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self):
        """all-reduce what has not been reduced yet, then the fused Adam (1/world folded in)"""
        self.reducer.reduce_rest()
        self.reducer.wait()
        self.opt.step(grad_scale=1.0 / hdist.world_size())

    # DataParallel-compatible checkpoint keys ("module." prefix, reference agent/barGen2.py:168-177)
    def state_dict(self):
        return {"module." + k: v.detach().clone() for k, v in self.module.state_dict().items()}

    def load_state_dict(self, sd):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        # parameters are views of the flat buffer and load_state_dict copies in place, so the flat buffer follows.
        # Only the refiner's entries may be absent or surplus (reference checkpoints carry them; this build's default
        # generator has no refiner): anything else is a checkpoint of another layout
        res = self.module.load_state_dict(sd, strict=False)
        bad = [k for k in list(res.missing_keys) + list(res.unexpected_keys) if not k.startswith("refiner.")]
        if bad:
            raise RuntimeError("checkpoint does not match the network: %s" % bad[:5])


class AgentBase(object):
    def __init__(self, config):
        self.config = config
        self.pretraining_step_size = self.config.pretraining_step_size
        self.batch_size = self.config.batch_size
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise RuntimeError("the BarGen agents of this build need a ROCm GPU (no CPU fallback for the HIP hot path)")
        ndev = torch.cuda.device_count()
        if self.local_rank >= ndev:
            # one process per GPU.  Only the tests (gloo ranks stacked on the single GPU of a test box) may share a
            # device, and they say so; under RCCL two ranks on one device is a duplicate-GPU error or a hang
            if os.environ.get("MGVAE_STACK_RANKS", "0") == "0":
                raise RuntimeError("LOCAL_RANK=%d but only %d GPU(s) are visible: one process per GPU "
                                   "(MGVAE_STACK_RANKS=1 lets gloo test ranks share a device)" % (self.local_rank, ndev))
            self.local_rank %= max(1, ndev)
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        if self.world > 1 and not torch.distributed.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
        self.logger = self.set_logger()
        self.iteration = 0
        self.epoch = 0
        # one seed for the whole job: with config.seed = None rank 0 draws it (reference agent/barGen2.py:53) and
        # broadcasts it, so python's ``random`` (agent/barGen.py's per-epoch div_flag) and torch's host RNG agree on
        # every rank and all ranks take the same branches, i.e. issue the same collectives in the same order
        self.manual_seed = hdist.shared_seed(getattr(self.config, "seed", None))

    # ---------------------------------------------------------------- reference surface
    def get_lr(self, optimizer):
        for g in optimizer.param_groups:
            return g["lr"]

    def set_logger(self):
        logger = logging.getLogger()
        logger.setLevel(logging.DEBUG)
        path = os.path.abspath(getattr(self.config, "log_file", "train_epoch.log"))
        if self.rank == 0 and not any(getattr(h, "_mgvae_path", None) == path for h in logger.handlers):
            fh = logging.FileHandler(filename=path)
            fh._mgvae_path = path
            fh.setLevel(logging.WARNING)
            fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
            logger.addHandler(fh)
        return logger

    def make_dataset(self):
        """NoteDataset (the reference's per-sample npz files) or, with ``config.packed_data_file``, the bit-packed one"""
        if getattr(self.config, "packed_data_file", None):
            from data.bar_dataset import PackedNoteDataset
            return PackedNoteDataset(self.config.root_path, self.config)
        from data.bar_dataset import NoteDataset
        return NoteDataset(self.config.root_path, self.config)

    def make_batch(self, samples):
        return collate_batch(samples)

    def free(self, module):
        for p in module.parameters():
            p.requires_grad = True

    def frozen(self, module):
        for p in module.parameters():
            p.requires_grad = False

    def seed_everything(self):
        torch.manual_seed(self.manual_seed)
        torch.cuda.manual_seed_all(self.manual_seed)
        random.seed(self.manual_seed)
        HF.manual_seed(self.manual_seed, self.rank)
        HF.set_compute_dtype(getattr(self.config, "compute_dtype", "f32"))
        if self.rank == 0:
            print("seed: ", self.manual_seed)

    def make_loader(self, dataset):
        sampler = None
        if self.world > 1:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(dataset, num_replicas=self.world, rank=self.rank, shuffle=False)
        # The reference runs ONE loader worker (agent/barGen2.py:41), and so does this build by default.  What differs is
        # how the worker comes to life.  torch's default on Linux is fork(): the child of a process that has initialised
        # the GPU runtime inherits libhsa-runtime64 / libamdhip64 state (queue and doorbell mappings, signal pools, the
        # runtimes' atexit and static destructors) WITHOUT the runtimes' helper threads, and it runs those destructors
        # when it exits at the end of an epoch -- that is the `DataLoader worker (pid N) is killed by signal:
        # Segmentation fault` round 2 recorded (the parent had initialised HIP, dlopen'ed libmgvae_hip.so and held pinned
        # host rings when it forked).  So no worker is ever forked from this process: workers are SPAWNED (a fresh
        # interpreter that never touches the GPU; the dataset and the module-level ``collate_batch`` travel by pickle)
        # and kept alive across epochs (``persistent_workers``: one start-up per run instead of one fork per epoch).
        # config.num_workers = 0 loads in-process.
        workers = int(getattr(self.config, "num_workers", 1))
        extra = {}
        if workers > 0:
            extra = dict(multiprocessing_context="spawn", persistent_workers=True)
        return DataLoader(dataset, batch_size=self.batch_size, shuffle=False, num_workers=workers, sampler=sampler,
                          pin_memory=self.config.pin_memory, collate_fn=collate_batch, **extra)

    def close_loader(self):
        """stop the persistent loader workers now (they would otherwise be stopped when the loader is collected)"""
        it = getattr(getattr(self, "dataloader", None), "_iterator", None)
        if it is not None and hasattr(it, "_shutdown_workers"):
            it._shutdown_workers()
            self.dataloader._iterator = None

    def to_device(self, *tensors):
        nb = bool(self.config.async_loading)
        out = tuple(t.to(self.device, non_blocking=nb) for t in tensors)
        if out[0].dtype == torch.uint8:         # bit-packed rolls: expand to fp32 on the device
            b = out[0].shape[0]
            out = (HF.unpack_bits(out[0], (b, 1, 96, 60)), HF.unpack_bits(out[1], (b, 1, 96, 60)),
                   HF.unpack_bits(out[2], (b, 1, 384, 60)), out[3])
        return out

    def run(self):
        try:
            self.train()
        except KeyboardInterrupt:
            print("You have entered CTRL+C.. Wait to finalize")

    # ---------------------------------------------------------------- checkpoints
    def _ckpt_path(self, name):
        return os.path.join(self.config.root_path, self.config.checkpoint_dir, name)

    def _save(self, state, epoch):
        if self.rank != 0:
            return
        os.makedirs(self._ckpt_path(""), exist_ok=True)
        tmp = self._ckpt_path("checkpoint_{}.pth.tar".format(epoch))
        torch.save(state, tmp)
        shutil.copyfile(tmp, self._ckpt_path("checkpoint.pth.tar"))

    def _load(self, file_name):
        filename = self._ckpt_path(file_name)
        try:
            print("Loading checkpoint '{}'".format(filename))
            return torch.load(filename, map_location=self.device, weights_only=False)
        except OSError:
            print("No checkpoint exists from '{}'. Skipping...".format(self.config.checkpoint_dir))
            print("**First time to train**")
            return None

    # ---------------------------------------------------------------- sampling (reference agent/barGen2.py:317-336)
    def sample_phrases(self, generator, n_phrases=10):
        """autoregressive bar-by-bar sampling: 4 bars per phrase, each bar conditioned on the
        previous binarised bar and the previous 4-bar phrase; returns n_phrases arrays [384, 60]"""
        outputs = []
        pre_phrase = torch.zeros(1, 1, 384, 60, device=self.device)
        pre_bar = torch.zeros(1, 1, 96, 60, device=self.device)
        phrase_idx = [330] + [i for i in range(n_phrases - 2, -1, -1)]
        with torch.no_grad():
            for idx in range(n_phrases):
                bars = []
                pos = torch.tensor([phrase_idx[idx]], device=self.device, dtype=torch.long)
                for _ in range(4):
                    out = generator(HF.randn((1, 1152), 1.0, self.device), pre_bar, pre_phrase, pos, False)
                    out = out[0] if isinstance(out, tuple) else out
                    pre_bar = torch.gt(out, 0.3).float()
                    bars.append(pre_bar.reshape(96, 60))
                pre_phrase = torch.cat(bars, dim=0).reshape(1, 1, 384, 60)
                outputs.append(pre_phrase.reshape(384, 60).cpu().numpy())
        return outputs
