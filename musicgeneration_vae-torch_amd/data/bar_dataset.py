"""Piano-roll bar dataset (reference: data/bar_dataset.py:9-25): one ``*.npz`` per entry under
``root_dir/config.data_path`` with keys note, pre_note [k,1,96,60], pre_phrase [k,1,384,60] and
position [k]; a batch is the concatenation of its items along axis 0 (agent ``make_batch``)."""
import os
import types

import numpy as np
from torch.utils.data import Dataset


def _plain_config(config):
    """the scalar settings of a Config object as a plain namespace.  Loader workers of this build are spawned, not forked
    (agent/base.py::make_loader), so the dataset travels to them by pickle -- and a Config subclass defined inside a
    function, as experiments and tests do, does not pickle."""
    out = types.SimpleNamespace()
    for k in dir(config):
        if not k.startswith("_"):
            v = getattr(config, k)
            if isinstance(v, (int, float, str, bool, type(None))):
                setattr(out, k, v)
    return out


class _Picklable(Dataset):
    def __getstate__(self):
        state = dict(self.__dict__)
        state["config"] = _plain_config(state["config"])
        return state


class NoteDataset(_Picklable):
    def __init__(self, root_dir, config):
        self.root_dir = root_dir
        self.config = config
        self.file_list = sorted(os.listdir(os.path.join(self.root_dir, config.data_path)))
        self.num_iterations = (len(self.file_list) + config.batch_size - 1) // config.batch_size

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        path = os.path.join(self.root_dir, self.config.data_path, self.file_list[idx])
        with np.load(path) as data:
            return {k: data[k] for k in ('note', 'pre_note', 'pre_phrase', 'position')}


class TestDataset(NoteDataset):
    """reference data/bar_dataset.py:27-43: entries 100..199 of the listing"""

    def __init__(self, root_dir, config):
        super().__init__(root_dir, config)
        self.file_list = self.file_list[100:200]
        self.num_iterations = (len(self.file_list) + config.batch_size - 1) // config.batch_size


# ---------------------------------------------------------------------------------------------------------
# Bit-packed variant (MI355X build).  The reference's layout -- one npz of fp32 rolls per sample,
# one loader worker -- ships 138 KB per sample and cannot feed 8 GPUs (>= 10^4 bars/s = 1.4 GB/s of fp32).
# Rolls are {0,1}, so one packed file holds, per sample, 720 + 720 + 2880 bytes and an int16 position: 32x less
# to read, pin and copy; the device expands the bits (hipops.functional.unpack_bits).
PACKED_KEYS = ("note_bits", "pre_note_bits", "pre_phrase_bits", "position")


def pack_dataset(src_dir, dst_file):
    """convert a directory of the reference's per-sample npz files into one packed npz; returns the sample count"""
    note, pre, phrase, pos = [], [], [], []
    for name in sorted(os.listdir(src_dir)):
        with np.load(os.path.join(src_dir, name)) as d:
            for k, dst, cells in (("note", note, 5760), ("pre_note", pre, 5760), ("pre_phrase", phrase, 23040)):
                a = np.asarray(d[k])
                if not np.isin(a, (0, 1)).all():
                    raise ValueError("%s[%s] is not a {0,1} roll: cannot bit-pack" % (name, k))
                dst.append(np.packbits(a.reshape(a.shape[0], cells).astype(np.uint8), axis=1, bitorder="little"))
            pos.append(np.asarray(d["position"]).astype(np.int16).reshape(-1))
    np.savez(dst_file, note_bits=np.concatenate(note), pre_note_bits=np.concatenate(pre),
             pre_phrase_bits=np.concatenate(phrase), position=np.concatenate(pos))
    return int(sum(len(p) for p in pos))


class PackedNoteDataset(_Picklable):
    """same role as NoteDataset over ONE packed file (``config.packed_data_file`` under ``root_dir``, made by
    ``pack_dataset``); an item is ONE sample (dict of uint8 rows + position), so ``batch_size`` counts samples here,
    whereas NoteDataset's items are files that may hold several samples each"""

    def __init__(self, root_dir, config):
        self.root_dir, self.config = root_dir, config
        with np.load(os.path.join(root_dir, config.packed_data_file)) as d:
            self.arrays = {k: np.ascontiguousarray(d[k]) for k in PACKED_KEYS}
        n = len(self.arrays["position"])
        if self.arrays["note_bits"].shape != (n, 720) or self.arrays["pre_phrase_bits"].shape != (n, 2880):
            raise ValueError("packed dataset has unexpected shapes")
        self.num_iterations = (n + config.batch_size - 1) // config.batch_size

    def __len__(self):
        return len(self.arrays["position"])

    def __getitem__(self, idx):
        return {k: v[idx:idx + 1] for k, v in self.arrays.items()}
