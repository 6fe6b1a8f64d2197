"""CPU: the bit-packed input pipeline's host side (data/bar_dataset.py::pack_dataset / PackedNoteDataset and the
agents' make_batch) reproduces the reference-format dataset bit for bit."""
import os
import sys

import numpy as np
import pytest

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "musicgeneration_vae-torch_amd"))


def _make(root, n_files=5, per_file=2, seed=3):
    d = os.path.join(root, "data", "dataset")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(seed)
    for i in range(n_files):
        np.savez(os.path.join(d, "bar_%03d.npz" % i),
                 note=(rng.random((per_file, 1, 96, 60)) < 0.05).astype(np.float32),
                 pre_note=(rng.random((per_file, 1, 96, 60)) < 0.05).astype(np.float32),
                 pre_phrase=(rng.random((per_file, 1, 384, 60)) < 0.05).astype(np.float32),
                 position=rng.integers(0, 332, size=(per_file,)))
    return d


def test_pack_dataset_round_trip(tmp_path):
    from data.bar_dataset import NoteDataset, PackedNoteDataset, pack_dataset
    root = str(tmp_path)
    src = _make(root)

    class Cfg:
        data_path = "data/dataset"
        packed_data_file = "data/packed.npz"
        batch_size = 4

    n = pack_dataset(src, os.path.join(root, Cfg.packed_data_file))
    assert n == 10
    plain, packed = NoteDataset(root, Cfg), PackedNoteDataset(root, Cfg)
    assert len(packed) == 10 and packed.num_iterations == 3
    k = 0
    for i in range(len(plain)):
        item = plain[i]
        for j in range(item["note"].shape[0]):
            p = packed[k]
            for key, bits, cells in (("note", "note_bits", 5760), ("pre_note", "pre_note_bits", 5760), ("pre_phrase", "pre_phrase_bits", 23040)):
                got = np.unpackbits(p[bits][0], bitorder="little")[:cells].astype(np.float32)
                assert np.array_equal(got, item[key][j].reshape(-1)), (i, j, key)
            assert int(p["position"][0]) == int(item["position"][j])
            k += 1
    # 32x smaller than the fp32 rolls
    assert os.path.getsize(os.path.join(root, Cfg.packed_data_file)) * 20 < 10 * (5760 * 2 + 23040) * 4


def test_pack_dataset_rejects_non_binary(tmp_path):
    from data.bar_dataset import pack_dataset
    d = os.path.join(str(tmp_path), "src"); os.makedirs(d)
    np.savez(os.path.join(d, "a.npz"), note=np.full((1, 1, 96, 60), 0.5, np.float32), pre_note=np.zeros((1, 1, 96, 60), np.float32),
             pre_phrase=np.zeros((1, 1, 384, 60), np.float32), position=np.zeros((1,), np.int64))
    with pytest.raises(ValueError):
        pack_dataset(d, os.path.join(str(tmp_path), "p.npz"))
