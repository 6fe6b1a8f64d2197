"""Piano-roll bar dataset (reference: data/bar_dataset.py:9-25): one ``*.npz`` per entry under
``root_dir/config.data_path`` with keys note, pre_note [k,1,96,60], pre_phrase [k,1,384,60] and
position [k]; a batch is the concatenation of its items along axis 0 (agent ``make_batch``)."""
import os

import numpy as np
from torch.utils.data import Dataset


class NoteDataset(Dataset):
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
