"""Host-side data path of the train / eval loop: offline sharding + negative sampling and the
per-line index selection.  Integer work only, bit-exact with the reference:
the stdlib `random` (MT19937) call sequence is part of the contract, so it is reproduced call for call.

  prepare_training_data / prepare_testing_data  <->  src/prepare_data.py:14-66
  DatasetTrain / DatasetTest / NewsDataset      <->  src/dataset.py:6-89
"""
import os
import random

import numpy as np
from torch.utils.data import Dataset, IterableDataset


def get_sample(all_elements, num_sample):
    """src/prepare_data.py:7-11: sample without replacement; a too-short list is replicated first."""
    if num_sample > len(all_elements):
        return random.sample(all_elements * (num_sample // len(all_elements) + 1), num_sample)
    return random.sample(all_elements, num_sample)


def shard_training_lines(lines, nGPU, npratio, seed):
    """One output line per positive impression with `npratio` sampled negatives, global shuffle, shard i % nGPU
    (src/prepare_data.py:15,31-41).  Returns the per-shard line lists."""
    random.seed(seed)
    behaviors = []
    for line in lines:
        iid, uid, time, history, imp = line.strip().split("\t")
        pos, neg = [], []
        for news_id, label in (x.split("-") for x in imp.split(" ")):
            if label == "0":
                neg.append(news_id)
            elif label == "1":
                pos.append(news_id)
        if not pos or not neg:
            continue
        for pos_id in pos:
            neg_str = " ".join(get_sample(neg, npratio))
            behaviors.append("\t".join([iid, uid, time, history, pos_id, neg_str]) + "\n")
    random.shuffle(behaviors)
    shards = [[] for _ in range(nGPU)]
    for i, line in enumerate(behaviors):
        shards[i % nGPU].append(line)
    return shards


def prepare_training_data(train_data_dir, nGPU, npratio, seed):
    """File form of shard_training_lines: behaviors.tsv -> behaviors_np{npratio}_{rank}.tsv; returns #samples."""
    with open(os.path.join(train_data_dir, "behaviors.tsv"), "r", encoding="utf-8") as f:
        shards = shard_training_lines(f, nGPU, npratio, seed)
    for i, shard in enumerate(shards):
        with open(os.path.join(train_data_dir, f"behaviors_np{npratio}_{i}.tsv"), "w") as f:
            f.writelines(shard)
    return sum(len(s) for s in shards)


def shard_testing_lines(lines, nGPU):
    """Round-robin i % nGPU without shuffling (src/prepare_data.py:52-64)."""
    shards = [[] for _ in range(nGPU)]
    for i, line in enumerate(lines):
        shards[i % nGPU].append(line)
    return shards


def prepare_testing_data(test_data_dir, nGPU):
    with open(os.path.join(test_data_dir, "behaviors.tsv"), "r", encoding="utf-8") as f:
        shards = shard_testing_lines(f, nGPU)
    for i, shard in enumerate(shards):
        with open(os.path.join(test_data_dir, f"behaviors_{i}.tsv"), "w") as f:
            f.writelines(shard)
    return sum(len(s) for s in shards)


def common_batch_count(shard_sizes, batch_size):
    """Batches every rank can run: the reference lets ranks disagree by one batch (latent hang);
    truncating to the common count keeps the gradient all-reduce matched."""
    return min(shard_sizes) // batch_size


class DatasetTrain(IterableDataset):
    """src/dataset.py:6-53.  Yields (history_features [H, F], history_mask [H] f32, candidate_features [1+K, F], label)."""

    def __init__(self, filename, news_index, news_combined, args):
        super().__init__()
        self.filename = filename
        self.news_index = news_index
        self.news_combined = news_combined
        self.args = args

    def trans_to_nindex(self, nids):
        return [self.news_index[i] if i in self.news_index else 0 for i in nids]       # unknown news -> 0

    def pad_to_fix_len(self, x, fix_length, padding_front=True, padding_value=0):
        if padding_front:                                                            # keep the LAST fix_length clicks
            pad_x = [padding_value] * (fix_length - len(x)) + x[-fix_length:]
            mask = [0] * (fix_length - len(x)) + [1] * min(fix_length, len(x))
        else:
            pad_x = x[-fix_length:] + [padding_value] * (fix_length - len(x))
            mask = [1] * min(fix_length, len(x)) + [0] * (fix_length - len(x))
        return pad_x, np.array(mask, dtype="float32")

    def line_to_indices(self, line):
        """The integer part of line_mapper: (history idx [H], mask, sample idx [1+K], label)."""
        line = line.strip().split("\t")
        hist, mask = self.pad_to_fix_len(self.trans_to_nindex(line[3].split()), self.args.user_log_length)
        pos = self.trans_to_nindex(line[4].split())
        neg = self.trans_to_nindex(line[5].split())
        label = random.randint(0, self.args.npratio)                                  # src/dataset.py:45
        return hist, mask, neg[:label] + pos + neg[label:], label

    def line_mapper(self, line):
        hist, mask, sample, label = self.line_to_indices(line)
        return self.news_combined[hist], mask, self.news_combined[sample], label

    def __iter__(self):
        return map(self.line_mapper, open(self.filename))


class DatasetTest(DatasetTrain):
    """src/dataset.py:56-78.  `news_scoring` holds encoded news vectors; yields per-impression variable-length candidates."""

    def __init__(self, filename, news_index, news_scoring, args):
        IterableDataset.__init__(self)
        self.filename = filename
        self.news_index = news_index
        self.news_scoring = news_scoring
        self.args = args

    def line_to_indices(self, line):
        line = line.strip().split("\t")
        hist, mask = self.pad_to_fix_len(self.trans_to_nindex(line[3].split()), self.args.user_log_length)
        cand = self.trans_to_nindex([i.split("-")[0] for i in line[4].split()])
        labels = np.array([int(i.split("-")[1]) for i in line[4].split()])
        return hist, mask, cand, labels

    def line_mapper(self, line):
        hist, mask, cand, labels = self.line_to_indices(line)
        return self.news_scoring[hist], mask, self.news_scoring[cand], labels


class NewsDataset(Dataset):
    """src/dataset.py:81-89."""

    def __init__(self, data):
        self.data = data

    def __getitem__(self, idx):
        return self.data[idx]

    def __len__(self):
        return self.data.shape[0]
