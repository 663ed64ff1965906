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

    def pad_to_fix_len(self, x, fix_length):
        """Front padding (the only mode the reference uses, src/dataset.py:17-20): keep the LAST fix_length clicks."""
        pad_x = [0] * (fix_length - len(x)) + x[-fix_length:]
        mask = [0] * (fix_length - len(x)) + [1] * min(fix_length, len(x))
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


class IndexedTrainShard:
    """Row f1 of SURVEY §8: the integer part of `DatasetTrain` for a WHOLE shard file, parsed once.

    The reference re-parses the shard line by line every epoch inside the training loop (src/dataset.py:26-53, ~10^4
    lines/s in Python) and ships [H, F] / [1+K, F] feature blocks per sample.  Here the shard becomes four index arrays
    that are uploaded once; per epoch only the label positions are drawn -- `random.randint(0, npratio)` per line in
    file order, the reference's own RNG call (src/dataset.py:45), so the (label, sample_news) stream is bit-identical
    to iterating `DatasetTrain` -- and `ops.assemble_batch` gathers the feature rows on the device.

      hist  [n, H] int32  news indices, front padded (src/dataset.py:17-24,40)      mask [n, H] float32
      pos   [n]    int32  the clicked news                                           neg  [n, K] int32
    """

    def __init__(self, filename, news_index, args):
        H, K = args.user_log_length, args.npratio
        tr = DatasetTrain(filename, news_index, None, args)
        hist, mask, pos, neg = [], [], [], []
        with open(filename) as f:
            for ln, line in enumerate(f):
                fld = line.strip().split("\t")
                h, m = tr.pad_to_fix_len(tr.trans_to_nindex(fld[3].split()), H)
                p, ng = tr.trans_to_nindex(fld[4].split()), tr.trans_to_nindex(fld[5].split())
                if len(p) != 1 or len(ng) != K:
                    raise ValueError(f"{filename}:{ln + 1}: expected 1 positive and {K} negatives per line "
                                     f"(src/prepare_data.py:31-35 writes such lines), got {len(p)} / {len(ng)}")
                hist.append(h); mask.append(m); pos.append(p[0]); neg.append(ng)
        n = len(pos)
        self.hist = np.asarray(hist, dtype=np.int32).reshape(n, H)
        self.mask = np.asarray(mask, dtype=np.float32).reshape(n, H)
        self.pos = np.asarray(pos, dtype=np.int32)
        self.neg = np.asarray(neg, dtype=np.int32).reshape(n, K)
        self.npratio = K

    def __len__(self):
        return self.pos.shape[0]

    def draw_labels(self):
        """One epoch's label positions, in file order: the `random.randint(0, npratio)` stream of src/dataset.py:45."""
        K = self.npratio
        return np.fromiter((random.randint(0, K) for _ in range(len(self))), dtype=np.int64, count=len(self))


class IndexedTestShard:
    """The integer part of `DatasetTest` (src/dataset.py:64-74) for a whole shard: history indices / mask per impression and
    the variable-length candidate lists in CSR form (cand, label, offsets)."""

    def __init__(self, filename, news_index, args):
        H = args.user_log_length
        tr = DatasetTest(filename, news_index, None, args)
        hist, mask, cand, lab, off = [], [], [], [], [0]
        with open(filename) as f:
            for line in f:
                h, m, c, l = tr.line_to_indices(line)
                hist.append(h); mask.append(m); cand += c; lab += l.tolist(); off.append(len(cand))
        n = len(hist)
        self.hist = np.asarray(hist, dtype=np.int32).reshape(n, H)
        self.mask = np.asarray(mask, dtype=np.float32).reshape(n, H)
        self.cand = np.asarray(cand, dtype=np.int32)
        self.label = np.asarray(lab, dtype=np.int32)
        self.offsets = np.asarray(off, dtype=np.int32)

    def __len__(self):
        return self.hist.shape[0]


class NewsDataset(Dataset):
    """src/dataset.py:81-89."""

    def __init__(self, data):
        self.data = data

    def __getitem__(self, idx):
        return self.data[idx]

    def __len__(self):
        return self.data.shape[0]
