"""On-disk formats either side of the path: the per-news title-embedding matrix and the
checkpoint dict.  Load-compatible with the reference in both directions.

  title_embeddings.{bpemb,bert}.npy.gz   src/preprocess.py:154-158 writes `np.save` of a float [N+1, T*D] matrix through
                                         gzip; :227-239 reads it back whole (gzip + np.load) at every start-up.
  epoch-k.pt                             src/main.py:118-142: {'model_state_dict', 'category_dict', 'subcategory_dict'}
                                         (train.checkpoint_dict / train.load_checkpoint).

The gzip stream of a 1.2-2.3 GB matrix cannot be memory-mapped or read in parallel and dominates start-up, so the first
read inflates it ONCE, in bounded chunks, into a plain `.npy` next to it (or in `cache_dir`) and every later start-up maps
that file: rows are paged in on demand, several ranks on one node share the page cache, and `np.load` never holds a
second full copy.
"""
import gzip
import os

import numpy as np

_NAMES = {"bpemb": "title_embeddings.bpemb.npy.gz", "bert": "title_embeddings.bert.npy.gz"}
_CHUNK = 64 << 20


def write_news_embeddings(data_dir, embeddings, kind="bpemb"):
    """src/preprocess.py:154-158: flatten to [N+1, T*D] and np.save through gzip."""
    arr = np.asarray(embeddings)
    arr = arr.reshape(arr.shape[0], -1)
    path = os.path.join(data_dir, _NAMES[kind])
    with gzip.GzipFile(path, "w") as f:
        np.save(f, arr)
    return path


def _inflate_once(gz_path, npy_path):
    """gzip -> .npy in 64 MB chunks (never the whole matrix in memory); atomic rename so concurrent ranks are safe."""
    tmp = f"{npy_path}.{os.getpid()}.tmp"
    with gzip.GzipFile(gz_path, "r") as src, open(tmp, "wb") as dst:
        while True:
            buf = src.read(_CHUNK)
            if not buf:
                break
            dst.write(buf)
    os.replace(tmp, npy_path)


def read_news_embeddings(data_dir, kind="bpemb", cache_dir=None, mmap=True):
    """src/preprocess.py:227-239 (`read_news_embeddings` / `read_news_embeddings_bert`): the [N+1, T*D] matrix.

    mmap=True (default): a read-only `np.memmap` of the inflated cache file (created on first use, reused while it is
    newer than the `.gz`).  mmap=False: the reference's own way, gzip + np.load, fully in memory.
    The values are identical either way (`allow_pickle` stays off: the file holds a plain array)."""
    gz_path = os.path.join(data_dir, _NAMES[kind])
    if not mmap:
        with gzip.GzipFile(gz_path, "r") as f:
            return np.load(f)
    cache_dir = cache_dir or data_dir
    npy_path = os.path.join(cache_dir, os.path.basename(gz_path)[:-3])
    if not os.path.exists(npy_path) or os.path.getmtime(npy_path) < os.path.getmtime(gz_path):
        os.makedirs(cache_dir, exist_ok=True)
        _inflate_once(gz_path, npy_path)
    return np.load(npy_path, mmap_mode="r")


def npy_header(path):
    """(shape, dtype, data offset) of a .npy file without touching its payload."""
    with open(path, "rb") as f:
        major, minor = np.lib.format.read_magic(f)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(f)
        if fortran:
            raise ValueError(f"{path}: Fortran-ordered arrays are not supported")
        return shape, dtype, f.tell()


def rows_in_blocks(embeddings, rows_per_block=8192):
    """Iterate (first_row, float32 block) over a (memory-mapped) matrix: bounded host memory while a table is uploaded."""
    n = embeddings.shape[0]
    for a in range(0, n, rows_per_block):
        yield a, np.ascontiguousarray(embeddings[a:a + rows_per_block], dtype=np.float32)
