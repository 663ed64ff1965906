"""On-disk formats either side of the path: the per-news title-embedding matrix and the
checkpoint dict.  Load-compatible with the reference in both directions.

  title_embeddings.{bpemb,bert}.npy.gz   src/preprocess.py:154-158 writes `np.save` of a float [N+1, T*D] matrix through
                                         gzip; :227-239 reads it back whole (gzip + np.load) at every start-up.
  epoch-k.pt                             src/main.py:118-142: {'model_state_dict', 'category_dict', 'subcategory_dict'}
                                         (train.checkpoint_dict / train.load_checkpoint).

The gzip stream of a 1.2-2.3 GB matrix cannot be memory-mapped or read in parallel and dominates start-up, so the first
read inflates it ONCE, in bounded chunks, into a plain `.npy` next to it (or in `cache_dir`) and every later start-up maps
that file: rows are paged in on demand, several ranks on one node share the page cache, and `np.load` never holds a
second full copy.  `model.NAML.TitleTable` uploads such a map block by block (`rows_in_blocks`) straight into the packed
compute-dtype table on the device: no full fp32 copy on the host or in HBM.

bf16 shards (`write_bf16_shards` / `read_bf16_shards`): the same matrix as the bf16 GEMM operand the kernels read --
token rows [rows*T, Dp] (D zero-padded to Dp, round-to-nearest-even as `v_cvt_pk_bf16_f32`), a few hundred MB per `.npy`
shard plus a `meta.json`.  Half the bytes on disk, no conversion at load (a shard is copied to the device as it lies), and
the ranks of a node map the same files.  A derived serving format: the fp32 `.npy.gz` stays the master (a checkpoint written
from a shard-loaded model holds the bf16 values widened to fp32).
"""
import gzip
import json
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
        yield a, np.array(embeddings[a:a + rows_per_block], dtype=np.float32, order="C", copy=True)      # (a private, writable block)


# ---------------------------------------------------------------------------------------------- bf16 shards
_SHARD_DIR = {"bpemb": "title_embeddings.bpemb.bf16", "bert": "title_embeddings.bert.bf16"}


def f32_to_bf16_bits(x):
    """fp32 -> bf16 bit patterns (uint16), round to nearest even -- what the device cast (`v_cvt_pk_bf16_f32`) gives for
    every finite value."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def padded_width(D, multiple=32):
    """Leading dimension of the bf16 GEMM operand for D feature columns (ops.pack: whole 32-deep k-steps)."""
    return (int(D) + multiple - 1) // multiple * multiple


class Bf16Shards:
    """The [rows, T*D] title-embedding matrix stored as bf16 token rows [rows*T, Dp] in `.npy` shards (memory-mapped)."""

    def __init__(self, path):
        with open(os.path.join(path, "meta.json")) as f:
            m = json.load(f)
        self.path, self.rows, self.T, self.D, self.Dp = path, int(m["rows"]), int(m["T"]), int(m["D"]), int(m["Dp"])
        self.files = [(int(a), int(n), os.path.join(path, name)) for a, n, name in m["shards"]]
        self.shape = (self.rows, self.T * self.D)
        self.dtype = np.dtype(np.uint16)

    def blocks(self):
        """(first news row, uint16 map [n*T, Dp]) per shard."""
        for a, n, name in self.files:
            blk = np.load(name, mmap_mode="r")
            if blk.shape != (n * self.T, self.Dp) or blk.dtype != np.uint16:
                raise ValueError(f"{name}: expected uint16 [{n * self.T}, {self.Dp}], found {blk.dtype} {blk.shape}")
            yield a, blk

    def to_float32(self):
        """The matrix back as fp32 [rows, T*D] (bf16 values widened): for checkpoints and tests, not for the hot path."""
        out = np.empty((self.rows, self.T, self.D), dtype=np.float32)
        for a, blk in self.blocks():
            n = blk.shape[0] // self.T
            wide = (np.asarray(blk[:, :self.D]).astype(np.uint32) << np.uint32(16)).view(np.float32)
            out[a:a + n] = wide.reshape(n, self.T, self.D)
        return out.reshape(self.rows, self.T * self.D)


def write_bf16_shards(data_dir, embeddings, D, kind="bpemb", rows_per_shard=8192):
    """Write the [rows, T*D] float matrix (array or memory map) as bf16 shards under `data_dir`; returns the shard directory.
    Host memory: one shard at a time."""
    rows, width = embeddings.shape
    D = int(D)
    if width % D:
        raise ValueError(f"row length {width} is not a multiple of the word dimension {D}")
    T, Dp = width // D, padded_width(D)
    path = os.path.join(data_dir, _SHARD_DIR[kind])
    os.makedirs(path, exist_ok=True)
    shards = []
    for k, (a, blk) in enumerate(rows_in_blocks(embeddings, rows_per_shard)):
        n = blk.shape[0]
        out = np.zeros((n * T, Dp), dtype=np.uint16)
        out[:, :D] = f32_to_bf16_bits(blk).reshape(n * T, D)
        name = f"shard-{k:05d}.npy"
        np.save(os.path.join(path, name), out)
        shards.append([a, n, name])
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"rows": rows, "T": T, "D": D, "Dp": Dp, "dtype": "bfloat16", "shards": shards}, f)
    return path


def read_bf16_shards(data_dir, kind="bpemb"):
    return Bf16Shards(os.path.join(data_dir, _SHARD_DIR[kind]))


def load_title_table(data_dir, kind="bpemb", prefer_bf16_shards=False, cache_dir=None):
    """What `embedding_matrix = read_news_embeddings(args.data_dir)` (src/main.py:62) should hand to `NAML.Model`: the bf16
    shards when they exist and are asked for, else the memory map of the inflated `.npy.gz`."""
    if prefer_bf16_shards and os.path.exists(os.path.join(data_dir, _SHARD_DIR[kind], "meta.json")):
        return read_bf16_shards(data_dir, kind)
    return read_news_embeddings(data_dir, kind, cache_dir=cache_dir)
