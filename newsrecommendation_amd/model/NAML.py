"""Drop-in for the reference's src/model/NAML.py (fork variant: per-news flattened title embedding)."""
import warnings

import numpy as np
import torch
from torch import nn

from .. import formats, ops
from .model_utils import AttentionPooling


def _cd(args):
    return getattr(args, "compute_dtype", "fp32")


class TitleTable(nn.Module):
    """The FROZEN per-news title-embedding table of src/model/NAML.py:104-107
    (`nn.Embedding.from_pretrained(..., freeze=True, padding_idx=0)`) held on the device ONLY as the GEMM operand the
    kernels read -- token rows [rows*T, Dp] in the compute dtype -- and filled block by block from its host source (SURVEY §8
    row f3): a numpy array, the `np.memmap` of `formats.read_news_embeddings`, or `formats.Bf16Shards`.  No full fp32 copy
    is made on the host or in HBM (the reference's `torch.from_numpy(w).float()` + `.cuda()` + a compute-dtype pack would
    hold 2.3 + 2.3 + 1.2 GB for 65 001 news x 30 tokens; this holds the 1.25 GB bf16 operand and one 150 MB block in flight).

    Checkpoint surface unchanged: `state_dict()` carries `weight` [rows, T*D] fp32 (read from the host source when a
    checkpoint is written), `load_state_dict` accepts it and re-uploads."""

    BLOCK_ROWS = 4096

    def __init__(self, source, D, compute_dtype):
        super().__init__()
        if len(source.shape) != 2 or source.shape[1] % int(D):
            raise ValueError(f"title table must be [rows, T*{D}], got {tuple(source.shape)}")
        self.source, self.D, self.compute_dtype = source, int(D), compute_dtype
        self.rows, self.T = int(source.shape[0]), int(source.shape[1]) // int(D)
        self._packed, self._device = None, None
        self.requires_grad = False
        self.shape = (self.rows, self.T * self.D)

    # -- device residency follows the module (model.cuda() / .to(device), src/main.py:79)
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        dev = fn(torch.empty(0)).device
        if dev != self._device:
            self._device, self._packed = dev, None
            if dev.type == "cuda":
                self._upload()
        return self

    def _upload(self):
        code = ops.dtype_code(self.compute_dtype)
        td, dev, T, D = ops.torch_dtype(code), self._device, self.T, self.D
        Dp = formats.padded_width(D) if code == ops.NR_BF16 else ops.round_up(D, ops.chunk(code))
        packed = torch.empty(self.rows * T, Dp, dtype=td, device=dev)
        with torch.cuda.device(dev):
            if isinstance(self.source, formats.Bf16Shards):
                src = self.source
                if code != ops.NR_BF16 or (src.T, src.D, src.Dp) != (T, D, Dp):
                    raise RuntimeError("bf16 shards hold the bf16 operand layout: compute_dtype must be 'bf16' and the shapes must match")
                for a, blk in src.blocks():                       # as it lies on disk: one H2D copy per shard, no kernel
                    packed[a * T:a * T + blk.shape[0]].copy_(torch.from_numpy(np.ascontiguousarray(blk)).view(torch.bfloat16))
            else:
                for a, blk in formats.rows_in_blocks(self.source, self.BLOCK_ROWS):
                    x = torch.from_numpy(blk).to(dev).view(-1, D)                     # [n*T, D] fp32, one block
                    dst = packed[a * T:a * T + x.shape[0]]
                    if code == ops.NR_F32 and Dp == D:
                        dst.copy_(x)
                    else:
                        ops.check(ops._lib.lib().nr_cast_pad(ops.ptr(x), x.shape[0], D, D, ops.ptr(dst), Dp, code, 0, ops._stream()),
                                  "nr_cast_pad")
                    del x
        self._packed = packed

    def packed(self, code):
        """[rows*T, Dp] operand in the compute dtype `code`."""
        if self._packed is None:
            raise RuntimeError("TitleTable: the table lives on the GPU only (libnrhip has no CPU fallback): move the model to a "
                               "cuda device first")
        if ops.torch_dtype(code) != self._packed.dtype:
            raise RuntimeError(f"TitleTable was packed as {self._packed.dtype}, asked for {ops.torch_dtype(code)}")
        return self._packed

    # -- state_dict surface: key `weight`, fp32 [rows, T*D]
    def host_float32(self):
        if isinstance(self.source, formats.Bf16Shards):
            return torch.from_numpy(self.source.to_float32())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                       # (a read-only memory map: torch warns, nothing writes to it)
            t = torch.from_numpy(np.asarray(self.source))
        return t if t.dtype == torch.float32 else t.float()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + "weight"] = self.host_float32()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key = prefix + "weight"
        if key not in state_dict:
            if strict:
                missing_keys.append(key)
            return
        w = state_dict[key]
        if tuple(w.shape) != self.shape:
            error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(w.shape)} from checkpoint, "
                              f"the shape in current model is {self.shape}.")
            return
        self.source = w.detach().float().cpu().numpy()
        if self._device is not None and self._device.type == "cuda":
            self._upload()


class NewsEncoder(nn.Module):
    """src/model/NAML.py:8-75: title row gather -> dropout -> Conv1d(k=3) -> pooling; category / subcategory
    Embedding -> Linear; 3-view additive pooling."""

    def __init__(self, args, embedding_matrix, num_category, num_subcategory):
        super().__init__()
        self.drop_rate = args.drop_rate
        self.num_words_title = args.num_words_title
        self.word_embedding_dim = args.word_embedding_dim
        self.use_category = args.use_category
        self.use_subcategory = args.use_subcategory
        self.compute_dtype = _cd(args)
        self.title_embeddings = embedding_matrix
        if args.use_category:
            self.category_emb = nn.Embedding(num_category + 1, args.category_emb_dim, padding_idx=0)
            self.category_dense = nn.Linear(args.category_emb_dim, args.news_dim)
        if args.use_subcategory:
            self.subcategory_emb = nn.Embedding(num_subcategory + 1, args.category_emb_dim, padding_idx=0)
            self.subcategory_dense = nn.Linear(args.category_emb_dim, args.news_dim)
        if args.use_category or args.use_subcategory:
            self.final_attn = AttentionPooling(args.news_dim, args.news_query_vector_dim, compute_dtype=_cd(args))
        self.cnn = nn.Conv1d(in_channels=args.word_embedding_dim, out_channels=args.news_dim, kernel_size=3, padding=1)
        self.attn = AttentionPooling(args.news_dim, args.news_query_vector_dim, compute_dtype=_cd(args))

    def forward(self, x, mask=None, needed=None):
        """x: [n, F] int32, columns = [news id, category id, subcategory id][:F] -> [n, news_dim] fp32.
        needed (beyond the reference): optional [n] flags; 0 = the caller multiplies this news vector by zero (a masked history
        slot): its title view is not encoded (zeros)."""
        code = ops.dtype_code(self.compute_dtype)
        if x.dtype != torch.int32:
            x = x.to(torch.int32)
        x = x.contiguous()
        p = self.drop_rate if self.training else 0.0
        needed = ops.needed_flags(needed)
        table = self.title_embeddings if isinstance(self.title_embeddings, TitleTable) else self.title_embeddings.weight
        ctx = ops.conv1d_k3_gather(table, self.cnn.weight, self.cnn.bias, x[:, 0],
                                   self.num_words_title, self.word_embedding_dim, code, p_in=p, needed=needed)
        all_vecs = [self.attn(ctx, mask, needed=needed)]
        col = 1
        if self.use_category:
            all_vecs.append(ops.gather_linear(self.category_emb.weight, self.category_dense.weight,
                                              self.category_dense.bias, x[:, col], code))
            col += 1
        if self.use_subcategory:
            all_vecs.append(ops.gather_linear(self.subcategory_emb.weight, self.subcategory_dense.weight,
                                              self.subcategory_dense.bias, x[:, col], code))
        if len(all_vecs) == 1:
            return all_vecs[0]
        return self.final_attn(torch.stack(all_vecs, dim=1))


class UserEncoder(nn.Module):
    """src/model/NAML.py:78-97."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.attn = AttentionPooling(args.news_dim, args.user_query_vector_dim, compute_dtype=_cd(args))
        self.pad_doc = nn.Parameter(torch.empty(1, args.news_dim).uniform_(-1, 1)).type(torch.FloatTensor)

    def forward(self, news_vecs, log_mask=None):
        code = ops.dtype_code(_cd(self.args))
        if self.args.user_log_mask:
            x = news_vecs if news_vecs.dtype == ops.torch_dtype(code) else ops.to_compute(news_vecs.float(), code)
            return self.attn(x, log_mask)
        return self.attn(ops.pad_blend(news_vecs, log_mask, self.pad_doc, code))


class Model(torch.nn.Module):
    """src/model/NAML.py:100-130."""

    def __init__(self, args, news_embeddings_weight, num_category, num_subcategory, **kwargs):
        super().__init__()
        self.args = args
        if args.freeze_embedding and getattr(args, "stream_title_table", True):
            # frozen (src/demo.sh:12): the table is uploaded block by block into the compute-dtype operand (row f3)
            news_embedding = TitleTable(news_embeddings_weight, args.word_embedding_dim, _cd(args))
        else:
            pretrained_embedding = torch.from_numpy(np.asarray(news_embeddings_weight)).float()
            news_embedding = nn.Embedding.from_pretrained(pretrained_embedding, freeze=args.freeze_embedding, padding_idx=0)
        self.news_encoder = NewsEncoder(args, news_embedding, num_category, num_subcategory)
        self.user_encoder = UserEncoder(args)
        self.loss_fn = nn.CrossEntropyLoss()

    def forward(self, history, history_mask, candidate, label):
        """history [B, H, F] int32; history_mask [B, H]; candidate [B, 1+K, F] int32; label [B] int64."""
        a = self.args
        F = history.shape[-1]
        B, C = candidate.shape[0], 1 + a.npratio
        # masked history slots reach the loss through a factor 0 (NAML.py:92-96): the encoder is told (flags), one launch
        want = not getattr(a, "encode_masked_slots", False)
        if candidate.is_cuda:
            ids, needed = ops.stack_rows(candidate.reshape(-1, F), history.reshape(-1, F), history_mask, flags=want)
        else:
            ids = torch.cat([candidate.reshape(-1, F), history.reshape(-1, F)], dim=0)
            needed = torch.cat([history_mask.new_ones(B * C), history_mask.reshape(-1)]) if want else None
        vecs = self.news_encoder(ids, needed=needed)
        cand_flat, hist_flat = ops.split_rows(vecs, B * C)             # its backward is not even a concatenation (see ops)
        cand_vecs = cand_flat.reshape(B, C, a.news_dim)
        hist_vecs = hist_flat.reshape(B, a.user_log_length, a.news_dim)
        user_vec = self.user_encoder(hist_vecs, history_mask)
        loss, score = ops.score_ce(cand_vecs, user_vec, label)
        return loss, score
