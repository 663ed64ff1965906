#!/usr/bin/env python3
"""Headline benchmark: NRMS train impressions/sec at per-GPU batch 512 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of the reference loop (src/main.py:98-110) on one resident synthetic MIND-shaped
batch: forward (dropout 0.2 on, as `model.train()`), loss, backward (incl. the word-embedding table gradient:
`freeze_embedding` defaults to False, src/parameters.py:47), gradient all-reduce-mean when N > 1, Adam step.
Weak scaling: every rank owns its own 512-impression shard per step (impressions are independent).

Other workloads the same script measures (each prints ONE JSON line of the same shape):
    --dtype fp32          the exact-fp32 mode that is held to 1e-4 against the CPU oracle
    --model NAML          BASELINE configs[2] (3 views, frozen [N+1, T*D] title table)
    --dense-batch         full histories and full-length titles (no padding for the sparsity shortcuts to use)
    --feed device         batches assembled on the device from news-index arrays every step
    --dp-mode ddp         DistributedDataParallel + torch.optim.Adam (the reference's objects) instead of the flat bucket
    --eval                BASELINE configs[4] per GPU: encode --eval-news news once, score --eval-impressions impressions,
                          ranking metrics on the device (a "step" is one pass over the impression shard)

The JSON line also carries
  roofline     - the dominant libnrhip kernel of the timed region, timed live with HIP events inside the library
                 (nr_prof_*), against the MI355X dense-MFMA / HBM peak; "step": the whole step against the binding (MFMA)
                 bound and its HBM bytes (PMC file) against the fused-minimum bytes of SURVEY §8(d);
  cpu_baseline - the CPU oracle (a port, torch-CPU fp32) timed on this host on a bounded sample.
"""
import argparse
import glob
import json
import os
import re
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_BF16 = 2500.0   # TFLOP/s dense (MI355X_MICROARCH.md, Chip-level parameters)
PEAK_MFMA_F32 = 157.3
PEAK_HBM = 8000.0         # GB/s

# algorithmic forward FLOPs per impression (55 titles + user encoder + scorer) and the fused-minimum bytes
FWD_GFLOP_PER_IMP = {"NRMS": 1.593, "NAML": 1.497}
FUSED_MIN_BYTES_PER_IMP_FWD = {"bf16": 1.085e6, "fp32": 2.075e6}


def make_args(dtype):
    return SimpleNamespace(num_words_title=30, user_log_length=50, npratio=4, word_embedding_dim=300, news_dim=400,
                           num_attention_heads=20, news_query_vector_dim=200, user_query_vector_dim=200, drop_rate=0.2,
                           user_log_mask=False, freeze_embedding=False, use_category=False, use_subcategory=False,
                           category_emb_dim=100, compute_dtype=dtype)


def synth_batches(args, B, V, n_batches, seed, device, dense=False):
    """Seeded MIND-shaped batches: title length ~U[5,30] zero padded, history length ~U[0,50]
    front padded, 1+K candidates, label ~U[0,K].  dense: every title 30 tokens, every history 50 clicks."""
    g = torch.Generator().manual_seed(seed)
    T, H, C = args.num_words_title, args.user_log_length, 1 + args.npratio
    out = []
    for _ in range(n_batches):
        hist = torch.randint(1, V, (B, H, T), generator=g, dtype=torch.int32)
        cand = torch.randint(1, V, (B, C, T), generator=g, dtype=torch.int32)
        if dense:
            mask = torch.ones(B, H)
        else:
            for t in (hist, cand):
                ln = torch.randint(5, T + 1, t.shape[:2], generator=g)
                t[torch.arange(T)[None, None, :] >= ln[..., None]] = 0
            hl = torch.randint(0, H + 1, (B,), generator=g)
            mask = (torch.arange(H)[None, :] >= (H - hl)[:, None]).float()
            hist[mask == 0] = 0
        label = torch.randint(0, C, (B,), generator=g, dtype=torch.int64)
        out.append(tuple(x.to(device) for x in (hist, mask, cand, label)))
    return out


def synth_batches_naml(args, B, n_news, n_batches, seed, device, dense=False):
    """NAML inputs: [news id, category id (<=17), subcategory id (<=264)] per slot."""
    g = torch.Generator().manual_seed(seed)
    H, C = args.user_log_length, 1 + args.npratio
    out = []
    for _ in range(n_batches):
        def ids(shape):
            return torch.stack([torch.randint(1, n_news + 1, shape, generator=g, dtype=torch.int32),
                                torch.randint(0, 18, shape, generator=g, dtype=torch.int32),
                                torch.randint(0, 265, shape, generator=g, dtype=torch.int32)], dim=-1)
        hist, cand = ids((B, H)), ids((B, C))
        hl = torch.full((B,), H) if dense else torch.randint(0, H + 1, (B,), generator=g)
        mask = (torch.arange(H)[None, :] >= (H - hl)[:, None]).float()
        hist[mask == 0] = 0
        label = torch.randint(0, C, (B,), generator=g, dtype=torch.int64)
        out.append(tuple(x.to(device) for x in (hist, mask, cand, label)))
    return out


def synth_news_table(args, n_news, V, seed):
    """news_combined [n_news+1, T] int32: title token ids, length ~U[5,30], zero padded; row 0 = the unknown news."""
    g = torch.Generator().manual_seed(seed)
    T = args.num_words_title
    comb = torch.randint(1, V, (n_news + 1, T), generator=g, dtype=torch.int32)
    ln = torch.randint(5, T + 1, (n_news + 1,), generator=g)
    comb[torch.arange(T)[None, :] >= ln[:, None]] = 0
    comb[0] = 0
    return comb


# ----------------------------------------------------------------------------------------- pricing
def batch_structure(batches, args, model_name):
    """What the device-side compactions of libnrhip see in these batches, recomputed on the host outside the timed region:
    share of token rows with a non-padding id (`*_live` NT GEMMs), share of 32-row slabs that touch a title with a
    non-zero upstream gradient (`gemm_tn3_live`), share of sequences the attention backward walks."""
    T = args.num_words_title
    live_rows, live_slabs, live_seq, live_titles, needed, needed_tiles, walked = [], [], [], [], [], [], []
    for hist, mask, cand, _ in batches:
        B, H = mask.shape
        C = cand.shape[1]
        nz_title = torch.cat([torch.ones(B * C, device=mask.device), (mask.reshape(-1) != 0).float()])          # dy != 0
        if model_name == "NRMS":
            ids = torch.cat([cand.reshape(B * C, T), hist.reshape(B * H, T)])
            live_rows.append(float((ids != 0).float().mean()))
            allpad = (ids == 0).all(-1)
        else:
            live_rows.append(1.0)
            allpad = torch.zeros_like(nz_title, dtype=torch.bool)
        live_titles.append(float((~allpad).float().mean()))
        needed.append(float(nz_title.mean()))
        M = nz_title.numel() * T
        row_nz = nz_title.repeat_interleave(T)
        pad = (-M) % 32
        slab_nz = torch.nn.functional.pad(row_nz, (0, pad)).view(-1, 32).amax(1)
        live_slabs.append(float(slab_nz.mean()))
        tile_nz = torch.nn.functional.pad(row_nz, (0, (-M) % 128)).view(-1, 128).amax(1)
        needed_tiles.append(float(tile_nz.mean()))
        near = torch.nn.functional.max_pool1d(nz_title[None, None], kernel_size=7, stride=1, padding=3)[0, 0] > 0
        live_seq.append(float((~(allpad & ~near)).float().mean()))
        walked.append(float((~(allpad & (nz_title == 0))).float().mean()))      # compact row storage: reach 0
    avg = lambda x: sum(x) / len(x)
    return {"live_token_rows": round(avg(live_rows), 4), "live_titles": round(avg(live_titles), 4), "needed_titles": round(avg(needed), 4),
            "needed_tiles": round(avg(needed_tiles), 4),
            "live_gradient_slabs": round(avg(live_slabs), 4), "attention_bwd_sequences": round(avg(live_seq), 4),
            "attention_bwd_sequences_compact": round(avg(walked), 4)}


def _dims(label, pat):
    m = re.search(pat, label)
    return tuple(int(x) for x in m.groups()) if m else None


def price_kernel(label, avg_ms, struct, dtype):
    """(bound, achieved, peak, unit, algorithmic work) of one kernel label.  GEMMs: 2*M*N*K over the rows they really
    contract (device-side live counts, see batch_structure); attention / pooling / row kernels: algorithmic bytes."""
    esz = 2 if "bf16" in label or dtype == "bf16" else 4
    s = avg_ms / 1e3
    name = label.split("[")[0]
    if name.startswith("gemm"):
        d = _dims(label, r"M(?:max)?=(\d+),N=(\d+),K=(\d+)")
        if d is None:
            return None
        M, N, K = d
        if name.endswith("_rows"):
            M = M * struct["live_token_rows"]                    # dense over the compactly stored live rows
        elif name.endswith("_live"):
            M = M * (struct["live_gradient_slabs"] if name.startswith("gemm_tn") else struct["live_token_rows"])
        elif name.endswith("_needed"):
            # row tiles (128 rows) / 32-row blocks (weights-in-registers kernels) with at least one needed title
            M = M * (struct["live_gradient_slabs"] if name.startswith("gemm_nt_wreg") else struct.get("needed_tiles", 1.0))
        fl = 2.0 * M * N * K
        peak = PEAK_MFMA_BF16 if "bf16" in label else PEAK_MFMA_F32
        # the skinny GEMMs of this path sit near the ridge (K = 200..400: ~200-300 FLOP per byte of activations moved): price
        # the launch against BOTH roofs and report the one that binds.  Bytes: the activation rows read + the rows written
        # (weights are a few hundred KB); "Mfull" = rows written even where nothing is computed (pooling dX writes zeros).
        ge = 2 if "bf16" in label else 4
        epi = _dims(label, r"epi=(\d+)")
        epi = epi[0] if epi else 0
        gap = _dims(label, r"gap=(\d+)")
        gap = gap[0] if gap else 0
        # gap > 0: the im2col-free conv operand -- row m is the K contiguous elements starting at buffer row m + m/gap of a
        # [M + M/gap, K/3] buffer, so consecutive rows OVERLAP: the launch reads each buffer row once, K/3 columns per row
        Ka = K // 3 if gap else K
        if name.startswith("gemm_tn"):
            by = M * (N + Ka) * ge                                # both operands are activations, the output is a weight gradient
        elif gap:
            by = M * (Ka + N) * ge
        elif epi == 2:
            by = M * K * ge + M * N * 4                           # + fp32 atomic rows (upper bound: before run merging)
        elif epi == 1:
            by = M * K * ge + d[0] * N * ge
        else:
            by = M * (K + N) * ge
        t_mfma, t_hbm = fl / (peak * 1e12), by / (PEAK_HBM * 1e9)
        if t_hbm > t_mfma:
            return {"bound": "hbm", "achieved": round(by / s / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s", "work": by}
        return {"bound": "mfma", "achieved": round(fl / s / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "work": fl}
    if name.startswith("attn") or name.startswith("mhsa_fused"):
        d = _dims(label, r"n=(\d+),L=(\d+),h=(\d+),d=(\d+)")
        n, L, h, dd = d
        rows, N = n * L, h * dd
        if name.startswith("mhsa_fused_bwd"):
            by = rows * (304 + N + N) * esz                      # read x rows + y-gradient, write packed dQ|dK|dV-free outputs
        elif name.startswith("mhsa_fused"):
            by = rows * (304 + N) * esz
        elif "gather" in name:
            by = rows * (3 * N + N) * esz                        # gather projected Q|K|V rows (L2 / MALL), write y
        elif name.startswith("attn_mfma_bwd_rows"):
            # compact row storage: dy of every sequence walked; Q|K|V read and dQ|dK|dV written for the LIVE rows only
            by = rows * (N * struct.get("attention_bwd_sequences_compact", 1.0) + 6 * N * struct["live_token_rows"]) * esz
        elif "fwd" in name:
            # read Q|K|V of the titles that have any live token (all-padding titles substitute the bias); write y of all
            # titles, or ("_live": the kernel walks the needed titles, a store-only kernel zero-fills the rest) of the needed ones
            wfrac = struct.get("needed_titles", 1.0) if name.endswith("_live") else 1.0
            by = rows * (3 * N * min(struct.get("live_titles", 1.0), wfrac) + N * wfrac) * esz
        else:
            by = rows * (3 * N + N + 3 * N) * esz                # read Q|K|V + dy, write dQ|dK|dV
            if name.endswith("_live"):
                by *= struct["attention_bwd_sequences"]
        return {"bound": "hbm", "achieved": round(by / s / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s", "work": by}
    if name.startswith("pool_fused"):
        # one pass per sequence that is needed (forward) / has a pooled gradient (backward): x in; e out (forward), or e in and
        # dpre + dX out (backward); out / alpha / g rows are small
        n, L, N, q = _dims(label, r"n=(\d+),L=(\d+),N=(\d+),q=(\d+)")
        frac = struct.get("needed_titles", 1.0)
        per_row = (N + q) if "fwd" in name else (N + q + q + N)
        by = n * frac * L * per_row * esz
        return {"bound": "hbm", "achieved": round(by / s / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s", "work": by}
    if name.startswith("pool_core"):
        n, L, N, q = _dims(label, r"n=(\d+),L=(\d+),N=(\d+),q=(\d+)")
        by = n * L * (N + q) * esz * (1 if "fwd" in name else 1) + n * L * q * esz * (0 if "fwd" in name else 1)
        return {"bound": "hbm", "achieved": round(by / s / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s", "work": by}
    if name.startswith("conv_rows"):
        n, T, Dp = _dims(label, r"n=(\d+),T=(\d+),Dp=(\d+)")
        frac = struct.get("live_gradient_slabs", 1.0) if name.endswith("_needed") else 1.0     # titles near a needed one (approx.)
        by = n * frac * (T + 2) * Dp * esz * 2                  # gathered table rows read + token rows (and zero rows) written
        return {"bound": "hbm", "achieved": round(by / s / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s", "work": by}
    if name.startswith("rows_materialize"):
        M, K = _dims(label, r"M(?:max)?=(\d+),K=(\d+)")
        if name.endswith("_live"):
            M = M * struct["live_token_rows"]                    # only the live rows are written
        elif name.endswith("_needed"):
            M = M * struct.get("needed_tiles", 1.0)              # titles near a needed one (upper bound: tile granularity)
        by = M * K * esz
        return {"bound": "hbm", "achieved": round(by / s / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s", "work": by}
    return None


def pmc_file(workload=None):
    """The newest committed PMC summary (profiles/*_hbm_traffic_pmc.json) -- and only if it was collected on THIS workload:
    `workload` = (model, dtype, batch) must equal the file's step_total.{model,dtype,batch} (the passes profile the default
    NRMS bf16 B=512 train step; NAML / eval / fp32 / dense lines get no PMC traffic from it)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.json")))
    if not files:
        return None
    pm = json.load(open(files[-1]))
    pm["_source"] = os.path.relpath(files[-1], ROOT)
    tot = pm.get("step_total") or {}
    if workload is not None and (tot.get("model"), tot.get("dtype"), tot.get("batch")) != tuple(workload):
        return None
    return pm


def pmc_traffic(label, workload=None):
    """HBM bytes per launch of the kernel behind `label`, from the committed rocprofv3 PMC passes of this command
    (profiles/*_hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, FETCH doubled as
    MI355X_MICROARCH.md prescribes for 16-byte coalesced reads on gfx950).  None when no matching entry exists or the
    file belongs to another workload."""
    pm = pmc_file(workload)
    if pm is None:
        return None
    key = {"gemm_nt[": "gemm_nt_kernelIDF16bLi0ELi0", "gemm_nt_dma": "gemm_nt_dma_kernel", "gemm_nt_wreg": "gemm_nt_wreg_kernel", "gemm_tn2": "tn2::gemm_tn2_kernel", "gemm_tn3": "tn3::gemm_tn3_kernel",
           "attn_mfma_bwd": "b16::bwd_kernel", "attn_mfma_fwd": "b16::fwd_kernel", "gemm_nt_wide": "gemm_nt_wide_kernel",
           "mhsa_fused_fwd": "fused_fwd", "mhsa_fused_bwd": "fused_bwd", "pool_fused_fwd": "pool_fused_fwd_kernel",
           "pool_fused_bwd": "pool_fused_bwd_kernel", "rows_materialize_live": "gather_live_rows_kernel"}
    want = next((v for k, v in key.items() if label.startswith(k)), None)
    if want is None:
        return None
    if (label.startswith("gemm_nt_dma") or label.startswith("gemm_nt_wreg")) and "epi=" in label:      # one instantiation per epilogue: <EPI, ...>
        want += "<" + label.split("epi=")[1].split(",")[0] + ","
    for k in pm["kernels"]:
        if want in k["kernel"]:
            return round((k["fetch_GB_x2_gfx950_16B_correction"] + k["write_GB"]) * 1e9)
    return None


def roofline_of(prof, dtype, struct, workload=None):
    """The kernel with the largest total time in the timed region, priced.  workload = (model, dtype, batch) of a plain
    train run (PMC traffic is attached only from a PMC file of that very workload), None otherwise."""
    if not prof:
        return None
    label, (cnt, ms) = max(prof.items(), key=lambda kv: kv[1][1])
    pr = price_kernel(label, ms / cnt, struct, dtype)
    out = {"kernel": label, "avg_ms": round(ms / cnt, 4), "launches": cnt,
           "traffic": pmc_traffic(label, workload) if workload is not None else None}
    if pr is None:
        out.update({"bound": "hbm", "achieved": None, "peak": PEAK_HBM, "unit": "GB/s", "frac": None})
        return out
    out.update({"bound": pr["bound"], "achieved": pr["achieved"], "peak": pr["peak"], "unit": pr["unit"],
                "frac": round(pr["achieved"] / pr["peak"], 4),
                ("algorithmic_flops" if pr["bound"] == "mfma" else "algorithmic_bytes"): int(pr["work"])})
    return out


def step_roofline(model_name, dtype, batch, ms_per_step, plain=True):
    """Whole training step against the binding bound of SURVEY §8(d): algorithmic FLOPs (3 x forward) / time / dense MFMA
    peak, and the HBM bytes a step moves (sum over all dispatches of the committed PMC passes) against the fused minimum
    (forward minimum x 3: activations are read again and their gradients written in the backward)."""
    fl = 3.0 * FWD_GFLOP_PER_IMP[model_name] * 1e9 * batch
    peak = PEAK_MFMA_BF16 if dtype == "bf16" else PEAK_MFMA_F32
    out = {"bound": "mfma", "algorithmic_flops": fl, "achieved": round(fl / (ms_per_step / 1e3) / 1e12, 1), "peak": peak, "unit": "TFLOP/s"}
    out["frac"] = round(out["achieved"] / peak, 4)
    pm = pmc_file((model_name, dtype, batch))
    fused_min = 3.0 * FUSED_MIN_BYTES_PER_IMP_FWD[dtype] * batch
    out["fused_min_bytes"] = int(fused_min)
    tot = pm.get("step_total") if pm else None
    if plain and tot:
        out["hbm_bytes_per_step_pmc"] = int(tot["bytes_per_step"])
        out["hbm_bytes_source"] = pm["_source"] + " (committed rocprofv3 --pmc passes of this command; NOT measured in this run)"
        out["wasted_traffic_ratio"] = round(tot["bytes_per_step"] / fused_min, 2)
        out["hbm_frac_of_peak_at_measured_time"] = round(tot["bytes_per_step"] / (ms_per_step / 1e3) / 1e9 / PEAK_HBM, 4)
    return out


# ----------------------------------------------------------------------------------------- CPU baseline
def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(args, V, seed, B=512, timed_steps=3):
    """The CPU oracle (oracle/nr_oracle.py, torch-CPU fp32) on the SAME workload: NRMS train step (fwd + bwd + Adam,
    Bernoulli dropout masks drawn per step) at B = 512.  torch's intra-op pool oversubscribes on the many
    small ops of this graph, so the thread count is swept first (one step at B = 128 per candidate, fastest kept), then one
    warm-up and `timed_steps` timed steps at B = 512 (SURVEY §8d asks for 3 + 5; 1 + 3 keeps the default run within minutes
    -- the `sample` string says what was run)."""
    from oracle import nr_oracle as O
    cores = physical_cores()
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(V, args.word_embedding_dim, generator=g) * 0.4
    table[0] = 0
    sd = O.init_state_dict("NRMS", args, table, seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(params.values(), lr=1e-4)
    T, N, p = args.num_words_title, args.news_dim, args.drop_rate

    def make_step(b):
        hist, mask, cand, label = synth_batches(args, b, V, 1, seed + 1, "cpu")[0]

        def step():
            keep = {"cand_word": torch.bernoulli(torch.full((b * 5, T, 300), 1 - p)),
                    "cand_ctx": torch.bernoulli(torch.full((b * 5, T, N), 1 - p)),
                    "hist_word": torch.bernoulli(torch.full((b * 50, T, 300), 1 - p)),
                    "hist_ctx": torch.bernoulli(torch.full((b * 50, T, N), 1 - p))}
            loss, _ = O.nrms_forward(hist, mask, cand, label, params, args, keep=keep)
            opt.zero_grad()
            loss.backward()
            opt.step()
        return step

    small = make_step(128)
    sweep = {}
    for th in sorted({t for t in (16, 32, 64, 128) if t <= max(cores, 16)} | {min(cores, 128)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        small()
        sweep[th] = round(time.perf_counter() - t0, 2)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    step = make_step(B)
    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(B * timed_steps / dt, 2), "unit": "impressions/s", "cores": threads, "kind": "port",
            "sample": f"NRMS train step (fwd+bwd+Adam, dropout 0.2) at B={B}: {timed_steps} timed steps after 1 warm-up ({warm:.1f} s), "
                      f"torch-CPU fp32, {threads} threads = fastest of the sweep {sweep} (seconds per B=128 step; host has {cores} physical cores)"}


def cpu_eval_baseline(args, V, n_news, n_imp, seed, budget_s=20.0):
    """The oracle's eval path on a bounded sample: encode a slice of the corpus, score + rank a slice of the impressions."""
    from oracle import nr_oracle as O
    import numpy as np
    cores = physical_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(V, args.word_embedding_dim, generator=g) * 0.4
    table[0] = 0
    sd = O.init_state_dict("NRMS", args, table, seed=0)
    ev = SimpleNamespace(**vars(args))
    ev.user_log_mask = True
    comb = synth_news_table(args, 4096, V, seed + 1)
    with torch.no_grad():
        t0 = time.perf_counter()
        nv = O.nrms_news_encoder(comb[:2048], sd, ev)
        t_news = (time.perf_counter() - t0) / 2048
        nv = torch.cat([nv, nv, nv[:1]])[: comb.shape[0]]
        rnd = np.random.RandomState(seed)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s / 2 and k < 4000:
            B = 64
            hist = torch.from_numpy(rnd.randint(0, comb.shape[0], (B, 50)))
            mask = torch.from_numpy((rnd.rand(B, 50) < 0.6).astype("float32"))
            uv = O.nrms_user_encoder(nv[hist], mask, sd, ev)
            for b in range(B):
                c = rnd.randint(2, 101)
                s = (nv[torch.from_numpy(rnd.randint(0, comb.shape[0], c))] @ uv[b]).numpy()
                y = (rnd.rand(c) < 0.15).astype("int64")
                if y.mean() not in (0, 1):
                    O.auc_score(y, s), O.mrr_score(y, s), O.ndcg_score(y, s, 5), O.ndcg_score(y, s, 10)
            k += B
        t_imp = (time.perf_counter() - t0) / max(k, 1)
    total = t_news * n_news + t_imp * n_imp
    return {"value": round(n_imp / total, 2), "unit": "impressions/s", "cores": cores, "kind": "port",
            "sample": f"oracle eval path extrapolated from 2048 encoded news ({t_news * 1e3:.2f} ms/news) and {k} scored + ranked "
                      f"impressions ({t_imp * 1e3:.2f} ms/impression) to {n_news} news + {n_imp} impressions, {cores} threads"}


class ArrayShard:
    """Synthetic stand-in with the field contract of data.IndexedTrainShard / IndexedTestShard (index arrays, no files)."""

    def __init__(self, n, **arrays):
        self.n = n
        self.__dict__.update(arrays)

    def __len__(self):
        return self.n

    def draw_labels(self):
        import random
        import numpy as np
        return np.fromiter((random.randint(0, self.npratio) for _ in range(self.n)), dtype=np.int64, count=self.n)


# ----------------------------------------------------------------------------------------- eval workload
def run_eval(a, args, device, rank, world, dist_on, steps=None, warmup=None, cpu=True):
    """BASELINE configs[4], one GPU's share: encode the news corpus once (sharded over the ranks + all_gather when N > 1),
    then score `--eval-impressions` impressions (candidates ~U[2,100]) and rank them, all on the device."""
    import numpy as np
    import torch.distributed as dist
    from newsrecommendation_amd import _lib, train as TR
    from newsrecommendation_amd.model import NRMS
    args.user_log_mask = True                                # src/demo.sh:26 evaluates with the masked user encoder
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    table = torch.randn(a.vocab, args.word_embedding_dim, generator=g) * 0.4
    table[0] = 0
    model = NRMS.Model(args, table.numpy()).to(device).eval()
    comb = synth_news_table(args, a.eval_news, a.vocab, 3).numpy()
    rnd = np.random.RandomState(10 + rank)
    n = a.eval_impressions
    counts = rnd.randint(2, 101, n)
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum(counts)
    hl = rnd.randint(0, 51, n)
    hist = rnd.randint(1, a.eval_news + 1, (n, 50)).astype(np.int32)
    mask = (np.arange(50)[None, :] >= (50 - hl)[:, None]).astype(np.float32)
    hist[mask == 0] = 0
    cand_np, label_np = rnd.randint(1, a.eval_news + 1, off[-1]).astype(np.int32), (rnd.rand(off[-1]) < 0.1).astype(np.int32)
    if getattr(a, "eval_feed", "resident") == "resident":
        # inputs resident in HBM when the clock starts, as for the training lines (the index arrays of the shard and the corpus:
        # 90 MB; `--eval-feed host` re-uploads them from pageable numpy arrays in every pass, as a one-shot evaluation does)
        dev_t = lambda x: torch.as_tensor(x, device=device)
        sh = ArrayShard(n, hist=dev_t(hist), mask=dev_t(mask), cand=dev_t(cand_np), label=dev_t(label_np), offsets=off)
        comb_in = dev_t(comb)
    else:
        sh = ArrayShard(n, hist=hist, mask=mask, cand=cand_np, label=label_np, offsets=off)
        comb_in = comb

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    a = SimpleNamespace(**vars(a))
    a.steps, a.warmup = (steps or a.steps), (a.warmup if warmup is None else warmup)
    nv_keep = None

    def one_pass():
        nv = TR.encode_news(model, comb_in, a.eval_batch, device, shard_over_ranks=dist_on)
        scores, sums = TR.score_shard(model, nv, sh, a.eval_batch, device)
        return sums

    for _ in range(a.warmup):
        one_pass()
    fence()
    prof_on = (not a.no_prof) and rank == 0
    if prof_on:
        _lib.prof_enable(2)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sums = one_pass()
    fence()
    dt = time.perf_counter() - t0
    prof = {}
    if prof_on:
        _lib.prof_enable(False)
        prof = _lib.prof_collect()
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    sums = sums.cpu().tolist()
    struct = {"live_token_rows": float((comb != 0).mean()), "live_titles": 1.0, "needed_titles": 1.0, "needed_tiles": 1.0, "live_gradient_slabs": 1.0,
              "attention_bwd_sequences": 1.0}
    out = {"metric": "eval impressions/sec (full-corpus encode + scoring + ranking metrics), NRMS", "value": round(n * world * a.steps / dt, 1),
           "unit": "impressions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": f"NRMS eval: encode {a.eval_news} news (sharded over ranks) + score {n} impressions per GPU, candidates ~U[2,100], "
                                  "history 50, user_log_mask=True; AUC/MRR/nDCG on the device",
                      "news": a.eval_news, "impressions_per_gpu": n, "candidates": int(off[-1]), "eval_batch": a.eval_batch, "parallelism": f"dp{world}",
                      "feed": getattr(a, "eval_feed", "resident"),
                      "scored_impressions": int(sums[0]), "mean_auc": round(sums[1] / max(sums[0], 1), 4)}}
    out["roofline"] = roofline_of(prof, a.dtype, struct)
    if prof:
        out["kernel_ms_per_step"] = {k: round(ms / a.steps, 4) for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:10]}
    if world == 1 and not a.no_cpu_baseline and cpu:
        out["cpu_baseline"] = cpu_eval_baseline(make_args("fp32"), a.vocab, a.eval_news, n, 7)
    del model, nv_keep
    from newsrecommendation_amd import ops as _ops
    _ops.projected_tables._c.clear()
    _ops.table_cache.invalidate()
    _ops.pack_cache.clear()
    torch.cuda.empty_cache()
    return out


def _rss():
    """(anonymous resident MB, peak resident MB) of this process, from /proc/self/status."""
    anon = peak = None
    try:
        for line in open("/proc/self/status"):
            if line.startswith("RssAnon:"):
                anon = int(line.split()[1]) / 1024.0
            elif line.startswith("VmHWM:"):
                peak = int(line.split()[1]) / 1024.0
    except OSError:
        pass
    return anon, peak


def build_naml_from_disk(a, args, device):
    """BASELINE configs[2] start-up through SURVEY §8 row f3: the [N+1, T*D] fp32 title matrix is written block by block as a
    plain `.npy` (the inflate-once cache file of formats.read_news_embeddings), memory-mapped, and streamed by
    NAML.TitleTable into the packed bf16 table on the device.  Returns (model, load statistics)."""
    import numpy as np
    import tempfile
    from newsrecommendation_amd.model import NAML
    rows, width = a.naml_news + 1, args.num_words_title * args.word_embedding_dim
    tmp = tempfile.mkdtemp(prefix="nr_bench_")
    path = os.path.join(tmp, "title_embeddings.bpemb.npy")
    mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(rows, width))
    g = torch.Generator().manual_seed(1)
    for r0 in range(0, rows, 4096):
        r1 = min(rows, r0 + 4096)
        mm[r0:r1] = (torch.randn(r1 - r0, width, generator=g) * 0.4).numpy()
    mm[0] = 0
    mm.flush()
    del mm
    anon0, _ = _rss()
    torch.cuda.synchronize()
    dev0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    table = np.load(path, mmap_mode="r")
    model = NAML.Model(args, table, 17, 264).to(device)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    anon1, peak = _rss()
    stats = {"source": "memory-mapped .npy (formats.read_news_embeddings cache layout), streamed by NAML.TitleTable",
             "file_GB": round(rows * width * 4 / 1e9, 2), "load_s": round(dt, 2),
             "device_MB_after_load": round((torch.cuda.memory_allocated() - dev0) / 1e6, 1),
             "host_anon_rss_growth_MB": None if anon0 is None else round(anon1 - anon0, 1), "host_peak_rss_MB": peak and round(peak, 1)}
    return model, stats, tmp


def run_train(a, device, rank, world, dist_on, model_name, dtype, dense=False, steps=None, warmup=None, plain_headline=True):
    """One train workload: W untimed warm-up steps, exactly K timed steps between barrier + synchronize fences, MAX over
    ranks.  Returns the JSON dict on rank 0 (None elsewhere)."""
    import shutil
    import torch.distributed as dist
    from newsrecommendation_amd import _lib, ops as _ops, parallel, train as TR
    from newsrecommendation_amd.model import NRMS
    steps, warmup = (steps or a.steps), (a.warmup if warmup is None else warmup)
    args = make_args(dtype)
    args.freeze_embedding = bool(a.freeze_embedding)
    args.compact_history = bool(a.compact_history)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    table_load, tmpdir = None, None
    if model_name == "NRMS":
        table = torch.randn(a.vocab, args.word_embedding_dim, generator=g) * 0.4
        table[0] = 0
        model = NRMS.Model(args, table.numpy()).to(device)
    else:
        args.use_category = args.use_subcategory = True
        args.freeze_embedding = True                      # src/demo.sh:12
        model, table_load, tmpdir = build_naml_from_disk(a, args, device)
    model.train()
    if a.deterministic:
        _ops.set_deterministic(True, elements=sum(p.numel() for p in model.parameters() if p.requires_grad) + (1 << 20), device=device)
    net, bucket, opt = model, None, None
    if a.dp_mode == "flat":
        bucket = parallel.FlatBucket(model, lr=1e-4)       # rank-0 broadcast; per step ONE all-reduce + ONE fused Adam kernel
    else:
        # src/main.py:76 (defaults); fused=True is the same update rule in one multi-tensor kernel instead of ~6
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
        if dist_on:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index])   # src/main.py:82
    if model_name == "NRMS":
        batches = synth_batches(args, a.batch, a.vocab, 4, 100 + rank, device, dense=dense)
    else:
        batches = synth_batches_naml(args, a.batch, a.naml_news, 4, 100 + rank, device, dense=dense)

    feed = None
    if a.feed == "device":
        if model_name != "NRMS":
            raise SystemExit("--feed device is wired for NRMS inputs")
        import numpy as np
        n_news, n_lines = 65000, a.batch * (steps + warmup)
        comb = synth_news_table(args, n_news, a.vocab, 3).numpy()
        rnd = np.random.RandomState(100 + rank)
        hl = np.full(n_lines, 50) if dense else rnd.randint(0, 51, n_lines)
        hist = rnd.randint(1, n_news + 1, (n_lines, 50)).astype(np.int32)
        mask = (np.arange(50)[None, :] >= (50 - hl)[:, None]).astype(np.float32)
        hist[mask == 0] = 0
        sh = ArrayShard(n_lines, hist=hist, mask=mask, pos=rnd.randint(1, n_news + 1, n_lines).astype(np.int32),
                        neg=rnd.randint(1, n_news + 1, (n_lines, 4)).astype(np.int32), npratio=4)
        feed = TR.DeviceFeed(sh, comb, a.batch, device)
        feed.start_epoch()

    def step(i):
        hist, mask, cand, label = feed.batch(i) if feed is not None else batches[i % len(batches)]
        loss, score = net(hist, mask, cand, label)
        if opt is not None:
            opt.zero_grad()
        loss.backward()
        if bucket is not None:
            bucket.step()
        else:
            opt.step()
        return loss

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel timing (HIP events inside the library).  A pair of event records costs ~3 us of stream time, and bracketing
    # the dozen large launches of a step costs the timed region 2-3 %: so the WARM-UP steps (all but the first, which packs
    # and allocates) are bracketed broadly -- that is where kernel_ms_per_step and the choice of the dominant kernel come
    # from -- and the TIMED region brackets only that dominant kernel, which `roofline` is computed from.
    # --all-kernels: every launch, in the timed region (a diagnostic run, not a headline number).
    prof_on = (not a.no_prof) and rank == 0
    warm_prof, dominant = {}, None
    for i in range(warmup):
        if prof_on and not a.all_kernels and i == 1:
            torch.cuda.synchronize()
            _lib.prof_enable(2)
        loss = step(i)
    if prof_on and not a.all_kernels and warmup >= 2:
        _lib.prof_enable(False)
        warm_prof = _lib.prof_collect()
        if warm_prof:
            dominant = max(warm_prof.items(), key=lambda kv: kv[1][1])[0]
    fence()
    if prof_on:
        if a.all_kernels:
            _lib.prof_enable(1)
        elif dominant is not None:
            _lib.prof_enable(3, only=dominant)
        else:
            _lib.prof_enable(2)                     # no warm-up to choose from: the launches over >= 65 536 rows
    # A generation-2 pass of Python's cycle collector over this process's heap takes ~50 ms (measured: it landed in the first
    # timed step of the --feed device run and made 20 steps read 4.9 instead of 3.0 ms): collect BEFORE the clock starts, as
    # timeit does; the collector stays enabled.  NR_BENCH_STEP_TIMES=1 prints the host time of every enqueue.
    import gc
    gc.collect()
    fence()
    t0 = time.perf_counter()
    trace = [] if os.environ.get("NR_BENCH_STEP_TIMES") else None
    for i in range(steps):
        loss = step(warmup + i)
        if trace is not None:
            trace.append(time.perf_counter())
    fence()
    dt = time.perf_counter() - t0
    if trace is not None and rank == 0:
        print("host enqueue per step (ms):", [round(1e3 * (b - a), 2) for a, b in zip([t0] + trace[:-1], trace)],
              "drain", round(1e3 * (t0 + dt - trace[-1]), 2), file=sys.stderr)
    prof = {}
    if prof_on:
        _lib.prof_enable(False)
        prof = _lib.prof_collect()
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    out = None
    if rank == 0:
        total = a.batch * world * steps
        ms_step = dt / steps * 1e3
        if feed is not None:
            batches = [feed.batch(i) for i in range(warmup, warmup + 4)]
        struct = batch_structure(batches, args, model_name)
        plain = plain_headline and not (dense or a.deterministic or a.compact_history)
        out = {"metric": f"train impressions/sec @ batch {a.batch}, " + model_name, "value": round(total / dt, 1), "unit": "impressions/s",
               "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": model_name + " train step (fwd+bwd+Adam), MIND-small shapes: title_len=30, history=50, "
                                      "npratio=4, 300-d " + ("word table" if model_name == "NRMS" else "per-news title rows, 3 views")
                                      + (", DENSE batch (no padding)" if dense else ""),
                          "per_gpu_batch": a.batch, "global_batch": a.batch * world,
                          **({"vocab_rows": a.vocab} if model_name == "NRMS" else {"title_table_rows": a.naml_news + 1, "table_load": table_load}),
                          "dropout": args.drop_rate, "freeze_embedding": args.freeze_embedding,
                          "compact_history": bool(a.compact_history), "dense_batch": bool(dense), "feed": a.feed,
                          "deterministic": bool(a.deterministic),
                          "optimizer": "flat bucket + HIP fused Adam" if bucket is not None else "DDP + torch.optim.Adam(fused)",
                          "parallelism": f"dp{world}", "final_loss": round(final_loss, 4), "batch_structure": struct}}
        if dist_on and a.dist_backend != "nccl":
            out["config"]["rehearsal"] = f"{a.dist_backend} backend, ranks share GPUs: NOT a measurement"
        # the committed PMC passes profile the default (sparse, resident, non-deterministic) workload only
        out["roofline"] = roofline_of(prof, dtype, struct, workload=(model_name, dtype, a.batch) if plain else None)
        if out["roofline"] is not None:
            out["roofline"]["step"] = step_roofline(model_name, dtype, a.batch, ms_step, plain=plain)
        # the per-kernel table: from the timed region when it was bracketed broadly, else from the warm-up steps (the
        # dominant kernel's entry is then replaced by its timed-region measurement)
        table, tsteps, src = (prof, steps, "timed region") if (a.all_kernels or not warm_prof) else (dict(warm_prof), warmup - 1, "warm-up steps")
        if table:
            if src == "warm-up steps" and prof:
                for k, (c, ms) in prof.items():
                    table[k] = (round(c * tsteps / steps), ms * tsteps / steps)
            tot = sum(ms for _, ms in table.values())
            top = sorted(table.items(), key=lambda kv: -kv[1][1])[:(None if a.all_kernels else 12)]
            out["kernel_ms_per_step"] = {k: round(ms / tsteps, 4) for k, (c, ms) in top}
            out["kernel_ms_per_step"]["_all_timed_libnrhip_kernels"] = round(tot / tsteps, 4)
            out["kernel_ms_per_step"]["_measured_in"] = src + (" (dominant kernel: timed region)" if src == "warm-up steps" else "")
            fr = {}
            for k, (c, ms) in top:
                pr = price_kernel(k, ms / max(c, 1), struct, dtype)
                if pr is not None:
                    fr[k] = {"bound": pr["bound"], "frac": round(pr["achieved"] / pr["peak"], 3)}
            out["kernel_roofline_frac"] = fr
    # leave the device as found: the next workload of this process builds its own model
    if a.deterministic:
        _ops.set_deterministic(False)
    del model, net, bucket, opt, batches, feed
    _ops.pack_cache.clear()
    _ops.table_cache.invalidate()
    torch.cuda.empty_cache()
    if tmpdir:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return out


def brief(out):
    """What an `also` entry keeps of a workload's line."""
    if out is None:
        return None
    r = out.get("roofline") or {}
    b = {"metric": out["metric"], "value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"], "steps": out["steps"],
         "warmup": out["warmup"], "dtype": out["dtype"], "workload": out["config"]["workload"],
         "dominant_kernel": {k: r.get(k) for k in ("kernel", "avg_ms", "bound", "achieved", "peak", "unit", "frac")}}
    if "step" in r:
        b["step_frac_of_mfma_peak"] = r["step"].get("frac")
    for k in ("table_load", "mean_auc", "scored_impressions"):
        if k in out["config"]:
            b[k] = out["config"][k]
    if "kernel_ms_per_step" in out:
        b["kernel_ms_per_step_top5"] = dict(list(out["kernel_ms_per_step"].items())[:5])
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--vocab", type=int, default=30000)
    ap.add_argument("--model", default="NRMS", choices=["NRMS", "NAML"],
                    help="NRMS = the headline config; NAML = BASELINE config[2] (multi-view, frozen [N+1, T*D] title table)")
    ap.add_argument("--naml-news", type=int, default=65000)
    ap.add_argument("--freeze-embedding", action="store_true")
    ap.add_argument("--dense-batch", action="store_true", help="full histories and full-length titles (no padding)")
    ap.add_argument("--feed", default="resident", choices=["resident", "device"],
                    help="resident: 4 pre-built batches cycle (the headline); device: every step assembles a NEW batch on the device "
                         "from news-index arrays (train.DeviceFeed, SURVEY §8 row f1)")
    ap.add_argument("--dp-mode", default="flat", choices=["flat", "ddp"],
                    help="flat: parallel.FlatBucket (one all-reduce + fused HIP Adam); ddp: DistributedDataParallel + torch.optim.Adam")
    ap.add_argument("--deterministic", action="store_true",
                    help="ops.set_deterministic(True): fixed-point integer-atomic gradient accumulation (bit-reproducible gradients)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one rank per GPU: the measured configuration); gloo: rehearsal of the N > 1 path only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--compact-history", action="store_true",
                    help="opt-in: encode only history slots with mask != 0 (their vectors reach the loss through a factor 0); "
                         "NOT the headline configuration -- the reference encodes all 55 titles per impression")
    ap.add_argument("--all-kernels", action="store_true", help="list every kernel label in kernel_ms_per_step (default: top 12)")
    ap.add_argument("--eval", action="store_true", help="eval workload (BASELINE configs[4]) instead of the train step")
    ap.add_argument("--eval-news", type=int, default=100000)
    ap.add_argument("--eval-impressions", type=int, default=125000, help="impressions per GPU (1M over 8 GPUs)")
    ap.add_argument("--eval-batch", type=int, default=2048)
    ap.add_argument("--eval-feed", default="resident", choices=["resident", "host"],
                    help="eval: index arrays resident on the device before the clock starts (default), or uploaded from host arrays in every pass")
    ap.add_argument("--no-also", action="store_true",
                    help="headline only: skip the other BASELINE configs (NAML, eval, dense batch, fp32) that the default 1-GPU "
                         "run measures in the same process and attaches under `also`")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if a.gpus != world and dist_on:
        raise SystemExit(f"--gpus {a.gpus} != WORLD_SIZE {world}")
    # one rank per GPU; `--dist-backend gloo` is for rehearsing the N > 1 code path on a box with fewer GPUs than ranks
    # (ranks then share a card and the collectives go through the host): never a measurement
    dev_index = local_rank if a.dist_backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if dist_on:
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)   # 'nccl' == RCCL on ROCm (src/main.py:31)
        else:
            dist.init_process_group(a.dist_backend)

    if a.eval:
        if a.steps == 20 and a.warmup == 5:
            a.steps, a.warmup = 3, 1
        out = run_eval(a, make_args(a.dtype), device, rank, world, dist_on)
        if out is not None:
            print(json.dumps(out))
        if dist_on:
            dist.destroy_process_group()
        return

    out = run_train(a, device, rank, world, dist_on, a.model, a.dtype, dense=a.dense_batch)
    # The default 1-GPU run (the driver's BENCH line) also puts the other BASELINE configs on the same clock, in this
    # process, after the headline's timed region: configs[2] NAML, configs[4] eval (one GPU's share), the dense batch (what
    # the step costs when the sparsity shortcuts find nothing) and the fp32 mode (the one held to 1e-4).  Headline keys are
    # untouched; each entry is a full measurement with its own warm-up, fences and dominant kernel.
    headline_default = (a.model == "NRMS" and a.dtype == "bf16" and not a.dense_batch and not a.deterministic and a.feed == "resident"
                        and not a.compact_history and a.dp_mode == "flat" and not a.freeze_embedding)
    if out is not None and world == 1 and headline_default and not a.no_also:
        also = {}
        t_also = time.perf_counter()
        for key, fn in (("naml", lambda: run_train(a, device, rank, world, False, "NAML", "bf16", steps=10, warmup=3)),
                        ("eval", lambda: run_eval(a, make_args("bf16"), device, rank, world, False, steps=2, warmup=1, cpu=False)),
                        ("dense", lambda: run_train(a, device, rank, world, False, "NRMS", "bf16", dense=True, steps=10, warmup=3)),
                        ("fp32", lambda: run_train(a, device, rank, world, False, "NRMS", "fp32", steps=5, warmup=2))):
            try:
                also[key] = brief(fn())
            except Exception as e:                      # the headline line must survive a failing side workload
                also[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        also["_seconds"] = round(time.perf_counter() - t_also, 1)
        out["also"] = also
    if out is not None and world == 1 and not a.no_cpu_baseline and a.model == "NRMS":
        out["cpu_baseline"] = cpu_baseline(make_args(a.dtype), a.vocab, 7)
    if out is not None:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
