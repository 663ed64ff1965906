#!/usr/bin/env python3
"""Headline benchmark: NRMS train impressions/sec at per-GPU batch 512 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of the reference loop (src/main.py:98-110) on one resident
synthetic MIND-shaped batch: forward (dropout 0.2 on, as `model.train()`), loss, backward (incl. the
word-embedding table gradient: `freeze_embedding` defaults to False, src/parameters.py:47), gradient
all-reduce when N > 1 (DistributedDataParallel over RCCL, as src/main.py:82), Adam step.
Weak scaling: every rank owns its own 512-impression shard per step (impressions are independent).

The JSON line also carries
  roofline     - the dominant libnrhip kernel of the timed region, timed live with HIP events inside the
                 library (nr_prof_*), against the MI355X dense-MFMA / HBM peak;
  cpu_baseline - the CPU oracle (a port, torch-CPU fp32) timed on this host on a bounded sample.
"""
import argparse
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


def make_args(dtype):
    return SimpleNamespace(num_words_title=30, user_log_length=50, npratio=4, word_embedding_dim=300, news_dim=400,
                           num_attention_heads=20, news_query_vector_dim=200, user_query_vector_dim=200, drop_rate=0.2,
                           user_log_mask=False, freeze_embedding=False, use_category=False, use_subcategory=False,
                           category_emb_dim=100, compute_dtype=dtype)


def synth_batches(args, B, V, n_batches, seed, device):
    """Seeded MIND-shaped batches: title length ~U[5,30] zero padded, history length ~U[0,50]
    front padded, 1+K candidates, label ~U[0,K]."""
    g = torch.Generator().manual_seed(seed)
    T, H, C = args.num_words_title, args.user_log_length, 1 + args.npratio
    out = []
    for _ in range(n_batches):
        hist = torch.randint(1, V, (B, H, T), generator=g, dtype=torch.int32)
        cand = torch.randint(1, V, (B, C, T), generator=g, dtype=torch.int32)
        for t in (hist, cand):
            ln = torch.randint(5, T + 1, t.shape[:2], generator=g)
            t[torch.arange(T)[None, None, :] >= ln[..., None]] = 0
        hl = torch.randint(0, H + 1, (B,), generator=g)
        mask = (torch.arange(H)[None, :] >= (H - hl)[:, None]).float()
        hist[mask == 0] = 0
        label = torch.randint(0, C, (B,), generator=g, dtype=torch.int64)
        out.append(tuple(x.to(device) for x in (hist, mask, cand, label)))
    return out


def synth_batches_naml(args, B, n_news, n_batches, seed, device):
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
        hl = torch.randint(0, H + 1, (B,), generator=g)
        mask = (torch.arange(H)[None, :] >= (H - hl)[:, None]).float()
        hist[mask == 0] = 0
        label = torch.randint(0, C, (B,), generator=g, dtype=torch.int64)
        out.append(tuple(x.to(device) for x in (hist, mask, cand, label)))
    return out


def gemm_flops(label):
    m = re.search(r"M=(\d+),N=(\d+),K=(\d+)", label)
    M, N, K = (int(x) for x in m.groups())
    return 2.0 * M * N * K


def attn_bytes(label, esz):
    m = re.search(r"n=(\d+),L=(\d+),h=(\d+),d=(\d+)", label)
    n, L, h, d = (int(x) for x in m.groups())
    rows, N = n * L, h * d
    if "_fwd" in label.split("[")[0]:
        return rows * (3 * N + N) * esz            # read Q|K|V, write y
    return rows * (3 * N + N + 3 * N) * esz        # read Q|K|V + dy, write dQ|dK|dV


def pmc_traffic(label):
    """HBM bytes per launch of the kernel behind `label`, from the committed rocprofv3 PMC passes of this command
    (profiles/*_hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, FETCH doubled as
    MI355X_MICROARCH.md prescribes for 16-byte coalesced reads on gfx950).  None when no matching entry exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.json")))
    if not files:
        return None
    key = {"gemm_nt[": "gemm_nt_kernelIDF16bLi0ELi0", "gemm_nt_dma": "gemm_nt_dma_kernel", "gemm_tn2": "tn2::gemm_tn2_kernel", "gemm_tn3": "tn3::gemm_tn3_kernel",
           "attn_mfma_bwd": "b16::bwd_kernel", "attn_mfma_fwd": "b16::fwd_kernel", "gemm_nt_wide": "gemm_nt_wide_kernel"}
    want = next((v for k, v in key.items() if label.startswith(k)), None)
    if want is None:
        return None
    if label.startswith("gemm_nt_dma") and "epi=" in label:      # one instantiation per epilogue: <EPI, NT16, PK>
        want += "<" + label.split("epi=")[1].split(",")[0] + ","
    for k in json.load(open(files[-1]))["kernels"]:
        if want in k["kernel"]:
            return round((k["fetch_GB_x2_gfx950_16B_correction"] + k["write_GB"]) * 1e9)
    return None


def live_sequence_fraction(batches, args):
    """Share of the B*(1+K+H) title sequences the attention backward really processes (label "attn_mfma_bwd_live"): it leaves
    out the all-padding sequences whose own and 3 neighbours' upstream gradients are exactly zero (masked history slots far
    from any live title) -- nothing downstream reads their dQ|dK|dV rows.  Recomputed here from the batch tensors, outside the
    timed region, to price that kernel with the bytes it moves rather than with the dense figure."""
    fr = []
    for hist, mask, cand, _ in batches:
        B, H = mask.shape
        C = cand.shape[1]
        nz = torch.cat([torch.ones(B * C, device=mask.device), (mask.reshape(-1) != 0).float()])
        allpad = torch.cat([(cand == 0).all(-1).reshape(-1), (hist == 0).all(-1).reshape(-1)])
        near = torch.nn.functional.max_pool1d(nz[None, None], kernel_size=7, stride=1, padding=3)[0, 0] > 0
        keep = ~(allpad & ~near)
        fr.append(float(keep.float().mean()))
    return sum(fr) / len(fr)


def roofline_of(prof, dtype, live_frac=1.0):
    """Pick the kernel with the largest total time in the timed region and price it."""
    if not prof:
        return None
    label, (cnt, ms) = max(prof.items(), key=lambda kv: kv[1][1])
    avg_s = ms / cnt / 1e3
    esz = 2 if dtype == "bf16" else 4
    if label.startswith("gemm") and "_live[" not in label:      # "_live": row count known on the device only
        fl = gemm_flops(label)
        peak = PEAK_MFMA_BF16 if "bf16" in label else PEAK_MFMA_F32
        ach = fl / avg_s / 1e12
        return {"kernel": label, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": pmc_traffic(label), "avg_ms": round(ms / cnt, 4), "launches": cnt,
                "algorithmic_flops": fl}
    if label.startswith("attn"):
        by = attn_bytes(label, esz)
        if "_live[" in label:
            by = int(by * live_frac)
        ach = by / avg_s / 1e9
        return {"kernel": label, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM, 4), "traffic": pmc_traffic(label), "avg_ms": round(ms / cnt, 4), "launches": cnt,
                "algorithmic_bytes": by}
    return {"kernel": label, "bound": "hbm", "achieved": None, "peak": PEAK_HBM, "unit": "GB/s", "frac": None,
            "traffic": None, "avg_ms": round(ms / cnt, 4), "launches": cnt}


def cpu_baseline(args, V, seed):
    """The CPU oracle (oracle/nr_oracle.py, torch-CPU fp32) on a bounded sample of the same workload:
    NRMS train step (fwd + bwd + Adam, Bernoulli dropout masks drawn per step) at B=64, 1 warm-up + 3 timed."""
    from oracle import nr_oracle as O
    Bs = 64
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(V, args.word_embedding_dim, generator=g) * 0.4
    table[0] = 0
    sd = O.init_state_dict("NRMS", args, table, seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(params.values(), lr=1e-4)
    hist, mask, cand, label = synth_batches(args, Bs, V, 1, seed + 1, "cpu")[0]
    T, N, p = args.num_words_title, args.news_dim, args.drop_rate

    def step():
        keep = {"cand_word": torch.bernoulli(torch.full((Bs * 5, T, 300), 1 - p)),
                "cand_ctx": torch.bernoulli(torch.full((Bs * 5, T, N), 1 - p)),
                "hist_word": torch.bernoulli(torch.full((Bs * 50, T, 300), 1 - p)),
                "hist_ctx": torch.bernoulli(torch.full((Bs * 50, T, N), 1 - p))}
        loss, _ = O.nrms_forward(hist, mask, cand, label, params, args, keep=keep)
        opt.zero_grad()
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(Bs * n / dt, 2), "unit": "impressions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"NRMS train step (fwd+bwd+Adam, dropout 0.2) at B={Bs}, {n} timed steps after 1 warm-up, torch-CPU fp32"}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--compact-history", action="store_true",
                    help="opt-in: encode only history slots with mask != 0 (their vectors reach the loss through a factor 0); "
                         "NOT the headline configuration -- the reference encodes all 55 titles per impression")
    ap.add_argument("--all-kernels", action="store_true", help="list every kernel label in kernel_ms_per_step (default: top 12)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if a.gpus != world and dist_on:
        raise SystemExit(f"--gpus {a.gpus} != WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)   # 'nccl' == RCCL on ROCm (src/main.py:31)

    from newsrecommendation_amd import _lib
    from newsrecommendation_amd.model import NAML, NRMS

    args = make_args(a.dtype)
    args.freeze_embedding = bool(a.freeze_embedding)
    args.compact_history = bool(a.compact_history)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    if a.model == "NRMS":
        table = torch.randn(a.vocab, args.word_embedding_dim, generator=g) * 0.4
        table[0] = 0
        model = NRMS.Model(args, table.numpy()).to(device)
    else:
        args.use_category = args.use_subcategory = True
        args.freeze_embedding = True                      # src/demo.sh:12
        table = torch.randn(a.naml_news + 1, args.num_words_title * args.word_embedding_dim, generator=g) * 0.4
        table[0] = 0
        model = NAML.Model(args, table.numpy(), 17, 264).to(device)
    model.train()
    # src/main.py:76 (defaults); fused=True is the same update rule in one multi-tensor kernel instead of ~6
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    net = model
    if dist_on:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])   # src/main.py:82
    if a.model == "NRMS":
        batches = synth_batches(args, a.batch, a.vocab, 4, 100 + rank, device)
    else:
        batches = synth_batches_naml(args, a.batch, a.naml_news, 4, 100 + rank, device)

    def step(i):
        hist, mask, cand, label = batches[i % len(batches)]
        loss, score = net(hist, mask, cand, label)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        loss = step(i)
    fence()
    prof_on = (not a.no_prof) and rank == 0
    if prof_on:
        # --all-kernels: bracket every launch; default: the launches over >= 65 536 rows (every kernel that can be the
        # dominant one) -- event records around the ~60 small launches of a step cost ~0.2 ms of stream time
        _lib.prof_enable(1 if a.all_kernels else 2)
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
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
    final_loss = float(loss.item())

    if rank == 0:
        total = a.batch * world * a.steps
        out = {"metric": f"train impressions/sec @ batch {a.batch}, " + a.model, "value": round(total / dt, 1), "unit": "impressions/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": a.model + " train step (fwd+bwd+Adam), MIND-small shapes: title_len=30, history=50, "
                                      "npratio=4, 300-d " + ("word table" if a.model == "NRMS" else "per-news title rows, 3 views"),
                          "per_gpu_batch": a.batch, "global_batch": a.batch * world, "vocab_rows": a.vocab,
                          "dropout": args.drop_rate, "freeze_embedding": args.freeze_embedding,
                          "compact_history": bool(a.compact_history),
                          "parallelism": f"dp{world}", "final_loss": round(final_loss, 4)}}
        out["roofline"] = roofline_of(prof, a.dtype, live_sequence_fraction(batches, args) if a.model == "NRMS" else 1.0)
        if prof:
            tot = sum(ms for _, ms in prof.values())
            out["kernel_ms_per_step"] = {k: round(ms / a.steps, 4) for k, (c, ms) in
                                         sorted(prof.items(), key=lambda kv: -kv[1][1])[:(None if a.all_kernels else 12)]}
            out["kernel_ms_per_step"]["_all_timed_libnrhip_kernels"] = round(tot / a.steps, 4)
        if world == 1 and not a.no_cpu_baseline and a.model == "NRMS":
            out["cpu_baseline"] = cpu_baseline(args, a.vocab, 7)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
