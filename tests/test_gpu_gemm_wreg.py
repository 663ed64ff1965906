"""GPU: the weights-in-registers NT GEMMs (csrc/nr_gemm.hip, gemm_nt_wreg_kernel) and the id-sorted table-gradient
scatter against (a) fp64 torch products of the same bf16 operands and (b) the tiled kernels they replace
(`nr_set_option("NT_WREG", 0)`, `("NO_SCATTER_SORT", 1)`), at ragged shapes: row counts that are no multiple of the
16/32-row step, column counts that leave a wave / a column group partly empty, K at both ends of an instantiation's
range, sequence lengths 16..64 for the in-kernel list of needed 32-row blocks, all-needed / none-needed flags.

Tolerances: outputs are bf16 (8 significant bits): |diff| <= 2^-7 * max|ref| against fp64 and against the other kernel
(both accumulate in fp32 over K <= 416, so they differ by the last bf16 bit at most); fp32 gradients 1e-4 relative.
"""
import pytest
import torch

from newsrecommendation_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bf(x):
    return x.to(torch.bfloat16)


class _opt:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = _lib.get_option(self.name)
        _lib.set_option(self.name, self.value)

    def __exit__(self, *a):
        _lib.set_option(self.name, self.old)


def _weights(N, K, g, scale=0.1):
    Kr = (K + 31) // 32 * 32
    full = torch.zeros(N, Kr, device=DEV, dtype=torch.bfloat16)
    full[:, :K] = _bf(torch.randn(N, K, device=DEV, generator=g) * scale)
    return full[:, :K]                                        # row stride Kr, zero padded: what ops.pack produces


@pytest.mark.parametrize("M", [64, 1000, 4112])
@pytest.mark.parametrize("N,K", [(320, 296), (648, 304), (1200, 304), (1208, 320)])
def test_store_epilogue_dense_rows_ragged_shapes(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, device=DEV, generator=g) * 0.5)
    w = _weights(N, K, g)
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    c = ops.gemm_nt(a, w, bias=bias)
    with _opt("NT_WREG", 0):
        c0 = ops.gemm_nt(a, w, bias=bias)
    tol = 2.0 ** -7 * ref.abs().max().item()
    assert (c.double() - ref).abs().max().item() <= tol
    assert (c.double() - c0.double()).abs().max().item() <= tol
    assert torch.isfinite(c.float()).all()


def _pool_ref(x, w1, b1, w2, b2, mask):
    """AttentionPooling.forward, src/model/model_utils.py:13-31, in fp64 on the given (bf16-representable) inputs."""
    x = x.double()
    e = torch.tanh(x @ w1.double().t() + b1.double())
    alpha = torch.exp(e @ w2.double().reshape(-1, 1) + b2.double())
    if mask is not None:
        alpha = alpha * mask.double().unsqueeze(2)
    alpha = alpha / (alpha.sum(1, keepdim=True) + 1e-8)
    return (x * alpha).sum(1)


@pytest.mark.parametrize("n,L", [(40, 30), (37, 50), (64, 16), (33, 64), (2048, 30)])
@pytest.mark.parametrize("flags", ["none", "mixed", "all_dead"])
def test_pooling_fc1_and_dx_kernels_with_needed_blocks(n, L, flags):
    """att_fc1 + tanh (STORE_TANH, list of needed 32-row blocks) and the pooling dX GEMM (POOLBWD epilogue with alpha and the
    pooled-gradient rows riding in the stages, zeros for dead blocks) through ops.additive_pool, forward and backward."""
    g = torch.Generator(device=DEV).manual_seed(n * L)
    N, q = 400, 200
    x = _bf(torch.randn(n, L, N, device=DEV, generator=g) * 0.5)
    w1 = (torch.randn(q, N, device=DEV, generator=g) * 0.05).requires_grad_(True)
    b1 = (torch.randn(q, device=DEV, generator=g) * 0.05).requires_grad_(True)
    w2 = (torch.randn(1, q, device=DEV, generator=g) * 0.1).requires_grad_(True)
    b2 = torch.zeros(1, device=DEV, requires_grad=True)
    mask = (torch.rand(n, L, device=DEV, generator=g) < 0.8).float()
    mask[:, 0] = 1
    if flags == "none":
        needed = None
        gout = torch.randn(n, N, device=DEV, generator=g)
    else:
        keep = torch.rand(n, device=DEV, generator=g) < (0.6 if flags == "mixed" else -1.0)
        needed = ops.needed_flags(keep)
        gout = torch.randn(n, N, device=DEV, generator=g) * keep.float().unsqueeze(1)     # unneeded vectors get no gradient

    def run():
        xx = x.clone().requires_grad_(True)
        for p in (w1, b1, w2, b2):
            p.grad = None
        out = ops.additive_pool(xx, w1, b1, w2, b2, ops.NR_BF16, mask=mask, needed=needed)
        out.backward(gout)
        return out.detach(), xx.grad.detach().float(), w1.grad.clone(), b1.grad.clone(), w2.grad.clone()

    got = run()
    with _opt("NT_WREG", 0):
        old = run()
    names = ["out", "dx", "dw1", "db1", "dw2"]
    for nm, a, b in zip(names, got, old):
        assert torch.isfinite(a).all(), nm
        scale = b.abs().max().item()
        tol = (2.0 ** -6 if nm in ("out", "dx") else 5e-3) * scale + 1e-6
        assert (a - b).abs().max().item() <= tol, (nm, (a - b).abs().max().item(), scale)
    # forward against fp64 on the rows that are computed
    ref = _pool_ref(x, _bf(w1.detach()), b1.detach(), w2.detach(), b2.detach(), mask)
    rows = torch.ones(n, dtype=torch.bool, device=DEV) if needed is None else needed.bool()
    if rows.any():
        assert (got[0][rows].double() - ref[rows]).abs().max().item() <= 2e-2 * ref[rows].abs().max().item() + 1e-4
    if needed is not None and (~rows).any():
        assert got[0][~rows].abs().max().item() == 0.0
        assert got[1][~rows].abs().max().item() == 0.0          # dx rows of unneeded sequences: exact zeros, written


@pytest.mark.parametrize("n,L,pad", [(160, 30, 0.7), (138, 30, 0.3), (137, 30, 0.5), (512, 30, 0.97)])
def test_qkv_projection_over_compacted_rows_and_sorted_scatter(n, L, pad):
    """Gather-source MHSA at >= 4096 token rows: the QKV projection runs over the compacted live rows (row numbers riding
    with the stages), the table gradient over the same rows sorted by token id.  Against the tiled kernels / batch order."""
    V, D, heads, dh = 997, 300, 20, 20
    g = torch.Generator(device=DEV).manual_seed(n)
    ids = torch.randint(1, V, (n, L), device=DEV, generator=g, dtype=torch.int32)
    ids[torch.rand(n, L, device=DEV, generator=g) < pad] = 0
    ids[::7] = 0                                               # some all-padding sequences
    table = torch.randn(V, D, device=DEV, generator=g) * 0.4
    table[0] = 0
    table.requires_grad_(True)
    N = heads * dh
    ws = [(torch.randn(N, D, device=DEV, generator=g) * 0.05).requires_grad_(True) for _ in range(3)]
    bs = [(torch.randn(N, device=DEV, generator=g) * 0.05).requires_grad_(True) for _ in range(3)]
    gy = _bf(torch.randn(n, L, N, device=DEV, generator=g) * 0.1)

    def run():
        torch.manual_seed(5)                                   # the dropout seeds are drawn from torch's CPU generator
        for p in [table] + ws + bs:
            p.grad = None
        y = ops.mhsa(None, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], heads=heads, code=ops.NR_BF16, ids=ids, table=table,
                     p_in=0.2, p_out=0.2)
        y.backward(gy)
        return [y.detach().float(), table.grad.clone()] + [w.grad.clone() for w in ws] + [b.grad.clone() for b in bs]

    got = run()
    with _opt("NT_WREG", 0):
        tiled = run()
    with _opt("NO_SCATTER_SORT", 1):
        unsorted = run()
    for other in (tiled, unsorted):
        for i, (a, b) in enumerate(zip(got, other)):
            if i == 6:
                continue        # db_K is analytically zero (softmax is invariant to a key bias): what is left is rounding noise
            scale = b.abs().max().item()
            tol = (2.0 ** -6 if i == 0 else 2e-3) * scale + 1e-7
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= tol, (i, (a - b).abs().max().item(), scale)
    assert got[1][0].abs().max().item() == 0.0                 # padding_idx row gets no gradient


def test_sorted_scatter_is_bit_identical_to_batch_order_in_deterministic_mode():
    """Fixed-point accumulation is order independent: with it the table gradient must not depend on the row order at all."""
    n, L, V, D, heads, dh = 160, 30, 499, 300, 20, 20
    g = torch.Generator(device=DEV).manual_seed(3)
    ids = torch.randint(1, V, (n, L), device=DEV, generator=g, dtype=torch.int32)
    ids[torch.rand(n, L, device=DEV, generator=g) < 0.6] = 0
    table = (torch.randn(V, D, device=DEV, generator=g) * 0.4)
    table[0] = 0
    table.requires_grad_(True)
    N = heads * dh
    ws = [(torch.randn(N, D, device=DEV, generator=g) * 0.05).requires_grad_(True) for _ in range(3)]
    bs = [torch.zeros(N, device=DEV, requires_grad=True) for _ in range(3)]
    gy = _bf(torch.randn(n, L, N, device=DEV, generator=g) * 0.1)

    def run():
        torch.manual_seed(9)
        table.grad = None
        y = ops.mhsa(None, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], heads=heads, code=ops.NR_BF16, ids=ids, table=table,
                     p_in=0.2, p_out=0.0)
        y.backward(gy)
        return table.grad.clone()

    ops.set_deterministic(True, elements=1 << 22)
    try:
        a = run()
        a2 = run()
        with _opt("NO_SCATTER_SORT", 1):
            b = run()
    finally:
        ops.set_deterministic(False)
    assert torch.equal(a, a2)
    assert torch.equal(a, b)


@pytest.mark.parametrize("n,L,q,masked", [(160, 30, 200, False), (211, 24, 200, False), (137, 32, 200, False), (150, 30, 136, False),
                                           (2048, 30, 200, False), (1024, 30, 200, True), (800, 24, 208, False), (641, 32, 224, False),
                                           (640, 32, 224, True)])
@pytest.mark.parametrize("flags", ["none", "mixed", "all_dead"])
def test_fused_pooling_forward_against_the_two_kernel_path(n, L, q, masked, flags):
    """pool_fused_fwd_kernel (round 3): fc1 + tanh + fc2 + softmax + weighted sum in one pass over x, a stage = one sequence
    (32 rows: the L real ones + 32 - L rows of the NEXT sequence riding along), software-pipelined over three sequences.
    Against the fc1 GEMM + pool_core_fwd pair (`NO_POOL_FUSED` = 1) and against fp64: out, alpha-dependent gradients and e-
    dependent gradients (the backward reads the e and alpha the forward wrote).  The x rows of unneeded sequences hold 1e30:
    they ride along in their neighbours' stages (and n * L is no multiple of 32: the last stage is clamped) and must not leak.
    Sequence counts that leave workgroups with 0, 1, 2 and many steps; q below a full column group.
    The rows-multiple-of-32 cases with >= 16 384 rows also take the fused BACKWARD core (pool_fused_bwd_kernel: dA on the matrix
    cores from x fragments loaded straight into registers, dpre written in place over e in the LDS ring and to global memory,
    dX from the ring, dw2 / db2 partial sums); `masked` keeps the forward on the two-kernel path (the fused forward takes no
    mask) and the backward on the fused one (it needs none: alpha carries it)."""
    g = torch.Generator(device=DEV).manual_seed(n * L + q)
    N = 400
    assert n * L >= 4096
    x = _bf(torch.randn(n, L, N, device=DEV, generator=g) * 0.5)
    w1 = (torch.randn(q, N, device=DEV, generator=g) * 0.05).requires_grad_(True)
    b1 = (torch.randn(q, device=DEV, generator=g) * 0.05).requires_grad_(True)
    w2 = (torch.randn(1, q, device=DEV, generator=g) * 0.1).requires_grad_(True)
    b2 = (torch.randn(1, device=DEV, generator=g) * 0.1).requires_grad_(True)
    if flags == "none":
        needed, keep = None, torch.ones(n, dtype=torch.bool, device=DEV)
    else:
        keep = torch.rand(n, device=DEV, generator=g) < (0.55 if flags == "mixed" else -1.0)
        needed = ops.needed_flags(keep)
        x[~keep] = 1e30        # (finite: at these sizes the backward's dense dW1 = dpre^T . x multiplies these rows by exact zeros)
    gout = torch.randn(n, N, device=DEV, generator=g) * keep.float().unsqueeze(1)
    mask = None
    if masked:
        mask = (torch.rand(n, L, device=DEV, generator=g) < 0.8).float()
        mask[:, 0] = 1

    def run():
        xx = x.clone().requires_grad_(True)
        for p in (w1, b1, w2, b2):
            p.grad = None
        _lib.prof_enable(1)
        try:
            _lib.prof_collect()
            out = ops.additive_pool(xx, w1, b1, w2, b2, ops.NR_BF16, mask=mask, needed=needed)
            out.backward(gout)
            torch.cuda.synchronize()
            labels = set(_lib.prof_collect().keys())
        finally:
            _lib.prof_enable(0)
        return labels, (out.detach(), xx.grad.detach().float()[keep], w1.grad.clone(), b1.grad.clone(), w2.grad.clone(), b2.grad.clone())

    lab1, got = run()
    with _opt("NO_POOL_FUSED", 1):
        lab0, old = run()
    has = lambda labs, p: any(l.startswith(p) for l in labs)
    assert has(lab1, "pool_fused_fwd") != masked and has(lab1, "pool_core_fwd") == masked
    assert has(lab0, "pool_core_fwd") and not has(lab0, "pool_fused_fwd") and not has(lab0, "pool_fused_bwd")
    if (n * L) % 32 == 0 and n * L >= 16384 and 192 < q <= 224:
        assert has(lab1, "pool_fused_bwd") and not has(lab1, "pool_core_bwd")
    else:
        assert has(lab1, "pool_core_bwd") and not has(lab1, "pool_fused_bwd")
    for nm, a, b in zip(["out", "dx", "dw1", "db1", "dw2", "db2"], got, old):
        assert torch.isfinite(a).all(), nm
        if a.numel() == 0:
            continue
        scale = b.abs().max().item()
        tol = (2.0 ** -7 if nm in ("out", "dx") else 2e-3) * scale + 1e-6
        if nm == "db2":      # analytically 0 (the softmax weights sum to 1: a shift of every logit changes nothing): rounding noise
            tol = 1e-4 * old[4].abs().max().item() + 1e-6
        assert (a - b).abs().max().item() <= tol, (nm, (a - b).abs().max().item(), scale)
    if keep.any():
        xr = torch.where(keep[:, None, None], x, torch.zeros_like(x))
        ref = _pool_ref(xr, _bf(w1.detach()), b1.detach(), w2.detach(), b2.detach(), mask)
        assert (got[0][keep].double() - ref[keep]).abs().max().item() <= 1e-2 * ref[keep].abs().max().item() + 1e-4
    if (~keep).any():
        assert got[0][~keep].abs().max().item() == 0.0
