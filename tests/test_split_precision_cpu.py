"""CPU check of the arithmetic behind csrc/split_gemm.hip: a float32 operand is the exact sum of
three bfloat16 pieces, and the six leading bf16 products (exact in float32) accumulated in float32 are as close to
float64 as a plain float32 GEMM.  Mirrors tools/probe_split_precision.py; the HIP kernel itself is tested in
tests/test_split_gemm_gpu.py."""
import torch

PRODUCTS = {6: [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)], 3: [(1, 0), (0, 1), (0, 0)], 1: [(0, 0)]}   # PROD_A / PROD_B


def _split(x):
    parts, r = [], x.clone()
    for _ in range(3):
        p = r.bfloat16().float()
        parts.append(p)
        r = r - p
    return parts, r


def test_three_bf16_pieces_hold_a_float32_exactly():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.logspace(-20, 20, 4096)
    parts, rest = _split(x)
    assert torch.equal(rest, torch.zeros_like(x))
    assert torch.equal((parts[0] + parts[1]) + parts[2], x)


def test_six_products_match_float32_accuracy():
    g = torch.Generator().manual_seed(1)
    M, K, N = 128, 2304, 128                      # a 256-channel 3x3 layer's reduction
    a, b = torch.randn(M, K, generator=g), 0.02 * torch.randn(K, N, generator=g)
    ref = a.double() @ b.double()
    A, _ = _split(a)
    Bp, _ = _split(b)
    err = {}
    for n, terms in PRODUCTS.items():
        acc = torch.zeros(M, N)
        for i, j in terms:
            acc += A[i] @ Bp[j]
        err[n] = float((acc.double() - ref).abs().max() / ref.abs().max())
    plain = float(((a @ b).double() - ref).abs().max() / ref.abs().max())
    assert err[6] <= 2.0 * plain + 1e-7, (err, plain)
    assert err[3] <= 2e-5 and err[1] >= 1e-4, err
    assert err[6] < err[3] < err[1]
