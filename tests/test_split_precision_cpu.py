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


def _tile_of_block(bx, by, gx, gy):
    """Python mirror of csrc/split_gemm.hip::split_tile_of_block (UFR_SPLIT_XCD=1)."""
    nwg, orig = gx * gy, by * gx + bx
    q, r, xcd, idx = nwg // 8, nwg % 8, orig % 8, orig // 8
    wgid = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
    return wgid % gx, wgid // gx


def test_xcd_tile_order_is_a_bijection_with_contiguous_runs():
    for gx, gy in ((2, 480), (4, 120), (8, 8), (1, 7), (3, 5), (2, 3), (5, 13), (32, 32), (1, 1), (7, 1)):
        seen, runs = set(), {}
        for by in range(gy):
            for bx in range(gx):
                tx, ty = _tile_of_block(bx, by, gx, gy)
                assert 0 <= tx < gx and 0 <= ty < gy
                seen.add((tx, ty))
                runs.setdefault((by * gx + bx) % 8, []).append(ty * gx + tx)
        assert len(seen) == gx * gy, (gx, gy)                       # every tile computed exactly once
        for tiles in runs.values():                                 # one XCD = one contiguous run of tiles
            assert tiles == list(range(tiles[0], tiles[0] + len(tiles))), (gx, gy)


def _igemm_tile(bx, by, bz, gx, gy, gz):
    """Python mirror of csrc/igemm.hip::xcd_tile: XCD-contiguous runs, phase / split-K slice (z) fastest inside a run."""
    n, orig = gx * gy * gz, (bz * gy + by) * gx + bx
    q, r, xcd, idx = n // 8, n % 8, orig % 8, orig // 8
    w = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
    rem = w // gz
    return rem % gx, rem // gx, w % gz


def test_igemm_tile_order_is_a_bijection_that_spreads_the_phases_over_the_xcds():
    """A stride-2 data gradient's four phases reduce over 1, 2, 2 and 4 taps: every XCD must get its share of each."""
    for gx, gy, gz in ((2, 120, 4), (4, 30, 4), (1, 480, 1), (7, 30, 2), (3, 5, 4), (9, 30, 1), (1, 1, 4), (4, 8, 16)):
        seen, per_xcd = set(), {}
        for bz in range(gz):
            for by in range(gy):
                for bx in range(gx):
                    t = _igemm_tile(bx, by, bz, gx, gy, gz)
                    assert 0 <= t[0] < gx and 0 <= t[1] < gy and 0 <= t[2] < gz
                    seen.add(t)
                    per_xcd.setdefault(((bz * gy + by) * gx + bx) % 8, []).append(t[2])
        assert len(seen) == gx * gy * gz, (gx, gy, gz)
        for zs in per_xcd.values():
            counts = [zs.count(z) for z in range(gz)]
            assert max(counts) - min(counts) <= 1, (gx, gy, gz, counts)
