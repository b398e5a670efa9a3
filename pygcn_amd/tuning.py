"""Dispatch thresholds of the host code — ONE table, each with the measurement it came from.

Nothing here changes a result: every threshold chooses between two routes that compute the same
sums (tests/test_fused_gpu.py::test_every_fallback_route_matches_the_oracle takes each of them).
All were measured on MI355X at config C4 (10⁷ vertices, 1.09·10⁸ entries, F = 256 fp32) unless
stated; device-side twins live in gcn_spmm.hip (`use_row_flags`).

| name | value | decides | measured |
|---|---|---|---|
| HINT_WIDE_MAX_SHARE   | 3/4 | wide kernel (rows ≥ 528 B): skip the flagged-zero rows of the dense operand below this share of non-zero rows | probing the bitmap costs 3 % of a launch; at ≥ 3/4 non-zero rows the skipped gathers no longer pay for it (round 1, `tools/grad_sparsity_probe.py`) |
| HINT_NARROW_MAX_SHARE | 1/8 | narrow kernel: the same, through lane compaction | the `mbcnt` + `ds_permute` compaction costs ≈ 12 % of a launch (C5, bf16 F = 128) |
| SPARSE_GEMM_MAX_SHARE | 1/3 | layer-by-layer path with row compaction ON: run the gradient GEMMs on a row list | list build (nonzero + padding) ≈ 0.5 ms + gather-fused GEMM at 1/3 of the rows ≈ full-height GEMM |
| SPARSE_FLAGS_MAX_SHARE | 1/8 | same path: ask the transpose product for per-row output flags | the flag bytes cost one store per row; below 1/8 the compacted GEMMs repay it |
| REASSOC_MAX_WIDTH_RATIO | 2 | first layer as (Â·X)·W (or the restricted product (Â·X)[R2]) only while Fin ≤ 2·Fout | the product gathers rows of width Fin instead of Fout: at 1433 → 16 (Cora) the transpose product is 90× cheaper |
| MIN_ROWS | 2¹⁷ | row compaction / K-split weight gradient only from this many vertices | below ≈ 10⁵ rows every GEMM of the step is launch-bound (≤ 20 µs): C2 / C3-sized probes |
| K_SPLIT | 128 | slabs of the hipBLASLt weight-gradient fallback | `tools/gemm_probe.py`: 8.7 ms vs 21.5 ms stream-K at N = 10⁷ |
| LONG_THRESH_BF16 | 1 024 | chunk length of long rows when the dense operand is stored in bf16 (fp32 storage keeps the C-ABI default, 256: the sequential fp32 chain of a chunk bounds the rounding error of the 1e-5 contract — at 1 024 the Â·1 = 1 residual at C4 is 1.4·10⁻⁵) | C5 forward product 41.5 → 38.9–39.7 ms (`bench.py --config c5 --spmm-only --long-thresh …`: 512: 39.9–40.3, 2 048: 39.3, 16 384: 39.2): 4× fewer chunk partials (1.7 GB of fp32 slabs written and re-read at 256) and long rows (7.4·10⁵ at 256); bf16 storage rounds to 2⁻⁸ anyway |
| ROWGRAD_MIN_ROWS | 16 384 | `model(x, adj)` runs as one autograd node and returns a RowSelectable (structural `output[idx]` gradient) from this many vertices; below, the plain layer-by-layer composition | Cora-sized epochs are launch-bound (C2: 0.85 ms layer-by-layer against 1.04 ms through the node's generic fallbacks for 1433 → 16 → 7); the wrapper subclass costs ≈ 30 µs of dispatch |
"""
from fractions import Fraction

HINT_WIDE_MAX_SHARE = Fraction(3, 4)
HINT_NARROW_MAX_SHARE = Fraction(1, 8)
SPARSE_GEMM_MAX_SHARE = Fraction(1, 3)
SPARSE_FLAGS_MAX_SHARE = Fraction(1, 8)
REASSOC_MAX_WIDTH_RATIO = 2
MIN_ROWS = 1 << 17
K_SPLIT = 128
ROWGRAD_MIN_ROWS = 16384
LONG_THRESH_BF16 = 1024


def below(count, total, share):
    """count / total < share, in integers (no rounding at the boundary)."""
    return count * share.denominator < total * share.numerator
