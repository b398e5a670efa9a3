"""Edge-case table shared by make_golden.py (generator) and the tests (data only)."""
EDGE_CASES = [
    # name, n_rows, n_cols, nnz, fin, fout, bias, kwargs for inputs.random_coo
    ("tiny_f1", 17, 17, 40, 3, 1, True, {}),
    ("f7", 50, 50, 300, 16, 7, True, {}),
    ("f16_empty_rows", 64, 64, 500, 9, 16, True, {"empty_rows": (0, 5, 6, 63)}),
    ("f100_dups", 80, 80, 600, 12, 100, True, {"duplicates": 200}),
    ("f256_hub", 300, 300, 1500, 20, 256, True, {"hub_row": 7, "hub_deg": 1200}),
    ("f256_nobias", 128, 128, 900, 33, 256, False, {}),
    ("rect_f64", 40, 90, 350, 5, 64, True, {}),
    ("f300_wide", 70, 70, 500, 6, 300, True, {}),
    ("all_empty", 10, 10, 0, 4, 8, True, {}),
]
