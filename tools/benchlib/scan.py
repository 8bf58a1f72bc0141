"""AV1 default (zig-zag) coefficient scan of a w x h transform block, generated instead of stored: the order
`av1_scan_orders[tx_size][DCT_DCT]` of the reference (coefficients.h:2197 -> default_scan_NxN, e.g. :86, :228) walks the
anti-diagonals of the block alternately down-left and up-right, starting 0, 1, w, 2w, w+1, 2, ...
tests/test_bench_inputs.py pins this generator to the reference's tables (where the reference build is present)."""
import numpy as np


def zigzag_scan(w, h):
    """scan[i] = raster position of the i-th coefficient; iscan[pos] = its scan index (int16, as the reference's)."""
    order = []
    for d in range(w + h - 1):
        cells = [(d - c, c) for c in range(max(0, d - h + 1), min(w - 1, d) + 1)]    # (row, col) with col ascending
        if d % 2 == 1:
            cells.reverse()                                                           # odd diagonals run down-left
        order += [r * w + c for r, c in cells]
    scan = np.array(order, dtype=np.int16)
    iscan = np.empty_like(scan)
    iscan[scan] = np.arange(scan.size, dtype=np.int16)
    return scan, iscan
