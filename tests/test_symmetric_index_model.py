"""Index model of the mirrored loads of the SYMMETRIC tile kernels (no GPU).

Three places read a row through a reflection: the first tile of the streaming forward and of k_inv_sym (left of the row
start, sample -1 - i is sample i) and the single-level right-looking inverse (right of the row end, sample t >= n is
sample 2n - 1 - t; modwt/MODWTTransform.java:644-657).  A tile covers positions far beyond what its stored outputs need
-- the tile that starts near the row end of a 2049-sample row reaches position 4102, whose mirror index is negative --
so every mirrored load is guarded.  This test restates the index arithmetic of those loads (the same expressions as in
csrc/vw_kernels.hip.h: k_fwd_stream / inv_fetch and csrc/vw_sym_kernels.hip.h: sym_fetch_pair) for every lane of every
tile of many row lengths and requires every index to lie inside the row; it also checks that the guarded positions
are exactly those no stored output depends on."""
import pytest

CHUNK = 512


def left_mirror_indices(t, vec2):
    """sym_fetch_pair / k_fwd_stream halo for a pair at t < 0 (t even): indices read."""
    assert t < 0 and t % 2 == 0
    return [-t - 2, -t - 1] if vec2 else [-t - 1, -t - 2]


def right_mirror_indices(t, n, vec2):
    """inv_fetch<kSymmetric> for the pair (t, t + 1), t even: indices read (guards included)."""
    idx = []
    if vec2:
        if t < n:
            idx += [t, t + 1]
        elif t <= 2 * n - 2:
            idx += [2 * n - 2 - t, 2 * n - 1 - t]
    else:
        if t < n:
            idx.append(t)
        elif t <= 2 * n - 1:
            idx.append(2 * n - 1 - t)
        if t + 1 < n:
            idx.append(t + 1)
        elif t <= 2 * n - 2:
            idx.append(2 * n - 2 - t)
    return idx


@pytest.mark.parametrize("n", [512, 513, 514, 600, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 2050, 4096, 4097, 5000, 16384])
@pytest.mark.parametrize("ppt", [1, 2, 4, 8])
def test_mirrored_loads_stay_inside_the_row(n, ppt):
    vec2 = n % 2 == 0
    tile = ppt * CHUNK
    ntiles = (n + tile - 1) // tile
    # left halo of the first tile (forward and k_inv_sym): 256 lanes, pair 2 * tid of the halo chunk
    for tid in range(256):
        for i in left_mirror_indices(-CHUNK + 2 * tid, vec2):
            assert 0 <= i < n
    # right-looking single-level inverse: tile + one halo chunk, limited to the filter's reach (at most 20 taps)
    for taps in (2, 8, 20):
        reach = (taps - 1 + 1) & ~1
        plimit = tile + reach
        for t_idx in range(ntiles):
            t0 = t_idx * tile
            for p in range(0, min(plimit, tile + CHUNK), 2):
                t = t0 + p
                for i in right_mirror_indices(t, n, vec2):
                    assert 0 <= i < n, (n, ppt, taps, t, i)
                # a position the guard skips feeds no stored output: outputs t' < n read up to t' + taps - 1 < 2n - 1
                if t > 2 * n - 1:
                    assert t - (taps - 1) >= n
