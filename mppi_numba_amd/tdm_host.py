"""Pure-numpy map preprocessing used by TDM_Numba (no GPU, no C library): the
part of /root/reference/mppi_numba/terrain.py that the reference also runs on
the host, once per map change (terrain.py:408-495 PMF -> one-hot CVaR bin /
risk traction map; terrain.py:511-583 zero-traction padding).  Kept separate
so that it can be tested against the golden fixtures on a machine without a GPU.
"""
import numpy as np


def padding_info(grid_shape, max_map_dim, max_speed_padding, dt, res):
    """(valid_rows, valid_cols, pad_cells) for a grid whose last two dims are
    (rows, cols): ring width ceil(max_speed_padding*dt/res), and how much of the
    grid fits max_map_dim once the ring is added (terrain.py:562-583)."""
    rows, cols = grid_shape[-2], grid_shape[-1]
    pad_cells = int(np.ceil(max_speed_padding * dt / res))
    max_rows = max_map_dim[0] - 2 * pad_cells
    max_cols = max_map_dim[1] - 2 * pad_cells
    return min(max_rows, rows), min(max_cols, cols), pad_cells, max_rows, max_cols


def padded_limits(xlimits, ylimits, valid_rows, valid_cols, pad_cells, res):
    px = np.array([xlimits[0] - pad_cells * res, xlimits[0] + (valid_cols + pad_cells) * res])
    py = np.array([ylimits[0] - pad_cells * res, ylimits[0] + (valid_rows + pad_cells) * res])
    return px, py


def pad_pmf(pmf_grid, valid_rows, valid_cols, pad):
    """Ring of cells with all mass in bin 0 (zero traction) around the PMF grid."""
    bins = pmf_grid.shape[0]
    out = np.zeros((bins, valid_rows + 2 * pad, valid_cols + 2 * pad), dtype=np.int8)
    out[0] = np.int8(100)
    out[:, pad:pad + valid_rows, pad:pad + valid_cols] = pmf_grid[:, :valid_rows, :valid_cols]
    return out


def pad_layer(grid, valid_rows, valid_cols, pad):
    """Zero ring around a (1, rows, cols) int8 layer (risk traction map)."""
    out = np.zeros((1, valid_rows + 2 * pad, valid_cols + 2 * pad), dtype=np.int8)
    out[:, pad:pad + valid_rows, pad:pad + valid_cols] = grid[:, :valid_rows, :valid_cols]
    return out


def pad_mask(mask, valid_rows, valid_cols, pad, pad_val=0):
    out = pad_val * np.ones((valid_rows + 2 * pad, valid_cols + 2 * pad), dtype=np.int8)
    out[pad:pad + valid_rows, pad:pad + valid_cols] = mask[:valid_rows, :valid_cols]
    return out


def _cumulative(pmf_grid, bin_values_f32):
    cum = 0.01 * pmf_grid.cumsum(axis=0).astype(float)  # reaches 1.0
    weighted_cum = np.cumsum(0.01 * pmf_grid.astype(float) * bin_values_f32.reshape((-1, 1, 1)), axis=0)
    return cum, weighted_cum


def cvar_traction(pmf_grid, bin_values_f32, alpha):
    """Mean of the worst alpha fraction of each cell's traction PMF (plain mean
    for alpha == 1, which the reference special-cases because a float cumsum
    may stop short of 1.0; terrain.py:426-441)."""
    _, rows, cols = pmf_grid.shape
    cum, weighted_cum = _cumulative(pmf_grid, bin_values_f32)
    if alpha == 1.0:
        return weighted_cum[-1]
    rr, cc = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    upto = np.argmax(cum >= alpha, axis=0)
    return weighted_cum[upto, rr, cc] / (cum[upto, rr, cc] + 1e-6)


def one_hot_cvar_pmf(pmf_grid, bin_values_f32, alpha):
    """use_det_dynamics: PMF with all mass (100) in the first bin whose value is
    >= the CVaR_alpha traction of the cell (terrain.py:408-452)."""
    bins, rows, cols = pmf_grid.shape
    target = cvar_traction(pmf_grid, bin_values_f32, alpha)
    which = np.argmax(target <= bin_values_f32.reshape((-1, 1, 1)), axis=0)
    rr, cc = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    out = np.zeros((bins, rows, cols), dtype=np.int8)
    out[which, rr, cc] = np.int8(100)
    return out


def risk_traction_map(pmf_grid, bin_values_f32, bounds_f32, alpha):
    """use_nom_dynamics_with_speed_map: int8 (1, rows, cols) map of
    100*(CVaR_alpha traction - lo)/(hi - lo), truncated (terrain.py:470-491)."""
    _, rows, cols = pmf_grid.shape
    target = cvar_traction(pmf_grid, bin_values_f32, alpha)
    span = bounds_f32[1] - bounds_f32[0]
    if alpha == 1.0:
        # the reference scales the mean as (100*(mean - lo))/range (terrain.py:476-478) but the
        # CVaR as 100*((cvar - lo)/range) (terrain.py:488-490): after truncation they can differ by one
        scaled = 100 * (target - bounds_f32[0]) / span
    else:
        scaled = 100 * np.asarray((target - bounds_f32[0]) / span)
    return np.reshape(scaled, (1, rows, cols)).astype(np.int8)


def bin_table(bin_values_dev, bounds_dev):
    """int8 traction value written for each PMF bin by the sampling kernel:
    np.int8(100.*(bin_values[b]-lo)/(hi-lo)), truncating, evaluated in the dtype
    of the device copies (terrain.py:689)."""
    span = bounds_dev[1] - bounds_dev[0]
    return np.array([np.int8(int(100.0 * float(bin_values_dev[b] - bounds_dev[0]) / float(span)))
                     for b in range(len(bin_values_dev))], dtype=np.int8)


def traction_scale(bounds_dev):
    """(lo, ratio): traction = lo + ratio*int8, ratio = 0.01*(hi-lo) with the
    subtraction in the array dtype and the product in float64 (mppi.py:674-675)."""
    return float(bounds_dev[0]), float(0.01 * float(bounds_dev[1] - bounds_dev[0]))


def as_device_float(values):
    """The dtype a cuda.to_device of `values` would have had (float32/float64 kept,
    everything else float64)."""
    arr = np.asarray(values)
    return arr if arr.dtype in (np.float32, np.float64) else arr.astype(np.float64)


def cvar_bin_of_distribution(values, pmf, alpha):
    """Index of the first bin whose value is >= the mean of the worst `alpha` fraction of a
    (values, pmf) distribution; the plain mean for alpha == 1 (terrain.py:226-257).  None when
    no bin qualifies (the reference then trips its own assert)."""
    if alpha == 1.0:
        expected = 0.0
        for val, mass in zip(values, pmf):
            expected += mass * val
    else:
        cum, expected, hit = 0.0, 0.0, False
        for val, mass in zip(values, pmf):
            cum += mass
            expected += mass * val
            if cum >= alpha:
                if cum > 0:
                    expected /= cum
                hit = True
                break
        if not hit:
            return None
    for idx, val in enumerate(values):
        if expected <= val:
            return idx
    return None


def semantic_pmf_grid(semantic_grid, terrain_of_id, terrain2pmf, num_pmf_bins, bounds_f32, mode, alpha):
    """PMF grid (and, in speed-map mode, the unpadded int8 risk traction layer) of a grid
    of semantic ids (terrain.py:219-325).  mode: 'tdm' | 'det' | 'speed'.  Every cell of
    one terrain type gets the same column, so the work is per unique id."""
    rows, cols = semantic_grid.shape
    pmf_grid = np.zeros((num_pmf_bins, rows, cols), dtype=np.int8)
    ids = np.unique(semantic_grid)
    risk = None
    if mode == "det":
        for sid in ids:
            values, pmf = terrain2pmf[terrain_of_id(sid)]
            column = np.zeros(num_pmf_bins, dtype=np.int8)
            chosen = cvar_bin_of_distribution(values, pmf, alpha)
            if chosen is not None:
                column[chosen] = 100
            assert column.sum() == 100
            pmf_grid[:, semantic_grid == sid] = column.reshape(-1, 1)
    elif mode == "speed":
        pmf_grid[-1, :, :] = np.int8(100)
        layers = len(terrain2pmf[terrain_of_id(ids[0])][1])
        pmf_f = np.zeros((layers, rows, cols), dtype=float)  # sums to 1 along axis 0
        val_f = np.zeros((layers, rows, cols), dtype=float)
        for sid in ids:
            values, pmf = terrain2pmf[terrain_of_id(sid)]
            mask = semantic_grid == sid
            pmf_f[:, mask] = np.reshape(pmf, (layers, 1))
            val_f[:, mask] = np.reshape(values, (layers, 1))
        cum = pmf_f.cumsum(axis=0)
        wv_cum = np.cumsum(pmf_f * val_f, axis=0)
        span = bounds_f32[1] - bounds_f32[0]
        if alpha == 1.0:
            scaled = 100 * (wv_cum[-1] - bounds_f32[0]) / span
        else:
            layer = np.argmax(cum >= alpha, axis=0)
            rr, cc = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
            cvar = wv_cum[layer, rr, cc] / cum[layer, rr, cc]  # (no epsilon in this entry point)
            scaled = 100 * np.asarray((cvar - bounds_f32[0]) / span)
        risk = np.reshape(scaled, (1, rows, cols)).astype(np.int8)
    elif mode == "tdm":
        for sid in ids:
            values, pmf = terrain2pmf[terrain_of_id(sid)]
            column = np.int8(np.asarray(pmf) * 100)
            column[-1] = np.int8(100) - np.sum(column[:-1])
            assert column.sum() == 100
            pmf_grid[:, semantic_grid == sid] = column.reshape(-1, 1)
    else:
        raise AssertionError("TDM cannot be set up")
    return pmf_grid, risk
