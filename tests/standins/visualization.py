"""Host-side stand-in for the reference's `mppi_numba/visualization.py` (matplotlib drawing
of a traction distribution map and of terrain densities; reference visualization.py:10-198).

Not on the hot path: it exists so that the notebooks' plotting cells run where the reference
is not checked out (mppi_numba/__init__.py prefers the reference's own file when present).
Same class / function names, arguments and return values; the cell patches are built with
numpy instead of per-cell Python lists.
"""
import copy

import numpy as np
import matplotlib.pyplot as plt
from matplotlib.collections import LineCollection, PolyCollection


class TDM_Visualizer(object):
    """Draws the (padded) semantic grid of a TDM_Numba: one coloured square per cell."""

    PREFERRED_MAX_FIG_WIDTH = 12
    PREFERRED_MAX_FIG_HEIGHT = 8
    PADDING_ID = -1

    def __init__(self, tdm, tdm_contains_semantic_grid=True):
        dims = tdm.get_padded_grid_xy_dim()
        assert dims is not None, "Cannot get padded grid dimension from TDM."
        self.num_rows, self.num_cols = (int(d) for d in dims)
        self.pad_width = tdm.pad_cells
        self.cell_dimensions = copy.deepcopy(tdm.cell_dimensions)
        self.xlimits = copy.deepcopy(tdm.padded_xlimits)
        self.ylimits = copy.deepcopy(tdm.padded_ylimits)
        self.num_pmf_bins = copy.deepcopy(tdm.num_pmf_bins)
        self.bin_values = copy.deepcopy(tdm.bin_values)
        self.bin_values_bounds = copy.deepcopy(tdm.bin_values_bounds)
        self.semantic_grid_initialized = False
        if tdm_contains_semantic_grid:
            self.semantic_grid_initialized = tdm.semantic_grid_initialized
            self.id2name = copy.deepcopy(tdm.id2name)
            self.name2terrain = copy.deepcopy(tdm.name2terrain)
            self.id2terrain_fn = copy.deepcopy(tdm.id2terrain_fn)
            self.terrain2pmf = copy.deepcopy(tdm.terrain2pmf)
            self.id2rgb = {sid: self.id2terrain_fn(sid).rgb for sid in self.id2name}
            self.id2name[self.PADDING_ID] = "Padding"
            self.id2rgb[self.PADDING_ID] = (0, 0, 0,)
            self.semantic_grid = self._padded(tdm.semantic_grid)

    def _padded(self, grid):
        """The grid (cropped like the TDM crops it) inside a ring of PADDING_ID cells."""
        p = self.pad_width
        out = np.full((self.num_rows, self.num_cols), float(self.PADDING_ID))
        out[p:self.num_rows - p, p:self.num_cols - p] = \
            np.asarray(grid)[:self.num_rows - 2 * p, :self.num_cols - 2 * p]
        return out

    # -- geometry ---------------------------------------------------------------------
    def cell_xy(self, ix, iy):
        """Centre of cell (ix, iy)."""
        w, h = self.cell_dimensions
        return self.xlimits[0] + (ix + 0.5) * w, self.ylimits[0] + (iy + 0.5) * h

    def cell_verts(self, ix, iy):
        w, h = self.cell_dimensions
        x, y = self.cell_xy(ix, iy)
        return [(x - 0.5 * w, y - 0.5 * h), (x - 0.5 * w, y + 0.5 * h),
                (x + 0.5 * w, y + 0.5 * h), (x + 0.5 * w, y - 0.5 * h)]

    def get_all_cell_verts(self, semantic_grid=None):
        grid = self.semantic_grid if semantic_grid is None else semantic_grid
        rows, cols = grid.shape
        w, h = self.cell_dimensions
        x0 = self.xlimits[0] + w * np.arange(cols)
        y0 = self.ylimits[0] + h * np.arange(rows)
        gx, gy = np.meshgrid(x0, y0)                       # row-major: iy outer, ix inner
        corners = np.array([(0, 0), (0, 1), (1, 1), (1, 0)], dtype=float) * (w, h)
        return np.stack((gx.reshape(-1, 1) + corners[:, 0], gy.reshape(-1, 1) + corners[:, 1]), axis=-1)

    def get_terrain_rgbs(self, id2rgb_map=None, semantic_grid=None):
        if id2rgb_map is None or semantic_grid is None:
            id2rgb_map, semantic_grid = self.id2rgb, self.semantic_grid
        return [id2rgb_map[sid] for sid in np.asarray(semantic_grid).reshape(-1)]

    def calc_auto_figsize(self, xlimits, ylimits):
        width, height = xlimits[1] - xlimits[0], ylimits[1] - ylimits[0]
        if width > height:
            return (self.PREFERRED_MAX_FIG_WIDTH, height * self.PREFERRED_MAX_FIG_WIDTH / width)
        return (width * self.PREFERRED_MAX_FIG_HEIGHT / height, self.PREFERRED_MAX_FIG_HEIGHT)

    # -- drawing ----------------------------------------------------------------------
    def draw_base_grid(self, figsize, ax=None):
        """Cell borders; creates the figure unless an axis is given.  Returns (fig, ax)."""
        w, h = self.cell_dimensions
        xs = self.xlimits[0] + w * np.arange(self.num_cols + 1)
        ys = self.ylimits[0] + h * np.arange(self.num_rows + 1)
        horizontal = [[(xs[0], y), (xs[-1], y)] for y in ys]
        vertical = [[(x, ys[0]), (x, ys[-1])] for x in xs]
        if ax is None:
            fig, ax = plt.subplots(figsize=figsize)
        else:
            fig = plt.gcf()
        ax.add_collection(LineCollection(horizontal + vertical, color="black", linewidths=0.5, alpha=0.5))
        ax.set_xlim(xs[0] - 1, xs[-1] + 1)
        ax.set_ylim(ys[0] - 1, ys[-1] + 1)
        ax.set_aspect('equal', adjustable='box')
        ax.axis('off')
        return fig, ax

    def draw_semantic_patches(self, ax, semantic_grid=None, id2rgb_map=None):
        if semantic_grid is None or id2rgb_map is None:
            semantic_grid, id2rgb_map = None, None
        ax.add_collection(PolyCollection(self.get_all_cell_verts(semantic_grid=semantic_grid),
                                         facecolors=self.get_terrain_rgbs(id2rgb_map, semantic_grid)))

    def draw(self, figsize=(10, 10), ax=None, semantic_grid=None, id2rgb_map=None):
        if not self.semantic_grid_initialized and semantic_grid is None and id2rgb_map is None:
            print("Semantic grid not initialized. Cannot invoke draw() function")
            return
        if figsize is None and ax is None:
            figsize = self.figsize = self.calc_auto_figsize(self.xlimits, self.ylimits)
        fig, ax = self.draw_base_grid(figsize, ax=ax)
        if self.semantic_grid_initialized:
            self.draw_semantic_patches(ax)
        elif semantic_grid is not None and id2rgb_map is not None:
            self.draw_semantic_patches(ax, semantic_grid=self._padded(semantic_grid), id2rgb_map=id2rgb_map)
        else:
            print("Colors not shown as semantic grid is not initialized.")
        return fig, ax


def vis_density(ax, density, terrain, vis_cvar_alpha=0.3, show_cvar=False, color='b', show_legend=True,
                title=None, hist_alpha=0.5, fontsize=12):
    """Histogram of a density's cached samples (and optionally its lower-tail threshold)."""
    _, thres = density.cvar(alpha=vis_cvar_alpha)
    if density.sample_initialized:
        ax.hist(density.samples, bins=100, density=True, color=color, alpha=hist_alpha, label=terrain.name)
    if show_cvar:
        ax.plot([thres, thres], [0, 5], 'k--', linewidth=2,
                label='{}-th Percentile'.format(int(vis_cvar_alpha * 100.0)))
    if density.sample_bounds is not None:
        ax.set_xlim(density.sample_bounds)
    if title is not None:
        ax.set_title(title, fontsize=fontsize)
    ax.set_xlabel("Traction", fontsize=fontsize)
    ax.set_ylabel("Density", fontsize=fontsize)
    if show_legend:
        ax.legend(fontsize=fontsize)
    return ax


def vis_density_as_pmf(ax, density, terrain, num_bins, include_min_max=True, color='b', title=None,
                       hist_alpha=0.5):
    """Stem plot of the binned PMF the planner consumes."""
    values, pmf = density.get_pmf(num_bins=num_bins, include_min_max=include_min_max)
    markers, stems, base = ax.stem(values, pmf, label=terrain.name)
    markers.set_color(color)
    stems.set_color(color)
    base.set_color('r')
    if density.pmf_bounds is not None:
        ax.set_xlim(density.pmf_bounds)
    if title is not None:
        ax.set_title(title)
    ax.set_xlabel("Traction")
    ax.set_ylabel("PMF")
    ax.legend()
    return ax
