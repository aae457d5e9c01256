"""Host-side stand-in for the reference's `mppi_numba/density.py` (sample-based traction
densities used by the notebooks to make up terrain models; reference density.py:8-107).

Not on the hot path and not a device module: it exists so that
`from mppi_numba.density import Density, GaussianMixture` works where the reference is not
checked out (mppi_numba/__init__.py prefers the reference's own file when it finds one).
Same constructor arguments, attributes and return values; the sampling is vectorised, so
the stream of numpy random numbers consumed differs from the reference's one-draw-at-a-time
loop (the notebooks do not seed it).
"""
import numpy as np


class Density(object):
    """A scalar density known through a sampler.  Statistics are estimated from a cached
    batch of `num_samples` draws, taken on first use."""

    def __init__(self, sample_bounds, pmf_bounds, sample_fn, num_samples=1e4):
        self.sample_bounds = sample_bounds  # support of the sampler
        self.pmf_bounds = pmf_bounds        # range over which get_pmf() bins
        self.sample_fn = sample_fn
        self.num_samples = num_samples
        self.samples = None
        self.sample_initialized = False

    # -- sampling -------------------------------------------------------------------
    def sample(self, num):
        return self.sample_fn(num)

    def initialize_samples(self, num_samples):
        self.samples = self.sample(num_samples)
        self.sample_initialized = True

    def _cached(self, samples=None):
        if samples is not None:
            return np.asarray(samples)
        if not self.sample_initialized:
            self.initialize_samples(self.num_samples)
        return self.samples

    # -- statistics -----------------------------------------------------------------
    def mean(self, samples=None):
        return np.mean(self._cached(samples))

    def var(self, samples=None):
        return np.var(self._cached(samples))

    def cvar(self, alpha, front=True, samples=None):
        """(mean of the worst `alpha` tail, its threshold): lower tail when `front`."""
        assert 0.0 < alpha <= 1.0, "Alpha must be in (0,1]"
        data = self._cached(samples)
        thres = np.percentile(data, 100.0 * (alpha if front else 1.0 - alpha))
        tail = data[data < thres] if front else data[data > thres]
        assert tail.size > 0
        return np.mean(tail), thres

    def get_pmf(self, num_bins, include_min_max=True):
        """Histogram PMF over `pmf_bounds`: (bin centre values, masses summing to 1).  With
        `include_min_max` a zero-mass bin is added at each bound (the planner wants exact 0
        and nominal tractions to be representable)."""
        data = self._cached()
        lo, hi = self.pmf_bounds
        counts, _ = np.histogram(data, num_bins, range=(lo, hi), density=True)
        width = (hi - lo) / num_bins
        values = np.arange(lo, hi, width) + width / 2
        if include_min_max:
            values = np.concatenate(([lo], values, [hi]))
            counts = np.concatenate(([0.0], counts, [0.0]))
        return values, counts / np.sum(counts)


class GaussianMixture(Density):
    """Mixture of Gaussians truncated to `sample_bounds` (rejection sampling)."""

    def __init__(self, sample_bounds, pmf_bounds, weights, means, stds, num_samples=1e3):
        assert sum(weights) == 1
        assert len(weights) == len(means) == len(stds)
        assert len(sample_bounds) == 2 and len(pmf_bounds) == 2
        assert sample_bounds[1] >= sample_bounds[0] and pmf_bounds[1] >= pmf_bounds[0]
        assert pmf_bounds[0] <= sample_bounds[0] and pmf_bounds[1] >= sample_bounds[1]
        self.num_components = len(weights)
        w = np.asarray(weights, dtype=float)
        mu = np.asarray(means, dtype=float)
        sd = np.asarray(stds, dtype=float)

        def sample_fn(num):
            num = int(num)
            kept = np.empty(0)
            while kept.size < num:
                want = max(64, 2 * (num - kept.size))
                comp = np.random.choice(self.num_components, size=want, p=w)
                draw = np.random.normal(mu[comp], sd[comp])
                ok = (draw >= sample_bounds[0]) & (draw <= sample_bounds[1])
                kept = np.concatenate((kept, draw[ok]))
            return kept[:num]

        super().__init__(sample_bounds, pmf_bounds, sample_fn, num_samples)
