from __future__ import annotations


class Gaussian:
    """Homoscedastic observation noise; ``num_datapoints`` is accepted for signature
    compatibility with the reference's call sites (``im_moea.py:286``)."""

    def __init__(self, num_datapoints=None, obs_stddev=1.0):
        self.num_datapoints = num_datapoints
        self.obs_stddev = float(obs_stddev)


class Bernoulli:
    """Probit-free logistic link for binary labels in {0, 1}."""

    def __init__(self, num_datapoints=None):
        self.num_datapoints = num_datapoints
