"""Figures of a run: decision-space animation, fitness statistics over generations, and
objective-space animations for two and three objectives.
API parity: reference ``src/evox/vis_tools/plot.py:6-577``.

Figures are assembled as plain dictionaries in the plotly JSON schema (``data`` / ``layout`` /
``frames``).  When the ``plotly`` package is importable the functions return a
``plotly.graph_objects.Figure`` built from that dictionary; otherwise they return a
:class:`FigureSpec`, which can be dumped to JSON or to a self-contained HTML page that loads
plotly.js in the browser."""
from __future__ import annotations

import json

import numpy as np
import torch

_PLAY = {"frame": {"duration": 200, "redraw": False}, "fromcurrent": True,
         "transition": {"duration": 200, "easing": "linear"}, "mode": "immediate"}
_PAUSE = {"frame": {"duration": 0, "redraw": False}, "mode": "immediate", "transition": {"duration": 0}}
_LEGEND = {"x": 1, "y": 1, "xanchor": "auto"}
_MARGIN = {"l": 0, "r": 0, "t": 0, "b": 0}
_BLUE, _RED = "#636EFA", "#EF553B"


class FigureSpec:
    """A plotly-schema figure without the plotly dependency."""

    def __init__(self, spec):
        self.spec = spec

    @property
    def data(self):
        return self.spec["data"]

    @property
    def layout(self):
        return self.spec["layout"]

    @property
    def frames(self):
        return self.spec.get("frames", [])

    def to_dict(self):
        return self.spec

    def to_json(self):
        return json.dumps(self.spec)

    def write_html(self, path):
        page = ("<html><head><script src='https://cdn.plot.ly/plotly-2.32.0.min.js'></script></head><body>"
                "<div id='fig'></div><script>const s = %s; Plotly.newPlot('fig', s.data, s.layout)"
                ".then(() => Plotly.addFrames('fig', s.frames || []));</script></body></html>" % self.to_json())
        with open(path, "w") as f:
            f.write(page)


def _finish(spec):
    try:
        import plotly.graph_objects as go
    except ImportError:
        return FigureSpec(spec)
    return go.Figure(spec)


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().float().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


def _lst(a):
    return [float(v) for v in np.asarray(a).reshape(-1)]


def _padded_range(values):
    lo, hi = float(np.min(values)), float(np.max(values))
    pad = 0.1 * (hi - lo)
    return [lo - pad, hi + pad]


def _slider(n):
    steps = [{"label": i, "method": "animate",
              "args": [[str(i)], {"frame": {"duration": 200, "redraw": False}, "mode": "immediate",
                                  "transition": {"duration": 200}}]} for i in range(n)]
    return [{"currentvalue": {"prefix": "Generation: "}, "pad": {"b": 1, "t": 10}, "len": 0.8, "x": 0.2, "y": 0,
             "yanchor": "top", "xanchor": "left", "steps": steps}]


def _buttons():
    return [{"type": "buttons", "x": 0.2, "xanchor": "right", "y": 0, "yanchor": "top", "direction": "left",
             "pad": {"r": 10, "t": 30},
             "buttons": [{"args": [None, _PLAY], "label": "Play", "method": "animate"},
                         {"args": [[None], _PAUSE], "label": "Pause", "method": "animate"}]}]


def _animated(frames, layout, kwargs):
    layout = dict(legend=_LEGEND, margin=_MARGIN, sliders=_slider(len(frames)), updatemenus=_buttons(), **layout)
    layout.update(kwargs)
    return _finish({"data": frames[0]["data"], "layout": layout, "frames": frames})


def plot_dec_space(population_history, **kwargs):
    """Animated scatter of the first two decision variables, one frame per generation."""
    hist = [_np(p) for p in population_history]
    everything = np.concatenate(hist, axis=0)
    frames = [{"name": str(i), "data": [{"type": "scatter", "x": _lst(p[:, 0]), "y": _lst(p[:, 1]), "mode": "markers",
                                         "marker": {"color": _BLUE}}]} for i, p in enumerate(hist)]
    return _animated(frames, {"xaxis": {"range": _padded_range(everything[:, 0])},
                              "yaxis": {"range": _padded_range(everything[:, 1])}}, kwargs)


def _fitness_stats(fitness_history):
    hist = [_np(f).reshape(-1) for f in fitness_history]
    return {"Min": [float(h.min()) for h in hist], "Max": [float(h.max()) for h in hist],
            "Median": [float(np.median(h)) for h in hist], "Average": [float(h.mean()) for h in hist]}


def plot_obj_space_1d(fitness_history, animation=True, **kwargs):
    return (plot_obj_space_1d_animation if animation else plot_obj_space_1d_no_animation)(fitness_history, **kwargs)


def plot_obj_space_1d_no_animation(fitness_history, **kwargs):
    stats = _fitness_stats(fitness_history)
    gens = list(range(len(fitness_history)))
    data = [{"type": "scatter", "x": gens, "y": ys, "mode": "lines", "name": name} for name, ys in stats.items()]
    layout = dict(legend=_LEGEND, margin=_MARGIN)
    layout.update(kwargs)
    return _finish({"data": data, "layout": layout})


def plot_obj_space_1d_animation(fitness_history, **kwargs):
    """The four statistics curves grow generation by generation; the current population's
    fitness is overlaid as a column of markers."""
    stats = _fitness_stats(fitness_history)
    hist = [_np(f).reshape(-1) for f in fitness_history]
    n = len(hist)
    frames = []
    for i in range(n):
        gens = list(range(i + 1))
        data = [{"type": "scatter", "x": gens, "y": ys[: i + 1], "mode": "lines", "name": name}
                for name, ys in stats.items()]
        data.append({"type": "scatter", "x": [i] * len(hist[i]), "y": _lst(hist[i]), "mode": "markers",
                     "name": "Population", "marker": {"color": _BLUE, "opacity": 0.4}})
        frames.append({"name": str(i), "data": data})
    allv = np.concatenate(hist)
    return _animated(frames, {"xaxis": {"range": [0, max(n - 1, 1)]}, "yaxis": {"range": _padded_range(allv)}}, kwargs)


def _sorted_points(f, sort_points):
    if sort_points:
        order = np.lexsort(tuple(f[:, j] for j in reversed(range(f.shape[1]))))
        f = f[order]
    return f


def plot_obj_space_2d(fitness_history, problem_pf=None, sort_points=False, **kwargs):
    hist = [_sorted_points(_np(f), sort_points) for f in fitness_history]
    everything = np.concatenate(hist, axis=0)
    pf = _np(problem_pf) if problem_pf is not None else None
    if pf is not None:
        everything = np.concatenate([everything, pf], axis=0)
    frames = []
    for i, f in enumerate(hist):
        data = []
        if pf is not None:
            data.append({"type": "scatter", "x": _lst(pf[:, 0]), "y": _lst(pf[:, 1]), "mode": "markers",
                         "name": "Pareto Front", "marker": {"color": _RED, "size": 2}})
        data.append({"type": "scatter", "x": _lst(f[:, 0]), "y": _lst(f[:, 1]), "mode": "markers", "name": "Population",
                     "marker": {"color": _BLUE}})
        frames.append({"name": str(i), "data": data})
    return _animated(frames, {"xaxis": {"range": _padded_range(everything[:, 0])},
                              "yaxis": {"range": _padded_range(everything[:, 1])}}, kwargs)


def plot_obj_space_3d(fitness_history, sort_points=False, problem_pf=None, **kwargs):
    hist = [_sorted_points(_np(f), sort_points) for f in fitness_history]
    everything = np.concatenate(hist, axis=0)
    pf = _np(problem_pf) if problem_pf is not None else None
    if pf is not None:
        everything = np.concatenate([everything, pf], axis=0)
    frames = []
    for i, f in enumerate(hist):
        data = []
        if pf is not None:
            data.append({"type": "scatter3d", "x": _lst(pf[:, 0]), "y": _lst(pf[:, 1]), "z": _lst(pf[:, 2]),
                         "mode": "markers", "name": "Pareto Front", "marker": {"color": _RED, "size": 1}})
        data.append({"type": "scatter3d", "x": _lst(f[:, 0]), "y": _lst(f[:, 1]), "z": _lst(f[:, 2]), "mode": "markers",
                     "name": "Population", "marker": {"color": _BLUE, "size": 2}})
        frames.append({"name": str(i), "data": data})
    scene = {"xaxis": {"range": _padded_range(everything[:, 0])}, "yaxis": {"range": _padded_range(everything[:, 1])},
             "zaxis": {"range": _padded_range(everything[:, 2])}, "aspectmode": "cube"}
    return _animated(frames, {"scene": scene}, kwargs)
