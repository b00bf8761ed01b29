from . import plot
from .plot import FigureSpec, plot_dec_space, plot_obj_space_1d, plot_obj_space_2d, plot_obj_space_3d

__all__ = ["plot", "FigureSpec", "plot_dec_space", "plot_obj_space_1d", "plot_obj_space_2d", "plot_obj_space_3d"]
