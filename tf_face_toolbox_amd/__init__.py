"""tf_face_toolbox_amd -- MI355X-native engine for the data-parallel training step of
medivhna/TF_Face_Toolbox (data_parallel.py + nets/ + loss.py).  Host code mirrors the
reference's Python interface; every FLOP runs in libfte.so (hand-written HIP, gfx950)."""
from .nets.net_base import net_select, Network            # noqa: F401
from .data_parallel import Singular, DataParallel, DataParallel_margin   # noqa: F401

__version__ = '0.1'
