"""Host-side helpers the reference's drivers import next to the model (`from utils import EMA, rmse, mae, sd, pearson`,
main_qm9.py:14, main_pdbbind.py:14).  The basis generators of the reference's utils/sbf.py have no counterpart here: the
basis constants are compiled into the HIP library (csrc/basis_constants.h)."""
from .ema import EMA
from .metrics import mae, pearson, rmse, sd

__all__ = ["EMA", "rmse", "mae", "sd", "pearson"]
