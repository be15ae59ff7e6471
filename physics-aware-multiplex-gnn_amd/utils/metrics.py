"""Evaluation metrics of the PDBbind driver (reference utils/metrics.py:6-23; used at main_pdbbind.py:37-39):
y = measured affinities, f = predictions, both 1-D numpy arrays.

  rmse    sqrt(mean((y-f)^2))
  mae     mean(|y-f|)
  sd      residual standard deviation of the least-squares line y ~ a*f + b:  sqrt(sum((y - (a f + b))^2) / (n-1))
  pearson Pearson correlation coefficient of (y, f)

`sd` is evaluated in closed form (the reference fits sklearn's LinearRegression, which solves the same normal
equations); no sklearn dependency."""
import math

import numpy as np


def _flat(a):
    return np.asarray(a, dtype=np.float64).reshape(-1)


def rmse(y, f):
    y, f = _flat(y), _flat(f)
    return math.sqrt(float(np.mean((y - f) ** 2)))


def mae(y, f):
    y, f = _flat(y), _flat(f)
    return float(np.mean(np.abs(y - f)))


def sd(y, f):
    y, f = _flat(y), _flat(f)
    fc, yc = f - f.mean(), y - y.mean()
    var = float(fc @ fc)
    slope = float(fc @ yc) / var if var > 0.0 else 0.0
    resid = yc - slope * fc                      # intercept = mean(y) - slope * mean(f)
    return math.sqrt(float(resid @ resid) / (len(y) - 1))


def pearson(y, f):
    y, f = _flat(y), _flat(f)
    fc, yc = f - f.mean(), y - y.mean()
    return float(fc @ yc) / math.sqrt(float(fc @ fc) * float(yc @ yc))
