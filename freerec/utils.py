"""`freerec.utils`: loggers, timing, pickles (DeepFM/main.py:259 `debugLogger`; the Coach's own logging)."""
import logging
import os
import pickle
import time
from contextlib import contextmanager

_LOG = logging.getLogger("freerec")
if not _LOG.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("[%(asctime)s] %(message)s", "%H:%M:%S"))
    _LOG.addHandler(_h)
    _LOG.setLevel(os.environ.get("FREEREC_LOG", "WARNING"))


def infoLogger(msg):
    _LOG.info(msg)
    return msg


def debugLogger(msg):
    _LOG.debug(msg)
    return msg


def warnLogger(msg):
    _LOG.warning(msg)
    return msg


def mkdirs(*paths):
    for p in paths:
        os.makedirs(p, exist_ok=True)


def export_pickle(obj, path):
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def import_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def timemeter(fn_or_name="block"):
    """`@freerec.utils.timemeter` on a function (its call sites in the reference), or `with timemeter("name"):`."""
    if callable(fn_or_name):
        import functools
        fn = fn_or_name

        @functools.wraps(fn)
        def wrapper(*a, **k):
            t0 = time.time()
            out = fn(*a, **k)
            infoLogger(f"[Wall TIME] >>> {fn.__qualname__} takes {time.time() - t0:.6f} seconds ...")
            return out
        return wrapper

    @contextmanager
    def cm():
        t0 = time.time()
        yield
        infoLogger(f"[Wall TIME] >>> {fn_or_name} takes {time.time() - t0:.6f} seconds ...")
    return cm()


def set_seed(seed):
    import random

    import numpy as np
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed
