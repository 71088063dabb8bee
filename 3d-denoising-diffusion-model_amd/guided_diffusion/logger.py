"""
Minimal stand-in for the reference's logger (logger.py:442-471 `configure`,
`log`, `get_dir`): the only three calls the inference script makes
(scripts/test.py:23, :25, :169).  Rank-suffixed log file like logger.py:457-465.
"""

import os
import sys
import time

_state = {"dir": None, "file": None}


def configure(dir=None, **_ignored):
    if not dir:
        dir = os.path.join(os.getcwd(), "ddpm3d-" + time.strftime("%Y-%m-%d-%H-%M-%S"))
    os.makedirs(dir, exist_ok=True)
    rank = int(os.environ.get("RANK", "0"))
    name = "log.txt" if rank == 0 else "log-rank%03d.txt" % rank
    _state["dir"] = dir
    _state["file"] = open(os.path.join(dir, name), "a")


def get_dir():
    return _state["dir"]


def log(*args):
    line = " ".join(str(a) for a in args)
    print(line, file=sys.stdout, flush=True)
    if _state["file"] is not None:
        _state["file"].write(line + "\n")
        _state["file"].flush()
