"""Kernel-form options of libnerfacc_hip.so (include/nerfacc_hip.h: nfa_set_option / nfa_get_option).

Every form of a kernel gives bit-identical results; an option only forces which one serves a call (lanes
per ray of the count pass, where the grid image lives, the emit form, the tile plan of the streaming
kernels ...).  The table is process-wide and lives in the shared library, which both host faces (the torch
extension and the ctypes binding) load, so one call here steers either.  `NFA_<NAME>` environment variables
seed the table once at load time; nothing reads the environment per call.

    with nerfacc_amd.options(split_p=16, emit="rays"):
        estimator.sampling(...)

No reference counterpart (the reference has one form per kernel).
"""
from __future__ import annotations

import contextlib
import ctypes
from typing import Dict, Optional, Union

_EMIT_NAMES = {1: "rays", 2: "samples", 3: "tiles"}


def _lib():
    from .cuda._backend import load_library

    return load_library()


def _canon(name: str) -> str:
    n = name.lower()
    return n[4:] if n.startswith("nfa_") else n


def set_option(name: str, value: Union[None, int, str] = None) -> None:
    """force `name` to `value`; None / "" / "auto" hands the choice back to the library"""
    v = None if value is None else str(value).encode()
    if _lib().nfa_set_option(name.encode(), v) != 0:
        raise ValueError("nerfacc_amd: " + _lib().nfa_last_error().decode())


def get_option(name: str) -> Optional[Union[int, str]]:
    """the forced value of `name`, or None while the library chooses"""
    val, is_set = ctypes.c_int64(0), ctypes.c_int32(0)
    if _lib().nfa_get_option(name.encode(), ctypes.byref(val), ctypes.byref(is_set)) != 0:
        raise ValueError("nerfacc_amd: " + _lib().nfa_last_error().decode())
    if not is_set.value:
        return None
    return _EMIT_NAMES[val.value] if _canon(name) == "emit" else int(val.value)


def reset_options() -> None:
    """every option back to its state when the library was loaded (environment-seeded or automatic)"""
    _lib().nfa_reset_options()


def list_options() -> Dict[str, str]:
    """name -> one-line description of every option the library knows"""
    L = _lib()
    return {L.nfa_option_name(i).decode(): L.nfa_option_doc(i).decode() for i in range(L.nfa_option_count())}


@contextlib.contextmanager
def options(**forced):
    """force some options for the duration of a `with` block, then restore what was there before"""
    before = {k: get_option(k) for k in forced}
    try:
        for k, v in forced.items():
            set_option(k, v)
        yield
    finally:
        for k, v in before.items():
            set_option(k, v)


def release_workspace() -> None:
    """Free the calling host thread's retained traversal workspace (at most 64 MB per device and stream; larger
    calls allocate per call) and drop any count pass `sample_occgrid` launched ahead for the next ray slice."""
    from .cuda import _backend

    c = _backend._C
    if hasattr(c, "release_workspace"):
        c.release_workspace()
