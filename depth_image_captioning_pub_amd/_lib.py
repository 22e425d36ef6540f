"""ctypes binding of libdic_hip.so (include/dic.h).  Fails loudly when the library is missing:
there is no CPU fallback anywhere in the product path."""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdic_hip.so")
LIB_EXPERIMENTS_PATH = os.path.join(HERE, "libdic_experiments.so")      # parked kernels + ablation switches: scripts/ only
HEADER = os.path.join(os.path.dirname(HERE), "include", "dic.h")

ABI_VERSION = 200       # DIC_ABI_VERSION of the include/dic.h these bindings were written against (load() refuses any other library)
_lib: Optional[C.CDLL] = None

c_fp = C.c_void_p       # device pointers travel as void* (tensor.data_ptr())
c_ll = C.c_longlong
c_sz = C.c_size_t


class DicError(RuntimeError):
    pass


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("DIC_LIB") == "experiments":      # development scripts (scripts/): same sources + -DDIC_EXPERIMENTS
        path = LIB_EXPERIMENTS_PATH
    if not os.path.exists(path):
        raise DicError(f"{path} is missing: build it with `python -m depth_image_captioning_pub_amd.build"
                       f"{' --experiments' if path != LIB_PATH else ''}` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.dic_version.restype = C.c_int
    lib.dic_last_error.restype = C.c_char_p
    got = lib.dic_version()
    if got != ABI_VERSION:      # argument lists / struct layouts differ between versions: calling on would pass garbage pointers
        raise DicError(f"{path} has ABI version {got}, these bindings need {ABI_VERSION} (include/dic.h DIC_ABI_VERSION): rebuild it "
                       "with `python -m depth_image_captioning_pub_amd.build --force`")
    lib.dic_struct_bytes.restype = C.c_size_t
    _lib = lib
    return lib


def check_struct(lib: C.CDLL, which: int, mirror) -> None:
    """The ctypes mirror of a struct of include/dic.h must have the size the library compiled (dic_struct_bytes)."""
    want = lib.dic_struct_bytes(which)
    if want != C.sizeof(mirror):
        raise DicError(f"ctypes mirror {mirror.__name__} is {C.sizeof(mirror)} bytes, the library's struct {want}: the binding is stale")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dic_last_error().decode("utf-8", "replace")
        raise DicError(f"{what} failed with code {rc}: {msg}")


def declared_symbols() -> list:
    """Function names declared in include/dic.h (used by the CPU test that the .so exports them all)."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dic_[a-z0-9_]+)\s*\(", text)))


def ptr(t) -> C.c_void_p:
    """Device (or host) pointer of a torch tensor; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def stream_ptr() -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
