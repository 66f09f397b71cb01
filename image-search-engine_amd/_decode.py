"""Image decode worker for the process pool of ``Describer``.  Kept free of torch and
of the HIP library on purpose: spawned workers import only this module, PIL and numpy.

Decoded pixels travel back through a ring of fixed-size slots in one memory-mapped file under /dev/shm
(``SlotRing``; the parent owns the file, workers map it once): the result message is then a shape, not
half a megabyte of pickled pixels through a pipe.  An image larger than a slot, or a pool without a ring,
falls back to returning the array itself."""
from __future__ import annotations

import mmap
import os
import tempfile
import weakref

import numpy as np


def _decode_rgb(path):
    from PIL import Image

    with Image.open(str(path)) as im:
        return np.asarray(im.convert("RGB"))


def read_image_bgr(path):
    """BGR uint8 HxWx3 like cv2.imread(IMREAD_COLOR) (backend/descriptors.py:64-68), decoded with PIL.
    Returns (path, array, None) or (path, None, error text): exceptions do not cross the process
    boundary as objects the parent would have to unpickle."""
    try:
        rgb = _decode_rgb(path)
        return path, np.ascontiguousarray(rgb[:, :, ::-1]).astype(np.uint8), None
    except Exception as e:  # the caller prints and skips, as the reference does
        return path, None, f"Problem opening image ({e.__class__.__name__})"


class SlotRing:
    """``nslots`` slots of ``slot_bytes`` in one file under ``root`` (parent side: creates and removes it)."""

    def __init__(self, nslots: int, slot_bytes: int, root: str = "/dev/shm"):
        need = nslots * slot_bytes
        st = os.statvfs(root)
        if st.f_bavail * st.f_frsize < 2 * need:
            raise OSError(f"{root} has less than {2 * need} bytes free")
        fd, self.path = tempfile.mkstemp(prefix="ise_decode_", dir=root)
        try:
            os.ftruncate(fd, need)
            self.map = mmap.mmap(fd, need)
        except Exception:
            os.close(fd)
            os.unlink(self.path)
            raise
        os.close(fd)
        self.nslots, self.slot_bytes = nslots, slot_bytes
        self.buf = np.frombuffer(self.map, dtype=np.uint8)
        # the file must not outlive the process even when close() is never reached (finalizers also run at exit)
        self._unlink = weakref.finalize(self, _unlink_quiet, self.path)

    def view(self, slot: int, shape) -> np.ndarray:
        """The pixels a worker left in ``slot`` (no copy: valid until the slot is handed out again)."""
        n = int(np.prod(shape))
        return self.buf[slot * self.slot_bytes: slot * self.slot_bytes + n].reshape(shape)

    def close(self) -> None:
        path, self.path = getattr(self, "path", None), None
        if path:
            self.buf = None
            try:
                self.map.close()
            except BufferError:  # a view is still alive somewhere: the mapping goes with it
                pass
            self._unlink()

    def __del__(self):
        self.close()


def _unlink_quiet(path: str) -> None:
    try:
        os.unlink(path)
    except OSError:
        pass


_ring = {}  # worker side: path -> (mmap, uint8 view); one entry per pool lifetime


def _worker_ring(path: str):
    ent = _ring.get(path)
    if ent is None:
        _ring.clear()  # a new ring replaces the old one: let its mapping go
        fd = os.open(path, os.O_RDWR)
        try:
            m = mmap.mmap(fd, 0)
        finally:
            os.close(fd)
        ent = _ring[path] = (m, np.frombuffer(m, dtype=np.uint8))
    return ent[1]


def read_image_bgr_into(path, ring_path: str, slot: int, slot_bytes: int):
    """As ``read_image_bgr``, but the pixels are left in slot ``slot`` of the ring file: returns
    (path, (h, w, 3), None).  An image that does not fit a slot comes back as the array itself."""
    try:
        rgb = _decode_rgb(path)
        if rgb.nbytes > slot_bytes:
            return path, np.ascontiguousarray(rgb[:, :, ::-1]).astype(np.uint8), None
        buf = _worker_ring(ring_path)
        dst = buf[slot * slot_bytes: slot * slot_bytes + rgb.nbytes].reshape(rgb.shape)
        np.copyto(dst, rgb[:, :, ::-1])  # the channel swap and the hand-over are one pass
        return path, tuple(int(v) for v in rgb.shape), None
    except Exception as e:
        return path, None, f"Problem opening image ({e.__class__.__name__})"
