"""numpy restatement of the packed candidate key of include/ise_knn.h (tests only):
key = ord(score) << 32 | global row id, score = squared L2 or -inner product,
ord() = order-preserving map float32 -> uint32; unfilled = 0xFFFFFFFFFFFFFFFF."""
import numpy as np

from oracle import knn_oracle as ko

PAD = np.uint64(0xFFFFFFFFFFFFFFFF)


def ord_f32(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return np.where(u >> 31 != 0, ~u, u ^ np.uint32(0x80000000)).astype(np.uint32)


def unord_f32(o):
    o = np.asarray(o, dtype=np.uint32)
    u = np.where(o & np.uint32(0x80000000) != 0, o ^ np.uint32(0x80000000), ~o).astype(np.uint32)
    return u.view(np.float32)


def encode(D, I, metric):
    """(D float32, I int64 global ids, -1 = unfilled) -> uint64 keys, same shape."""
    sc = D if metric == ko.METRIC_L2 else -D
    keys = (ord_f32(sc).astype(np.uint64) << np.uint64(32)) | (I.astype(np.uint64) & np.uint64(0xFFFFFFFF))
    return np.where(I < 0, PAD, keys)


def decode(keys, metric):
    keys = np.asarray(keys, dtype=np.uint64)
    pad = keys == PAD
    sc = unord_f32((keys >> np.uint64(32)).astype(np.uint32))
    D = np.where(metric == ko.METRIC_L2, sc, -sc).astype(np.float32)
    fl = np.float32(np.finfo(np.float32).max)
    D = np.where(pad, fl if metric == ko.METRIC_L2 else -fl, D).astype(np.float32)
    I = np.where(pad, -1, (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)).astype(np.int64)
    return D, I
