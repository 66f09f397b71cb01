"""Feature extraction with the reference's surface (backend/descriptors.py:24-204):
``SupportsDescribe``, ``Describer``, ``describe_dataset``, ``CNNDescriptor``.

What changes under the surface: the CNN runs BATCHED on PyTorch-ROCm (the
reference does one image per forward pass and one host<->device round trip per
image, backend/descriptors.py:185-196) and preprocessing (resize 224, ImageNet
normalise, HWC->CHW) runs on the device.  What is kept deliberately:
  * images are BGR uint8 and go into the RGB-trained normalisation unswapped
    (quirk 5.9-1: backend/descriptors.py:65,153-159,185);
  * ``describe(image)`` returns a flat CPU float32 tensor of 2048 values
    (backend/descriptors.py:196);
  * per-image failures are printed and skipped (backend/descriptors.py:94-96).
cv2 / albumentations / torchvision are not in this image: decoding uses PIL, the
resize is ``F.interpolate(bilinear, align_corners=False)`` on float pixels, which
differs from cv2's 11-bit fixed-point uint8 resize by <= 1 LSB before normalisation
(semantic, not bit, parity for preprocessing).
"""
from __future__ import annotations

import collections
import threading
from collections import defaultdict
from typing import Protocol

import numpy as np
import torch
import torch.nn.functional as F

from .config import Config, DnnModels
from .resnet import fold_batchnorm_, load_resnet50_weights, resnet50_features
from .utils import chunkIt

config = Config()

_MEAN = (0.485, 0.456, 0.406)  # albumentations A.Normalize() defaults (ImageNet)
_STD = (0.229, 0.224, 0.225)


class SupportsDescribe(Protocol):
    def describe(self, image: np.ndarray) -> np.ndarray: ...


class _PendingFeatures:
    """Features of a batch whose device work may still be running."""

    def __init__(self, host: torch.Tensor, event, release=None):
        self._host, self._event, self._release = host, event, release

    def result(self) -> torch.Tensor:
        if self._event is not None:
            self._event.synchronize()
            self._event = None
            # out of the page-locked buffer (it goes back to the descriptor's pool right here); a plain memcpy:
            # a torch CPU op of this size would wake the intra-op thread pool, whose workers then spin on the
            # decoders' cores
            self._host = torch.from_numpy(self._host.numpy().copy())
            release, self._release = self._release, None
            if release is not None:
                release()
        return self._host


class _DescribeRequest:
    __slots__ = ("image", "result", "error", "alone", "done", "event")

    def __init__(self, image):
        self.image, self.result, self.error, self.alone, self.done = image, None, None, False, False
        self.event = threading.Event()


class CNNDescriptor:
    """backend/descriptors.py:142-204 with a batched device path.

    ``out_dim``: None keeps the reference's 2048-d ``flatten`` node.  BASELINE
    config 2 asks for 512-d embeddings, which the reference's node does not have;
    ``out_dim=512`` applies a fixed seeded 2048->512 Gaussian projection
    (scaled 1/sqrt(2048)) after the network, on the device.
    """

    def __init__(self, model=DnnModels.RESNET, seed: int = 0, out_dim: int | None = None,
                 device: str | None = None, dtype: torch.dtype = torch.float32, fold_bn: bool = True,
                 weights_path=None):
        """``weights_path``: a torchvision ResNet-50 ``state_dict`` file (e.g. IMAGENET1K_V2, what the
        reference downloads at backend/descriptors.py:161-163) loaded with ``weights_only=True``
        before BatchNorm is folded; ``fc.*`` entries are ignored.  None = seeded random init
        (BASELINE config 2; there is no network here)."""
        self.model = model
        self.weights_path = weights_path
        self.seed = seed
        self.out_dim = out_dim
        self.device = torch.device(device or config.DEVICE)
        self.dtype = dtype
        self.fold_bn = fold_bn  # inference only: BatchNorm folded into the convolutions
        self.preprocessor = None
        self.feature_extractor = None
        self.projection = None
        self._stage = {"bufs": [None, None], "events": [None, None], "turn": 0}
        self._stage_lock = threading.Lock()  # Flask request threads share one descriptor
        self._init_combining()
        self._graphs, self._graph_lock = {}, threading.Lock()
        self.initialize_model()

    def _init_combining(self):
        self._cq, self._cq_leader, self._cq_lock = collections.deque(), False, threading.Lock()
        self.combined_batches = self.combined_calls = 0

    def __getstate__(self):  # copies and pickles leave the staging buffers and their lock behind
        d = dict(self.__dict__)
        for name in ("_stage", "_stage_lock", "_cq", "_cq_leader", "_cq_lock", "_graphs", "_graph_lock"):
            d.pop(name, None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._stage = {"bufs": [None, None], "events": [None, None], "turn": 0}
        self._stage_lock = threading.Lock()
        self._init_combining()
        self._graphs, self._graph_lock = {}, threading.Lock()
        if getattr(self.preprocessor, "__self__", None) is not None:
            self.preprocessor = self._preprocess_batch

    def initialize_model(self):
        if self.model == DnnModels.RESNET:
            self.preprocessor = self._preprocess_batch
            net = resnet50_features(self.seed)
            if self.weights_path is not None:
                load_resnet50_weights(net, self.weights_path)
            if self.fold_bn:
                net = fold_batchnorm_(net)
            self.feature_extractor = net.to(self.device).to(memory_format=torch.channels_last)
            if self.out_dim is not None and self.out_dim != 2048:
                g = torch.Generator().manual_seed(self.seed + 1)
                self.projection = (torch.randn(2048, self.out_dim, generator=g) / 2048 ** 0.5).to(self.device)
        elif self.model == DnnModels.BiT:
            raise NotImplementedError("google/bit-50 needs a network fetch (backend/descriptors.py:171-172)")
        else:
            raise ValueError(f"Model '{self.model}' not recognized for feature extraction")
        self.feature_extractor.eval()
        self._mean = torch.tensor(_MEAN, device=self.device).view(1, 3, 1, 1) * 255.0
        self._std = torch.tensor(_STD, device=self.device).view(1, 3, 1, 1) * 255.0

    # -- preprocessing: A.Resize(224,224, INTER_LINEAR) -> A.Normalize() -> ToTensorV2
    def _staging(self, nbytes: int):
        """(buffer, slot): one of two page-locked host buffers used alternately for the batch upload; a
        buffer is handed out again only after the copy that last read it has completed (its event).
        Called with ``_stage_lock`` held."""
        st = self._stage
        i = st["turn"]
        st["turn"] = 1 - i
        if st["events"][i] is not None:
            st["events"][i].synchronize()
        buf = st["bufs"][i]
        if buf is None or buf.numel() < nbytes:
            cap = max(nbytes, 2 * (buf.numel() if buf is not None else 0), 1 << 20)
            buf = st["bufs"][i] = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        return buf, i

    def _preprocess_batch(self, images) -> torch.Tensor:
        """The whole batch goes up in ONE host-to-device copy (pixels packed back to back in a pinned
        buffer); images of the same height and width are then resized together.  Per image this is
        the arithmetic of the one-image path: uint8 -> float, bilinear resize on the float pixels,
        (v - 255 mean) / (255 std), channel order untouched; rows of the result follow the input order."""
        size = config.RESIZE_SIZE
        arrs = []
        for im in images:
            a = np.ascontiguousarray(im)
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("expected an HWC 3-channel uint8 image")
            if a.dtype != np.uint8:  # what torch.from_numpy(...).float() of the old path accepted
                a = a.astype(np.float32)
            arrs.append(a)
        if not arrs:
            return torch.empty((0, 3, size, size), device=self.device)
        if self.device.type != "cuda" or any(a.dtype != np.uint8 for a in arrs):
            out = []
            for a in arrs:
                t = torch.from_numpy(a).to(self.device).permute(2, 0, 1).unsqueeze(0).float()
                if t.shape[2] != size or t.shape[3] != size:
                    t = F.interpolate(t, size=(size, size), mode="bilinear", align_corners=False)
                out.append(t)
            x = torch.cat(out, 0)
        else:
            offs = np.cumsum([0] + [a.size for a in arrs])
            with self._stage_lock:
                host, slot = self._staging(int(offs[-1]))
                hv = host.numpy()
                for a, o in zip(arrs, offs[:-1]):
                    hv[o:o + a.size] = a.reshape(-1)
                dev = host[: int(offs[-1])].to(self.device, non_blocking=True)
                ev = torch.cuda.Event(blocking=True)
                ev.record(torch.cuda.current_stream(self.device))
                self._stage["events"][slot] = ev
            groups: dict[tuple, list] = {}
            for i, a in enumerate(arrs):
                groups.setdefault(a.shape[:2], []).append(i)
            x = torch.empty((len(arrs), 3, size, size), dtype=torch.float32, device=self.device)
            for (h, w), idx in groups.items():
                n = h * w * 3
                if len(idx) > 1 and all(offs[j] - offs[i] == n for i, j in zip(idx, idx[1:])):
                    t = dev[offs[idx[0]]: offs[idx[0]] + n * len(idx)].view(len(idx), h, w, 3)  # neighbours: a view
                else:
                    t = torch.stack([dev[offs[i]: offs[i] + n].view(h, w, 3) for i in idx])
                t = t.permute(0, 3, 1, 2).float()
                if h != size or w != size:
                    t = F.interpolate(t, size=(size, size), mode="bilinear", align_corners=False)
                if len(groups) == 1:
                    x = t
                else:
                    x[torch.as_tensor(idx, device=self.device)] = t
        x = (x - self._mean) / self._std  # (img - mean*255) / (std*255), channel order untouched
        return x.contiguous(memory_format=torch.channels_last)

    def _forward_eager(self, x: torch.Tensor) -> torch.Tensor:
        if self.dtype != torch.float32:
            with torch.autocast(self.device.type, dtype=self.dtype):
                f = self.feature_extractor(x).float()
        else:
            f = self.feature_extractor(x)
        if self.projection is not None:
            f = f @ self.projection
        return f

    GRAPH_BATCHES = (1, 2, 4, 8, 16, 32)

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        """Small batches replay a captured HIP graph of the network (one per batch size in GRAPH_BATCHES,
        captured at first use): at one image the eager forward is 2.4 ms of launches for 0.3 ms of device
        work -- the request path of backend/engine.py:78 lives there.  Larger batches are device-bound
        and run eagerly.  ``config.CNN_GRAPHS = False`` turns the graphs off."""
        b = x.shape[0]
        if x.is_cuda and b in self.GRAPH_BATCHES and getattr(config, "CNN_GRAPHS", True) and not torch.is_grad_enabled():
            with self._graph_lock:   # a graph's input and output buffers are its own: one replay at a time
                g = self._graphs.get(b)
                if g is None and b not in self._graphs:
                    g = self._graphs[b] = self._capture(b)
                if g is not None:
                    graph, x_static, out_static = g
                    x_static.copy_(x)
                    graph.replay()
                    return out_static.clone()
        return self._forward_eager(x)

    def _capture(self, b: int):
        """(graph, input buffer, output buffer) of one batch size, or None when capture is refused."""
        try:
            x_static = torch.zeros((b, 3, config.RESIZE_SIZE, config.RESIZE_SIZE), device=self.device).contiguous(
                memory_format=torch.channels_last)
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):      # the convolution library picks its kernels here, not in the capture
                for _ in range(3):
                    self._forward_eager(x_static)
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out_static = self._forward_eager(x_static)
            return graph, x_static, out_static
        except Exception as e:  # noqa: BLE001 -- eager is always there
            print(f"NOTE: no HIP graph for batches of {b} ({e.__class__.__name__}: {e}); running them eagerly")
            return None

    # -- batched entry points (new capability)
    @torch.no_grad()
    def extract_features_tensor(self, images_u8: torch.Tensor) -> torch.Tensor:
        """Device-resident batch: uint8 (B, H, W, 3) BGR tensor on the extractor's device ->
        (B, d) float32 on the device.  Same arithmetic as ``extract_features_batch`` without
        the per-image host work (used for BASELINE config 2's synthetic images)."""
        size = config.RESIZE_SIZE
        x = images_u8.permute(0, 3, 1, 2).float()
        if x.shape[2] != size or x.shape[3] != size:
            x = F.interpolate(x, size=(size, size), mode="bilinear", align_corners=False)
        x = ((x - self._mean) / self._std).contiguous(memory_format=torch.channels_last)
        return self._forward(x)

    @torch.no_grad()
    def extract_features_batch(self, images) -> torch.Tensor:
        """list of HWC BGR uint8 arrays -> (B, d) float32 tensor ON THE DEVICE."""
        return self._forward(self.preprocessor(images))

    def describe_batch(self, images) -> torch.Tensor:
        """(B, d) float32 CPU tensor."""
        return self.extract_features_batch(images).cpu()

    def describe_batch_async(self, images) -> "_PendingFeatures":
        """Launch ``describe_batch`` without waiting for the device: upload, network and the copy back
        into page-locked memory are enqueued; ``.result()`` waits for them and returns the (B, d) float32
        CPU tensor.  The images may be reused once this returns (their pixels are in the staging buffer)."""
        f = self.extract_features_batch(images)
        if not f.is_cuda:
            return _PendingFeatures(f, None)
        # page-locked result buffers from a free list (pinning memory is slow: they are kept): a buffer belongs
        # to the batch it was handed to until ``.result()`` has copied the features out, however many batches
        # other threads launch in between (describe_dataset with N_JOBS > 1 shares one descriptor).  Events
        # are blocking ones: a waiting host thread sleeps, it does not spin
        with self._stage_lock:
            free = self._stage.setdefault("out_free", [])
            buf = None
            for j, b in enumerate(free):
                if b.numel() >= f.numel():
                    buf = free.pop(j)
                    break
            if buf is None:
                buf = torch.empty(max(f.numel(), 1 << 18), dtype=torch.float32, pin_memory=True)
        host = buf[: f.numel()].view(f.shape)
        host.copy_(f, non_blocking=True)
        ev = torch.cuda.Event(blocking=True)
        ev.record(torch.cuda.current_stream(self.device))

        def release(buf=buf):
            with self._stage_lock:
                pool = self._stage.setdefault("out_free", [])
                if len(pool) < 8:
                    pool.append(buf)

        return _PendingFeatures(host, ev, release)

    # -- the reference's entry points
    def extract_features(self, image):
        if self.model != DnnModels.RESNET:
            raise Exception("Model not recognized")
        return self._forward(self.preprocessor([image])).cpu().flatten()

    def describe(self, image: np.ndarray):
        """One image -> flat (d,) CPU tensor (backend/descriptors.py:198-204).  Calls that arrive from
        several threads while the device is busy -- Flask request threads, backend/engine.py:78,137 -- are
        run together as one batch (``config.DESCRIBE_COMBINE_MAX`` images, 0 = never): a ResNet-50 forward
        costs about the same for 1 image as for 16.  A lone caller is served at once."""
        cmax = int(getattr(config, "DESCRIBE_COMBINE_MAX", 32))
        if cmax <= 1:
            with torch.no_grad():
                return self.extract_features(image)
        # One leader at a time; while the flag is up new calls queue behind it and sleep on their OWN
        # event (a shared condition variable would wake every waiter at every batch, and in Python each
        # of them wants the interpreter lock back).  A leader runs one batch -- its own request is at the
        # head of it -- wakes the callers it served and promotes the next head of the queue.
        req = _DescribeRequest(image)
        with self._cq_lock:
            self._cq.append(req)
            lead = not self._cq_leader
            if lead:
                self._cq_leader = True
        if not lead:
            req.event.wait()          # served, or promoted to leader
        if not req.done:
            with self._cq_lock:
                batch = [self._cq.popleft() for _ in range(min(len(self._cq), cmax))]
            feats, err = None, None
            try:
                if len(batch) == 1:
                    with torch.no_grad():
                        feats = [self.extract_features(batch[0].image)]
                else:
                    # batch sizes are rounded up to a power of two (the last image repeated): MIOpen
                    # tunes every new shape once, for up to a second -- five shapes instead of thirty-one
                    images = [b.image for b in batch]
                    images += [images[-1]] * ((1 << (len(images) - 1).bit_length()) - len(images))
                    feats = list(self.describe_batch(images))[: len(batch)]
                    self.combined_batches += 1
                    self.combined_calls += len(batch)
            except Exception as e:  # noqa: BLE001 -- handed to the caller(s) below
                err = e
            for i, b in enumerate(batch):
                if err is None:
                    b.result = feats[i].flatten()
                elif len(batch) == 1:
                    b.error = err
                else:
                    b.alone = True   # one bad image must not fail its neighbours: everyone retries alone
                b.done = True
                if b is not req:
                    b.event.set()
            with self._cq_lock:
                if self._cq:
                    self._cq[0].event.set()   # the next head leads (the flag stays up for it)
                else:
                    self._cq_leader = False
        if req.alone:
            with torch.no_grad():
                return self.extract_features(image)
        if req.error is not None:
            raise req.error
        return req.result

    extract = describe  # name used by BASELINE.json's north_star

    def warm_up(self, shape=(375, 500)) -> None:
        """Run every batch size concurrent ``describe`` calls can produce (1 and the powers of two up to
        ``config.DESCRIBE_COMBINE_MAX``) once: the convolution library tunes a new shape for up to a second,
        which a server wants at start-up, not inside a request."""
        cmax = max(1, int(getattr(config, "DESCRIBE_COMBINE_MAX", 32)))
        img = np.zeros(tuple(shape) + (3,), np.uint8)
        b = 1
        while b <= cmax:
            self.describe_batch([img] * b)
            b *= 2


class _NullContext:
    """``with`` wrapper for an executor that outlives the block."""

    def __init__(self, obj):
        self.obj = obj

    def __enter__(self):
        return self.obj

    def __exit__(self, *a):
        return False


class Descriptions(defaultdict):
    """What ``Describer.describe`` returns: the reference's ``{name: [per-image arrays]}`` dict
    (backend/descriptors.py:77,99-101) plus, beside it, the path each array came from."""

    def __init__(self):
        super().__init__(list)
        self.paths = defaultdict(list)


class Describer:
    """backend/descriptors.py:47-101.  Descriptors exposing ``describe_batch`` are fed
    ``batch_size`` images per call; results and skip-on-error behaviour are unchanged."""

    def __init__(self, descriptors: dict[str, SupportsDescribe], batch_size: int | None = None):
        self.descriptors = self._validate_descriptors(descriptors)
        self.batch_size = batch_size or config.DNN_BATCH_SIZE
        # paths whose description made it into the last describe_dataset() result, in result order:
        # row i of the index built from that result is described_paths[i] (the reference loses this
        # when an image is skipped)
        self.described_paths: list = []
        self._decode_lock = threading.Lock()  # the decode process pool and its slot ring: one describe() at a time

    def _validate_descriptors(self, descriptors):
        if not descriptors:
            raise Exception("No descriptors provided")
        return descriptors

    def read_image(self, path):
        """BGR uint8 HxWx3 like cv2.imread(IMREAD_COLOR) (decoded with PIL here)."""
        from PIL import Image

        try:
            with Image.open(str(path)) as im:
                rgb = np.asarray(im.convert("RGB"))
        except Exception:
            raise Exception("Problem opening image")
        return np.ascontiguousarray(rgb[:, :, ::-1]).astype(np.uint8)

    def _flush(self, run: "_DescribeRun", final: bool = False):
        """Describe the run's pending batch.  A descriptor with ``describe_batch_async`` only has its batch
        LAUNCHED here; the results are collected when the next batch has been launched (or at the end),
        so the device works on batch i while the host decodes, packs and uploads batch i + 1.  Lists stay
        in input order: a batch is always collected before the next one is.  Everything a call of
        ``describe`` carries from one batch to the next lives in ``run`` -- ``describe_dataset`` calls
        ``describe`` on ONE Describer from N_JOBS threads (backend/descriptors.py:125-129)."""
        if run.pending:
            paths, images, slots = zip(*run.pending)
            launched = {}
            for d_name, descriptor in self.descriptors.items():
                if hasattr(descriptor, "describe_batch_async") and getattr(config, "DESCRIBE_ASYNC", True):
                    try:
                        launched[d_name] = descriptor.describe_batch_async(list(images))
                    except Exception as e:  # collected below as a failed batch
                        launched[d_name] = e
            self._collect(run)
            run.uncollected = (paths, images, slots, launched)
            run.pending = []
        if final:
            self._collect(run)

    def _collect(self, run: "_DescribeRun"):
        prev, run.uncollected = run.uncollected, None
        if prev is None:
            return
        paths, images, slots, launched = prev
        descriptions = run.descriptions
        for d_name, descriptor in self.descriptors.items():
            if d_name in launched or hasattr(descriptor, "describe_batch"):
                try:
                    h = launched.get(d_name)
                    if isinstance(h, Exception):
                        raise h
                    feats = h.result() if h is not None else descriptor.describe_batch(list(images))
                    for f in feats:
                        descriptions[d_name].append(f.reshape(1, -1))
                    descriptions.paths[d_name].extend(paths)
                    continue
                except Exception as e:  # fall back to per-image so one bad image skips alone
                    print(f"ERROR: batched describe failed ({e}); retrying per image")
            for img_path, image in zip(paths, images):
                try:
                    description = descriptor.describe(image)
                    if description is None:
                        raise Exception(f"Couldn't describe image '{img_path}'.")
                    if description.ndim == 1:
                        description = description.reshape(1, -1)
                    descriptions[d_name].append(description)
                    descriptions.paths[d_name].append(img_path)
                except Exception as e:
                    print(f"ERROR: Problem describing image '{img_path}'\n '{e}'")
        del images, prev  # the ring views are dead: only now may their slots be handed out again
        run.release(slots)

    def _process_pool(self, procs: int):
        from concurrent.futures import ProcessPoolExecutor
        import multiprocessing as mp

        pool = getattr(self, "_pool", None)
        if pool is None or getattr(self, "_pool_size", 0) != procs:
            if pool is not None:
                pool.shutdown(wait=False, cancel_futures=True)
            # spawn, not fork: the parent has the GPU open
            self._pool = ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn"))
            self._pool_size = procs
        return self._pool

    def _slot_ring(self, nslots: int):
        """The pixel hand-over ring of the decode processes (``_decode.SlotRing``), or None when
        ``config.DECODE_SLOT_BYTES`` is 0 or /dev/shm cannot hold it (results are then pickled)."""
        slot_bytes = int(getattr(config, "DECODE_SLOT_BYTES", 3 << 20))
        ring = getattr(self, "_ring", None)
        if ring is not None and (ring.nslots < nslots or ring.slot_bytes != slot_bytes):
            ring.close()
            ring = self._ring = None
        if ring is None and slot_bytes > 0 and not getattr(self, "_ring_failed", False):
            from ._decode import SlotRing

            try:
                ring = self._ring = SlotRing(nslots, slot_bytes)
            except OSError as e:
                print(f"NOTE: decode processes return pixels through their pipes ({e})")
                self._ring_failed = True
        return ring

    def close(self):
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=False, cancel_futures=True)
            self._pool = None
        ring = getattr(self, "_ring", None)
        if ring is not None:
            ring.close()
            self._ring = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _safe_read(self, img_path):
        try:
            return img_path, self.read_image(img_path), None
        except Exception as e:
            return img_path, None, e

    def describe(self, images_paths, multiprocess=False) -> dict[str, list]:
        """Decode runs ``decode_workers`` images ahead on a thread pool (PIL releases the GIL
        while decoding) so the GPU batches are not starved by JPEG decode;
        order and skip-on-error behaviour are those of the reference's sequential loop.
        Re-entrant: concurrent calls on one Describer (``describe_dataset`` with N_JOBS > 1) share
        nothing but the decode process pool and its slot ring, which one call at a time uses."""
        procs = max(0, int(getattr(config, "DECODE_PROCESSES", 0)))
        if procs:  # the pool already decodes in parallel: calls that want it take turns
            with self._decode_lock:
                return self._describe(images_paths, procs)
        return self._describe(images_paths, 0)

    def _describe(self, images_paths, procs: int):
        from concurrent.futures import ThreadPoolExecutor

        paths = np.asarray(images_paths).ravel().tolist()
        workers = procs or max(1, int(getattr(config, "DECODE_WORKERS", 8)))
        window = max(workers * 4, self.batch_size)  # decoded images held ahead of the GPU: bounded
        # DECODE_PROCESSES > 0: a pool of spawned processes (the PIL thread pool tops out near 2 k images/s on
        # the GIL-bound array conversion, scripts/feed_rate.py); the pool lives as long as the Describer
        read = self._safe_read
        ring = None
        if procs:
            from ._decode import read_image_bgr as read
            from ._decode import read_image_bgr_into

            # slots come from a free list and go back to it when their image is skipped or its batch has been
            # collected: images holding one number at most window (in flight) + batch_size (pending, at the
            # moment of a launch) + batch_size (launched, not yet collected)
            ring = self._slot_ring(window + 2 * self.batch_size)
        run = _DescribeRun(ring.nslots if ring is not None else 0)
        with (_NullContext(self._process_pool(procs)) if procs else ThreadPoolExecutor(max_workers=workers)) as pool:
            inflight = collections.deque()
            it = iter(paths)
            while True:
                while len(inflight) < window:
                    nxt = next(it, None)
                    if nxt is None:
                        break
                    if ring is not None:
                        slot = run.take_slot()
                        inflight.append((pool.submit(read_image_bgr_into, nxt, ring.path, slot, ring.slot_bytes), slot))
                    else:
                        inflight.append((pool.submit(read, nxt), -1))
                if not inflight:
                    break
                fut, slot = inflight.popleft()
                img_path, image, err = fut.result()  # input order is kept
                if err is not None:
                    print(f"ERROR: Problem describing image '{img_path}'\n '{err}'")
                    run.release((slot,))
                    continue
                if isinstance(image, tuple):  # a shape: the pixels are in the ring (a view, dead once collected)
                    image = ring.view(slot, image)
                else:                         # the pixels came back themselves (larger than a slot): slot unused
                    run.release((slot,))
                    slot = -1
                run.pending.append((img_path, image, slot))
                if len(run.pending) >= self.batch_size:
                    self._flush(run)
        self._flush(run, final=True)
        return run.descriptions


class _DescribeRun:
    """State of ONE ``Describer.describe`` call: the batch being filled, the batch launched but not yet
    collected, the results so far, and the free slots of the decode ring."""

    def __init__(self, nslots: int):
        self.descriptions = Descriptions()
        self.pending: list = []      # (path, image, slot)
        self.uncollected = None      # (paths, images, slots, launched handles)
        self.free = collections.deque(range(nslots))

    def take_slot(self) -> int:
        return self.free.popleft()   # never empty: the ring is sized for everything that can hold a slot

    def release(self, slots) -> None:
        self.free.extend(s for s in slots if s >= 0)


def describe_dataset(describer: Describer, images_paths: np.ndarray, prediction=False) -> list:
    """backend/descriptors.py:104-139: chunk (n_jobs*2), thread-parallel describe, flatten
    into a list of per-image (1, d) arrays.  Quirk 5.9-5 is kept: an existing
    BOVW_CORNER_DESCRIPTIONS_PATH file short-circuits extraction even on the DNN path."""
    import joblib
    from joblib import Parallel, delayed

    n_jobs = 1 if prediction else config.N_JOBS
    if config.BOVW_CORNER_DESCRIPTIONS_PATH.exists() and not prediction:
        print("Loading corner description features from local file.")
        return joblib.load(str(config.BOVW_CORNER_DESCRIPTIONS_PATH))
    paths_chunks = chunkIt(images_paths, n_jobs * 2)
    print("Extracting features from {} images. Splitting in {} jobs.".format(images_paths.shape[0], len(paths_chunks)))
    with Parallel(backend="threading", n_jobs=n_jobs) as parallel:
        dicts = parallel(delayed(describer.describe)(paths, n_jobs > 1) for paths in paths_chunks)
    descriptions = []
    described = []
    for descriptions_dict in dicts:
        for key in descriptions_dict.keys():
            for image_description in descriptions_dict[key]:
                descriptions.append(image_description)
            described.extend(getattr(descriptions_dict, "paths", {}).get(key, []))
    describer.described_paths = described
    return descriptions
