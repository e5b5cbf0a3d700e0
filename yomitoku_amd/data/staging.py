"""Pinned staging + asynchronous H2D for decoded pages (MI355X ingestion)."""

from __future__ import annotations

import queue
import threading
from typing import Iterable, Iterator, List, Tuple

import numpy as np
import torch


class PageStager:
    """uint8 H x W x 3 BGR pages -> HBM through a ring of pinned host buffers and a dedicated copy stream.

    `upload(img)` returns the device tensor immediately; the copy runs on `self.stream` and the CALLER's current stream
    is made to wait for it (event), so kernels queued afterwards see the page while the host moves on to decode the
    next one.  A ring slot is reused only after its previous copy has completed (its event is synchronised first)."""

    def __init__(self, device, slots: int = 4):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("PageStager stages pages into HBM: it needs a HIP device")
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots: List[dict] = [{"buf": None, "event": None} for _ in range(max(2, int(slots)))]
        self._next = 0
        self._lock = threading.Lock()

    def upload(self, img: np.ndarray, wait: bool = True) -> torch.Tensor:
        """wait=True: the caller's current stream waits for the copy (event) before anything queued after this call.
        wait=False: the caller orders its consumer streams itself (against an event recorded on `self.stream`) and keeps
        the tensor alive until they are done with it (yomitoku_amd.serving does both per wave)."""
        if not isinstance(img, np.ndarray) or img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("page must be a uint8 H x W x 3 BGR array")
        n = int(img.size)
        with self._lock:
            slot = self._slots[self._next]
            self._next = (self._next + 1) % len(self._slots)
            if slot["event"] is not None:
                slot["event"].synchronize()  # the copy that last used this buffer has left it
            if slot["buf"] is None or slot["buf"].numel() < n:
                slot["buf"] = torch.empty(max(n, 1 << 23), dtype=torch.uint8, pin_memory=True)
            host = slot["buf"][:n].view(img.shape)
            np.copyto(host.numpy(), img)  # also makes a strided BGR view (img[:, :, ::-1]) contiguous
            with torch.cuda.stream(self.stream):
                # allocated on the COPY stream's pool: a block the caching allocator hands back here was freed in this
                # stream's order (or is held back until the streams recorded on it are done), so the DMA can never
                # overwrite data that kernels queued on the caller's stream still read
                out = torch.empty(img.shape, dtype=torch.uint8, device=self.device)
                out.copy_(host, non_blocking=True)
                event = torch.cuda.Event()
                event.record(self.stream)
            slot["event"] = event
        if wait:
            current = torch.cuda.current_stream(self.device)
            current.wait_event(event)
            out.record_stream(current)
        return out


_default_stagers = {}
_default_lock = threading.Lock()


def default_stager(device) -> PageStager:
    """The process-wide stager of a device (what imaging.page_to_device, i.e. every module's `__call__`, goes through)."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    with _default_lock:
        stager = _default_stagers.get(index)
        if stager is None:
            stager = _default_stagers[index] = PageStager(torch.device("cuda", index), slots=4)
    return stager


def stream_pages(paths: Iterable[str], device, prefetch: int = 4) -> Iterator[Tuple[str, int, np.ndarray, torch.Tensor]]:
    """(path, page index, host BGR page, device page) for every page of every image file in `paths`, decoded by a
    background thread (Pillow releases the GIL while decoding) and staged through a PageStager `prefetch` pages ahead of
    the consumer.  PDFs go through load_pdf (needs pypdfium2)."""
    from .functions import load_image, load_pdf

    stager = PageStager(device, slots=prefetch + 2)
    q: "queue.Queue" = queue.Queue(maxsize=max(1, int(prefetch)))
    done = object()

    def producer():
        try:
            with torch.cuda.device(stager.device):
                for path in paths:
                    pages = load_pdf(path) if str(path).lower().endswith(".pdf") else load_image(path)
                    for k, page in enumerate(pages):
                        q.put((str(path), k, page, stager.upload(page)))
        except BaseException as exc:  # noqa: BLE001 - re-raised in the consumer
            q.put(exc)
        finally:
            q.put(done)

    thread = threading.Thread(target=producer, name="ymk-ingest", daemon=True)
    thread.start()
    while True:
        item = q.get()
        if item is done:
            break
        if isinstance(item, BaseException):
            raise item
        path, k, page, dev_page = item
        # the upload was ordered against the PRODUCER's stream; order it against the consumer's too
        torch.cuda.current_stream(stager.device).wait_stream(stager.stream)
        yield path, k, page, dev_page
    thread.join()
