"""Host <-> HBM staging for whole volumes: pinned double buffers and one HIP copy stream each way.

At config-2 size one unit is 4.3 GB of camera counts in (8.6 GB as float32) and 3.2 GB out, against
~55 ms of kernels: over PCIe the copies take longer than the arithmetic, so the steady state of a
run over many units is set by whichever of {upload, kernels, download} is slowest -- provided they
overlap.  The reference uploads synchronously from pageable memory inside the step
(``torch.as_tensor(volume, device=...)``, ``shrimpy/preprocessing.py:316``) and downloads with
``.cpu()``; here the three run concurrently:

* ``up`` stream: pinned host slot -> device slot (``hipMemcpyAsync`` through torch), unit k+1;
* compute stream (the caller's current stream): kernels of unit k;
* ``down`` stream: result of unit k-1 -> pinned host slot.

Ordering is carried by HIP events only (no device-wide synchronise): a device slot is not
overwritten before the kernels that read it have finished, a host slot not before its copy has.
PyTorch is the plumbing here (pinned allocations, streams, events); there is no kernel in this file.
"""

from __future__ import annotations

import os

import numpy as np

from . import _lib

__all__ = ["VolumeStager", "EncodedVolume"]


class EncodedVolume:
    """One result as the chunk frames the device wrote (``io/device_codec.DeviceBloscEncoder``): ``frames[k]`` is a
    uint8 numpy view (pinned host memory) of chunk k's blosc frame, chunks being consecutive z-ranges of
    ``frame_bytes`` decoded bytes each.  Valid until the staging slot is reused."""

    def __init__(self, frames, frame_bytes: int, shape, dtype=np.float32):
        self.frames, self.frame_bytes, self.shape, self.dtype = frames, int(frame_bytes), tuple(shape), np.dtype(dtype)

    @property
    def nbytes(self) -> int:
        return int(sum(f.size for f in self.frames))


def _pinned_tensor(shape, dtype):
    """A CPU tensor over ``lsr_pinned_alloc`` memory of exactly the tensor's size.  The allocation belongs
    to the buffer object every view (tensor, numpy array) keeps alive, and is freed with it."""
    import ctypes
    import weakref

    import torch

    nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    ptr = ctypes.c_void_p()
    _lib.call("lsr_pinned_alloc", nbytes, ctypes.byref(ptr))
    buf = (ctypes.c_char * nbytes).from_address(ptr.value)
    weakref.finalize(buf, _lib.load().lsr_pinned_free, ptr.value)
    t = torch.frombuffer(buf, dtype=dtype).view(shape)
    if not t.is_pinned():      # torch does not recognise the range: treat it like any failed pin
        del t, buf
        raise _lib.LsrError("lsr_pinned_alloc", -1, "torch does not see the allocation as page-locked")
    return t


class VolumeStager:
    """``depth`` pinned host slots + device slots for raw stacks, ``depth`` pinned slots for results.

    Threading: ``stage_in`` may be called from a loader thread, ``acquire``/``release``/
    ``stage_out`` from the thread that launches the kernels, ``collect`` from a writer thread.
    Slot ``k`` serves units ``k, k + depth, ...``; calls for one slot must come in unit order.
    """

    def __init__(self, raw_shape, raw_dtype, out_shape, device, depth: int = 2, pin: str = "exact",
                 encode_frame_bytes: int | None = None, encode_blocksize: int = 0, decode_layout: dict | None = None,
                 codec_streams: str | None = None):
        """``depth`` slots each way (default 2; more only helps when load times vary a lot).

        ``encode_frame_bytes``: the result leaves the device as blosc-zstd chunk frames of that many decoded bytes
        each (``csrc/blosc_encode.hip``): ``stage_out`` runs the encoder behind the unit's kernels, ``collect`` returns
        an :class:`EncodedVolume` whose frames the writer stores as they are -- the host never sees the float32
        volume and never runs zstd.

        ``decode_layout`` (``dict(nbytes, blocksize, typesize)``, from ``ZarrArray.compressed_layout``): the raw stack
        arrives as the blosc-zstd chunk frames of the store.  ``host_in`` is then a pinned BYTE buffer the loader fills
        with ``ZarrArray.read_volume_frames`` (file reads only), ``stage_in`` takes the resulting ``CompressedVolume``,
        uploads the compressed bytes, the decoder (``csrc/blosc_decode.hip``) runs in front of the unit's kernels, and
        ``acquire`` checks the decoder's status word before the kernels use the stack.

        ``codec_streams`` (``LSR_CODEC_STREAMS``): ``"serial"`` (default) launches decoder and encoder on the compute
        stream, in line with the unit's kernels; ``"overlap"`` on streams of their own (see the comment at ``_serial``).

        ``pin``: ``"exact"`` takes page-locked allocations of exactly the slot size from the HIP runtime
        (``lsr_pinned_alloc`` = ``hipHostMalloc``) -- torch's caching host allocator rounds every pinned
        block up to a power of two (a 4.3 GB raw slot becomes 8 GB, a 3.2 GB result slot 4 GB: ~24 GB per
        rank instead of ~15 GB at config 2); ``"torch"`` uses ``pin_memory=True``; ``"none"`` keeps the
        slots pageable (copies are then synchronous: correct, slower).

        (Round 2 pinned ordinary torch allocations in place with ``hipHostRegister``.  That is the
        kernel's user-pointer path: the pages stay the process's own, and when their CPU mapping changes
        under a registered range -- huge-page collapse, NUMA balancing, compaction -- the driver evicts and
        restores the process's GPU queues with copies in flight.  The store-to-store tests that use these
        slots were the ones that aborted intermittently; driver-owned pinned memory has no such path.)"""
        import torch

        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.LsrError("VolumeStager", -1, f"device {self.device}: staging needs a HIP device "
                                "(pinned host memory and copy streams); a run without a GPU reads and writes volumes directly.")
        if depth < 2:
            raise ValueError("depth must be at least 2 (one slot in flight, one being filled)")
        self.depth = int(depth)
        self.raw_shape = tuple(int(v) for v in raw_shape)
        self.out_shape = tuple(int(v) for v in out_shape)
        dt = {np.dtype("uint16"): torch.uint16, np.dtype("float32"): torch.float32}.get(np.dtype(raw_dtype))
        if dt is None:
            raise TypeError(f"raw dtype {raw_dtype}: uint16 (camera counts) or float32")
        if pin not in ("exact", "torch", "none"):
            raise ValueError("pin must be 'exact', 'torch' or 'none'")
        self._encoders = None
        self.encode_frame_bytes = None
        out_slot_shape, out_slot_dtype = self.out_shape, torch.float32
        if encode_frame_bytes:
            from .io.device_codec import DeviceBloscEncoder

            nbytes = 4 * int(np.prod(self.out_shape))
            self.encode_frame_bytes = int(encode_frame_bytes)
            self._encoders = [DeviceBloscEncoder(nbytes, 4, self.encode_frame_bytes, self.device, encode_blocksize)
                              for _ in range(depth)]
            out_slot_shape, out_slot_dtype = (self._encoders[0].capacity,), torch.uint8
            self._table_host = [torch.empty((self._encoders[0].n_frames, 2), dtype=torch.int64).pin_memory()
                                for _ in range(depth)]
            self._table_ready = [None] * depth
        self._decoder = None
        in_slot_shape, in_slot_dtype = self.raw_shape, dt
        if decode_layout:
            from .io.device_codec import DeviceBloscDecoder

            raw_bytes = int(np.prod(self.raw_shape)) * np.dtype(raw_dtype).itemsize
            if int(decode_layout["typesize"]) != np.dtype(raw_dtype).itemsize:
                raise ValueError(f"chunk frames of {decode_layout['typesize']}-byte elements for a {np.dtype(raw_dtype)} stack")
            self._decoder = DeviceBloscDecoder(raw_bytes, decode_layout["nbytes"], decode_layout["blocksize"],
                                               decode_layout["typesize"], self.device)
            cap = self._decoder.comp_capacity + 16 * self._decoder.n_frames
            in_slot_shape, in_slot_dtype = (cap,), torch.uint8
            self._comp_dev = [torch.empty(cap, dtype=torch.uint8, device=self.device) for _ in range(depth)]
            self._table_in_dev = [torch.empty((self._decoder.n_frames, 2), dtype=torch.int64, device=self.device) for _ in range(depth)]
            self._table_in_host = [torch.empty((self._decoder.n_frames, 2), dtype=torch.int64).pin_memory() for _ in range(depth)]
            # (two status words per slot, used in turn: in serial mode a unit's word is read when its RESULT is collected,
            # and by then the host may have queued the decode of the unit that takes the slot next)
            self._status_host = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(2 * depth)]
        with torch.cuda.device(self.device):
            self._host_in = [self._host_slot(in_slot_shape, in_slot_dtype, pin) for _ in range(depth)]
            self._host_out = [self._host_slot(out_slot_shape, out_slot_dtype, pin) for _ in range(depth)]
        self._dev_in = [torch.empty(self.raw_shape, dtype=dt, device=self.device) for _ in range(depth)]
        self._up = torch.cuda.Stream(self.device)
        self._down = torch.cuda.Stream(self.device)
        # the encoder's launches get a stream of their own: on the download stream the frames of unit i + 1 (queued as soon
        # as its kernels are launched) would sit in front of the download of unit i (queued only once its frame table is
        # on the host) and hold it back by a whole unit of kernels
        # codec_streams: "serial" (default) runs the decoder and the encoder on the COMPUTE stream, in line with the unit's
        # kernels; "overlap" gives them streams of their own beside the kernels.  Neither codec kernel can share a CU
        # with the fused RL kernel (143 KB of LDS and 247 of 256 VGPRs per lane are taken), so "beside" means that
        # decode / encode workgroups and RL workgroups take CUs away from each other: traced (tools/probes/gpu_timeline.py,
        # profiles/r05_gpu_timeline.txt) an RL launch then takes 3.2 ms instead of 1.43, a decode 21 ms instead of 6.9, and
        # the GPU is busy 66 ms per config-4 unit for 45 ms of work.  The copies stay on their own streams either way.
        if codec_streams is None:
            codec_streams = os.environ.get("LSR_CODEC_STREAMS", "serial")
        if codec_streams not in ("serial", "overlap"):
            raise ValueError("codec_streams must be 'serial' or 'overlap'")
        self._serial = codec_streams == "serial"
        self._enc = torch.cuda.Stream(self.device) if self._encoders is not None and not self._serial else None
        self._pending_decode = [None] * depth
        self._decode_done = [None] * depth      # input side: (event, status word) of the unit acquired in the slot
        self._verdict = [None] * depth          # result side: the same pair, handed over at stage_out, read at collect
        self._status_turn = [0] * depth
        # LSR_STAGE_EVENTS=1: HIP events around the decoder, the unit's kernels and the encoder (serial mode), reported by
        # gpu_times() -- what the card spends on a unit without a profiler attached (rocprofv3 turns the device -> host
        # copies into shader kernels that starve the decoder: a decode then reads 28 ms instead of 7)
        self._ev = [] if os.environ.get("LSR_STAGE_EVENTS") == "1" else None
        self._ev_open = {}
        self._uploaded = [None] * depth      # recorded on `up` after the H2D copy of the slot
        self._consumed = [None] * depth      # recorded on the compute stream when the raw slot is dead
        self._downloaded = [None] * depth    # recorded on `down` after the D2H copy of the slot

    @staticmethod
    def _host_slot(shape, dtype, pin: str):
        import torch

        if pin == "torch":
            return torch.empty(shape, dtype=dtype, pin_memory=True)
        if pin == "none" or not all(shape):
            return torch.empty(shape, dtype=dtype)
        try:
            return _pinned_tensor(shape, dtype)
        except _lib.LsrError:
            return torch.empty(shape, dtype=dtype, pin_memory=True)   # fall back to torch's allocator

    def close(self) -> None:
        """Drain the copy streams and let go of the slots.  A pinned slot goes back to the driver when its
        last view dies -- a numpy view a caller still holds (``host_in`` / ``collect``) keeps it valid."""
        self.drain()
        self._host_in, self._host_out, self._dev_in = [], [], []
        self._encoders = self._decoder = None

    @property
    def takes_frames(self) -> bool:
        """True when ``host_in`` is a byte buffer for ``ZarrArray.read_volume_frames`` (``decode_layout`` was given)."""
        return self._decoder is not None

    # ---- host -> device --------------------------------------------------------------------
    def host_in(self, slot: int) -> np.ndarray:
        """The slot's pinned input buffer as a numpy view, once its previous upload has finished
        (loaders fill it in place: a copy into it costs more than the PCIe transfer)."""
        ev = self._uploaded[slot]
        if ev is not None:
            ev.synchronize()
        return self._host_in[slot].numpy()

    def stage_in(self, slot: int, data=None):
        """Enqueue the upload of the slot (after copying ``data`` into it, if given and not the
        slot's own buffer).  Returns at once; ``acquire`` makes the compute stream wait for it."""
        import torch

        view = self.host_in(slot)
        if self._decoder is not None:
            from .io.device_codec import CompressedVolume

            if not isinstance(data, CompressedVolume):
                raise TypeError("a stager with decode_layout takes the CompressedVolume that read_volume_frames returned "
                                "for this slot's buffer")
            if data.used > view.size or tuple(data.table.shape) != tuple(self._table_in_host[slot].shape):
                raise ValueError(f"compressed volume of {data.used} bytes / {data.table.shape[0]} chunks does not fit the slot")
            self._table_in_host[slot].numpy()[...] = data.table
            with torch.cuda.stream(self._up):
                if self._consumed[slot] is not None:
                    self._up.wait_event(self._consumed[slot])
                used = max(int(data.used), 1)
                self._comp_dev[slot][:used].copy_(self._host_in[slot][:used], non_blocking=True)
                self._table_in_dev[slot].copy_(self._table_in_host[slot], non_blocking=True)
                if not self._serial:
                    self._decoder.decode(self._comp_dev[slot], int(data.used), self._table_in_dev[slot], self._dev_in[slot])
                    self._status_host[slot].copy_(self._decoder.status, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._up)
            self._pending_decode[slot] = int(data.used) if self._serial else None
            self._uploaded[slot] = ev
            return slot
        if data is not None and not (isinstance(data, np.ndarray) and np.shares_memory(data, view)):
            arr = np.asarray(data)
            if arr.shape != view.shape:
                raise ValueError(f"expected raw shape {view.shape}, got {arr.shape}")
            np.copyto(view, arr, casting="same_kind")
        with torch.cuda.stream(self._up):
            if self._consumed[slot] is not None:
                self._up.wait_event(self._consumed[slot])
            self._dev_in[slot].copy_(self._host_in[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._up)
        self._uploaded[slot] = ev
        return slot

    def acquire(self, slot: int):
        """Device tensor of the slot; the current stream waits (on the GPU) for its upload."""
        import torch

        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._uploaded[slot])
        if self._decoder is not None:
            if self._pending_decode[slot] is not None:
                # serial mode: the decoder runs HERE, on the compute stream, between the previous unit's kernels and this
                # unit's -- see __init__ (codec_streams).  Nothing waits for it on the host: the unit's kernels are queued
                # right behind it (waiting here for its verdict left the card idle for 2-4 ms per unit while the host
                # caught up), and the verdict is read when the unit's result is collected -- a stack that did not decode
                # has been processed for nothing, never stored.
                e0 = self._mark(cur)
                self._decoder.decode(self._comp_dev[slot], self._pending_decode[slot], self._table_in_dev[slot], self._dev_in[slot])
                self._status_turn[slot] ^= 1
                word = self._status_host[slot + self.depth * self._status_turn[slot]]
                word.copy_(self._decoder.status, non_blocking=True)
                self._pending_decode[slot] = None
                done = torch.cuda.Event(enable_timing=self._ev is not None)
                done.record(cur)
                self._decode_done[slot] = (done, word)
                if self._ev is not None:
                    self._ev_open[slot] = (e0, done)
            else:
                self._decode_done[slot] = None
                # streams of their own: the decoder ran ahead on the upload stream; its verdict is known by now
                self._uploaded[slot].synchronize()
                self._decoder.check(int(self._status_host[slot].item()))
        return self._dev_in[slot]

    def _check_decoded(self, slot: int) -> None:
        """Serial codec streams: the decoder's status word of the unit in ``slot`` (its launches are long over when the
        unit's result is collected)."""
        verdict = self._verdict[slot] if self._decoder is not None else None
        if verdict is not None:
            done, word = verdict
            self._verdict[slot] = None
            done.synchronize()
            self._decoder.check(int(word.item()))

    def release(self, slot: int) -> None:
        """The kernels enqueued so far on the current stream are the last readers of the slot."""
        import torch

        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._consumed[slot] = ev

    # ---- device -> host --------------------------------------------------------------------
    def stage_out(self, slot: int, result) -> None:
        """Enqueue the download of ``result`` (as of now on the current stream) into the slot."""
        import torch

        if tuple(result.shape) != self.out_shape:
            raise ValueError(f"expected result shape {self.out_shape}, got {tuple(result.shape)}")
        prev = self._downloaded[slot]
        if prev is not None:
            prev.synchronize()                  # the writer may still be reading the host slot
        # the decoder's verdict on this unit's input travels with its result: (the input slot of the unit has the result
        # slot's index -- both are the unit's number modulo depth -- and the next unit to ACQUIRE that slot may do so before
        # this result is collected; the next one to stage a result OUT of it cannot)
        if self._decoder is not None:
            self._verdict[slot] = self._decode_done[slot]
            self._decode_done[slot] = None
        for dev_slot in self._dev_in:
            # a pipeline that hands its input back (e.g. zero RL iterations on a float32 stack):
            # the raw slot was released before this download was queued, so the upload two units on
            # could overwrite it under the copy -- detach the result from the slot first
            if result.untyped_storage().data_ptr() == dev_slot.untyped_storage().data_ptr():
                result = result.clone()
                break
        done = torch.cuda.Event(enable_timing=self._ev is not None)
        done.record(torch.cuda.current_stream(self.device))
        if self._encoders is not None:
            # frames are written on the download stream (beside the next unit's kernels); only the (offset, size)
            # table comes down now -- collect() then copies exactly the compressed bytes
            enc = self._enc if self._enc is not None else torch.cuda.current_stream(self.device)
            with torch.cuda.stream(enc):
                if self._enc is not None:
                    enc.wait_event(done)
                    result.record_stream(enc)
                _, table = self._encoders[slot].encode(result.contiguous())
                self._table_host[slot].copy_(table, non_blocking=True)
                ev = torch.cuda.Event(enable_timing=self._ev is not None)
                ev.record(enc)
                if self._ev is not None and slot in self._ev_open:
                    self._ev.append(self._ev_open.pop(slot) + (done, ev))
            self._table_ready[slot] = ev
            self._downloaded[slot] = None
            return
        with torch.cuda.stream(self._down):
            self._down.wait_event(done)
            self._host_out[slot].copy_(result, non_blocking=True)
            result.record_stream(self._down)    # keep the allocator from recycling it under the copy
            ev = torch.cuda.Event()
            ev.record(self._down)
        self._downloaded[slot] = ev

    def collect(self, slot: int):
        """The slot's result on the host (pinned numpy view), after its download has finished.
        Valid until the slot's next ``stage_out``.  With ``encode_frame_bytes``: an :class:`EncodedVolume`."""
        if self._encoders is not None:
            import torch

            self._table_ready[slot].synchronize()
            self._check_decoded(slot)
            table = self._table_host[slot].numpy()
            end = int(table[-1, 0] + table[-1, 1])
            if end > self._host_out[slot].numel():
                raise _lib.LsrError("VolumeStager.collect", -1, f"frames end at byte {end}, the slot holds {self._host_out[slot].numel()}")
            with torch.cuda.stream(self._down):
                self._host_out[slot][:end].copy_(self._encoders[slot].out[:end], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._down)
            self._downloaded[slot] = ev
            ev.synchronize()
            host = self._host_out[slot].numpy()
            return EncodedVolume([host[int(o):int(o) + int(n)] for o, n in table], self.encode_frame_bytes, self.out_shape)
        self._downloaded[slot].synchronize()
        self._check_decoded(slot)
        return self._host_out[slot].numpy()

    def warm_up(self) -> None:
        """One launch of the decoder (on a volume of absent chunks) and of the encoder (on zeros) before the first unit
        arrives, on the streams the real launches will use (which therefore follow them in stream order): the kernels'
        code objects are loaded and their LDS limits set while the loader reads -- the first unit of a config-4 run took
        144 ms on the caller's clock without this, 45 ms with it."""
        import torch

        cur = torch.cuda.current_stream(self.device)
        if self._decoder is not None:
            with torch.cuda.stream(cur if self._serial else self._up):
                self._table_in_dev[0].zero_()
                self._decoder.decode(self._comp_dev[0], 0, self._table_in_dev[0], self._dev_in[0])
        if self._encoders is not None:
            zeros = torch.zeros(self.out_shape, dtype=torch.float32, device=self.device)
            enc = cur if self._enc is None else self._enc
            with torch.cuda.stream(enc):
                if enc is not cur:
                    enc.wait_stream(cur)
                    zeros.record_stream(enc)
                self._encoders[0].encode(zeros)
        # slot 0's buffers were touched on the compute stream: its first real upload (upload stream, loader thread) and the
        # first real encode come behind this
        done = torch.cuda.Event()
        done.record(cur)
        self._up.wait_event(done)
        if self._enc is not None:
            self._enc.wait_event(done)

    def _mark(self, stream):
        if self._ev is None:
            return None
        import torch

        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def gpu_times(self) -> dict | None:
        """With ``LSR_STAGE_EVENTS=1`` (serial codec streams, decoder and encoder in use): median milliseconds the card
        spent per unit in the decoder, in the unit's kernels and in the encoder, by HIP events on the compute stream."""
        if not self._ev:
            return None
        self.drain()
        rows = np.array([[a.elapsed_time(b), b.elapsed_time(c), c.elapsed_time(d)] for a, b, c, d in self._ev])
        med = np.median(rows, axis=0)
        return {"per_unit_gpu_ms [decode, kernels, encode]": [[round(float(v), 1) for v in r] for r in rows],
                "units": len(rows), "decode_ms": round(float(med[0]), 2), "kernels_ms": round(float(med[1]), 2),
                "encode_ms": round(float(med[2]), 2), "sum_ms": round(float(med.sum()), 2)}

    def drain(self) -> None:
        self._up.synchronize()
        self._down.synchronize()
        if self._enc is not None:
            self._enc.synchronize()
