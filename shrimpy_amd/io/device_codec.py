"""Chunk codecs that run on the MI355X: blosc-zstd frames written (and read) in HBM.

The acquisition's format is Zarr v3 shards of blosc frames with the zstd compressor and byte shuffle
(``shrimpy/mantis/mantis_engine.py:474-481``; asserted in ``shrimpy/tests/test_mantis_integration.py:177-190``),
and the CLI writes its results the same way.  Round 4 measured what that costs on the host: zstd level 1 over the
1.75 GB float32 result of one config-4 unit is ~6 core-seconds -- 0.38 s per unit on a rank's 16 cores against
30 ms of kernels.  Here the frames are produced where the result already is:

* :class:`DeviceBloscEncoder` -- ``lsr_blosc_encode_device`` (``csrc/blosc_encode.hip``): a volume in HBM ->
  one c-blosc 1.x frame per Zarr chunk, compacted in one device buffer, plus the (offset, size) table.  The
  download then moves compressed bytes and the writer threads only ``pwrite`` them.
* the host twin (``lsr_blosc_encode_device_cpu``: the same bytes from host memory) serves CPU tensors and the
  GPU-less tests, where its frames are decoded by the system libzstd.

The frames are a subset of zstd (Huffman-coded literals, RLE and raw blocks, no sequences) that every zstd
decoder reads -- numcodecs / c-blosc behind iohub included.
"""

from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib

__all__ = ["DeviceBloscEncoder", "DeviceBloscDecoder", "CompressedVolume", "encode_frames_host", "decode_frames_host",
           "plan_frames", "frame_layout", "DecodeError"]


class DecodeError(ValueError):
    """A chunk did not decode (damaged bytes, or a frame that differs from the layout the decoder was planned for)."""

    stage = "load"      # what a pipeline reports for the unit, wherever in it the decoder's status word is read

    _CODES = {1: "corrupt stream", 2: "decodes to more than the chunk holds", 3: "not a zstd stream / unsupported feature",
              4: "size differs from the chunk layout"}

    def __init__(self, status: int, blocks_per_frame: int):
        self.block = (int(status) >> 8) - 1
        self.code = int(status) & 0xFF
        self.frame = self.block // max(int(blocks_per_frame), 1)
        super().__init__(f"chunk {self.frame} (blosc block {self.block}): {self._CODES.get(self.code, f'code {self.code}')}")


class CompressedVolume:
    """One (t, c) volume as it lies in the store: ``table[f] = (offset, size)`` of z-chunk f's blosc frame inside the
    byte buffer the reader filled (``size == 0``: the chunk is absent, i.e. zeros); ``used`` = bytes of the buffer in use."""

    def __init__(self, table: np.ndarray, used: int):
        self.table = np.ascontiguousarray(table, dtype=np.int64).reshape(-1, 2)
        self.used = int(used)


def frame_layout(frame) -> dict | None:
    """(nbytes, blocksize, typesize) of a blosc frame the device decoder takes (zstd, byte shuffle or none), else None."""
    from . import codecs

    if len(frame) < 16:
        return None
    h = codecs.blosc_header(bytes(frame[:16]))
    flags = h["flags"]
    if flags & 0x2:                         # stored frames carry no usable blocksize: any layout decodes them
        return None
    if (flags >> 5) != 4 or ((flags & 0x4) and not ((flags & 0x1) and h["typesize"] > 1)):
        return None
    if h["typesize"] not in (1, 2, 4) or h["blocksize"] <= 0 or h["nbytes"] % h["typesize"] or h["blocksize"] % h["typesize"]:
        return None
    return dict(nbytes=h["nbytes"], blocksize=h["blocksize"], typesize=h["typesize"])


def decode_frames_host(frames, frame_nbytes: int, blocksize: int, typesize: int, out_bytes: int) -> np.ndarray:
    """The host twin of :class:`DeviceBloscDecoder` (``lsr_blosc_decode_device_cpu``: the lane decoder of
    ``csrc/zstd_lane.hpp``, block by block): ``frames`` = the blosc frames of consecutive chunks (``b""`` = absent)."""
    table = np.zeros((len(frames), 2), dtype=np.int64)
    at = 0
    for f, fr in enumerate(frames):
        table[f] = (at, len(fr))
        at += len(fr)
    comp = np.frombuffer(b"".join(bytes(f) for f in frames) or b"\0", dtype=np.uint8)
    out = np.empty(int(out_bytes), dtype=np.uint8)
    status = ctypes.c_uint64(0)
    _lib.call("lsr_blosc_decode_device_cpu", comp.ctypes.data, at, table.ctypes.data, len(frames), int(frame_nbytes),
              int(blocksize), int(typesize), out.ctypes.data, out.size, None, 0, ctypes.byref(status), None)
    if status.value:
        raise DecodeError(status.value, -(-int(frame_nbytes) // int(blocksize)))
    return out



def plan_frames(src_bytes: int, typesize: int, frame_bytes: int, blocksize: int = 0) -> tuple[int, int, int]:
    """``(n_frames, scratch_bytes, out_capacity)`` of one encode call."""
    nf, sb, cap = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _lib.call("lsr_blosc_encode_device_plan", int(src_bytes), int(typesize), int(frame_bytes), int(blocksize),
              ctypes.byref(nf), ctypes.byref(sb), ctypes.byref(cap))
    return int(nf.value), int(sb.value), int(cap.value)


def encode_frames_host(array: np.ndarray, frame_bytes: int, blocksize: int = 0) -> list[bytes]:
    """The host twin: ``array`` (C-contiguous, 1/2/4-byte elements) cut into chunks of ``frame_bytes`` (the last one
    zero-padded), each as one blosc-zstd frame -- byte for byte what :class:`DeviceBloscEncoder` writes."""
    raw = np.ascontiguousarray(array).reshape(-1).view(np.uint8)
    typesize = int(np.asarray(array).dtype.itemsize)
    n_frames, _, cap = plan_frames(raw.size, typesize, frame_bytes, blocksize)
    out = np.empty(cap, dtype=np.uint8)
    frames = np.zeros(2 * n_frames, dtype=np.int64)
    _lib.call("lsr_blosc_encode_device_cpu", raw.ctypes.data, raw.size, typesize, int(frame_bytes), int(blocksize),
              None, 0, out.ctypes.data, out.size, frames.ctypes.data, None)
    return [out[frames[2 * f]:frames[2 * f] + frames[2 * f + 1]].tobytes() for f in range(n_frames)]


class DeviceBloscEncoder:
    """Frames of every chunk of one volume shape, written by the GPU.

    Buffers (block streams, the compacted frames, the frame table) are allocated once for the shape and reused:
    ``encode`` launches three kernels on the current stream and returns device tensors without synchronising.
    """

    def __init__(self, src_bytes: int, typesize: int, frame_bytes: int, device, blocksize: int = 0):
        import torch

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LsrError("DeviceBloscEncoder", -1, f"device {self.device}: the device encoder needs a HIP device "
                                "(CPU arrays go through encode_frames_host)")
        self.src_bytes, self.typesize, self.frame_bytes, self.blocksize = int(src_bytes), int(typesize), int(frame_bytes), int(blocksize)
        self.n_frames, scratch, cap = plan_frames(src_bytes, typesize, frame_bytes, blocksize)
        self.capacity = cap
        self._scratch = torch.empty(scratch, dtype=torch.uint8, device=self.device)
        self.out = torch.empty(cap, dtype=torch.uint8, device=self.device)
        self.frames = torch.zeros((self.n_frames, 2), dtype=torch.int64, device=self.device)

    def encode(self, volume):
        """``volume``: contiguous device tensor of ``src_bytes`` bytes.  Returns ``(out, frames)``: the uint8 buffer
        holding the frames and the ``(n_frames, 2)`` int64 table of (offset, size), both on the device; valid after
        the current stream reaches this point, until the next ``encode``."""
        if volume.device != self.device or not volume.is_contiguous():
            raise ValueError("volume must be a contiguous tensor on the encoder's device")
        if volume.numel() * volume.element_size() != self.src_bytes:
            raise ValueError(f"volume has {volume.numel() * volume.element_size()} bytes, the encoder was planned for {self.src_bytes}")
        if volume.element_size() != self.typesize:
            raise ValueError(f"element size {volume.element_size()} differs from the planned typesize {self.typesize}")
        _lib.call("lsr_blosc_encode_device", volume.data_ptr(), self.src_bytes, self.typesize, self.frame_bytes,
                  self.blocksize, self._scratch.data_ptr(), self._scratch.numel(), self.out.data_ptr(), self.out.numel(),
                  self.frames.data_ptr(), _lib.stream_ptr(self.device))
        return self.out, self.frames

    def encode_to_host(self, volume) -> list[bytes]:
        """Convenience (tests, small volumes): encode, synchronise, return the frames as ``bytes``."""
        out, frames = self.encode(volume)
        table = frames.cpu().numpy()
        end = int(table[-1, 0] + table[-1, 1])
        host = out[:end].cpu().numpy()
        return [host[o:o + n].tobytes() for o, n in table]


class DeviceBloscDecoder:
    """The chunks of one volume layout, decoded by the GPU (``lsr_blosc_decode_device``).

    ``decode(comp, table, out)`` launches on the current stream: ``comp`` = device uint8 tensor with the compressed
    chunks, ``table`` = device int64 ``(n_frames, 2)`` of (offset, size), ``out`` = the volume's device tensor.  The
    status word comes back with ``status_async`` / ``check``; nothing here synchronises."""

    def __init__(self, out_bytes: int, frame_nbytes: int, blocksize: int, typesize: int, device):
        import torch

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LsrError("DeviceBloscDecoder", -1, f"device {self.device}: the device decoder needs a HIP device "
                                "(host arrays go through decode_frames_host or the blosc codecs)")
        self.out_bytes, self.frame_nbytes, self.blocksize, self.typesize = int(out_bytes), int(frame_nbytes), int(blocksize), int(typesize)
        self.n_frames = -(-self.out_bytes // self.frame_nbytes)
        self.blocks_per_frame = -(-self.frame_nbytes // self.blocksize)
        sb = ctypes.c_int64()
        _lib.call("lsr_blosc_decode_device_plan", self.n_frames, self.frame_nbytes, self.blocksize, self.typesize, ctypes.byref(sb))
        self._scratch = torch.empty(int(sb.value), dtype=torch.uint8, device=self.device)
        self.status = torch.zeros(1, dtype=torch.int64, device=self.device)
        # what the compressed chunks of one volume can take at most (c-blosc: nbytes + 16 per frame; this package's
        # device encoder: 12 bytes per block more)
        self.comp_capacity = self.n_frames * (self.frame_nbytes + 16 + 12 * self.blocks_per_frame + 64)

    def decode(self, comp, comp_bytes: int, table, out) -> None:
        if out.numel() * out.element_size() != self.out_bytes or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous tensor of {self.out_bytes} bytes")
        if tuple(table.shape) != (self.n_frames, 2):
            raise ValueError(f"table must be ({self.n_frames}, 2), got {tuple(table.shape)}")
        _lib.call("lsr_blosc_decode_device", comp.data_ptr(), int(comp_bytes), table.data_ptr(), self.n_frames, self.frame_nbytes,
                  self.blocksize, self.typesize, out.data_ptr(), self.out_bytes, self._scratch.data_ptr(), self._scratch.numel(),
                  self.status.data_ptr(), _lib.stream_ptr(self.device))

    def check(self, status_value: int) -> None:
        if status_value:
            raise DecodeError(int(status_value), self.blocks_per_frame)

    def decode_from_host(self, frames, out) -> None:
        """Convenience (tests, small volumes): upload ``frames`` (list of bytes), decode into ``out``, synchronise, check."""
        import torch

        table = np.zeros((len(frames), 2), dtype=np.int64)
        at = 0
        for f, fr in enumerate(frames):
            table[f] = (at, len(fr))
            at += len(fr)
        comp = torch.frombuffer(bytearray(b"".join(bytes(f) for f in frames) or b"\0"), dtype=torch.uint8).to(self.device)
        self.decode(comp, at, torch.as_tensor(table).to(self.device), out)
        self.check(int(self.status.cpu().item()))
