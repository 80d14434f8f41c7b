"""The store format on either side of the hot path: Zarr v3 shards with a
CRC-32C index around blosc-zstd chunks -- what the acquisition engine writes
(``shrimpy/mantis/mantis_engine.py:474-481``, asserted in
``shrimpy/tests/test_mantis_integration.py:153-196``) -- plus bare zstd / gzip / blosc chunks.

The blosc frame codec is checked three ways: against frames a real c-blosc 1.21.0 produced
(committed under ``tests/golden/blosc_frames.npz`` with the script that made them,
``oracle/make_blosc_golden.py``), live against ``libblosc`` where one is loadable, and by round trip.
"""

import json
import os
from pathlib import Path

import numpy as np
import pytest

from shrimpy_amd.io import codecs
from shrimpy_amd.io.omezarr import UnsupportedCodec, ZarrArray, open_ome_zarr


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test patterns + the classic check value
    assert codecs.crc32c(b"123456789") == 0xE3069283
    assert codecs.crc32c(bytes(32)) == 0x8A9136AA
    assert codecs.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert codecs.crc32c(bytes(range(32))) == 0x46DD794E
    assert codecs.crc32c(b"") == 0


def test_crc32c_native_python_and_portable_agree():
    """chunk-level CRC-32C went through a pure-Python byte loop (10 MB/s under the GIL).  The native entry
    (SSE4.2 or tables) equals the Python loop on every length and alignment, chains over pieces, and is fast."""
    import ctypes
    import time

    from shrimpy_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(5)
    blob = rng.integers(0, 256, 4099, dtype=np.uint8)
    for start in (0, 1, 3, 7):
        for n in (0, 1, 7, 8, 9, 63, 64, 1000, 4092):
            piece = blob[start:start + n].tobytes()
            want = codecs._crc32c_python(piece)
            assert codecs.crc32c(piece) == want == codecs.crc32c(memoryview(piece))
            out = ctypes.c_uint32()
            buf = np.frombuffer(piece, dtype=np.uint8)
            assert lib.lsr_crc32c_host_portable(buf.ctypes.data if n else None, n, 0, ctypes.byref(out)) == 0
            assert out.value == want
    # in pieces: the seed carries the running value
    a, b = blob[:1234], blob[1234:]
    out = ctypes.c_uint32()
    assert lib.lsr_crc32c_host(a.ctypes.data, a.size, 0, ctypes.byref(out)) == 0
    assert lib.lsr_crc32c_host(b.ctypes.data, b.size, out.value, ctypes.byref(out)) == 0
    assert out.value == codecs._crc32c_python(blob.tobytes())
    big = rng.integers(0, 256, 64 << 20, dtype=np.uint8)
    t0 = time.perf_counter()
    codecs.crc32c(big)
    assert (time.perf_counter() - t0) < 1.0, "64 MB must take well under a second (the Python loop needs ~7)"


def test_blosc_decoder_reads_frames_made_by_c_blosc(golden_dir):
    """Frames from libblosc 1.21.0 (zstd / lz4 / zlib; no-, byte- and bit-shuffle; split and
    unsplit blocks; a leftover block; the memcpyed form) through the pure-Python decoder and through
    the native frame walker in liblsrecon."""
    g = np.load(golden_dir / "blosc_frames.npz")
    names = sorted({k.rsplit(".", 1)[0] for k in g.files})
    assert len(names) >= 12
    taken = 0
    for name in names:
        frame, want = g[name + ".frame"].tobytes(), g[name + ".data"]
        out = np.empty(want.nbytes, np.uint8)
        codecs._py_blosc_decode(frame, out)
        np.testing.assert_array_equal(out.view(want.dtype), want.reshape(-1), err_msg=name)
        np.testing.assert_array_equal(codecs.blosc_decode(frame).view(want.dtype), want.reshape(-1))
        if codecs._native_lib() is not None:      # the native walker (lsr_blosc_decode_host) on the same vectors
            out = np.full(want.nbytes, 0xEE, np.uint8)
            if codecs._native_decode(frame, out):
                taken += 1
                np.testing.assert_array_equal(out.view(want.dtype), want.reshape(-1), err_msg=name + " (native)")
            else:
                assert codecs.blosc_header(frame)["flags"] & 0x4, name      # only bit-shuffled frames are declined
    assert codecs._native_lib() is None or taken >= 8


@pytest.mark.parametrize("cname", ["zstd", "zlib"])
@pytest.mark.parametrize("shuffle", [0, 1, 2])
@pytest.mark.parametrize("dtype,n", [("uint16", 70001), ("float32", 4096), ("uint8", 5), ("uint16", 0)])
def test_blosc_python_encoder_round_trips(cname, shuffle, dtype, n):
    rng = np.random.default_rng(n + shuffle)
    a = rng.integers(80, 600, n).astype(dtype)
    frame = codecs._py_blosc_encode(a.view(np.uint8), a.itemsize, cname, 1, shuffle, 0)
    h = codecs.blosc_header(frame)
    assert (h["nbytes"], h["typesize"], h["compressor"], h["cbytes"]) == (a.nbytes, a.itemsize, cname, len(frame))
    out = np.empty(a.nbytes, np.uint8)
    codecs._py_blosc_decode(frame, out)
    np.testing.assert_array_equal(out.view(dtype), a)


@pytest.fixture
def libblosc(monkeypatch):
    """A real c-blosc through ctypes when the host has one (the build image: /opt/conda/lib)."""
    for cand in (os.environ.get("LSR_LIBBLOSC"), "/opt/conda/lib/libblosc.so.1"):
        if cand and os.path.exists(cand):
            monkeypatch.setenv("LSR_LIBBLOSC", cand)
            break
    monkeypatch.setattr(codecs, "_libblosc_tried", False)
    monkeypatch.setattr(codecs, "_libblosc", None)
    if codecs._blosc_lib() is None:
        pytest.skip("no libblosc on this host")
    yield
    codecs._libblosc_tried, codecs._libblosc = False, None


def test_blosc_python_codec_agrees_with_libblosc_both_ways(libblosc):
    rng = np.random.default_rng(2)
    for dtype, n, cname, shuffle in [("uint16", 33333, "zstd", 1), ("uint16", 565, "zstd", 2),
                                     ("float32", 300001, "lz4", 1), ("float64", 1000, "zlib", 2),
                                     ("uint8", 100000, "zstd", 0)]:
        a = rng.integers(0, 600, n).astype(dtype)
        frame = codecs.blosc_encode(a, a.itemsize, cname, 1, shuffle, backend="libblosc")
        out = np.empty(a.nbytes, np.uint8)
        codecs._py_blosc_decode(frame, out)
        np.testing.assert_array_equal(out.view(dtype), a)
        if cname != "lz4":
            mine = codecs._py_blosc_encode(a.view(np.uint8), a.itemsize, cname, 1, shuffle, 0)
            np.testing.assert_array_equal(codecs.blosc_decode(mine, backend="libblosc").view(dtype), a)


def test_blosc_decode_into_a_caller_buffer_and_errors():
    a = np.arange(5000, dtype=np.uint16)
    frame = codecs._py_blosc_encode(a.view(np.uint8), 2, "zstd", 1, 1, 0)
    dest = np.zeros((50, 100), np.uint16)
    assert codecs.blosc_decode(frame, out=dest) is dest
    np.testing.assert_array_equal(dest.reshape(-1), a)
    with pytest.raises(ValueError):
        codecs.blosc_decode(frame, out=np.zeros(10, np.uint16))
    with pytest.raises(ValueError):
        codecs.blosc_decode(frame[:8])
    with pytest.raises(ValueError):
        codecs.blosc_decode(frame, out=np.zeros((100, 100), np.uint16)[:, ::2])


# ------------------------------------------------------------------ arrays

ENGINE = dict(compress="blosc-zstd", shards="volume")   # mantis_engine.py:474-481


def _plate(path, version, shape=(2, 2, 40, 12, 20), dtype="uint16", chunks=None, **kw):
    rng = np.random.default_rng(7)
    data = rng.integers(80, 600, shape).astype(dtype)
    with open_ome_zarr(path, layout="hcs", mode="w", channel_names=["BF", "GFP"], version=version,
                       prefer_iohub=False) as plate:
        pos = plate.create_position("A", "1", "fov0")
        arr = pos.create_zeros("0", shape=shape, dtype=dtype, chunks=chunks, **kw)
        for t in range(shape[0]):
            for c in range(shape[1]):
                arr.write_volume(t, c, data[t, c])
    return data


def test_engine_layout_round_trips_and_declares_what_the_engine_declares(tmp_path):
    """Zarr v3, NGFF 0.5, position key ``A/1/fov0``, ``sharding_indexed`` around ``blosc`` with
    ``cname=zstd``, z-chunk != 1: the properties ``test_mantis_integration.py:153-196`` asserts."""
    path = tmp_path / "acq.ome.zarr"
    data = _plate(path, "0.5", chunks=(1, 1, 16, 12, 20), **ENGINE)
    meta = json.loads((path / "A" / "1" / "fov0" / "0" / "zarr.json").read_text())
    assert meta["zarr_format"] == 3
    (sh,) = meta["codecs"]
    assert sh["name"] == "sharding_indexed" and sh["configuration"]["index_location"] == "end"
    inner = {c["name"]: c.get("configuration", {}) for c in sh["configuration"]["codecs"]}
    assert inner["blosc"]["cname"] == "zstd" and inner["blosc"]["typesize"] == 2
    assert [c["name"] for c in sh["configuration"]["index_codecs"]] == ["bytes", "crc32c"]
    assert meta["chunk_grid"]["configuration"]["chunk_shape"] == [1, 1, 48, 12, 20]   # shard = padded volume
    with open_ome_zarr(path, prefer_iohub=False) as plate:
        assert plate.version == "0.5"
        (key, pos), = plate.positions()
        assert key == "A/1/fov0"
        arr = pos["0"]
        assert arr.chunks == (1, 1, 16, 12, 20) and arr.shards == (1, 1, 48, 12, 20) and arr.chunks[-3] != 1
        for t in range(2):
            for c in range(2):
                np.testing.assert_array_equal(arr.read_volume(t, c), data[t, c])
        out = np.empty((40, 12, 20), np.uint16)          # decode straight into a caller (pinned) buffer
        assert arr.read_volume(1, 0, out=out) is out
        np.testing.assert_array_equal(out, data[1, 0])
    # one file per (t, c) volume: index (3 chunks x 16 B + crc) at its end
    shard = path / "A" / "1" / "fov0" / "0" / "c" / "1" / "0" / "0" / "0" / "0"
    raw = shard.read_bytes()
    index = np.frombuffer(raw[-52:-4], "<u8").reshape(3, 2)
    assert codecs.crc32c(raw[-52:-4]) == int.from_bytes(raw[-4:], "little")
    assert index[0, 0] == 0 and index[1, 0] == index[0, 1] and int(index[:, 1].sum()) == len(raw) - 52
    assert codecs.blosc_header(raw[:16])["compressor"] == "zstd"


@pytest.mark.parametrize("kw", [
    dict(compress="blosc-zstd", shards=(1, 1, 32, 12, 20)),        # two shards along z, ragged tail
    dict(compress="blosc-zstd", shards=(2, 2, 48, 16, 20)),        # a shard that holds all four (t, c) volumes
    dict(compress=None, shards=(1, 1, 48, 12, 20)),                # sharded, inner chunks raw
    dict(compress="gzip", shards=(1, 2, 16, 12, 40)),
    dict(compress="zstd"), dict(compress="blosc-zstd"), dict(compress="gzip"),   # unsharded
])
def test_sharded_and_compressed_v3_arrays_round_trip(tmp_path, kw):
    chunks = (1, 1, 16, 4, 20) if kw.get("shards") == (2, 2, 48, 16, 20) else (1, 1, 16, 12, 20)
    data = _plate(tmp_path / "p.zarr", "0.5", chunks=chunks, **kw)
    with open_ome_zarr(tmp_path / "p.zarr", prefer_iohub=False) as plate:
        arr = plate["A/1/fov0"]["0"]
        np.testing.assert_array_equal(arr[:], data)


def test_engine_layout_through_libblosc_equals_the_python_codec(tmp_path, libblosc, monkeypatch):
    """With a C library doing the frames (the fast path on a real host) the store reads back the same,
    and a store written by either codec is readable by the other."""
    assert codecs.blosc_backend() == "libblosc"
    data = _plate(tmp_path / "c.zarr", "0.5", chunks=(1, 1, 16, 12, 20), **ENGINE)
    codecs._libblosc = None                         # read it back with the pure-Python decoder
    monkeypatch.setattr(codecs, "_native", None)
    monkeypatch.setattr(codecs, "_native_tried", True)
    assert codecs.blosc_backend() == "python"
    with open_ome_zarr(tmp_path / "c.zarr", prefer_iohub=False) as plate:
        np.testing.assert_array_equal(plate["A/1/fov0"]["0"][:], data)
    data2 = _plate(tmp_path / "py.zarr", "0.5", chunks=(1, 1, 16, 12, 20), **ENGINE)
    codecs._libblosc_tried = False
    assert codecs.blosc_backend() == "libblosc"
    with open_ome_zarr(tmp_path / "py.zarr", prefer_iohub=False) as plate:
        np.testing.assert_array_equal(plate["A/1/fov0"]["0"][:], data2)


@pytest.mark.parametrize("compress", ["blosc-zstd", "zstd", "zlib"])
def test_v2_compressors_round_trip(tmp_path, compress):
    data = _plate(tmp_path / "p.zarr", "0.4", chunks=(1, 1, 16, 12, 20), compress=compress)
    meta = json.loads((tmp_path / "p.zarr" / "A" / "1" / "fov0" / "0" / ".zarray").read_text())
    assert meta["compressor"]["id"] == compress.split("-")[0]
    with open_ome_zarr(tmp_path / "p.zarr", prefer_iohub=False) as plate:
        np.testing.assert_array_equal(plate["A/1/fov0"]["0"][:], data)


def test_missing_inner_chunks_and_shards_read_as_fill(tmp_path):
    with open_ome_zarr(tmp_path / "z.zarr", layout="hcs", mode="w", version="0.5", prefer_iohub=False) as plate:
        arr = plate.create_position("A", "1", "0").create_zeros(
            "0", shape=(1, 2, 40, 12, 20), dtype="uint16", chunks=(1, 1, 16, 12, 20), compress="blosc-zstd",
            shards=(1, 2, 48, 12, 20))
        vol = np.full((40, 12, 20), 7, np.uint16)
        arr.write_volume(0, 1, vol)                      # channel 0 of the shared shard stays absent
    with open_ome_zarr(tmp_path / "z.zarr", prefer_iohub=False) as plate:
        arr = plate["A/1/0"]["0"]
        assert not arr.read_volume(0, 0).any()           # the autofocus-failed all-zero volume case
        np.testing.assert_array_equal(arr.read_volume(0, 1), vol)


def test_corrupt_shard_index_is_detected(tmp_path):
    _plate(tmp_path / "p.zarr", "0.5", shape=(1, 1, 40, 12, 20), chunks=(1, 1, 16, 12, 20), **ENGINE)
    shard = tmp_path / "p.zarr" / "A" / "1" / "fov0" / "0" / "c" / "0" / "0" / "0" / "0" / "0"
    raw = bytearray(shard.read_bytes())
    raw[-10] ^= 0x40
    shard.write_bytes(bytes(raw))
    with open_ome_zarr(tmp_path / "p.zarr", prefer_iohub=False) as plate:
        with pytest.raises(OSError, match="checksum"):
            plate["A/1/fov0"]["0"].read_volume(0, 0)
    shard.write_bytes(bytes(raw[:20]))
    with open_ome_zarr(tmp_path / "p.zarr", prefer_iohub=False) as plate:
        with pytest.raises(OSError, match="shorter"):
            plate["A/1/fov0"]["0"].read_volume(0, 0)


def test_unknown_codecs_name_the_fix(tmp_path):
    _plate(tmp_path / "p.zarr", "0.5", shape=(1, 1, 8, 4, 4), chunks=(1, 1, 8, 4, 4))
    meta_path = tmp_path / "p.zarr" / "A" / "1" / "fov0" / "0" / "zarr.json"
    meta = json.loads(meta_path.read_text())
    meta["codecs"].append({"name": "pcodec"})
    meta_path.write_text(json.dumps(meta))
    with pytest.raises(UnsupportedCodec, match="install iohub"):
        ZarrArray(meta_path.parent, "0.5", "r")
    with pytest.raises(ValueError, match="multiple"):
        ZarrArray.create(tmp_path / "bad", "0.5", (1, 1, 8, 4, 4), (1, 1, 3, 4, 4), "uint16", None, (1, 1, 8, 4, 4))
    with pytest.raises(ValueError, match="0.5"):
        ZarrArray.create(tmp_path / "bad2", "0.4", (1, 1, 8, 4, 4), (1, 1, 4, 4, 4), "uint16", None, (1, 1, 8, 4, 4))


def test_io_thread_budget_defaults_to_the_core_share_and_can_be_split(monkeypatch):
    """Volume reads / writes fan out to min(16, cores the process may use); a streamed run may split
    the share between its reader and its writer, and an explicit budget is honoured."""
    from shrimpy_amd.io import omezarr

    monkeypatch.delenv("LSR_IO_THREADS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    cores = omezarr.host_cores()
    assert 1 <= cores <= (os.cpu_count() or 1)
    assert omezarr._io_threads("read") == omezarr._io_threads("write") == min(16, cores)
    # the local ranks of one launch divide the box between them
    monkeypatch.setattr(omezarr, "host_cores", lambda: 128)
    for local, share in ((1, 16), (8, 16), (16, 8), (64, 2), (512, 1)):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(local))
        assert omezarr.rank_cores() == max(1, 128 // local)
        assert omezarr._io_threads("read") == omezarr._io_threads("write") == share
    monkeypatch.undo()
    monkeypatch.delenv("LSR_IO_THREADS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    prev = omezarr.io_thread_budget(read=3, write=5)
    try:
        assert (omezarr._io_threads("read"), omezarr._io_threads("write")) == (3, 5)
    finally:
        omezarr.io_thread_budget(**prev)
    assert omezarr._io_threads("read") == min(16, cores)
    monkeypatch.setenv("LSR_IO_THREADS", "2,7")
    assert (omezarr._io_threads("read"), omezarr._io_threads("write")) == (2, 7)


def test_a_single_shard_spreads_its_inner_chunks_over_the_thread_budget(tmp_path, monkeypatch):
    """The acquisition stores a volume as ONE shard file; its inner chunks are encoded / decoded by
    the whole thread budget (they used to be handled by one thread, one after another).  The files
    written are the same bytes whatever the thread count; errors of a worker propagate."""
    import threading

    from shrimpy_amd.io import omezarr

    monkeypatch.delenv("LSR_IO_THREADS", raising=False)
    monkeypatch.setattr(omezarr.ZarrArray, "_POOL_MIN_BYTES", 0)
    seen = set()
    real = omezarr._BlockCodec.decode

    def spy(self, *a, **k):
        seen.add(threading.current_thread().name)
        return real(self, *a, **k)

    monkeypatch.setattr(omezarr._BlockCodec, "decode", spy)
    prev = omezarr.io_thread_budget(read=4, write=4)
    try:
        data = _plate(tmp_path / "par.zarr", "0.5", shape=(1, 2, 70, 12, 20), chunks=(1, 1, 8, 12, 20), **ENGINE)
        with open_ome_zarr(tmp_path / "par.zarr", prefer_iohub=False) as plate:
            arr = plate["A/1/fov0"]["0"]
            assert arr._inner_threads(1, 1 << 20, "read") == 4 and arr._inner_threads(2, 1 << 20, "read") == 2
            out = np.empty((70, 12, 20), np.uint16)
            np.testing.assert_array_equal(arr.read_volume(0, 1, out=out), data[0, 1])
        assert len({n for n in seen if n.startswith("lsr-shard")}) > 1
        omezarr.io_thread_budget(read=1, write=1)
        data1 = _plate(tmp_path / "ser.zarr", "0.5", shape=(1, 2, 70, 12, 20), chunks=(1, 1, 8, 12, 20), **ENGINE)
        np.testing.assert_array_equal(data, data1)
        for c in range(2):
            rel = Path("A/1/fov0/0/c/0") / str(c) / "0/0/0"
            assert (tmp_path / "par.zarr" / rel).read_bytes() == (tmp_path / "ser.zarr" / rel).read_bytes()
        # a truncated shard: the failing worker's error reaches the caller
        omezarr.io_thread_budget(read=4, write=4)
        shard = tmp_path / "par.zarr" / "A/1/fov0/0/c/0/0/0/0/0"
        raw = shard.read_bytes()
        ilen = 16 * 9 + 4
        shard.write_bytes(raw[:40] + raw[-ilen:])
        with open_ome_zarr(tmp_path / "par.zarr", prefer_iohub=False) as plate:
            with pytest.raises(OSError, match="runs past the end"):
                plate["A/1/fov0"]["0"].read_volume(0, 0)
    finally:
        omezarr.io_thread_budget(**prev)


def test_native_frame_walker_agrees_with_c_blosc_and_with_the_python_codec(libblosc):
    """``lsr_blosc_decode_host`` (csrc/blosc_frame.hip) on frames made by c-blosc itself -- 32 KB blocks,
    the acquisition's kind -- and by the Python encoder, for zstd / lz4 / zlib streams, shuffled and not:
    equal to the data and to the Python walker; what it does not take falls through; corrupt frames are
    reported."""
    if codecs._native_lib() is None:
        pytest.skip("liblsrecon is not built")
    rng = np.random.default_rng(4)
    for dtype, n, cname, shuffle in [("uint16", 1 << 20, "zstd", 1), ("uint16", 33333, "zstd", 1),
                                     ("float32", 300001, "lz4", 1), ("float64", 20000, "zlib", 1),
                                     ("uint8", 100000, "zstd", 0), ("uint16", 70001, "zstd", 0)]:
        a = rng.poisson(300, n).astype(dtype)
        for maker in ("libblosc", "python"):
            if maker == "python" and cname == "lz4":
                continue
            frame = codecs.blosc_encode(a, a.itemsize, cname, 1, shuffle, backend=maker)
            out = np.zeros(a.nbytes, np.uint8)
            assert codecs._native_decode(frame, out) is True
            np.testing.assert_array_equal(out.view(dtype), a)
            assert codecs.blosc_decode(frame, backend="lsrecon").view(dtype).tolist() == a.tolist()
            ref = np.zeros(a.nbytes, np.uint8)
            codecs._py_blosc_decode(frame, ref)                        # the Python statement of the same walk
            np.testing.assert_array_equal(ref, out)
    # bit shuffle: not taken (the Python codec decodes it)
    a = rng.integers(0, 600, 4096).astype(np.uint16)
    frame = codecs.blosc_encode(a, 2, "zstd", 1, 2, backend="libblosc")
    assert codecs._native_decode(frame, np.zeros(a.nbytes, np.uint8)) is False
    np.testing.assert_array_equal(codecs.blosc_decode(frame, backend=None).view(np.uint16), a)
    # corrupt frames are reported, not dereferenced
    frame = codecs.blosc_encode(rng.poisson(300, 50000).astype(np.uint16), 2, "zstd", 1, 1, backend="libblosc")
    out = np.zeros(100000, np.uint8)
    with pytest.raises(ValueError, match="runs past the end|does not decode"):
        codecs._native_decode(frame[:len(frame) // 2], out)
    bad = bytearray(frame)
    bad[40:60] = bytes(20)
    with pytest.raises(ValueError, match="corrupt blosc frame"):
        codecs._native_decode(bytes(bad), out)
    with pytest.raises(ValueError, match="destination has"):
        codecs._native_decode(frame, np.zeros(10, np.uint8))


def test_frames_with_impossible_block_sizes_are_refused_by_both_walkers():
    """A hostile or damaged header: blocksize 0, blocksize > nbytes, and split blocks whose size is not a
    multiple of the typesize (the T streams would not cover the block: the rest of the scratch would be
    copied out undecoded).  ``ValueError`` from the Python walker and from ``lsr_blosc_decode_host``."""
    import struct

    def header(typesize, nbytes, blocksize, flags):
        return struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, 16 + 4 + 4 + nbytes)

    zstd_split = 4 << 5                      # zstd, blocks may be split (0x10 clear), no shuffle flag
    cases = [header(2, 4096, 0, zstd_split | 1), header(2, 4096, 8192, zstd_split | 1),
             header(3, 3001, 1000, zstd_split | 1)]
    for head in cases:
        nbytes = struct.unpack_from("<I", head, 4)[0]
        frame = head + struct.pack("<i", 20) + bytes(nbytes + 64)
        with pytest.raises(ValueError, match="corrupt blosc frame"):
            codecs._py_blosc_decode(frame, np.zeros(nbytes, np.uint8))
        if codecs._native_lib() is not None:
            with pytest.raises(ValueError, match="corrupt blosc frame"):
                codecs._native_decode(frame, np.zeros(nbytes, np.uint8))


def test_concurrent_writers_of_one_shared_shard_keep_each_others_volumes(tmp_path):
    """A shard that holds several (t, c) volumes is rewritten by read-modify-write: writer threads (and
    ranks) holding different volumes of it take turns, and none loses the other's chunks."""
    from concurrent.futures import ThreadPoolExecutor

    shape = (4, 2, 24, 8, 20)
    rng = np.random.default_rng(9)
    data = rng.integers(80, 600, shape).astype("uint16")
    with open_ome_zarr(tmp_path / "p.zarr", layout="hcs", mode="w", channel_names=["BF", "GFP"], version="0.5",
                       prefer_iohub=False) as plate:
        arr = plate.create_position("A", "1", "fov0").create_zeros(
            "0", shape=shape, dtype="uint16", chunks=(1, 1, 8, 8, 20), compress="blosc-zstd", shards=(4, 2, 24, 8, 20))
        for _ in range(3):                         # every (t, c) at once, three times over
            with ThreadPoolExecutor(8) as pool:
                list(pool.map(lambda tc: arr.write_volume(tc[0], tc[1], data[tc]),
                              [(t, c) for t in range(4) for c in range(2)]))
    with open_ome_zarr(tmp_path / "p.zarr", prefer_iohub=False) as plate:
        np.testing.assert_array_equal(plate["A/1/fov0"]["0"][:], data)
    assert not list((tmp_path / "p.zarr").rglob("*.partial"))
    assert not list((tmp_path / "p.zarr").rglob("*.lock"))       # nothing but chunk files among the chunks


def _shared_shard_writer(path, t, seed, rounds):
    data = np.random.default_rng(seed).integers(80, 600, (2, 24, 8, 20)).astype("uint16")
    with open_ome_zarr(path, layout="hcs", mode="a", prefer_iohub=False) as plate:
        arr = plate["A/1/fov0"]["0"]
        for _ in range(rounds):
            for c in range(2):
                arr.write_volume(t, c, data[c])


def test_processes_writing_one_shared_shard_keep_each_others_volumes(tmp_path):
    """The same between PROCESSES (ranks of one host): four of them, one timepoint each, all in one shard file."""
    import multiprocessing as mp

    shape = (4, 2, 24, 8, 20)
    with open_ome_zarr(tmp_path / "p.zarr", layout="hcs", mode="w", channel_names=["BF", "GFP"], version="0.5",
                       prefer_iohub=False) as plate:
        plate.create_position("A", "1", "fov0").create_zeros(
            "0", shape=shape, dtype="uint16", chunks=(1, 1, 8, 8, 20), compress="blosc-zstd", shards=shape)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_shared_shard_writer, args=(str(tmp_path / "p.zarr"), t, 100 + t, 3)) for t in range(4)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    with open_ome_zarr(tmp_path / "p.zarr", prefer_iohub=False) as plate:
        got = plate["A/1/fov0"]["0"][:]
    for t in range(4):
        np.testing.assert_array_equal(got[t], np.random.default_rng(100 + t).integers(80, 600, (2, 24, 8, 20)).astype("uint16"))
    leftovers = [f.name for f in (tmp_path / "p.zarr").rglob("*") if f.suffix in (".partial", ".lock")]
    assert not leftovers, leftovers


def test_a_chunk_whose_crc32c_suffix_does_not_match_is_refused(tmp_path):
    from shrimpy_amd.io.omezarr import _BlockCodec

    codec = _BlockCodec("zstd", level=1)
    codec.params["crc32c"] = True
    block = np.arange(240, dtype="uint16").reshape(3, 8, 10)
    raw = bytearray(codec.encode(block))
    np.testing.assert_array_equal(codec.decode(bytes(raw), block.shape, block.dtype), block)
    raw[5] ^= 0x40
    with pytest.raises(ValueError, match="CRC-32C"):
        codec.decode(bytes(raw), block.shape, block.dtype)
    with pytest.raises(ValueError, match="shorter"):
        codec.decode(b"ab", block.shape, block.dtype)


def test_a_damaged_stream_is_a_value_error_whichever_decoder_meets_it():
    """A flipped byte inside a zlib / zstd stream: the decoders' own exception types (``zlib.error``, a provider's)
    leave the frame walkers as ``ValueError`` (found by ``tools/fuzz_blosc.py``)."""
    from shrimpy_amd.io import codecs

    data = np.random.default_rng(2).integers(0, 4, 30000, dtype=np.uint8)
    for cname in ("zlib", "zstd"):
        frame = bytearray(codecs.blosc_encode(data, 2, cname=cname, clevel=3, shuffle=1, blocksize=4096, backend="python"))
        np.testing.assert_array_equal(codecs.blosc_decode(bytes(frame), backend="python"), data)
        refused = 0
        for pos in range(len(frame) - 40, len(frame) - 4, 3):        # inside the last block's stream
            bad = bytearray(frame)
            bad[pos] ^= 0x5A
            try:
                codecs.blosc_decode(bytes(bad), backend="python")
            except ValueError:
                refused += 1
        assert refused > 0


@pytest.mark.parametrize("dtype,n", [("float32", 1000), ("uint16", (1 << 20) + 3), ("uint8", 7), ("float32", 0),
                                     ("float32", 300000), ("float64", 5000)])
def test_native_frame_encoder_writes_the_python_encoders_frames(dtype, n):
    """``lsr_blosc_encode_host`` (csrc/blosc_frame.hip): the writer's side of the blosc-zstd output the CLI defaults to.
    Same header fields and block structure as the Python encoder (the zstd streams themselves may differ with the zstd
    library behind each), decoded back by the native walker, by the Python walker and -- where one is loadable -- by
    c-blosc itself; an incompressible array takes the memcpyed form; many threads at once."""
    from concurrent.futures import ThreadPoolExecutor

    if codecs._native_lib() is None or not codecs._native_lib().lsr_blosc_host_encoder():
        pytest.skip("liblsrecon is not built or libzstd's compressor is not loadable")
    rng = np.random.default_rng(n + 1)
    a = np.round(rng.random(n) * 100).astype(dtype)
    for shuffle in (codecs.SHUFFLE_BYTE, codecs.SHUFFLE_NONE):
        for blocksize in (0, 4096, 1000):
            nat = codecs.blosc_encode(a, a.itemsize, shuffle=shuffle, blocksize=blocksize, backend="lsrecon")
            py = codecs._py_blosc_encode(a.view(np.uint8), a.itemsize, "zstd", 1, shuffle, blocksize)
            hn, hp = codecs.blosc_header(nat), codecs.blosc_header(py)
            for key in ("version", "versionlz", "typesize", "nbytes", "blocksize", "compressor"):
                assert hn[key] == hp[key], key
            assert hn["flags"] & ~2 == hp["flags"] & ~2 and hn["cbytes"] == len(nat)
            for backend in ("lsrecon", "python", None):
                np.testing.assert_array_equal(codecs.blosc_decode(nat, backend=backend), a.view(np.uint8))
    noise = rng.integers(0, 256, 50000, dtype=np.uint8)
    frame = codecs.blosc_encode(noise, 1, backend="lsrecon")
    assert codecs.blosc_header(frame)["flags"] & 2 and len(frame) == noise.size + 16        # memcpyed
    np.testing.assert_array_equal(codecs.blosc_decode(frame, backend="python"), noise)
    if n >= 1000:
        parts = [np.ascontiguousarray(a[i::4]) for i in range(4)]
        with ThreadPoolExecutor(4) as pool:
            frames = list(pool.map(lambda p: codecs.blosc_encode(p, p.itemsize, backend="lsrecon"), parts))
        for p, f in zip(parts, frames):
            np.testing.assert_array_equal(codecs.blosc_decode(f, backend="lsrecon"), p.view(np.uint8))
    with pytest.raises(codecs.CodecUnavailable):
        codecs.blosc_encode(a, a.itemsize, shuffle=codecs.SHUFFLE_BIT, backend="lsrecon")


def test_shard_lock_table_never_hands_two_locks_to_one_directory(tmp_path):
    """the table of per-directory locks was purged of unlocked entries past 4096 -- between a thread fetching
    its Lock and acquiring it, which let a second Lock for the same directory in.  Entries are now counted while held or
    waited for and dropped by the last thread out: many threads, many directories, never two in one critical section,
    and nothing left in the table afterwards."""
    import threading
    import time

    from shrimpy_amd.io import omezarr

    dirs = [tmp_path / f"d{i}" for i in range(40)]
    for d in dirs:
        d.mkdir()
    inside = {str(d): 0 for d in dirs}
    worst = [0]
    guard = threading.Lock()

    def work(seed):
        rng = np.random.default_rng(seed)
        for _ in range(200):
            d = dirs[int(rng.integers(0, len(dirs)))]
            with omezarr._shard_lock(d / "c0"):
                with guard:
                    inside[str(d)] += 1
                    worst[0] = max(worst[0], inside[str(d)])
                time.sleep(0)
                with guard:
                    inside[str(d)] -= 1

    threads = [threading.Thread(target=work, args=(s,)) for s in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert worst[0] == 1
    assert omezarr._shard_locks == {}
