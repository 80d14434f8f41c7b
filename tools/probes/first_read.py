"""Why is the first read of a freshly written chunk file slower than a repeat?  Times plain
readinto / mmap copies of one config-4 raw volume (64 chunk files of 33.5 MB) into a pinned slot."""
import json
import mmap
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

scratch = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
root = Path(tempfile.mkdtemp(prefix="lsr_fr_", dir=scratch))
nfiles, fbytes = 64, 32 * 256 * 2048 * 2
res = {}
try:
    src = np.random.default_rng(0).integers(0, 255, fbytes, dtype=np.uint8)
    dest = torch.empty(nfiles * fbytes, dtype=torch.uint8, pin_memory=True).numpy()
    dest[:] = 1
    plain = np.empty(nfiles * fbytes, dtype=np.uint8); plain[:] = 1

    def make(tag, threads=16):
        d = root / tag; d.mkdir()
        def w(i):
            with open(d / f"{i}", "wb", buffering=0) as f:
                f.write(memoryview(src))
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(w, range(nfiles)))
        return d

    def rd(d, buf, threads=16, how="readinto"):
        def r(i):
            view = memoryview(buf)[i * fbytes:(i + 1) * fbytes]
            if how == "readinto":
                with open(d / f"{i}", "rb", buffering=0) as f:
                    got = 0
                    while got < fbytes:
                        got += f.readinto(view[got:])
            elif how == "mmap":
                fd = os.open(d / f"{i}", os.O_RDONLY)
                m = mmap.mmap(fd, fbytes, prot=mmap.PROT_READ)
                np.copyto(np.frombuffer(view, dtype=np.uint8), np.frombuffer(m, dtype=np.uint8))
                del m
                os.close(fd)
            elif how == "mmap_populate":
                fd = os.open(d / f"{i}", os.O_RDONLY)
                m = mmap.mmap(fd, fbytes, flags=mmap.MAP_SHARED | mmap.MAP_POPULATE, prot=mmap.PROT_READ)
                np.copyto(np.frombuffer(view, dtype=np.uint8), np.frombuffer(m, dtype=np.uint8))
                del m
                os.close(fd)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(r, range(nfiles)))
        return round(time.perf_counter() - t0, 4)

    d = make("a"); res["readinto_pinned_x3"] = [rd(d, dest) for _ in range(3)]
    d = make("b"); res["readinto_plain_x3"] = [rd(d, plain) for _ in range(3)]
    d = make("c"); res["mmap_pinned_x3"] = [rd(d, dest, how="mmap") for _ in range(3)]
    d = make("d"); res["mmap_populate_pinned_x3"] = [rd(d, dest, how="mmap_populate") for _ in range(3)]
    d = make("e"); time.sleep(2.0); res["readinto_after_2s_x3"] = [rd(d, dest) for _ in range(3)]
    d = make("f"); res["readinto_32thr_x3"] = [rd(d, dest, threads=32) for _ in range(3)]
    d = make("g", threads=1); res["written_by_1thr_readinto_x3"] = [rd(d, dest) for _ in range(3)]
    d = make("h"); res["readinto_8thr_x3"] = [rd(d, dest, threads=8) for _ in range(3)]
    print(json.dumps(res))
finally:
    shutil.rmtree(root, ignore_errors=True)
