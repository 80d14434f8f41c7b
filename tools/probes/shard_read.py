"""How fast can a rank pull a volume's compressed shard (0.75 GB, one file) from tmpfs into a pinned slot?

    python tools/probes/shard_read.py [--dir /dev/shm] [--gb 0.75] [--files 8]

The streamed store-to-store run with both codecs on the device is bound by this read (0.048-0.056 s per config-4 unit =
13-15 GB/s).  Freshly written files, each read ONCE (as in the run), per variant:
  threads x pieces of one file (preadv into the pinned buffer), two files side by side, mmap + copy, readahead hints.
One JSON line per variant.
"""

from __future__ import annotations

import argparse
import json
import mmap
import os
import sys
import time

from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--gb", type=float, default=0.75)
    ap.add_argument("--files", type=int, default=6)
    args = ap.parse_args()
    import torch

    n = int(args.gb * 1e9) // 4096 * 4096
    pin = torch.cuda.is_available()
    pinned = [(torch.empty(n, dtype=torch.uint8).pin_memory() if pin else torch.empty(n, dtype=torch.uint8)).numpy() for _ in range(2)]
    plain = np.empty(n, np.uint8)
    root = Path(args.dir) / "lsr_shard_read"
    root.mkdir(exist_ok=True)
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, n, dtype=np.uint8)

    def fresh(tag, count):
        paths = []
        for k in range(count):
            p = root / f"{tag}_{k}.bin"
            with open(p, "wb") as f:
                f.write(blob)
            paths.append(p)
        return paths

    def read_pieces(path, dest, threads, pieces):
        step = -(-n // pieces) // 4096 * 4096 or n
        spans = [(o, min(step, n - o)) for o in range(0, n, step)]
        fd = os.open(path, os.O_RDONLY)
        try:
            def one(span):
                o, m = span
                got = 0
                mv = memoryview(dest)[o:o + m]
                while got < m:
                    k = os.preadv(fd, [mv[got:]], o + got)
                    if k <= 0:
                        raise OSError("short read")
                    got += k
            if threads <= 1:
                for s in spans:
                    one(s)
            else:
                with ThreadPoolExecutor(threads) as pool:
                    list(pool.map(one, spans))
        finally:
            os.close(fd)

    def timed(label, fn, count, bytes_each=n, **extra):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        print(json.dumps({"variant": label, "files": count, "s_per_file": round(dt / count, 4),
                          "GBps": round(count * bytes_each / dt / 1e9, 1), **extra}), flush=True)

    try:
        for threads, pieces in ((1, 1), (4, 64), (8, 64), (16, 64), (16, 16), (16, 256), (32, 64)):
            paths = fresh("a", args.files)
            timed(f"one file at a time, {threads} threads x {pieces} pieces -> pinned",
                  lambda: [read_pieces(p, pinned[0], threads, pieces) for p in paths], len(paths))
            for p in paths:
                p.unlink()
        paths = fresh("b", args.files)
        timed("one file at a time, 16 threads x 64 pieces -> pageable",
              lambda: [read_pieces(p, plain, 16, 64) for p in paths], len(paths))
        timed("the same files a second time", lambda: [read_pieces(p, pinned[0], 16, 64) for p in paths], len(paths))
        for p in paths:
            p.unlink()
        for each in (8, 16):
            paths = fresh("c", args.files // 2 * 2)

            def two():
                with ThreadPoolExecutor(2) as outer:
                    for i in range(0, len(paths), 2):
                        list(outer.map(lambda a: read_pieces(paths[a[0]], pinned[a[1]], each, 64), [(i, 0), (i + 1, 1)]))
            timed(f"two files side by side, {each} threads each -> pinned", two, len(paths))
            for p in paths:
                p.unlink()
        paths = fresh("d", args.files)

        def mapped():
            for p in paths:
                with open(p, "rb") as f, mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ) as m:
                    src = np.frombuffer(m, dtype=np.uint8)
                    step = n // 16
                    with ThreadPoolExecutor(16) as pool:
                        list(pool.map(lambda o: np.copyto(pinned[0][o:o + step], src[o:o + step]), range(0, step * 16, step)))
                    del src
        timed("mmap + 16 threads copying -> pinned", mapped, len(paths))
        for p in paths:
            p.unlink()
        for each in (8, 16):
            paths = fresh("f", args.files // 2 * 2)

            def mapped_one(path, dest, threads):
                with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ) as m:
                    src = np.frombuffer(m, dtype=np.uint8)
                    step = n // threads
                    with ThreadPoolExecutor(threads) as pool:
                        list(pool.map(lambda o: np.copyto(dest[o:o + step], src[o:o + step]), range(0, step * threads, step)))
                    del src

            def two_mapped():
                with ThreadPoolExecutor(2) as outer:
                    for i in range(0, len(paths), 2):
                        list(outer.map(lambda a: mapped_one(paths[a[0]], pinned[a[1]], each), [(i, 0), (i + 1, 1)]))
            timed(f"two files side by side through mappings, {each} threads each -> pinned", two_mapped, len(paths))
            for p in paths:
                p.unlink()
        # the write itself, for scale
        t0 = time.perf_counter()
        paths = fresh("e", args.files)
        dt = time.perf_counter() - t0
        print(json.dumps({"variant": "writing the files (one thread)", "files": len(paths), "s_per_file": round(dt / len(paths), 4),
                          "GBps": round(len(paths) * n / dt / 1e9, 1)}), flush=True)
    finally:
        for p in root.glob("*.bin"):
            p.unlink()
        root.rmdir()


if __name__ == "__main__":
    main()
