#!/usr/bin/env python3
"""How fast can this box take FASTQ bytes?  (diagnostic for the end-to-end leg: tksm sequence is bound by the file system once the
device work is hidden)  pwrite of 8 GiB from T threads into ONE file at disjoint offsets -- page cache, preallocated (fallocate),
O_DIRECT -- and into T files, on the temporary directory and on /dev/shm.
usage: python tools/fs_write_probe.py [GiB]"""
import mmap
import os
import sys
import tempfile
import threading
import time

GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
PIECE = 64 << 20
buf = mmap.mmap(-1, PIECE)            # page-aligned (O_DIRECT needs it)
buf.write(b"ACGT" * (PIECE // 4))
mv = memoryview(buf)


def run(base, threads, mode, one_file=True):
    total = int(GIB * (1 << 30)) // PIECE * PIECE
    d = tempfile.mkdtemp(prefix="fsprobe_", dir=base)
    paths = [os.path.join(d, f"f{i if not one_file else 0}") for i in range(threads)]
    flags = os.O_WRONLY | os.O_CREAT | (os.O_DIRECT if mode == "direct" else 0)
    fds = []
    try:
        for p in (paths[:1] if one_file else paths):
            fd = os.open(p, flags, 0o644)
            if mode == "fallocate":
                os.posix_fallocate(fd, 0, total if one_file else total // threads)
            fds.append(fd)
        n_pieces = total // PIECE
        nxt = [0]
        lock = threading.Lock()

        def w(i):
            fd = fds[0] if one_file else fds[i]
            while True:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= n_pieces:
                    return
                off = k * PIECE if one_file else (k // threads) * PIECE
                os.pwrite(fd, mv, off)
        t0 = time.time()
        th = [threading.Thread(target=w, args=(i,)) for i in range(threads)]
        [x.start() for x in th]
        [x.join() for x in th]
        dt = time.time() - t0
        return total / dt / 1e9
    finally:
        for fd in fds:
            os.close(fd)
        for p in set(paths):
            if os.path.exists(p):
                os.remove(p)
        os.rmdir(d)


def run_mmap(base, threads, prealloc):
    """the same bytes copied into a shared mapping of the output file (page faults instead of write(): no per-file write lock)"""
    import ctypes
    total = int(GIB * (1 << 30)) // PIECE * PIECE
    d = tempfile.mkdtemp(prefix="fsprobe_", dir=base)
    path = os.path.join(d, "f0")
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
    try:
        if prealloc:
            os.posix_fallocate(fd, 0, total)
        else:
            os.ftruncate(fd, total)
        mm = mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        dst = ctypes.addressof(ctypes.c_char.from_buffer(mm))
        src = ctypes.addressof(ctypes.c_char.from_buffer(buf))
        n_pieces = total // PIECE
        nxt = [0]
        lock = threading.Lock()

        def w(i):
            while True:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= n_pieces:
                    return
                ctypes.memmove(dst + k * PIECE, src, PIECE)
        t0 = time.time()
        th = [threading.Thread(target=w, args=(i,)) for i in range(threads)]
        [x.start() for x in th]
        [x.join() for x in th]
        dt = time.time() - t0
        return total / dt / 1e9
    finally:
        os.close(fd)
        os.remove(path)
        os.rmdir(d)


for base in (tempfile.gettempdir(), "/dev/shm"):
    for prealloc in (False, True):
        for threads in (1, 3, 8, 16):
            try:
                print(f"{base:10s} mmap{'+fallocate' if prealloc else '':10s} {threads} thread(s), one file: {run_mmap(base, threads, prealloc):6.2f} GB/s", flush=True)
            except Exception as e:
                print(f"{base:10s} mmap {threads}: {e!r}", flush=True)
for base in ():
    for mode in ("cache", "fallocate", "direct"):
        for threads, one in ((1, True), (3, True), (8, True), (8, False)):
            try:
                print(f"{base:10s} {mode:9s} {threads} thread(s), {'one file' if one else 'a file each'}: {run(base, threads, mode, one):6.2f} GB/s", flush=True)
            except OSError as e:
                print(f"{base:10s} {mode:9s} {threads} thread(s): {e}", flush=True)
