"""GPU: the upload path (lrge_hip_seqset_upload[_async], lrge_hip_host_alloc) and the per-context options.

The 2-bit pack must give the same packed image -- hence the same minimizers -- whatever memory the ASCII bases come from
(pageable host memory staged through the context's pinned buffers, pinned host memory, device memory) and whether the
call waits for the transfer or not; edge reads (empty, shorter than a word, non-ACGT bytes, lower case) included.
Reference: the bases are what mm_idx_reader_read / Aligner::map see (aligner.rs:171-185, :231-241); seq_nt4_table maps
A/a C/c G/g T/t/U/u to 0..3 and everything else to 4.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import to_arrays

pytestmark = pytest.mark.gpu


def _reads():
    rng = np.random.Generator(np.random.PCG64(5))
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(n))) for n in
            (5000, 31, 32, 33, 1, 64, 4097, 20000, 15, 16, 17, 127, 128, 129)]
    seqs += [b"", b"acgtnNRYuU-*" * 40, bytes(range(1, 256)) * 3, b"N" * 70 + seqs[0][:300], seqs[0].lower()]
    return seqs


class _DevBuf:
    """Device memory through the HIP runtime the library itself runs on (torch's bundled runtime is a different copy:
    initialised second in one process it reports no devices)."""

    def __init__(self, host):
        import ctypes as C
        # the copy liblrge_hip.so is linked against (already mapped), not the one bundled with torch
        paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l and "/torch/" not in l})
        self.hip = C.CDLL(paths[0] if paths else "libamdhip64.so")
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(max(host.size, 1))) == 0
        assert self.hip.hipMemcpy(self.ptr, C.c_void_p(host.ctypes.data), C.c_size_t(host.size), 1) == 0   # hipMemcpyHostToDevice

    def free(self):
        self.hip.hipFree(self.ptr)


@pytest.mark.parametrize("host_pack", [True, False], ids=["host-pack", "device-pack"])
def test_every_source_gives_the_same_minimizers(ctx, oracle, knobs, host_pack):
    """host_pack: sets that start in host memory are packed by the uploader thread's workers (AVX2 / scalar words) and travel
    packed (default); NO_HOST_PACK sends the ASCII and packs on the device (k_pack), as device-resident sources always do."""
    if not host_pack:
        knobs.set("NO_HOST_PACK", "1")
    seqs = _reads()
    bases, offs = to_arrays(seqs)
    ref = ctx.upload(bases, offs)                         # pageable, blocking
    x0, y0 = ref.sketch(0)
    # the oracle agrees (so the pack is right, not merely consistent)
    exp = [oracle.sketch(s, 5, 15, rid=i, is_hpc=False) for i, s in enumerate(seqs) if len(s)]
    ex = np.concatenate([e["x"] for e in exp]); ey = np.concatenate([e["y"] for e in exp])
    assert np.array_equal(x0, ex) and np.array_equal(y0, ey)
    pinned = ctx.host_alloc(bases.size)
    pinned.array[:] = bases
    dev = _DevBuf(bases)
    for src, wait in ((bases, False), (pinned, True), (pinned, False), (int(dev.ptr.value), True), (int(dev.ptr.value), False)):
        S = ctx.upload(src, offs, None, wait=wait)
        for preset in (0, 1):
            x, y = S.sketch(preset)
            if preset == 0:
                assert np.array_equal(x, x0) and np.array_equal(y, y0), (type(src).__name__, wait)
        S.free()
    # an async upload nobody consumes is drained by free(); one that is waited for explicitly stays valid
    S = ctx.upload(pinned, offs, None, wait=False)
    S.free()
    S = ctx.upload(pinned, offs, None, wait=False)
    S.wait()
    x, y = S.sketch(0)
    assert np.array_equal(x, x0)
    S.free(); ref.free(); pinned.free(); dev.free()


def test_large_pageable_upload_is_staged_in_chunks(ctx):
    """More than the 64 MB staging buffers hold: several chunks, both buffers in flight."""
    rng = np.random.Generator(np.random.PCG64(6))
    n, L = 2000, 80_000                                    # 160 MB
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * L, dtype=np.uint8)]
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    A = ctx.upload(bases, offs)
    pinned = ctx.host_alloc(bases.size); pinned.array[:] = bases
    B = ctx.upload(pinned, offs)
    xa, ya = A.sketch(0); xb, yb = B.sketch(0)
    assert len(xa) > 0.3 * n * L / 1.0 * 0.9 and np.array_equal(xa, xb) and np.array_equal(ya, yb)
    # the same set through the other pack (ASCII over PCIe, k_pack on the device) and in small host-pack chunks, two in flight
    ctx.set_option("NO_HOST_PACK", "1")
    C = ctx.upload(bases, offs); D = ctx.upload(pinned, offs, None, wait=False)
    ctx.set_option("NO_HOST_PACK", None)
    ctx.set_option("HOST_PACK_CHUNK_WORDS", str(1 << 17))
    E = ctx.upload(pinned, offs, None, wait=False); F = ctx.upload(pinned, offs, None, wait=False)
    ctx.set_option("HOST_PACK_CHUNK_WORDS", None)
    for S in (C, D, E, F):
        x, y = S.sketch(0)
        assert np.array_equal(x, xa) and np.array_equal(y, ya)
        S.free()
    A.free(); B.free(); pinned.free()


def test_options_are_per_context_and_debug_ones_never_come_from_the_environment():
    code = r'''
import numpy as np
from lrge_amd import engine, synth
g, q, t = synth.make_config("tiny_twoset")
qr, tr = engine.name_ranks(q.names, t.names)
a, b = engine.Context(0), engine.Context(0)
Qa, Ta = a.upload(q.bases, q.offsets, qr), a.upload(t.bases, t.offsets, tr)
Qb, Tb = b.upload(q.bases, q.offsets, qr), b.upload(t.bases, t.offsets, tr)
ref = engine.Index(a, Ta, 0).overlap_twoset(Qa)[0]
# LRGE_HIP_BATCH_ANCHORS came from the environment at creation: both contexts run many small batches
engine.Index(a, Ta, 0).overlap_twoset(Qa); assert a.counters()["batches"] > 3
# cleared on one context only
a.set_option("BATCH_ANCHORS", None)
engine.Index(a, Ta, 0).overlap_twoset(Qa); assert a.counters()["batches"] == 1
engine.Index(b, Tb, 0).overlap_twoset(Qb); assert b.counters()["batches"] > 3
# LRGE_HIP_DEBUG_MAX_SKIP=0 in the environment is ignored (it would change every chain); the explicit call is not
assert np.array_equal(engine.Index(b, Tb, 0).overlap_twoset(Qb)[0], ref)
b.set_option("DEBUG_MAX_SKIP", "0")
changed = engine.Index(b, Tb, 0).overlap_twoset(Qb)[0]
b.set_option("DEBUG_MAX_SKIP", None)
assert np.array_equal(engine.Index(b, Tb, 0).overlap_twoset(Qb)[0], ref)
print("OPT-OK", int(ref.sum()), int(changed.sum()))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, LRGE_HIP_BATCH_ANCHORS="20000", LRGE_HIP_DEBUG_MAX_SKIP="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0 and "OPT-OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
