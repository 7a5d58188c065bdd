"""The counter-based read generator (tools/synth, lrge_amd/synth_cb.py): test infrastructure that makes the
H. sapiens-scale configuration buildable in seconds.  Host twin: every read is a function of its index alone; device
twin: bit-identical to the host twin."""
import numpy as np
import pytest

from lrge_amd import synth, synth_cb


def test_host_twin_reads_depend_on_their_index_only():
    spec, Q, T = synth_cb.spec_of("tiny_hifi")
    whole = spec.host_reads(first=0, n=Q + T)
    idx = np.array([Q + T - 1, 0, 17, 3, 17], dtype=np.uint64)
    sub = spec.host_reads(idx=idx)
    ws, ss = whole.seqs(), sub.seqs()
    for k, i in enumerate(idx):
        assert ss[k] == ws[int(i)]
        assert sub.names[k] == whole.names[int(i)]
    # a second universe with another seed differs
    other = synth_cb.Spec(spec.gsize, spec.seed + 1, spec.platform).host_reads(first=0, n=4)
    assert other.seqs()[0] != ws[0]


@pytest.mark.parametrize("cfg", ["tiny_hifi", "tiny_ont"])
def test_host_twin_read_model(cfg):
    """Reads are error-laden copies of their genome interval on the stated strand, lengths follow the platform."""
    spec, Q, T = synth_cb.spec_of(cfg)
    rb = spec.host_reads(first=0, n=Q + T)
    p = synth.PLATFORMS[spec.platform]
    src_len = rb.ends - rb.starts
    assert src_len.min() >= p["lo"] and src_len.max() <= min(p["hi"], spec.gsize)
    assert (rb.starts >= 0).all() and (rb.ends <= spec.gsize).all()
    assert 0 < rb.strands.mean() < 1
    err = p["sub"] + p["ins"] + p["dele"]
    tot_mis, tot = 0, 0
    for i in range(0, rb.n, 7):
        g = spec.genome(int(rb.starts[i]), int(src_len[i]))
        src = synth._COMP[g[::-1]] if rb.strands[i] else g
        s = np.frombuffer(rb.seqs()[i], dtype=np.uint8)
        # emitted length = source length + insertions - deletions
        assert abs(len(s) - len(src)) < 6 * np.sqrt(len(src) * err) + 8
        # the first 40 bases agree except for the errors among them
        m = min(40, len(s), len(src))
        tot_mis += int((s[:m] != src[:m]).sum() > 0); tot += 1
    assert tot_mis < tot * (1 - (1 - err) ** 40) * 2 + 3
    assert set(np.unique(rb.bases)) <= set(b"ACGT")


def test_length_table_matches_the_platform_distribution():
    for platform in ("hifi", "ont"):
        tab = synth_cb.length_table(platform, 10**9)
        p = synth.PLATFORMS[platform]
        assert tab.size == 65536 and (np.diff(tab.astype(np.int64)) >= 0).all()
        want = p["mu"] if p["kind"] == "normal" else np.exp(p["mu"])
        assert abs(float(np.median(tab)) - want) < 0.01 * want


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,scale", [("tiny_hifi", 1.0), ("tiny_ont", 1.0), ("c5_human_tenth", 0.1)])
def test_device_twin_is_bit_identical_to_the_host_twin(cfg, scale):
    spec, Q, T = synth_cb.spec_of(cfg, scale)
    n = Q + T
    dr = spec.device_reads(0, n)
    try:
        h = spec.host_reads(first=0, n=n)
        assert np.array_equal(h.offsets, dr.offsets)
        assert np.array_equal(h.bases, dr.to_host())
        assert np.array_equal(h.starts, dr.starts) and np.array_equal(h.ends, dr.ends) and np.array_equal(h.strands, dr.strands)
        # a range that does not start at read 0
        d2 = spec.device_reads(Q, min(T, 50))
        assert np.array_equal(d2.to_host(), h.slice(Q, Q + min(T, 50)).bases)
        d2.free()
    finally:
        dr.free()


@pytest.mark.gpu
def test_device_generated_reads_upload_like_host_reads(ctx):
    """A set generated in HBM and packed in place gives the minimizers of the same reads uploaded from the host."""
    spec, Q, T = synth_cb.spec_of("tiny_hifi")
    dr = spec.device_reads(0, Q + T)
    h = spec.host_reads(first=0, n=Q + T)
    a = ctx.upload(dr.ptr, dr.offsets, dr.name_ranks())
    b = ctx.upload(h.bases, h.offsets, dr.name_ranks())
    dr.free()
    for preset in (0, 1):
        xa, ya = a.sketch(preset); xb, yb = b.sketch(preset)
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb) and xa.size > 0
    a.free(); b.free()
