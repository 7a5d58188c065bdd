"""Guards on the machine code of the hot kernels (no GPU needed: hipcc cross-compiles gfx950).  Three of round 6's gains were found by
reading the ISA -- accesses that looked like LDS or cached global loads in the source and were FLAT instructions with system-scope bits
in the binary -- and a qualifier added or lost in a later edit would bring them back without failing any parity test."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    from lrge_amd import build
    out = tmp_path_factory.mktemp("isa") / "dev.s"
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = ["hipcc"] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "lrge_amd", "csrc", "lrge_hip.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z[A-Za-z0-9_]+):", text, re.M)]      # functions: kernels and out-of-line device code
    kernels = {}
    for (a, name), (b, _) in zip(labels, labels[1:] + [(len(text), "")]):
        kernels[name] = text[a:b]
    return kernels


def _of(kernels, prefix):
    hit = {k: v for k, v in kernels.items() if k.startswith(prefix)}
    assert hit, prefix
    return hit


def test_wave_sketch_counter_is_an_lds_access(device_asm):
    for name, body in _of(device_asm, "_Z13k_sketch_wave").items():
        assert "ds_read_b32" in body and "ds_write_b32" in body, name
        assert not re.search(r"flat_(load|store)", body), name          # (the counter through a generic pointer)
        assert "sc0 sc1" not in body, name


def test_chain_kernels_reread_their_records_through_plain_loads(device_asm):
    for prefix in ("_Z11k_chain_lpg", "_Z10k_chain_hw"):
        for name, body in _of(device_asm, prefix).items():
            assert not re.search(r"(flat|global)_load_dword(x2)? .*sc0 sc1", body), name


def test_sketch_kernels_keep_their_state_in_registers(device_asm):
    # the HPC state machine's queues are shift registers in VGPRs: a ring indexed at run time would live in scratch memory
    for prefix in ("_Z13k_sketch_wave", "_Z14k_sketch_count", "_Z14k_sketch_write", "_Z15k_sketch_direct"):
        for name, body in _of(device_asm, prefix).items():
            assert not re.search(r"scratch_(load|store)", body), name
