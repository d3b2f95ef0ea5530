"""Every kernel of the library, as the compiler reports it for gfx950 (no GPU needed): no scratch -- a spilled VGPR is
HBM traffic on a path whose bound is HBM -- and at least 3 waves per SIMD.  Round 3 shipped one spilling instantiation
(the indexed Maxwell kernel, 5 VGPRs / 16 B per lane) that no profile covered; this guards every instantiation."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def rows():
    import kernel_resources

    return kernel_resources.kernel_resources()


def test_all_kernels_reported(rows):
    names = [r["name"] for r in rows]
    assert len(rows) >= 70, names
    for must in ("evaluate_kernel<2, true, false, 2, false>", "evaluate_kernel<2, true, false, 0, true>", "evaluate_kernel<3, true, true, 0, false>", "evaluate_uniaxial_kernel<1, true>", "stream_copy_kernel", "strain_kernel"):
        assert any(must in n for n in names), must
    for r in rows:
        for key in ("vgpr", "scratch", "occupancy", "vgpr_spill"):
            assert isinstance(r.get(key), int), (r["name"], key)


def test_no_scratch_no_vgpr_spills(rows):
    bad = [(r["name"], r["scratch"], r["vgpr_spill"]) for r in rows if r["scratch"] != 0 or r["vgpr_spill"] != 0]
    assert not bad, bad


def test_occupancy_at_least_three_waves_per_simd(rows):
    bad = [(r["name"], r["vgpr"], r["occupancy"]) for r in rows if r["occupancy"] < 3]
    assert not bad, bad


def test_headline_kernels_keep_four_waves(rows):
    """the streaming kernels of the BASELINE configurations are cut for 4 waves per SIMD (128 VGPRs)"""
    for law in (1, 2, 3, 4):
        for r in rows:
            if r["name"].startswith(f"void evaluate_kernel<{law}, true, false"):
                assert r["occupancy"] >= 4 and r["vgpr"] <= 128, r


def test_sgpr_spill_reloads_stay_bounded():
    """An SGPR that does not fit lives in a VGPR lane and every use of it is a VALU instruction (v_readlane).  Round 4: three more
    64-bit uniform words in the packed VonMises3D tile took that kernel's reloads from 57 to 365 and its executed VALU
    instructions up 22 % -- nothing in the register table shows it.  Static reloads of the streaming kernels, bounded."""
    import kernel_resources

    traffic = kernel_resources.sgpr_spill_traffic()
    head = [n for n in traffic if n.startswith("void evaluate_kernel<2, true, false")]
    assert len(head) == 6, sorted(traffic)  # HIST 0 / 1 / 2, each with the tangent written in full or as its 8 parameters (host tangent)
    for n in head:  # (round 5: the tangent writer exists twice -- with and without the need test -- and a tile runs one of the two)
        assert traffic[n][1] <= 200, (n, traffic[n])
    for n, (w, r) in traffic.items():
        if "evaluate_kernel<" in n and "true, false" in n:
            assert r <= 320, (n, w, r)
