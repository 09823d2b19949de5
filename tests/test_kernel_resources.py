"""Register / scratch budget of the kernels of the fused step (cross-compiled for gfx950: no GPU needed).

Scratch is HBM traffic (a 16-byte spill in EN3 was 24 MB of stores per launch, a stack slot in
EN1 doubled what it writes) and the patch passes only hold three workgroups per CU at their register caps, so the
hot kernels must stay scratch-free and inside the budgets their launch bounds assume."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def table():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for line in out.stdout.splitlines()[1:]:
        f = line.split()
        if len(f) < 6:
            continue
        rows.append((" ".join(f[:-5]), dict(vgpr=int(f[-5]), sgpr=int(f[-4]), scratch=int(f[-3]), lds=int(f[-2]), waves=int(f[-1]))))
    assert len(rows) > 40
    return rows


def pick(table, prefix):
    # (the tool cuts names at 48 characters: variants that only differ in a later template argument -- EN1 with the
    #  constant quasi-static mass or a per-element one -- share a name; the bench model runs the smaller one)
    hits = [v for k, v in table if k.startswith(prefix)]
    assert hits, prefix
    return min(hits, key=lambda r: r["lds"])


@pytest.mark.parametrize("kernel,max_vgpr,min_waves,max_lds", [
    ("E2_update_stress<desk::MathPortable, 1, 1, 0, 0>", 256, 2, 20480),   # E2<GEO>, first pass of two
    ("E2_update_stress<desk::MathOcml, 1, 1, 0, 0>", 256, 2, 0),
    ("E2_update_stress<desk::MathPortable, 1, 0, 0, 0>", 168, 3, 20480),   # first step of a call
    ("E2_update_stress<desk::MathPortable, 0, 1, 0, 0>", 256, 2, 20480),   # one pass (return mapping inline): the fused step's mode
    ("E2_update_stress<desk::MathPortable, 0, 1, 7, 0>", 256, 2, 20480),   # ... with the evp law known at compile time (the headline)
    ("E2_return_mapping<desk::MathPortable, 1>", 256, 2, 20480),        # second pass
    # (the patch passes' LDS is dynamic since round 5 -- sized per mesh, engine/launch.hpp: en1_lds_bytes / en3_lds_bytes --;
    #  what is bounded here is the register budget that lets three / four 256-lane or three 512-lane workgroups share a CU)
    ("EN1_mass_temperature_dvoldt<256, 1, 1>", 128, 4, 0),
    ("EN3_force_nodes<512, 1>", 80, 6, 1024),
    ("EN2_nmd_gather<1664, 896>", 64, 8, 20480),
    ("k_s2", 256, 2, 1024),
    ("k_s3_finalize", 64, 8, 4096),
    ("E1_geom_rotate_strainrate<68>", 96, 5, 1024),                      # the compute_dt reduction of every 10th step
])
def test_hot_kernels_stay_scratch_free_and_inside_their_budgets(table, kernel, max_vgpr, min_waves, max_lds):
    r = pick(table, kernel)
    assert r["scratch"] == 0, r
    assert r["vgpr"] <= max_vgpr and r["waves"] >= min_waves and r["lds"] <= max_lds, r


@pytest.mark.parametrize("kernel", ["E2_update_stress<desk::MathPortable, 0, 1, 7, 1>", "E2_update_stress<desk::MathPortable, 0, 1, 0, 1>"])
def test_three_wave_shape_of_the_fused_stress_update(table, kernel):
    """E2<GEO> held to three waves per SIMD for launches that it saves a round of workgroups (a strong-scaling shard,
    engine/launch.hpp: e2_three_waves): 168 VGPRs at the price of a few dozen bytes of scratch per lane -- bounded here."""
    r = pick(table, kernel)
    assert r["vgpr"] <= 168 and r["waves"] >= 3 and r["scratch"] <= 48, r


def test_pipelined_stress_update_budget(table):
    """the LDS-DMA pipeline of E2<GEO> (passes/e2.hpp): two workgroups of four wavefronts per CU -- all 256 VGPRs, the private
    LDS regions of its wavefronts + the libm tables within half a CU's LDS, a few dwords of scratch at most (20 B in round 4)"""
    for kernel in ("E2_update_stress_pipe<desk::MathPortable, 7, 4>", "E2_update_stress_pipe<desk::MathPortable, 0, 4>"):
        r = pick(table, kernel)
        assert r["waves"] >= 2 and r["lds"] <= 80 * 1024 and r["scratch"] <= 32, r


def test_patch_pass_lds_follows_the_mesh():
    """en1_lds_bytes / en3_lds_bytes (restated): the headline mesh's largest block of 64 nodes (1596 incidences, 290 patch
    nodes, 870 patch elements) fits three workgroups per CU in both passes, a block of 40 nodes four"""
    cap = lambda n: (n + 7) & ~7
    en1 = lambda inc, pn, pe, constm=True: cap(pn) * 56 + cap(pe) * (24 if constm else 32) + cap(inc) * 10
    en3 = lambda inc, pn: cap(pn) * 40 + cap(inc) * 24
    assert en1(1596, 290, 870) <= 160 * 1024 // 3 and en3(1596, 290) <= 160 * 1024 // 3
    assert en1(1048, 212, 608) <= 160 * 1024 // 4 and en3(1048, 212) <= 160 * 1024 // 4


@pytest.fixture(scope="module")
def table2d():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--2d"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for line in out.stdout.splitlines()[1:]:
        f = line.split()
        if len(f) < 6:
            continue
        rows.append((" ".join(f[:-5]), dict(vgpr=int(f[-5]), sgpr=int(f[-4]), scratch=int(f[-3]), lds=int(f[-2]), waves=int(f[-1]))))
    assert len(rows) > 40
    return rows


@pytest.mark.parametrize("kernel,max_vgpr,min_waves", [
    # the 2-D step's four launches (round 5: the patch passes' time follows the workgroups a CU holds -- DESIGN.md section 2.3 --,
    # so their register budgets are pinned like the 3-D ones: two patch elements per lane at most on meshes whose largest patch allows)
    ("k2p_temp_dvoldt<1, 2>", 96, 5),
    ("k2p_force<1, 2>", 80, 6),
    ("k2_stress<desk::MathPortable, 2, 7>", 128, 4),
    ("k2_node_avg_extent", 64, 8),
])
def test_2d_step_kernels_stay_scratch_free_and_inside_their_budgets(table2d, kernel, max_vgpr, min_waves):
    r = pick(table2d, kernel)
    assert r["scratch"] == 0, r
    assert r["vgpr"] <= max_vgpr and r["waves"] >= min_waves, r
