"""Register-allocation guard.  The resident / batched-big kernels keep a fit's samples in registers for its whole life;
hipcc's allocation for them is fragile (an unrelated edit once took the dlevmar_dif kernel from 6 to 161 spilled VGPRs,
a 30 % slowdown that no numerical test can see).  This test cross-compiles resident_fit.hip for gfx950 with
-Rpass-analysis=kernel-resource-usage (no GPU needed) and bounds the spills of the prepared-sample kernels."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_resident_kernels_do_not_spill_their_samples(tmp_path):
    src = os.path.join(ROOT, "brdf_amd", "csrc", "resident_fit.hip")
    # -DBRDF_DEV_WARD_ONLY: only the Ward instantiations (the benchmarked kernels; a minute instead of six for all three models,
    # whose table -- same structure, same numbers within a few registers -- is kept in profiles/r02_resident_kernel_resources.txt)
    cmd = ["hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-DBRDF_DEV_WARD_ONLY",
           "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", str(tmp_path / "r.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src), timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", out.stderr)]
    lds = [int(v) for v in re.findall(r"LDS Size \[bytes/block\]: (\d+)", out.stderr)]
    assert len(names) == len(spills) == len(lds) and len(names) >= 6
    worst = {}
    for nme, sp, l in zip(names, spills, lds):
        m = re.search(r"resident_fit_kernelILi(\d)ELi(\d)ELb(\d)ELb(\d)E", nme)
        if not m:
            continue
        assert l <= 160 * 1024, (nme, l)  # one workgroup per CU must fit the CU's LDS
        if m.group(3) == "1":  # FAST (prepared-sample) kernels: the ones every fit with positive cosines takes
            worst[nme] = sp
    assert len(worst) == 6  # Ward x (dif, bc_dif/bc_der, der) x (single fit, batched)
    assert max(worst.values()) <= 12, worst  # (today: 0-8; the batched dlevmar_bc_dif kernel keeps its control wave's samples in registers)
