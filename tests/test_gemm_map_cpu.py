"""CPU: the index maps and the work-item lists of the bf16 GEMM kernels (csrc/xq_gemm_map.hpp — the very functions the device code
calls) replayed by tests/gemm_map_emulator.cpp: one K tile through LDS-DMA image -> fragment reads -> MFMA lane layout -> epilogue
staging for every operand-layout / tile-width combination against a plain matrix product, LDS bank conflicts of every fragment read,
bijectivity of the XCD tile order, and every workgroup's item list of the persistent schedule (each (tile, K tile) exactly once; the
scalar tile walk of the persistent kernel == decode_item), and the staging address streams of its scalar cursor (retarget / step) against
the per-K-tile form base + kt * adv of the ring kernel (gm::StagerAddr is the device code's own arithmetic)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_maps_and_item_lists_replay_on_the_cpu(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++ here")
    exe = str(tmp_path / "gemm_map_emulator")
    subprocess.run([gxx, "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "gemm_map_emulator.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ALL OK"), out.stdout[-2000:]
    assert out.stdout.count(": ok;") == 6 and "item lists:" in out.stdout
    assert "bank-conflict cycles: b128 0, tr 0" in out.stdout
    assert out.stdout.count("duo-ok; bank-conflict cycles: b128 0, tr 0") == 4      # round 6: 128 x 256 tiles, uniform-delta staging, ragged edges moved back
    assert out.stdout.count("addresses compared, ok") == 10      # staging address streams: scalar cursor == base + kt * adv
